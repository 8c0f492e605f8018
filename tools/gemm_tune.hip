// Phase stamps of the dense-layer GEMMs (batch 256): where a workgroup's time goes.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -DPPO_TUNE_GEMM_STAMPS tools/gemm_tune.hip -o tools/gemm_tune
//   ./tools/gemm_tune            (s_memtime ticks are 10 ns)
#include "../ppo_amd/csrc/core.hip"
#include "../ppo_amd/csrc/gemm.hip"

#include <algorithm>
#include <vector>

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

__global__ void fill_kernel(float *p, size_t n, unsigned seed)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15;
        x *= 2246822519u;
        x ^= x >> 13;
        p[i] = ((x & 0xFFFF) / 32768.0f) - 1.0f;
    }
}

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 256, F = 3872, H = 256;
    float *x, *w, *h, *dh, *gw, *gx, *ws;
    CK(hipMalloc(&x, (size_t)B * F * 4));
    CK(hipMalloc(&gx, (size_t)B * F * 4));
    CK(hipMalloc(&w, (size_t)H * F * 4));
    CK(hipMalloc(&gw, (size_t)H * F * 4));
    CK(hipMalloc(&h, (size_t)B * H * 4));
    CK(hipMalloc(&dh, (size_t)B * H * 4));
    const size_t ws_bytes = ppo_gemm_workspace_bytes(B, H, F);
    CK(hipMalloc(&ws, ws_bytes));
    fill_kernel<<<1024, 256>>>(x, (size_t)B * F, 1);
    fill_kernel<<<1024, 256>>>(w, (size_t)H * F, 2);
    fill_kernel<<<64, 256>>>(dh, (size_t)B * H, 3);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char *names[3] = {"dense fwd", "dense dW ", "dense dX "};
    for (int which = 0; which < 3; ++which) {
        auto run = [&]() {
            if (which == 0) return ppo_gemm_f32(x, F, 1, 1, w, 1, F, 0, nullptr, nullptr, h, H, B, H, F, ws, ws_bytes, nullptr);
            if (which == 1) return ppo_gemm_f32(dh, 1, H, 0, x, F, 1, 1, nullptr, nullptr, gw, F, H, F, B, nullptr, 0, nullptr);
            return ppo_gemm_f32(dh, H, 1, 0, w, F, 1, 0, nullptr, x, gx, F, B, F, H, nullptr, 0, nullptr);
        };
        for (int i = 0; i < 5; ++i) run();
        CK(hipDeviceSynchronize());
        const int reps = 50;
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) run();
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.2f us per call (events, back to back)\n", names[which], ms * 1e3 / reps);
#ifdef PPO_TUNE_GEMM_STAMPS
        std::vector<unsigned long long> st(1024 * 8);
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(ppo::ppo_gemm_stamps), st.size() * 8));
        double ph[4] = {0, 0, 0, 0};
        unsigned long long first = ~0ull, last = 0, first_end = ~0ull;
        int n = 0;
        for (int b = 0; b < 256; ++b) {
            const unsigned long long *s = &st[b * 8];
            if (s[4] == 0 || s[4] < s[0]) continue;
            for (int k = 0; k < 4; ++k) ph[k] += (double)(s[k + 1] - s[k]);
            first = std::min(first, s[0]);
            last = std::max(last, s[4]);
            first_end = std::min(first_end, s[4]);
            ++n;
        }
        if (n)
            printf("   %d workgroups: issue prologue loads %.2f us | main groups %.2f us | last group %.2f us | epilogue %.2f us"
                   " | first start -> last end %.2f us (first end after %.2f us)\n",
                   n, ph[0] / n / 100, ph[1] / n / 100, ph[2] / n / 100, ph[3] / n / 100, (double)(last - first) / 100,
                   (double)(first_end - first) / 100);
        std::vector<unsigned long long> zeros(1024 * 8, 0);
        CK(hipMemcpyToSymbol(HIP_SYMBOL(ppo::ppo_gemm_stamps), zeros.data(), zeros.size() * 8));
#endif
    }
    return 0;
}

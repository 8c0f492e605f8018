// Tuning harness for the columns-regime GAE scan: times template/launch variants
// of the production kernel (it #includes the production source) with HIP events.
//   hipcc -O3 --offload-arch=gfx950 -I include tools/scan_tune.hip -o tools/scan_tune
#include "../ppo_amd/csrc/core.hip"
#include "../ppo_amd/csrc/gae_scan.hip"

#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

__global__ void fill_kernel(float *p, size_t n, unsigned seed)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15;
        x *= 2246822519u;
        x ^= x >> 13;
        p[i] = ((x & 0xFFFF) / 32768.0f) - 1.0f;
    }
}
__global__ void fill_done(uint8_t *p, size_t n)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned x = (unsigned)i * 2654435761u + 77u;
        x ^= x >> 15;
        x *= 2246822519u;
        x ^= x >> 13;
        p[i] = (x % 100u) == 0;
    }
}

template <int U>
float time_variant(const float *r, const float *v, const uint8_t *d, float *adv, float *ret, int N, int A,
                   int64_t ld, int block, int reps)
{
    using namespace ppo;
    const int packs = A / 4;
    const int grid = (packs + block - 1) / block;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto go = [&]() {
        hipLaunchKernelGGL((gae_columns_kernel<PPO_TERM_U8, 4, U>), dim3(grid), dim3(block), 0, 0, r, v,
                           v + (size_t)N * ld, (const void *)d, adv, ret, N, packs, ld, 0.999f, 0.999 * 0.95,
                           0.999 * 0.95);
    };
    for (int i = 0; i < 3; ++i) go();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) go();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int N = 256;
    std::vector<int> As = {1 << 18, 1 << 20, (1 << 20) + 4096, 1 << 22};
    std::vector<int> pads = {0, 64, 1024};
    for (int A : As)
        for (int pad : pads) {
            const int64_t ld = A + pad;
            float *r, *v, *adv, *ret;
            uint8_t *d;
            size_t n = (size_t)N * ld;
            CK(hipMalloc(&r, n * 4));
            CK(hipMalloc(&v, (n + ld) * 4));
            CK(hipMalloc(&adv, n * 4));
            CK(hipMalloc(&ret, n * 4));
            CK(hipMalloc(&d, n));
            fill_kernel<<<2048, 256>>>(r, n, 1);
            fill_kernel<<<2048, 256>>>(v, n + ld, 2);
            fill_done<<<2048, 256>>>(d, n);
            CK(hipDeviceSynchronize());
            for (int block : {64, 128, 256}) {
                float t4 = time_variant<4>(r, v, d, adv, ret, N, A, ld, block, 10);
                float t8 = time_variant<8>(r, v, d, adv, ret, N, A, ld, block, 10);
                float t16 = time_variant<16>(r, v, d, adv, ret, N, A, ld, block, 10);
                double gb = 17.0 * N * A / 1e6;
                printf("A=%8d pad=%5d block=%3d  U4 %8.1f GB/s  U8 %8.1f GB/s  U16 %8.1f GB/s\n", A, pad, block,
                       gb / t4, gb / t8, gb / t16);
                fflush(stdout);
            }
            CK(hipFree(r));
            CK(hipFree(v));
            CK(hipFree(adv));
            CK(hipFree(ret));
            CK(hipFree(d));
        }
    return 0;
}

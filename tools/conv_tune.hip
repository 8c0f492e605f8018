// Microbenchmark of the convolution kernels at the training shapes (batch 256), through the C ABI.
// It #includes the production sources, so it always measures what ships.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include tools/conv_tune.hip -o tools/conv_tune
//   ./tools/conv_tune [reps]        (also run under rocprofv3 --pmc ... for SQ counters)
#include "../ppo_amd/csrc/core.hip"
#include "../ppo_amd/csrc/conv3x3.hip"
#include "../ppo_amd/csrc/conv3x3_wgrad.hip"

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
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15;
        x *= 2246822519u;
        x ^= x >> 13;
        p[i] = ((x & 0xFFFF) / 32768.0f) - 1.0f;
    }
}

struct Geo {
    int cin, cout, hw, in_mode;
};

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    const int B = argc > 2 ? atoi(argv[2]) : 256;  // <= 256 (buffers are sized for 256)
    std::vector<Geo> geos = {{4, 16, 84, 2}, {16, 16, 42, 1}, {16, 32, 42, 0}, {32, 32, 21, 1}, {32, 32, 11, 1}};
    float *in, *out, *w, *bias, *dw, *db, *ws;
    const size_t max_act = (size_t)B * 16 * 84 * 84;
    CK(hipMalloc(&in, max_act * 4));
    CK(hipMalloc(&out, max_act * 4));
    CK(hipMalloc(&w, 32 * 32 * 9 * 4));
    CK(hipMalloc(&bias, 32 * 4));
    CK(hipMalloc(&dw, 32 * 32 * 9 * 4));
    CK(hipMalloc(&db, 32 * 4));
    const size_t ws_bytes = ppo_conv3x3_wgrad_workspace_bytes(32, 32);
    CK(hipMalloc(&ws, ws_bytes));
    float *dw_big;  // argmax scratch of the conv + pool runs
    CK(hipMalloc(&dw_big, max_act));
    fill_kernel<<<2048, 256>>>(in, max_act, 1);
    fill_kernel<<<2048, 256>>>(out, max_act, 2);
    fill_kernel<<<64, 256>>>(w, 32 * 32 * 9, 3);
    fill_kernel<<<1, 64>>>(bias, 32, 4);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%-34s %10s %10s\n", "kernel (batch 256)", "us", "TFLOP/s");
    for (const Geo &g : geos) {
        const double flops = 2.0 * 9 * g.cin * g.cout * g.hw * g.hw * B;
        for (int which = 0; which < 5; ++which) {
            if (which == 4 && g.hw != 84) continue;  // weight gradient with the max-pool backward inside (first stack)
            if (which == 1 && g.cin <= 5) continue;  // no backward-data for the observation conv
            if (which == 3 && !(g.hw == 84 || (g.hw == 42 && g.cout == 32) )) continue;  // conv + max-pool: stack-first layers
            auto go = [&]() {
                int rc = 0;
                if (which == 0)
                    rc = ppo_conv3x3_forward_f32(in, g.in_mode, w, bias, nullptr, out, B, g.cin, g.cout, g.hw, g.hw, nullptr);
                else if (which == 1)
                    rc = ppo_conv3x3_backward_data_f32(out, w, in, nullptr, in, B, g.cin, g.cout, g.hw, g.hw, nullptr);
                else if (which == 4) {
                    int n_slabs = 0;
                    rc = ppo_conv3x3_backward_weight_slabs_pooled_f32(in, g.in_mode, out, (const uint8_t *)dw_big, ws, ws_bytes, B,
                                                                      g.cin, g.cout, g.hw, g.hw, &n_slabs, nullptr);
                } else if (which == 3)
                    rc = ppo_conv3x3_pool_forward_f32(in, g.in_mode, w, bias, out, (uint8_t *)dw_big, B, g.cin, g.cout, g.hw, g.hw,
                                                      nullptr);
                else
                    rc = ppo_conv3x3_backward_weight_f32(in, g.in_mode, out, dw, db, ws, ws_bytes, B, g.cin, g.cout,
                                                         g.hw, g.hw, 0, nullptr);
                if (rc) {
                    printf("error: %s\n", ppo_last_error());
                    exit(1);
                }
            };
            go();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) go();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps;
            char name[64];
            snprintf(name, sizeof name, "%s %d->%d @%dx%d", which == 0 ? "fwd  " : (which == 1 ? "bwd-d" : (which == 2 ? "wgrad" : (which == 3 ? "fwd+p" : "wg+pb"))),
                     g.cin, g.cout, g.hw, g.hw);
            printf("%-34s %10.1f %10.1f\n", name, us, flops / us / 1e6);
#ifdef PPO_TUNE_STAMPS
            if (which <= 4) {
                static std::vector<unsigned long long> all(8 * 8 * 1024), zeros(8 * 8 * 1024, 0);
                CK(hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(ppo::ppo_tune_stamps), all.size() * 8));
                CK(hipMemcpyToSymbol(HIP_SYMBOL(ppo::ppo_tune_stamps), zeros.data(), zeros.size() * 8));
                unsigned long long st[8] = {0};
                for (size_t i = 0; i < all.size(); ++i) st[i % 8] += all[i];
                const double n = (double)st[5];
                printf("    per item-wave cycles: barrier+stage %.0f | group prologue %.0f | K loop %.0f | compute section %.0f | item %.0f  (item-waves %.0f)\n",
                       st[0] / n, st[1] / n, st[2] / n, st[3] / n, st[4] / n, n);
                if (which == 3) {
                    size_t waves = 0;
                    for (size_t i = 6; i < all.size(); i += 8) waves += all[i] != 0;
                    printf("    conv+pool per wave: prologue %.0f | lifetime %.0f cycles (%.1f us at 2.36 GHz)   (%zu waves per launch)\n",
                           (double)st[7] / waves, (double)st[6] / waves, (double)st[6] / waves / 2360.0, waves);
                }
                if (which == 2 || which == 4)
                    printf("    wgrad per wave: prologue %.0f | fold+slab %.0f | whole kernel %.0f   (waves %d, items per wave %.1f)\n",
                           st[6] / 2048.0, st[7] / 2048.0, st[3] / 2048.0, 2048, n / 2048.0);
            }
#endif
            fflush(stdout);
        }
    }
    return 0;
}

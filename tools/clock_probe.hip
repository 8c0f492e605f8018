// Does the shader clock depend on how much of the chip a kernel occupies?  A wave spins for a fixed number of s_memtime
// ticks (the counter tools/mfma_peak.hip found to run at the shader clock under load); the wall time of the launch gives the
// counter's frequency with 1, 16 and 256 workgroups resident, and after an idle gap.
// build: hipcc -O3 --offload-arch=gfx950 tools/clock_probe.hip -o tools/clock_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
__global__ void spin(unsigned long long ticks, unsigned long long *out)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long t = t0;
    while (t - t0 < ticks) t = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = t - t0;
}
int main()
{
    unsigned long long *out;
    hipMalloc(&out, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const unsigned long long ticks = 2000000;
    for (int idle_ms : {0, 50}) {
        for (int grid : {1, 16, 256, 2048}) {
            for (int rep = 0; rep < 3; ++rep) {
                if (idle_ms) std::this_thread::sleep_for(std::chrono::milliseconds(idle_ms));
                hipEventRecord(e0);
                hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, 0, ticks, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep == 2) printf("idle %2d ms before, grid %4d: %7.1f us for %llu ticks -> %.0f MHz\n", idle_ms, grid, ms * 1e3, ticks, ticks / (ms * 1e3));
            }
        }
    }
    return 0;
}

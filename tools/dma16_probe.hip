// Probe (gfx950): does the 16-byte LDS-DMA (global_load_lds_dwordx4) tolerate 4-byte-aligned global
// addresses and 4-byte-aligned LDS bases, and what does one request cost to issue vs the 4-byte form?
// Build: hipcc -O3 --offload-arch=gfx950 tools/dma16_probe.hip -o tools/dma16_probe ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

using gptr_t = const __attribute__((address_space(1))) void *;
using lptr_t = __attribute__((address_space(3))) void *;

__global__ void probe16(const float *src, int goff, int loff, float *out)
{
    __shared__ __align__(16) float s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = -1.f;
    __syncthreads();
    // lane l: 16 bytes from src[goff + 4 l ..] to s[loff + 4 l ..]
    __builtin_amdgcn_global_load_lds((gptr_t)(src + goff + 4 * threadIdx.x), (lptr_t)(s + loff), 16, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = s[i];
}

template <int SIZE>
__global__ void issue_cost(const float *src, long long *cycles, int reps)
{
    __shared__ __align__(16) float s[64 * 4 * 8];
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (SIZE == 16)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)(r * 8 + u) * 256 + 4 * threadIdx.x),
                                                 (lptr_t)(s + u * 256), 16, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)(r * 8 + u) * 256 + threadIdx.x),
                                                 (lptr_t)(s + u * 256), 4, 0, 0);
        }
    }
    long long t1 = clock64();
    __syncthreads();
    if (threadIdx.x == 0) cycles[blockIdx.x] = (t1 - t0) + (long long)(s[5] == 12345.f);
}

int main()
{
    const int n = 1 << 20;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, n * 4);
    hipMalloc(&o, 1024 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<float> r(1024);
    int bad = 0;
    for (int goff = 0; goff < 4; ++goff)
        for (int loff = 0; loff < 4; ++loff) {
            hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, 0, d, goff, loff, o);
            if (hipDeviceSynchronize() != hipSuccess) {
                printf("goff %d loff %d: launch failed\n", goff, loff);
                return 1;
            }
            hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
            int wrong = 0;
            for (int i = 0; i < 256; ++i) wrong += r[loff + i] != (float)(goff + i);
            wrong += (loff > 0 && r[loff - 1] != -1.f) + (r[loff + 256] != -1.f);
            printf("goff %d loff %d: %s (first %g %g %g %g %g)\n", goff, loff, wrong ? "MISMATCH" : "ok", r[0], r[1], r[2],
                   r[3], r[4]);
            bad += wrong != 0;
        }
    long long *c;
    hipMalloc(&c, 8 * 64);
    long long hc[64];
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(issue_cost<4>, dim3(1), dim3(64), 0, 0, d, c, 64);
        hipDeviceSynchronize();
        hipMemcpy(hc, c, 8, hipMemcpyDeviceToHost);
        printf("size 4 : %.1f cycles per request (1 wave)\n", hc[0] / (64.0 * 8));
        hipLaunchKernelGGL(issue_cost<16>, dim3(1), dim3(64), 0, 0, d, c, 64);
        hipDeviceSynchronize();
        hipMemcpy(hc, c, 8, hipMemcpyDeviceToHost);
        printf("size 16: %.1f cycles per request (1 wave)\n", hc[0] / (64.0 * 8));
    }
    hipLaunchKernelGGL(issue_cost<16>, dim3(1), dim3(512), 0, 0, d, c, 64);
    hipDeviceSynchronize();
    hipMemcpy(hc, c, 8, hipMemcpyDeviceToHost);
    printf("size 16: %.1f cycles per request per wave (8 waves in the workgroup)\n", hc[0] / (64.0 * 8));
    hipLaunchKernelGGL(issue_cost<4>, dim3(1), dim3(512), 0, 0, d, c, 64);
    hipDeviceSynchronize();
    hipMemcpy(hc, c, 8, hipMemcpyDeviceToHost);
    printf("size 4 : %.1f cycles per request per wave (8 waves in the workgroup)\n", hc[0] / (64.0 * 8));
    printf(bad ? "RESULT: some alignment combinations fail\n" : "RESULT: all alignment combinations ok\n");
    return 0;
}

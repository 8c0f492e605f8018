// What does this part sustain on v_mfma_f32_16x16x4_f32?  A loop of independent MFMAs per wave, 2/4/8 waves
// per CU on every CU, timed with HIP events, plus s_memtime ticks per MFMA so the shader clock during the run
// can be read off.  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void mfma_loop(float *out, unsigned long long *ticks, int iters)
{
    f32x4 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int CHAINS>
void run(int waves_per_cu, int iters)
{
    const int cus = 256;
    float *out;
    unsigned long long *ticks;
    hipMalloc(&out, cus * waves_per_cu * 64 * 4);
    hipMalloc(&ticks, cus * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(cus), dim3(64 * waves_per_cu), 0, 0, out, ticks, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(cus), dim3(64 * waves_per_cu), 0, 0, out, ticks, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t[256];
    hipMemcpy(t, ticks, 256 * 8, hipMemcpyDeviceToHost);
    double tick = 0;
    for (int i = 0; i < 256; ++i) tick += t[i];
    tick /= 256;
    const double mfmas_per_wave = (double)iters * 8 * CHAINS;
    const double flops = mfmas_per_wave * 2048.0 * cus * waves_per_cu;
    printf("chains %d waves/CU %d: %.3f ms  %.1f TFLOP/s  | %.1f s_memtime ticks per MFMA per wave, s_memtime rate %.0f MHz\n",
           CHAINS, waves_per_cu, ms, flops / (ms * 1e-3) / 1e12, tick / mfmas_per_wave, tick / (ms * 1e3));
    hipFree(out);
    hipFree(ticks);
}

int main()
{
    for (int rep = 0; rep < 2; ++rep) {
        run<4>(4, 20000);
        run<4>(8, 20000);
        run<2>(8, 40000);
        run<1>(8, 80000);
        run<4>(16, 10000);
    }
    return 0;
}

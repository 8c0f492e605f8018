// What does this part sustain on v_mfma_f32_16x16x4_f32?  A loop of independent MFMAs per wave, 2/4/8 waves
// per CU on every CU, timed with HIP events, plus s_memtime ticks per MFMA so the shader clock during the run
// can be read off.  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// DATA: 0 = constant small operands (the original probe), 1 = per-lane pseudo-random operands that change every
// iteration (a multiply-free xorshift on the bit patterns keeps them finite and of order 1): does the sustained rate
// depend on what flows through the multipliers?
template <int CHAINS, int DATA>
__global__ void mfma_loop(float *out, unsigned long long *ticks, int iters)
{
    f32x4 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    unsigned ua = 0x3f000000u | ((threadIdx.x * 2654435761u + blockIdx.x * 40503u) & 0x007fffffu);
    unsigned ub = 0x3f000000u | ((threadIdx.x * 2246822519u + blockIdx.x * 9176u) & 0x007fffffu);
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < iters; ++i) {
        if (DATA) {  // two VALU-cheap updates per 8 * CHAINS MFMAs: new mantissas, sign flips, magnitude in [0.5, 1)
            ua = (ua ^ (ua << 7) ^ (ua >> 9)) & 0x807fffffu | 0x3f000000u;
            ub = (ub ^ (ub << 5) ^ (ub >> 11)) & 0x807fffffu | 0x3f000000u;
            a = __uint_as_float(ua) * 0.05f;
            b = __uint_as_float(ub);
        }
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

template <int CHAINS, int DATA>
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
    hipLaunchKernelGGL((mfma_loop<CHAINS, DATA>), dim3(cus), dim3(64 * waves_per_cu), 0, 0, out, ticks, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop<CHAINS, DATA>), dim3(cus), dim3(64 * waves_per_cu), 0, 0, out, ticks, iters);
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
    printf("%s chains %d waves/CU %d: %.3f ms  %.1f TFLOP/s  | %.1f s_memtime ticks per MFMA per wave, s_memtime rate %.0f MHz\n",
           DATA ? "random operands  " : "constant operands", CHAINS, waves_per_cu, ms, flops / (ms * 1e-3) / 1e12, tick / mfmas_per_wave, tick / (ms * 1e3));
    hipFree(out);
    hipFree(ticks);
}

int main()
{
    for (int rep = 0; rep < 2; ++rep) {
        run<4, 0>(8, 20000);
        run<4, 1>(8, 20000);
        run<4, 0>(16, 10000);
        run<4, 1>(16, 10000);
        run<4, 1>(16, 100000);  // ~0.9 s: long enough for a power limit to bite
    }
    return 0;
}

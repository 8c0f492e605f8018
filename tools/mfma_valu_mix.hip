// Do fp32 MFMAs and VALU instructions of OTHER waves on the same SIMD overlap on gfx950?  Each wave runs a loop of
// 8 x (one v_mfma_f32_16x16x4_f32 on one of 4 independent accumulators + K independent v_fma_f32 / v_max_f32 /
// DPP moves), with 1, 2, 4 waves per SIMD on every CU.  If the matrix pipe hides the VALU work the time per MFMA stays
// 32 cycles until K x 4 cycles exceed it; if they add, it is 32 + 4 K (+ issue).  Prints ns and cycles (at 2.4 GHz)
// per MFMA per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_mix.hip -o tools/mfma_valu_mix
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// KIND: 0 = v_fma_f32, 1 = v_max_f32, 2 = v_mov_b32_dpp row_shr:1, 3 = ds_read_b32 (LDS), 4 = s_nop (no VALU: issue only)
// BF16: v_mfma_f32_16x16x32_bf16 (16 cycles) instead of v_mfma_f32_16x16x4_f32 (32 cycles)
template <int K, int KIND, bool MFMA, bool BF16 = false>
__global__ void mix_loop(float *out, int iters)
{
    __shared__ float lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = a * (j + 1);
    bf16x8 ha, hb;
#pragma unroll
    for (int j = 0; j < 8; ++j) ha[j] = (__bf16)(a + j), hb[j] = (__bf16)(b - j);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MFMA && !BF16) acc[u % 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % 4], 0, 0, 0);
            if (MFMA && BF16) acc[u % 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[u % 4], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float &x = v[(u * K + k) % 8];
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a));
                if (KIND == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(a));
                if (KIND == 2) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x));
                if (KIND == 3) asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"((threadIdx.x & 63) * 4));
                if (KIND == 4) asm volatile("s_nop 3");
            }
            if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)");
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, int KIND, bool MFMA, bool BF16 = false>
double run(int waves_per_simd, int iters)
{
    const int cus = 256, wpc = 4 * waves_per_simd;
    float *out;
    hipMalloc(&out, cus * wpc * 64 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((mix_loop<K, KIND, MFMA, BF16>), dim3(cus), dim3(64 * wpc), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix_loop<K, KIND, MFMA, BF16>), dim3(cus), dim3(64 * wpc), 0, 0, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    // per SIMD: waves_per_simd waves x iters x 8 steps
    return ms * 1e6 / ((double)waves_per_simd * iters * 8);  // ns per (MFMA + K others) step per SIMD
}

template <int KIND, bool BF16 = false>
void table(const char *name)
{
    const int iters = 4000;
    printf("\n%s beside each %s: ns per step per SIMD (cycles at 2.4 GHz); [without the MFMA]\n", name,
           BF16 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_16x16x4_f32");
    printf("waves/SIMD |      K=0      |      K=1      |      K=2      |      K=4      |      K=8\n");
    for (int w : {1, 2, 4}) {
        const double t0 = run<0, KIND, true, BF16>(w, iters), t1 = run<1, KIND, true, BF16>(w, iters), t2 = run<2, KIND, true, BF16>(w, iters),
                     t4 = run<4, KIND, true, BF16>(w, iters), t8 = run<8, KIND, true, BF16>(w, iters);
        const double n1 = run<1, KIND, false>(w, iters), n2 = run<2, KIND, false>(w, iters), n4 = run<4, KIND, false>(w, iters),
                     n8 = run<8, KIND, false>(w, iters);
        printf("    %d      | %5.1f (%5.1f) | %5.1f (%5.1f) [%4.1f] | %5.1f (%5.1f) [%4.1f] | %5.1f (%5.1f) [%4.1f] | %5.1f (%5.1f) [%4.1f]\n", w, t0,
               t0 * 2.4, t1, t1 * 2.4, n1 * 2.4, t2, t2 * 2.4, n2 * 2.4, t4, t4 * 2.4, n4 * 2.4, t8, t8 * 2.4, n8 * 2.4);
    }
}

int main()
{
    table<0>("v_fma_f32");
    table<1>("v_max_f32");
    table<2>("s_nop 1 + v_mov_b32_dpp");
    table<3>("ds_read_b32 (+ wait)");
    table<4>("s_nop 3");
    table<0, true>("v_fma_f32");
    table<3, true>("ds_read_b32 (+ wait)");
    return 0;
}

// Strided f32 GEMM on the f32 MFMA for the dense layer and the heads of the
// policy/value network (reference: torch.nn.Linear at rl/models.py:84,364-366 and
// its autograd):
//
//   C[m,n] = epi( sum_k fa(A[m,k]) * fb(B[k,n]) + bias[n] )
//
// A and B are addressed through (row, col) element strides, so x @ W^T (forward),
// dY @ W (input gradient) and dY^T @ X (weight gradient) are the same kernel.
// fa / fb: optional ReLU on load (the encoder keeps pre-activations and applies
// ReLU when they are consumed); epi: optional [mask_src > 0] gate (ReLU backward).
//
// 64x64 output tile per 256-thread workgroup, 2x2 waves of 32x32, K staged through
// LDS in 16-deep slabs.  Small grids are filled with split-K:
// slices write [S][M][N] partials and gemm_finalize sums them in slice order
// (deterministic) and applies the epilogue.
#include "common.h"
#include "mfma.h"

namespace ppo {
namespace {

constexpr int BM = 64, BN = 64, BK = 16;
// Two LDS images per operand tile, chosen by which global axis is contiguous so that the
// staging writes stay (nearly) conflict-free while the MFMA reads are conflict-free:
//   k-major  [BK][PITCH_M]  (tile axis contiguous in memory)  PITCH_M = 16 mod 32
//   m-major  [BM][PITCH_K]  (k contiguous in memory)          PITCH_K =  2 mod 32
constexpr int PITCH_M = 80, PITCH_K = 34;
constexpr int TILE_WORDS = BM * PITCH_K > BK * PITCH_M ? BM * PITCH_K : BK * PITCH_M;

struct GemmArgs {
    const float *A;
    const float *B;
    float *C;            // [M, N] row-major (ldc) or partials [S][M][N]
    const float *bias;   // [N] or null
    const float *mask;   // [M, N] (ldc) or null: C = mask > 0 ? C : 0
    int M, N, K;
    int64_t a_sm, a_sk, b_sk, b_sn, ldc;
    int relu_a, relu_b;
    int k_per_slice;
    int split;
};

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p)
{
    __shared__ float s_a[TILE_WORDS];
    __shared__ float s_b[TILE_WORDS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l15 = lane & 15;
    const int g = lane >> 4;
    const int wm = (wave >> 1) * 32;
    const int wn = (wave & 1) * 32;
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * p.k_per_slice;
    const int kend = min(p.K, kbeg + p.k_per_slice);

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging maps: put consecutive threads along the unit-stride axis of each operand
    const bool a_kfast = p.a_sk == 1;
    const bool b_kfast = p.b_sk == 1;

    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;  // 0 .. 1023
            {
                const int kk = a_kfast ? (idx & 15) : (idx >> 6);
                const int mm = a_kfast ? (idx >> 4) : (idx & 63);
                const int gm = m0 + mm, gk = k0 + kk;
                float v = 0.f;
                if (gm < p.M && gk < kend) {
                    v = p.A[gm * p.a_sm + gk * p.a_sk];
                    if (p.relu_a) v = fmaxf(v, 0.f);
                }
                s_a[a_kfast ? mm * PITCH_K + kk : kk * PITCH_M + mm] = v;
            }
            {
                const int kk = b_kfast ? (idx & 15) : (idx >> 6);
                const int nn = b_kfast ? (idx >> 4) : (idx & 63);
                const int gn = n0 + nn, gk = k0 + kk;
                float v = 0.f;
                if (gn < p.N && gk < kend) {
                    v = p.B[gk * p.b_sk + gn * p.b_sn];
                    if (p.relu_b) v = fmaxf(v, 0.f);
                }
                s_b[b_kfast ? nn * PITCH_K + kk : kk * PITCH_M + nn] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[i] = a_kfast ? s_a[(wm + i * 16 + l15) * PITCH_K + ks * 4 + g]
                               : s_a[(ks * 4 + g) * PITCH_M + wm + i * 16 + l15];
#pragma unroll
            for (int j = 0; j < 2; ++j)
                b[j] = b_kfast ? s_b[(wn + j * 16 + l15) * PITCH_K + ks * 4 + g]
                               : s_b[(ks * 4 + g) * PITCH_M + wn + j * 16 + l15];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
        }
    }

    float *C = p.C;
    if (p.split > 1) C += (size_t)blockIdx.z * p.M * p.N;
    const int64_t ldc = p.split > 1 ? p.N : p.ldc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + i * 16 + g * 4 + r;
                const int n = n0 + wn + j * 16 + l15;
                if (m < p.M && n < p.N) {
                    float v = acc[i][j][r];
                    if (p.split == 1) {
                        if (p.bias) v += p.bias[n];
                        if (p.mask) v = p.mask[m * p.ldc + n] > 0.f ? v : 0.f;
                    }
                    C[m * ldc + n] = v;
                }
            }
}

__global__ __launch_bounds__(256) void gemm_finalize_kernel(const float *__restrict__ partial, int split, int M, int N,
                                                            const float *__restrict__ bias,
                                                            const float *__restrict__ mask, float *__restrict__ C,
                                                            int64_t ldc)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * N) return;
    const int m = idx / N, n = idx % N;
    float v = 0.f;
    for (int s = 0; s < split; ++s) v += partial[(size_t)s * M * N + idx];
    if (bias) v += bias[n];
    if (mask) v = mask[m * ldc + n] > 0.f ? v : 0.f;
    C[m * ldc + n] = v;
}

// out[n] = sum_m f(X[m, n])   (bias gradients)
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ X, int M, int N, int64_t ldx,
                                                     float *__restrict__ out, int accumulate)
{
    __shared__ float s[256];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;  // 4 row partitions
    float v = 0.f;
    if (n < N)
        for (int m = part; m < M; m += 4) v += X[m * ldx + n];
    s[threadIdx.x] = v;
    __syncthreads();
    if (part == 0 && n < N) {
        const float t = (s[threadIdx.x] + s[threadIdx.x + 64]) + (s[threadIdx.x + 128] + s[threadIdx.x + 192]);
        out[n] = accumulate ? out[n] + t : t;
    }
}

}  // namespace
}  // namespace ppo

extern "C" size_t ppo_gemm_workspace_bytes(int M, int N, int K)
{
    (void)K;
    return (size_t)16 * M * N * sizeof(float);  // up to 16 split-K slices
}

extern "C" int ppo_gemm_f32(const float *A, int64_t a_sm, int64_t a_sk, int relu_a, const float *B, int64_t b_sk,
                            int64_t b_sn, int relu_b, const float *bias, const float *mask, float *C, int64_t ldc,
                            int M, int N, int K, void *workspace, size_t workspace_bytes, void *stream)
{
    using namespace ppo;
    if (M < 0 || N < 0 || K < 0) return fail(PPO_E_INVALID, "ppo_gemm_f32: negative dimension");
    if (M == 0 || N == 0) return PPO_OK;
    if (!A || !B || !C) return fail(PPO_E_INVALID, "ppo_gemm_f32: null pointer");
    if (ldc < N) return fail(PPO_E_INVALID, "ppo_gemm_f32: ldc < N");
    hipStream_t st = as_stream(stream);
    const int gm = (M + BM - 1) / BM, gn = (N + BN - 1) / BN;
    // split K until the grid covers the chip (256 CUs), at least 256-deep slices
    int split = 1;
    while (gm * gn * split < 256 && split < 16 && K / (split * 2) >= 256) split *= 2;
    if (split > 1 && (!workspace || workspace_bytes < (size_t)split * M * N * sizeof(float))) split = 1;
    int kps = (K + split - 1) / split;
    kps = (kps + BK - 1) / BK * BK;
    GemmArgs p{A, B, split > 1 ? static_cast<float *>(workspace) : C, bias, mask, M, N, K, a_sm, a_sk, b_sk, b_sn,
               ldc, relu_a, relu_b, kps, split};
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(gn, gm, split), dim3(256), 0, st, p);
    int rc = check_launch("gemm_f32_kernel");
    if (rc) return rc;
    if (split > 1) {
        hipLaunchKernelGGL(gemm_finalize_kernel, dim3((M * N + 255) / 256), dim3(256), 0, st,
                           static_cast<const float *>(workspace), split, M, N, bias, mask, C, ldc);
        rc = check_launch("gemm_finalize_kernel");
    }
    return rc;
}

extern "C" int ppo_colsum_f32(const float *X, int M, int N, int64_t ldx, float *out, int accumulate, void *stream)
{
    using namespace ppo;
    if (M < 0 || N <= 0 || !X || !out) return fail(PPO_E_INVALID, "ppo_colsum_f32: bad arguments");
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64), dim3(256), 0, as_stream(stream), X, M, N, ldx, out, accumulate);
    return check_launch("colsum_kernel");
}

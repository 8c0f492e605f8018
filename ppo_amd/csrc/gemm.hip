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
// 64x64 output tile per 256-thread workgroup, 2x2 waves of 32x32, K staged through LDS in 32-deep
// slabs.  Each operand slab is fetched with two 16-byte loads per thread along whichever axis is
// contiguous in memory, one slab ahead (registers hold slab s+1 while the MFMAs run on slab s).
// Operands that are not 16-byte friendly (heads: 13 rows) take a scalar path.  Small grids are
// filled with split-K: slices write [S][M][N] partials and gemm_finalize sums them in slice order
// (deterministic) and applies the epilogue.
#include "common.h"
#include "mfma.h"

namespace ppo {
namespace {

#ifndef PPO_TUNE_GEMM_BK
#define PPO_TUNE_GEMM_BK 32
#endif
// 32-deep slabs; 64-deep ones (half the barriers) measured slower: dense forward 25.9 vs 19.3 us, dX 19.7 vs 18.5 us
constexpr int BM = 64, BN = 64, BK = PPO_TUNE_GEMM_BK;
// Two LDS images per operand tile, chosen by which global axis is contiguous:
//   k-major  [BK][PITCH_M]  (tile axis contiguous in memory)  PITCH_M = 16 mod 32: conflict-free MFMA reads,
//                                                             16-byte aligned rows for ds_write_b128
//   m-major  [BM][PITCH_K]  (k contiguous in memory)          PITCH_K =  2 mod 32: conflict-free MFMA reads
constexpr int PITCH_M = 80, PITCH_K = BK + 2;
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
    int vec_a, vec_b;    // operand may be fetched with aligned float4 loads along its contiguous axis
};

// One operand slab (64 tile rows x 32 k) -> registers -> LDS.  `tfast`: the tile axis (m or n) is the
// contiguous one, else k is.  rows = extent of the tile axis, t0/k0 origin, st/sk element strides.
constexpr int kSlabF4 = BM * BK / 4 / 256;  // float4 per thread and operand slab
struct Slab {
    float4 v[kSlabF4];
};

__device__ __forceinline__ void slab_load(Slab &s, const float *__restrict__ P, int64_t st, int64_t sk, bool kfast,
                                          bool vec, int t0, int rows, int k0, int kend, int tid)
{
#pragma unroll
    for (int e = 0; e < kSlabF4; ++e) {
        const int q = tid + e * 256;  // BM * BK / 4 float4 per slab
        int tt, kk;
        if (kfast) {
            kk = (q % (BK / 4)) * 4;  // BK / 4 float4 along k
            tt = q / (BK / 4);        // 64 rows
        } else {
            tt = (q & 15) * 4;  // 16 float4 along the tile axis
            kk = q >> 4;        // BK k
        }
        const int gt = t0 + tt, gk = k0 + kk;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vec) {
            const bool in = kfast ? (gt < rows && gk + 3 < kend) : (gk < kend && gt + 3 < rows);
            if (in) {
                r = *reinterpret_cast<const float4 *>(P + gt * st + gk * sk);
            } else {  // ragged edge: element-wise
                float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int g_t = kfast ? gt : gt + i, g_k = kfast ? gk + i : gk;
                    if (g_t < rows && g_k < kend) t[i] = P[g_t * st + g_k * sk];
                }
                r = make_float4(t[0], t[1], t[2], t[3]);
            }
        } else {
            float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int g_t = kfast ? gt : gt + i, g_k = kfast ? gk + i : gk;
                if (g_t < rows && g_k < kend) t[i] = P[g_t * st + g_k * sk];
            }
            r = make_float4(t[0], t[1], t[2], t[3]);
        }
        s.v[e] = r;
    }
}

__device__ __forceinline__ void slab_store(const Slab &s, float *__restrict__ lds, bool kfast, bool relu, int tid)
{
#pragma unroll
    for (int e = 0; e < kSlabF4; ++e) {
        const int q = tid + e * 256;
        float4 r = s.v[e];
        if (relu) r = make_float4(fmaxf(r.x, 0.f), fmaxf(r.y, 0.f), fmaxf(r.z, 0.f), fmaxf(r.w, 0.f));
        if (kfast) {
            const int kk = (q % (BK / 4)) * 4, tt = q / (BK / 4);
            float *d = lds + tt * PITCH_K + kk;  // [tile row][k]
            d[0] = r.x;
            d[1] = r.y;
            d[2] = r.z;
            d[3] = r.w;
        } else {
            const int tt = (q & 15) * 4, kk = q >> 4;
            *reinterpret_cast<float4 *>(lds + kk * PITCH_M + tt) = r;  // [k][tile row], 16-byte aligned
        }
    }
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p)
{
    __shared__ __align__(16) float s_a[TILE_WORDS];
    __shared__ __align__(16) float s_b[TILE_WORDS];
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

    const bool a_kfast = p.a_sk == 1;
    const bool b_kfast = p.b_sk == 1;
    Slab ra, rb;
    if (kbeg < kend) {
        slab_load(ra, p.A, p.a_sm, p.a_sk, a_kfast, p.vec_a, m0, p.M, kbeg, kend, tid);
        slab_load(rb, p.B, p.b_sn, p.b_sk, b_kfast, p.vec_b, n0, p.N, kbeg, kend, tid);
    }
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();  // everyone is done reading the previous slab
        slab_store(ra, s_a, a_kfast, p.relu_a, tid);
        slab_store(rb, s_b, b_kfast, p.relu_b, tid);
        __syncthreads();
        if (k0 + BK < kend) {  // fetch the next slab while the MFMAs run
            slab_load(ra, p.A, p.a_sm, p.a_sk, a_kfast, p.vec_a, m0, p.M, k0 + BK, kend, tid);
            slab_load(rb, p.B, p.b_sn, p.b_sk, b_kfast, p.vec_b, n0, p.N, k0 + BK, kend, tid);
        }
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
    // slices are added in slice order (deterministic); eight loads are in flight at a time (the rolled loop issued one
    // load per trip and waited for it: 10.8 us for 32 slices of a 256 x 256 product)
    float v = 0.f;
    const size_t stride = (size_t)M * N;
    int s = 0;
    for (; s + 8 <= split; s += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = partial[(size_t)(s + u) * stride + idx];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; s < split; ++s) v += partial[(size_t)s * stride + idx];
    if (bias) v += bias[n];
    if (mask) v = mask[m * ldc + n] > 0.f ? v : 0.f;
    C[m * ldc + n] = v;
}

// Skinny products (the fused policy / value / advantage heads: 13 output columns, or 13 rows, or K = 13):
// a 64x64 MFMA tile would be mostly padding and only 4 workgroups would run, each a long latency-bound K
// chain (measured 15 us for 0.85 MFLOP).  Two CUDA-core style kernels instead, both with every operand load
// of a thread issued before the first use and a fixed summation order:
//  * gemm_rows_kernel   N <= 32, both operands contiguous along k (x @ W^T): one wave per output row; lanes
//                       stride k, so A's row is read once, coalesced, and each of the N dot products is a
//                       wave reduction.
//  * gemm_small_kernel  anything else small: 16 k-partitions per output element, 16 outputs per workgroup
//                       (consecutive along n: coalesced B rows), partials combined through LDS in partition order.
constexpr int kRowsMaxN = 32, kRowsMaxKPerLane = 8;

__global__ __launch_bounds__(256) void gemm_rows_kernel(GemmArgs p)
{
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= p.M) return;
    const float *a = p.A + m * p.a_sm;
    const int kpl = (p.K + 63) / 64;  // <= kRowsMaxKPerLane
    float av[kRowsMaxKPerLane];
#pragma unroll
    for (int i = 0; i < kRowsMaxKPerLane; ++i) {
        const int k = lane + 64 * i;
        float v = (i < kpl && k < p.K) ? a[k] : 0.f;
        av[i] = p.relu_a ? fmaxf(v, 0.f) : v;
    }
    // 4 output columns per pass: their 4 * kpl loads are all issued before the first use
    for (int n0 = 0; n0 < p.N; n0 += 4) {
        float bv[4][kRowsMaxKPerLane];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float *b = p.B + min(n0 + j, p.N - 1) * p.b_sn;
#pragma unroll
            for (int i = 0; i < kRowsMaxKPerLane; ++i) {
                const int k = lane + 64 * i;
                bv[j][i] = (i < kpl && k < p.K) ? b[k] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < kRowsMaxKPerLane; ++i) {
                float x = bv[j][i];
                if (p.relu_b) x = fmaxf(x, 0.f);
                acc = fmaf(av[i], x, acc);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            const int n = n0 + j;
            if (lane == 0 && n < p.N) {
                float v = acc;
                if (p.bias) v += p.bias[n];
                if (p.mask) v = p.mask[m * p.ldc + n] > 0.f ? v : 0.f;
                p.C[m * p.ldc + n] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void gemm_small_kernel(GemmArgs p)
{
    __shared__ float s[256];
    const int o = blockIdx.x * 16 + (threadIdx.x & 15);
    const int part = threadIdx.x >> 4;
    const bool live = o < p.M * p.N;
    const int m = live ? o / p.N : 0, n = live ? o - m * p.N : 0;
    const float *a = p.A + m * p.a_sm;
    const float *b = p.B + n * p.b_sn;
    const int per = (p.K + 15) / 16;
    const int k0 = part * per, k1 = min(p.K, k0 + per);
    float acc = 0.f;
    if (live) {
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            float av[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                av[u] = a[(k + u) * p.a_sk];
                bv[u] = b[(k + u) * p.b_sk];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (p.relu_a) av[u] = fmaxf(av[u], 0.f);
                if (p.relu_b) bv[u] = fmaxf(bv[u], 0.f);
                acc = fmaf(av[u], bv[u], acc);
            }
        }
        for (; k < k1; ++k) {
            float a0 = a[k * p.a_sk], b0 = b[k * p.b_sk];
            if (p.relu_a) a0 = fmaxf(a0, 0.f);
            if (p.relu_b) b0 = fmaxf(b0, 0.f);
            acc = fmaf(a0, b0, acc);
        }
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    if (part == 0 && live) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += s[q * 16 + threadIdx.x];
        if (p.bias) v += p.bias[n];
        if (p.mask) v = p.mask[m * p.ldc + n] > 0.f ? v : 0.f;
        p.C[m * p.ldc + n] = v;
    }
}

// out[n] = sum_m X[m, n]   (bias gradients): 16 row partitions per column, combined in LDS in a fixed order
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ X, int M, int N, int64_t ldx,
                                                     float *__restrict__ out, int accumulate)
{
    __shared__ float s[256];
    const int n = blockIdx.x * 16 + (threadIdx.x & 15);
    const int part = threadIdx.x >> 4;  // 16 row partitions
    float v = 0.f;
    if (n < N)
        for (int m = part; m < M; m += 16) v += X[m * ldx + n];
    s[threadIdx.x] = v;
    __syncthreads();
    if (part == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += s[q * 16 + threadIdx.x];
        out[n] = accumulate ? out[n] + t : t;
    }
}

inline bool vec_ok(const float *P, int64_t s_tile, int64_t s_k)
{
    // float4 along the contiguous axis: the other stride and the base must keep 16-byte alignment
    const int64_t other = s_k == 1 ? s_tile : s_k;
    return (s_k == 1 || s_tile == 1) && other % 4 == 0 && aligned(P, 16);
}

}  // namespace
}  // namespace ppo

extern "C" size_t ppo_gemm_workspace_bytes(int M, int N, int K)
{
    (void)K;
    return (size_t)32 * M * N * sizeof(float);  // up to 32 split-K slices
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
    // The choice of kernel must not depend on the batch size, or a rollout forward (batch A) and a training
    // forward (batch 256) of the same sample would round differently.  N and K of a forward product are model
    // dimensions; M is the batch unless A is read transposed (a_sm == 1: a weight-gradient product dY^T X,
    // where M is a model dimension and K the batch).
    if ((N <= 16 || K <= 16 || (M <= 16 && a_sm == 1)) && (int64_t)M * N <= (1 << 22)) {
        GemmArgs q{A, B, C, bias, mask, M, N, K, a_sm, a_sk, b_sk, b_sn, ldc, relu_a, relu_b, K, 1, 0, 0};
        if (N <= kRowsMaxN && a_sk == 1 && b_sk == 1 && K <= 64 * kRowsMaxKPerLane) {
            hipLaunchKernelGGL(gemm_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, st, q);
            return check_launch("gemm_rows_kernel");
        }
        hipLaunchKernelGGL(gemm_small_kernel, dim3((M * N + 15) / 16), dim3(256), 0, st, q);
        return check_launch("gemm_small_kernel");
    }
    const int gm = (M + BM - 1) / BM, gn = (N + BN - 1) / BN;
    // split K until two workgroups per CU are in flight (one 4-wave workgroup per CU cannot hide the operand
    // loads of its own K chain), slices at least 64 deep
    int split = 1;
    while (gm * gn * split < 512 && split < 32 && K / (split * 2) >= 64) split *= 2;
    if (split > 1 && (!workspace || workspace_bytes < (size_t)split * M * N * sizeof(float))) split = 1;
    int kps = (K + split - 1) / split;
    kps = (kps + BK - 1) / BK * BK;
    GemmArgs p{A, B, split > 1 ? static_cast<float *>(workspace) : C, bias, mask, M, N, K, a_sm, a_sk, b_sk, b_sn,
               ldc, relu_a, relu_b, kps, split, vec_ok(A, a_sm, a_sk) ? 1 : 0, vec_ok(B, b_sn, b_sk) ? 1 : 0};
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
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 15) / 16), dim3(256), 0, as_stream(stream), X, M, N, ldx, out, accumulate);
    return check_launch("colsum_kernel");
}

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
// 64x64 output tile per 512-thread workgroup, 2x4 waves of 32x16, K staged through LDS in 32-deep
// slabs.  Each operand slab is fetched with two 16-byte loads per thread along whichever axis is
// contiguous in memory, one slab ahead (registers hold slab s+1 while the MFMAs run on slab s).
// Operands that are not 16-byte friendly (heads: 13 rows) take a scalar path.  Small grids are
// filled with split-K: slices write [S][M][N] partials and gemm_finalize sums them in slice order
// (deterministic) and applies the epilogue.
#include <algorithm>

#include "common.h"
#include "loss_rows.h"
#include "policy_act.h"
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
    int vec_c;           // C (and bias, mask) can move as aligned float4 along n
};

// One operand slab (64 tile rows x 32 k) -> registers -> LDS.  `tfast`: the tile axis (m or n) is the
// contiguous one, else k is.  rows = extent of the tile axis, t0/k0 origin, st/sk element strides.
#ifndef PPO_TUNE_GEMM_WAVES
#define PPO_TUNE_GEMM_WAVES 8
#endif
// 8 waves per workgroup = two per SIMD: one wave's LDS write -> barrier -> read turnaround runs under the other's MFMAs
// (with 4 waves a slab took 1890 cycles for 1024 of MFMA work); wave tile 32 x 16 (4 waves: 32 x 32)
constexpr int kGemmWaves = PPO_TUNE_GEMM_WAVES, kGemmThreads = kGemmWaves * 64;
static_assert(kGemmWaves == 4 || kGemmWaves == 8, "2 x 2 waves of 32 x 32 or 2 x 4 waves of 32 x 16");
constexpr int kWaveN = kGemmWaves == 8 ? 16 : 32, kNJ = kWaveN / 16;  // columns per wave, 16-column MFMA tiles per wave
constexpr int kSlabF4 = BM * BK / 4 / kGemmThreads;  // float4 per thread and operand slab
struct Slab {
    float4 v[kSlabF4];
};

// Operands are read through buffer descriptors (common.h: a predicated global load serialises the whole prefetch,
// 1.5 us per slab): an element outside the matrix gets the offset kOutside, which the hardware range check turns into
// zeros.  Offsets are 32-bit byte offsets: the host checks the operands are below kBufferBytes.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t operand_rsrc(const float *P) { return buffer_of(P); }

template <bool kfast, bool vec>
__device__ __forceinline__ void slab_load(Slab &s, __amdgpu_buffer_rsrc_t P, int st, int sk, int t0, int rows, int k0,
                                          int kend, int tid)
{
#pragma unroll
    for (int e = 0; e < kSlabF4; ++e) {
        const int q = tid + e * kGemmThreads;  // BM * BK / 4 float4 per slab
        int tt, kk;
        if (kfast) {
            kk = (q % (BK / 4)) * 4;  // BK / 4 float4 along k
            tt = q / (BK / 4);        // 64 rows
        } else {
            tt = (q & 15) * 4;  // 16 float4 along the tile axis
            kk = q >> 4;        // BK k
        }
        const int gt = t0 + tt, gk = k0 + kk;
        if (vec) {
            // the host picks `vec` only when the extent along the contiguous axis is a multiple of 4, so a float4
            // is either wholly inside or wholly outside
            const int off = (gt < rows && gk < kend) ? (gt * st + gk * sk) * 4 : kOutside;
            s.v[e] = buffer_f32x4(P, off);
        } else {
            float t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int g_t = kfast ? gt : gt + i, g_k = kfast ? gk + i : gk;
                const int off = (g_t < rows && g_k < kend) ? (g_t * st + g_k * sk) * 4 : kOutside;
                t[i] = buffer_f32(P, off);
            }
            s.v[e] = make_float4(t[0], t[1], t[2], t[3]);
        }
    }
}

__device__ __forceinline__ void slab_store(const Slab &s, float *__restrict__ lds, bool kfast, bool relu, int tid)
{
#pragma unroll
    for (int e = 0; e < kSlabF4; ++e) {
        const int q = tid + e * kGemmThreads;
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

// Workgroup -> (n tile, m tile, K slice).  Consecutive workgroup ids land on consecutive XCDs (8 of them, each
// with its own L2), so the id is read as (xcd, slot) and everything that shares operand bytes gets the same xcd:
//  * split-K products (dense forward: 16 tiles x 16 slices of x and W): an XCD owns whole K slices, so each byte
//    of both operands crosses the fabric once instead of once per XCD that holds a tile of its row / column block;
//  * unsplit products (dense dW / dX: 4 x 61 tiles, the 4 MB operand indexed by the long axis): the tiles along
//    the short axis, which all read the same block of the big operand, share an XCD.
// The grid is padded to a multiple of 8 per group; padded ids return at once.
constexpr int kXcds = 8;
struct TileId {
    int tm, tn, z;
    bool live;
};
__device__ __forceinline__ TileId tile_of(int id, int gm, int gn, int split)
{
    const int xcd = id % kXcds, slot = id / kXcds;
    TileId t;
    if (split > 1) {
        const int per = (split + kXcds - 1) / kXcds;  // K slices per XCD
        t.z = xcd * per + slot % per;
        const int tile = slot / per;
        t.tn = tile % gn;
        t.tm = tile / gn;
        t.live = t.z < split && t.tm < gm;
    } else if (gn >= gm) {
        t.z = 0;
        t.tm = slot % gm;
        t.tn = xcd + kXcds * (slot / gm);
        t.live = t.tn < gn;
    } else {
        t.z = 0;
        t.tn = slot % gn;
        t.tm = xcd + kXcds * (slot / gn);
        t.live = t.tm < gm;
    }
    return t;
}
inline int tile_grid(int gm, int gn, int split)
{
    if (split > 1) return kXcds * ((split + kXcds - 1) / kXcds) * gm * gn;
    const int lo = gn >= gm ? gm : gn, hi = gn >= gm ? gn : gm;
    return kXcds * ((hi + kXcds - 1) / kXcds) * lo;
}

// DEPTH operand slabs are in flight in registers per thread (the loads of slab s + DEPTH are issued when slab s
// is consumed): with one slab ahead a workgroup's K chain ran at the load latency, 1.85 us per 32-deep slab against
// 0.43 us of MFMA work.  The LDS tile is double-buffered, so a slab costs one barrier.
#ifndef PPO_TUNE_GEMM_DEPTH
#define PPO_TUNE_GEMM_DEPTH 4
#endif
constexpr int DEPTH = PPO_TUNE_GEMM_DEPTH;

// Diagnostic build only (tools/gemm_tune -DPPO_TUNE_GEMM_STAMPS): s_memtime at the phase boundaries of wave 0 of
// every workgroup, into a buffer nothing else reads.
#ifdef PPO_TUNE_GEMM_STAMPS
__device__ unsigned long long ppo_gemm_stamps[1024 * 8];
#define PPO_GSTAMP(slot)                                                                          \
    do {                                                                                          \
        unsigned long long t_;                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if (threadIdx.x == 0 && blockIdx.x < 1024) ppo_gemm_stamps[blockIdx.x * 8 + (slot)] = t_; \
    } while (0)
#else
#define PPO_GSTAMP(slot)
#endif
static_assert(DEPTH % 2 == 0, "the LDS buffer of a slab is its slot's parity");

// a_kfast / b_kfast: which axis of each operand is contiguous in memory (k, or the tile axis); vec: both operands
// can be fetched with aligned float4 loads
template <bool a_kfast, bool b_kfast, bool vec>
__global__ __launch_bounds__(kGemmThreads) void gemm_f32_kernel(GemmArgs p, int gm, int gn)
{
    __shared__ __align__(16) float s_a[2][TILE_WORDS];
    __shared__ __align__(16) float s_b[2][TILE_WORDS];
    const TileId t = tile_of(blockIdx.x, gm, gn, p.split);
    if (!t.live) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l15 = lane & 15;
    const int g = lane >> 4;
    const int wm = (wave / (BN / kWaveN)) * 32;
    const int wn = (wave % (BN / kWaveN)) * kWaveN;
    const int m0 = t.tm * BM;
    const int n0 = t.tn * BN;
    const int kbeg = t.z * p.k_per_slice;
    const int kend = min(p.K, kbeg + p.k_per_slice);

    f32x4 acc[2][kNJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < kNJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // The K slice is walked in groups of DEPTH slabs, rounded up: slabs past kend read zeros through the range
    // check (no memory traffic, a few idle MFMAs), which keeps every load of the main loop unconditional - with a
    // load that may or may not have been issued the compiler has to wait as if it had not, i.e. for almost everything.
    const int nslabs = kbeg < kend ? (kend - kbeg + BK - 1) / BK : 0;
    const int ngroups = (nslabs + DEPTH - 1) / DEPTH;
    const __amdgpu_buffer_rsrc_t bufA = operand_rsrc(p.A), bufB = operand_rsrc(p.B);
    Slab ra[DEPTH], rb[DEPTH];

    auto fetch = [&](int d, int slab) {
        slab_load<a_kfast, vec>(ra[d], bufA, (int)p.a_sm, (int)p.a_sk, m0, p.M, kbeg + slab * BK, kend, tid);
        slab_load<b_kfast, vec>(rb[d], bufB, (int)p.b_sn, (int)p.b_sk, n0, p.N, kbeg + slab * BK, kend, tid);
    };
    auto stage = [&](int d) {
        // buffer d & 1 was last read two slabs ago; every wave finished that before the previous slab's barrier
        slab_store(ra[d], s_a[d & 1], a_kfast, p.relu_a, tid);
        slab_store(rb[d], s_b[d & 1], b_kfast, p.relu_b, tid);
        __syncthreads();
    };
    auto multiply = [&](int d) {
        // every operand read of the slab is issued before its first MFMA (32 registers): read-then-multiply per K
        // step left the MFMA pipe idle for an LDS round trip in each of the 8 steps (one wave per SIMD: nobody
        // else fills the gap)
        const float *sa = s_a[d & 1], *sb = s_b[d & 1];
        float a[BK / 4][2], b[BK / 4][kNJ];
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[ks][i] = a_kfast ? sa[(wm + i * 16 + l15) * PITCH_K + ks * 4 + g]
                                   : sa[(ks * 4 + g) * PITCH_M + wm + i * 16 + l15];
#pragma unroll
            for (int j = 0; j < kNJ; ++j)
                b[ks][j] = b_kfast ? sb[(wn + j * 16 + l15) * PITCH_K + ks * 4 + g]
                                   : sb[(ks * 4 + g) * PITCH_M + wn + j * 16 + l15];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < kNJ; ++j) acc[i][j] = mfma16(a[ks][i], b[ks][j], acc[i][j]);
    };

    PPO_GSTAMP(0);
    if (ngroups > 0) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) fetch(d, d);
        PPO_GSTAMP(1);
        for (int grp = 1; grp < ngroups; ++grp) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                stage(d);
                fetch(d, grp * DEPTH + d);
                multiply(d);
            }
        }
        PPO_GSTAMP(2);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {  // last group: nothing left to fetch
            stage(d);
            multiply(d);
        }
    }
    PPO_GSTAMP(3);

    float *C = p.C;
    if (p.split > 1) C += (size_t)t.z * p.M * p.N;
    const int64_t ldc = p.split > 1 ? p.N : p.ldc;
    const bool finish = p.split == 1;  // bias and gate belong to whoever writes the final value
    if (p.vec_c) {
        // The accumulators go through LDS once so that a thread owns four consecutive columns: bias, gate and the
        // result move as float4 (4 + 4 + 4 memory instructions per thread instead of 16 + 16 + 16 dword ones, each
        // of which costs the CU's one memory pipe about as much as a float4 does).
        constexpr int PITCH_C = BN + 4;
        static_assert(BM * PITCH_C <= 2 * TILE_WORDS, "the output tile is staged in the A operand's buffers");
        float *sc = &s_a[0][0];
        const int row0 = tid >> 4, col = (tid & 15) * 4;
        constexpr int kRowsPerPass = kGemmThreads / 16, kPasses = BM / kRowsPerPass;
        float4 gate[kPasses], bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool col_in = n0 + col < p.N;  // N is a multiple of 4 here
        if (finish && p.mask) {
            const __amdgpu_buffer_rsrc_t bufM = operand_rsrc(p.mask);
#pragma unroll
            for (int q = 0; q < kPasses; ++q) {
                const int m = m0 + row0 + kRowsPerPass * q;
                const int off = (m < p.M && col_in) ? (m * (int)p.ldc + n0 + col) * 4 : kOutside;
                gate[q] = buffer_f32x4(bufM, off);
            }
        }
        if (finish && p.bias && col_in) bias4 = *reinterpret_cast<const float4 *>(p.bias + n0 + col);
        __syncthreads();  // every wave is done with the last slab
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < kNJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sc[(wm + i * 16 + g * 4 + r) * PITCH_C + wn + j * 16 + l15] = acc[i][j][r];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kPasses; ++q) {
            const int row = row0 + kRowsPerPass * q, m = m0 + row;
            float4 v = *reinterpret_cast<const float4 *>(sc + row * PITCH_C + col);
            if (finish) {
                v.x += bias4.x, v.y += bias4.y, v.z += bias4.z, v.w += bias4.w;
                if (p.mask) {
                    v.x = gate[q].x > 0.f ? v.x : 0.f;
                    v.y = gate[q].y > 0.f ? v.y : 0.f;
                    v.z = gate[q].z > 0.f ? v.z : 0.f;
                    v.w = gate[q].w > 0.f ? v.w : 0.f;
                }
            }
            if (m < p.M && col_in) *reinterpret_cast<float4 *>(C + m * ldc + n0 + col) = v;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < kNJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm + i * 16 + g * 4 + r;
                    const int n = n0 + wn + j * 16 + l15;
                    if (m < p.M && n < p.N) {
                        float v = acc[i][j][r];
                        if (finish) {
                            if (p.bias) v += p.bias[n];
                            if (p.mask) v = p.mask[m * p.ldc + n] > 0.f ? v : 0.f;
                        }
                        C[m * ldc + n] = v;
                    }
                }
    }
    PPO_GSTAMP(4);
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

// NB output columns per pass: the NB * KPL operand loads of a pass are all issued before the first use (with the
// heads' 13 columns and K = 256 that is ONE pass of 64 loads per lane; the rolled 4-column form paid four
// dependent load round trips and measured 14 us for 0.85 MFLOP).  The order of the additions does not depend on NB.
template <int NB, int KPL>
__global__ __launch_bounds__(256) void gemm_rows_kernel(GemmArgs p)
{
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (m >= p.M) return;
    // range-checked buffer loads (k >= K reads zero): no branch around any load
    const __amdgpu_buffer_rsrc_t bufA = operand_rsrc(p.A), bufB = operand_rsrc(p.B);
    bool kin[KPL];
#pragma unroll
    for (int i = 0; i < KPL; ++i) kin[i] = lane + 64 * i < p.K;
    float av[KPL];
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        const int off = kin[i] ? (m * (int)p.a_sm + lane + 64 * i) * 4 : kOutside;
        av[i] = buffer_f32(bufA, off);  // ReLU at use
    }
    for (int n0 = 0; n0 < p.N; n0 += NB) {
        float bv[NB][KPL];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int row = min(n0 + j, p.N - 1) * (int)p.b_sn;
#pragma unroll
            for (int i = 0; i < KPL; ++i) {
                const int off = kin[i] ? (row + lane + 64 * i) * 4 : kOutside;
                bv[j][i] = buffer_f32(bufB, off);
            }
        }
        // keeps the (loop-invariant) ReLU of A's row below the loads of B: hoisted above them it made the wave wait
        // for A's row before the first load of B was issued
#pragma unroll
        for (int i = 0; i < KPL; ++i) asm volatile("" : "+v"(av[i]));
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < KPL; ++i) {
                float x = bv[j][i];
                if (p.relu_b) x = fmaxf(x, 0.f);
                acc = fmaf(p.relu_a ? fmaxf(av[i], 0.f) : av[i], x, acc);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            bv[j][0] = acc;
        }
        // lane j finishes column n0 + j: one bias load, one gate load and one store per wave and pass
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j) v = lane == j ? bv[j][0] : v;
        const int n = n0 + lane;
        if (lane < NB && n < p.N) {
            if (p.bias) v += p.bias[n];
            if (p.mask) v = p.mask[m * p.ldc + n] > 0.f ? v : 0.f;
            p.C[m * p.ldc + n] = v;
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
    if (n < N) {
        // rows part, part + 16, ... added in that order; eight loads in flight per trip (one per trip measured
        // 5.4 us for a 256 x 256 sum: sixteen dependent round trips)
        for (int m = part; m < M; m += 16 * 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int mm = m + 16 * u;
                t[u] = mm < M ? X[mm * ldx + n] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
    }
    s[threadIdx.x] = v;
    __syncthreads();
    if (part == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += s[q * 16 + threadIdx.x];
        out[n] = accumulate ? out[n] + t : t;
    }
}

// Dense layer's split-K finalize and the fused heads in one launch (the tail of every forward pass): a wave owns a
// row m; lane l holds columns l, l + 64, ... of h[m] = sum_s partial[s][m][:] + bias (slices added in slice order, eight
// loads per column in flight, exactly as gemm_finalize_kernel), writes them, and the row's NH head outputs follow as in
// gemm_rows_kernel (lane-strided fma chain, xor-shuffle reduction, lane j finishes column j): bit-identical to the
// two launches it replaces, one launch boundary and one round trip of h less.
// NA > 0: the action step of the rollout (policy_act.h) runs on the finished head row too - the row sits in the wave's
// lanes, every lane runs the same per-sample body on broadcast values and lane 0 stores (one launch less per env step
// and group: the sampling launch was 12 us of latency for 128 x 6 logits).
struct ActTail {
    float temperature;
    uint64_t seed, offset;
    ActOut out;
    // ... or, in a TRAINING forward (loss != 0), the discrete PPO loss on the finished head row (loss_rows.h: the body of
    // ppo_loss_kernel, same bits): d loss / d heads and the statistics row leave from here and the loss launch is gone
    int loss;
    PpoLossP lp;
    float *dheads;
    const int32_t *index;
};

// NHT: head columns handled (16 or 32: the 15-action suites have 2 * 15 + 1 = 31 heads; their head weights take 128 registers)
template <int KPL, int NA, int NHT>
__global__ __launch_bounds__(256) void finalize_heads_kernel(const float *__restrict__ partial, int split, int M, int H,
                                                            const float *__restrict__ bias, float *__restrict__ h,
                                                            const float *__restrict__ Wh, const float *__restrict__ bh, int NH,
                                                            int relu_h, float *__restrict__ heads, ActTail act)
{
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (m >= M) return;
    const __amdgpu_buffer_rsrc_t pb = buffer_of(partial), wb = buffer_of(Wh), bb = buffer_of(bias, bias != nullptr);
    const int stride = M * H;
    int off[KPL];  // element offset of (m, k) inside a slice, or -1
#pragma unroll
    for (int i = 0; i < KPL; ++i) off[i] = lane + 64 * i < H ? m * H + lane + 64 * i : -1;
    // the head weights of this lane's columns are requested first (NH <= NHT columns, one pass): they depend on nothing
    // and arrive under the slice sums
    float bv[NHT][KPL];
#pragma unroll
    for (int j = 0; j < NHT; ++j) {
        const int row = min(j, NH - 1) * H;
#pragma unroll
        for (int i = 0; i < KPL; ++i) bv[j][i] = buffer_f32(wb, off[i] < 0 ? kOutside : (row + lane + 64 * i) * 4);
    }
    float hv[KPL];
#pragma unroll
    for (int i = 0; i < KPL; ++i) hv[i] = 0.f;
    // the bias is requested with the first slices (it used to be a third dependent round trip behind them)
    float bsv[KPL];
#pragma unroll
    for (int i = 0; i < KPL; ++i) bsv[i] = buffer_f32(bb, off[i] < 0 ? kOutside : (lane + 64 * i) * 4);
    int s = 0;
    // sixteen slices (the dense layer's split) in flight at once: one round trip to the partials, which sit in another
    // XCD's L2 or in memory, instead of two; the additions keep the slice order
    for (; s + 16 <= split; s += 16) {
        float t[KPL][16];
#pragma unroll
        for (int i = 0; i < KPL; ++i)
#pragma unroll
            for (int u = 0; u < 16; ++u) t[i][u] = buffer_f32(pb, off[i] < 0 ? kOutside : ((s + u) * stride + off[i]) * 4);
#pragma unroll
        for (int i = 0; i < KPL; ++i)
#pragma unroll
            for (int u = 0; u < 16; ++u) hv[i] += t[i][u];
    }
    for (; s + 8 <= split; s += 8) {
        float t[KPL][8];
#pragma unroll
        for (int i = 0; i < KPL; ++i)
#pragma unroll
            for (int u = 0; u < 8; ++u) t[i][u] = buffer_f32(pb, off[i] < 0 ? kOutside : ((s + u) * stride + off[i]) * 4);
#pragma unroll
        for (int i = 0; i < KPL; ++i)
#pragma unroll
            for (int u = 0; u < 8; ++u) hv[i] += t[i][u];
    }
    for (; s < split; ++s)
#pragma unroll
        for (int i = 0; i < KPL; ++i) hv[i] += buffer_f32(pb, off[i] < 0 ? kOutside : (s * stride + off[i]) * 4);
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        if (bias) hv[i] += bsv[i];
        if (off[i] >= 0) h[off[i]] = hv[i];
    }
    float out = 0.f;
#pragma unroll
    for (int j = 0; j < NHT; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < KPL; ++i) acc = fmaf(relu_h ? fmaxf(hv[i], 0.f) : hv[i], bv[j][i], acc);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        out = lane == j ? acc : out;
    }
    if (lane < NH) {
        if (bh) out += bh[lane];
        heads[m * NH + lane] = out;
    }
    if constexpr (NA > 0) {
        if (act.loss) {
            ppo_loss_row_z<NA>(act.lp, [&](int i) { return __shfl(out, i, 64); }, act.dheads + (size_t)m * NH, m,
                               act.index ? act.index[m] : m, lane == 0);
            return;
        }
        // raw logits and recorded values: each lane stores its own column (a lane-0 store loop would read the other
        // lanes from inside a divergent branch); the per-sample body then only broadcasts logits in uniform control flow
        if (act.out.raw_policy && lane < NA) act.out.raw_policy[(size_t)m * NA + lane] = out;
        if (act.out.values && lane >= NA && lane < NA + act.out.vh) act.out.values[(size_t)m * act.out.vh + lane - NA] = out;
        ActOut o = act.out;
        o.raw_policy = nullptr;
        o.values = nullptr;
        policy_act_row<NA>([&](int i) { return __shfl(out, i, 64); }, m, NA, act.temperature, nullptr, act.seed, act.offset, 0, o,
                           lane == 0);
    }
}

// Backward of the fused heads in one launch (was: a 13-row product, a 13-column product and two column sums, four
// latency-bound launches at the head of every backward pass):
//   dh[b][k]  = (sum_j dheads[b][j] * Wh[j][k]) * [gate[b][k] > 0]      rows: one wave per sample (blocks 0 .. XB-1)
//   dWh[j][k] = sum_b dheads[b][j] * f(hin[b][k])                        columns: 16 per workgroup, 16 batch partitions
//   dbh[j]    = sum_b dheads[b][j];   db_next[k] = sum_b dh[b][k]        (b = part, part + 16, ...; partitions added in order:
//                                                                         the order ppo_colsum_f32 uses, so both are bit-identical to it)
// NH <= 16.  Range-checked buffer reads throughout (common.h).
struct HeadsBwdArgs {
    const float *dheads, *hin, *gate, *Wh;
    float *dh, *dWh, *dbh, *db_next;
    int B, H, NH, relu_in, xb;
};

__global__ __launch_bounds__(256) void heads_backward_kernel(HeadsBwdArgs a)
{
    __shared__ float s_red[18][256];
    const int tid = threadIdx.x, lane = tid & 63;
    const __amdgpu_buffer_rsrc_t db = buffer_of(a.dheads), wb = buffer_of(a.Wh), hb = buffer_of(a.hin),
                                 gb = buffer_of(a.gate, a.gate != nullptr);
    const bool gate_off = a.gate == nullptr;
    if ((int)blockIdx.x < a.xb) {
        const int b = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(tid >> 6);
        if (b >= a.B) return;
        const float dl = buffer_f32(db, lane < a.NH ? (b * a.NH + lane) * 4 : kOutside);  // lane j holds dheads[b][j]
        float dj[16];
#pragma unroll
        for (int j = 0; j < 16; ++j)  // (the builtin moves ints: bit casts, not conversions); zeros beyond NH
            dj[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dl), j));
        // four columns per lane and trip (the 256 hidden units: ONE trip), all their loads issued before the first use - the
        // one-column loop paid a dependent round trip per 64 columns
        for (int k0 = lane; k0 < a.H; k0 += 256) {
            float w[4][16], gv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + 64 * i;
#pragma unroll
                for (int j = 0; j < 16; ++j) w[i][j] = buffer_f32(wb, (j < a.NH && k < a.H) ? (j * a.H + k) * 4 : kOutside);
                gv[i] = buffer_f32(gb, k < a.H ? (b * a.H + k) * 4 : kOutside);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + 64 * i;
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) acc = fmaf(dj[j], w[i][j], acc);
                if (k < a.H) a.dh[b * a.H + k] = (gv[i] > 0.f || gate_off) ? acc : 0.f;
            }
        }
        return;
    }
    const int kk = tid & 15, part = tid >> 4;
    const int k = ((int)blockIdx.x - a.xb) * 16 + kk;
    const bool col = k < a.H;
    float w[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) w[j] = buffer_f32(wb, (col && j < a.NH) ? (j * a.H + k) * 4 : kOutside);
    float dW[16], dbn = 0.f, dbj = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) dW[j] = 0.f;
    constexpr int SPT = 8;  // samples per trip: their loads are all issued first (a 256-sample minibatch: two round trips)
    for (int b0 = part; b0 < a.B; b0 += 16 * SPT) {
        float x[SPT], gv[SPT], d[SPT][16];
#pragma unroll
        for (int u = 0; u < SPT; ++u) {
            const int b = b0 + 16 * u;
            const bool in = b < a.B;
            x[u] = buffer_f32(hb, (in && col) ? (b * a.H + k) * 4 : kOutside);
            gv[u] = buffer_f32(gb, (in && col) ? (b * a.H + k) * 4 : kOutside);
#pragma unroll
            for (int j = 0; j < 16; ++j) d[u][j] = buffer_f32(db, (in && j < a.NH) ? (b * a.NH + j) * 4 : kOutside);
        }
#pragma unroll
        for (int u = 0; u < SPT; ++u) {
            const float act = a.relu_in ? fmaxf(x[u], 0.f) : x[u];
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                dW[j] = fmaf(d[u][j], act, dW[j]);
                t = fmaf(d[u][j], w[j], t);
            }
            dbn += (gv[u] > 0.f || gate_off) ? t : 0.f;  // rows past B contribute t = 0
            float dsel = 0.f;  // column kk of dheads (only block xb writes the sum); a select chain, not an indexed register
#pragma unroll
            for (int j = 0; j < 16; ++j) dsel = j == kk ? d[u][j] : dsel;
            dbj += dsel;
        }
    }
    // partitions are combined in partition order (deterministic)
#pragma unroll
    for (int j = 0; j < 16; ++j) s_red[j][tid] = dW[j];
    s_red[16][tid] = dbn;
    s_red[17][tid] = dbj;
    __syncthreads();
    if (part == 0) {
        float tot[18];
#pragma unroll
        for (int v = 0; v < 18; ++v) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += s_red[v][q * 16 + kk];
            tot[v] = t;
        }
        if (col) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < a.NH) a.dWh[j * a.H + k] = tot[j];
            if (a.db_next) a.db_next[k] = tot[16];
        }
        if ((int)blockIdx.x == a.xb && a.dbh && kk < a.NH) a.dbh[kk] = tot[17];
    }
}

inline bool vec_ok(const float *P, int64_t s_tile, int64_t s_k, int tile_extent, int k_extent)
{
    // float4 along the contiguous axis: the other stride and the base keep 16-byte alignment, and the extent along
    // the contiguous axis is a multiple of 4 (no float4 straddles the edge)
    const int64_t other = s_k == 1 ? s_tile : s_k;
    const int extent = s_k == 1 ? k_extent : tile_extent;
    return (s_k == 1 || s_tile == 1) && other % 4 == 0 && extent % 4 == 0 && aligned(P, 16);
}

}  // namespace
}  // namespace ppo

extern "C" size_t ppo_gemm_workspace_bytes(int M, int N, int K)
{
    (void)K;
    return (size_t)32 * M * N * sizeof(float);  // up to 32 split-K slices
}

namespace ppo {
namespace {
// the fused heads that may ride on a split-K finalize (ppo_dense_heads_forward_f32)
struct HeadsTail {
    const float *Wh, *bh;
    float *heads;
    int NH, relu_h;
    bool fused;  // out: the finalize launch produced the heads too
    const ActTail *act = nullptr;  // sample actions from the head row in the same launch (n_actions in act_n)
    int act_n = 0;
    bool act_fused = false;  // out
};

int gemm_dispatch(const float *A, int64_t a_sm, int64_t a_sk, int relu_a, const float *B, int64_t b_sk, int64_t b_sn,
                  int relu_b, const float *bias, const float *mask, float *C, int64_t ldc, int M, int N, int K,
                  void *workspace, size_t workspace_bytes, void *stream, HeadsTail *tail)
{
    if (M < 0 || N < 0 || K < 0) return fail(PPO_E_INVALID, "ppo_gemm_f32: negative dimension");
    if (M == 0 || N == 0) return PPO_OK;
    if (!A || !B || !C) return fail(PPO_E_INVALID, "ppo_gemm_f32: null pointer");
    if (ldc < N) return fail(PPO_E_INVALID, "ppo_gemm_f32: ldc < N");
    // the kernels address their operands with 32-bit byte offsets
    const int64_t a_last = (int64_t)(M - 1) * a_sm + (int64_t)(K - 1) * a_sk, b_last = (int64_t)(N - 1) * b_sn + (int64_t)(K - 1) * b_sk;
    if (K > 0 && (a_last * 4 + 16 > (int64_t)kBufferBytes || b_last * 4 + 16 > (int64_t)kBufferBytes || a_sm < 0 || a_sk < 0 ||
                  b_sn < 0 || b_sk < 0))
        return fail(PPO_E_INVALID, "ppo_gemm_f32: operand spans 2 GiB or more (or a negative stride); split the batch");
    hipStream_t st = as_stream(stream);
    // The choice of kernel must not depend on the batch size, or a rollout forward (batch A) and a training
    // forward (batch 256) of the same sample would round differently.  N and K of a forward product are model
    // dimensions; M is the batch unless A is read transposed (a_sm == 1: a weight-gradient product dY^T X,
    // where M is a model dimension and K the batch).
    if ((N <= 16 || K <= 16 || (M <= 16 && a_sm == 1)) && (int64_t)M * N <= (1 << 22)) {
        GemmArgs q{A, B, C, bias, mask, M, N, K, a_sm, a_sk, b_sk, b_sn, ldc, relu_a, relu_b, K, 1, 0, 0, 0};
        if (N <= kRowsMaxN && a_sk == 1 && b_sk == 1 && K <= 64 * kRowsMaxKPerLane) {
            if (K <= 256 && N <= 16)
                hipLaunchKernelGGL((gemm_rows_kernel<16, 4>), dim3((M + 3) / 4), dim3(256), 0, st, q);
            else if (K <= 256)
                hipLaunchKernelGGL((gemm_rows_kernel<8, 4>), dim3((M + 3) / 4), dim3(256), 0, st, q);
            else
                hipLaunchKernelGGL((gemm_rows_kernel<4, kRowsMaxKPerLane>), dim3((M + 3) / 4), dim3(256), 0, st, q);
            return check_launch("gemm_rows_kernel");
        }
        hipLaunchKernelGGL(gemm_small_kernel, dim3((M * N + 15) / 16), dim3(256), 0, st, q);
        return check_launch("gemm_small_kernel");
    }
    const int gm = (M + BM - 1) / BM, gn = (N + BN - 1) / BN;
    // Few tiles: split K into slices of kSliceK (the slice count depends on K and on the tile counts of the model
    // dimensions only through `few`, which for a forward product x @ W^T is a property of N: a rollout forward and a
    // training forward of the same sample must round alike whatever the batch).
    constexpr int kSliceK = 256;
    int split = 1;
    const bool few = gn * 4 <= 32;  // at most 8 column tiles: even a 2048-row batch leaves CUs idle without slicing
    if (few && K >= 2 * kSliceK) split = std::min(32, (K + kSliceK - 1) / kSliceK);
    if (split > 1 && (!workspace || workspace_bytes < (size_t)split * M * N * sizeof(float))) split = 1;
    int kps = (K + split - 1) / split;
    kps = (kps + BK - 1) / BK * BK;
    if (split > 1) split = (K + kps - 1) / kps;  // no empty slices
    GemmArgs p{A, B, split > 1 ? static_cast<float *>(workspace) : C, bias, mask, M, N, K, a_sm, a_sk, b_sk, b_sn,
               ldc, relu_a, relu_b, kps, split, vec_ok(A, a_sm, a_sk, M, K) ? 1 : 0, vec_ok(B, b_sn, b_sk, N, K) ? 1 : 0, 0};
    p.vec_c = N % 4 == 0 && ldc % 4 == 0 && aligned(C, 16) && aligned(p.C, 16) && (!bias || aligned(bias, 16)) &&
              (!mask || (aligned(mask, 16) && (int64_t)M * ldc * 4 < (int64_t)kBufferBytes));
    const dim3 grid(tile_grid(gm, gn, split));
    const bool vec = p.vec_a && p.vec_b;
    const int variant = (a_sk == 1 ? 4 : 0) | (b_sk == 1 ? 2 : 0) | (vec ? 1 : 0);
#define PPO_GEMM_CASE(v, ak, bk, vc)                                                                  \
    case v:                                                                                           \
        hipLaunchKernelGGL((gemm_f32_kernel<ak, bk, vc>), grid, dim3(kGemmThreads), 0, st, p, gm, gn); \
        break;
    switch (variant) {
        PPO_GEMM_CASE(0, false, false, false)
        PPO_GEMM_CASE(1, false, false, true)
        PPO_GEMM_CASE(2, false, true, false)
        PPO_GEMM_CASE(3, false, true, true)
        PPO_GEMM_CASE(4, true, false, false)
        PPO_GEMM_CASE(5, true, false, true)
        PPO_GEMM_CASE(6, true, true, false)
        PPO_GEMM_CASE(7, true, true, true)
    }
#undef PPO_GEMM_CASE
    int rc = check_launch("gemm_f32_kernel");
    if (rc) return rc;
    if (split > 1) {
        if (tail && !mask && ldc == N && N <= 256 && tail->NH <= 32 && (int64_t)split * M * N * 4 < (int64_t)kBufferBytes) {
#define PPO_FINALIZE(NA, NHT, ACT)                                                                                     \
    hipLaunchKernelGGL((finalize_heads_kernel<4, NA, NHT>), dim3((M + 3) / 4), dim3(256), 0, st,                       \
                       static_cast<const float *>(workspace), split, M, N, bias, C, tail->Wh, tail->bh, tail->NH,      \
                       tail->relu_h, tail->heads, ACT)
            tail->act_fused = tail->act != nullptr;
            const bool wide = tail->NH > 16;  // (a 512-row launch of the 31-head net made four launches before: 90 us bracketed)
#define PPO_FIN2(NA, ACT)                       \
    do {                                        \
        if (wide) PPO_FINALIZE(NA, 32, ACT);    \
        else PPO_FINALIZE(NA, 16, ACT);         \
    } while (0)
            switch (tail->act ? tail->act_n : 0) {  // the action counts of the benchmark suites; others sample in their own launch
                case 4: PPO_FIN2(4, *tail->act); break;
                case 6: PPO_FIN2(6, *tail->act); break;
                case 15: PPO_FIN2(15, *tail->act); break;
                case 18: PPO_FIN2(18, *tail->act); break;
                default: PPO_FIN2(0, ActTail{}); tail->act_fused = false;
            }
#undef PPO_FIN2
#undef PPO_FINALIZE
            tail->fused = true;
            return check_launch("finalize_heads_kernel");
        }
        hipLaunchKernelGGL(gemm_finalize_kernel, dim3((M * N + 255) / 256), dim3(256), 0, st,
                           static_cast<const float *>(workspace), split, M, N, bias, mask, C, ldc);
        rc = check_launch("gemm_finalize_kernel");
    }
    return rc;
}
}  // namespace
}  // namespace ppo

extern "C" int ppo_gemm_f32(const float *A, int64_t a_sm, int64_t a_sk, int relu_a, const float *B, int64_t b_sk,
                            int64_t b_sn, int relu_b, const float *bias, const float *mask, float *C, int64_t ldc,
                            int M, int N, int K, void *workspace, size_t workspace_bytes, void *stream)
{
    return ppo::gemm_dispatch(A, a_sm, a_sk, relu_a, B, b_sk, b_sn, relu_b, bias, mask, C, ldc, M, N, K, workspace,
                              workspace_bytes, stream, nullptr);
}

extern "C" int ppo_dense_heads_forward_f32(const float *x, int relu_x, const float *W, const float *b, const float *Wh,
                                           const float *bh, int relu_h, float *h, float *heads, int M, int K, int H, int NH,
                                           void *workspace, size_t workspace_bytes, void *stream)
{
    using namespace ppo;
    if (M < 0 || K < 0 || H <= 0 || NH <= 0) return fail(PPO_E_INVALID, "ppo_dense_heads_forward_f32: bad dimension");
    if (M == 0) return PPO_OK;
    if (!x || !W || !Wh || !h || !heads) return fail(PPO_E_INVALID, "ppo_dense_heads_forward_f32: null pointer");
    HeadsTail tail{Wh, bh, heads, NH, relu_h, false};
    int rc = gemm_dispatch(x, K, 1, relu_x, W, 1, K, 0, b, nullptr, h, H, M, H, K, workspace, workspace_bytes, stream, &tail);
    if (rc || tail.fused) return rc;
    // the dense product ran unsplit (no workspace, short K, wide layer): the heads are their own launch
    return gemm_dispatch(h, H, 1, relu_h, Wh, 1, H, 0, bh, nullptr, heads, NH, M, NH, H, nullptr, 0, stream, nullptr);
}

extern "C" int ppo_policy_act_f32(const float *heads, int B, int ldo, int n_actions, float temperature, const float *uniform,
                                  uint64_t seed, uint64_t offset, int greedy, float *log_policy, int32_t *actions,
                                  float *log_pac, float *raw_policy, float *values, int n_value_heads, void *stream);

extern "C" int ppo_dense_heads_act_forward_f32(const float *x, int relu_x, const float *W, const float *b, const float *Wh,
                                               const float *bh, int relu_h, float *h, float *heads, int M, int K, int H,
                                               int NH, void *workspace, size_t workspace_bytes, int n_actions,
                                               float temperature, uint64_t seed, uint64_t offset, float *log_policy,
                                               int32_t *actions, float *log_pac, float *raw_policy, float *values,
                                               int n_value_heads, void *stream)
{
    using namespace ppo;
    if (M < 0 || K < 0 || H <= 0 || NH <= 0) return fail(PPO_E_INVALID, "ppo_dense_heads_act_forward_f32: bad dimension");
    if (n_actions <= 0 || n_actions > kMaxActions || n_value_heads < 0 || NH < n_actions + n_value_heads)
        return fail(PPO_E_INVALID, "ppo_dense_heads_act_forward_f32: bad head layout (n_actions=%d value heads=%d of %d)",
                    n_actions, n_value_heads, NH);
    if (!(temperature > 0.f)) return fail(PPO_E_INVALID, "ppo_dense_heads_act_forward_f32: temperature must be > 0");
    if (M == 0) return PPO_OK;
    if (!x || !W || !Wh || !h || !heads) return fail(PPO_E_INVALID, "ppo_dense_heads_act_forward_f32: null pointer");
    const ActTail act{temperature, seed, offset, ActOut{log_policy, actions, log_pac, raw_policy, values, n_value_heads}, 0,
                      PpoLossP{}, nullptr, nullptr};
    HeadsTail tail{Wh, bh, heads, NH, relu_h, false};
    tail.act = &act;
    tail.act_n = n_actions;
    int rc = gemm_dispatch(x, K, 1, relu_x, W, 1, K, 0, b, nullptr, h, H, M, H, K, workspace, workspace_bytes, stream, &tail);
    if (rc) return rc;
    if (!tail.fused)
        rc = gemm_dispatch(h, H, 1, relu_h, Wh, 1, H, 0, bh, nullptr, heads, NH, M, NH, H, nullptr, 0, stream, nullptr);
    if (rc || tail.act_fused) return rc;
    return ppo_policy_act_f32(heads, M, NH, n_actions, temperature, nullptr, seed, offset, 0, log_policy, actions, log_pac,
                              raw_policy, values, n_value_heads, stream);
}

extern "C" int ppo_ppo_loss_f32(const float *heads, int B, int ldo, int n_actions, int n_value_heads, const int32_t *actions,
                                const float *old_log_pac, const float *old_log_policy, const float *advantages,
                                const float *returns, float eps_clip, float ent_coef, float vf_coef, float grad_scale,
                                float *dheads, float *stats, const int32_t *index, void *stream);

extern "C" int ppo_dense_heads_loss_forward_f32(const float *x, int relu_x, const float *W, const float *b, const float *Wh,
                                                const float *bh, int relu_h, float *h, float *heads, int M, int K, int H,
                                                int NH, void *workspace, size_t workspace_bytes, int n_actions,
                                                int n_value_heads, const int32_t *actions, const float *old_log_pac,
                                                const float *old_log_policy, const float *advantages, const float *returns,
                                                float eps_clip, float ent_coef, float vf_coef, float grad_scale, float *dheads,
                                                float *stats, const int32_t *index, void *stream)
{
    using namespace ppo;
    if (M < 0 || K < 0 || H <= 0 || NH <= 0) return fail(PPO_E_INVALID, "ppo_dense_heads_loss_forward_f32: bad dimension");
    if (n_actions <= 0 || n_actions > kMaxActions || n_value_heads < 0 || NH < n_actions + n_value_heads)
        return fail(PPO_E_INVALID, "ppo_dense_heads_loss_forward_f32: bad head layout (n_actions=%d value heads=%d of %d)",
                    n_actions, n_value_heads, NH);
    if (M == 0) return PPO_OK;
    if (!x || !W || !Wh || !h || !heads || !actions || !old_log_pac || !advantages || !dheads || (n_value_heads > 0 && !returns))
        return fail(PPO_E_INVALID, "ppo_dense_heads_loss_forward_f32: null pointer");
    ActTail act{};
    act.loss = 1;
    act.lp = PpoLossP{NH, n_actions, n_value_heads, actions, old_log_pac, old_log_policy, advantages, returns, eps_clip,
                      ent_coef, vf_coef, grad_scale, stats};
    act.dheads = dheads, act.index = index;
    HeadsTail tail{Wh, bh, heads, NH, relu_h, false};
    tail.act = &act;
    tail.act_n = n_actions;
    int rc = gemm_dispatch(x, K, 1, relu_x, W, 1, K, 0, b, nullptr, h, H, M, H, K, workspace, workspace_bytes, stream, &tail);
    if (rc) return rc;
    if (!tail.fused)
        rc = gemm_dispatch(h, H, 1, relu_h, Wh, 1, H, 0, bh, nullptr, heads, NH, M, NH, H, nullptr, 0, stream, nullptr);
    if (rc || tail.act_fused) return rc;
    return ppo_ppo_loss_f32(heads, M, NH, n_actions, n_value_heads, actions, old_log_pac, old_log_policy, advantages, returns,
                            eps_clip, ent_coef, vf_coef, grad_scale, dheads, stats, index, stream);
}

extern "C" int ppo_heads_backward_f32(const float *dheads, const float *hin, int relu_in, const float *gate, const float *Wh,
                                      float *dh, float *dWh, float *dbh, float *db_next, int B, int H, int NH, void *stream)
{
    using namespace ppo;
    if (B <= 0 || H <= 0 || NH <= 0 || NH > 16) return fail(PPO_E_INVALID, "ppo_heads_backward_f32: needs B, H > 0 and 1 <= NH <= 16");
    if (!dheads || !hin || !Wh || !dh || !dWh) return fail(PPO_E_INVALID, "ppo_heads_backward_f32: null pointer");
    if ((int64_t)B * H * 4 + 16 > (int64_t)kBufferBytes) return fail(PPO_E_INVALID, "ppo_heads_backward_f32: B * H spans 2 GiB; split the batch");
    HeadsBwdArgs a{dheads, hin, gate, Wh, dh, dWh, dbh, db_next, B, H, NH, relu_in, (B + 3) / 4};
    hipLaunchKernelGGL(heads_backward_kernel, dim3(a.xb + (H + 15) / 16), dim3(256), 0, as_stream(stream), a);
    return check_launch("heads_backward_kernel");
}

extern "C" int ppo_colsum_f32(const float *X, int M, int N, int64_t ldx, float *out, int accumulate, void *stream)
{
    using namespace ppo;
    if (M < 0 || N <= 0 || !X || !out) return fail(PPO_E_INVALID, "ppo_colsum_f32: bad arguments");
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 15) / 16), dim3(256), 0, as_stream(stream), X, M, N, ldx, out, accumulate);
    return check_launch("colsum_kernel");
}

// Truncated-horizon (TVF) return estimator on gfx950 — replaces the reference's
// rl/returns_truncated.py:623-693 (_calculate_sampled_return_multi_fast) + :558-620 (_n_step_estimate)
// + :142-174 (_interpolate), called from rl/tvf.py:250-262.
//
//   ret[t,a,k] = mean_c ( S_n[t,a] + boot_{k,c}[t,a] * D_n[t,a] )       n = min(sample[k,c], h_k)
//   D_i[t] = prod_{j<i, t+j<N} gamma*(1-done[t+j])     S_n[t] = sum_{i<n, t+i<N} r[t+i] * D_i[t]
//   boot = interp(V[t+n, a, :], h_k - n)  for t < N-n,   interp(V[N, a, :], h_k - (N-t))  otherwise
//
// The host resolves everything that does not depend on (t, a): which n are needed, and for every
// (k, c) and every (k, N-t) the interpolation plan (zero / exact column / two columns and their
// float64 weights) exactly as the reference's searchsorted logic decides it.
//
// tvf_column_kernel (the product path): ONE workgroup per env column a keeps that column's value samples
// V[0..N, a, :] in LDS (257 x 108 floats = 112 KB at the config size, rows padded to an odd stride so that
// 64 lanes reading one column of 64 consecutive rows hit 64 banks), so value_samples cross HBM once and
// every bootstrap gather is an LDS read.  Time is walked in chunks of T rows (lanes = t):
//   walk    one wave, lane = t: the running S and D over i in the reference's order and precision (float32
//           arrays, the step factor gamma*(1-done) formed in float64), stored at the needed n as [nd][t];
//   terms   16 waves, lane = t, (k, c) uniform per wave => n, the interpolation plan and its weights are
//           scalar loads; per term one 8-byte LDS read (S, D), one or two 4-byte LDS reads of the value
//           column, the float64 blend, and the reference's float32 accumulation order (no contraction)
//           => bit-identical to the reference; rows t >= N-n take their per-(k, N-t) tail plan per lane;
//   store   results go through an LDS tile [t][k] so that out[t, a, :] leaves as whole 4K-byte rows.
// HBM traffic = the algorithmic 4*(N+1)*A*V read + 4*N*A*K written.  What bounds it is VALU
// issue, not HBM: a term is ~15 wave instructions (address, 2-3 LDS reads, 2 cvt + 3 float64 ops for the
// NumPy-promotion blend, cvt, mul, 2 adds), K*C of them per (t, a).
//
// tvf_prefix_kernel + tvf_gather_kernel (fallback when a column does not fit in LDS: very long rollouts or
// several hundred heads): one thread per (t, a) walks S/D into an [nd][t, a] workspace, then one thread per
// (t, a, k) gathers from global memory; same arithmetic, same bits.
#include <cstdlib>

#include "common.h"

namespace ppo {
namespace {

__global__ __launch_bounds__(256) void tvf_prefix_kernel(const float *__restrict__ rewards,
                                                         const uint8_t *__restrict__ dones, int N, int A, double gamma,
                                                         const int32_t *__restrict__ nd_of_n, int max_n, int ND,
                                                         float *__restrict__ cS, float *__restrict__ cD)
{
#pragma clang fp contract(off)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * A) return;
    const int t = idx / A;
    const int a = idx - t * A;
    float S = 0.f, D = 1.f;
    const size_t NA = (size_t)N * A;  // workspace is [nd][t, a]: neighbouring threads write neighbouring floats
    float *oS = cS + idx;
    float *oD = cD + idx;
    for (int i = 0; i < max_n; ++i) {
        if (t + i < N) {  // rows past the end of the rollout stop updating (s[:N-i] += ..., :667-669)
            const size_t j = (size_t)(t + i) * A + a;
            const float term = rewards[j] * D;
            S = S + term;
            const double step = dones[j] ? 0.0 : gamma;  // gamma * (1 - bool) is float64
            D = (float)((double)D * step);                // ... rounded back into the float32 array
        }
        const int nd = nd_of_n[i + 1];
        if (nd >= 0) {
            oS[nd * NA] = S;
            oD[nd * NA] = D;
        }
    }
}

// A plan row (mode, i0, i1) + weights (w0, w1) says how to evaluate interp(values[..., :], target):
// mode 0 -> zero, 1 -> values[i0], 2 -> values[i0]*w0 + values[i1]*w1.
// The two-column case is evaluated in float64 and rounded once: in the reference the interpolation factor
// is a NumPy float64 scalar (a quotient of NumPy integers), and under NumPy >= 2 promotion (NEP 50, the NumPy
// the golden vectors were produced with) float32_array * float64_scalar is a float64 array; the float32
// destination rounds it (rl/returns_truncated.py:171-174, 609-614).
__device__ __forceinline__ float apply_plan(const float *__restrict__ row, const int32_t *__restrict__ p,
                                            const double *__restrict__ w)
{
#pragma clang fp contract(off)
    const int mode = p[0];
    if (mode == 0) return 0.f;
    const float v0 = row[p[1]];
    if (mode == 1) return v0;
    const float v1 = row[p[2]];
    const double x0 = (double)v0 * w[0];
    const double x1 = (double)v1 * w[1];
    return (float)(x0 + x1);
}

__global__ __launch_bounds__(256) void tvf_gather_kernel(const float *__restrict__ values, int N, int A, int V, int K,
                                                         int C, int ND, const int32_t *__restrict__ n_eff,
                                                         const int32_t *__restrict__ nd_index,
                                                         const int32_t *__restrict__ main_plan,
                                                         const double *__restrict__ main_w,
                                                         const int32_t *__restrict__ tail_plan,
                                                         const double *__restrict__ tail_w,
                                                         const uint8_t *__restrict__ k_zero, float inv_c,
                                                         const float *__restrict__ cS, const float *__restrict__ cD,
                                                         float *__restrict__ out)
{
#pragma clang fp contract(off)
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * A * K) return;
    const int k = idx % K;
    const int64_t ta = idx / K;
    const int t = ta / A;
    const int a = ta - (int64_t)t * A;
    float total = 0.f;
    const size_t NA = (size_t)N * A;
    if (!k_zero[k]) {
        const float *last_row = values + ((size_t)N * A + a) * V;
        for (int c = 0; c < C; ++c) {
            const int kc = k * C + c;
            const int n = n_eff[kc];
            const int nd = nd_index[kc];
            const float S = cS[(size_t)nd * NA + ta];
            const float D = cD[(size_t)nd * NA + ta];
            float boot;
            if (t < N - n) {
                boot = apply_plan(values + ((size_t)(t + n) * A + a) * V, main_plan + 3 * kc, main_w + 2 * kc);
            } else {
                const int kj = k * (N + 1) + (N - t);
                boot = apply_plan(last_row, tail_plan + 3 * kj, tail_w + 2 * kj);
            }
            const float md = boot * D;
            const float term = S + md;
            total = total + term;
        }
        total = total * inv_c;
    }
    out[idx] = total;
}

// ---------------------------------------------------------------------------------------------------------
// The plan as 32-byte records, so that a sample's whole plan is ONE scalar load (x8) instead of five from five
// arrays with a wait between them, and a lane's tail plan two 16-byte loads.  Packed by a small launch in front of the
// column kernel, into the caller's workspace.
struct alignas(16) MainRec {  // everything a term needs as ready-made LDS byte offsets: no scalar arithmetic per term
    int32_t off_sd;  // nd * T * 8: the (S, D) row of this n in the chunk table
    int32_t off_v;   // (n * VS + i0) * 4: from a lane's row t to V[t + n, i0]
    int32_t lim;     // (N * VS + i0) * 4: the same column of the last row V[N] - the clamp of off_v + row
    int32_t thr;     // N - n: rows t >= thr bootstrap from V[N] (their tail plan) instead
    double w0, w1;
    int32_t mode, pad0, pad1, pad2;
};
struct alignas(8) LaneRec {  // the same, 24 bytes: kept in vector registers, one record per lane (see the kernel)
    uint32_t a;  // off_sd / 8 | thr << 16   (off_sd / 8 = nd * T < 20480 and thr <= N < 40960 while the column fits in LDS)
    uint32_t b;  // off_v | mode << 24            (lim = off_v + thr * VS * 4)
    double w0, w1;
};
struct alignas(32) TailRec {
    int32_t mode, i0, i1, pad;
    double w0, w1;
};

__global__ __launch_bounds__(256) void tvf_pack_kernel(const int32_t *__restrict__ n_eff, const int32_t *__restrict__ nd_index,
                                                       const int32_t *__restrict__ main_plan,
                                                       const double *__restrict__ main_w,
                                                       const int32_t *__restrict__ tail_plan,
                                                       const double *__restrict__ tail_w, int n_main, int n_tail,
                                                       int N, int VS, int T, MainRec *__restrict__ mrec,
                                                       LaneRec *__restrict__ lrec, TailRec *__restrict__ trec)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_main) {
        MainRec r;
        const int n = n_eff[i], i0 = main_plan[3 * i + 1];
        r.off_sd = nd_index[i] * T * 8;
        r.off_v = (n * VS + i0) * 4;
        r.lim = (N * VS + i0) * 4;
        r.thr = N - n;
        r.mode = main_plan[3 * i], r.pad0 = r.pad1 = r.pad2 = 0;
        r.w0 = main_w[2 * i], r.w1 = main_w[2 * i + 1];
        mrec[i] = r;
        LaneRec l;
        l.a = (uint32_t)(r.off_sd >> 3) | (uint32_t)r.thr << 16;
        l.b = (uint32_t)r.off_v | (uint32_t)r.mode << 24;
        l.w0 = r.w0, l.w1 = r.w1;
        lrec[i] = l;
    } else if (i < n_main + n_tail) {
        const int j = i - n_main;
        TailRec r;
        r.mode = tail_plan[3 * j], r.i0 = tail_plan[3 * j + 1], r.i1 = tail_plan[3 * j + 2], r.pad = 0;
        r.w0 = tail_w[2 * j], r.w1 = tail_w[2 * j + 1];
        trec[j] = r;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Column-resident form.  Dynamic LDS (floats): vals [(N+1) x VS] | rw [N + PAD] | (8-byte aligned) stepd [N + PAD]
// doubles | needed [PAD / 32 + 2] words | (8-byte aligned) sd [ND x T] float2, re-used as the output tile [T x KS]
// floats.  VS = V|1, KS = K|1 (odd strides); PAD = max_n rounded up to 8, + 8: the walk reads rewards / step factors
// past the end of the rollout, where they are 0 / 1.0 - an update with those is the identity, bit for bit, so the
// "rows past the end stop updating" rule of the reference needs neither a clamp nor a select.
struct ColumnLds {
    int VS, KS, T, PAD;
    size_t rw_off, step_off, need_off, sd_off, bytes;  // offsets in floats
};

inline ColumnLds column_lds(int N, int V, int K, int ND, int max_n, int T)
{
    ColumnLds c;
    c.VS = V | 1;
    c.KS = K | 1;
    c.T = T;
    c.PAD = ((max_n + 7) & ~7) + 8;
    size_t f = (size_t)(N + 1) * c.VS + 2;  // + 2: the paired read of (i0, i0 + 1) may touch one float past the last row
    c.rw_off = f;
    f += (size_t)N + c.PAD;
    f = (f + 1) & ~(size_t)1;
    c.step_off = f;
    f += 2 * ((size_t)N + c.PAD);
    c.need_off = f;
    f += (size_t)c.PAD / 32 + 2;
    f = (f + 1) & ~(size_t)1;
    c.sd_off = f;
    const size_t sd = (size_t)ND * T * 8, tile = (size_t)T * c.KS * 4;
    c.bytes = f * 4 + (sd > tile ? sd : tile);
    return c;
}

constexpr int kColumnThreads = 1024;
constexpr int kColumnWaves = kColumnThreads / kWave;
constexpr size_t kLdsLimit = 160 * 1024;

struct ColumnArgs {  // the scalars; the pointers stay separate __restrict__ parameters (scalar loads need that)
    double gamma;
    int N, A, V, K, C, ND, T, max_n, PAD, rw_off, step_off, need_off, sd_off, skip;
    float inv_c;
};

template <int MAXR>  // result registers per lane: K <= MAXR * kColumnWaves
__global__ __launch_bounds__(kColumnThreads) void tvf_column_kernel(
    const float *__restrict__ rewards, const uint8_t *__restrict__ dones, const float *__restrict__ values,
    const int32_t *__restrict__ nd_of_n, const MainRec *__restrict__ mrec, const LaneRec *__restrict__ lrec,
    const TailRec *__restrict__ trec, const uint8_t *__restrict__ k_zero, float *__restrict__ out, const ColumnArgs p)
{
#pragma clang fp contract(off)
    extern __shared__ float lds[];
    const int N = p.N, A = p.A, V = p.V, K = p.K, C = p.C, T = p.T, max_n = p.max_n;
    const int a = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int VS = V | 1, KS = K | 1;
    float *vals = lds;
    float *rw = lds + p.rw_off;
    double *stepd = reinterpret_cast<double *>(lds + p.step_off);
    uint32_t *needed = reinterpret_cast<uint32_t *>(lds + p.need_off);  // bit b of word w: prefix length 32 w + b + 1 is used
    float2 *sd = reinterpret_cast<float2 *>(lds + p.sd_off);
    float *tile = lds + p.sd_off;

    // ---- the column: V[0..N, a, :] (rows of V floats, contiguous in HBM), four loads in flight per thread
    if ((V & 3) == 0) {
        const int nq = V >> 2, total = (N + 1) * nq;
        for (int q0 = tid; q0 < total; q0 += 4 * kColumnThreads) {
            float4 v[4];
            int dst[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + u * kColumnThreads;
                const int qc = q < total ? q : total - 1;
                const int row = qc / nq, cq = qc - row * nq;
                v[u] = *reinterpret_cast<const float4 *>(values + ((size_t)row * A + a) * V + 4 * cq);
                dst[u] = q < total ? row * VS + 4 * cq : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (dst[u] >= 0) {
                    float *d = vals + dst[u];
                    d[0] = v[u].x;
                    d[1] = v[u].y;
                    d[2] = v[u].z;
                    d[3] = v[u].w;
                }
            }
        }
    } else {
        const int total = (N + 1) * V;
        for (int q = tid; q < total; q += kColumnThreads) {
            const int row = q / V, col = q - row * V;
            vals[(size_t)row * VS + col] = values[((size_t)row * A + a) * V + col];
        }
    }
    for (int t = tid; t < N + p.PAD; t += kColumnThreads) {
        const bool in = t < N;
        const float r = in ? rewards[(size_t)t * A + a] : 0.f;
        const bool done = in && dones[(size_t)t * A + a];
        rw[t] = r;
        stepd[t] = in ? (done ? 0.0 : p.gamma) : 1.0;  // gamma * (1 - bool) is float64 (rl/returns_truncated.py:672)
    }
    if (tid < 2) vals[(size_t)(N + 1) * VS + tid] = 0.f;
    for (int i = tid; i < ((max_n + 63) & ~63); i += kColumnThreads) {  // whole waves: one ballot covers 64 prefix lengths
        const bool used = i < max_n && nd_of_n[i + 1] >= 0;
        const uint64_t bits = __ballot(used);
        if (lane == 0) {
            needed[i >> 5] = (uint32_t)bits;
            needed[(i >> 5) + 1] = (uint32_t)(bits >> 32);
        }
    }
    // which of this wave's heads are the h = 0 ones (value 0 by definition): one bit per result register
    uint32_t kz = 0;
#pragma unroll
    for (int j = 0; j < MAXR; ++j) {
        const int k = j * kColumnWaves + wave;
        if (k >= K || k_zero[k]) kz |= 1u << j;
    }
    kz = __builtin_amdgcn_readfirstlane(kz);
    // This wave's plan records, ONE PER LANE in vector registers (record j * C + c of the wave in lane (j * C + c) % 64
    // of register set (j * C + c) / 64), loaded once: a term then gets its record with six v_readlane and no memory
    // access at all.  As scalar loads they cost one L2 round trip per group of terms - the table (K * C * 48 bytes) is
    // larger than the scalar cache and every wave streams its own part of it once per chunk - and that round trip,
    // not instruction issue, set the pace.  Up to kLaneC samples per head; more take the scalar-load path below.
    constexpr int kLaneC = 8;
    constexpr int NSET = (MAXR * kLaneC + kWave - 1) / kWave;
    const bool lane_recs = C <= kLaneC;
    uint32_t ra[NSET], rb[NSET], rw0l[NSET], rw0h[NSET], rw1l[NSET], rw1h[NSET];
#pragma unroll
    for (int s2 = 0; s2 < NSET; ++s2) {
        const int ri = s2 * kWave + lane;
        const int j = ri / C, c = ri - j * C;
        const int k = j * kColumnWaves + wave;
        const bool ok = lane_recs && j < MAXR && k < K;
        const LaneRec r = lrec[ok ? (size_t)k * C + c : 0];
        ra[s2] = r.a, rb[s2] = r.b;
        rw0l[s2] = (uint32_t)__double2loint(r.w0), rw0h[s2] = (uint32_t)__double2hiint(r.w0);
        rw1l[s2] = (uint32_t)__double2loint(r.w1), rw1h[s2] = (uint32_t)__double2hiint(r.w1);
    }
    __syncthreads();

    const float *last_row = vals + (size_t)N * VS;
    const bool lane_on = lane < T;
    const char *sd_lane_b = reinterpret_cast<const char *>(sd + (lane_on ? lane : 0));
    for (int t0 = 0; t0 < N; t0 += T) {
        const int t = t0 + lane;
        // ---- walk: S_n[t], D_n[t] at every needed n, reference order (rl/returns_truncated.py:667-672).  The slots of
        // the needed n ascend with n, so the slot is a counter; "n is needed" is one bit of a mask in SGPRs.  Eight
        // steps at a time: their reads are issued together at immediate offsets from one address.
        if (wave == 0 && !(p.skip & 1)) {
            float S = 0.f, D = 1.f;
            int slot = 0;
            uint32_t need = 0;
            const int tw = t < N ? t : N;  // idle lanes walk the padding
            for (int i0 = 0; i0 < max_n; i0 += 8) {
                if ((i0 & 31) == 0) need = __builtin_amdgcn_readfirstlane(needed[i0 >> 5]);
                const float *rp = rw + tw + i0;
                const double *sp = stepd + tw + i0;
                float r[8];
                double st[8];
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    r[b] = rp[b];
                    st[b] = sp[b];
                }
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const float term = r[b] * D;
                    S = S + term;
                    D = (float)((double)D * st[b]);  // float64 product rounded back into the float32 array
                    if ((need >> ((i0 & 31) + b)) & 1u) {  // uniform: n = i0 + b + 1 is used by some (k, c)
                        if (lane_on) sd[slot * T + lane] = make_float2(S, D);
                        ++slot;
                    }
                }
            }
        }
        __syncthreads();
        // ---- terms.  Four (k, c) samples at a time: their scalar plan loads, then their LDS reads, are issued
        // together (a term is a chain of scalar load -> LDS read -> float64 blend; one at a time a wave spends its
        // life waiting, and the four waves of a SIMD cannot cover it).
        const int t_hi = t0 + T - 1;  // the last row of this chunk: decides, per n, whether any lane needs its tail plan
        const int jt = t < N ? N - t : 0;  // tail-plan row of this lane (row 0 is all zeros: idle lanes)
        const uint32_t row_t = (uint32_t)(t < N ? t : N) * VS * 4;
        // the tail bootstrap of every head of this wave for this lane's row: V[N] at horizon h_k - (N - t), used by the
        // terms with n >= N - t.  All loads first, then the blends.
        float tail_boot[MAXR];
#pragma unroll
        for (int j = 0; j < MAXR; ++j) tail_boot[j] = 0.f;
        if (t_hi >= N - max_n && !(p.skip & 2)) {  // uniform
#pragma unroll
            for (int g = 0; g < MAXR; g += 8) {  // eight heads at a time: 56 registers of plan in flight
                TailRec tr[8];
                float a0[8], a1[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = (kz >> (g + j)) & 1u ? 0 : (g + j) * kColumnWaves + wave;
                    tr[j] = trec[(size_t)k * (N + 1) + jt];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    a0[j] = last_row[tr[j].i0];
                    a1[j] = last_row[tr[j].i1];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double x0 = (double)a0[j] * tr[j].w0;
                    const double x1 = (double)a1[j] * tr[j].w1;
                    const float blend = (float)(x0 + x1);
                    tail_boot[g + j] = tr[j].mode == 0 ? 0.f : (tr[j].mode == 1 ? a0[j] : blend);
                }
            }
        }
        // one copy of the term code for all of a wave's heads (unrolled over j it is 8-16 copies, ~60 KB of instructions
        // that sixteen waves walk at different places: the instruction cache thrashes); the per-head registers are
        // then picked by uniform selects
        float res[MAXR];
#pragma unroll
        for (int j = 0; j < MAXR; ++j) res[j] = 0.f;
#pragma unroll 1
        for (int j = 0; j < MAXR; ++j) {
            const int k = j * kColumnWaves + wave;
            float total = 0.f;
            float tb = 0.f;
#pragma unroll
            for (int jj = 0; jj < MAXR; ++jj) tb = jj == j ? tail_boot[jj] : tb;
            if (!((kz >> j) & 1u) && !(p.skip & 2)) {
                // one term: boot from the pair (V[t+n, i0], V[t+n, i0+1]) - a two-column plan always names neighbours
                auto term = [&](const MainRec &r, float2 e, float2 v) {
                    float boot;
                    if (r.mode == 2) {  // uniform branches: the operands are already here
                        const double x0 = (double)v.x * r.w0;
                        const double x1 = (double)v.y * r.w1;
                        boot = (float)(x0 + x1);
                    } else {
                        boot = r.mode == 1 ? v.x : 0.f;
                    }
                    if (t_hi >= r.thr) boot = t >= r.thr ? tb : boot;  // uniform test, then per lane
                    const float md = boot * e.y;
                    const float tm = e.x + md;
                    total = total + tm;
                };
                auto read_sd = [&](const MainRec &r) {
                    return *reinterpret_cast<const float2 *>(sd_lane_b + r.off_sd);
                };
                auto read_pair = [&](const MainRec &r) {
                    uint32_t addr = row_t + (uint32_t)r.off_v;
                    addr = addr < (uint32_t)r.lim ? addr : (uint32_t)r.lim;
                    const float *q = reinterpret_cast<const float *>(reinterpret_cast<const char *>(vals) + addr);
                    return make_float2(q[0], q[1]);
                };
                auto lane_rec = [&](int ri) {  // record ri of this wave out of the lanes
                    MainRec r;
                    uint32_t a = 0, b = 0, w0l = 0, w0h = 0, w1l = 0, w1h = 0;
                    const int ln = ri & (kWave - 1);
#pragma unroll
                    for (int s2 = 0; s2 < NSET; ++s2) {
                        if (NSET == 1 || (ri >> 6) == s2) {  // uniform
                            a = __builtin_amdgcn_readlane(ra[s2], ln), b = __builtin_amdgcn_readlane(rb[s2], ln);
                            w0l = __builtin_amdgcn_readlane(rw0l[s2], ln), w0h = __builtin_amdgcn_readlane(rw0h[s2], ln);
                            w1l = __builtin_amdgcn_readlane(rw1l[s2], ln), w1h = __builtin_amdgcn_readlane(rw1h[s2], ln);
                        }
                    }
                    r.off_sd = (int32_t)((a & 0xffffu) << 3);
                    r.thr = (int32_t)(a >> 16);
                    r.off_v = (int32_t)(b & 0xffffffu);
                    r.mode = (int32_t)(b >> 24);
                    r.lim = r.off_v + r.thr * VS * 4;
                    r.w0 = __hiloint2double((int)w0h, (int)w0l);
                    r.w1 = __hiloint2double((int)w1h, (int)w1l);
                    return r;
                };
                int c = 0;
                if (lane_recs) {
                    for (; c + 4 <= C; c += 4) {
                        MainRec r[4];
                        float2 e[4], v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) r[u] = lane_rec(j * C + c + u);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            e[u] = read_sd(r[u]);
                            v[u] = read_pair(r[u]);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) term(r[u], e[u], v[u]);
                    }
                    for (; c < C; ++c) {
                        const MainRec r = lane_rec(j * C + c);
                        term(r, read_sd(r), read_pair(r));
                    }
                }
                const MainRec *mk = mrec + (size_t)k * C;  // more than kLaneC samples per head: records by scalar loads
                for (; c < C; ++c) {
                    const MainRec r = mk[c];
                    term(r, read_sd(r), read_pair(r));
                }
                total = total * p.inv_c;
            }
#pragma unroll
            for (int jj = 0; jj < MAXR; ++jj) res[jj] = jj == j ? total : res[jj];
        }
        __syncthreads();  // every wave is done reading sd: the region becomes the output tile
#pragma unroll
        for (int j = 0; j < MAXR; ++j) {
            const int k = j * kColumnWaves + wave;
            if (k < K && lane_on) tile[lane * KS + k] = res[j];
        }
        __syncthreads();
        const int rows = (p.skip & 4) ? 0 : (N - t0) < T ? (N - t0) : T;
        for (int q = tid; q < rows * K; q += kColumnThreads) {
            const int tl = q / K, k = q - tl * K;
            out[((size_t)(t0 + tl) * A + a) * K + k] = tile[tl * KS + k];
        }
        __syncthreads();
    }
}

template <int MAXR>
int launch_column(const ColumnLds &L, hipStream_t st, const float *rewards, const uint8_t *dones, const float *values,
                  const int32_t *nd_of_n, const MainRec *mrec, const LaneRec *lrec, const TailRec *trec,
                  const uint8_t *k_zero, float *out, ColumnArgs args)
{
    auto kern = tvf_column_kernel<MAXR>;
    // timing aid (tools/tvf_phases.sh, a -DPPO_TUNE_TIMING_AIDS build): bit 0 skips the walk, 1 the terms, 2 the stores -
    // results are then garbage; the shipped library has no such switch
#ifdef PPO_TUNE_TIMING_AIDS
    static const int skip = getenv("PPO_AMD_TVF_SKIP") ? atoi(getenv("PPO_AMD_TVF_SKIP")) : 0;
#else
    constexpr int skip = 0;
#endif
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)kLdsLimit);
        if (e != hipSuccess) return fail(PPO_E_HIP, "tvf_column: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    args.T = L.T;
    args.PAD = L.PAD;
    args.rw_off = (int)L.rw_off;
    args.step_off = (int)L.step_off;
    args.need_off = (int)L.need_off;
    args.sd_off = (int)L.sd_off;
    args.skip = skip;
    hipLaunchKernelGGL(kern, dim3(args.A), dim3(kColumnThreads), L.bytes, st, rewards, dones, values, nd_of_n, mrec, lrec,
                       trec, k_zero, out, args);
    return check_launch("tvf_column_kernel");
}

}  // namespace
}  // namespace ppo

extern "C" size_t ppo_tvf_returns_workspace_bytes(int N, int A, int ND, int K, int C)
{
    // the larger of: the packed plan records of the column kernel; the [nd][t, a] S / D cache of the fallback kernels
    const size_t records = (size_t)K * C * (48 + 24) + (size_t)K * (N + 1) * 32 + 32;
    const size_t cache = (size_t)2 * N * A * (ND < 1 ? 1 : ND) * sizeof(float);
    return records > cache ? records : cache;
}

extern "C" int ppo_tvf_returns_f32(const float *rewards, const uint8_t *dones, const float *value_samples, int N, int A,
                                   int V, int K, int C, double gamma, const int32_t *n_eff, const int32_t *nd_index,
                                   const int32_t *nd_of_n, int max_n, int ND, const int32_t *main_plan,
                                   const double *main_w, const int32_t *tail_plan, const double *tail_w,
                                   const uint8_t *k_zero, void *workspace, size_t workspace_bytes, float *out,
                                   void *stream)
{
    using namespace ppo;
    if (N <= 0 || A <= 0 || V <= 0 || K <= 0 || C <= 0 || max_n < 1 || max_n > N || ND < 1)
        return fail(PPO_E_INVALID, "ppo_tvf_returns_f32: bad shape (N=%d A=%d V=%d K=%d C=%d max_n=%d ND=%d)", N, A, V,
                    K, C, max_n, ND);
    if (!aligned(workspace, 32)) return fail(PPO_E_INVALID, "ppo_tvf_returns_f32: workspace must be 32-byte aligned");
    if (!rewards || !dones || !value_samples || !n_eff || !nd_index || !nd_of_n || !main_plan || !main_w || !tail_plan ||
        !tail_w || !k_zero || !workspace || !out)
        return fail(PPO_E_INVALID, "ppo_tvf_returns_f32: null pointer");
    if (workspace_bytes < ppo_tvf_returns_workspace_bytes(N, A, ND, K, C))
        return fail(PPO_E_INVALID, "ppo_tvf_returns_f32: workspace too small");
    hipStream_t st = as_stream(stream);
    // the column-resident kernel whenever a column (+ a chunk of the S/D table) fits in a CU's LDS
    static const bool column_ok = !(getenv("PPO_AMD_TVF_COLUMN") && atoi(getenv("PPO_AMD_TVF_COLUMN")) == 0);
    if (column_ok && K <= 16 * kColumnWaves) {
        for (int T = 64; T >= 16; T >>= 1) {
            const ColumnLds L = column_lds(N, V, K, ND, max_n, T);
            if (L.bytes > kLdsLimit) continue;
            // the lane-resident plan records hold nd * T and N - n in 16 bits each; a column that fits in LDS is far
            // inside both (sd alone would be 512 KB, vals 256 KB) - refuse rather than wrap if that ever changes
            if ((size_t)ND * T >= 65536 || N >= 65536) break;
            ColumnArgs ca{};
            ca.gamma = gamma;
            ca.N = N, ca.A = A, ca.V = V, ca.K = K, ca.C = C, ca.ND = ND, ca.max_n = max_n;
            ca.inv_c = (float)(1.0 / C);
            const int n_main = K * C, n_tail = K * (N + 1);
            MainRec *mrec = static_cast<MainRec *>(workspace);
            LaneRec *lrec = reinterpret_cast<LaneRec *>(mrec + n_main);
            TailRec *trec = reinterpret_cast<TailRec *>(
                (reinterpret_cast<uintptr_t>(lrec + n_main) + 31) & ~(uintptr_t)31);
            hipLaunchKernelGGL(tvf_pack_kernel, dim3((n_main + n_tail + 255) / 256), dim3(256), 0, st, n_eff, nd_index,
                               main_plan, main_w, tail_plan, tail_w, n_main, n_tail, N, L.VS, L.T, mrec, lrec, trec);
            int rc = check_launch("tvf_pack_kernel");
            if (rc) return rc;
            return K <= 8 * kColumnWaves
                       ? launch_column<8>(L, st, rewards, dones, value_samples, nd_of_n, mrec, lrec, trec, k_zero, out, ca)
                       : launch_column<16>(L, st, rewards, dones, value_samples, nd_of_n, mrec, lrec, trec, k_zero, out, ca);
        }
    }
    float *cS = static_cast<float *>(workspace);
    float *cD = cS + (size_t)N * A * ND;
    hipLaunchKernelGGL(tvf_prefix_kernel, dim3((N * A + 255) / 256), dim3(256), 0, st, rewards, dones, N, A, gamma,
                       nd_of_n, max_n, ND, cS, cD);
    int rc = check_launch("tvf_prefix_kernel");
    if (rc) return rc;
    const int64_t total = (int64_t)N * A * K;
    hipLaunchKernelGGL(tvf_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, value_samples, N, A, V,
                       K, C, ND, n_eff, nd_index, main_plan, main_w, tail_plan, tail_w, k_zero, (float)(1.0 / C), cS, cD,
                       out);
    return check_launch("tvf_gather_kernel");
}

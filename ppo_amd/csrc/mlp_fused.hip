// The MLP policy / value networks (StandardMLP + DualHeadNet heads, rl/models.py:148-169, 364-384, 433-508) as THREE
// launches per training minibatch instead of ~14, and ONE per inference forward instead of 5 (gfx950).
//
//   x [B, F] -> fc1 -> tanh -> fc2 -> act (tanh | relu) -> heads [B, NH]          (F = 377, H = 256, NH = 163 for the
//                                                                                    Humanoid + 128 TVF heads config)
//
// At a 256-row minibatch every product here is 20 - 50 MFLOP: the op-by-op path spent 8 - 20 us per launch on launch
// latency, not arithmetic (profiles/r03d_bench_line_humanoid.json: 56 launches per optimiser step at <= 0.03 of a
// roofline each).  Layout of the fused path:
//
//   A  mlp_rows_kernel      one 16-wave workgroup per tile of 16 rows; the tile's activations never leave LDS:
//                           x -> a1 -> hact -> heads -> [loss: the per-sample bodies of loss_rows.h, one wave per row]
//                           -> dheads -> dh (through act') -> da1 (through tanh').  Weights stream from L2 (every tile
//                           reads all of them: 0.8 MB forward, 0.43 MB backward).  Inference stops after the heads.
//                           What the weight-gradient launch needs (a1, hact, dheads, dh, da1) is written once, coalesced.
//   B  mlp_wgrad_kernel     dW = dY^T X over the whole minibatch, one wave per 16x16 tile of a weight matrix (K = rows, in
//                           fixed order: deterministic, no slabs, no atomics), the bias gradients (column sums) and
//                           log_std's, and per-workgroup partial sums of g^2 in the optimiser's workspace.
//   C  adam_kernel          optim.hip, re-reducing B's partials (ppo_adam_step_presummed_f32): no separate sum-of-squares
//                           launch.
//
// MFMA mapping (v_mfma_f32_16x16x4_f32, exact f32).  A layer is D[n, row] = sum_k W[n, k] X[row, k]: M = output feature
// (A operand = 16 rows of W), N = the tile's 16 rows (B operand = activations in LDS), K = input feature.  A lane fetches
// FOUR consecutive k of its W row with one 16-byte load and four consecutive k of its activation row with one
// ds_read_b128, and spends them on four MFMAs: K slot g of MFMA i carries k0 + 4 g + i - a sum over k does not care which
// slot a term rides in, as long as A and B agree.  Activation rows are padded to a stride of 4 (mod 32) floats, which
// makes those 16-byte reads (8 lanes per pass) conflict-free.  The result lands as lane = row, 4 consecutive features:
// bias + activation, then one 16-byte LDS store in [row][feature] order - the next layer's B operand as it stands.
#include "common.h"
#include "loss_rows.h"
#include "mfma.h"

namespace ppo {
namespace {

#ifndef PPO_TUNE_MLP_STOP  // timing aid (tools/mlp_phases.sh builds): leave the rows kernel after phase n; 0 = the product
#define PPO_TUNE_MLP_STOP 0
#endif
constexpr int kRows = 16;        // rows of a tile
constexpr int kMlpWaves = 16;
constexpr int kMlpThreads = kMlpWaves * 64;

enum { MLP_LOSS_NONE = 0, MLP_LOSS_VALUE, MLP_LOSS_DISTIL, MLP_LOSS_GAUSS, MLP_LOSS_PPO };
enum { ACT_NONE = 0, ACT_TANH, ACT_RELU };

struct MlpArgs {
    const float *x;          // [*, F]
    const int32_t *x_index;  // nullable: minibatch row b -> row of x (x is the whole batch); null: x holds the B rows
    const int32_t *index;    // nullable: minibatch row b -> row of the per-sample loss arrays
    int B, F, H, NH, act;    // act: the encoder activation behind fc2 (ACT_TANH | ACT_RELU); fc1 is followed by tanh
    const float *w1, *b1, *w2, *b2, *wh, *bh;  // bh nullable (head_bias off)
    float *heads;            // [B, NH], nullable when training
    float *h_pre, *hact_out; // [B, H] each, nullable (include_features / the weight-gradient launch)
    float *a1_out, *dheads, *dh, *da1;         // training: [B, H], [B, NH], [B, H], [B, H]
    float *x_out;            // training with an index: the gathered rows [B, F] (the weight-gradient launch reads them)
    float *dlog_std_rows;    // gaussian loss: [B, nA]
    int loss;
    ValueLossP lv;
    DistilLossP ld;
    GaussLossP lg;
    PpoLossP lp;
    int ldx, ldh, ldo;       // LDS row strides (floats), each = 4 (mod 32)
    int lds_floats;          // floats of LDS the tiles take; 64 per wave follow (the sink of the cache-warming requests)
};

__host__ __device__ inline int lds_stride(int n) { int v = (n + 15) / 16 * 16; return v + ((4 - v % 32) + 32) % 32; }

// Every phase of the rows kernel is a chain of dependent memory round trips (weights -> MFMAs -> bias -> LDS -> barrier),
// and at one 16-row tile per CU nothing else hides them: measured per layer 6 - 10 us for 1.5 us of MFMA work, 15 us for
// the loss (index -> target rows), 16 us for a store phase at the end (tools/mlp_phases.sh).  So: a layer's first blocks
// of weights and its bias are REQUESTED while the previous phase still runs (the ring below survives the barrier in
// registers), the loss's rows are touched at kernel start, and results leave from the epilogues as 16-byte stores.
constexpr int kFwdPF = 6;  // blocks of a forward layer's weights in flight per wave (16 bytes per lane each)
constexpr int kBwdPF = 4;  // ... of a backward layer's (four 4-byte loads per lane each); 8 / 6 spill at 16 waves per workgroup

struct FwdRing {
    float4 wq[kFwdPF];
    float bias[4];
};
struct BwdRing {
    float wq[kBwdPF][4];
};

// the requests of a wave's first tile (tile index = wave) of a forward layer: blocks 0 .. PF-1 and the bias quad
__device__ __forceinline__ void fwd_issue(const float *__restrict__ W, const float *__restrict__ bias, int N, int K, int wave,
                                          int lane, FwdRing &ring)
{
    const int l15 = lane & 15, g = lane >> 4;
    const int K16 = (K + 15) / 16;
    const __amdgpu_buffer_rsrc_t wb = buffer_of(W), bb = buffer_of(bias, bias != nullptr);
    const int n0 = wave * 16;
    const bool live = n0 < N;
    const int nrow = n0 + l15 < N ? n0 + l15 : N - 1;
    const int wbase = (nrow * K + 4 * g) * 4;
#pragma unroll
    for (int u = 0; u < kFwdPF; ++u) ring.wq[u] = buffer_f32x4(wb, live && u < K16 ? wbase + u * 64 : kOutside);
#pragma unroll
    for (int r = 0; r < 4; ++r) ring.bias[r] = buffer_f32(bb, live && n0 + 4 * g + r < N ? (n0 + 4 * g + r) * 4 : kOutside);
}

__device__ __forceinline__ void bwd_issue(const float *__restrict__ W, int N, int C, int wave, int lane, BwdRing &ring)
{
    const int l15 = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t wb = buffer_of(W);
    const int c0 = wave * 16;
    const bool live = c0 < C;
    const int c = c0 + l15 < C ? c0 + l15 : C - 1;
#pragma unroll
    for (int u = 0; u < kBwdPF; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = u * 16 + 4 * g + i;
            ring.wq[u][i] = buffer_f32(wb, live && o < N ? (o * C + c) * 4 : kOutside);
        }
}

// ---- forward layer: out[row][n] = act(sum_k W[n][k] in[row][k] + bias[n]) for the tile's 16 rows, n < N.
// `ring`: the first tile's first blocks and bias, requested by fwd_issue before the barrier in front of this call.
// `next()`: called when this wave's last K loop is done (its ring registers are free): requests the next phase's weights.
// pre_s (nullable): the values before the activation, same layout as out_s.
template <int ACT, class Next>
__device__ __forceinline__ void fwd_layer(const float *__restrict__ W, const float *__restrict__ bias, int N, int K,
                                          const float *in_s, int ldi, float *out_s, int ldo, float *pre_s, int wave, int lane,
                                          FwdRing &ring, Next next)
{
    const int l15 = lane & 15, g = lane >> 4;
    const int K16 = (K + 15) / 16;
    // a 16-byte read that runs past the end of a row continues into the next row - or, for the last row, into whatever
    // follows the matrix in the flat parameter buffer (at most 12 bytes: finite numbers): those k meet the zeros of the
    // activation rows' padding
    const __amdgpu_buffer_rsrc_t wb = buffer_of(W), bb = buffer_of(bias, bias != nullptr);
    for (int t = wave; t * 16 < N; t += kMlpWaves) {
        const int n0 = t * 16;
        const int nrow = n0 + l15 < N ? n0 + l15 : N - 1;
        const int wbase = (nrow * K + 4 * g) * 4;
        const float *xrow = in_s + l15 * ldi + 4 * g;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (t != wave) {  // (only nets wider than 16 tiles: the first tile's requests were made ahead of time)
#pragma unroll
            for (int u = 0; u < kFwdPF; ++u) ring.wq[u] = buffer_f32x4(wb, u < K16 ? wbase + u * 64 : kOutside);
#pragma unroll
            for (int r = 0; r < 4; ++r) ring.bias[r] = buffer_f32(bb, n0 + 4 * g + r < N ? (n0 + 4 * g + r) * 4 : kOutside);
        }
        // a ring of PF blocks in flight, refilled one by one: block kb's registers take block kb + PF's load as soon as its
        // four MFMAs have consumed them
        for (int kb0 = 0; kb0 < K16; kb0 += kFwdPF) {
#pragma unroll
            for (int u = 0; u < kFwdPF; ++u) {
                const int kb = kb0 + u;
                if (kb < K16) {
                    const float4 xv = *reinterpret_cast<const float4 *>(xrow + kb * 16);
                    const float4 wv = ring.wq[u];
                    acc = mfma16(wv.x, xv.x, acc);
                    acc = mfma16(wv.y, xv.y, acc);
                    acc = mfma16(wv.z, xv.z, acc);
                    acc = mfma16(wv.w, xv.w, acc);
                }
                ring.wq[u] = buffer_f32x4(wb, kb + kFwdPF < K16 ? wbase + (kb + kFwdPF) * 64 : kOutside);
            }
        }
        // lane: row l15, features n0 + 4 g .. + 3
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = n0 + 4 * g + r < N ? acc[r] + ring.bias[r] : 0.f;
        if (t + kMlpWaves >= (N + 15) / 16) next();  // after the bias has been consumed: `next` may overwrite the ring
        if (pre_s) *reinterpret_cast<float4 *>(pre_s + l15 * ldo + n0 + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (ACT == ACT_TANH) v[r] = tanhf(v[r]);
            if (ACT == ACT_RELU) v[r] = fmaxf(v[r], 0.f);
        }
        *reinterpret_cast<float4 *>(out_s + l15 * ldo + n0 + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (wave * 16 >= N) next();  // waves without a tile in this layer still take part in the next one
}

// ---- backward-data layer: din[row][c] = (sum_o dout[row][o] W[o][c]) * act'(saved[row][c]), c < C, o < N
template <class Next>
__device__ __forceinline__ void bwd_layer(const float *__restrict__ W, int N, int C, int act, const float *dout_s, int ldd,
                                          const float *saved_s, float *din_s, int ldc, int wave, int lane, BwdRing &ring,
                                          Next next)
{
#pragma clang fp contract(off)  // torch's tanh_backward rounds t * t before the subtraction
    const int l15 = lane & 15, g = lane >> 4;
    const int N16 = (N + 15) / 16;
    const __amdgpu_buffer_rsrc_t wb = buffer_of(W);
    for (int t = wave; t * 16 < C; t += kMlpWaves) {
        const int c0 = t * 16;
        const int c = c0 + l15 < C ? c0 + l15 : C - 1;
        const float *drow = dout_s + l15 * ldd + 4 * g;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        auto load = [&](float (&w)[4], int ob) {  // W[ob * 16 + 4 g + i][c], i = 0..3: rows past N read zeros
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = ob * 16 + 4 * g + i;
                w[i] = buffer_f32(wb, o < N ? (o * C + c) * 4 : kOutside);
            }
        };
        if (t != wave) {
#pragma unroll
            for (int u = 0; u < kBwdPF; ++u) load(ring.wq[u], u);
        }
        for (int ob0 = 0; ob0 < N16; ob0 += kBwdPF) {
#pragma unroll
            for (int u = 0; u < kBwdPF; ++u) {
                const int ob = ob0 + u;
                if (ob < N16) {
                    const float4 dv = *reinterpret_cast<const float4 *>(drow + ob * 16);
                    acc = mfma16(ring.wq[u][0], dv.x, acc);
                    acc = mfma16(ring.wq[u][1], dv.y, acc);
                    acc = mfma16(ring.wq[u][2], dv.z, acc);
                    acc = mfma16(ring.wq[u][3], dv.w, acc);
                }
                load(ring.wq[u], ob + kBwdPF);
            }
        }
        if (t + kMlpWaves >= (C + 15) / 16) next();
        const float4 sv = *reinterpret_cast<const float4 *>(saved_s + l15 * ldc + c0 + 4 * g);
        const float sq[4] = {sv.x, sv.y, sv.z, sv.w};
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float d = acc[r];
            if (act == ACT_TANH) {
                const float tt = sq[r] * sq[r];
                d = d * (1.f - tt);
            } else if (act == ACT_RELU) {
                d = sq[r] > 0.f ? d : 0.f;
            }
            v[r] = c0 + 4 * g + r < C ? d : 0.f;
        }
        *reinterpret_cast<float4 *>(din_s + l15 * ldc + c0 + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (wave * 16 >= C) next();
}

// Rows of a finished LDS tile to HBM, one wave per row, lanes along the row: whole cache lines per store instruction
// (from the MFMA epilogues a store instruction would scatter sixty-four 16-byte pieces over 16 rows - partial-line writes,
// measured at ~10 us per array).  Called right after the barrier that completes the tile; nothing waits for these stores
// (lds_barrier does not), so they drain under the next phase.
__device__ __forceinline__ void store_rows(float *dst, const float *src_s, int lds, int n, int r0, int rows, int wave, int lane)
{
#ifdef PPO_TUNE_MLP_NOSTORE  // timing aid: no result leaves the kernel
    return;
#endif
    if (!dst) return;
    for (int r = wave; r < rows; r += kMlpWaves) {
        float *d = dst + (size_t)(r0 + r) * n;
        const float *sr = src_s + r * lds;
        if ((n & 3) == 0) {
            for (int c = lane * 4; c < n; c += 256) *reinterpret_cast<float4 *>(d + c) = *reinterpret_cast<const float4 *>(sr + c);
        } else {
            for (int c = lane; c < n; c += 64) d[c] = sr[c];
        }
    }
}

// The phases hand their results over through LDS only, so the barrier between them needs the LDS traffic drained
// (lgkmcnt) and nothing else.  __syncthreads() also carries a workgroup-scope fence, i.e. s_waitcnt vmcnt(0): every
// barrier then waited for the epilogues' HBM stores to be acknowledged and for the next layer's weight requests - the very
// round trips the requests were issued early to hide.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(kMlpThreads) void mlp_rows_kernel(const MlpArgs a)
{
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * kRows;
    const int rows = a.B - r0 < kRows ? a.B - r0 : kRows;
    const int F = a.F, H = a.H, NH = a.NH, ldx = a.ldx, ldh = a.ldh, ldo = a.ldo;
    float *xs = smem;
    float *a1s = xs + kRows * ldx;
    float *has = a1s + kRows * ldh;   // fc2's output after the encoder activation
    float *os = has + kRows * ldh;
    float *dos = os + kRows * ldo;    // training only from here on
    float *dhs = dos + kRows * ldo;
    float *da1s = dhs + kRows * ldh;
    const bool train = a.loss != MLP_LOSS_NONE;

    // ---- touch every cache line the kernel will need later, all loads independent: the optimiser step just rewrote the
    // weights, so this XCD's L2 does not hold them, and the dependent chains further down (a layer's weight ring, the
    // loss's index -> target rows) would each pay HBM round trips instead of L2 hits.
    // Fire-and-forget: each touch is a 4-byte-per-lane LDS-DMA request into a dead LDS word per lane (no destination
    // register, so nothing in the program ever waits for it; summing loaded values instead made every touch a full
    // s_waitcnt vmcnt(0) round trip, ~6 of them in a row before the first layer could start).
    {
        using gptr_t = const __attribute__((address_space(1))) void *;
        using lptr_t = __attribute__((address_space(3))) void *;
        float *sink = smem + a.lds_floats + wave * 64;  // 64 floats per wave behind the tiles
        auto touch = [&](const float *w, size_t n) {
            for (size_t i = (size_t)tid * 32; i < n; i += (size_t)kMlpThreads * 32)
                __builtin_amdgcn_global_load_lds((gptr_t)(w + i), (lptr_t)sink, 4, 0, 0);
        };
        touch(a.w1, (size_t)H * F);
        touch(a.w2, (size_t)H * H);
        touch(a.wh, (size_t)NH * H);
    }
    FwdRing fr;
    BwdRing br;
    fwd_issue(a.w1, a.b1, H, F, wave, lane, fr);
    // ---- the tile's rows of x (through the minibatch index), zero beyond F and beyond the batch: one wave per row, the
    // row's loads unconditional (clamped address, then a select) and all in flight together
    for (int r = wave; r < kRows; r += kMlpWaves) {
        const int b = r0 + (r < rows ? r : 0);
        const float *src = a.x + (size_t)(a.x_index ? a.x_index[b] : b) * F;
        constexpr int XU = 8;  // 8 x 64 = 512 features per pass
        for (int k0 = 0; k0 < ldx; k0 += 64 * XU) {
            float v[XU];
#pragma unroll
            for (int u = 0; u < XU; ++u) {
                const int k = k0 + u * 64 + lane;
                v[u] = src[k < F ? k : F - 1];
            }
#pragma unroll
            for (int u = 0; u < XU; ++u) {
                const int k = k0 + u * 64 + lane;
                if (k < ldx) xs[r * ldx + k] = (r < rows && k < F) ? v[u] : 0.f;
            }
        }
    }
    if (train)
        for (int i = tid; i < kRows * ldo; i += kMlpThreads) dos[i] = 0.f;  // columns past NH stay zero (K padding of dh)
    lds_barrier();
    if (PPO_TUNE_MLP_STOP == 1) return;
    store_rows(a.x_out, xs, ldx, F, r0, rows, wave, lane);
    fwd_layer<ACT_TANH>(a.w1, a.b1, H, F, xs, ldx, a1s, ldh, nullptr, wave, lane, fr,
                        [&] { fwd_issue(a.w2, a.b2, H, H, wave, lane, fr); });
    lds_barrier();
    if (PPO_TUNE_MLP_STOP == 2) return;
    if (train) store_rows(a.a1_out, a1s, ldh, H, r0, rows, wave, lane);
    auto issue_heads = [&] { fwd_issue(a.wh, a.bh, NH, H, wave, lane, fr); };
    float *hps = a.h_pre ? dos : nullptr;  // (inference: the pre-activation tile borrows LDS the training phases own)
    if (a.act == ACT_TANH) fwd_layer<ACT_TANH>(a.w2, a.b2, H, H, a1s, ldh, has, ldh, hps, wave, lane, fr, issue_heads);
    else fwd_layer<ACT_RELU>(a.w2, a.b2, H, H, a1s, ldh, has, ldh, hps, wave, lane, fr, issue_heads);
    lds_barrier();
    if (PPO_TUNE_MLP_STOP == 3) return;
    store_rows(a.hact_out, has, ldh, H, r0, rows, wave, lane);
    store_rows(a.h_pre, hps, ldh, H, r0, rows, wave, lane);
    fwd_layer<ACT_NONE>(a.wh, a.bh, NH, H, has, ldh, os, ldo, nullptr, wave, lane, fr, [&] {
        if (train) bwd_issue(a.wh, NH, H, wave, lane, br);  // the first backward layer's weights ride through the loss
    });
    lds_barrier();
    if (PPO_TUNE_MLP_STOP == 4) return;
    store_rows(a.heads, os, ldo, NH, r0, rows, wave, lane);
    if (!train) return;

    // ---- loss: one wave per row (the bodies of the stand-alone loss kernels)
    if (wave < rows) {
        const int b = r0 + wave, sb = a.index ? a.index[b] : b;
        const float *z = os + wave * ldo;
        float *dz = dos + wave * ldo;
        if (a.loss == MLP_LOSS_VALUE) value_loss_row(a.lv, z, dz, b, sb, lane);
        else if (a.loss == MLP_LOSS_DISTIL) distil_loss_row(a.ld, z, dz, b, sb, lane);
        else if (a.loss == MLP_LOSS_GAUSS)
            gaussian_loss_row(a.lg, z, dz, a.dlog_std_rows ? a.dlog_std_rows + (size_t)b * a.lg.nA : nullptr, b, sb, lane);
        else if (lane == 0) ppo_loss_row<0>(a.lp, z, dz, b, sb);
    }
    lds_barrier();
    if (PPO_TUNE_MLP_STOP == 5) return;
    store_rows(a.dheads, dos, ldo, NH, r0, rows, wave, lane);
    // ---- backward-data: dh = (dheads Wh) * act'(h);  da1 = (dh W2) * tanh'(a1)
    bwd_layer(a.wh, NH, H, a.act, dos, ldo, has, dhs, ldh, wave, lane, br, [&] { bwd_issue(a.w2, H, H, wave, lane, br); });
    lds_barrier();
    if (PPO_TUNE_MLP_STOP == 6) return;
    store_rows(a.dh, dhs, ldh, H, r0, rows, wave, lane);
    bwd_layer(a.w2, H, H, ACT_TANH, dhs, ldh, a1s, da1s, ldh, wave, lane, br, [] {});
    lds_barrier();
    store_rows(a.da1, da1s, ldh, H, r0, rows, wave, lane);
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradients of the three layers in one launch.  dW[o][k] = sum_rows dY[row][o] X[row][k]: M = o (A = dY^T), N = k
// (B = X), K = rows in blocks of 16, ascending: one wave owns a 16x16 tile of one dW from the first row to the last, so the
// sum has a fixed order.  Lanes read dY[row][o0 + l15] and X[row][k0 + l15]: 64-byte segments per 16 lanes.  X of the first
// layer is the copy of the minibatch's rows the rows kernel wrote (it gathered them through the index once).
struct WgradProblem {
    const float *dy;  // [B, N]
    const float *x;   // [B, K]
    float *dw;        // [N, K]
    float *db;        // [N] nullable
    int N, K, tiles_k, tile0;  // tile0: first linear tile id of this problem
};
struct MlpWgradArgs {
    WgradProblem p[3];
    int B, n_tiles;
    const float *dlog_std_rows;  // [B, nA] nullable (gaussian policy phase)
    float *dlog_std;             // [nA] nullable: written (zeros when there are no rows) so that a stale gradient never leaks
    int nA;
    float *partials;             // [gridDim.x] sums of g^2, the optimiser's workspace
    int tile_wgs;                // workgroups that walk tiles; the last workgroup takes the column sums
    const float *stats;          // [B, n_stats] per-sample loss statistics (nullable) ...
    float *stat_sums;            // ... column-summed into this row (added to it when stat_accumulate)
    int n_stats, stat_accumulate;
};

__global__ __launch_bounds__(256) void mlp_wgrad_kernel(const MlpWgradArgs a)
{
    __shared__ float s_part[4];
    __shared__ float s_cols[8][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    float sq = 0.f;
    if ((int)blockIdx.x < a.tile_wgs) {
        for (int tile = blockIdx.x * 4 + wave; tile < a.n_tiles; tile += a.tile_wgs * 4) {
            const WgradProblem &p = tile >= a.p[2].tile0 ? a.p[2] : (tile >= a.p[1].tile0 ? a.p[1] : a.p[0]);
            const int lt = tile - p.tile0;
            const int o0 = (lt / p.tiles_k) * 16, k0 = (lt % p.tiles_k) * 16;
            const int N = p.N, K = p.K;
            // range-checked buffer reads (common.h): written as "load from a clamped address, then select" the compiler
            // sank every load into a branch with s_waitcnt vmcnt(0) behind it - 150 full waits per tile, 39 us per launch
            const bool o_ok = o0 + l15 < N, k_ok = k0 + l15 < K;
            const __amdgpu_buffer_rsrc_t dyb = buffer_of(p.dy), xb = buffer_of(p.x);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            float asum = 0.f;  // this lane's share of the bias gradient: sum over its rows of dY[row][o0 + l15]
            constexpr int PF = 4;  // row blocks per group, two groups in flight
            float av0[PF][4], bv0[PF][4], av1[PF][4], bv1[PF][4];
            auto load = [&](float (&aa)[PF][4], float (&bb)[PF][4], int rb0) {
#pragma unroll
                for (int u = 0; u < PF; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = (rb0 + u) * 16 + 4 * g + i;
                        const bool in = row < a.B;
                        aa[u][i] = buffer_f32(dyb, in && o_ok ? (row * N + o0 + l15) * 4 : kOutside);
                        bb[u][i] = buffer_f32(xb, in && k_ok ? (row * K + k0 + l15) * 4 : kOutside);
                    }
            };
            auto compute = [&](const float (&aa)[PF][4], const float (&bb)[PF][4]) {
#pragma unroll
                for (int u = 0; u < PF; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc = mfma16(aa[u][i], bb[u][i], acc);
                        asum += aa[u][i];
                    }
            };
            const int RB = (a.B + 15) / 16;
            load(av0, bv0, 0);
            for (int rb0 = 0; rb0 < RB; rb0 += 2 * PF) {
                load(av1, bv1, rb0 + PF);
                compute(av0, bv0);
                load(av0, bv0, rb0 + 2 * PF);
                compute(av1, bv1);
            }
            // lane: dW[o0 + 4 g + r][k0 + l15]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + 4 * g + r;
                if (o < N && k_ok) {
                    p.dw[(size_t)o * K + k0 + l15] = acc[r];
                    sq += acc[r] * acc[r];
                }
            }
            // the bias gradient falls out of the A operand: lanes (l15, g = 0..3) hold the four row-quarters of column o
            asum += __shfl_xor(asum, 16, 64);
            asum += __shfl_xor(asum, 32, 64);
            if (k0 == 0 && p.db && g == 0 && o_ok) {
                p.db[o0 + l15] = asum;
                sq += asum * asum;
            }
        }
    } else {
        // the narrow column sums, one workgroup: log_std's gradient rows [B, nA] and the loss statistics [B, n_stats].
        // thread = (row eighth, column): 8 partial sums per column, folded in a fixed order
        const int col = tid & 31, part = tid >> 5;
        const int lo = (a.B * part) / 8, hi = (a.B * (part + 1)) / 8;
        for (int pass = 0; pass < 2; ++pass) {
            const float *src = pass == 0 ? a.dlog_std_rows : a.stats;
            const int nc = pass == 0 ? a.nA : a.n_stats;
            float *dst = pass == 0 ? a.dlog_std : a.stat_sums;
            if (!dst) continue;
            for (int c0 = 0; c0 < nc; c0 += 32) {
                const int c = c0 + col;
                float s = 0.f;
                if (src && c < nc) {
                    // 16 loads in flight, then their sum in row order (a plain `s += src[...]` loop was one memory round
                    // trip per row: 39 us for 32 rows, the whole launch waiting on this one workgroup)
                    for (int rr = lo; rr < hi; rr += 16) {
                        float v[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(rr + u < hi ? rr + u : hi - 1) * nc + c];
#pragma unroll
                        for (int u = 0; u < 16; ++u) s += rr + u < hi ? v[u] : 0.f;
                    }
                }
                s_cols[part][col] = s;
                __syncthreads();
                if (part == 0 && c < nc) {
                    float t = s_cols[0][col];
#pragma unroll
                    for (int q = 1; q < 8; ++q) t += s_cols[q][col];
                    if (pass == 0) {
                        dst[c] = t;
                        sq += t * t;
                    } else {
                        dst[c] = (a.stat_accumulate ? dst[c] : 0.f) + t;  // not a gradient: stays out of the sum of squares
                    }
                }
                __syncthreads();
            }
        }
    }
    sq = wave_sum(sq);
    if (lane == 0) s_part[wave] = sq;
    __syncthreads();
    if (tid == 0) a.partials[blockIdx.x] = ((s_part[0] + s_part[1]) + s_part[2]) + s_part[3];
}

size_t mlp_lds_bytes(int F, int H, int NH, bool train)
{
    const int ldx = lds_stride(F), ldh = lds_stride(H), ldo = lds_stride(NH);
    size_t fl = (size_t)kRows * (ldx + 2 * ldh + ldo);
    fl += train ? (size_t)kRows * (ldo + 2 * ldh) : (size_t)kRows * ldh;  // inference: room for the pre-activation tile
    return (fl + kMlpWaves * 64) * 4;
}

int launch_rows(MlpArgs &a, hipStream_t st)
{
    a.ldx = lds_stride(a.F), a.ldh = lds_stride(a.H), a.ldo = lds_stride(a.NH);
    const size_t lds = mlp_lds_bytes(a.F, a.H, a.NH, a.loss != MLP_LOSS_NONE);
    a.lds_floats = (int)(lds / 4) - kMlpWaves * 64;
    if (lds > 160 * 1024) return fail(PPO_E_INVALID, "mlp_rows: a tile of %d features / %d hidden / %d heads needs %zu bytes of LDS", a.F, a.H, a.NH, lds);
    static bool ready = false;
    if (!ready) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_rows_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail(PPO_E_HIP, "mlp_rows: hipFuncSetAttribute: %s", hipGetErrorString(e));
        ready = true;
    }
    hipLaunchKernelGGL(mlp_rows_kernel, dim3((a.B + kRows - 1) / kRows), dim3(kMlpThreads), lds, st, a);
    return check_launch("mlp_rows_kernel");
}

int check_net(const char *who, const ppo_mlp_net *n, int B)
{
    if (!n) return fail(PPO_E_INVALID, "%s: null network", who);
    if (B < 0 || n->F <= 0 || n->H <= 0 || n->NH <= 0 || n->H % 16 != 0)
        return fail(PPO_E_INVALID, "%s: bad shape (B=%d F=%d H=%d NH=%d; H must be a multiple of 16)", who, B, n->F, n->H, n->NH);
    if (n->act != 1 && n->act != 2) return fail(PPO_E_INVALID, "%s: act must be 1 (tanh) or 2 (relu)", who);
    if (!n->w1 || !n->b1 || !n->w2 || !n->b2 || !n->wh) return fail(PPO_E_INVALID, "%s: null parameter", who);
    return PPO_OK;
}

void fill_net(MlpArgs &a, const ppo_mlp_net *n)
{
    a.F = n->F, a.H = n->H, a.NH = n->NH, a.act = n->act;
    a.w1 = n->w1, a.b1 = n->b1, a.w2 = n->w2, a.b2 = n->b2, a.wh = n->wh, a.bh = n->bh;
}

}  // namespace
}  // namespace ppo

extern "C" int ppo_mlp_supported(int F, int H, int NH)
{
    return F > 0 && H > 0 && NH > 0 && H % 16 == 0 && ppo::mlp_lds_bytes(F, H, NH, true) <= 160 * 1024;
}

extern "C" int ppo_mlp_forward_f32(const float *x, const ppo_mlp_net *net, const int32_t *index, int B, float *heads,
                                   float *h_pre, float *hact, void *stream)
{
    using namespace ppo;
    int rc = check_net("ppo_mlp_forward_f32", net, B);
    if (rc) return rc;
    if (B == 0) return PPO_OK;
    if (!x || !heads) return fail(PPO_E_INVALID, "ppo_mlp_forward_f32: null pointer");
    MlpArgs a{};
    fill_net(a, net);
    a.x = x, a.x_index = index, a.index = index, a.B = B, a.heads = heads, a.h_pre = h_pre, a.hact_out = hact, a.loss = MLP_LOSS_NONE;
    return launch_rows(a, as_stream(stream));
}

extern "C" size_t ppo_mlp_train_workspace_floats(int B, int F, int H, int NH) { return (size_t)B * (4 * (size_t)H + NH + (size_t)F); }

extern "C" int ppo_mlp_train_f32(const float *x, const ppo_mlp_net *net, const ppo_mlp_grads *grads, const int32_t *index,
                                 int64_t x_rows, int B, const ppo_mlp_loss *loss, float *workspace, float *heads,
                                 float *stat_sums, int n_stats, int stat_accumulate, float *partials, int *n_partials,
                                 void *stream)
{
    using namespace ppo;
    const int x_indexed = x_rows > 0;
    int rc = check_net("ppo_mlp_train_f32", net, B);
    if (rc) return rc;
    if (B == 0) {
        if (n_partials) *n_partials = 0;
        return PPO_OK;
    }
    if (!grads || !x || !loss || !workspace || !partials || !n_partials)
        return fail(PPO_E_INVALID, "ppo_mlp_train_f32: null pointer");
    if (x_indexed && !index) return fail(PPO_E_INVALID, "ppo_mlp_train_f32: x_indexed without an index");
    if (stat_sums && (n_stats <= 0 || n_stats > 256 || !loss->stats))
        return fail(PPO_E_INVALID, "ppo_mlp_train_f32: stat_sums needs loss->stats and 1 <= n_stats <= 256");
    if (!grads->dw1 || !grads->db1 || !grads->dw2 || !grads->db2 || !grads->dwh)
        return fail(PPO_E_INVALID, "ppo_mlp_train_f32: null gradient");
    const int H = net->H, NH = net->NH, F = net->F;
    MlpArgs a{};
    fill_net(a, net);
    a.x = x, a.x_index = x_indexed ? index : nullptr, a.index = index, a.B = B, a.heads = heads;
    float *ws = workspace;
    a.a1_out = ws, ws += (size_t)B * H;
    a.hact_out = ws, ws += (size_t)B * H;
    a.dh = ws, ws += (size_t)B * H;
    a.da1 = ws, ws += (size_t)B * H;
    a.dheads = ws, ws += (size_t)B * NH;
    a.x_out = x_indexed ? ws : nullptr;  // rows gathered once by the rows kernel: the weight gradients then read plain rows
    a.dlog_std_rows = nullptr;
    switch (loss->kind) {
        case PPO_MLP_LOSS_VALUE:
            if (loss->n_tvf > 0 && (loss->tvf_stride <= 0 || loss->tvf_col < 0 || loss->tvf_col + (loss->n_tvf - 1) * loss->tvf_stride >= NH))
                return fail(PPO_E_INVALID, "ppo_mlp_train_f32: TVF columns exceed the head row");
            if (loss->value_col < 0 || loss->value_col + loss->n_value_heads > NH || !(loss->tvf_keep_prob > 0.f))
                return fail(PPO_E_INVALID, "ppo_mlp_train_f32: bad value-loss arguments");
            a.loss = MLP_LOSS_VALUE;
            a.lv = ValueLossP{NH, loss->value_col, loss->returns ? loss->n_value_heads : 0, loss->n_value_heads > 0 ? loss->returns : nullptr,
                              loss->vf_coef, loss->tvf_col, loss->tvf_returns ? loss->n_tvf : 0, loss->n_tvf > 0 ? loss->tvf_stride : 1,
                              loss->n_tvf > 0 ? loss->tvf_returns : nullptr, loss->tvf_weights, loss->tvf_coef, loss->grad_scale,
                              loss->stats, loss->tvf_keep_prob, loss->seed, loss->offset};
            break;
        case PPO_MLP_LOSS_DISTIL:
            if (loss->n_actions <= 0 || loss->n_actions > kMaxActions || loss->n_pred <= 0 || loss->pred_stride <= 0 ||
                loss->pred_col < loss->n_actions || loss->pred_col + (loss->n_pred - 1) * loss->pred_stride >= NH ||
                !loss->targets || !loss->old_policy)
                return fail(PPO_E_INVALID, "ppo_mlp_train_f32: bad distil-loss arguments");
            a.loss = MLP_LOSS_DISTIL;
            a.ld = DistilLossP{NH, loss->n_actions, loss->pred_col, loss->n_pred, loss->pred_stride, loss->vector_targets,
                               loss->targets, loss->weights, loss->old_policy, loss->log_std, loss->beta, loss->grad_scale, loss->stats};
            break;
        case PPO_MLP_LOSS_GAUSSIAN:
            if (loss->n_actions <= 0 || loss->n_value_heads < 0 || NH < loss->n_actions + loss->n_value_heads || !loss->actions_f ||
                !loss->old_log_pac || !loss->advantages || !loss->log_std || (loss->n_value_heads > 0 && !loss->returns) ||
                !loss->dlog_std_rows || !grads->dlog_std)
                return fail(PPO_E_INVALID, "ppo_mlp_train_f32: bad gaussian-loss arguments");
            a.loss = MLP_LOSS_GAUSS;
            a.lg = GaussLossP{NH, loss->n_actions, loss->n_value_heads, loss->actions_f, loss->old_log_pac, loss->advantages,
                              loss->returns, loss->log_std, loss->eps_clip, loss->vf_coef, loss->grad_scale, loss->stats};
            a.dlog_std_rows = loss->dlog_std_rows;
            break;
        case PPO_MLP_LOSS_PPO:
            if (loss->n_actions <= 0 || loss->n_actions > kMaxActions || loss->n_value_heads < 0 ||
                NH < loss->n_actions + loss->n_value_heads || !loss->actions_i || !loss->old_log_pac || !loss->advantages ||
                (loss->n_value_heads > 0 && !loss->returns))
                return fail(PPO_E_INVALID, "ppo_mlp_train_f32: bad ppo-loss arguments");
            a.loss = MLP_LOSS_PPO;
            a.lp = PpoLossP{NH, loss->n_actions, loss->n_value_heads, loss->actions_i, loss->old_log_pac, loss->old_log_policy,
                            loss->advantages, loss->returns, loss->eps_clip, loss->ent_coef, loss->vf_coef, loss->grad_scale, loss->stats};
            break;
        default:
            return fail(PPO_E_INVALID, "ppo_mlp_train_f32: unknown loss kind %d", loss->kind);
    }
    hipStream_t st = as_stream(stream);
    rc = launch_rows(a, st);
    if (rc) return rc;

    MlpWgradArgs w{};
    w.B = B;
    int tile0 = 0;
    auto problem = [&](int q, const float *dy, const float *xx, float *dw, float *db, int N, int K) {
        WgradProblem &p = w.p[q];
        p.dy = dy, p.x = xx, p.dw = dw, p.db = db, p.N = N, p.K = K;
        p.tiles_k = (K + 15) / 16, p.tile0 = tile0;
        tile0 += ((N + 15) / 16) * p.tiles_k;
    };
    problem(0, a.da1, x_indexed ? a.x_out : x, grads->dw1, grads->db1, H, F);
    problem(1, a.dh, a.a1_out, grads->dw2, grads->db2, H, H);
    problem(2, a.dheads, a.hact_out, grads->dwh, grads->dbh, NH, H);
    w.n_tiles = tile0;
    w.dlog_std_rows = a.loss == MLP_LOSS_GAUSS ? a.dlog_std_rows : nullptr;
    w.dlog_std = grads->dlog_std, w.nA = grads->n_log_std;
    w.partials = partials;
    w.stats = stat_sums ? loss->stats : nullptr, w.stat_sums = stat_sums, w.n_stats = n_stats, w.stat_accumulate = stat_accumulate;
    int wgs = (tile0 + 3) / 4;
    if (wgs > 255) wgs = 255;  // the optimiser re-reduces at most 256 partials
    w.tile_wgs = wgs;
    *n_partials = wgs + 1;
    hipLaunchKernelGGL(mlp_wgrad_kernel, dim3(wgs + 1), dim3(256), 0, st, w);
    return check_launch("mlp_wgrad_kernel");
}

// Weight/bias gradient of the 3x3 convolution on the f32 MFMA (gfx950).
//
//   dW[o,i,ky,kx] = sum_{n,y,x} dy[n,o,y,x] * f(in[n,i,y+ky-1,x+kx-1])      db[o] = sum_{n,y,x} dy[n,o,y,x]
//
// (what autograd computes for torch.nn.Conv2d at rl/impala.py:61-62,96; f is the same fused load
// transform as the forward kernel: identity / ReLU / uint8/255).
//
// GEMM view: M = output channel o (MFMA "i", operand A = dy), N = j = tap*CINP + i plus one all-ones
// column whose result is db (MFMA "j", operand B = shifted input), K = pixels, 4 adjacent pixels of a
// row per v_mfma_f32_16x16x4_f32.
//
// A 512-thread workgroup (8 waves) walks (image, row band) items.  Both bands sit in LDS, planar, plane
// stride = 2 (mod 32) banks so that 16 channels x 2 adjacent pixels are conflict-free; an extra plane holds
// 0.0 (padding columns) and, where the bias column fits into the padding of the last N tile anyway (4 input
// channels: 37 of 48 columns), another holds 1.0, so the K loop has no selects or branches.  Where 9*CINP is a
// multiple of 16 (16 / 32 input channels) a ones column would cost a whole extra N tile (10 instead of 9, 19
// instead of 18 tiles: 5-11 % more MFMAs); there db is the running sum of the A operand each lane reads anyway.
// Float inputs (every layer but the uint8 observation convolution) use the RUN layout of the forward kernels: a
// channel's band rows sit back to back with the image's own row stride and NO halo columns, in the dy band too, so a
// plane is one contiguous run that 16-byte LDS-DMA requests move (the 4-byte request per row of the halo-column layout
// was what bounded these kernels: 256 requests per 42x42 item, the K loop waiting at the item barrier for them; cutting
// 11 % of the MFMAs changed nothing).  K runs over the flat pixel index 4 at a time with no per-row padding (21 -> 24
// columns cost 14 % of the K steps); the x-1 / x+1 taps of a pixel in the first / last column read the neighbouring
// row's pixel and are zeroed by a select that exists only in the (compile-time known) steps that touch a row end
// (the first version chose 1/0/x per lane and skipped tiles per wave: hipcc turned that into exec-masked
// branches and accumulator copies, 30 VALU per MFMA).  wave % 4 owns a fixed set of N tiles (the wave
// count per tile set is a compile-time property of the code path the wave takes), wave / 4 splits the
// band's K steps in two contiguous halves.  Accumulators stay in registers across all items of the
// workgroup; at the end the second K half is added through LDS and one [COUT][JP] slab per workgroup is
// written.  conv3x3_wgrad_reduce sums the slabs in a fixed order (deterministic, no atomics) into
// PyTorch's [o][i][3][3] layout.
#include "common.h"
#include "conv_stage.h"
#include "mfma.h"

namespace ppo {
namespace {

constexpr int pad_mod32(int v, int target) { return v + ((target - v % 32) + 32) % 32; }

#ifndef PPO_STAMP  // the stamp macros come from conv3x3.hip in the tools/conv_tune -DPPO_TUNE_STAMPS build
#define PPO_STAMP(var)
#define PPO_STAMP_ADD(slot, t1, t0)
#endif
// Double-buffered staging pays where the bands arrive by LDS-DMA and a workgroup sees several items; uint8
// observations are staged through registers (the copy cannot overlap the K loop) and the 11x11 / 8x8 layers
// are one item per workgroup (measured: 62 vs 76 us and 20.9 vs 21.8 us single-buffered).
#ifndef PPO_TUNE_WGRAD_NBUF
#define PPO_TUNE_WGRAD_NBUF 2
#endif
#ifndef PPO_TUNE_WGRAD_PERCU
#define PPO_TUNE_WGRAD_PERCU 2
#endif
#ifndef PPO_TUNE_WGRAD_NBUF_ONEBAND
#define PPO_TUNE_WGRAD_NBUF_ONEBAND 1
#endif
template <int IN_MODE, int NBANDS>
constexpr int wgrad_nbuf()
{
    return IN_MODE == IN_U8 ? 1 : (NBANDS == 1 ? PPO_TUNE_WGRAD_NBUF_ONEBAND : PPO_TUNE_WGRAD_NBUF);
}
// contiguous-run band layout (16-byte LDS-DMA) wherever both bands arrive by DMA
template <int IN_MODE, bool DY_POOLED>
constexpr bool wgrad_run() { return true; }  // (the halo-column layout remains for reference behind RUN = false)
constexpr int kWgradWaves = 8;
// Four extra waves (one per SIMD) that do nothing but issue the NEXT item's LDS-DMA while the eight compute waves run
// this item's K loop.  Stamps of the 16 -> 16 42x42 layer (one resident workgroup per CU, 12 k cycles per item): K loop
// 60 %, item barrier 23 %, DMA issue 13 % - every compute wave stopped feeding the MFMA pipe to issue its share of ~100
// requests, all of them at the same time, and the skew of that issue phase came back as barrier wait.  Measured on the
// whole training step (tools/model_speed.py, batch 256): 0 staging waves 1.112 ms, 1 -> 1.178 (one wave cannot issue an
// item's requests within a K loop: the compute waves wait for IT), 2 -> 1.113, 4 -> 1.089, 8 -> 1.123.  Only where both
// bands arrive by DMA into a second buffer (not the uint8 first layer, not the pooled-gradient gather, not one-item
// workgroups).
#ifndef PPO_TUNE_WGRAD_DMA_WAVE
#define PPO_TUNE_WGRAD_DMA_WAVE 4
#endif
template <int IN_MODE, int NBANDS, bool DY_POOLED>
constexpr bool wgrad_dma_wave()
{
    return PPO_TUNE_WGRAD_DMA_WAVE && wgrad_nbuf<IN_MODE, NBANDS>() == 2 && IN_MODE != IN_U8 && !DY_POOLED;
}
constexpr int kWgradDmaWaves = PPO_TUNE_WGRAD_DMA_WAVE;  // staging waves (0 = the compute waves stage)
template <int IN_MODE, int NBANDS, bool DY_POOLED>
constexpr int wgrad_threads() { return (kWgradWaves + (wgrad_dma_wave<IN_MODE, NBANDS, DY_POOLED>() ? kWgradDmaWaves : 0)) * 64; }

template <int CIN, int COUT, int H, int W, int TR, int NBUF_, bool RUN_>
struct WgradCfg {
    static constexpr int CINP = (CIN + 3) / 4 * 4;
    static constexpr int MTC = COUT / 16;
    // bias gradient: a ones column of B where it is free (the last N tile has a spare column), else lane sums of A
    static constexpr bool BIAS_IN_TILE = (9 * CINP) % 16 != 0;
    static constexpr int NJ = 9 * CINP + (BIAS_IN_TILE ? 1 : 0);
    static constexpr int NTT = (NJ + 15) / 16;
    static constexpr int JP = ((9 * CINP + 1) + 15) / 16 * 16;  // slab row: the N tiles, then (or inside them) column 9*CINP = db
    // The 8 waves are WT tile owners x KG K-split groups.  WT is chosen so the N tiles divide (almost) evenly:
    // 3 tiles (4 input channels) -> every wave owns all 3 and takes an eighth of K; 10 tiles (16 channels) ->
    // 2 owners x 5 tiles, K in quarters; 19 tiles (32 channels) -> 4 owners x 5/5/5/4, K in halves.  (With a
    // fixed 4 x 2 split the 10-tile layers ran 3/3/2/2 tiles per owner: a third of the K loop spent waiting at
    // the item barrier.)
    // 9 tiles (16 channels, db by lane sums) -> one owner, K in eighths; 18 tiles (32 channels) -> 2 owners x 9, K in quarters.
    static constexpr int WT = (NTT <= 4 || NTT == 9) ? 1 : ((NTT <= 12 || NTT == 18) ? 2 : 4);
    static constexpr int KG = kWgradWaves / WT;
    static constexpr int NTW_MAX = (NTT + WT - 1) / WT;  // tiles of owners with index < REM
    static constexpr int REM = NTT % WT == 0 ? WT : NTT % WT;
    static constexpr int WIDTH = W;
    static constexpr bool RUN = RUN_;                   // contiguous rows, no halo columns, K over the flat band index
    static_assert(RUN || W % 4 == 0, "the halo-column layout steps whole rows 4 pixels at a time");
    static constexpr int G = 4;                         // RUN: guard floats in front of the rows of an x plane
    static constexpr int PWD = W;                       // dy row pitch
    static constexpr int PWX = RUN ? W : W + 2;         // x row pitch (halo-column layout: pixel column c at c + 1)
    static constexpr int ROWS = TR + 2;
    static constexpr int NBANDS = (H + TR - 1) / TR;
    static constexpr int STEPS = RUN ? (TR * W + 3) / 4 : TR * (W / 4);
    // RUN: step s reads dy[4s .. 4s+3] and x[4s + G - 1 + ky*W + kx + 0..3]
    static constexpr int XRUN = 4 * STEPS + 2 * W + 2 * G > ROWS * W + 2 * G ? 4 * STEPS + 2 * W + 2 * G : ROWS * W + 2 * G;
    // plane stride 2 (mod 32) banks: 16 channels x 2 adjacent pixels conflict-free; the observation convolution
    // (<= 8 channels, staged four pixels per lane with 16-byte LDS stores) takes 4 (mod 32) for the alignment
    static constexpr int XPLANE = pad_mod32(RUN ? XRUN : ROWS * PWX, (RUN && CINP <= 8) ? 4 : 2);
    static constexpr int DPLANE = pad_mod32(RUN ? 4 * STEPS : TR * PWD, 2);
    static constexpr int LDS_X = (CINP + 2) * XPLANE;  // + ones plane (BIAS_IN_TILE) + zeros plane
    static constexpr int LDS_D = COUT * DPLANE;
    static constexpr int LDS_RED = COUT * JP;  // ONE cross-K-group reduction image (reuses the staging space): groups fold in turn
    static constexpr int NBUF = NBUF_;              // staging buffers: 2 = the next item's bands are in flight
    static constexpr int LDS_BUF = LDS_X + LDS_D;   //   (LDS-DMA) while this item's K loop runs, one barrier per item
    static constexpr int LDS_WORDS = NBUF * LDS_BUF > LDS_RED ? NBUF * LDS_BUF : LDS_RED;
    static constexpr size_t LDS_BYTES = (size_t)LDS_WORDS * 4;
    static_assert(COUT % 16 == 0, "COUT must be a multiple of 16");
};

// The K loop of K-split group KGI for a wave that owns NTW tiles.  Fully unrolled: the step range of a group
// is a compile-time constant, so every LDS offset is an immediate off a per-lane base register, and the
// operands of step i+PF are requested before the MFMAs of step i issue (the rolled loop the compiler produced
// from a runtime range exposed the whole LDS latency on every step: 58 % of the MFMA rate in its K loop).
template <class C, int NTW, bool RELU, int KGI>
__device__ __forceinline__ void wgrad_k_loop(const float *__restrict__ s_x, const float *__restrict__ s_d,
                                             const int (&joff)[C::NTW_MAX], const int (&ml)[C::NTW_MAX],
                                             const int (&mr)[C::NTW_MAX], int aoff, int g,
                                             f32x4 (&acc)[C::MTC][C::NTW_MAX], float (&asum)[C::MTC], float floor)
{
    constexpr int SPR = C::RUN ? 1 : C::PWD / 4;  // steps per row (halo-column layout)
    constexpr int S0 = KGI * C::STEPS / C::KG, S1 = (KGI + 1) * C::STEPS / C::KG;
    constexpr int LEN = S1 - S0;
    constexpr int PF = 2;
    float a[PF + 1][C::MTC], b[PF + 1][NTW];
    int xb[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) xb[t] = joff[t] + g;
    auto load = [&](int i, float (&aa)[C::MTC], float (&bb)[NTW]) {
        const int s = S0 + i;
        const int pd = 4 * s;                                     // dy: linear in s (rows are 4*SPR wide, or flat)
        const int px = C::RUN ? 4 * s : 4 * s + 2 * (s / SPR);    // x rows are PWD + 2 wide, or flat with the same pitch
#pragma unroll
        for (int m = 0; m < C::MTC; ++m) aa[m] = s_d[m * 16 * C::DPLANE + aoff + pd];
#pragma unroll
        for (int t = 0; t < NTW; ++t) bb[t] = s_x[xb[t] + px];
    };
#pragma unroll
    for (int i = 0; i < PF && i < LEN; ++i) load(i, a[i], b[i]);
#pragma unroll
    for (int i = 0; i < LEN; ++i) {
        if (i + PF < LEN) load(i + PF, a[(i + PF) % (PF + 1)], b[(i + PF) % (PF + 1)]);
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            float bv = b[i % (PF + 1)][t];
            // the band was staged raw by LDS-DMA: the forward's ReLU-on-load is applied here (floor 0: relu1's own v_med3;
            // floor -inf: the identity, for a problem of the batch whose forward read its input raw)
            if (RELU) bv = __builtin_amdgcn_fmed3f(bv, floor, __builtin_inff());
            if constexpr (C::RUN) {
                // pixel 4s + e sits in column (4s + e) % W: in the first column the x-1 taps, in the last the x+1 taps
                // read a neighbouring row's pixel.  ml / mr hold the lane's pixel phase g when its column's tap is an
                // x-1 / x+1 one (else -1), so one compare + select per operand, in the steps that touch a row end only.
                // (i is a fully unrolled loop index: everything below folds to constants per step)
                constexpr int W = C::WIDTH;
                const int c0 = (4 * (S0 + i)) % W;
                const int eL = c0 == 0 ? 0 : (W - c0 < 4 ? W - c0 : -1);          // phase whose column is 0
                const int eR = (W - 1 - c0 < 4) ? W - 1 - c0 : -1;                // phase whose column is W-1
                if (eL >= 0) bv = ml[t] == eL ? 0.f : bv;
                if (eR >= 0) bv = mr[t] == eR ? 0.f : bv;
            }
#pragma unroll
            for (int m = 0; m < C::MTC; ++m) acc[m][t] = mfma16(a[i % (PF + 1)][m], bv, acc[m][t]);
        }
        if constexpr (!C::BIAS_IN_TILE) {
#pragma unroll
            for (int m = 0; m < C::MTC; ++m) asum[m] += a[i % (PF + 1)][m];  // db: step order, then lanes, K groups, slabs
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// dispatch on the (wave-uniform) K group and tile count of this wave
template <class C, bool RELU, int KGI = 0>
__device__ __forceinline__ void wgrad_k_loops(const float *__restrict__ s_x, const float *__restrict__ s_d,
                                              const int (&joff)[C::NTW_MAX], const int (&ml)[C::NTW_MAX],
                                              const int (&mr)[C::NTW_MAX], int aoff, int g, int wt, int kg,
                                              f32x4 (&acc)[C::MTC][C::NTW_MAX], float (&asum)[C::MTC], float floor)
{
    constexpr int NLO = C::NTW_MAX > 1 ? C::NTW_MAX - 1 : 1;
    if (kg == KGI) {
        if (wt < C::REM) wgrad_k_loop<C, C::NTW_MAX, RELU, KGI>(s_x, s_d, joff, ml, mr, aoff, g, acc, asum, floor);
        else if (C::NTW_MAX > 1) wgrad_k_loop<C, NLO, RELU, KGI>(s_x, s_d, joff, ml, mr, aoff, g, acc, asum, floor);
    } else if constexpr (KGI + 1 < C::KG) {
        wgrad_k_loops<C, RELU, KGI + 1>(s_x, s_d, joff, ml, mr, aoff, g, wt, kg, acc, asum, floor);
    }
}

// Up to kWgradBatch independent problems of one geometry per launch (blockIdx.y picks one): the layers of a stack share
// their geometry and their gradients are all known once its backward-data pass is through, and one launch of 4 x the
// workgroups has one ramp and one tail where four launches have four (per-launch constant ~10 us).
// set by ppo_conv3x3_backward_weight_slabs_pooled_indexed_f32 around its dispatch (see WgradBatch::in_index)
thread_local const int32_t *t_wgrad_in_index = nullptr;
constexpr int kWgradBatch = 5;  // a stack's four block convolutions + the next stack's first convolution (same geometry)
struct WgradBatch {
    const void *in[kWgradBatch];
    const float *dy[kWgradBatch];  // DY_POOLED: the POOLED gradient g [n, COUT, H/2, W/2] ...
    float *partial[kWgradBatch];
    float relu_floor[kWgradBatch]; // IN_RELU launches: the lower clamp of the input on load, per problem - 0 = the forward's
                                   // ReLU, -inf = the raw input (a first convolution riding in a block batch)
    const uint8_t *argmax;         // ... and the pooling's argmax (problem 0 only): dy = maxpool_backward(g, argmax)
    const int32_t *in_index;       // nullable (uint8 input only): image i of the launch is image in_index[i] of `in` - the
                                   // minibatch permutation, so that the observations need no gathered copy
};

template <int CIN, int COUT, int H, int W, int TR, int IN_MODE, bool DY_POOLED = false>
__global__ __launch_bounds__((wgrad_threads<IN_MODE, (H + TR - 1) / TR, DY_POOLED>())) void conv3x3_wgrad_kernel(WgradBatch batch,
                                                                                                                int n_images)
{
    using C = WgradCfg<CIN, COUT, H, W, TR, wgrad_nbuf<IN_MODE, (H + TR - 1) / TR>(), wgrad_run<IN_MODE, DY_POOLED>()>;
    constexpr bool DMAW = wgrad_dma_wave<IN_MODE, (H + TR - 1) / TR, DY_POOLED>();
    const void *__restrict__ in_ = batch.in[blockIdx.y];
    const float *__restrict__ dy = batch.dy[blockIdx.y];
    float *__restrict__ partial = batch.partial[blockIdx.y];
    const float relu_floor = batch.relu_floor[blockIdx.y];
    extern __shared__ __align__(16) float smem[];
    float *s_x = smem;
    float *s_d = smem + C::LDS_X;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    PPO_STAMP(t_k0)
    const int wave = tid >> 6;
    const bool is_dma = DMAW && wave >= kWgradWaves;  // a staging wave: no tiles, no K group
    const int wt = wave % C::WT;  // owner of N tiles wt, wt + WT, ...
    const int kg = is_dma ? -1 : wave / C::WT;  // K-split group
    const int l15 = lane & 15;
    const int g = lane >> 4;

    // per-lane B offsets of this wave's n-tiles: j = (wt + 4*t)*16 + l15 -> (tap, ci); the bias column reads
    // the ones plane, padding columns the zeros plane
    int joff[C::NTW_MAX], ml[C::NTW_MAX], mr[C::NTW_MAX];
#pragma unroll
    for (int t = 0; t < C::NTW_MAX; ++t) {
        const int j = (wt + C::WT * t) * 16 + l15;
        const int tap = j / C::CINP;
        const int ci = j % C::CINP;
        // halo-column layout: x of pixel (row r, column c) sits at r * PWX + c + 1; RUN: at G + r * W + c, i.e. the
        // tap (ky, kx) of dy pixel q is at q + (G - 1) + ky * W + kx
        joff[t] = j < 9 * C::CINP ? ci * C::XPLANE + (tap / 3) * C::PWX + (tap % 3) + (C::RUN ? C::G - 1 : 0)
                                  : ((C::BIAS_IN_TILE && j == 9 * C::CINP) ? C::CINP * C::XPLANE : (C::CINP + 1) * C::XPLANE);
        ml[t] = (j < 9 * C::CINP && tap % 3 == 0) ? g : -1;
        mr[t] = (j < 9 * C::CINP && tap % 3 == 2) ? g : -1;
    }
    float asum[C::MTC];
#pragma unroll
    for (int m = 0; m < C::MTC; ++m) asum[m] = 0.f;
    const int aoff = l15 * C::DPLANE + g;  // A: channel l15 of the m-tile, pixel +g

    f32x4 acc[C::MTC][C::NTW_MAX];
#pragma unroll
    for (int m = 0; m < C::MTC; ++m)
#pragma unroll
        for (int t = 0; t < C::NTW_MAX; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // zero everything once (halo columns, padded channels and padded dy columns are never written by the
    // LDS-DMA staging), then the constant planes
    if (!is_dma) zero_lds<C::NBUF * C::LDS_BUF, kWgradWaves * 64>(smem, tid);
    __syncthreads();
    if constexpr (C::BIAS_IN_TILE)
        for (int b = 0; b < C::NBUF; ++b)
            for (int i = tid; i < C::XPLANE && !is_dma; i += kWgradWaves * 64) s_x[b * C::LDS_BUF + C::CINP * C::XPLANE + i] = 1.0f;

    const int n_items = n_images * C::NBANDS;
    constexpr int SW = DMAW ? kWgradDmaWaves : kWgradWaves;  // waves that share the staging of an item
    const int stid = DMAW ? tid - kWgradWaves * 64 : tid;
    auto stage = [&](int item, float *bx, float *bd) {
        const int img = item / C::NBANDS;
        const int y0 = (item % C::NBANDS) * TR;
        if constexpr (C::RUN) {
            if constexpr (IN_MODE == IN_U8) {
                // uint8 observations: four pixels per lane (dword load, exact x / 255, 16-byte LDS store)
                uint32_t raw[FlatU8Map<CIN, W, C::ROWS, kWgradWaves * 64>::Q];
                band_u8x4_load<CIN, H, W, C::ROWS, kWgradWaves * 64>(in_, batch.in_index ? batch.in_index[img] : img, y0, tid, raw);
                band_u8x4_store<CIN, W, C::ROWS, C::XPLANE, C::G, kWgradWaves * 64>(raw, bx, tid);
            } else {
                stage_band_chunk_dma<CIN, H, W, C::ROWS, C::XPLANE, C::G, SW>(static_cast<const float *>(in_), img, y0, bx, stid);
            }
            if constexpr (DY_POOLED) {
                static_assert(C::LDS_X % 2 == 0 && C::LDS_BUF % 2 == 0 && C::DPLANE % 2 == 0, "stage_dy_pooled stores float2 pairs");
                stage_dy_pooled<COUT, H, W, TR, C::PWD, C::DPLANE, kWgradWaves>(dy, batch.argmax, img, y0, bd, tid);
            } else {
                // the same routine for the TR rows y0 .. y0 + TR - 1 of dy: its first row is (y0 + 1) - 1, no guard
                stage_band_chunk_dma<COUT, H, W, TR, C::DPLANE, 0, SW>(dy, img, y0 + 1, bd, stid);
            }
            return;
        } else if constexpr (IN_MODE == IN_U8)
            stage_band<CIN, C::CINP, H, W, C::ROWS, C::PWX, C::XPLANE, 1, IN_MODE, kWgradWaves>(in_, img, y0, bx, tid);
        else
            stage_band_dma<CIN, H, W, C::ROWS, C::PWX, C::XPLANE, 1, kWgradWaves>(static_cast<const float *>(in_), img, y0, bx, tid);
        if constexpr (DY_POOLED) {
            static_assert(C::LDS_X % 2 == 0 && C::LDS_BUF % 2 == 0, "stage_dy_pooled stores float2 pairs");
            stage_dy_pooled<COUT, H, W, TR, C::PWD, C::DPLANE, kWgradWaves>(dy, batch.argmax, img, y0, bd, tid);
        }
        else {
            stage_band_dma<COUT, H, W, TR, C::PWD, C::DPLANE, 0, kWgradWaves>(dy, img, y0, bd, tid);
        }
    };
    constexpr bool RELU = IN_MODE == IN_RELU;
    PPO_STAMP(t_k1)
    PPO_STAMP_ADD(6, t_k1, t_k0)  // prologue: lane constants, LDS zeroing
    if constexpr (C::NBUF == 2) {
        int buf = 0;
        if ((int)blockIdx.x < n_items && (!DMAW || is_dma)) stage(blockIdx.x, s_x, s_d);
        for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
            const float *bx = s_x + buf * C::LDS_BUF, *bd = s_d + buf * C::LDS_BUF;
            PPO_STAMP(t_top)
#ifdef PPO_TUNE_STAMPS_VM  // experiment build: how much of the barrier wait is this wave's own DMA still in flight?
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PPO_STAMP(t_vm)
            PPO_STAMP_ADD(6, t_vm, t_top)
#endif
            __syncthreads();  // this item's bands have landed (vmcnt(0)); every wave is done with the other buffer
            PPO_STAMP(t_bar)
            if (item + (int)gridDim.x < n_items && (!DMAW || is_dma))
                stage(item + gridDim.x, s_x + (buf ^ 1) * C::LDS_BUF, s_d + (buf ^ 1) * C::LDS_BUF);
            PPO_STAMP(t_staged)
            if (!is_dma) wgrad_k_loops<C, RELU>(bx, bd, joff, ml, mr, aoff, g, wt, kg, acc, asum, relu_floor);
            PPO_STAMP(t_end)
            PPO_STAMP_ADD(0, t_bar, t_top)      // barrier wait
            PPO_STAMP_ADD(1, t_staged, t_bar)   // DMA issue of the next item
            PPO_STAMP_ADD(2, t_end, t_staged)   // K loop
            PPO_STAMP_ADD(4, t_end, t_top)      // whole item
            if (lane == 0) { PPO_STAMP_ADD(5, 1ull, 0ull) }
            buf ^= 1;
        }
    } else {
        // Register-staged operands (uint8 observations, the pooled gradient's gather) are fetched for the NEXT item
        // before this item's K loop and written to LDS after it, so their load latency hides under the MFMAs (the
        // single-buffered loop exposed it: staging was 65 % of an item).  DMA-staged operands keep the plain order.
        constexpr bool PRE_X = C::RUN && IN_MODE == IN_U8, PRE_D = C::RUN && DY_POOLED;
        using XM = FlatU8Map<CIN, (PRE_X ? W : 4), C::ROWS, kWgradWaves * 64>;
        using DM = PooledMap<COUT, (PRE_D ? H : 2), (PRE_D ? W : 4), (PRE_D ? TR : 2), kWgradWaves>;
        uint32_t xraw[PRE_X ? XM::Q : 1];
        PooledRaw draw[PRE_D ? DM::Q : 1];
        auto prefetch = [&](int item) {
            const int img = item / C::NBANDS, y0 = (item % C::NBANDS) * TR;
            if constexpr (PRE_X)
                band_u8x4_load<CIN, H, W, C::ROWS, kWgradWaves * 64>(in_, batch.in_index ? batch.in_index[img] : img, y0, tid, xraw);
            if constexpr (PRE_D) dy_pooled_load<COUT, H, W, TR, kWgradWaves>(dy, batch.argmax, img, y0, tid, draw);
        };
        if ((PRE_X || PRE_D) && (int)blockIdx.x < n_items) prefetch(blockIdx.x);
        for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
            PPO_STAMP(t_top)
            __syncthreads();
            PPO_STAMP(t_bar)
            if constexpr (PRE_X || PRE_D) {
                const int img = item / C::NBANDS, y0 = (item % C::NBANDS) * TR;
                if constexpr (PRE_X) band_u8x4_store<CIN, W, C::ROWS, C::XPLANE, C::G, kWgradWaves * 64>(xraw, s_x, tid);
                else stage_band_chunk_dma<CIN, H, W, C::ROWS, C::XPLANE, C::G, kWgradWaves>(static_cast<const float *>(in_), img, y0, s_x, tid);
                if constexpr (PRE_D) dy_pooled_store<COUT, H, W, TR, C::PWD, C::DPLANE, kWgradWaves>(draw, y0, s_d, tid);
                else stage_band_chunk_dma<COUT, H, W, TR, C::DPLANE, 0, kWgradWaves>(dy, img, y0 + 1, s_d, tid);
                __syncthreads();
                if (item + (int)gridDim.x < n_items) prefetch(item + gridDim.x);
            } else {
                stage(item, s_x, s_d);
                __syncthreads();
            }
            PPO_STAMP(t_staged)
            wgrad_k_loops<C, RELU>(s_x, s_d, joff, ml, mr, aoff, g, wt, kg, acc, asum, relu_floor);
            PPO_STAMP(t_end)
            PPO_STAMP_ADD(0, t_bar, t_top)      // barrier wait
            PPO_STAMP_ADD(1, t_staged, t_bar)   // staging (single-buffered: exposed)
            PPO_STAMP_ADD(2, t_end, t_staged)   // K loop
            PPO_STAMP_ADD(4, t_end, t_top)      // whole item
            if (lane == 0) { PPO_STAMP_ADD(5, 1ull, 0ull) }
        }
    }

    // fold the K groups through LDS: groups 1.. park their accumulators in images of their own, group 0
    // adds them in group order (fixed order: deterministic), then one slab per workgroup:
    // partial[wg][co][j]; a lane holds rows g*4+r (co), column l15 (j) of each of its tiles
    float *s_red = smem;
    PPO_STAMP(t_k2)
    __syncthreads();  // every wave is done reading the staging buffers
    // db by lane sums: a lane summed dy[co = m*16 + l15][pixels = g (mod 4)] over its K steps; the four pixel phases are
    // added across the lane groups (fixed order), K groups through the reduction images, workgroups through the slab
    const bool db_lane = !C::BIAS_IN_TILE && wt == 0 && g == 0;
    if constexpr (!C::BIAS_IN_TILE) {
#pragma unroll
        for (int m = 0; m < C::MTC; ++m) {
            float v = asum[m];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            asum[m] = v;
        }
    }
    // one image, the groups take turns (group order: deterministic); 2 barriers per group, once per kernel — a
    // set of KG - 1 images was up to 117 KB of LDS, more than the staging buffers and the reason for one workgroup per CU
#pragma unroll 1
    for (int src = 1; src < C::KG; ++src) {
        float *img = s_red;
        if (kg == src) {
            if (db_lane) {
#pragma unroll
                for (int m = 0; m < C::MTC; ++m) img[(m * 16 + l15) * C::JP + 9 * C::CINP] = asum[m];
            }
#pragma unroll
            for (int m = 0; m < C::MTC; ++m)
#pragma unroll
                for (int t = 0; t < C::NTW_MAX; ++t)
                    if (wt + C::WT * t < C::NTT)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            img[(m * 16 + g * 4 + r) * C::JP + (wt + C::WT * t) * 16 + l15] = acc[m][t][r];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int m = 0; m < C::MTC; ++m)
#pragma unroll
                for (int t = 0; t < C::NTW_MAX; ++t)
                    if (wt + C::WT * t < C::NTT)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[m][t][r] += img[(m * 16 + g * 4 + r) * C::JP + (wt + C::WT * t) * 16 + l15];
            if (db_lane) {
#pragma unroll
                for (int m = 0; m < C::MTC; ++m) asum[m] += img[(m * 16 + l15) * C::JP + 9 * C::CINP];
            }
        }
        if (src + 1 < C::KG) __syncthreads();  // the image is free again
    }
    if (kg == 0) {
        float *slab = partial + (size_t)blockIdx.x * COUT * C::JP;
#pragma unroll
        for (int m = 0; m < C::MTC; ++m)
#pragma unroll
            for (int t = 0; t < C::NTW_MAX; ++t)
                if (wt + C::WT * t < C::NTT)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        slab[(m * 16 + g * 4 + r) * C::JP + (wt + C::WT * t) * 16 + l15] = acc[m][t][r];
        if (db_lane) {
#pragma unroll
            for (int m = 0; m < C::MTC; ++m) slab[(m * 16 + l15) * C::JP + 9 * C::CINP] = asum[m];
        }
    }
    PPO_STAMP(t_k3)
    PPO_STAMP_ADD(7, t_k3, t_k2)  // K-group fold + slab write
    PPO_STAMP_ADD(3, t_k3, t_k0)  // whole kernel, per wave
}

// dW[o][i][tap] = sum_wg partial[wg][o][tap*CINP + i]; db[o] = sum_wg partial[wg][o][9*CINP].
// A thread owns 4 consecutive j of one output channel (16-byte loads); 64 such quads per workgroup,
// 4 threads per quad each summing a quarter of the slabs, combined in LDS in a fixed order.
// Sum of one thread's partition of the slabs (slabs lo .. hi - 1 of one float4 of the [cout][jp] image): even slabs
// into one accumulator, odd ones into another, added at the end (the order the rolled two-at-a-time loop used, so the
// result is bit-identical to it).  All (up to 16) loads of a trip are issued before the first add - the rolled loop
// paid a memory round trip per pair, 8 of them in a row for 256 slabs - through a range-checked buffer descriptor
// (common.h): a slab index past hi reads zeros.
__device__ __forceinline__ float4 slab_partition_sum(const float *slabs, int byte_off, int byte_stride, int lo, int hi, bool live)
{
    const __amdgpu_buffer_rsrc_t buf = buffer_of(slabs);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    for (int q = lo; q < hi; q += 16) {
        float4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = buffer_f32x4(buf, (live && q + u < hi) ? byte_off + (q + u) * byte_stride : kOutside);
#pragma unroll
        for (int u = 0; u < 16; u += 2) {
            s0.x += v[u].x; s0.y += v[u].y; s0.z += v[u].z; s0.w += v[u].w;
            s1.x += v[u + 1].x; s1.y += v[u + 1].y; s1.z += v[u + 1].z; s1.w += v[u + 1].w;
        }
    }
    return make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
}

__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const float *__restrict__ partial, int n_slabs,
                                                                   int cout, int cin, int cinp, int jp,
                                                                   float *__restrict__ dw, float *__restrict__ db,
                                                                   int accumulate)
{
    __shared__ float4 s[256];
    const int q4 = jp / 4;  // quads per channel row
    const int idx = blockIdx.x * 16 + (threadIdx.x & 15);  // 16 output quads per workgroup
    const int part = threadIdx.x >> 4;                      // 16 slab partitions per quad
    const bool live = idx < cout * q4;
    const int co = live ? idx / q4 : 0;
    const int j0 = live ? (idx % q4) * 4 : 0;
    const size_t stride = (size_t)cout * jp;
    const int per = (n_slabs + 15) / 16;
    const int lo = part * per;
    const int hi = lo + per < n_slabs ? lo + per : n_slabs;
    s[threadIdx.x] = slab_partition_sum(partial, (co * jp + j0) * 4, (int)stride * 4, lo, hi, live);
    __syncthreads();
    if (part == 0 && live) {
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 16; ++q) {  // fixed order over the slab partitions
            const float4 a = s[q * 16 + threadIdx.x];
            sum[0] += a.x;
            sum[1] += a.y;
            sum[2] += a.z;
            sum[3] += a.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = j0 + e;
            if (j == 9 * cinp) {
                if (db) db[co] = accumulate ? db[co] + sum[e] : sum[e];
            } else if (j < 9 * cinp) {
                const int tap = j / cinp;
                const int ci = j % cinp;
                if (ci < cin) {
                    float *dst = dw + ((size_t)co * cin + ci) * 9 + tap;
                    *dst = accumulate ? *dst + sum[e] : sum[e];
                }
            }
        }
    }
}

#ifndef PPO_TUNE_WGRAD_MAXSLABS
#define PPO_TUNE_WGRAD_MAXSLABS 512
#endif
constexpr int kWgradMaxSlabs = PPO_TUNE_WGRAD_MAXSLABS;

// The same reduction for up to kMaxJobs layers in one launch (blockIdx.y = layer): a backward pass defers its
// 15 slab reductions to one launch at the end instead of 15 small ones between the wgrad kernels.
constexpr int kMaxJobs = 32;
struct ReduceJobs {
    ppo_wgrad_job j[kMaxJobs];
};

__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_jobs_kernel(ReduceJobs jobs)
{
    __shared__ float4 s[256];
    const ppo_wgrad_job &job = jobs.j[blockIdx.y];
    const int cout = job.cout, cin = job.cin, n_slabs = job.n_slabs, accumulate = job.accumulate;
    const int cinp = (cin + 3) / 4 * 4;
    const int jp = ((9 * cinp + 1) + 15) / 16 * 16;
    const int q4 = jp / 4;
    if ((int)blockIdx.x * 16 >= cout * q4) return;  // this layer needs fewer blocks than the widest one
    const int idx = blockIdx.x * 16 + (threadIdx.x & 15);
    const int part = threadIdx.x >> 4;
    const bool live = idx < cout * q4;
    const int co = live ? idx / q4 : 0;
    const int j0 = live ? (idx % q4) * 4 : 0;
    const size_t stride = (size_t)cout * jp;
    const int per = (n_slabs + 15) / 16;
    const int lo = part * per;
    const int hi = lo + per < n_slabs ? lo + per : n_slabs;
    s[threadIdx.x] = slab_partition_sum(job.slabs, (co * jp + j0) * 4, (int)stride * 4, lo, hi, live);
    __syncthreads();
    if (part == 0 && live) {
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 16; ++q) {  // fixed order over the slab partitions (same order as the single-layer kernel)
            const float4 a = s[q * 16 + threadIdx.x];
            sum[0] += a.x;
            sum[1] += a.y;
            sum[2] += a.z;
            sum[3] += a.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = j0 + e;
            if (j == 9 * cinp) {
                if (job.dbias) job.dbias[co] = accumulate ? job.dbias[co] + sum[e] : sum[e];
            } else if (j < 9 * cinp) {
                const int tap = j / cinp;
                const int ci = j % cinp;
                if (ci < cin) {
                    float *d = job.dweight + ((size_t)co * cin + ci) * 9 + tap;
                    *d = accumulate ? *d + sum[e] : sum[e];
                }
            }
        }
    }
}

template <int CIN, int COUT, int H, int W, int TR, int IN_MODE, bool DY_POOLED = false>
int launch_wgrad(const void *in, const float *dy, float *dw, float *db, float *workspace, size_t workspace_bytes,
                 int n_images, int accumulate, hipStream_t st, int *n_slabs_out = nullptr,
                 const WgradBatch *more = nullptr, int count = 1, const uint8_t *argmax = nullptr)
{
    using C = WgradCfg<CIN, COUT, H, W, TR, wgrad_nbuf<IN_MODE, (H + TR - 1) / TR>(), wgrad_run<IN_MODE, DY_POOLED>()>;
    auto kern = conv3x3_wgrad_kernel<CIN, COUT, H, W, TR, IN_MODE, DY_POOLED>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        if (e != hipSuccess) return fail(PPO_E_HIP, "conv3x3_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int n_items = n_images * C::NBANDS;
    // as many workgroups as fit on the chip at once (LDS-limited), at most kWgradMaxSlabs slabs
    int per_cu = (int)((160 * 1024) / C::LDS_BYTES);
    per_cu = per_cu < 1 ? 1 : (per_cu > PPO_TUNE_WGRAD_PERCU ? PPO_TUNE_WGRAD_PERCU : per_cu);
#ifndef PPO_TUNE_WGRAD_BIGSLABS
    if ((size_t)COUT * C::JP * sizeof(float) > 16 * 1024) per_cu = 1;  // big slabs: keep the reduce traffic down
#endif
    // `count` problems share the chip (blockIdx.y): each gets 1 / count of the resident workgroups, so a batched launch
    // is ONE wave of workgroups, each walking count x as many items — a quarter of the slabs to write and reduce
    // (4 x 256 slabs of 39 KB were 1.6-2.3 x the tensors themselves for the 21x21 / 11x11 layers) and a quarter of the
    // per-workgroup prologue / K-group fold
    int grid = (256 * per_cu) / count;  // rounded DOWN: 5 x 52 workgroups would need a second wave for the last four
    if (grid < 1) grid = 1;
    if (grid > kWgradMaxSlabs) grid = kWgradMaxSlabs;
    if (grid > n_items) grid = n_items;
    const size_t need = (size_t)grid * COUT * C::JP * sizeof(float);
    if (need > workspace_bytes)
        return fail(PPO_E_INVALID, "conv3x3_wgrad: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
    WgradBatch batch{};
    if (more) {
        batch = *more;  // `count` problems, each with a workspace of workspace_bytes
    } else {
        batch.in[0] = in;
        batch.dy[0] = dy;
        batch.partial[0] = workspace;
    }
    batch.argmax = argmax;
    batch.in_index = IN_MODE == IN_U8 ? t_wgrad_in_index : nullptr;
    hipLaunchKernelGGL(kern, dim3(grid, count), dim3(wgrad_threads<IN_MODE, C::NBANDS, DY_POOLED>()), C::LDS_BYTES, st, batch,
                       n_images);
    int rc = check_launch("conv3x3_wgrad_kernel");
    if (rc) return rc;
    if (n_slabs_out) {  // slabs only: the caller reduces later (ppo_conv3x3_wgrad_reduce_f32)
        *n_slabs_out = grid;
        return PPO_OK;
    }
    const int total = COUT * (C::JP / 4);  // one thread quad-group per 4 consecutive j
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((total + 15) / 16), dim3(256), 0, st, workspace, grid, COUT,
                       CIN, C::CINP, C::JP, dw, db, accumulate);
    return check_launch("conv3x3_wgrad_reduce_kernel");
}

template <int IN_MODE>
int dispatch_wgrad(int cin, int cout, int h, int w_, const void *in, const float *dy, float *dw, float *db,
                   float *ws, size_t ws_bytes, int n, int accumulate, hipStream_t st, int *n_slabs_out = nullptr,
                   const WgradBatch *more = nullptr, int count = 1, const uint8_t *argmax = nullptr)
{
#define PPO_WGRAD_POOLED_CASE(CI, CO, HH, WW, TR)                                                    \
    if constexpr (IN_MODE != IN_RELU) {                                                              \
        if (argmax && cin == CI && cout == CO && h == HH && w_ == WW)                                \
            return launch_wgrad<CI, CO, HH, WW, TR, IN_MODE, true>(in, dy, dw, db, ws, ws_bytes, n, accumulate, st, n_slabs_out, \
                                                                   nullptr, 1, argmax);              \
    }
    PPO_WGRAD_POOLED_CASE(4, 16, 84, 84, 6)
    PPO_WGRAD_POOLED_CASE(5, 16, 84, 84, 6)
    PPO_WGRAD_POOLED_CASE(3, 16, 64, 64, 8)
    PPO_WGRAD_POOLED_CASE(4, 16, 64, 64, 8)
#undef PPO_WGRAD_POOLED_CASE
    if (argmax)
        return fail(PPO_E_INVALID, "conv3x3_wgrad: no pooled-gradient kernel for cin=%d cout=%d h=%d w=%d in_mode=%d", cin,
                    cout, h, w_, IN_MODE);
#define PPO_WGRAD_CASE(ALLOWED, CI, CO, HH, WW, TR)                                                  \
    if constexpr (ALLOWED) {                                                                         \
        if (cin == CI && cout == CO && h == HH && w_ == WW)                                          \
            return launch_wgrad<CI, CO, HH, WW, TR, IN_MODE>(in, dy, dw, db, ws, ws_bytes, n, accumulate, st, n_slabs_out, \
                                                             more, count);                                   \
    }
    constexpr bool FIRST = IN_MODE != IN_RELU;
    constexpr bool UP = IN_MODE == IN_NONE;
    constexpr bool SAME = IN_MODE == IN_RELU;
    PPO_WGRAD_CASE(FIRST, 4, 16, 84, 84, 6)
    PPO_WGRAD_CASE(FIRST, 5, 16, 84, 84, 6)
    PPO_WGRAD_CASE(FIRST, 3, 16, 64, 64, 8)
    PPO_WGRAD_CASE(FIRST, 4, 16, 64, 64, 8)
    PPO_WGRAD_CASE(UP, 16, 32, 42, 42, 7)
    PPO_WGRAD_CASE(UP, 16, 32, 32, 32, 8)
    PPO_WGRAD_CASE(UP, 32, 32, 21, 21, 7)
    PPO_WGRAD_CASE(UP, 32, 32, 16, 16, 8)
    PPO_WGRAD_CASE(SAME, 16, 16, 42, 42, 7)
    PPO_WGRAD_CASE(SAME, 16, 16, 32, 32, 8)
    PPO_WGRAD_CASE(SAME, 32, 32, 21, 21, 7)
    PPO_WGRAD_CASE(SAME, 32, 32, 16, 16, 8)
    PPO_WGRAD_CASE(SAME, 32, 32, 11, 11, 11)
    PPO_WGRAD_CASE(SAME, 32, 32, 8, 8, 8)
#undef PPO_WGRAD_CASE
    return fail(PPO_E_INVALID, "conv3x3_wgrad: unsupported geometry cin=%d cout=%d h=%d w=%d in_mode=%d", cin, cout,
                h, w_, IN_MODE);
}

}  // namespace
}  // namespace ppo

extern "C" size_t ppo_conv3x3_wgrad_workspace_bytes(int cin, int cout)
{
    const int cinp = (cin + 3) / 4 * 4;
    const int jp = ((9 * cinp + 1) + 15) / 16 * 16;
    return (size_t)ppo::kWgradMaxSlabs * cout * jp * sizeof(float);
}

extern "C" int ppo_conv3x3_backward_weight_f32(const void *in, int in_mode, const float *dy, float *dweight,
                                               float *dbias, void *workspace, size_t workspace_bytes, int n,
                                               int cin, int cout, int h, int w, int accumulate, void *stream)
{
    using namespace ppo;
    if (n <= 0) return fail(PPO_E_INVALID, "ppo_conv3x3_backward_weight_f32: n must be positive");
    if (!in || !dy || !dweight || !workspace) return fail(PPO_E_INVALID, "ppo_conv3x3_backward_weight_f32: null pointer");
    hipStream_t st = as_stream(stream);
    float *ws = static_cast<float *>(workspace);
    switch (in_mode) {
        case IN_NONE: return dispatch_wgrad<IN_NONE>(cin, cout, h, w, in, dy, dweight, dbias, ws, workspace_bytes, n, accumulate, st);
        case IN_RELU: return dispatch_wgrad<IN_RELU>(cin, cout, h, w, in, dy, dweight, dbias, ws, workspace_bytes, n, accumulate, st);
        case IN_U8: return dispatch_wgrad<IN_U8>(cin, cout, h, w, in, dy, dweight, dbias, ws, workspace_bytes, n, accumulate, st);
    }
    return fail(PPO_E_INVALID, "ppo_conv3x3_backward_weight_f32: unknown in_mode %d", in_mode);
}

extern "C" int ppo_conv3x3_backward_weight_slabs_f32(const void *in, int in_mode, const float *dy, void *workspace,
                                                     size_t workspace_bytes, int n, int cin, int cout, int h, int w,
                                                     int *n_slabs, void *stream)
{
    using namespace ppo;
    if (n <= 0) return fail(PPO_E_INVALID, "ppo_conv3x3_backward_weight_slabs_f32: n must be positive");
    if (!in || !dy || !workspace || !n_slabs) return fail(PPO_E_INVALID, "ppo_conv3x3_backward_weight_slabs_f32: null pointer");
    hipStream_t st = as_stream(stream);
    float *ws = static_cast<float *>(workspace);
    switch (in_mode) {
        case IN_NONE: return dispatch_wgrad<IN_NONE>(cin, cout, h, w, in, dy, nullptr, nullptr, ws, workspace_bytes, n, 0, st, n_slabs);
        case IN_RELU: return dispatch_wgrad<IN_RELU>(cin, cout, h, w, in, dy, nullptr, nullptr, ws, workspace_bytes, n, 0, st, n_slabs);
        case IN_U8: return dispatch_wgrad<IN_U8>(cin, cout, h, w, in, dy, nullptr, nullptr, ws, workspace_bytes, n, 0, st, n_slabs);
    }
    return fail(PPO_E_INVALID, "ppo_conv3x3_backward_weight_slabs_f32: unknown in_mode %d", in_mode);
}

extern "C" int ppo_conv3x3_backward_weight_slabs_batch_f32(const void *const *ins, int in_mode, const float *const *dys,
                                                           void *const *workspaces, size_t workspace_bytes, int count,
                                                           int n, int cin, int cout, int h, int w, int *n_slabs,
                                                           void *stream)
{
    using namespace ppo;
    if (n <= 0 || count < 1 || count > kWgradBatch)
        return fail(PPO_E_INVALID, "ppo_conv3x3_backward_weight_slabs_batch_f32: n must be positive, count in 1..%d", kWgradBatch);
    if (!ins || !dys || !workspaces || !n_slabs)
        return fail(PPO_E_INVALID, "ppo_conv3x3_backward_weight_slabs_batch_f32: null pointer");
    WgradBatch b{};
    for (int k = 0; k < count; ++k) {
        if (!ins[k] || !dys[k] || !workspaces[k])
            return fail(PPO_E_INVALID, "ppo_conv3x3_backward_weight_slabs_batch_f32: null pointer in problem %d", k);
        b.in[k] = ins[k];
        b.dy[k] = dys[k];
        b.partial[k] = static_cast<float *>(workspaces[k]);
    }
    hipStream_t st = as_stream(stream);
    float *ws = b.partial[0];
    switch (in_mode) {
        case IN_NONE: return dispatch_wgrad<IN_NONE>(cin, cout, h, w, b.in[0], b.dy[0], nullptr, nullptr, ws, workspace_bytes, n, 0, st, n_slabs, &b, count);
        case IN_RELU: return dispatch_wgrad<IN_RELU>(cin, cout, h, w, b.in[0], b.dy[0], nullptr, nullptr, ws, workspace_bytes, n, 0, st, n_slabs, &b, count);
        case IN_U8: return dispatch_wgrad<IN_U8>(cin, cout, h, w, b.in[0], b.dy[0], nullptr, nullptr, ws, workspace_bytes, n, 0, st, n_slabs, &b, count);
    }
    return fail(PPO_E_INVALID, "ppo_conv3x3_backward_weight_slabs_batch_f32: unknown in_mode %d", in_mode);
}

extern "C" int ppo_conv3x3_backward_weight_slabs_batch_mixed_f32(const void *const *ins, const int *relu, const float *const *dys,
                                                                 void *const *workspaces, size_t workspace_bytes, int count,
                                                                 int n, int cin, int cout, int h, int w, int *n_slabs,
                                                                 void *stream)
{
    using namespace ppo;
    if (n <= 0 || count < 1 || count > kWgradBatch)
        return fail(PPO_E_INVALID, "ppo_conv3x3_backward_weight_slabs_batch_mixed_f32: n must be positive, count in 1..%d", kWgradBatch);
    if (!ins || !relu || !dys || !workspaces || !n_slabs)
        return fail(PPO_E_INVALID, "ppo_conv3x3_backward_weight_slabs_batch_mixed_f32: null pointer");
    WgradBatch b{};
    for (int k = 0; k < count; ++k) {
        if (!ins[k] || !dys[k] || !workspaces[k])
            return fail(PPO_E_INVALID, "ppo_conv3x3_backward_weight_slabs_batch_mixed_f32: null pointer in problem %d", k);
        b.in[k] = ins[k];
        b.dy[k] = dys[k];
        b.partial[k] = static_cast<float *>(workspaces[k]);
        b.relu_floor[k] = relu[k] ? 0.f : -__builtin_inff();
    }
    return dispatch_wgrad<IN_RELU>(cin, cout, h, w, b.in[0], b.dy[0], nullptr, nullptr, b.partial[0], workspace_bytes, n, 0,
                                   as_stream(stream), n_slabs, &b, count);
}

extern "C" int ppo_conv3x3_backward_weight_slabs_pooled_f32(const void *in, int in_mode, const float *g,
                                                            const uint8_t *argmax, void *workspace, size_t workspace_bytes,
                                                            int n, int cin, int cout, int h, int w, int *n_slabs,
                                                            void *stream)
{
    using namespace ppo;
    if (n <= 0) return fail(PPO_E_INVALID, "ppo_conv3x3_backward_weight_slabs_pooled_f32: n must be positive");
    if (!in || !g || !argmax || !workspace || !n_slabs)
        return fail(PPO_E_INVALID, "ppo_conv3x3_backward_weight_slabs_pooled_f32: null pointer");
    hipStream_t st = as_stream(stream);
    float *ws = static_cast<float *>(workspace);
    switch (in_mode) {
        case IN_NONE: return dispatch_wgrad<IN_NONE>(cin, cout, h, w, in, g, nullptr, nullptr, ws, workspace_bytes, n, 0, st, n_slabs, nullptr, 1, argmax);
        case IN_U8: return dispatch_wgrad<IN_U8>(cin, cout, h, w, in, g, nullptr, nullptr, ws, workspace_bytes, n, 0, st, n_slabs, nullptr, 1, argmax);
    }
    return fail(PPO_E_INVALID, "ppo_conv3x3_backward_weight_slabs_pooled_f32: in_mode %d has no pooled-gradient kernel", in_mode);
}

extern "C" int ppo_conv3x3_backward_weight_slabs_pooled_indexed_f32(const void *in, const int32_t *index, int in_mode,
                                                                    const float *g, const uint8_t *argmax, void *workspace,
                                                                    size_t workspace_bytes, int n, int cin, int cout, int h,
                                                                    int w, int *n_slabs, void *stream)
{
    if (index && in_mode != ppo::IN_U8)
        return ppo::fail(PPO_E_INVALID, "ppo_conv3x3_backward_weight_slabs_pooled_indexed_f32: the index applies to uint8 observations");
    ppo::t_wgrad_in_index = index;
    const int rc = ppo_conv3x3_backward_weight_slabs_pooled_f32(in, in_mode, g, argmax, workspace, workspace_bytes, n, cin, cout, h,
                                                                w, n_slabs, stream);
    ppo::t_wgrad_in_index = nullptr;
    return rc;
}

extern "C" int ppo_conv3x3_backward_weight_pooled_supported(int cin, int cout, int h, int w)
{
    return cout == 16 && ((h == 84 && w == 84 && (cin == 4 || cin == 5)) || (h == 64 && w == 64 && (cin == 3 || cin == 4)));
}

extern "C" int ppo_conv3x3_wgrad_reduce_f32(const ppo_wgrad_job *jobs, int n_jobs, void *stream)
{
    using namespace ppo;
    if (n_jobs < 0 || n_jobs > kMaxJobs) return fail(PPO_E_INVALID, "ppo_conv3x3_wgrad_reduce_f32: 0..%d jobs per call", kMaxJobs);
    if (n_jobs == 0) return PPO_OK;
    if (!jobs) return fail(PPO_E_INVALID, "ppo_conv3x3_wgrad_reduce_f32: null jobs");
    ReduceJobs rj;
    int max_blocks = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const ppo_wgrad_job &j = jobs[i];
        if (!j.slabs || !j.dweight || j.n_slabs <= 0 || j.cin <= 0 || j.cout <= 0 || j.cout % 16)
            return fail(PPO_E_INVALID, "ppo_conv3x3_wgrad_reduce_f32: bad job %d", i);
        rj.j[i] = j;
        const int cinp = (j.cin + 3) / 4 * 4;
        const int jp = ((9 * cinp + 1) + 15) / 16 * 16;
        const int blocks = (j.cout * (jp / 4) + 15) / 16;
        max_blocks = blocks > max_blocks ? blocks : max_blocks;
    }
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_jobs_kernel, dim3(max_blocks, n_jobs), dim3(256), 0, as_stream(stream), rj);
    return check_launch("conv3x3_wgrad_reduce_jobs_kernel");
}

// 3x3 / stride 1 / pad 1 convolution as an implicit GEMM on the f32 MFMA
// (v_mfma_f32_16x16x4_f32), NCHW float32 — the IMPALA-CNN contraction
// (reference: rl/impala.py:61-62,96 via torch.nn.Conv2d).
//
// One kernel serves forward and backward-data:
//   forward        out[n,o,y,x] = b[o] + sum_{i,ky,kx} f(in[n,i,y+ky-1,x+kx-1]) * w[o,i,ky,kx]  (+ residual)
//   backward-data  dx[n,i,y,x]  = (sum_{o,ky,kx} dy[n,o,y-ky+1,x-kx+1] * w[o,i,ky,kx]) * [pre[n,i,y,x] > 0] (+ dres)
// The second is the first with the weight tensor read transposed and flipped.
// f() is the input transform: identity, ReLU (the pre-activation residual blocks of
// rl/impala.py:73-78 store pre-activations; ReLU is applied when the operand is read), or
// uint8 -> x/255 (rl/models.py:842-848).
//
// Mapping: a workgroup of 8 (16-channel layers) or 4 (32-channel layers) waves walks (image, band of TR
// output rows) items for all output channels.  The input band sits in LDS as planar [ci][row][col] with one
// halo ROW above and below (zero outside the image) and NO halo columns: rows keep the image's row stride,
// so a channel's band is one contiguous run that 16-byte LDS-DMA requests move, and the x-1 / x+1 taps at
// the image edge are zeroed by the med3 that also applies ReLU-on-read (plane stride = 16 mod 32 banks).
// The weights live in REGISTERS for the whole
// kernel (each lane's A-operand slice wa[NT][K/4], loaded once: the filter bank is at most 36 KB and
// every wave needs all of it for every item), as does the bias.  GEMM view: M = output channel
// (MFMA "i"), N = pixel (MFMA "j"), K = 9*CIN with k = tap*CINP + ci.
// A wave holds MT pixel tiles x NT channel tiles of 16x16 accumulators; every activation read is a
// ds_read_b32 at an immediate offset from a per-lane base, software-pipelined two K steps ahead
// (hand-unrolled, sched_barrier-fenced); output stores are deferred into the next item's K loop.
//
// Pipeline (float inputs): the band is DOUBLE-BUFFERED and filled by LDS-DMA (global_load_lds): the
// next item's band is requested right after the barrier that publishes the current one, so HBM latency,
// the store drain of the previous epilogue and the MFMA work of the current item overlap, with one
// barrier per item.  (Measured before this: waves spent ~50 % of their life in s_waitcnt/s_barrier.)
// uint8 observations (first conv only) are staged through registers with the /255 fused.
#include "common.h"
#include "conv_stage.h"
#include "mfma.h"

namespace ppo {

// conv1_pool.hip: the first layer on uint8 observations, pooled out of the MFMA accumulators
bool conv1_pool_supported(int cin, int cout, int h, int w, bool train);
int conv1_pool_forward(const void *in, const int32_t *in_index, const float *w, bool packed, const float *bias, float *out,
                       uint8_t *argmax, int n, int cin, int h, int w_, hipStream_t st);

namespace {

// Waves per workgroup (NW) is a per-geometry choice: 8 for the 16-channel layers (several workgroups fit a
// CU), 4 for the 32-channel layers, whose 144 weight registers per lane allow only 8 waves per CU in total:
// two independent 4-wave workgroups then interleave their barrier / epilogue phases with each other's K
// loops, where one 8-wave workgroup leaves the MFMA pipe idle while all its waves sit in the same phase
// (stamps build: 29 % of an item outside the K loop).

// Diagnostic build only (tools/conv_tune -DPPO_TUNE_STAMPS): s_memtime stamps around the phases of an
// item, summed per phase over all waves into a buffer nothing else reads (cdna_hip_programming.md §7).
#ifdef PPO_TUNE_STAMPS
__device__ unsigned long long ppo_tune_stamps[8 * 8 * 1024];  // [workgroup][wave][slot], no atomics
#define PPO_STAMP(var)                                                                                  \
    unsigned long long var;                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");                         \
    __builtin_amdgcn_sched_barrier(0);
#define PPO_STAMP_ADD(slot, t1, t0) \
    if (lane == 0) ppo_tune_stamps[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + slot] += (t1) - (t0);
#else
#define PPO_STAMP(var)
#define PPO_STAMP_ADD(slot, t1, t0)
#endif

// Set by the *_packed_f32 entry points around their dispatch: `weight` is then the pre-packed A operand
// (ppo_conv3x3_pack_weights_f32) instead of the raw [O][I][3][3] tensor.  Host-side, thread-local, consumed by the
// launch a few frames down the same call.
thread_local int t_weights_packed = 0;
// images of the next conv3x3_pool launch are read through this index (ppo_conv3x3_pool_forward_packed_indexed_f32): the
// minibatch gather of the observations happens in the first convolution's own loads
thread_local const int32_t *t_in_index = nullptr;

// Packed A operand of a kernel with (kernel-side) CIN input and COUT output channels: for K step s = tap*CINP/4 + cs
// and channel tile n, lane (l15, g) holds w(co = n*16 + l15, ci = cs*4 + g, tap); four consecutive steps form one
// float4 so a wave's load is 1 KB contiguous:  packed[((s/4 * NT + n) * 64 + lane) * 4 + s % 4].
__host__ __device__ constexpr int packed_floats(int cin, int cout)
{
    const int ks = 9 * (((cin + 3) / 4 * 4) / 4), nt = ((cout + 15) / 16 * 16) / 16;
    return (ks + 3) / 4 * 4 * nt * 64;
}

struct PackJobs {
    ppo_pack_job j[32];
};

__global__ __launch_bounds__(256) void pack_weights_kernel(PackJobs jobs)
{
    const ppo_pack_job &job = jobs.j[blockIdx.y];
    // kernel-side channel counts: the transposed (backward-data) kernel contracts over the forward's outputs
    const int cin = job.transposed ? job.cout : job.cin, cout = job.transposed ? job.cin : job.cout;
    const int cinp = (cin + 3) / 4 * 4, nt = ((cout + 15) / 16 * 16) / 16;
    const int ks = 9 * (cinp / 4);
    const int total = (ks + 3) / 4 * 4 * nt * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int e = i & 3, lane = (i >> 2) & 63, rest = i >> 8;  // rest = s4 * nt + n
        const int n = rest % nt, s = (rest / nt) * 4 + e;
        float v = 0.f;
        if (s < ks) {
            const int tap = s / (cinp / 4), ci = (s % (cinp / 4)) * 4 + (lane >> 4), co = n * 16 + (lane & 15);
            if (ci < cin && co < cout)
                v = job.transposed ? job.weight[((size_t)ci * cout + co) * 9 + (8 - tap)]  // raw [O = ci][I = co], flipped
                                   : job.weight[((size_t)co * cin + ci) * 9 + tap];
        }
        job.packed[i] = v;
    }
}

template <int CIN, int COUT, int H, int W, int TR, bool DOUBLE>
struct ConvCfg {
    static constexpr int CINP = (CIN + 3) / 4 * 4;  // k-steps of 4 channels
    static constexpr int COUTP = (COUT + 15) / 16 * 16;
    static constexpr int NT = COUTP / 16;
    static constexpr int ROWS = TR + 2;             // one halo row above and below; no halo columns
    static constexpr int G = 4;                     // guard floats around the rows of a plane
    static constexpr int PLANE_RAW = ROWS * W + 2 * G;
    static constexpr int PLANE = PLANE_RAW + ((16 - PLANE_RAW % 32) + 32) % 32;  // = 16 (mod 32)
    static constexpr int K = 9 * CINP;
    static constexpr int NBANDS = (H + TR - 1) / TR;
    static constexpr int NPIX = TR * W;
    static constexpr int MTILES = (NPIX + 15) / 16;
    static constexpr int LDS_IN = CINP * PLANE;  // floats, one band buffer
    static constexpr int NBUF = DOUBLE ? 2 : 1;
    static constexpr size_t LDS_BYTES = (size_t)(NBUF * LDS_IN) * 4;
};

// WIDE epilogue (maps whose channel planes and bands are whole float4s): the accumulators of a wave's tiles are
// transposed through a wave-private LDS scratch from the MFMA layout (lane = pixel x 4 channels) to lane = 4
// consecutive pixels of one channel, so the ReLU gate, the residual and the result travel as ONE 16-byte access per
// lane and tile instead of four 4-byte ones.  The per-lane dword form issued 24-48 memory instructions per group and
// wave: the CU's one memory pipeline, not the MFMA pipe, set the pace (stamps: 40 % of a backward-data item spent
// issuing the gate / residual loads).  Same arithmetic per element, so the bits do not change.
constexpr int kScrPitch = 20;  // floats per scratch row: 16 pixels + 4, so channel rows 4 apart are 16 banks apart
template <int MT, int NT>
constexpr int conv_scratch_floats() { return MT * NT * 16 * kScrPitch; }  // per wave

template <int CIN, int COUT, int H, int W, int TR, int MT, int NW, int IN_MODE, bool TRANSPOSED, bool PACKED, bool WIDE>
__global__ __launch_bounds__(NW * 64) void conv3x3_kernel(
    const void *__restrict__ in_, const float *__restrict__ w, const float *__restrict__ bias,
    const float *__restrict__ residual, const float *__restrict__ mask_src, float *__restrict__ out,
    int n_images)
{
    constexpr bool DMA = IN_MODE != IN_U8;
    constexpr int kConvWaves = NW, kConvThreads = NW * 64;
    using C = ConvCfg<CIN, COUT, H, W, TR, DMA>;
    extern __shared__ __align__(16) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l15 = lane & 15;
    const int g = lane >> 4;

    // ---- weights: the MFMA A operand of K step s is the same for every pixel tile, so each lane keeps
    // its A values for ALL steps in registers for the lifetime of the workgroup (NT * KS VGPRs: 36 for
    // 16->16, 144 for 32->32) for the lifetime of the workgroup: no LDS reads for A in the K loop.
    // They get there through LDS: the raw [O][I][3][3] tensor is copied in with coalesced loads (rows padded
    // to an odd stride so the per-lane gather below is bank-conflict free), then each lane picks its values.
    // (Gathering straight from global memory cost ~64 cache lines per load instruction: measured ~8 us of a
    // 38 us launch for the 32->32 layers, all of it before the first MFMA.)
    constexpr int KS = 9 * (C::CINP / 4);
    constexpr int WR = TRANSPOSED ? CIN : COUT;        // rows of the raw tensor
    constexpr int WL = (TRANSPOSED ? COUT : CIN) * 9;  // row length
    constexpr int WLP = WL | 1;                        // odd LDS row stride
    // The transposed (backward-data) gather already touches few cache lines per request (lanes step through
    // the contiguous [I][3][3] part of the tensor) and measured slightly faster straight from global memory.
    constexpr bool VIA_LDS = !TRANSPOSED;
    float wa[C::NT][KS];
    if constexpr (PACKED) {
        // `w` is this kernel's A operand already in per-lane order (ppo_conv3x3_pack_weights_f32): 16-byte coalesced
        // loads, no LDS round trip, no barrier — they are in flight while the band buffers are zeroed and the
        // first band is requested, and are first needed in the K loop.
        const float4 *pw = reinterpret_cast<const float4 *>(w);
#pragma unroll
        for (int s4 = 0; s4 < (KS + 3) / 4; ++s4)
#pragma unroll
            for (int n = 0; n < C::NT; ++n) {
                const float4 v = pw[(s4 * C::NT + n) * 64 + lane];
                if (4 * s4 + 0 < KS) wa[n][4 * s4 + 0] = v.x;
                if (4 * s4 + 1 < KS) wa[n][4 * s4 + 1] = v.y;
                if (4 * s4 + 2 < KS) wa[n][4 * s4 + 2] = v.z;
                if (4 * s4 + 3 < KS) wa[n][4 * s4 + 3] = v.w;
            }
    } else {
        if constexpr (VIA_LDS) {
            for (int i = tid; i < WR * WL; i += kConvThreads) smem[(i / WL) * WLP + i % WL] = w[i];
            __syncthreads();
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int tap = s / (C::CINP / 4);
            const int ci = (s % (C::CINP / 4)) * 4 + g;  // this lane group contracts k = 4s + g
#pragma unroll
            for (int n = 0; n < C::NT; ++n) {
                const int co = n * 16 + l15;
                float v = 0.f;
                if (ci < CIN && co < COUT)
                    v = TRANSPOSED ? w[((size_t)ci * COUT + co) * 9 + (8 - tap)]  // w[o=ci][i=co], taps flipped
                                   : smem[co * WLP + ci * 9 + tap];
                wa[n][s] = v;
            }
        }
        if constexpr (VIA_LDS) __syncthreads();  // every lane has its weights: the region is reused for the input bands
    }
    // this lane's bias values (channels n*16 + g*4 + r), loaded once
    float bias_r[C::NT][4];
#pragma unroll
    for (int n = 0; n < C::NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = n * 16 + g * 4 + r;
            bias_r[n][r] = (bias && co < COUT) ? bias[co] : 0.f;
        }
    // WIDE: lane = (channel ch of a 16-channel tile, 4 consecutive pixels quad*4..+3)
    const int ch = lane >> 2, quad = lane & 3;
    float bias_c[C::NT];
#pragma unroll
    for (int n = 0; n < C::NT; ++n) bias_c[n] = (WIDE && bias && n * 16 + ch < COUT) ? bias[n * 16 + ch] : 0.f;
    float *scr = smem + C::NBUF * C::LDS_IN + wave * conv_scratch_floats<MT, C::NT>();  // wave-private, WIDE only
    if constexpr (DMA) {
        zero_lds<C::NBUF * C::LDS_IN, kConvThreads>(smem, tid);  // halo columns + padded channels stay zero
        __syncthreads();
    }

    const int n_items = n_images * C::NBANDS;
    auto stage = [&](int item, float *dst) {
        const int img = item / C::NBANDS;
        const int y0 = (item % C::NBANDS) * TR;
        if constexpr (DMA)
            stage_band_chunk_dma<CIN, H, W, C::ROWS, C::PLANE, C::G, kConvWaves>(static_cast<const float *>(in_), img, y0, dst, tid);
        else
            stage_band_flat<CIN, C::CINP, H, W, C::ROWS, C::PLANE, C::G, IN_MODE, kConvWaves>(in_, img, y0, dst, tid);
    };
    if (DMA && (int)blockIdx.x < n_items) stage(blockIdx.x, smem);

    // ---- per-lane constants of this wave's pixel-tile groups (identical for every item)
    constexpr int GROUPS = (C::MTILES + MT - 1) / MT;
    constexpr int NGW = (GROUPS + kConvWaves - 1) / kConvWaves;  // group slots per wave
    int pix[NGW][MT];   // pixel index inside the band = its offset inside a channel plane of the band
    int lofs[NGW][MT];  // LDS offset of the pixel's 3x3 window origin (+ this lane group's channel plane)
    // The band has no halo columns: at the image's left (right) edge the x-1 (x+1) taps read a neighbouring
    // row's pixel and are forced to 0 instead.  hi_l / hi_r are the upper clamp of a med3(v, lo, hi): 0 for
    // an edge pixel, +inf otherwise; with lo = 0 the same instruction is also the ReLU-on-read, with
    // lo = -hi it only masks (clamp to [0, 0] or to [-inf, +inf]).
    float hi_l[NGW][MT], hi_r[NGW][MT];
#pragma unroll
    for (int q = 0; q < NGW; ++q)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int p = ((wave + q * kConvWaves) * MT + m) * 16 + l15;
            pix[q][m] = p;
            const int pc = p < C::NPIX ? p : 0;
            lofs[q][m] = C::G + pc - 1 + g * C::PLANE;  // window origin = (row, x - 1); rows are W apart
            hi_l[q][m] = (pc % W == 0) ? 0.f : INFINITY;
            hi_r[q][m] = (pc % W == W - 1) ? 0.f : INFINITY;
        }

    // results of the last group a wave computed are STORED one barrier later (after the next item's
    // barrier, before its DMA request): the barrier's vmcnt(0) then only ever waits for memory
    // operations that were issued a whole item ago, never for stores it has just issued.
    float pend[MT][C::NT][4];  // WIDE: element e = pixel pend_off + e of channel n*16 + ch
    int pend_off[MT];
    bool pend_live[MT];
    float *pend_base = out;
#pragma unroll
    for (int m = 0; m < MT; ++m) pend_live[m] = false;
    auto flush = [&]() {
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if (pend_live[m]) {
                if constexpr (WIDE) {
#pragma unroll
                    for (int n = 0; n < C::NT; ++n)
                        if (n * 16 + ch < COUT)
                            *reinterpret_cast<float4 *>(pend_base + pend_off[m] + (n * 16 + ch) * (H * W)) =
                                make_float4(pend[m][n][0], pend[m][n][1], pend[m][n][2], pend[m][n][3]);
                    pend_live[m] = false;
                    continue;
                }
#pragma unroll
                for (int n = 0; n < C::NT; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = n * 16 + g * 4 + r;
                        if (co < COUT) {
#ifdef PPO_TUNE_NO_STORE  // tools/conv_tune ablation build only: keep the value alive, skip the store
                            asm volatile("" ::"v"(pend[m][n][r]));
#else
                            pend_base[pend_off[m] + co * (H * W)] = pend[m][n][r];
#endif
                        }
                    }
                pend_live[m] = false;
            }
    };

    const bool gate_off = mask_src == nullptr;
    int buf = 0;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int img = item / C::NBANDS;
        const int y0 = (item % C::NBANDS) * TR;
        const int plim = min(C::NPIX, (H - y0) * W);  // pixels of this band that exist in the image
        const size_t img_off = (size_t)img * COUT * H * W + (size_t)y0 * W;
        float *s_in = smem + buf * C::LDS_IN;
        PPO_STAMP(t_top)
        if constexpr (DMA) {
            __syncthreads();  // this item's band has landed (vmcnt(0)) and every wave is done with the other buffer
            flush();
#ifndef PPO_TUNE_NO_STAGE  // tools/conv_tune ablation build only
            if (item + (int)gridDim.x < n_items) stage(item + gridDim.x, smem + (buf ^ 1) * C::LDS_IN);
#endif
        } else {
            __syncthreads();  // previous item's readers are done with s_in
            flush();
            stage(item, s_in);
            __syncthreads();
        }
        PPO_STAMP(t_staged)
        PPO_STAMP_ADD(0, t_staged, t_top)  // barrier + staging issue

        // ---- MFMA main loop: MT pixel tiles per group
#pragma unroll
        for (int q = 0; q < NGW; ++q) {
            if (wave + q * kConvWaves >= GROUPS) break;
            PPO_STAMP(t_g0)
            flush();  // an earlier group of this item (NGW > 1)
            f32x4 acc[C::NT][MT];
#pragma unroll
            for (int n = 0; n < C::NT; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

            // epilogue operands (ReLU gate source, residual) are requested NOW and consumed after the K
            // loop, so their memory latency hides under the MFMAs
            bool live[MT];
            float gate[MT][C::NT][4], res[MT][C::NT][4];
            // (range-checked buffer reads, common.h: lanes without an output element, and absent tensors, read zeros;
            // a zero gate only ever meets an element that is not stored, or gate_off)
            const __amdgpu_buffer_rsrc_t mask_img = buffer_of(mask_src ? mask_src + img_off : nullptr, mask_src != nullptr);
            const __amdgpu_buffer_rsrc_t res_img = buffer_of(residual ? residual + img_off : nullptr, residual != nullptr);
            int p4[MT];  // WIDE: first of this lane's 4 pixels of tile m
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if constexpr (WIDE) {
                    p4[m] = ((wave + q * kConvWaves) * MT + m) * 16 + quad * 4;
                    live[m] = p4[m] < plim;  // plim is a multiple of 4: a quad is whole or absent
#pragma unroll
                    for (int n = 0; n < C::NT; ++n) {
                        const int oi = p4[m] + (n * 16 + ch) * (H * W);
                        const int off = (live[m] && n * 16 + ch < COUT) ? oi * 4 : kOutside;
                        const float4 gv = buffer_f32x4(mask_img, off);
                        const float4 rv = buffer_f32x4(res_img, off);
                        gate[m][n][0] = gv.x, gate[m][n][1] = gv.y, gate[m][n][2] = gv.z, gate[m][n][3] = gv.w;
                        res[m][n][0] = rv.x, res[m][n][1] = rv.y, res[m][n][2] = rv.z, res[m][n][3] = rv.w;
                    }
                    continue;
                }
                live[m] = pix[q][m] < plim;
#pragma unroll
                for (int n = 0; n < C::NT; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = n * 16 + g * 4 + r;
                        const int oi = pix[q][m] + co * (H * W);
                        const int off = (live[m] && co < COUT) ? oi * 4 : kOutside;
                        gate[m][n][r] = buffer_f32(mask_img, off);
                        res[m][n][r] = buffer_f32(res_img, off);
                    }
            }
            int base[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) base[m] = lofs[q][m] + buf * C::LDS_IN;

            PPO_STAMP(t_g1)
            PPO_STAMP_ADD(1, t_g1, t_g0)  // group prologue (gate/residual requests)
            // K loop, fully unrolled (every LDS offset is an immediate) and software-pipelined by hand:
            // the operands of step s+PF are requested before the MFMAs of step s issue, and a
            // sched_barrier per step keeps the compiler from re-batching the reads (left alone it either
            // waits on each step's reads right before its MFMAs, or hoists all of them and spills).
            if constexpr (!TRANSPOSED) {
                // Forward: steps go through the pipeline in blocks of SB: [wait for block j's operands, requested
                // a whole block of MFMAs ago] -> [their ReLU / edge-mask med3s] -> [request block j+1] -> [MFMAs
                // of block j].  At the wait only block j's reads are outstanding, so the compiler's conservative
                // lgkmcnt(0) is exact, and no MFMA waits on a med3 issued just before it.  Measured: forward
                // 0.559 -> 0.537 ms per 256 batch; the transposed (backward-data) kernels, whose epilogue
                // prefetches compete for the same counters, were 5 % slower with it and keep the per-step form.
                constexpr int SB = (MT * C::NT >= 4) ? 2 : 4;
                constexpr int NB = (KS + SB - 1) / SB;
                float raw[2][SB][MT];
                auto load_block = [&](int j, float (&r)[SB][MT]) {
#pragma unroll
                    for (int u = 0; u < SB; ++u) {
                        const int s = j * SB + u;
                        if (s < KS) {
                            const int tap = s / (C::CINP / 4);
                            const int cs = s % (C::CINP / 4);
                            const int tap_off = (tap / 3) * W + (tap % 3);
#pragma unroll
                            for (int m = 0; m < MT; ++m) r[u][m] = smem[base[m] + cs * 4 * C::PLANE + tap_off];
                        }
                    }
                };
                load_block(0, raw[0]);
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    float bv[SB][MT];
#pragma unroll
                    for (int u = 0; u < SB; ++u) {
                        const int s = j * SB + u;
                        if (s < KS) {
                            const int kx = (s / (C::CINP / 4)) % 3;
#pragma unroll
                            for (int m = 0; m < MT; ++m) {
                                float x = raw[j & 1][u][m];
                                if (kx != 1) {
                                    const float hi = kx == 0 ? hi_l[q][m] : hi_r[q][m];
                                    x = __builtin_amdgcn_fmed3f(x, IN_MODE == IN_RELU ? 0.f : -hi, hi);
                                } else if (IN_MODE == IN_RELU) {
                                    x = relu1(x);
                                }
                                bv[u][m] = x;
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (j + 1 < NB) load_block(j + 1, raw[(j + 1) & 1]);
#pragma unroll
                    for (int u = 0; u < SB; ++u) {
                        const int s = j * SB + u;
                        if (s < KS) {
#pragma unroll
                            for (int m = 0; m < MT; ++m)
#pragma unroll
                                for (int n = 0; n < C::NT; ++n) acc[n][m] = mfma16(wa[n][s], bv[u][m], acc[n][m]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                constexpr int PF = 2;
                float b[PF + 1][MT];
                auto load_step = [&](int s, float (&bb)[MT]) {
                    const int tap = s / (C::CINP / 4);
                    const int cs = s % (C::CINP / 4);
                    const int tap_off = (tap / 3) * W + (tap % 3);
    #pragma unroll
                    for (int m = 0; m < MT; ++m) bb[m] = smem[base[m] + cs * 4 * C::PLANE + tap_off];
                };
    #pragma unroll
                for (int s = 0; s < PF; ++s) load_step(s, b[s]);
    #pragma unroll
                for (int s = 0; s < KS; ++s) {
                    if (s + PF < KS) load_step(s + PF, b[(s + PF) % (PF + 1)]);
    #pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        float bv = b[s % (PF + 1)][m];
                        // one med3 per operand: ReLU on read (pre-activation input) and/or the edge mask
                        const int kx = (s / (C::CINP / 4)) % 3;
                        if (kx != 1) {
                            const float hi = kx == 0 ? hi_l[q][m] : hi_r[q][m];
                            bv = __builtin_amdgcn_fmed3f(bv, IN_MODE == IN_RELU ? 0.f : -hi, hi);
                        } else if (IN_MODE == IN_RELU) {
                            bv = relu1(bv);
                        }
    #pragma unroll
                        for (int n = 0; n < C::NT; ++n) acc[n][m] = mfma16(wa[n][s], bv, acc[n][m]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            PPO_STAMP(t_g2)
            PPO_STAMP_ADD(2, t_g2, t_g1)  // K loop

            // ---- epilogue: lane holds pixel (lane & 15) x channels g*4 .. g*4+3 of each tile; the values
            // are parked in registers and written by the next flush()
            pend_base = out + img_off;
            if constexpr (WIDE) {
                // MFMA layout -> scratch rows [channel][16 pixels]; DS operations of a wave complete in order, so the
                // reads below see the writes of the other lanes without a barrier
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < C::NT; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            scr[((m * C::NT + n) * 16 + g * 4 + r) * kScrPitch + l15] = acc[n][m][r];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    pend_live[m] = live[m];
                    pend_off[m] = p4[m];
#pragma unroll
                    for (int n = 0; n < C::NT; ++n) {
                        const float4 o = *reinterpret_cast<const float4 *>(scr + ((m * C::NT + n) * 16 + ch) * kScrPitch + quad * 4);
                        const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float val = ov[e] + bias_c[n];
                            val = (gate[m][n][e] > 0.f || gate_off) ? val : 0.f;
                            pend[m][n][e] = val + res[m][n][e];
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            } else {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                pend_live[m] = live[m];
                pend_off[m] = pix[q][m];
#pragma unroll
                for (int n = 0; n < C::NT; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float val = acc[n][m][r] + bias_r[n][r];
                        val = (gate[m][n][r] > 0.f || gate_off) ? val : 0.f;
                        pend[m][n][r] = val + res[m][n][r];
                    }
            }
            }
        }
        PPO_STAMP(t_end)
        PPO_STAMP_ADD(3, t_end, t_staged)  // whole compute section of the item (all groups + epilogues)
        PPO_STAMP_ADD(4, t_end, t_top)     // whole item
        if (lane == 0) { PPO_STAMP_ADD(5, 1ull, 0ull) }  // item-waves counted
        if constexpr (DMA) buf ^= 1;
    }
    flush();
}

template <int CIN, int COUT, int H, int W, int TR, int MT, int NW, int IN_MODE, bool TRANSPOSED, bool PACKED, bool WIDE>
int launch_conv_impl(const void *in, const float *w, const float *bias, const float *residual,
                     const float *mask_src, float *out, int n_images, hipStream_t st)
{
    using C = ConvCfg<CIN, COUT, H, W, TR, IN_MODE != IN_U8>;
    auto kern = conv3x3_kernel<CIN, COUT, H, W, TR, MT, NW, IN_MODE, TRANSPOSED, PACKED, WIDE>;
    constexpr int kConvThreads = NW * 64;
    // the band buffers (+ the WIDE epilogue's per-wave scratch), or the padded weight image the prologue stages
    // through the same region
    constexpr size_t kWeightImage = (size_t)(TRANSPOSED ? CIN : COUT) * (((TRANSPOSED ? COUT : CIN) * 9) | 1) * 4;
    constexpr size_t kBands = C::LDS_BYTES + (WIDE ? (size_t)NW * conv_scratch_floats<MT, C::NT>() * 4 : 0);
    constexpr size_t kLdsBytes = kBands > kWeightImage ? kBands : kWeightImage;
    static int wg_per_cu = 0;  // resident workgroups per CU (LDS- and VGPR-limited), queried once
    if (wg_per_cu == 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
        if (e != hipSuccess) return fail(PPO_E_HIP, "conv3x3: hipFuncSetAttribute: %s", hipGetErrorString(e));
        int nb = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, kConvThreads, kLdsBytes);
        if (e != hipSuccess) return fail(PPO_E_HIP, "conv3x3: occupancy query: %s", hipGetErrorString(e));
        wg_per_cu = nb < 1 ? 1 : (nb > 24 / NW ? 24 / NW : nb);
#ifdef PPO_TUNE_WG_PER_CU  // tools/conv_tune experiment build only
        wg_per_cu = PPO_TUNE_WG_PER_CU < wg_per_cu ? PPO_TUNE_WG_PER_CU : wg_per_cu;
#endif
    }
    const int n_items = n_images * C::NBANDS;
    int grid = 256 * wg_per_cu;
    if (grid > n_items) grid = n_items;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kConvThreads), kLdsBytes, st, in, w, bias, residual, mask_src, out,
                       n_images);
    return check_launch("conv3x3_kernel");
}

// raw-weight and packed-weight forms are separate instantiations (a runtime branch between the two prologues cost
// the backward-data kernels 8 %: 1.56 -> 1.70 ms per minibatch)
template <int CIN, int COUT, int H, int W, int TR, int MT, int NW, int IN_MODE, bool TRANSPOSED>
int launch_conv(const void *in, const float *w, const float *bias, const float *residual, const float *mask_src,
                float *out, int n_images, hipStream_t st)
{
    // the 16-byte epilogue needs whole float4s: channel planes and bands a multiple of 4 floats, 16-byte aligned tensors
    constexpr bool kWideGeo = (H * W) % 4 == 0 && (TR * W) % 4 == 0 && IN_MODE != IN_U8;
#ifdef PPO_TUNE_NO_WIDE
    const bool wide = false;
#else
    const bool wide = kWideGeo && aligned(out, 16) && (!residual || aligned(residual, 16)) && (!mask_src || aligned(mask_src, 16));
#endif
    if constexpr (kWideGeo) {
        if (wide) {
            if (t_weights_packed)
                return launch_conv_impl<CIN, COUT, H, W, TR, MT, NW, IN_MODE, TRANSPOSED, true, true>(in, w, bias, residual,
                                                                                                     mask_src, out, n_images, st);
            return launch_conv_impl<CIN, COUT, H, W, TR, MT, NW, IN_MODE, TRANSPOSED, false, true>(in, w, bias, residual,
                                                                                                  mask_src, out, n_images, st);
        }
    }
    if (t_weights_packed)
        return launch_conv_impl<CIN, COUT, H, W, TR, MT, NW, IN_MODE, TRANSPOSED, true, false>(in, w, bias, residual, mask_src,
                                                                                                out, n_images, st);
    return launch_conv_impl<CIN, COUT, H, W, TR, MT, NW, IN_MODE, TRANSPOSED, false, false>(in, w, bias, residual, mask_src,
                                                                                             out, n_images, st);
}

// Supported layer geometries: Atari 84x84 (rl/atari.py) and Procgen 64x64 (rl/procgen.py)
// through the three IMPALA stacks (16, 32, 32 channels; rl/models.py:873).
// TR / MT are chosen so that the pixel-tile groups of a band divide evenly over the 8 waves.
template <int IN_MODE, bool TRANSPOSED>
int dispatch_conv(int cin, int cout, int h, int w_, const void *in, const float *w, const float *bias,
                  const float *residual, const float *mask_src, float *out, int n, hipStream_t st)
{
// FIRST: the obs conv (uint8 or float obs, never ReLU-on-load, never transposed);
// UP: the channel-changing stack-first conv; SAME: everything else.
#define PPO_CONV_CASE(ALLOWED, CI, CO, HH, WW, TR, MT, NW)                                              \
    if constexpr (ALLOWED) {                                                                           \
        if (cin == CI && cout == CO && h == HH && w_ == WW)                                            \
            return launch_conv<CI, CO, HH, WW, TR, MT, NW, IN_MODE, TRANSPOSED>(in, w, bias, residual, \
                                                                                 mask_src, out, n, st); \
    }
#ifndef PPO_NW32
#define PPO_NW32 4
#endif
    constexpr bool FIRST = !TRANSPOSED && IN_MODE != IN_RELU;
    constexpr bool UP = !TRANSPOSED && IN_MODE == IN_NONE;
    constexpr bool DOWN = TRANSPOSED;
    constexpr bool SAME = IN_MODE != IN_U8;
    PPO_CONV_CASE(FIRST, 4, 16, 84, 84, 12, 4, 8)   // 63 tiles -> 16 groups of 4: 2 per wave
    PPO_CONV_CASE(FIRST, 5, 16, 84, 84, 12, 4, 8)
    PPO_CONV_CASE(FIRST, 3, 16, 64, 64, 16, 4, 8)   // 64 tiles -> 16 groups
    PPO_CONV_CASE(FIRST, 4, 16, 64, 64, 16, 4, 8)
    PPO_CONV_CASE(UP, 16, 32, 42, 42, 6, 2, 8)      // 252 px = 16 tiles -> 8 groups of 2: 1 per wave
    PPO_CONV_CASE(UP, 16, 32, 32, 32, 8, 2, 8)      // 256 px = 16 tiles
    PPO_CONV_CASE(DOWN, 32, 16, 42, 42, 6, 2, 8)
    PPO_CONV_CASE(DOWN, 32, 16, 32, 32, 8, 2, 8)
#ifndef PPO_TUNE_C16_MT
#define PPO_TUNE_C16_MT 2
#define PPO_TUNE_C16_NW 8
#endif
    PPO_CONV_CASE(SAME, 16, 16, 42, 42, 6, PPO_TUNE_C16_MT, PPO_TUNE_C16_NW)
    PPO_CONV_CASE(SAME, 16, 16, 32, 32, 8, 2, 8)
    PPO_CONV_CASE(SAME, 32, 32, 21, 21, 6, 2, PPO_NW32)    // 126 px = 8 tiles -> 4 groups of 2: 1 per wave
    PPO_CONV_CASE(SAME, 32, 32, 16, 16, 8, 2, PPO_NW32)    // 128 px = 8 tiles
    PPO_CONV_CASE(SAME, 32, 32, 11, 11, 11, 2, PPO_NW32)   // 121 px = 8 tiles
    PPO_CONV_CASE(SAME, 32, 32, 8, 8, 8, 1, PPO_NW32)      // 64 px = 4 tiles: 1 per wave
#undef PPO_CONV_CASE
    return fail(PPO_E_INVALID, "conv3x3: unsupported geometry cin=%d cout=%d h=%d w=%d transposed=%d", cin,
                cout, h, w_, (int)TRANSPOSED);
}

// ---------------------------------------------------------------------------------------------------------
// Stack-first convolution fused with the 3x3 / stride 2 / pad 1 max-pool that always follows it
// (rl/impala.py:104-105: x = firstconv(x); x = max_pool2d(x, 3, 2, 1)).  The pre-pool map is the largest
// tensor of the network (16 x 84 x 84 floats per sample: 27 % of the forward's HBM traffic when written and
// read back) and nothing else ever needs it: the backward pass uses the pool's argmax only.  Here a
// workgroup computes the 2*PR+1 convolution rows behind PR pooled rows into LDS, pools them from LDS and
// writes only the pooled map (+ the uint8 argmax when training).  One convolution row per band is computed
// twice (1/(2 PR) more MFMA work).  Same K loop, weight registers and band staging as conv3x3_kernel.
template <int CIN, int COUT, int H, int W, int PR, bool DOUBLE>
struct ConvPoolCfg {
    static constexpr int CINP = (CIN + 3) / 4 * 4;
    static constexpr int NT = (COUT + 15) / 16;
    static constexpr int HO = (H + 1) / 2, WO = (W + 1) / 2;
    static constexpr int NBANDS = (HO + PR - 1) / PR;
    static constexpr int CR = 2 * PR + 1;       // convolution rows per band
    static constexpr int ROWS = CR + 2;         // input rows per band
    static constexpr int G = 4;
    static constexpr int PLANE_RAW = ROWS * W + 2 * G;
    static constexpr int PLANE = PLANE_RAW + ((16 - PLANE_RAW % 32) + 32) % 32;  // = 16 (mod 32)
    static constexpr int NPIX = CR * W;
    static constexpr int OPLANE = NPIX + ((4 - NPIX % 16) + 16) % 16;  // = 4 (mod 16): conflict-free channel-strided stores
    static constexpr int MTILES = (NPIX + 15) / 16;
    static constexpr int LDS_IN = CINP * PLANE;
    static constexpr int NBUF = DOUBLE ? 2 : 1;
    static constexpr int LDS_OUT = COUT * OPLANE;
    static constexpr size_t LDS_BYTES = (size_t)(NBUF * LDS_IN + LDS_OUT) * 4;
};

template <int CIN, int COUT, int H, int W, int PR, int MT, int NW, int IN_MODE, bool PACKED>
__global__ __launch_bounds__(NW * 64) void conv3x3_pool_kernel(const void *__restrict__ in_, const float *__restrict__ w,
                                                              const float *__restrict__ bias, float *__restrict__ out,
                                                              uint8_t *__restrict__ argmax, int n_images,
                                                              const int32_t *__restrict__ in_index)
{
    constexpr bool DMA = IN_MODE != IN_U8;
    constexpr int kWaves = NW, kThreads = NW * 64;
    using C = ConvPoolCfg<CIN, COUT, H, W, PR, DMA>;
    extern __shared__ __align__(16) float smem[];
    float *s_out = smem + C::NBUF * C::LDS_IN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l15 = lane & 15;
    const int g = lane >> 4;
    PPO_STAMP(t_entry)

    // weights -> registers through LDS (see conv3x3_kernel)
    constexpr int KS = 9 * (C::CINP / 4);
    constexpr int WL = CIN * 9, WLP = WL | 1;
    float wa[C::NT][KS];
    if constexpr (PACKED) {  // pre-packed A operand (see conv3x3_kernel)
        const float4 *pw = reinterpret_cast<const float4 *>(w);
#pragma unroll
        for (int s4 = 0; s4 < (KS + 3) / 4; ++s4)
#pragma unroll
            for (int n = 0; n < C::NT; ++n) {
                const float4 v = pw[(s4 * C::NT + n) * 64 + lane];
                if (4 * s4 + 0 < KS) wa[n][4 * s4 + 0] = v.x;
                if (4 * s4 + 1 < KS) wa[n][4 * s4 + 1] = v.y;
                if (4 * s4 + 2 < KS) wa[n][4 * s4 + 2] = v.z;
                if (4 * s4 + 3 < KS) wa[n][4 * s4 + 3] = v.w;
            }
    } else {
        for (int i = tid; i < COUT * WL; i += kThreads) smem[(i / WL) * WLP + i % WL] = w[i];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int tap = s / (C::CINP / 4);
            const int ci = (s % (C::CINP / 4)) * 4 + g;
#pragma unroll
            for (int n = 0; n < C::NT; ++n) {
                const int co = n * 16 + l15;
                wa[n][s] = (ci < CIN && co < COUT) ? smem[co * WLP + ci * 9 + tap] : 0.f;
            }
        }
        __syncthreads();
    }
    float bias_r[C::NT][4];
#pragma unroll
    for (int n = 0; n < C::NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = n * 16 + g * 4 + r;
            bias_r[n][r] = (bias && co < COUT) ? bias[co] : 0.f;
        }
    if constexpr (DMA) {
        zero_lds<C::NBUF * C::LDS_IN, kThreads>(smem, tid);
        __syncthreads();
    }

    const int n_items = n_images * C::NBANDS;
    // image `img` of the launch is image in_index[img] of the input tensor (the minibatch permutation), or img itself
    auto src_img = [&](int img) { return in_index ? in_index[img] : img; };
    auto stage = [&](int item, float *dst) {
        const int img = src_img(item / C::NBANDS);
        const int y0 = 2 * (item % C::NBANDS) * PR - 1;  // band row r holds image row y0 + r - 1 = 2*yo0 - 2 + r
        if constexpr (DMA)
            stage_band_chunk_dma<CIN, H, W, C::ROWS, C::PLANE, C::G, kWaves>(static_cast<const float *>(in_), img, y0, dst, tid);
        else
            stage_band_flat<CIN, C::CINP, H, W, C::ROWS, C::PLANE, C::G, IN_MODE, kWaves>(in_, img, y0, dst, tid);
    };
    if (DMA && (int)blockIdx.x < n_items) stage(blockIdx.x, smem);
    // uint8 observations: the next item's band waits in registers while this item computes (four pixels per
    // register where the row length allows dword loads)
    constexpr bool U8X4 = !DMA && IN_MODE == IN_U8 && W % 4 == 0;
    using FM = FlatMap<C::CINP, W, C::ROWS, kWaves>;
    using FM4 = FlatU8Map<CIN, (U8X4 ? W : 4), C::ROWS, kThreads>;
    uint32_t raw[DMA ? 1 : (U8X4 ? FM4::Q : FM::Q)];
    if constexpr (U8X4) {  // padded channels and the guards are never written by the dword path
        zero_lds<C::NBUF * C::LDS_IN, kThreads>(smem, tid);
        __syncthreads();
    }
    auto prefetch = [&](int item) {
        if constexpr (U8X4)
            band_u8x4_load<CIN, H, W, C::ROWS, kThreads>(in_, src_img(item / C::NBANDS), 2 * (item % C::NBANDS) * PR - 1, tid, raw);
        else if constexpr (!DMA)
            band_flat_load<CIN, C::CINP, H, W, C::ROWS, IN_MODE, kWaves>(in_, src_img(item / C::NBANDS),
                                                                         2 * (item % C::NBANDS) * PR - 1, tid, raw);
    };
    if (!DMA && (int)blockIdx.x < n_items) prefetch(blockIdx.x);

    PPO_STAMP(t_loop)
    PPO_STAMP_ADD(7, t_loop, t_entry)  // prologue: weights, zeroing, first prefetch issue
    constexpr int GROUPS = (C::MTILES + MT - 1) / MT;
    constexpr int NGW = (GROUPS + kWaves - 1) / kWaves;
    int buf = 0;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int img = item / C::NBANDS;
        const int yo0 = (item % C::NBANDS) * PR;
        PPO_STAMP(t_top)
        if constexpr (DMA) {
            __syncthreads();  // band landed; previous item's pooling reads of s_out are done
            if (item + (int)gridDim.x < n_items) stage(item + gridDim.x, smem + (buf ^ 1) * C::LDS_IN);
        } else {
            __syncthreads();  // previous item's readers are done with the band and s_out
            if constexpr (U8X4) band_u8x4_store<CIN, W, C::ROWS, C::PLANE, C::G, kThreads>(raw, smem, tid);
            else band_flat_store<C::CINP, W, C::ROWS, C::PLANE, C::G, IN_MODE, kWaves>(raw, smem, tid);
            if (item + (int)gridDim.x < n_items) prefetch(item + gridDim.x);
            __syncthreads();
        }
        PPO_STAMP(t_staged)
        // ---- convolution rows -> s_out.  The last band of a map may hold fewer pooled rows (21 = 5 x 4 + 1): only the
        // 2 pr_n + 1 convolution rows its windows read are computed (a whole band was 9 rows for 1 pooled row: 11 % of
        // the 42x42 layer's MFMAs)
        const int npix_need = (2 * min(PR, C::HO - yo0) + 1) * W;
#pragma unroll 1
        for (int q = 0; q < NGW; ++q) {
            const int grp = wave + q * kWaves;
            if (grp >= GROUPS || grp * MT * 16 >= npix_need) break;
            f32x4 acc[C::NT][MT];
            int base[MT], pixv[MT];
            float hi_l[MT], hi_r[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int p = (grp * MT + m) * 16 + l15;
                pixv[m] = p;
                const int pc = p < C::NPIX ? p : 0;
                base[m] = C::G + pc - 1 + g * C::PLANE + buf * C::LDS_IN;
                hi_l[m] = (pc % W == 0) ? 0.f : INFINITY;
                hi_r[m] = (pc % W == W - 1) ? 0.f : INFINITY;
#pragma unroll
                for (int n = 0; n < C::NT; ++n) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            constexpr int PF = 2;
            float b[PF + 1][MT];
            auto load_step = [&](int s, float (&bb)[MT]) {
                const int tap = s / (C::CINP / 4);
                const int cs = s % (C::CINP / 4);
                const int tap_off = (tap / 3) * W + (tap % 3);
#pragma unroll
                for (int m = 0; m < MT; ++m) bb[m] = smem[base[m] + cs * 4 * C::PLANE + tap_off];
            };
#pragma unroll
            for (int s = 0; s < PF; ++s) load_step(s, b[s]);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + PF < KS) load_step(s + PF, b[(s + PF) % (PF + 1)]);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    float bv = b[s % (PF + 1)][m];
                    const int kx = (s / (C::CINP / 4)) % 3;
                    if (kx != 1) {
                        const float hi = kx == 0 ? hi_l[m] : hi_r[m];
                        bv = __builtin_amdgcn_fmed3f(bv, -hi, hi);
                    }
#pragma unroll
                    for (int n = 0; n < C::NT; ++n) acc[n][m] = mfma16(wa[n][s], bv, acc[n][m]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (pixv[m] < C::NPIX) {
#pragma unroll
                    for (int n = 0; n < C::NT; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int co = n * 16 + g * 4 + r;
                            if (co < COUT) s_out[co * C::OPLANE + pixv[m]] = acc[n][m][r] + bias_r[n][r];
                        }
                }
        }
        PPO_STAMP(t_conv)
        __syncthreads();  // all convolution rows of the band are in s_out
        PPO_STAMP(t_bar2)
        // ---- pool from LDS: ties to the first tap in row-major window order, padding excluded, NaN propagates
        // (same rule as maxpool_fwd_kernel / F.max_pool2d)
        // A wave pass covers RPW pooled rows of WO outputs (lane -> (row in pass, xo)); rows are (co, pr) pairs
        // with pr fastest, so the only divisions are by compile-time constants.
        const int pr_n = min(PR, C::HO - yo0);
        if constexpr (C::WO % 2 == 0 && W % 4 == 0 && C::OPLANE % 4 == 0) {
            // Even pooled width: a thread produces TWO neighbouring outputs (xo = 2 xp, 2 xp + 1).  Their windows cover
            // columns 4 xp - 1 .. 4 xp + 3 of three rows: one 16-byte LDS read + one 4-byte read per row (15 values for
            // 18 taps) instead of 18 strided 4-byte reads, and the pair leaves as one 8-byte store (+ one 2-byte argmax
            // store).  Tasks are the flat (channel, pooled row, pair) index, so every lane works: the row-per-wave form
            // left a third of the lanes idle at 42 outputs per row and spent 58 % of an item here (stamps).
            constexpr int XP = C::WO / 2, NTASK = COUT * PR * XP;
            for (int task = tid; task < NTASK; task += kThreads) {
                const int co = task / (PR * XP), rem = task % (PR * XP);
                const int pr = rem / XP, xp = rem % XP;
                if (pr >= pr_n) continue;
                const int yo = yo0 + pr;
                const float *row0 = s_out + co * C::OPLANE + 2 * pr * W + 4 * xp;  // column 4 xp of conv row 2 pr
                float v[3][5];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float4 q = *reinterpret_cast<const float4 *>(row0 + ky * W);
                    v[ky][0] = xp > 0 ? row0[ky * W - 1] : 0.f;
                    v[ky][1] = q.x, v[ky][2] = q.y, v[ky][3] = q.z, v[ky][4] = q.w;
                }
                float best[2];
                int tap[2];
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    const int xo = 2 * xp + o;
                    float win[3][3];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) win[ky][kx] = v[ky][2 * o + kx];
                    pool_window_max<H % 2 == 0, W % 2 == 0>(win, yo > 0, 2 * yo + 1 < H, xo > 0, 2 * xo + 1 < W, best[o], tap[o]);
                }
                const size_t oi = ((size_t)(img * COUT + co) * C::HO + yo) * C::WO + 2 * xp;
                *reinterpret_cast<float2 *>(out + oi) = make_float2(best[0], best[1]);
                if (argmax) *reinterpret_cast<unsigned short *>(argmax + oi) = (unsigned short)(tap[0] | (tap[1] << 8));
            }
        } else {
        constexpr int RPW = 64 / C::WO;
        const int sub = lane / C::WO, xo = lane % C::WO;
        for (int u0 = wave * RPW; u0 < COUT * PR; u0 += kWaves * RPW) {
            const int u = u0 + sub;
            const int co = u / PR, pr = u % PR;
            if (sub < RPW && co < COUT && pr < pr_n) {
                const float *src = s_out + co * C::OPLANE + 2 * pr * W + 2 * xo - 1;  // window origin (cr = 2 pr, ix = 2 xo - 1)
                const int yo = yo0 + pr;
                float best;
                int best_tap;
                pool_window_lds<H % 2 == 0, W % 2 == 0>(src, W, yo > 0, 2 * yo + 1 < H, xo > 0, 2 * xo + 1 < W, best, best_tap);
                const size_t oi = ((size_t)(img * COUT + co) * C::HO + yo) * C::WO + xo;
#ifdef PPO_TUNE_POOL_NOSTORE  // tools/conv_tune ablation build only
                asm volatile("" ::"v"(best), "v"(best_tap), "v"(oi));
#else
                out[oi] = best;
                if (argmax) argmax[oi] = (uint8_t)best_tap;
#endif
            }
        }
        }
        PPO_STAMP(t_end)
        PPO_STAMP_ADD(0, t_staged, t_top)   // barrier(s) + staging
        PPO_STAMP_ADD(1, t_conv, t_staged)  // convolution groups -> s_out
        PPO_STAMP_ADD(2, t_bar2, t_conv)    // barrier before pooling
        PPO_STAMP_ADD(3, t_end, t_bar2)     // pooling + stores
        PPO_STAMP_ADD(4, t_end, t_top)      // whole item
        if (lane == 0) { PPO_STAMP_ADD(5, 1ull, 0ull) }
        if constexpr (DMA) buf ^= 1;
    }
    PPO_STAMP(t_exit)
    PPO_STAMP_ADD(6, t_exit, t_entry)  // wave lifetime
}

template <int CIN, int COUT, int H, int W, int PR, int MT, int NW, int IN_MODE, bool PACKED>
int launch_conv_pool_impl(const void *in, const float *w, const float *bias, float *out, uint8_t *argmax, int n_images,
                          hipStream_t st)
{
    using C = ConvPoolCfg<CIN, COUT, H, W, PR, IN_MODE != IN_U8>;
    auto kern = conv3x3_pool_kernel<CIN, COUT, H, W, PR, MT, NW, IN_MODE, PACKED>;
    constexpr size_t kWeightImage = (size_t)COUT * ((CIN * 9) | 1) * 4;
    constexpr size_t kLdsBytes = C::LDS_BYTES > kWeightImage ? C::LDS_BYTES : kWeightImage;
    static_assert(kLdsBytes <= 160 * 1024, "band + pre-pool rows must fit the 160 KB of LDS");
    static int wg_per_cu = 0;
    if (wg_per_cu == 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
        if (e != hipSuccess) return fail(PPO_E_HIP, "conv3x3_pool: hipFuncSetAttribute: %s", hipGetErrorString(e));
        int nb = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NW * 64, kLdsBytes);
        if (e != hipSuccess) return fail(PPO_E_HIP, "conv3x3_pool: occupancy query: %s", hipGetErrorString(e));
        wg_per_cu = nb < 1 ? 1 : (nb > 3 ? 3 : nb);
#ifdef PPO_TUNE_CP_WG  // tools/conv_tune experiment build only: force the grid multiple and report what the runtime said
        fprintf(stderr, "conv3x3_pool<%d,%d,%d>: occupancy query %d workgroups per CU, LDS %zu B; forcing %d\n", CIN, COUT, H, nb,
                kLdsBytes, PPO_TUNE_CP_WG);
        wg_per_cu = PPO_TUNE_CP_WG;
#endif
    }
    const int n_items = n_images * C::NBANDS;
    int grid = 256 * wg_per_cu;
    if (grid > n_items) grid = n_items;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), kLdsBytes, st, in, w, bias, out, argmax, n_images, t_in_index);
    return check_launch("conv3x3_pool_kernel");
}

template <int CIN, int COUT, int H, int W, int PR, int MT, int NW, int IN_MODE>
int launch_conv_pool(const void *in, const float *w, const float *bias, float *out, uint8_t *argmax, int n_images,
                     hipStream_t st)
{
    if (t_weights_packed)
        return launch_conv_pool_impl<CIN, COUT, H, W, PR, MT, NW, IN_MODE, true>(in, w, bias, out, argmax, n_images, st);
    return launch_conv_pool_impl<CIN, COUT, H, W, PR, MT, NW, IN_MODE, false>(in, w, bias, out, argmax, n_images, st);
}

template <int IN_MODE>
int dispatch_conv_pool(int cin, int cout, int h, int w_, const void *in, const float *w, const float *bias, float *out,
                       uint8_t *argmax, int n, hipStream_t st)
{
#define PPO_CP_CASE(ALLOWED, CI, CO, HH, WW, PR, MT, NW)                                                   \
    if constexpr (ALLOWED) {                                                                                 \
        if (cin == CI && cout == CO && h == HH && w_ == WW)                                                  \
            return launch_conv_pool<CI, CO, HH, WW, PR, MT, NW, IN_MODE>(in, w, bias, out, argmax, n, st);  \
    }
    constexpr bool FLOAT = IN_MODE == IN_NONE;
    if constexpr (IN_MODE == IN_U8) {
        if (conv1_pool_supported(cin, cout, h, w_, argmax != nullptr))
            return conv1_pool_forward(in, t_in_index, w, t_weights_packed != 0, bias, out, argmax, n, cin, h, w_, st);
    }
#ifndef PPO_CP_PR84  // 84x84 first layer: pooled rows per item / pixel tiles per wave group
#define PPO_CP_PR84 3   // 7 rows x 84 = 37 tiles -> 8 groups of 5; 51 KB of LDS: three workgroups per CU
#define PPO_CP_MT84 5
#endif
#ifndef PPO_CP_PR42
#define PPO_CP_PR42 4   // 9 rows x 42 = 24 tiles -> 8 groups of 3: every wave busy (PR 3: 19 tiles -> 7 groups, 75.7 vs 73.3 us)
#define PPO_CP_MT42 3
#endif
#ifndef PPO_CP_PR21
#define PPO_CP_PR21 6   // 13 rows x 21 = 18 tiles -> 6 groups of 3
#define PPO_CP_MT21 3
#define PPO_CP_NW21 8
#endif
    PPO_CP_CASE(true, 4, 16, 84, 84, PPO_CP_PR84, PPO_CP_MT84, 8)
    PPO_CP_CASE(true, 5, 16, 84, 84, PPO_CP_PR84, PPO_CP_MT84, 8)
    PPO_CP_CASE(true, 3, 16, 64, 64, 4, 5, 8)     // 9 rows x 64 = 36 tiles -> 8 groups of 5 (as the 84x84 case)
    PPO_CP_CASE(true, 4, 16, 64, 64, 4, 5, 8)
    PPO_CP_CASE(FLOAT, 16, 32, 42, 42, PPO_CP_PR42, PPO_CP_MT42, 8)
    PPO_CP_CASE(FLOAT, 16, 32, 32, 32, 4, 3, 8)   // 9 rows x 32 = 18 tiles -> 6 groups
    PPO_CP_CASE(FLOAT, 32, 32, 21, 21, PPO_CP_PR21, PPO_CP_MT21, PPO_CP_NW21)
    PPO_CP_CASE(FLOAT, 32, 32, 16, 16, 8, 3, 8)   // 17 rows x 16 = 17 tiles
#undef PPO_CP_CASE
    return fail(PPO_E_INVALID, "conv3x3_pool: unsupported geometry cin=%d cout=%d h=%d w=%d in_mode=%d", cin, cout, h,
                w_, IN_MODE);
}

}  // namespace
}  // namespace ppo

extern "C" int ppo_conv3x3_forward_f32(const void *in, int in_mode, const float *weight, const float *bias,
                                       const float *residual, float *out, int n, int cin, int cout, int h,
                                       int w, void *stream)
{
    using namespace ppo;
    if (n < 0) return fail(PPO_E_INVALID, "ppo_conv3x3_forward_f32: n < 0");
    if (n == 0) return PPO_OK;
    if (!in || !weight || !out) return fail(PPO_E_INVALID, "ppo_conv3x3_forward_f32: null pointer");
    hipStream_t st = as_stream(stream);
    switch (in_mode) {
        case IN_NONE: return dispatch_conv<IN_NONE, false>(cin, cout, h, w, in, weight, bias, residual, nullptr, out, n, st);
        case IN_RELU: return dispatch_conv<IN_RELU, false>(cin, cout, h, w, in, weight, bias, residual, nullptr, out, n, st);
        case IN_U8: return dispatch_conv<IN_U8, false>(cin, cout, h, w, in, weight, bias, residual, nullptr, out, n, st);
    }
    return fail(PPO_E_INVALID, "ppo_conv3x3_forward_f32: unknown in_mode %d", in_mode);
}

extern "C" int ppo_conv3x3_backward_data_f32(const float *dy, const float *weight, const float *relu_src,
                                             const float *dres, float *dx, int n, int cin, int cout, int h,
                                             int w, void *stream)
{
    using namespace ppo;
    if (n < 0) return fail(PPO_E_INVALID, "ppo_conv3x3_backward_data_f32: n < 0");
    if (n == 0) return PPO_OK;
    if (!dy || !weight || !dx) return fail(PPO_E_INVALID, "ppo_conv3x3_backward_data_f32: null pointer");
    // the transposed op contracts over the forward op's output channels
    return dispatch_conv<IN_NONE, true>(cout, cin, h, w, dy, weight, nullptr, dres, relu_src, dx, n,
                                        as_stream(stream));
}

extern "C" int ppo_conv3x3_pool_forward_f32(const void *in, int in_mode, const float *weight, const float *bias,
                                            float *out, uint8_t *argmax, int n, int cin, int cout, int h, int w,
                                            void *stream)
{
    using namespace ppo;
    if (n < 0) return fail(PPO_E_INVALID, "ppo_conv3x3_pool_forward_f32: n < 0");
    if (n == 0) return PPO_OK;
    if (!in || !weight || !out) return fail(PPO_E_INVALID, "ppo_conv3x3_pool_forward_f32: null pointer");
    hipStream_t st = as_stream(stream);
    switch (in_mode) {
        case IN_NONE: return dispatch_conv_pool<IN_NONE>(cin, cout, h, w, in, weight, bias, out, argmax, n, st);
        case IN_U8: return dispatch_conv_pool<IN_U8>(cin, cout, h, w, in, weight, bias, out, argmax, n, st);
    }
    return fail(PPO_E_INVALID, "ppo_conv3x3_pool_forward_f32: in_mode %d (the stack-first convolution reads raw input)", in_mode);
}

extern "C" int ppo_conv3x3_pool_forward_packed_indexed_f32(const void *in, const int32_t *index, int in_mode, const float *packed,
                                                           const float *bias, float *out, uint8_t *argmax, int n, int cin,
                                                           int cout, int h, int w, void *stream)
{
    if (index && in_mode != ppo::IN_U8)
        return ppo::fail(PPO_E_INVALID, "ppo_conv3x3_pool_forward_packed_indexed_f32: the index applies to uint8 observations");
    ppo::t_weights_packed = 1;
    ppo::t_in_index = index;
    const int rc = ppo_conv3x3_pool_forward_f32(in, in_mode, packed, bias, out, argmax, n, cin, cout, h, w, stream);
    ppo::t_in_index = nullptr;
    ppo::t_weights_packed = 0;
    return rc;
}

extern "C" size_t ppo_conv3x3_packed_floats(int cin, int cout, int transposed)
{
    return (size_t)(transposed ? ppo::packed_floats(cout, cin) : ppo::packed_floats(cin, cout));
}

extern "C" int ppo_conv3x3_pack_weights_f32(const ppo_pack_job *jobs, int n_jobs, void *stream)
{
    using namespace ppo;
    if (n_jobs < 0 || n_jobs > 32) return fail(PPO_E_INVALID, "ppo_conv3x3_pack_weights_f32: 0..32 jobs per call");
    if (n_jobs == 0) return PPO_OK;
    if (!jobs) return fail(PPO_E_INVALID, "ppo_conv3x3_pack_weights_f32: null jobs");
    PackJobs pj;
    for (int i = 0; i < n_jobs; ++i) {
        if (!jobs[i].weight || !jobs[i].packed || jobs[i].cin <= 0 || jobs[i].cout <= 0 || !aligned(jobs[i].packed, 16))
            return fail(PPO_E_INVALID, "ppo_conv3x3_pack_weights_f32: bad job %d", i);
        pj.j[i] = jobs[i];
    }
    hipLaunchKernelGGL(pack_weights_kernel, dim3(40, n_jobs), dim3(256), 0, as_stream(stream), pj);
    return check_launch("pack_weights_kernel");
}

extern "C" int ppo_conv3x3_forward_packed_f32(const void *in, int in_mode, const float *packed, const float *bias,
                                              const float *residual, float *out, int n, int cin, int cout, int h,
                                              int w, void *stream)
{
    ppo::t_weights_packed = 1;
    const int rc = ppo_conv3x3_forward_f32(in, in_mode, packed, bias, residual, out, n, cin, cout, h, w, stream);
    ppo::t_weights_packed = 0;
    return rc;
}

extern "C" int ppo_conv3x3_backward_data_packed_f32(const float *dy, const float *packed, const float *relu_src,
                                                    const float *dres, float *dx, int n, int cin, int cout, int h,
                                                    int w, void *stream)
{
    ppo::t_weights_packed = 1;
    const int rc = ppo_conv3x3_backward_data_f32(dy, packed, relu_src, dres, dx, n, cin, cout, h, w, stream);
    ppo::t_weights_packed = 0;
    return rc;
}

extern "C" int ppo_conv3x3_pool_forward_packed_f32(const void *in, int in_mode, const float *packed, const float *bias,
                                                   float *out, uint8_t *argmax, int n, int cin, int cout, int h,
                                                   int w, void *stream)
{
    ppo::t_weights_packed = 1;
    const int rc = ppo_conv3x3_pool_forward_f32(in, in_mode, packed, bias, out, argmax, n, cin, cout, h, w, stream);
    ppo::t_weights_packed = 0;
    return rc;
}

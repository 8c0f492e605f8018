// The residual blocks of an IMPALA stack as ONE kernel with the image resident in LDS (gfx950).
//
//   q0 = p  + conv1_0(relu(conv0_0(relu(p))))        (rl/impala.py:66-84, ImpalaResidualBlock.forward)
//   q1 = q0 + conv1_1(relu(conv0_1(relu(q0))))       (rl/impala.py:110-114: two blocks per stack)
//
// Why: at a minibatch of 256 every launch on the forward's dependent chain costs ~10 us whatever it computes
// (kernel boundary, weight prologue, first band, tail), and the four 32->32 convolutions of the
// 11x11 stack compute for under 4 us each.  A 32-channel 11x11 map is 15 KB, so a workgroup keeps its image in LDS
// across all four convolutions: no cross-workgroup exchange, no kernel boundary, intermediate maps go to HBM only
// when the backward pass needs them (training) and are never read back.
//
// The backward-data pass of the same blocks is the same kernel with the flipped / transposed packed weights, no ReLU
// on read, and each layer's output gated by its forward pre-activation:
//   da1 = conv1_1^T(g) * [a1 > 0];  g1 = g + conv0_1^T(da1) * [q0 > 0];  da0 = conv1_0^T(g1) * [a0 > 0];
//   g0 = g1 + conv0_0^T(da0) * [p > 0]                       (all four are written: the weight gradients read them)
//
// Per image: X <- p (LDS-DMA);  Y = conv(relu(X));  X += conv(relu(Y));  Y = conv(relu(X));  X += conv(relu(Y)).
// Each convolution is the implicit GEMM of conv3x3.hip (same flat band layout with TR = H, same K order, same
// epilogue arithmetic: results are bit-identical to the four separate launches); its weights are the pre-packed A
// operand (ppo_conv3x3_pack_weights_f32), re-loaded into registers at each layer switch.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "conv_stage.h"
#include "mfma.h"

// Channel-tile groups per workgroup (see StackCfg).  Measured: 2 -> forward 0.458 ms and train step 1.41 ms per 256
// samples, 1 -> 0.465 / 1.44.
#ifndef PPO_TAIL_SPLIT
#define PPO_TAIL_SPLIT 2
#endif

namespace ppo {
namespace {

constexpr int kScrPitchS = 20;  // floats per row of the transposed-epilogue scratch (see conv3x3.hip kScrPitch)

struct StackTailArgs {
    const float *in;       // [n, C, H, W] block input p
    const float *w[4];     // packed weights: block0.conv0, block0.conv1, block1.conv0, block1.conv1
    const float *bias[4];  // forward only
    const float *mask[4];  // backward only: pre-activation map gating each layer's output ([n, C, H, W])
    float *save[4];        // forward: a0, q0, a1, q1 ([n, C, H, W] each); the last is required, the others nullable
                           // backward: da1, dq1_in, da0, dq0_in (all required: the weight gradients read them)
    int n_images;
};

// NW pixel-tile waves x NSPLIT channel-tile groups: wave = ng * NW + pw computes MT pixel tiles x NT / NSPLIT channel
// tiles.  NSPLIT = 2 puts two waves on every SIMD of the CU the image owns, so one wave's operand reads, med3s and
// epilogue issue under the other's MFMAs.
template <int C, int H, int W, int MT, int NW, int NSPLIT>
struct StackCfg {
    static constexpr int ROWS = H + 2;
    static constexpr int G = 4;
    static constexpr int PLANE_RAW = ROWS * W + 2 * G;
    static constexpr int PLANE = PLANE_RAW + ((16 - PLANE_RAW % 32) + 32) % 32;  // = 16 (mod 32)
    static constexpr int NT = C / 16;            // channel tiles of the layer
    static constexpr int NTL = NT / NSPLIT;      // ... of one wave
    static constexpr int WAVES = NW * NSPLIT;
    static constexpr int KS = 9 * (C / 4);
    static constexpr int NPIX = H * W;
    static constexpr int MTILES = (NPIX + 15) / 16;
    static constexpr int LDS_MAP = C * PLANE;  // floats per resident map
    static constexpr size_t LDS_BYTES = (size_t)2 * LDS_MAP * 4;
    static_assert(C % 16 == 0, "channel tiles of 16");
    static_assert(MTILES == MT * NW, "one group of MT pixel tiles per wave");
    static_assert(NT % NSPLIT == 0, "channel tiles split evenly");
};

template <int C, int H, int W, int MT, int NW, int NSPLIT, bool BACKWARD>
__global__ __launch_bounds__(NW * NSPLIT * 64) void stack_tail_kernel(StackTailArgs a)
{
    using S = StackCfg<C, H, W, MT, NW, NSPLIT>;
    constexpr int NT = S::NTL, KS = S::KS, PLANE = S::PLANE, G = S::G;
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = (tid >> 6) % NW;        // pixel-tile group
    const int n0 = ((tid >> 6) / NW) * NT;   // first channel tile of this wave

    zero_lds<2 * S::LDS_MAP, S::WAVES * 64>(smem, tid);  // guards and halo rows of both maps stay zero
    __syncthreads();

    int pix[MT], lofs[MT];
    float hi_l[MT], hi_r[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int p = (wave * MT + m) * 16 + l15;
        pix[m] = p;
        const int pc = p < S::NPIX ? p : 0;
        lofs[m] = G + pc - 1 + g * PLANE;
        hi_l[m] = (pc % W == 0) ? 0.f : INFINITY;
        hi_r[m] = (pc % W == W - 1) ? 0.f : INFINITY;
    }

    // A layer's A operand (16-byte coalesced loads of the packed weights) is requested when the PREVIOUS layer's K loop
    // ends - its registers are free from then on - so the L2 round trip runs under that layer's epilogue and the barrier:
    // loaded at the top of its own layer, all eight waves of the one resident workgroup sat through it, ~1 us per layer.
    float wa[NT][KS];
    auto load_weights = [&](int layer) {
        const float4 *pw = reinterpret_cast<const float4 *>(a.w[layer]);
#pragma unroll
        for (int s4 = 0; s4 < KS / 4; ++s4)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float4 v = pw[(s4 * S::NT + n0 + n) * 64 + lane];
                wa[n][4 * s4 + 0] = v.x;
                wa[n][4 * s4 + 1] = v.y;
                wa[n][4 * s4 + 2] = v.z;
                wa[n][4 * s4 + 3] = v.w;
            }
    };
    load_weights(0);

    for (int img = blockIdx.x; img < a.n_images; img += gridDim.x) {
        __syncthreads();  // the previous image's last readers of X are done
        stage_band_chunk_dma<C, H, W, S::ROWS, PLANE, G, S::WAVES>(a.in, img, 0, smem, tid);
        const size_t img_off = (size_t)img * C * H * W;

#pragma unroll 1
        for (int layer = 0; layer < 4; ++layer) {
            float bias_r[NT][4];
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) bias_r[n][r] = BACKWARD ? 0.f : a.bias[layer][(n0 + n) * 16 + g * 4 + r];
            // backward: the ReLU gate of this layer's output (its forward pre-activation), requested now and consumed
            // after the K loop
            float gate[BACKWARD ? MT : 1][NT][4];
            if constexpr (BACKWARD) {
                // range-checked buffer reads (common.h): a predicated load is a branch plus a full memory wait each
                const __amdgpu_buffer_rsrc_t mask = buffer_of(a.mask[layer] + img_off);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            gate[m][n][r] = buffer_f32(mask, pix[m] < S::NPIX ? (((n0 + n) * 16 + g * 4 + r) * (H * W) + pix[m]) * 4
                                                                              : kOutside);  // unused beyond the map
            }

            __syncthreads();  // the source map is complete (DMA landed / previous epilogue's LDS writes)
            const int odd = layer & 1;
            const int src = odd ? S::LDS_MAP : 0;  // even layers read X, odd layers read Y
            int base[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) base[m] = lofs[m] + src;

            f32x4 acc[NT][MT];
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

            // K loop: the forward block pipeline of conv3x3.hip (ReLU on read, edge taps masked by the same med3)
            constexpr int SB = (MT * NT >= 4) ? 2 : 4;
            constexpr int NB = (KS + SB - 1) / SB;
            float raw[2][SB][MT];
            auto load_block = [&](int j, float (&r)[SB][MT]) {
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int s = j * SB + u;
                    if (s < KS) {
                        const int tap = s / (C / 4), cs = s % (C / 4);
                        const int tap_off = (tap / 3) * W + (tap % 3);
#pragma unroll
                        for (int m = 0; m < MT; ++m) r[u][m] = smem[base[m] + cs * 4 * PLANE + tap_off];
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
                        const int kx = (s / (C / 4)) % 3;
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            float x = raw[j & 1][u][m];
                            if (kx != 1) {
                                const float hi = kx == 0 ? hi_l[m] : hi_r[m];
                                x = __builtin_amdgcn_fmed3f(x, BACKWARD ? -hi : 0.f, hi);
                            } else if (!BACKWARD) {
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
                            for (int n = 0; n < NT; ++n) acc[n][m] = mfma16(wa[n][s], bv[u][m], acc[n][m]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }

            // ---- epilogue: lane holds pixel l15 x channels g*4..g*4+3 of each tile.  Even layers write Y, odd layers
            // add the block input (X, same positions) and overwrite it: nobody else reads those positions of X
            // during an odd layer, and the next layer's readers wait at its barrier.
            __builtin_amdgcn_sched_barrier(0);
            load_weights((layer + 1) & 3);  // the next layer's (after the fourth: the next image's first) A operand
            float *dst = smem + (odd ? 0 : S::LDS_MAP);
            float *save = a.save[layer];
            // the block input of an odd layer is read for ALL of the lane's elements before the first of them is
            // overwritten (only this lane touches these positions): read-add-write per element was one LDS round
            // trip after another, the compiler cannot move a read above the previous element's write
            __builtin_amdgcn_sched_barrier(0);  // not hoisted into the K loop (register pressure)
            float skip[MT][NT][4];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        skip[m][n][r] = odd ? smem[((n0 + n) * 16 + g * 4 + r) * PLANE + G + W + (pix[m] < S::NPIX ? pix[m] : 0)] : 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (pix[m] < S::NPIX) {
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int co = (n0 + n) * 16 + g * 4 + r;
                            const int lo = co * PLANE + G + W + pix[m];
                            float val = acc[n][m][r] + bias_r[n][r];
                            if constexpr (BACKWARD) val = gate[m][n][r] > 0.f ? val : 0.f;
                            if (odd) val = val + skip[m][n][r];
                            dst[lo] = val;
                            if (save) save[img_off + (size_t)co * (H * W) + pix[m]] = val;
                        }
                }
            }
        }
    }
}

template <int C, int H, int W, int MT, int NW, int NSPLIT, bool BACKWARD>
int launch_stack_tail(const StackTailArgs &args, hipStream_t st)
{
    using S = StackCfg<C, H, W, MT, NW, NSPLIT>;
    auto kern = stack_tail_kernel<C, H, W, MT, NW, NSPLIT, BACKWARD>;
    static int wg_per_cu = 0;
    if (wg_per_cu == 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)S::LDS_BYTES);
        if (e != hipSuccess) return fail(PPO_E_HIP, "stack_tail: hipFuncSetAttribute: %s", hipGetErrorString(e));
        int nb = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, S::WAVES * 64, S::LDS_BYTES);
        if (e != hipSuccess) return fail(PPO_E_HIP, "stack_tail: occupancy query: %s", hipGetErrorString(e));
        wg_per_cu = nb < 1 ? 1 : nb;
    }
    int grid = 256 * wg_per_cu;
    if (grid > args.n_images) grid = args.n_images;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(S::WAVES * 64), S::LDS_BYTES, st, args);
    return check_launch("stack_tail_kernel");
}

// ---------------------------------------------------------------------------------------------------------
// The same four convolutions for the 16-channel stack, whose map (16 x 42 x 42 floats = 113 KB) leaves no room for a
// second one: every layer runs IN PLACE with the map moving up one row per layer.
//
// A 3x3 convolution computed in bands of TR rows, top to bottom, no longer needs input row y - 1 once output row y is
// known, so output row y can take the storage of input row y - 1.  Per band: K loop over input rows y0 - 1 .. y0 + TR
// (accumulators in registers) -> ONE workgroup barrier (every wave has read the rows about to be overwritten) -> the
// band's outputs are written over input rows y0 - 1 .. y0 + TR - 2.  The next band reads rows >= y0 + TR - 1: untouched,
// so it starts without another barrier.  A plane holds H + 6 rows: layer l reads its input at row offset 5 - l and
// leaves its output at 4 - l (four layers, four rows of head-room); the row just below each new map is cleared at the
// layer boundary (it still holds the last row of the previous map).  The skip connection of the odd layers (the block
// input: p or q0; backward: g or g1) is re-read from HBM — the kernel's own input or what layer 1 wrote two layers
// earlier — as 16-byte loads in the transposed epilogue of conv3x3.hip's WIDE path, which this kernel shares, as it
// does the K loop and the arithmetic order: results are bit-identical to the four launches.
//
// What it removes from the four launches: three kernel boundaries with their weight prologues and ragged tails, and
// the HBM -> LDS band staging (with its halo re-reads and the DMA waits at every item barrier) of layers 2-4.
template <int C, int H, int W, int TR>
struct ShiftCfg {
    static_assert(C == 16, "one channel tile per wave");
    static constexpr int NW = 16;                // waves: two groups of GW, working half a step apart
    static constexpr int GW = 8;                 // waves of a group = pixel-tile groups of a band (two per SIMD: a lone
                                                 // wave leaves the MFMA pipe idle while it prepares its next operands)
    static constexpr int G = 4;
    static constexpr int ROWS = H + 6;           // 4 rows of head-room + a halo row above and below
    static constexpr int PLANE_RAW = ROWS * W + 2 * G;
    static constexpr int PLANE = PLANE_RAW + ((16 - PLANE_RAW % 32) + 32) % 32;  // = 16 (mod 32)
    static constexpr int KS = 9 * (C / 4);
    static constexpr int NBANDS = (H + TR - 1) / TR;
    static constexpr int NPIX = TR * W;          // pixels of a band
    static constexpr int MTILES = (NPIX + 15) / 16;       // the last tile of a band may be partial (whole quads)
    static constexpr int MT = MTILES / GW;                // pixel tiles per wave
    static constexpr int LDS_MAP = C * PLANE;
    static constexpr int SCR = MT * 16 * kScrPitchS;      // per-wave transpose scratch (floats)
    // the groups' epilogues alternate (a barrier apart), so wave i of either group uses scratch i
    static constexpr size_t LDS_BYTES = (size_t)(LDS_MAP + GW * SCR) * 4;
    static_assert(H % TR == 0 && MTILES == MT * GW, "whole bands, tiles split evenly over a group's waves");
    static_assert((H * W) % 4 == 0 && NPIX % 4 == 0 && W % 2 == 0, "float4 epilogue on whole quads");
    static_assert(LDS_BYTES <= 160 * 1024, "map + scratch must fit the CU's LDS");
};

// Two wave groups alternate: in every half-step one group runs the K loop of a band (two waves per SIMD) while the
// other runs the epilogue of the band before it (scratch transpose, gate / skip,
// LDS overwrite, HBM stores), then a workgroup barrier.  Group hs % 2 owns band hs.  The rows an epilogue overwrites
// (input rows y0 - 1 .. y0 + TR - 2 of its band) are disjoint from the rows the other group's K loop of the NEXT band
// reads (>= y0 + TR - 1), and were last read two half-steps (two barriers) ago by the K loop of the band before, so one
// barrier per half-step is all the ordering there is.  With every wave in the same phase (the first form of this
// kernel) the MFMA pipe idled through every barrier and epilogue: 9.1 k cycles per band for 4.6 k of MFMA work.
template <int C, int H, int W, int TR, bool BACKWARD>
__global__ __launch_bounds__(1024) void stack_shift_kernel(StackTailArgs a)
{
    using S = ShiftCfg<C, H, W, TR>;
    constexpr int KS = S::KS, PLANE = S::PLANE, G = S::G, MT = S::MT, NW = S::NW;
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4, wave = tid >> 6;
    const int grp = __builtin_amdgcn_readfirstlane(wave / S::GW), wv = wave % S::GW;
    const int ch = lane >> 2, quad = lane & 3;  // transposed epilogue: channel ch, pixels quad*4 .. +3 of a tile
    float *scr = smem + S::LDS_MAP + wv * S::SCR;

    // per-lane constants of this wave's pixel tiles inside a band (identical for every band, layer and image)
    int lofs[MT], p4[MT];
    float hi_l[MT], hi_r[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int p = (wv * MT + m) * 16 + l15;
        lofs[m] = G + p - 1 + g * PLANE;  // window origin (row above, x - 1) relative to the band's first input row
        hi_l[m] = (p % W == 0) ? 0.f : INFINITY;
        hi_r[m] = (p % W == W - 1) ? 0.f : INFINITY;
        p4[m] = (wv * MT + m) * 16 + quad * 4;
    }

    for (int img = blockIdx.x; img < a.n_images; img += gridDim.x) {
        const size_t img_off = (size_t)img * C * H * W;
        __syncthreads();  // the previous image's last readers are done
        // head-room + top halo rows (0 .. 4) and the bottom halo row (H + 5) are zero; rows 5 .. H + 4 take the image
        for (int i = tid; i < C * 6 * W; i += NW * 64) {
            const int c = i / (6 * W), r = (i / W) % 6, x = i % W;
            smem[c * PLANE + G + (r < 5 ? r : H + 5) * W + x] = 0.f;
        }
        for (int i = tid; i < C * 2 * G; i += NW * 64)  // guards in front of and behind the rows
            smem[(i / (2 * G)) * PLANE + ((i % (2 * G)) < G ? (i % G) : G + S::ROWS * W + (i % G))] = 0.f;
        {   // X <- in: a channel's H * W floats are one run (flat layout), 16-byte LDS-DMA requests
            using gptr_t = const __attribute__((address_space(1))) void *;
            using lptr_t = __attribute__((address_space(3))) void *;
            constexpr int N4 = H * W / 4, REQ = (N4 + 63) / 64;
            const int w0 = __builtin_amdgcn_readfirstlane(wave);
            for (int c = w0; c < C; c += NW) {
                const float *g0 = a.in + img_off + (size_t)c * H * W;
                float *l0 = smem + c * PLANE + G + 5 * W;
#pragma unroll
                for (int q = 0; q < REQ; ++q)
                    if (q * 64 + lane < N4)
                        __builtin_amdgcn_global_load_lds((gptr_t)(g0 + (q * 64 + lane) * 4), (lptr_t)(l0 + q * 256), 16, 0, 0);
            }
        }

#pragma unroll 1
        for (int layer = 0; layer < 4; ++layer) {
            float wa[KS];
            const float4 *pw = reinterpret_cast<const float4 *>(a.w[layer]);
#pragma unroll
            for (int s4 = 0; s4 < KS / 4; ++s4) {
                const float4 v = pw[s4 * 64 + lane];
                wa[4 * s4 + 0] = v.x, wa[4 * s4 + 1] = v.y, wa[4 * s4 + 2] = v.z, wa[4 * s4 + 3] = v.w;
            }
            const float bias_c = BACKWARD ? 0.f : a.bias[layer][ch];
            const int odd = layer & 1;
            // skip connection of the odd layers: the block input = the kernel's input (layer 1) or layer 1's output (layer 3)
            const float *res = odd ? (layer == 1 ? a.in : a.save[1]) + img_off : nullptr;
            const float *mask = BACKWARD ? a.mask[layer] + img_off : nullptr;
            float *save = a.save[layer] ? a.save[layer] + img_off : nullptr;
            const __amdgpu_buffer_rsrc_t mask_b = buffer_of(mask, mask != nullptr), res_b = buffer_of(res, res != nullptr);
            const int in_row = 5 - layer;  // plane row of this layer's input row y = 0

            // state a band carries from its K half-step to its epilogue half-step
            f32x4 acc[MT];
            float4 gate4[MT], res4[MT];

            __syncthreads();  // the source map is complete (DMA landed / previous layer's LDS writes and row clear)
#pragma unroll 1
            for (int hs = 0; hs <= S::NBANDS; ++hs) {
                if ((hs & 1) == grp) {
                    if (hs < S::NBANDS) {
                        // ---------------- K half-step of band hs
                        const int y0 = hs * TR;
                        const int chan_off = ch * (H * W) + y0 * W;
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            const bool live = p4[m] < S::NPIX;  // a quad is whole or absent (NPIX % 4 == 0)
                            // range-checked buffer reads (common.h); an absent tensor reads zeros
                            const int off = live ? (chan_off + p4[m]) * 4 : kOutside;
                            if constexpr (BACKWARD) gate4[m] = buffer_f32x4(mask_b, off);
                            res4[m] = buffer_f32x4(res_b, off);
                        }
                        int base[MT];
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            base[m] = lofs[m] + (y0 - 1 + in_row) * W;
                            acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                        // the block pipeline of conv3x3.hip / stack_tail_kernel
                        constexpr int SB = 2;
                        constexpr int NB = (KS + SB - 1) / SB;
                        float raw[2][SB][MT];
                        auto load_block = [&](int j, float (&r)[SB][MT]) {
#pragma unroll
                            for (int u = 0; u < SB; ++u) {
                                const int s = j * SB + u;
                                if (s < KS) {
                                    const int tap = s / (C / 4), cs = s % (C / 4);
                                    const int tap_off = (tap / 3) * W + (tap % 3);
#pragma unroll
                                    for (int m = 0; m < MT; ++m) r[u][m] = smem[base[m] + cs * 4 * PLANE + tap_off];
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
                                    const int kx = (s / (C / 4)) % 3;
#pragma unroll
                                    for (int m = 0; m < MT; ++m) {
                                        float x = raw[j & 1][u][m];
                                        if (kx != 1) {
                                            const float hi = kx == 0 ? hi_l[m] : hi_r[m];
                                            x = __builtin_amdgcn_fmed3f(x, BACKWARD ? -hi : 0.f, hi);
                                        } else if (!BACKWARD) {
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
                                    for (int m = 0; m < MT; ++m) acc[m] = mfma16(wa[s], bv[u][m], acc[m]);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else if (layer < 3) {
                        // last half-step, no band left for this group: the row below the new map still holds the old
                        // map's last row (read for the last time one barrier ago): it is the next layer's bottom halo
                        for (int i = tid % (S::GW * 64); i < C * W; i += S::GW * 64)
                            smem[(i / W) * PLANE + G + (H + in_row - 1) * W + i % W] = 0.f;
                    }
                } else if (hs >= 1) {
                    // ---------------- epilogue half-step of band hs - 1
                    const int y0 = (hs - 1) * TR;
                    const int chan_off = ch * (H * W) + y0 * W;
                    // MFMA layout -> [channel][16 pixels] rows of the wave's scratch (DS operations of a wave are in order)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) scr[(m * 16 + g * 4 + r) * kScrPitchS + l15] = acc[m][r];
                    __builtin_amdgcn_wave_barrier();
                    float *out_row = smem + ch * PLANE + G + (y0 + in_row - 1) * W;  // output row y sits where input row y - 1 sat
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        if (p4[m] >= S::NPIX) continue;
                        const float4 o = *reinterpret_cast<const float4 *>(scr + (m * 16 + ch) * kScrPitchS + quad * 4);
                        const float ov[4] = {o.x, o.y, o.z, o.w};
                        const float gv[4] = {gate4[m].x, gate4[m].y, gate4[m].z, gate4[m].w};
                        const float rv[4] = {res4[m].x, res4[m].y, res4[m].z, res4[m].w};
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float val = ov[e] + bias_c;
                            if constexpr (BACKWARD) val = gv[e] > 0.f ? val : 0.f;
                            v[e] = val + rv[e];
                        }
                        if (layer < 3) {  // the next layer's input (8-byte stores: rows are W = 2 (mod 4) floats apart)
                            *reinterpret_cast<float2 *>(out_row + p4[m]) = make_float2(v[0], v[1]);
                            *reinterpret_cast<float2 *>(out_row + p4[m] + 2) = make_float2(v[2], v[3]);
                        }
                        if (save) *reinterpret_cast<float4 *>(save + chan_off + p4[m]) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                __syncthreads();
            }
        }
    }
}

template <int C, int H, int W, int TR, bool BACKWARD>
int launch_stack_shift(const StackTailArgs &args, hipStream_t st)
{
    using S = ShiftCfg<C, H, W, TR>;
    auto kern = stack_shift_kernel<C, H, W, TR, BACKWARD>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)S::LDS_BYTES);
        if (e != hipSuccess) return fail(PPO_E_HIP, "stack_shift: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    int grid = 256;  // one workgroup per CU (150 KB of LDS each)
    if (grid > args.n_images) grid = args.n_images;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(S::NW * 64), S::LDS_BYTES, st, args);
    return check_launch("stack_shift_kernel");
}

// ---------------------------------------------------------------------------------------------------------
// A whole stack in one launch: first convolution + 3x3/s2 max-pool + the two residual blocks, for the stack whose
// input map fits LDS next to its pre-pool map (32 channels at 21x21 -> 11x11: 2 x 63.5 KB).
//   c = conv(in) + b   (no ReLU on read: rl/impala.py:96-97 feeds the previous stack's output as is)
//   p = maxpool3x3s2(c), idx = argmax tap (ties to the first tap in row-major order, padding excluded, NaN propagates:
//       the rule of conv3x3_pool_kernel / maxpool_fwd_kernel / F.max_pool2d)
//   then the residual blocks on p exactly as stack_tail_kernel (the small maps reuse the input map's LDS).
// resident_conv is the layer body of stack_tail_kernel as a function (that kernel keeps its own copy: it is tuned
// and measured as it stands).
// Compile-time loop: f(integral_constant<I>) for I in [I, N).  (With two convolution geometries in one kernel a
// `#pragma unroll` K loop exceeds the unroller's budget and stays a real loop, with the weights in scratch.)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct LaneMap {  // where this lane's pixel tiles live in a resident map
    template <int MT>
    struct T {
        int pix[MT], lofs[MT];
        float hi_l[MT], hi_r[MT];
    };
};

template <int H, int W, int PLANE, int G, int MT>
__device__ __forceinline__ void lane_map_init(LaneMap::T<MT> &lm, int pixel_wave, int l15, int g)
{
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int p = (pixel_wave * MT + m) * 16 + l15;
        lm.pix[m] = p;
        const int pc = p < H * W ? p : 0;
        lm.lofs[m] = G + pc - 1 + g * PLANE;
        lm.hi_l[m] = (pc % W == 0) ? 0.f : INFINITY;
        lm.hi_r[m] = (pc % W == W - 1) ? 0.f : INFINITY;
    }
}

// dst = conv(f(src)) + bias (+ dst's previous value when `residual`), f = ReLU (RELU_IN) or identity; both maps in LDS at
// smem + src_off / dst_off; `save` (nullable) receives the HBM copy of this image's result.
// GATED (backward-data): the result is zeroed where `mask` (this image's forward pre-activation map) is <= 0, before
// the residual add; bias is then null.
template <int C, int H, int W, int MT, int NTOT, int NT, bool RELU_IN, bool GATED = false>
__device__ __forceinline__ void resident_conv(float *smem, int src_off, int dst_off, bool residual, const float *wpk,
                                              const float *bias, float *save, const LaneMap::T<MT> &lm, int n0, int lane,
                                              const float *mask = nullptr)
{
    constexpr int ROWS = H + 2, G = 4, PLANE_RAW = ROWS * W + 2 * G;
    constexpr int PLANE = PLANE_RAW + ((16 - PLANE_RAW % 32) + 32) % 32;
    constexpr int KS = 9 * (C / 4);
    const int g = lane >> 4;
    float wa[NT][KS];
    const float4 *pw = reinterpret_cast<const float4 *>(wpk);
#pragma unroll
    for (int s4 = 0; s4 < KS / 4; ++s4)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const float4 v = pw[(s4 * NTOT + n0 + n) * 64 + lane];
            wa[n][4 * s4 + 0] = v.x;
            wa[n][4 * s4 + 1] = v.y;
            wa[n][4 * s4 + 2] = v.z;
            wa[n][4 * s4 + 3] = v.w;
        }
    float bias_r[NT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_r[n][r] = (GATED || !bias) ? 0.f : bias[(n0 + n) * 16 + g * 4 + r];
    float gate[GATED ? MT : 1][NT][4];
    if constexpr (GATED) {
        const __amdgpu_buffer_rsrc_t mask_b = buffer_of(mask);  // range-checked reads (common.h)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    gate[m][n][r] = buffer_f32(mask_b, lm.pix[m] < H * W ? (((n0 + n) * 16 + g * 4 + r) * (H * W) + lm.pix[m]) * 4
                                                                         : kOutside);  // unused beyond the map
    }

    __syncthreads();  // the source map is complete
    int base[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) base[m] = lm.lofs[m] + src_off;
    f32x4 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int SB = (MT * NT >= 4) ? 2 : 4;
    constexpr int NB = (KS + SB - 1) / SB;
    float raw[2][SB][MT];
    auto load_block = [&](int j, float (&r)[SB][MT]) {
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int s = j * SB + u;
            if (s < KS) {
                const int tap = s / (C / 4), cs = s % (C / 4);
                const int tap_off = (tap / 3) * W + (tap % 3);
#pragma unroll
                for (int m = 0; m < MT; ++m) r[u][m] = smem[base[m] + cs * 4 * PLANE + tap_off];
            }
        }
    };
    load_block(0, raw[0]);
    static_for<0, NB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        float bv[SB][MT];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int s = j * SB + u;
            if (s < KS) {
                const int kx = (s / (C / 4)) % 3;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    float x = raw[j & 1][u][m];
                    if (kx != 1) {
                        const float hi = kx == 0 ? lm.hi_l[m] : lm.hi_r[m];
                        x = __builtin_amdgcn_fmed3f(x, RELU_IN ? 0.f : -hi, hi);
                    } else if (RELU_IN) {
                        x = relu1(x);
                    }
                    bv[u][m] = x;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (j + 1 < NB) load_block(j + 1, raw[(j + 1) & 1]);
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int s = j * SB + u;
            if (s < KS) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[n][m] = mfma16(wa[n][s], bv[u][m], acc[n][m]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (lm.pix[m] < H * W) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = (n0 + n) * 16 + g * 4 + r;
                    const int lo = dst_off + co * PLANE + G + W + lm.pix[m];
                    float val = acc[n][m][r] + bias_r[n][r];
                    if constexpr (GATED) val = gate[m][n][r] > 0.f ? val : 0.f;
                    if (residual) val = val + smem[lo];
                    smem[lo] = val;
                    if (save) save[(size_t)co * (H * W) + lm.pix[m]] = val;
                }
        }
    }
}

struct StackFullArgs {
    const float *in;       // [n, C, HI, WI] the previous stack's output
    const float *w[5];     // packed forward weights: firstconv, block0.conv0, block0.conv1, block1.conv0, block1.conv1
    const float *bias[5];
    float *pooled;         // [n, C, HO, WO] p (nullable: inference)
    uint8_t *argmax;       // [n, C, HO, WO] winning tap 0..8 (nullable)
    float *save[4];        // a0, q0, a1, q1 on [n, C, HO, WO]; q1 required
    // chained form: `in` is the PREVIOUS stack's pooled map and its two residual blocks run first on the resident
    // [C, HI, WI] map (as stack_tail_kernel), so that stack's output never leaves LDS on its way into this one
    const float *pre_w[4];
    const float *pre_bias[4];
    float *pre_save[4];    // the previous stack's a0, q0, a1, q1 (nullable: inference)
    int has_pre;
    int n_images;
};

template <int C, int HI, int WI, int MTI, int HO, int WO, int MTO, int NW, int NSPLIT>
__global__ __launch_bounds__(NW * NSPLIT * 64) void stack_full_kernel(StackFullArgs a)
{
    using SI = StackCfg<C, HI, WI, MTI, NW, NSPLIT>;
    using SO = StackCfg<C, HO, WO, MTO, NW, NSPLIT>;
    static_assert(HO == (HI + 1) / 2 && WO == (WI + 1) / 2, "3x3 / stride 2 / pad 1 pooling");
    static_assert(2 * SO::LDS_MAP <= SI::LDS_MAP, "the small maps reuse the input map's LDS");
    constexpr int NT = SI::NTL, WAVES = SI::WAVES, THREADS = WAVES * 64;
    constexpr int A_OFF = 0, B_OFF = SI::LDS_MAP, X_OFF = 0, Y_OFF = SO::LDS_MAP;
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = tid >> 6, pixel_wave = wave % NW, n0 = (wave / NW) * NT;

    LaneMap::T<MTI> lmi;
    LaneMap::T<MTO> lmo;
    lane_map_init<HI, WI, SI::PLANE, SI::G, MTI>(lmi, pixel_wave, l15, g);
    lane_map_init<HO, WO, SO::PLANE, SO::G, MTO>(lmo, pixel_wave, l15, g);

    if (a.has_pre) zero_lds<SI::LDS_MAP, THREADS>(smem + B_OFF, tid);  // halo rows / guards of the blocks' second map

    for (int img = blockIdx.x; img < a.n_images; img += gridDim.x) {
        __syncthreads();  // the previous image's last readers of the small maps are done
        zero_lds<SI::LDS_MAP, THREADS>(smem + A_OFF, tid);  // guards of the input map (its halo rows: the stage below)
        __syncthreads();
        stage_band_chunk_dma<C, HI, WI, SI::ROWS, SI::PLANE, SI::G, WAVES>(a.in, img, 0, smem + A_OFF, tid);
        if (a.has_pre) {
            // ---- the previous stack's residual blocks: A <-> B ping-pong, result back in A
            const size_t big_img = (size_t)img * C * HI * WI;
#pragma unroll 1
            for (int layer = 0; layer < 4; ++layer) {
                const int odd = layer & 1;
                float *save = a.pre_save[layer] ? a.pre_save[layer] + big_img : nullptr;
                resident_conv<C, HI, WI, MTI, SI::NT, NT, true>(smem, odd ? B_OFF : A_OFF, odd ? A_OFF : B_OFF, odd != 0,
                                                                a.pre_w[layer], a.pre_bias[layer], save, lmi, n0, lane);
            }
        }
        // ---- first convolution: A -> B (no ReLU on read, no residual, not saved: only its pooled form leaves the CU)
        resident_conv<C, HI, WI, MTI, SI::NT, NT, false>(smem, A_OFF, B_OFF, false, a.w[0], a.bias[0], nullptr, lmi, n0, lane);
        __syncthreads();  // B is complete and nobody reads A any more
        zero_lds<2 * SO::LDS_MAP, THREADS>(smem + X_OFF, tid);  // X and Y (inside A's region): halo rows and guards
        __syncthreads();
        // ---- pool B -> X (+ HBM copy and argmax); a wave pass covers RPW pooled rows of WO outputs
        {
            constexpr int RPW = 64 / WO;
            const int sub = lane / WO, xo = lane % WO;
            const size_t out_img = (size_t)img * C * HO * WO;
            for (int u0 = wave * RPW; u0 < C * HO; u0 += WAVES * RPW) {
                const int u = u0 + sub;
                const int co = u / HO, yo = u % HO;
                if (sub < RPW && co < C) {
                    const float *src = smem + B_OFF + co * SI::PLANE + SI::G + WI + (2 * yo - 1) * WI + 2 * xo - 1;
                    float best;
                    int best_tap;
                    pool_window_lds<HI % 2 == 0, WI % 2 == 0>(src, WI, yo > 0, 2 * yo + 1 < HI, xo > 0, 2 * xo + 1 < WI, best, best_tap);
                    smem[X_OFF + co * SO::PLANE + SO::G + WO + yo * WO + xo] = best;
                    const size_t oi = out_img + ((size_t)co * HO + yo) * WO + xo;
                    if (a.pooled) a.pooled[oi] = best;
                    if (a.argmax) a.argmax[oi] = (uint8_t)best_tap;
                }
            }
        }
        // ---- the residual blocks on the pooled map (each layer starts with the barrier that publishes its source)
        const size_t out_img = (size_t)img * C * HO * WO;
#pragma unroll 1
        for (int layer = 0; layer < 4; ++layer) {
            const int odd = layer & 1;
            float *save = a.save[layer] ? a.save[layer] + out_img : nullptr;
            resident_conv<C, HO, WO, MTO, SO::NT, NT, true>(smem, odd ? Y_OFF : X_OFF, odd ? X_OFF : Y_OFF, odd != 0,
                                                            a.w[1 + layer], a.bias[1 + layer], save, lmo, n0, lane);
        }
    }
}

// Backward-data of the same stack in one launch: the blocks' gated transposed chain on the pooled map (as
// stack_tail_kernel<.., BACKWARD>), max-pool backward into the pre-pool gradient dc, and the first convolution's
// transposed pass, which yields the gradient of the previous stack's output:
//   da1, g1, da0, g0 as ppo_impala_stack_tail_backward_f32;  dc = maxpool_bwd(g0, argmax)  (summation order of
//   maxpool_bwd_kernel: windows (oy0, k), (oy0, k + 1), (oy0 + 1, k), (oy0 + 1, k + 1));  g_prev = conv_first^T(dc).
// All of them go to HBM as well: the weight-gradient kernels read them.
struct StackFullBwdArgs {
    const float *g;         // [n, C, HO, WO] d loss / d (stack output)
    const float *w[5];      // backward-data packed weights: block1.conv1, block1.conv0, block0.conv1, block0.conv0, firstconv
    const float *mask[4];   // a1, q0, a0, p  ([n, C, HO, WO])
    const uint8_t *argmax;  // [n, C, HO, WO]
    float *save[4];         // da1, g1, da0, g0
    float *dc;              // [n, C, HI, WI]
    float *g_prev;          // [n, C, HI, WI]
    int n_images;
};

template <int C, int HI, int WI, int MTI, int HO, int WO, int MTO, int NW, int NSPLIT>
__global__ __launch_bounds__(NW * NSPLIT * 64) void stack_full_bwd_kernel(StackFullBwdArgs a)
{
    using SI = StackCfg<C, HI, WI, MTI, NW, NSPLIT>;
    using SO = StackCfg<C, HO, WO, MTO, NW, NSPLIT>;
    static_assert(2 * SO::LDS_MAP <= SI::LDS_MAP, "the small maps live inside the first big map's LDS");
    constexpr int NT = SI::NTL, WAVES = SI::WAVES, THREADS = WAVES * 64;
    constexpr int A_OFF = 0, B_OFF = SI::LDS_MAP, X_OFF = 0, Y_OFF = SO::LDS_MAP;
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = tid >> 6, pixel_wave = wave % NW, n0 = (wave / NW) * NT;

    LaneMap::T<MTI> lmi;
    LaneMap::T<MTO> lmo;
    lane_map_init<HI, WI, SI::PLANE, SI::G, MTI>(lmi, pixel_wave, l15, g);
    lane_map_init<HO, WO, SO::PLANE, SO::G, MTO>(lmo, pixel_wave, l15, g);

    zero_lds<SI::LDS_MAP, THREADS>(smem + B_OFF, tid);  // dc's halo rows and guards: only its interior is ever written

    for (int img = blockIdx.x; img < a.n_images; img += gridDim.x) {
        __syncthreads();  // the previous image's transposed first convolution has read B and written A
        zero_lds<2 * SO::LDS_MAP, THREADS>(smem + X_OFF, tid);  // X, Y: halo rows and guards (A's interior overlaps them)
        __syncthreads();
        stage_band_chunk_dma<C, HO, WO, SO::ROWS, SO::PLANE, SO::G, WAVES>(a.g, img, 0, smem + X_OFF, tid);
        const size_t small_img = (size_t)img * C * HO * WO, big_img = (size_t)img * C * HI * WI;
#pragma unroll 1
        for (int layer = 0; layer < 4; ++layer) {
            const int odd = layer & 1;
            resident_conv<C, HO, WO, MTO, SO::NT, NT, false, true>(smem, odd ? Y_OFF : X_OFF, odd ? X_OFF : Y_OFF, odd != 0,
                                                                   a.w[layer], nullptr, a.save[layer] + small_img, lmo, n0,
                                                                   lane, a.mask[layer] + small_img);
        }
        __syncthreads();  // g0 is complete in X
        // ---- max-pool backward: gather per pre-pool pixel, X (g0) + argmax -> B (dc) and HBM
        {
            const uint8_t *am = a.argmax + small_img;
            for (int e = tid; e < C * HI * WI; e += THREADS) {
                const int co = e / (HI * WI), r = e % (HI * WI);
                const int iy = r / WI, ix = r % WI;
                const int k = ix >> 1, oy0 = iy >> 1;
                const int ky0 = (iy & 1) ? 2 : 1;
                const bool has_oy1 = (iy & 1) && (oy0 + 1 < HO);
                const bool odd_x = ix & 1;
                const bool has_ox1 = odd_x && (k + 1 < WO);
                const int kx0 = odd_x ? 2 : 1;  // this pixel's tap column inside window ox = k
                const float *gx = smem + X_OFF + co * SO::PLANE + SO::G + WO;
                const uint8_t *ac = am + co * (HO * WO);
                float sum = 0.f;
                {
                    const int i00 = oy0 * WO + k;
                    if (ac[i00] == ky0 * 3 + kx0) sum += gx[i00];
                    if (has_ox1 && ac[i00 + 1] == ky0 * 3 + 0) sum += gx[i00 + 1];
                }
                if (has_oy1) {
                    const int i10 = (oy0 + 1) * WO + k;
                    if (ac[i10] == kx0) sum += gx[i10];
                    if (has_ox1 && ac[i10 + 1] == 0) sum += gx[i10 + 1];
                }
                smem[B_OFF + co * SI::PLANE + SI::G + WI + r] = sum;
                a.dc[big_img + e] = sum;
            }
        }
        // ---- first convolution, transposed: B (dc) -> A (overwrites the small maps; its barrier publishes B)
        resident_conv<C, HI, WI, MTI, SI::NT, NT, false, false>(smem, B_OFF, A_OFF, false, a.w[4], nullptr,
                                                                a.g_prev + big_img, lmi, n0, lane);
    }
}

template <int C, int HI, int WI, int MTI, int HO, int WO, int MTO, int NW, int NSPLIT>
int launch_stack_full_bwd(const StackFullBwdArgs &args, hipStream_t st)
{
    using SI = StackCfg<C, HI, WI, MTI, NW, NSPLIT>;
    constexpr size_t kLds = 2 * (size_t)SI::LDS_MAP * 4;
    auto kern = stack_full_bwd_kernel<C, HI, WI, MTI, HO, WO, MTO, NW, NSPLIT>;
    static bool ready = false;
    if (!ready) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)kLds);
        if (e != hipSuccess) return fail(PPO_E_HIP, "stack_full_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        ready = true;
    }
    const int grid = args.n_images < 256 ? args.n_images : 256;  // one workgroup per CU (LDS)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SI::WAVES * 64), kLds, st, args);
    return check_launch("stack_full_bwd_kernel");
}

template <int C, int HI, int WI, int MTI, int HO, int WO, int MTO, int NW, int NSPLIT>
int launch_stack_full(const StackFullArgs &args, hipStream_t st)
{
    using SI = StackCfg<C, HI, WI, MTI, NW, NSPLIT>;
    constexpr size_t kLds = 2 * (size_t)SI::LDS_MAP * 4;
    auto kern = stack_full_kernel<C, HI, WI, MTI, HO, WO, MTO, NW, NSPLIT>;
    static bool ready = false;
    if (!ready) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)kLds);
        if (e != hipSuccess) return fail(PPO_E_HIP, "stack_full: hipFuncSetAttribute: %s", hipGetErrorString(e));
        ready = true;
    }
    const int grid = args.n_images < 256 ? args.n_images : 256;  // one workgroup per CU (LDS)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SI::WAVES * 64), kLds, st, args);
    return check_launch("stack_full_kernel");
}


// ---------------------------------------------------------------------------------------------------------
// The chained launch for SMALL inference batches (a rollout group of 128 images on a 256-CU chip): one image on TWO CUs.
//
// One workgroup per image leaves half the chip idle at 128 images, and the chain of nine convolutions of one image is
// MFMA-bound on its one CU (rocprofv3 timeline of the rollout: 2 x 124 us of a 536 us env step in this launch, at 0.31 of
// peak).  Rows cannot be split: the halo of nine chained convolutions and a stride-2 pool is most of a 21-row map.  So the
// pair of workgroups of an image splits the OUTPUT CHANNELS of the five convolutions on the 21x21 map (82 % of the
// launch's FLOPs): workgroup h computes channels 16 h .. 16 h + 15 of every layer for all pixels - the same K loop, K
// order and epilogue arithmetic per output element as stack_full_kernel, hence the same bits - and after each of those
// layers the two exchange their halves through HBM / L2: own half written next to the LDS copy (resident_conv's `save`),
// a release fence, a flag; a bounded spin on the partner's flag (acquire), then the partner's half comes in by LDS-DMA.
// The 11x11 part (pool + four convolutions, 18 % of the FLOPs, 1 us of MFMA work per layer) is cheaper to compute twice
// than to exchange: both workgroups run it on the whole pooled map, channel tile by channel tile, and h = 0 stores it.
//
// Pairing and progress: workgroups b and b ^ 8 form a pair (same XCD under the round-robin placement, adjacent in that
// XCD's dispatch order), so the resident set of an in-order dispatch is always whole pairs plus at most one workgroup
// per XCD whose partner is next in line; whole pairs finish without anybody else, which frees the slot.  The spin is
// BOUNDED all the same: a partner that never shows up sets ctl[2] (the host checks it) and the wave goes on with
// garbage instead of hanging the device.  ctl[0] is the launch number (flag values never repeat: no resets), advanced
// by the last workgroup of a launch to leave (ctl[1] counts them), so every launch has the same arguments and can sit in a
// recorded launch plan.
struct ChainSplitArgs {
    const float *in;         // [n, C, HI, WI] the previous stack's pooled map
    const float *pre_w[4];   // packed forward weights of its two residual blocks
    const float *pre_bias[4];
    const float *w[5];       // firstconv + the four block convolutions of this stack
    const float *bias[5];
    float *out;              // q1 [n, C, HO, WO]
    float *xbuf;             // [n][2][C][HI * WI] exchange slots
    uint32_t *flags;         // [n][2]
    uint32_t *ctl;           // launch number, workgroups done, error, fault injection (tests: see the kernel)
    int n_images;
    int skip;                // timing aid, PPO_TUNE_TIMING_AIDS builds only (PPO_AMD_SPLIT_SKIP): 1 no waiting for the
                             // partner, 2 no 11x11 part - garbage out; always 0 in the shipped library
};

constexpr int kSplitWaves = 8;  // 28 pixel tiles of the 21x21 map, four per wave (the eighth wave's are past the map); one
                                // channel tile (16 of 32) per workgroup.  (14 waves of two tiles spilled: 128-register cap)
constexpr uint32_t kSpinLimit = 1u << 21;

template <int C, int HI, int WI, int HO, int WO>
__global__ __launch_bounds__(kSplitWaves * 64) void stack_chain_split_kernel(ChainSplitArgs a)
{
    static_assert(C == 32, "two channel tiles: one per workgroup of a pair");
    using SI = StackCfg<C, HI, WI, 7, 4, 2>;   // geometry constants only (PLANE, LDS_MAP, ...)
    using SO = StackCfg<C, HO, WO, 2, 4, 2>;
    constexpr int MTI = 4, MTO = 2, WAVES = kSplitWaves, THREADS = WAVES * 64;
    static_assert(SI::MTILES <= MTI * WAVES && SO::MTILES == MTO * 4 && WAVES == 8, "every pixel tile has a wave");
    static_assert(2 * SO::LDS_MAP <= SI::LDS_MAP, "the small maps reuse the input map's LDS");
    constexpr int A_OFF = 0, B_OFF = SI::LDS_MAP, X_OFF = 0, Y_OFF = SO::LDS_MAP;
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = tid >> 6;
    const int b = blockIdx.x;
    const int h = (b >> 3) & 1;                      // which half of the output channels
    const int pair0 = (b & 7) | ((b >> 4) << 3);     // partner: b ^ 8
    const int n_pairs = (gridDim.x >> 4) << 3;

    LaneMap::T<MTI> lmi;
    LaneMap::T<MTO> lmo;
    lane_map_init<HI, WI, SI::PLANE, SI::G, MTI>(lmi, wave, l15, g);
    lane_map_init<HO, WO, SO::PLANE, SO::G, MTO>(lmo, wave & 3, l15, g);

    const uint32_t base = __hip_atomic_load(a.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 8u;  // 5 exchanges per launch
    // ctl[3] != 0 (set by tests only, DualHeadNet.chain_split_inject_fault): the second workgroup of every pair withholds
    // its flags and the spin gives up early - the path a partner that never arrives takes, driven on purpose
    const uint32_t fault = __hip_atomic_load(a.ctl + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t spin_limit = fault ? 1u << 10 : kSpinLimit;
    bool gave_up = false;  // thread 0's: after one timeout this workgroup no longer waits (its results are garbage anyway)
    zero_lds<SI::LDS_MAP, THREADS>(smem + B_OFF, tid);  // halo rows / guards of the second big map

    for (int img = pair0; img < a.n_images; img += n_pairs) {
        uint32_t *flag_own = a.flags + 2 * img + h, *flag_other = a.flags + 2 * img + (1 - h);
        float *slots = a.xbuf + (size_t)img * 2 * C * HI * WI;
        uint32_t step = 0;
        // own half of a layer's result is in LDS and in its slot: publish it, wait for the partner's, fetch that
        auto exchange = [&](int dst_off, float *slot, bool fetch) {
            ++step;
            // Both workgroups of a pair sit on one XCD, i.e. behind one L2, and the vector L1 is write-through: a slot
            // store is visible to the partner once it has been acknowledged (vmcnt(0)), PROVIDED the partner's reads do
            // not hit its own L1 - they carry sc1.  (A device-scope fence pair here - buffer_wbl2 / buffer_inv, the
            // whole L2 written back ten times per image - made the launch 4 x slower than the one it replaces.)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                if (!(fault && h == 1))
                    __hip_atomic_store(flag_own, base + step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint32_t it = 0;
                while (!(a.skip & 1) && !gave_up &&
                       __hip_atomic_load(flag_other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < base + step) {
                    // the partner never came: say so and go on (garbage, but the device lives).  Somebody else's timeout
                    // (looked at every 1024 polls) ends this wait too: the launch's results are void already, so the whole
                    // launch costs one spin limit, not one per exchange and image
                    if (++it > spin_limit ||
                        ((it & 1023u) == 0 && __hip_atomic_load(a.ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                        __hip_atomic_store(a.ctl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        gave_up = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            __syncthreads();
            if (fetch)
                stage_band_chunk_dma<16, HI, WI, SI::ROWS, SI::PLANE, SI::G, WAVES, 16>(slot + (size_t)(1 - h) * 16 * HI * WI, 0, 0,
                                                                                       smem + dst_off + (1 - h) * 16 * SI::PLANE, tid);
        };

        __syncthreads();  // the previous image's last readers of the small maps are done
        zero_lds<SI::LDS_MAP, THREADS>(smem + A_OFF, tid);
        __syncthreads();
        stage_band_chunk_dma<C, HI, WI, SI::ROWS, SI::PLANE, SI::G, WAVES>(a.in, img, 0, smem + A_OFF, tid);
        // ---- the previous stack's residual blocks: A <-> B ping-pong, result back in A
#pragma unroll 1
        for (int layer = 0; layer < 4; ++layer) {
            const int odd = layer & 1;
            float *slot = slots + (size_t)(layer & 1) * C * HI * WI;
            resident_conv<C, HI, WI, MTI, 2, 1, true>(smem, odd ? B_OFF : A_OFF, odd ? A_OFF : B_OFF, odd != 0, a.pre_w[layer],
                                                      a.pre_bias[layer], slot, lmi, h, lane);
            exchange(odd ? A_OFF : B_OFF, slot, true);
        }
        // ---- first convolution of this stack: A -> B (no ReLU on read, no residual)
        resident_conv<C, HI, WI, MTI, 2, 1, false>(smem, A_OFF, B_OFF, false, a.w[0], a.bias[0], slots, lmi, h, lane);
        exchange(B_OFF, slots, h == 0);
        if (h != 0 || (a.skip & 2)) continue;  // the 11x11 part is workgroup 0's alone: its partner is done with this image (its CU is
                               // free for another stream's kernels); the fetch of the last exchange is all it waited for
        __syncthreads();  // B is complete (the fetch landed: vmcnt(0) at the barrier) and nobody reads A any more
        zero_lds<2 * SO::LDS_MAP, THREADS>(smem + X_OFF, tid);
        __syncthreads();
        // ---- pool B -> X
        {
            constexpr int RPW = 64 / WO;
            const int sub = lane / WO, xo = lane % WO;
            for (int u0 = wave * RPW; u0 < C * HO; u0 += WAVES * RPW) {
                const int u = u0 + sub;
                const int co = u / HO, yo = u % HO;
                if (sub < RPW && co < C) {
                    const float *src = smem + B_OFF + co * SI::PLANE + SI::G + WI + (2 * yo - 1) * WI + 2 * xo - 1;
                    float best;
                    int best_tap;
                    pool_window_lds<HI % 2 == 0, WI % 2 == 0>(src, WI, yo > 0, 2 * yo + 1 < HI, xo > 0, 2 * xo + 1 < WI, best, best_tap);
                    smem[X_OFF + co * SO::PLANE + SO::G + WO + yo * WO + xo] = best;
                }
            }
        }
        // ---- the residual blocks on the pooled map (4 pixel-tile waves x 2 channel tiles, as stack_full_kernel)
        float *out_img = a.out + (size_t)img * C * HO * WO;
#pragma unroll 1
        for (int layer = 0; layer < 4; ++layer) {
            const int odd = layer & 1;
            resident_conv<C, HO, WO, MTO, 2, 1, true>(smem, odd ? Y_OFF : X_OFF, odd ? X_OFF : Y_OFF, odd != 0, a.w[1 + layer],
                                                      a.bias[1 + layer], layer == 3 ? out_img : nullptr, lmo, wave >> 2, lane);
        }
    }
    // the last workgroup to leave advances the launch number
    __syncthreads();
    if (tid == 0) {
        const uint32_t done = __hip_atomic_fetch_add(a.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            __hip_atomic_store(a.ctl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(a.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int C, int HI, int WI, int HO, int WO>
int launch_chain_split(const ChainSplitArgs &args, hipStream_t st)
{
    using SI = StackCfg<C, HI, WI, 7, 4, 2>;
    constexpr size_t kLds = 2 * (size_t)SI::LDS_MAP * 4;
    auto kern = stack_chain_split_kernel<C, HI, WI, HO, WO>;
    static bool ready = false;
    if (!ready) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)kLds);
        if (e != hipSuccess) return fail(PPO_E_HIP, "stack_chain_split: hipFuncSetAttribute: %s", hipGetErrorString(e));
        ready = true;
    }
    int groups = (args.n_images + 7) / 8;       // 8 pairs = 16 workgroups per group: b and b ^ 8
    if (groups > 16) groups = 16;               // at most one workgroup per CU; more images: pairs walk them
    hipLaunchKernelGGL(kern, dim3(16 * groups), dim3(kSplitWaves * 64), kLds, st, args);
    return check_launch("stack_chain_split_kernel");
}

}  // namespace
}  // namespace ppo

extern "C" int ppo_impala_stack_tail_supported(int channels, int h, int w)
{
    // 84x84 observations: 21x21 and 11x11; 64x64 (procgen): 16x16 and 8x8
    if (channels == 16) return (h == 42 && w == 42) || (h == 32 && w == 32);  // in-place shifted form (stack_shift_kernel)
    return channels == 32 && ((h == 11 && w == 11) || (h == 21 && w == 21) || (h == 16 && w == 16) || (h == 8 && w == 8));
}

extern "C" int ppo_impala_stack_tail_forward_f32(const float *in, const float *const *packed_weights,
                                                 const float *const *biases, float *a0, float *q0, float *a1, float *q1,
                                                 int n_images, int channels, int h, int w, void *stream)
{
    using namespace ppo;
    if (n_images < 0) return fail(PPO_E_INVALID, "ppo_impala_stack_tail_forward_f32: negative batch");
    if (n_images == 0) return PPO_OK;
    if (!in || !packed_weights || !biases || !q1)
        return fail(PPO_E_INVALID, "ppo_impala_stack_tail_forward_f32: null pointer");
    StackTailArgs args;
    args.in = in;
    for (int l = 0; l < 4; ++l) {
        if (!packed_weights[l] || !biases[l])
            return fail(PPO_E_INVALID, "ppo_impala_stack_tail_forward_f32: null weights / bias of layer %d", l);
        if (!aligned(packed_weights[l], 16))
            return fail(PPO_E_ALIGN, "ppo_impala_stack_tail_forward_f32: packed weights must be 16-byte aligned");
        args.w[l] = packed_weights[l];
        args.bias[l] = biases[l];
        args.mask[l] = nullptr;
    }
    if (!aligned(in, 4)) return fail(PPO_E_ALIGN, "ppo_impala_stack_tail_forward_f32: input must be 4-byte aligned");
    args.save[0] = a0;
    args.save[1] = q0;
    args.save[2] = a1;
    args.save[3] = q1;
    args.n_images = n_images;
    hipStream_t st = as_stream(stream);
    if (channels == 16) {
        // the skip connection of the second block is re-read from q0, so q0 is required even when nothing else is kept
        if (!q0) return fail(PPO_E_INVALID, "ppo_impala_stack_tail_forward_f32: the 16-channel form needs q0 (its second block re-reads it)");
        if (!aligned(in, 16) || !aligned(q0, 16) || !aligned(q1, 16) || (a0 && !aligned(a0, 16)) || (a1 && !aligned(a1, 16)))
            return fail(PPO_E_ALIGN, "ppo_impala_stack_tail_forward_f32: the 16-channel form needs 16-byte aligned maps");
        if (h == 42 && w == 42) return launch_stack_shift<16, 42, 42, 6, false>(args, st);
        if (h == 32 && w == 32) return launch_stack_shift<16, 32, 32, 8, false>(args, st);
    }
    if (channels == 32 && h == 11 && w == 11) return launch_stack_tail<32, 11, 11, 2, 4, PPO_TAIL_SPLIT, false>(args, st);
    if (channels == 32 && h == 21 && w == 21) return launch_stack_tail<32, 21, 21, 7, 4, PPO_TAIL_SPLIT, false>(args, st);
    if (channels == 32 && h == 16 && w == 16) return launch_stack_tail<32, 16, 16, 4, 4, PPO_TAIL_SPLIT, false>(args, st);
    if (channels == 32 && h == 8 && w == 8) return launch_stack_tail<32, 8, 8, 1, 4, PPO_TAIL_SPLIT, false>(args, st);
    return fail(PPO_E_INVALID, "ppo_impala_stack_tail_forward_f32: no kernel for %d channels at %dx%d", channels, h, w);
}

extern "C" int ppo_impala_stack_tail_backward_f32(const float *g, const float *const *packed_weights_t,
                                                  const float *const *masks, float *da1, float *g1, float *da0, float *g0,
                                                  int n_images, int channels, int h, int w, void *stream)
{
    using namespace ppo;
    if (n_images < 0) return fail(PPO_E_INVALID, "ppo_impala_stack_tail_backward_f32: negative batch");
    if (n_images == 0) return PPO_OK;
    if (!g || !packed_weights_t || !masks || !da1 || !g1 || !da0 || !g0)
        return fail(PPO_E_INVALID, "ppo_impala_stack_tail_backward_f32: null pointer");
    StackTailArgs args;
    args.in = g;
    for (int l = 0; l < 4; ++l) {
        if (!packed_weights_t[l] || !masks[l])
            return fail(PPO_E_INVALID, "ppo_impala_stack_tail_backward_f32: null weights / mask of layer %d", l);
        if (!aligned(packed_weights_t[l], 16))
            return fail(PPO_E_ALIGN, "ppo_impala_stack_tail_backward_f32: packed weights must be 16-byte aligned");
        args.w[l] = packed_weights_t[l];
        args.bias[l] = nullptr;
        args.mask[l] = masks[l];
    }
    args.save[0] = da1;
    args.save[1] = g1;
    args.save[2] = da0;
    args.save[3] = g0;
    args.n_images = n_images;
    hipStream_t st = as_stream(stream);
    if (channels == 16) {
        for (const void *ptr : {(const void *)g, (const void *)da1, (const void *)g1, (const void *)da0, (const void *)g0,
                                (const void *)masks[0], (const void *)masks[1], (const void *)masks[2], (const void *)masks[3]})
            if (!aligned(ptr, 16))
                return fail(PPO_E_ALIGN, "ppo_impala_stack_tail_backward_f32: the 16-channel form needs 16-byte aligned maps");
        if (h == 42 && w == 42) return launch_stack_shift<16, 42, 42, 6, true>(args, st);
        if (h == 32 && w == 32) return launch_stack_shift<16, 32, 32, 8, true>(args, st);
    }
    if (channels == 32 && h == 11 && w == 11) return launch_stack_tail<32, 11, 11, 2, 4, PPO_TAIL_SPLIT, true>(args, st);
    if (channels == 32 && h == 21 && w == 21) return launch_stack_tail<32, 21, 21, 7, 4, PPO_TAIL_SPLIT, true>(args, st);
    if (channels == 32 && h == 16 && w == 16) return launch_stack_tail<32, 16, 16, 4, 4, PPO_TAIL_SPLIT, true>(args, st);
    if (channels == 32 && h == 8 && w == 8) return launch_stack_tail<32, 8, 8, 1, 4, PPO_TAIL_SPLIT, true>(args, st);
    return fail(PPO_E_INVALID, "ppo_impala_stack_tail_backward_f32: no kernel for %d channels at %dx%d", channels, h, w);
}

extern "C" int ppo_impala_stack_full_supported(int channels, int h, int w)
{
    return channels == 32 && ((h == 21 && w == 21) || (h == 16 && w == 16));
}

extern "C" int ppo_impala_stack_full_forward_f32(const float *in, const float *const *packed_weights,
                                                 const float *const *biases, float *pooled, uint8_t *argmax, float *a0,
                                                 float *q0, float *a1, float *q1, int n_images, int channels, int h, int w,
                                                 void *stream)
{
    using namespace ppo;
    if (n_images < 0) return fail(PPO_E_INVALID, "ppo_impala_stack_full_forward_f32: negative batch");
    if (n_images == 0) return PPO_OK;
    if (!in || !packed_weights || !biases || !q1)
        return fail(PPO_E_INVALID, "ppo_impala_stack_full_forward_f32: null pointer");
    StackFullArgs args;
    args.in = in;
    for (int l = 0; l < 5; ++l) {
        if (!packed_weights[l] || !biases[l])
            return fail(PPO_E_INVALID, "ppo_impala_stack_full_forward_f32: null weights / bias of layer %d", l);
        if (!aligned(packed_weights[l], 16))
            return fail(PPO_E_ALIGN, "ppo_impala_stack_full_forward_f32: packed weights must be 16-byte aligned");
        args.w[l] = packed_weights[l];
        args.bias[l] = biases[l];
    }
    args.pooled = pooled;
    args.argmax = argmax;
    args.save[0] = a0;
    args.save[1] = q0;
    args.save[2] = a1;
    args.save[3] = q1;
    args.has_pre = 0;
    for (int l = 0; l < 4; ++l) {
        args.pre_w[l] = nullptr;
        args.pre_bias[l] = nullptr;
        args.pre_save[l] = nullptr;
    }
    args.n_images = n_images;
    if (channels == 32 && h == 21 && w == 21)
        return launch_stack_full<32, 21, 21, 7, 11, 11, 2, 4, 2>(args, as_stream(stream));
    if (channels == 32 && h == 16 && w == 16)
        return launch_stack_full<32, 16, 16, 4, 8, 8, 1, 4, 2>(args, as_stream(stream));
    return fail(PPO_E_INVALID, "ppo_impala_stack_full_forward_f32: no kernel for %d channels at %dx%d", channels, h, w);
}

extern "C" int ppo_impala_stack_chain_forward_f32(const float *in, const float *const *pre_packed_weights,
                                                  const float *const *pre_biases, float *pre_a0, float *pre_q0,
                                                  float *pre_a1, float *pre_q1, const float *const *packed_weights,
                                                  const float *const *biases, float *pooled, uint8_t *argmax, float *a0,
                                                  float *q0, float *a1, float *q1, int n_images, int channels, int h, int w,
                                                  void *stream)
{
    using namespace ppo;
    if (n_images < 0) return fail(PPO_E_INVALID, "ppo_impala_stack_chain_forward_f32: negative batch");
    if (n_images == 0) return PPO_OK;
    if (!in || !pre_packed_weights || !pre_biases || !packed_weights || !biases || !q1)
        return fail(PPO_E_INVALID, "ppo_impala_stack_chain_forward_f32: null pointer");
    StackFullArgs args;
    args.in = in;
    for (int l = 0; l < 5; ++l) {
        if (!packed_weights[l] || !biases[l] || (l < 4 && (!pre_packed_weights[l] || !pre_biases[l])))
            return fail(PPO_E_INVALID, "ppo_impala_stack_chain_forward_f32: null weights / bias of layer %d", l);
        if (!aligned(packed_weights[l], 16) || (l < 4 && !aligned(pre_packed_weights[l], 16)))
            return fail(PPO_E_ALIGN, "ppo_impala_stack_chain_forward_f32: packed weights must be 16-byte aligned");
        args.w[l] = packed_weights[l];
        args.bias[l] = biases[l];
        if (l < 4) {
            args.pre_w[l] = pre_packed_weights[l];
            args.pre_bias[l] = pre_biases[l];
        }
    }
    args.pre_save[0] = pre_a0;
    args.pre_save[1] = pre_q0;
    args.pre_save[2] = pre_a1;
    args.pre_save[3] = pre_q1;
    args.has_pre = 1;
    args.pooled = pooled;
    args.argmax = argmax;
    args.save[0] = a0;
    args.save[1] = q0;
    args.save[2] = a1;
    args.save[3] = q1;
    args.n_images = n_images;
    if (channels == 32 && h == 21 && w == 21)
        return launch_stack_full<32, 21, 21, 7, 11, 11, 2, 4, 2>(args, as_stream(stream));
    if (channels == 32 && h == 16 && w == 16)
        return launch_stack_full<32, 16, 16, 4, 8, 8, 1, 4, 2>(args, as_stream(stream));
    return fail(PPO_E_INVALID, "ppo_impala_stack_chain_forward_f32: no kernel for %d channels at %dx%d", channels, h, w);
}

extern "C" int ppo_impala_stack_full_backward_f32(const float *g, const float *const *packed_weights_t,
                                                  const float *const *masks, const uint8_t *argmax, float *da1, float *g1,
                                                  float *da0, float *g0, float *dc, float *g_prev, int n_images,
                                                  int channels, int h, int w, void *stream)
{
    using namespace ppo;
    if (n_images < 0) return fail(PPO_E_INVALID, "ppo_impala_stack_full_backward_f32: negative batch");
    if (n_images == 0) return PPO_OK;
    if (!g || !packed_weights_t || !masks || !argmax || !da1 || !g1 || !da0 || !g0 || !dc || !g_prev)
        return fail(PPO_E_INVALID, "ppo_impala_stack_full_backward_f32: null pointer");
    StackFullBwdArgs args;
    args.g = g;
    for (int l = 0; l < 5; ++l) {
        if (!packed_weights_t[l] || (l < 4 && !masks[l]))
            return fail(PPO_E_INVALID, "ppo_impala_stack_full_backward_f32: null weights / mask of layer %d", l);
        if (!aligned(packed_weights_t[l], 16))
            return fail(PPO_E_ALIGN, "ppo_impala_stack_full_backward_f32: packed weights must be 16-byte aligned");
        args.w[l] = packed_weights_t[l];
        if (l < 4) args.mask[l] = masks[l];
    }
    args.argmax = argmax;
    args.save[0] = da1;
    args.save[1] = g1;
    args.save[2] = da0;
    args.save[3] = g0;
    args.dc = dc;
    args.g_prev = g_prev;
    args.n_images = n_images;
    if (channels == 32 && h == 21 && w == 21)
        return launch_stack_full_bwd<32, 21, 21, 7, 11, 11, 2, 4, 2>(args, as_stream(stream));
    if (channels == 32 && h == 16 && w == 16)
        return launch_stack_full_bwd<32, 16, 16, 4, 8, 8, 1, 4, 2>(args, as_stream(stream));
    return fail(PPO_E_INVALID, "ppo_impala_stack_full_backward_f32: no kernel for %d channels at %dx%d", channels, h, w);
}

extern "C" size_t ppo_impala_stack_chain_split_workspace_bytes(int n_images, int channels, int h, int w)
{
    // exchange slots [n][2][C][h*w] floats, flags [n][2], control words [4] (all 16-byte aligned inside the block)
    const size_t slots = (size_t)n_images * 2 * channels * h * w * sizeof(float);
    const size_t flags = ((size_t)n_images * 2 * sizeof(uint32_t) + 15) & ~(size_t)15;
    return slots + flags + 16;
}

extern "C" int ppo_impala_stack_chain_split_forward_f32(const float *in, const float *const *pre_packed_weights,
                                                        const float *const *pre_biases,
                                                        const float *const *packed_weights, const float *const *biases,
                                                        float *q1, void *workspace, size_t workspace_bytes, int n_images,
                                                        int channels, int h, int w, void *stream)
{
    using namespace ppo;
    if (n_images < 0) return fail(PPO_E_INVALID, "ppo_impala_stack_chain_split_forward_f32: negative batch");
    if (n_images == 0) return PPO_OK;
    if (!in || !pre_packed_weights || !pre_biases || !packed_weights || !biases || !q1 || !workspace)
        return fail(PPO_E_INVALID, "ppo_impala_stack_chain_split_forward_f32: null pointer");
    if (!(channels == 32 && h == 21 && w == 21))
        return fail(PPO_E_INVALID, "ppo_impala_stack_chain_split_forward_f32: no kernel for %d channels at %dx%d", channels, h, w);
    if (workspace_bytes < ppo_impala_stack_chain_split_workspace_bytes(n_images, channels, h, w) || !aligned(workspace, 16))
        return fail(PPO_E_INVALID, "ppo_impala_stack_chain_split_forward_f32: workspace too small or misaligned");
    ChainSplitArgs args;
    args.in = in;
    for (int l = 0; l < 5; ++l) {
        if (!packed_weights[l] || !biases[l] || (l < 4 && (!pre_packed_weights[l] || !pre_biases[l])))
            return fail(PPO_E_INVALID, "ppo_impala_stack_chain_split_forward_f32: null weights / bias of layer %d", l);
        if (!aligned(packed_weights[l], 16) || (l < 4 && !aligned(pre_packed_weights[l], 16)))
            return fail(PPO_E_ALIGN, "ppo_impala_stack_chain_split_forward_f32: packed weights must be 16-byte aligned");
        args.w[l] = packed_weights[l];
        args.bias[l] = biases[l];
        if (l < 4) {
            args.pre_w[l] = pre_packed_weights[l];
            args.pre_bias[l] = pre_biases[l];
        }
    }
    args.out = q1;
    const size_t slots = (size_t)n_images * 2 * channels * h * w * sizeof(float);
    const size_t flags = ((size_t)n_images * 2 * sizeof(uint32_t) + 15) & ~(size_t)15;
    args.xbuf = static_cast<float *>(workspace);
    args.flags = reinterpret_cast<uint32_t *>(static_cast<char *>(workspace) + slots);
    args.ctl = reinterpret_cast<uint32_t *>(static_cast<char *>(workspace) + slots + flags);
    args.n_images = n_images;
#ifdef PPO_TUNE_TIMING_AIDS  // tools/build_variant.sh builds only: the shipped library has no switch that voids results
    static const int skip = getenv("PPO_AMD_SPLIT_SKIP") ? atoi(getenv("PPO_AMD_SPLIT_SKIP")) : 0;
    args.skip = skip;
#else
    args.skip = 0;
#endif
    return launch_chain_split<32, 21, 21, 11, 11>(args, as_stream(stream));
}

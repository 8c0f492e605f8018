// One IMPALA residual block (rl/impala.py:66-84: q' = q + conv1(relu(conv0(relu(q))))) as ONE launch for small
// inference batches, band by band, the intermediate map never leaving LDS (gfx950).
//
// Why: a rollout group is 128 images.  There the 16-channel stack runs as four conv3x3_kernel launches of 20 us each
// with 6.6 us of MFMA work in them (rocprofv3 timeline): at 7 bands per image a workgroup sees one or two
// items of ~1 us, and the launch is its fixed cost - ramp, weight prologue, first band's HBM round trip, tail.  The
// in-place whole-stack kernel (stack_shift_kernel) is MFMA-bound on the one CU an image gets and leaves half the chip
// idle at 128 images.  Fusing the two convolutions of a block halves the launches and keeps every CU busy: per
// (image, band of TR rows) item
//   X <- rows y0-2 .. y0+TR+1 of q (LDS-DMA; rows outside the image are zero)
//   Y  = conv0(relu(X)) + b0 on rows y0-1 .. y0+TR (one halo row each side recomputed: (TR+2)/TR of its work),
//        rows outside the image forced to 0 - they are conv1's zero padding, not a convolution of padding
//   out rows y0 .. y0+TR-1 = conv1(relu(Y)) + b1 + X
// Same implicit GEMM, K order (tap-major, channel groups inner), packed A operand and epilogue arithmetic per output
// element as conv3x3_kernel, so the result is bit-identical to the two launches.
#include "common.h"
#include "conv_stage.h"
#include "mfma.h"
#include <type_traits>

#ifndef PPO_TUNE_BLOCK_HALF_MAX
#define PPO_TUNE_BLOCK_HALF_MAX 256  // largest batch run as half-image items
#endif
#ifndef PPO_TUNE_BLOCK
#define PPO_TUNE_BLOCK 0  // 1 / 2 / 3: timing aids (wrong results), tools/build_variant.sh
#endif

namespace ppo {
namespace {

constexpr int pad_plane(int raw) { return raw + ((16 - raw % 32) + 32) % 32; }  // = 16 (mod 32)

template <int C, int H, int W, int TR, int NW_>
struct BlockCfg {
    static_assert(C == 16, "one channel tile");
    static constexpr int NW = NW_;
    static constexpr int G = 4;
    static constexpr int ROWS = TR + 4;  // input rows of an item
    static constexpr int PLANE_X = pad_plane(ROWS * W + 2 * G), PLANE_Y = pad_plane((TR + 2) * W + 2 * G);
    static constexpr int KS = 9 * (C / 4);
    static constexpr int NBANDS = (H + TR - 1) / TR;
    static constexpr int NPIX0 = (TR + 2) * W, NPIX1 = TR * W;
    static constexpr int NT0 = (NPIX0 + 15) / 16, NT1 = (NPIX1 + 15) / 16;  // pixel tiles; tile t belongs to wave t % NW
    static constexpr int MT0 = (NT0 + NW - 1) / NW, MT1 = (NT1 + NW - 1) / NW;
    static constexpr int LDS_X = C * PLANE_X, LDS_Y = C * PLANE_Y;
    static constexpr size_t LDS_BYTES = (size_t)(LDS_X + LDS_Y) * 4;
    static constexpr int WAVES_PER_SIMD = 160 * 1024 / (int)LDS_BYTES >= 2 ? NW / 2 : NW / 4;  // for the register budget
};

// acc[m] += sum_s wa[s] x f(src[window of pixel tile m at step s]); f = ReLU, edge taps masked (conv3x3.hip's pipeline)
template <int C, int W, int PLANE, int MT>
__device__ __forceinline__ void band_k_loop(const float *__restrict__ src, const int (&base)[MT], const float (&hi_l)[MT],
                                            const float (&hi_r)[MT], const float (&wa)[9 * (C / 4)], f32x4 (&acc)[MT])
{
    constexpr int KS = 9 * (C / 4);
    constexpr int SB = 2;  // four waves per SIMD hide the LDS latency; deeper blocks would spill at 128 VGPRs
#if PPO_TUNE_BLOCK == 1 || PPO_TUNE_BLOCK >= 4  // timing aid: no K loop (4: nor staging, 5: nor stores, 6: neither)
    constexpr int NB = 1;
#else
    constexpr int NB = (KS + SB - 1) / SB;
#endif
    float raw[2][SB][MT];
    auto load_block = [&](int j, float (&r)[SB][MT]) {
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int s = j * SB + u;
            if (s < KS) {
                const int tap = s / (C / 4), cs = s % (C / 4);
                const int tap_off = (tap / 3) * W + (tap % 3);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#if PPO_TUNE_BLOCK == 2  // timing aid: no LDS reads
                    r[u][m] = hi_l[m];
#else
                    r[u][m] = src[base[m] + cs * 4 * PLANE + tap_off];
#endif
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
#if PPO_TUNE_BLOCK == 3  // timing aid: no ReLU / edge mask
                    if (false) {
#else
                    if (kx != 1) {
#endif
                        const float hi = kx == 0 ? hi_l[m] : hi_r[m];
                        x = __builtin_amdgcn_fmed3f(x, 0.f, hi);
                    } else {
#if PPO_TUNE_BLOCK != 3
                        x = relu1(x);
#endif
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
                    acc[m] = mfma16(wa[s], bv[u][m], acc[m]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// One phase of an item for a wave that owns MTV pixel tiles (tiles wave, wave + NW, ...): PHASE 0 = conv0 into Y,
// PHASE 1 = conv1 + skip connection into HBM.
template <class S, int C, int H, int W, int TR, int PHASE, int MTV>
__device__ __forceinline__ void block_phase(float *__restrict__ smem, const float (&wa)[S::KS], const float (&bias)[4],
                                            float *__restrict__ out_img, int y0, int wave, int l15, int g)
{
    constexpr int G = S::G, NW = S::NW;
    constexpr int NPIX = PHASE == 0 ? S::NPIX0 : S::NPIX1;
    constexpr int PLANE = PHASE == 0 ? S::PLANE_X : S::PLANE_Y;
    constexpr int MAP = PHASE == 0 ? 0 : S::LDS_X;
    int base[MTV];
    float hl[MTV], hr[MTV];
    f32x4 acc[MTV];
#pragma unroll
    for (int m = 0; m < MTV; ++m) {
        const int p = (m * NW + wave) * 16 + l15;
        const int pc = p < NPIX ? p : 0;
        base[m] = MAP + G + pc - 1 + g * PLANE;  // output row r reads the source map's rows r .. r + 2
        hl[m] = (pc % W == 0) ? 0.f : INFINITY;
        hr[m] = (pc % W == W - 1) ? 0.f : INFINITY;
        acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    band_k_loop<C, W, PLANE, MTV>(smem, base, hl, hr, wa, acc);
    // the epilogue's pixel indices are recomputed from an opaque copy of the lane id: kept from above they would stay live
    // across the K loop, MTV registers the 16-wave form does not have
    int l15e = l15, pix[MTV];
    asm volatile("" : "+v"(l15e));
#pragma unroll
    for (int m = 0; m < MTV; ++m) pix[m] = (m * NW + wave) * 16 + l15e;
    if constexpr (PHASE == 0) {
        float *sY = smem + S::LDS_X;
#pragma unroll
        for (int m = 0; m < MTV; ++m) {
            if (pix[m] < NPIX) {
                const int yy = y0 - 1 + pix[m] / W;  // Y row r' = image row y0 - 1 + r'
                const bool inside = yy >= 0 && yy < H;
#pragma unroll
                for (int r = 0; r < 4; ++r) sY[(g * 4 + r) * S::PLANE_Y + G + pix[m]] = inside ? acc[m][r] + bias[r] : 0.f;
            }
        }
    } else {
        const int plim = min(NPIX, (H - y0) * W);
#pragma unroll
        for (int m = 0; m < MTV; ++m) {
            if (pix[m] < plim) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = g * 4 + r;
                    float val = acc[m][r] + bias[r];
                    val = val + smem[co * S::PLANE_X + G + 2 * W + pix[m]];  // X row r + 2 = image row y0 + r
#if PPO_TUNE_BLOCK >= 5
                    if (val == 12345.678f)
#endif
                    out_img[(size_t)co * (H * W) + pix[m]] = val;
                }
            }
        }
    }
}

template <int C, int H, int W, int TR, int NW>
__global__ __launch_bounds__((NW * 64), 4) void conv3x3_block_kernel(
    const float *__restrict__ in, const float *__restrict__ w0, const float *__restrict__ b0, const float *__restrict__ w1,
    const float *__restrict__ b1, float *__restrict__ out, int n_images)
{
    using S = BlockCfg<C, H, W, TR, NW>;
    constexpr int KS = S::KS, G = S::G, MT0 = S::MT0, MT1 = S::MT1;
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // A operands in the packed per-lane order (ppo_conv3x3_pack_weights_f32), fetched per phase: a workgroup sees one or
    // two items, and holding both layers' 72 registers would cost half the waves per SIMD
    float wa[KS];
    auto load_weights = [&](const float *w) {
        asm volatile("" : "+s"(w));  // opaque per call: hoisted out of the item loop both sets would stay live
        const float4 *p = reinterpret_cast<const float4 *>(w);
#pragma unroll
        for (int s4 = 0; s4 < KS / 4; ++s4) {
            const float4 u = p[s4 * 64 + lane];
            wa[4 * s4 + 0] = u.x, wa[4 * s4 + 1] = u.y, wa[4 * s4 + 2] = u.z, wa[4 * s4 + 3] = u.w;
        }
    };
    float bias0[4], bias1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias0[r] = b0[g * 4 + r], bias1[r] = b1[g * 4 + r];
    // this wave's share of the pixel tiles: all MT of them, or one fewer when the tile count is no multiple of NW
    const int nv0 = (S::NT0 - wave + NW - 1) / NW, nv1 = (S::NT1 - wave + NW - 1) / NW;

    zero_lds<S::LDS_X + S::LDS_Y, NW * 64>(smem, tid);  // guards stay zero
    const int n_items = n_images * S::NBANDS;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int img = item / S::NBANDS;
        const int y0 = (item % S::NBANDS) * TR;
        __syncthreads();  // zeroing done / the previous item's readers of X and Y are done
        // X rows 0 .. TR+3 <- image rows y0-2 .. y0+TR+1 (the stager's first band row is its y0 argument - 1)
        int stid = tid;
        asm volatile("" : "+v"(stid));  // the stager's per-lane offsets are recomputed per item, not kept across the K loops
#if PPO_TUNE_BLOCK != 4 && PPO_TUNE_BLOCK != 6
        stage_band_chunk_dma<C, H, W, S::ROWS, S::PLANE_X, G, NW>(in, img, y0 - 1, smem, stid);
#endif
        load_weights(w0);
        __syncthreads();  // landed
        if (nv0 == MT0) block_phase<S, C, H, W, TR, 0, MT0>(smem, wa, bias0, nullptr, y0, wave, l15, g);
        else if constexpr (MT0 > 1) block_phase<S, C, H, W, TR, 0, MT0 - 1>(smem, wa, bias0, nullptr, y0, wave, l15, g);
        load_weights(w1);
        __syncthreads();  // Y complete
        float *o = out + (size_t)img * C * H * W + (size_t)y0 * W;
        if (nv1 == MT1) block_phase<S, C, H, W, TR, 1, MT1>(smem, wa, bias1, o, y0, wave, l15, g);
        else if constexpr (MT1 > 1) block_phase<S, C, H, W, TR, 1, MT1 - 1>(smem, wa, bias1, o, y0, wave, l15, g);
    }
}

template <int C, int H, int W, int TR, int NW>
int launch_block(const float *in, const float *w0, const float *b0, const float *w1, const float *b1, float *out, int n,
                 hipStream_t st)
{
    using S = BlockCfg<C, H, W, TR, NW>;
    auto kern = conv3x3_block_kernel<C, H, W, TR, NW>;
    static int wg_per_cu = 0;
    if (wg_per_cu == 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)S::LDS_BYTES);
        if (e != hipSuccess) return fail(PPO_E_HIP, "conv3x3_block: hipFuncSetAttribute: %s", hipGetErrorString(e));
        int nb = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NW * 64, S::LDS_BYTES);
        if (e != hipSuccess) return fail(PPO_E_HIP, "conv3x3_block: occupancy query: %s", hipGetErrorString(e));
        wg_per_cu = nb < 1 ? 1 : (nb > 3 ? 3 : nb);
    }
    const int n_items = n * S::NBANDS;
    int grid = 256 * wg_per_cu;
    if (grid > n_items) grid = n_items;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), S::LDS_BYTES, st, in, w0, b0, w1, b1, out, n);
    return check_launch("conv3x3_block_kernel");
}

}  // namespace
}  // namespace ppo

extern "C" int ppo_conv3x3_block_supported(int channels, int h, int w)
{
    return channels == 16 && ((h == 42 && w == 42) || (h == 32 && w == 32));
}

extern "C" int ppo_conv3x3_block_forward_packed_f32(const float *in, const float *packed0, const float *bias0,
                                                    const float *packed1, const float *bias1, float *out, int n,
                                                    int channels, int h, int w, void *stream)
{
    using namespace ppo;
    if (n < 0) return fail(PPO_E_INVALID, "ppo_conv3x3_block_forward_packed_f32: n < 0");
    if (n == 0) return PPO_OK;
    if (!in || !packed0 || !bias0 || !packed1 || !bias1 || !out)
        return fail(PPO_E_INVALID, "ppo_conv3x3_block_forward_packed_f32: null pointer");
    if (!aligned(packed0, 16) || !aligned(packed1, 16))
        return fail(PPO_E_ALIGN, "ppo_conv3x3_block_forward_packed_f32: packed weights must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    // at most two half-image items per CU: half images on 16 waves (1.1 x the MFMA work of the two convolutions: conv0's
    // two halo rows and the last tile); more images: bands of TR rows on 8 waves, two workgroups per CU (1.3 x, but the
    // next item's staging runs under the other workgroup's K loops)
    if (channels == 16 && h == 42 && w == 42)
        return n <= PPO_TUNE_BLOCK_HALF_MAX ? launch_block<16, 42, 42, 21, 16>(in, packed0, bias0, packed1, bias1, out, n, st)
                                            : launch_block<16, 42, 42, 6, 8>(in, packed0, bias0, packed1, bias1, out, n, st);
    if (channels == 16 && h == 32 && w == 32)
        return n <= PPO_TUNE_BLOCK_HALF_MAX ? launch_block<16, 32, 32, 16, 16>(in, packed0, bias0, packed1, bias1, out, n, st)
                                            : launch_block<16, 32, 32, 8, 8>(in, packed0, bias0, packed1, bias1, out, n, st);
    return fail(PPO_E_INVALID, "ppo_conv3x3_block_forward_packed_f32: no kernel for %d channels at %dx%d", channels, h, w);
}

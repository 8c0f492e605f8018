// 3x3 / stride 1 / pad 1 convolution as an implicit GEMM on the f32 MFMA
// (v_mfma_f32_16x16x4_f32), NCHW float32 — the IMPALA-CNN contraction
// (reference: rl/impala.py:61-62,96 via torch.nn.Conv2d).
//
// One kernel serves forward and backward-data:
//   forward        out[n,o,y,x] = b[o] + sum_{i,ky,kx} f(in[n,i,y+ky-1,x+kx-1]) * w[o,i,ky,kx]  (+ residual)
//   backward-data  dx[n,i,y,x]  = (sum_{o,ky,kx} dy[n,o,y-ky+1,x-kx+1] * w[o,i,ky,kx]) * [pre[n,i,y,x] > 0] (+ dres)
// The second is the first with the weight tensor read transposed and flipped.
// f() is the input transform fused into the LDS staging: identity, ReLU (the
// pre-activation residual blocks of rl/impala.py:73-78 store pre-activations
// and apply ReLU on load), or uint8 -> x/255 (rl/models.py:842-848).
//
// Mapping: a 256-thread workgroup owns a band of TR output rows of one image
// for all output channels.  The input band (+1-pixel halo, zero padded) sits in
// LDS as planar [ci][row][col]; the weights sit in LDS as [k/2][co][2] with
// k = tap*CINP + ci.  GEMM view: M = output channel (MFMA "i"), N = pixel
// (MFMA "j"), K = 9*CIN.  A wave holds MT pixel tiles x NT channel tiles of
// 16x16 accumulators; every operand read is a conflict-free ds_read_b32 at a
// compile-time offset from a per-lane base (plane stride = 16 mod 32 banks).
#include "common.h"
#include "conv_stage.h"
#include "mfma.h"

namespace ppo {
namespace {

template <int CIN, int COUT, int H, int W, int TR>
struct ConvCfg {
    static constexpr int CINP = (CIN + 3) / 4 * 4;  // k-steps of 4 channels
    static constexpr int COUTP = (COUT + 15) / 16 * 16;
    static constexpr int NT = COUTP / 16;
    static constexpr int PW = W + 2;
    static constexpr int ROWS = TR + 2;
    static constexpr int PLANE_RAW = ROWS * PW;
    static constexpr int PLANE = PLANE_RAW + ((16 - PLANE_RAW % 32) + 32) % 32;  // = 16 (mod 32)
    static constexpr int K = 9 * CINP;
    static constexpr int NBANDS = (H + TR - 1) / TR;
    static constexpr int NPIX = TR * W;
    static constexpr int MTILES = (NPIX + 15) / 16;
    static constexpr int LDS_IN = CINP * PLANE;  // floats
    static constexpr int LDS_W = K * COUTP;      // floats
    static constexpr size_t LDS_BYTES = (size_t)(LDS_IN + LDS_W) * 4;
};

template <int CIN, int COUT, int H, int W, int TR, int MT, int IN_MODE, bool TRANSPOSED>
__global__ __launch_bounds__(256, 3) void conv3x3_kernel(
    const void *__restrict__ in_, const float *__restrict__ w, const float *__restrict__ bias,
    const float *__restrict__ residual, const float *__restrict__ mask_src, float *__restrict__ out,
    int n_images)
{
    using C = ConvCfg<CIN, COUT, H, W, TR>;
    extern __shared__ __align__(16) float smem[];
    float *s_in = smem;
    float *s_w = smem + C::LDS_IN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l15 = lane & 15;
    const int g = lane >> 4;

    // ---- weights -> LDS, once per workgroup: s_w[((k >> 1) * COUTP + co) * 2 + (k & 1)]
    // (8 independent loads in flight per thread, then the LDS stores: see conv_stage.h)
    for (int base = 0; base < C::K * C::COUTP; base += 256 * 8) {
        float v[8];
        int off[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 256 + tid;
            const int k = idx / C::COUTP;
            const int co = idx % C::COUTP;
            const int tap = k / C::CINP;
            const int ci = k % C::CINP;
            off[u] = idx < C::K * C::COUTP ? ((k >> 1) * C::COUTP + co) * 2 + (k & 1) : -1;
            v[u] = 0.f;
            if (idx < C::K * C::COUTP && ci < CIN && co < COUT) {
                v[u] = TRANSPOSED ? w[((size_t)ci * COUT + co) * 9 + (8 - tap)]  // w[o=ci][i=co], taps flipped
                                  : w[((size_t)co * CIN + ci) * 9 + tap];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (off[u] >= 0) s_w[off[u]] = v[u];
    }

    const int n_items = n_images * C::NBANDS;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int img = item / C::NBANDS;
        const int band = item % C::NBANDS;
        const int y0 = band * TR;

        __syncthreads();  // previous item's readers are done with s_in (and s_w is complete)
        // ---- input band (+halo) -> LDS with the fused input transform
        stage_band<CIN, C::CINP, H, W, C::ROWS, C::PW, C::PLANE, 1, IN_MODE, 256>(in_, img, y0, s_in, tid);
        __syncthreads();

        // ---- MFMA main loop: MT pixel tiles per step of the wave
        constexpr int GROUPS = (C::MTILES + MT - 1) / MT;
        constexpr int TAP_UNROLL = C::CINP >= 16 ? 1 : 9;
        for (int grp = wave; grp < GROUPS; grp += 4) {
            int pix[MT];
            int base[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int p = (grp * MT + m) * 16 + l15;
                pix[m] = p;
                const int pc = p < C::NPIX ? p : 0;
                base[m] = (pc / W) * C::PW + (pc % W) + g * C::PLANE;
            }
            f32x4 acc[C::NT][MT];
#pragma unroll
            for (int n = 0; n < C::NT; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

            // big-CIN layers: keep the tap loop rolled so the compiler cannot hoist all
            // 9*CIN/4 steps of LDS reads at once (it spills otherwise)
#pragma unroll TAP_UNROLL
            for (int tap = 0; tap < 9; ++tap) {
                const int tap_off = (tap / 3) * C::PW + (tap % 3);
#pragma unroll
                for (int cs = 0; cs < C::CINP / 4; ++cs) {
                    const int k0 = tap * C::CINP + cs * 4;  // this lane group contracts k = k0 + g
                    float a[C::NT], b[MT];
#pragma unroll
                    for (int n = 0; n < C::NT; ++n)
                        a[n] = s_w[(((k0 >> 1) + (g >> 1)) * C::COUTP + n * 16 + l15) * 2 + (g & 1)];
#pragma unroll
                    for (int m = 0; m < MT; ++m) b[m] = s_in[base[m] + cs * 4 * C::PLANE + tap_off];
#pragma unroll
                    for (int n = 0; n < C::NT; ++n)
#pragma unroll
                        for (int m = 0; m < MT; ++m) acc[n][m] = mfma16(a[n], b[m], acc[n][m]);
                }
            }

            // ---- epilogue: lane holds pixel (lane & 15) x channels g*4 .. g*4+3 of each tile
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int p = pix[m];
                const int y = y0 + p / W;
                const int x = p % W;
                if (p < C::NPIX && y < H) {
#pragma unroll
                    for (int n = 0; n < C::NT; ++n) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int co = n * 16 + g * 4 + r;
                            if (co < COUT) {
                                const size_t oi = (((size_t)img * COUT + co) * H + y) * W + x;
                                float val = acc[n][m][r];
                                if (bias) val += bias[co];
                                if (mask_src) val = mask_src[oi] > 0.f ? val : 0.f;
                                if (residual) val += residual[oi];
                                out[oi] = val;
                            }
                        }
                    }
                }
            }
        }
    }
}

template <int CIN, int COUT, int H, int W, int TR, int MT, int IN_MODE, bool TRANSPOSED>
int launch_conv(const void *in, const float *w, const float *bias, const float *residual,
                const float *mask_src, float *out, int n_images, hipStream_t st)
{
    using C = ConvCfg<CIN, COUT, H, W, TR>;
    auto kern = conv3x3_kernel<CIN, COUT, H, W, TR, MT, IN_MODE, TRANSPOSED>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        if (e != hipSuccess) return fail(PPO_E_HIP, "conv3x3: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int n_items = n_images * C::NBANDS;
    const int wg_per_cu = (int)((160 * 1024) / C::LDS_BYTES) > 4 ? 4 : (int)((160 * 1024) / C::LDS_BYTES);
    int grid = 256 * (wg_per_cu < 1 ? 1 : wg_per_cu);
    if (grid > n_items) grid = n_items;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), C::LDS_BYTES, st, in, w, bias, residual, mask_src, out,
                       n_images);
    return check_launch("conv3x3_kernel");
}

// Supported layer geometries: Atari 84x84 (rl/atari.py) and Procgen 64x64 (rl/procgen.py)
// through the three IMPALA stacks (16, 32, 32 channels; rl/models.py:873).
template <int IN_MODE, bool TRANSPOSED>
int dispatch_conv(int cin, int cout, int h, int w_, const void *in, const float *w, const float *bias,
                  const float *residual, const float *mask_src, float *out, int n, hipStream_t st)
{
// FIRST: the obs conv (uint8 or float obs, never ReLU-on-load, never transposed);
// UP: the channel-changing stack-first conv; SAME: everything else.
#define PPO_CONV_CASE(ALLOWED, CI, CO, HH, WW, TR, MT)                                             \
    if constexpr (ALLOWED) {                                                                       \
        if (cin == CI && cout == CO && h == HH && w_ == WW)                                        \
            return launch_conv<CI, CO, HH, WW, TR, MT, IN_MODE, TRANSPOSED>(in, w, bias, residual, \
                                                                             mask_src, out, n, st); \
    }
    constexpr bool FIRST = !TRANSPOSED && IN_MODE != IN_RELU;
    constexpr bool UP = !TRANSPOSED && IN_MODE == IN_NONE;
    constexpr bool DOWN = TRANSPOSED;
    constexpr bool SAME = IN_MODE != IN_U8;
    PPO_CONV_CASE(FIRST, 4, 16, 84, 84, 12, 2)
    PPO_CONV_CASE(FIRST, 5, 16, 84, 84, 12, 2)
    PPO_CONV_CASE(FIRST, 3, 16, 64, 64, 16, 2)
    PPO_CONV_CASE(FIRST, 4, 16, 64, 64, 16, 2)
    PPO_CONV_CASE(UP, 16, 32, 42, 42, 14, 2)
    PPO_CONV_CASE(UP, 16, 32, 32, 32, 16, 2)
    PPO_CONV_CASE(DOWN, 32, 16, 42, 42, 14, 2)
    PPO_CONV_CASE(DOWN, 32, 16, 32, 32, 16, 2)
    PPO_CONV_CASE(SAME, 16, 16, 42, 42, 14, 2)
    PPO_CONV_CASE(SAME, 16, 16, 32, 32, 16, 2)
    PPO_CONV_CASE(SAME, 32, 32, 21, 21, 11, 2)
    PPO_CONV_CASE(SAME, 32, 32, 16, 16, 16, 2)
    PPO_CONV_CASE(SAME, 32, 32, 11, 11, 11, 2)
    PPO_CONV_CASE(SAME, 32, 32, 8, 8, 8, 2)
#undef PPO_CONV_CASE
    return fail(PPO_E_INVALID, "conv3x3: unsupported geometry cin=%d cout=%d h=%d w=%d transposed=%d", cin,
                cout, h, w_, (int)TRANSPOSED);
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

// Weight/bias gradient of the 3x3 convolution on the f32 MFMA (gfx950).
//
//   dW[o,i,ky,kx] = sum_{n,y,x} dy[n,o,y,x] * f(in[n,i,y+ky-1,x+kx-1])      db[o] = sum_{n,y,x} dy[n,o,y,x]
//
// (what autograd computes for torch.nn.Conv2d at rl/impala.py:61-62,96; f is the
// same fused load transform as the forward kernel: identity / ReLU / uint8/255).
//
// GEMM view: M = output channel o (MFMA "i", operand A = dy), N = j = tap*CINP + i
// plus one all-ones column whose result is db (MFMA "j", operand B = shifted input),
// K = pixels.  A 256-thread workgroup walks (image, row band) items; both bands sit
// in LDS (planar, plane stride = 2 mod 32 banks so that 16 channels x 2 adjacent
// pixels are conflict-free); the four waves split the N tiles and keep their
// accumulators in registers across all items of the workgroup.  Each workgroup
// then writes one partial [COUT][JP] slab; conv3x3_wgrad_reduce sums the slabs in
// a fixed order (deterministic, no atomics) into PyTorch's [o][i][3][3] layout.
#include "common.h"
#include "conv_stage.h"
#include "mfma.h"

namespace ppo {
namespace {

constexpr int pad_mod32(int v, int target) { return v + ((target - v % 32) + 32) % 32; }

template <int CIN, int COUT, int H, int W, int TR>
struct WgradCfg {
    static constexpr int CINP = (CIN + 3) / 4 * 4;
    static constexpr int MTC = COUT / 16;
    static constexpr int NJ = 9 * CINP + 1;  // + the ones column (bias gradient)
    static constexpr int NTT = (NJ + 15) / 16;
    static constexpr int JP = NTT * 16;
    static constexpr int NTW = (NTT + 3) / 4;  // n-tiles per wave (max)
    static constexpr int PWD = (W + 3) / 4 * 4;
    static constexpr int PWX = PWD + 2;
    static constexpr int ROWS = TR + 2;
    static constexpr int XPLANE = pad_mod32(ROWS * PWX, 2);
    static constexpr int DPLANE = pad_mod32(TR * PWD, 2);
    static constexpr int NBANDS = (H + TR - 1) / TR;
    static constexpr int LDS_X = CINP * XPLANE;
    static constexpr int LDS_D = COUT * DPLANE;
    static constexpr size_t LDS_BYTES = (size_t)(LDS_X + LDS_D) * 4;
    static_assert(COUT % 16 == 0, "COUT must be a multiple of 16");
};

template <int CIN, int COUT, int H, int W, int TR, int IN_MODE>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(const void *__restrict__ in_,
                                                                const float *__restrict__ dy,
                                                                float *__restrict__ partial, int n_images)
{
    using C = WgradCfg<CIN, COUT, H, W, TR>;
    extern __shared__ __align__(16) float smem[];
    float *s_x = smem;
    float *s_d = smem + C::LDS_X;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l15 = lane & 15;
    const int g = lane >> 4;

    // per-lane B offsets of this wave's n-tiles: j = (wave + 4*t)*16 + l15 -> (tap, ci)
    int joff[C::NTW];
    int jkind[C::NTW];  // 0: input element, 1: ones column, 2: padding (zero)
#pragma unroll
    for (int t = 0; t < C::NTW; ++t) {
        const int j = (wave + 4 * t) * 16 + l15;
        const int tap = j / C::CINP;
        const int ci = j % C::CINP;
        joff[t] = ci * C::XPLANE + (tap / 3) * C::PWX + (tap % 3);
        jkind[t] = j < 9 * C::CINP ? 0 : (j == 9 * C::CINP ? 1 : 2);
        if (jkind[t] != 0) joff[t] = 0;
    }
    const int aoff = l15 * C::DPLANE + g;  // A: channel l15 of the m-tile, pixel +g

    f32x4 acc[C::MTC][C::NTW];
#pragma unroll
    for (int m = 0; m < C::MTC; ++m)
#pragma unroll
        for (int t = 0; t < C::NTW; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int n_items = n_images * C::NBANDS;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int img = item / C::NBANDS;
        const int y0 = (item % C::NBANDS) * TR;

        __syncthreads();
        stage_band<CIN, C::CINP, H, W, C::ROWS, C::PWX, C::XPLANE, 1, IN_MODE, 256>(in_, img, y0, s_x, tid);
        stage_band<COUT, COUT, H, W, TR, C::PWD, C::DPLANE, 0, IN_NONE, 256>(dy, img, y0, s_d, tid);
        __syncthreads();

        // K loop over the band's pixels, 4 adjacent columns per MFMA
#pragma unroll 1
        for (int r = 0; r < TR; ++r) {
#pragma unroll
            for (int x4 = 0; x4 < C::PWD / 4; ++x4) {
                const int pd = r * C::PWD + x4 * 4;  // dy pixel offset (lane adds g via aoff)
                const int px = r * C::PWX + x4 * 4 + g;
                float a[C::MTC], b[C::NTW];
#pragma unroll
                for (int m = 0; m < C::MTC; ++m) a[m] = s_d[m * 16 * C::DPLANE + aoff + pd];
#pragma unroll
                for (int t = 0; t < C::NTW; ++t) {
                    const float xv = s_x[joff[t] + px];
                    b[t] = jkind[t] == 0 ? xv : (jkind[t] == 1 ? 1.0f : 0.0f);
                }
#pragma unroll
                for (int m = 0; m < C::MTC; ++m)
#pragma unroll
                    for (int t = 0; t < C::NTW; ++t)
                        if ((wave + 4 * t) < C::NTT) acc[m][t] = mfma16(a[m], b[t], acc[m][t]);
            }
        }
    }

    // partial[wg][co][j]; lane holds rows g*4+r (co), column l15 (j) of each tile
    float *slab = partial + (size_t)blockIdx.x * COUT * C::JP;
#pragma unroll
    for (int m = 0; m < C::MTC; ++m)
#pragma unroll
        for (int t = 0; t < C::NTW; ++t) {
            const int nt = wave + 4 * t;
            if (nt < C::NTT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[(m * 16 + g * 4 + r) * C::JP + nt * 16 + l15] = acc[m][t][r];
            }
        }
}

// dW[o][i][tap] = sum_wg partial[wg][o][tap*CINP + i]; db[o] = sum_wg partial[wg][o][9*CINP]
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const float *__restrict__ partial, int n_slabs,
                                                                   int cout, int cin, int cinp, int jp,
                                                                   float *__restrict__ dw, float *__restrict__ db,
                                                                   int accumulate)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int nj = 9 * cinp + 1;
    if (idx >= cout * nj) return;
    const int co = idx / nj;
    const int j = idx % nj;
    const size_t stride = (size_t)cout * jp;
    const float *p = partial + (size_t)co * jp + j;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int s = 0;
    for (; s + 4 <= n_slabs; s += 4) {
        s0 += p[(size_t)(s + 0) * stride];
        s1 += p[(size_t)(s + 1) * stride];
        s2 += p[(size_t)(s + 2) * stride];
        s3 += p[(size_t)(s + 3) * stride];
    }
    for (; s < n_slabs; ++s) s0 += p[(size_t)s * stride];
    const float sum = (s0 + s1) + (s2 + s3);
    if (j == 9 * cinp) {
        if (db) db[co] = accumulate ? db[co] + sum : sum;
    } else {
        const int tap = j / cinp;
        const int ci = j % cinp;
        if (ci < cin) {
            float *d = dw + ((size_t)co * cin + ci) * 9 + tap;
            *d = accumulate ? *d + sum : sum;
        }
    }
}

template <int CIN, int COUT, int H, int W, int TR, int IN_MODE>
int launch_wgrad(const void *in, const float *dy, float *dw, float *db, float *workspace, size_t workspace_bytes,
                 int n_images, int accumulate, hipStream_t st)
{
    using C = WgradCfg<CIN, COUT, H, W, TR>;
    auto kern = conv3x3_wgrad_kernel<CIN, COUT, H, W, TR, IN_MODE>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        if (e != hipSuccess) return fail(PPO_E_HIP, "conv3x3_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int n_items = n_images * C::NBANDS;
    int grid = n_items < 256 ? n_items : 256;
    const size_t need = (size_t)grid * COUT * C::JP * sizeof(float);
    if (need > workspace_bytes)
        return fail(PPO_E_INVALID, "conv3x3_wgrad: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), C::LDS_BYTES, st, in, dy, workspace, n_images);
    int rc = check_launch("conv3x3_wgrad_kernel");
    if (rc) return rc;
    const int total = COUT * (9 * C::CINP + 1);
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, workspace, grid,
                       COUT, CIN, C::CINP, C::JP, dw, db, accumulate);
    return check_launch("conv3x3_wgrad_reduce_kernel");
}

template <int IN_MODE>
int dispatch_wgrad(int cin, int cout, int h, int w_, const void *in, const float *dy, float *dw, float *db,
                   float *ws, size_t ws_bytes, int n, int accumulate, hipStream_t st)
{
#define PPO_WGRAD_CASE(ALLOWED, CI, CO, HH, WW, TR)                                                  \
    if constexpr (ALLOWED) {                                                                         \
        if (cin == CI && cout == CO && h == HH && w_ == WW)                                          \
            return launch_wgrad<CI, CO, HH, WW, TR, IN_MODE>(in, dy, dw, db, ws, ws_bytes, n, accumulate, st); \
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
    return (size_t)256 * cout * jp * sizeof(float);
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

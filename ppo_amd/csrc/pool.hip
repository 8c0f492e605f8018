// 3x3 / stride 2 / pad 1 max pooling, NCHW float32 (reference: F.max_pool2d at
// rl/impala.py:105) and its gradient.  HBM-bound elementwise-class kernels: one thread
// per output element, coalesced along x.
//
// Forward optionally records which of the 9 window taps won (uint8), which is all the
// backward pass needs: the gradient of an input element is the sum of the gradients of
// the (at most four) windows that contain it and chose it.  Ties go to the first tap in
// row-major window order, as in PyTorch's kernel (strict '>' scan).
#include "common.h"

namespace ppo {
namespace {

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                          uint8_t *__restrict__ argmax, int planes, int H, int W,
                                                          int Ho, int Wo)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)planes * Ho * Wo;
    if (idx >= total) return;
    const int ox = idx % Wo;
    const int oy = (idx / Wo) % Ho;
    const int64_t pl = idx / ((int64_t)Wo * Ho);
    const float *src = in + pl * H * W;
    float best = -INFINITY;
    int best_tap = 0;
    bool found = false;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy - 1 + ky;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * ox - 1 + kx;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                const float v = src[iy * W + ix];
                if (!found || v > best || v != v) {
                    best = v;
                    best_tap = ky * 3 + kx;
                    found = true;
                }
            }
        }
    }
    out[idx] = best;
    if (argmax) argmax[idx] = (uint8_t)best_tap;
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float *__restrict__ dout,
                                                          const uint8_t *__restrict__ argmax, float *__restrict__ din,
                                                          int planes, int H, int W, int Ho, int Wo)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)planes * H * W;
    if (idx >= total) return;
    const int ix = idx % W;
    const int iy = (idx / W) % H;
    const int64_t pl = idx / ((int64_t)W * H);
    const float *g = dout + pl * Ho * Wo;
    const uint8_t *a = argmax + pl * Ho * Wo;
    // windows containing row iy: oy = iy/2 (tap row iy - 2oy + 1) and, for odd iy, oy = (iy+1)/2 (tap row 0)
    float sum = 0.f;
#pragma unroll
    for (int sy = 0; sy < 2; ++sy) {
        const int oy = (iy >> 1) + sy;
        const int ky = iy - (2 * oy - 1);
        if (sy == 1 && !(iy & 1)) continue;
        if (oy >= Ho || ky < 0 || ky > 2) continue;
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            const int ox = (ix >> 1) + sx;
            const int kx = ix - (2 * ox - 1);
            if (sx == 1 && !(ix & 1)) continue;
            if (ox >= Wo || kx < 0 || kx > 2) continue;
            if (a[oy * Wo + ox] == ky * 3 + kx) sum += g[oy * Wo + ox];
        }
    }
    din[idx] = sum;
}

}  // namespace
}  // namespace ppo

extern "C" int ppo_maxpool3x3s2_forward_f32(const float *in, float *out, uint8_t *argmax, int n, int c, int h, int w,
                                            void *stream)
{
    using namespace ppo;
    if (n < 0 || c <= 0 || h <= 0 || w <= 0) return fail(PPO_E_INVALID, "ppo_maxpool3x3s2_forward_f32: bad shape");
    if (n == 0) return PPO_OK;
    if (!in || !out) return fail(PPO_E_INVALID, "ppo_maxpool3x3s2_forward_f32: null pointer");
    const int ho = (h + 1) / 2, wo = (w + 1) / 2;
    const int64_t total = (int64_t)n * c * ho * wo;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), in,
                       out, argmax, n * c, h, w, ho, wo);
    return check_launch("maxpool_fwd_kernel");
}

extern "C" int ppo_maxpool3x3s2_backward_f32(const float *dout, const uint8_t *argmax, float *din, int n, int c, int h,
                                             int w, void *stream)
{
    using namespace ppo;
    if (n < 0 || c <= 0 || h <= 0 || w <= 0) return fail(PPO_E_INVALID, "ppo_maxpool3x3s2_backward_f32: bad shape");
    if (n == 0) return PPO_OK;
    if (!dout || !argmax || !din) return fail(PPO_E_INVALID, "ppo_maxpool3x3s2_backward_f32: null pointer");
    const int ho = (h + 1) / 2, wo = (w + 1) / 2;
    const int64_t total = (int64_t)n * c * h * w;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       dout, argmax, din, n * c, h, w, ho, wo);
    return check_launch("maxpool_bwd_kernel");
}

// 3x3 / stride 2 / pad 1 max pooling, NCHW float32 (reference: F.max_pool2d at
// rl/impala.py:105) and its gradient.  HBM-bound elementwise-class kernels.
//
// Forward optionally records which of the 9 window taps won (uint8), which is all the
// backward pass needs: the gradient of an input element is the sum of the gradients of
// the (at most four) windows that contain it and chose it.  Ties go to the first tap in
// row-major window order, as in PyTorch's kernel (strict '>' scan).
//
// Indexing: blockIdx.y = (image, channel) plane, blockIdx.x * blockDim.x + threadIdx.x walks the
// plane, so the only integer division per thread is by the row length.  The backward kernel
// produces two adjacent input columns per thread (they share their candidate windows) and stores
// them with one 8-byte access when the row length is even.
#include "common.h"

namespace ppo {
namespace {

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                          uint8_t *__restrict__ argmax, int H, int W, int Ho, int Wo)
{
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= Ho * Wo) return;
    const int64_t pl = blockIdx.y;
    const int oy = o / Wo;
    const int ox = o - oy * Wo;
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
    out[pl * Ho * Wo + o] = best;
    if (argmax) argmax[pl * Ho * Wo + o] = (uint8_t)best_tap;
}

// one thread -> input columns 2k, 2k+1 of one row
template <bool PAIR_STORE>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float *__restrict__ dout,
                                                          const uint8_t *__restrict__ argmax, float *__restrict__ din,
                                                          int H, int W, int Ho, int Wo)
{
    const int Wh = (W + 1) / 2;  // column pairs per row
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= H * Wh) return;
    const int64_t pl = blockIdx.y;
    const int iy = t / Wh;
    const int k = t - iy * Wh;
    const float *g = dout + pl * Ho * Wo;
    const uint8_t *a = argmax + pl * Ho * Wo;
    // rows: oy0 = iy/2 holds tap row (iy even ? 1 : 2); for odd iy, oy0+1 holds tap row 0
    const int oy0 = iy >> 1;
    const int ky0 = (iy & 1) ? 2 : 1;
    const bool has_oy1 = (iy & 1) && (oy0 + 1 < Ho);
    // columns: ix0 = 2k is tap col 1 of ox = k; ix1 = 2k+1 is tap col 2 of ox = k and tap col 0 of ox = k+1
    const bool has_ox1 = (k + 1 < Wo);
    // All four (argmax, gradient) pairs are read unconditionally through range-checked descriptors (common.h): a
    // window that does not exist reads gradient 0.  (Reading the gradient only when the argmax matched made every
    // load a branch with a full memory wait behind it.)  The additions keep the order of the conditional form.
    const __amdgpu_buffer_rsrc_t gb = buffer_of(g), ab = buffer_of(a);
    const int i00 = oy0 * Wo + k, i10 = i00 + Wo;
    const int o00 = i00, o01 = has_ox1 ? i00 + 1 : -1, o10 = has_oy1 ? i10 : -1, o11 = (has_oy1 && has_ox1) ? i10 + 1 : -1;
    const int t00 = __builtin_amdgcn_raw_buffer_load_b8(ab, o00, 0, 0);
    const int t01 = __builtin_amdgcn_raw_buffer_load_b8(ab, o01 < 0 ? kOutside : o01, 0, 0);
    const int t10 = __builtin_amdgcn_raw_buffer_load_b8(ab, o10 < 0 ? kOutside : o10, 0, 0);
    const int t11 = __builtin_amdgcn_raw_buffer_load_b8(ab, o11 < 0 ? kOutside : o11, 0, 0);
    const float g00 = buffer_f32(gb, o00 * 4);
    const float g01 = buffer_f32(gb, o01 < 0 ? kOutside : o01 * 4);
    const float g10 = buffer_f32(gb, o10 < 0 ? kOutside : o10 * 4);
    const float g11 = buffer_f32(gb, o11 < 0 ? kOutside : o11 * 4);
    float s0 = 0.f, s1 = 0.f;
    s0 += t00 == ky0 * 3 + 1 ? g00 : 0.f;
    s1 += t00 == ky0 * 3 + 2 ? g00 : 0.f;
    s1 += t01 == ky0 * 3 + 0 ? g01 : 0.f;
    s0 += t10 == 1 ? g10 : 0.f;
    s1 += t10 == 2 ? g10 : 0.f;
    s1 += t11 == 0 ? g11 : 0.f;
    float *dst = din + pl * H * W + iy * W + 2 * k;
    if (PAIR_STORE) {
        *reinterpret_cast<float2 *>(dst) = make_float2(s0, s1);
    } else {
        dst[0] = s0;
        if (2 * k + 1 < W) dst[1] = s1;
    }
}

// Even H and W (every IMPALA stack input except the odd 21x21 / 11x11 ones): one thread -> the 2x2 input block
// (rows 2j, 2j+1; columns 2k, 2k+1).  The four windows that can select these elements are (j,k), (j,k+1),
// (j+1,k), (j+1,k+1): 4 (argmax, gradient) pairs feed 4 outputs, where the row-pair form above reads 6.
__global__ __launch_bounds__(256) void maxpool_bwd2x2_kernel(const float *__restrict__ dout,
                                                             const uint8_t *__restrict__ argmax,
                                                             float *__restrict__ din, int H, int W, int Ho, int Wo,
                                                             int64_t n_windows)
{
    // one thread per pooling window, flat over (plane, j, k): a (Ho*Wo)-per-plane grid left the last workgroup of every
    // 21x21 plane 72 % idle.  The four (argmax, gradient) pairs are read unconditionally through range-checked
    // descriptors based at the thread's plane (common.h): a neighbour that does not exist reads gradient 0.
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_windows) return;
    const int64_t pl = idx / (Ho * Wo);
    const int t = (int)(idx - pl * (Ho * Wo));
    const int j = t / Wo;
    const int k = t - j * Wo;
    const bool right = k + 1 < Wo, down = j + 1 < Ho;
    const __amdgpu_buffer_rsrc_t gb = buffer_of(dout), ab = buffer_of(argmax);  // whole tensors: the host checks < 2 GiB
    const int o00 = (int)idx, o01 = right ? o00 + 1 : -1, o10 = down ? o00 + Wo : -1, o11 = (right && down) ? o00 + Wo + 1 : -1;
    const int t00 = __builtin_amdgcn_raw_buffer_load_b8(ab, o00, 0, 0);
    const int t01 = __builtin_amdgcn_raw_buffer_load_b8(ab, o01 < 0 ? kOutside : o01, 0, 0);
    const int t10 = __builtin_amdgcn_raw_buffer_load_b8(ab, o10 < 0 ? kOutside : o10, 0, 0);
    const int t11 = __builtin_amdgcn_raw_buffer_load_b8(ab, o11 < 0 ? kOutside : o11, 0, 0);
    const float g00 = buffer_f32(gb, o00 * 4);
    const float g01 = buffer_f32(gb, o01 < 0 ? kOutside : o01 * 4);
    const float g10 = buffer_f32(gb, o10 < 0 ? kOutside : o10 * 4);
    const float g11 = buffer_f32(gb, o11 < 0 ? kOutside : o11 * 4);
    // taps are ky*3+kx; summation order per element = window order (j,k), (j,k+1), (j+1,k), (j+1,k+1), the same
    // order the row-pair kernel uses
    const float r00 = (t00 == 4 ? g00 : 0.f);
    float r01 = (t00 == 5 ? g00 : 0.f);
    r01 += (t01 == 3 ? g01 : 0.f);
    float r10 = (t00 == 7 ? g00 : 0.f);
    r10 += (t10 == 1 ? g10 : 0.f);
    float r11 = (t00 == 8 ? g00 : 0.f);
    r11 += (t01 == 6 ? g01 : 0.f);
    r11 += (t10 == 2 ? g10 : 0.f);
    r11 += (t11 == 0 ? g11 : 0.f);
    float *dst = din + pl * H * W + (2 * j) * W + 2 * k;
    *reinterpret_cast<float2 *>(dst) = make_float2(r00, r01);
    *reinterpret_cast<float2 *>(dst + W) = make_float2(r10, r11);
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
    if ((int64_t)n * c > 65535 * 1024LL) return fail(PPO_E_INVALID, "ppo_maxpool3x3s2_forward_f32: too many planes");
    const int ho = (h + 1) / 2, wo = (w + 1) / 2;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((ho * wo + 255) / 256, n * c), dim3(256), 0, as_stream(stream), in, out,
                       argmax, h, w, ho, wo);
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
    const int threads = h * ((w + 1) / 2);
    const dim3 grid((threads + 255) / 256, n * c);
    if (w % 2 == 0 && h % 2 == 0 && aligned(din, 8) && (int64_t)n * c * ho * wo * 4 + 16 < (int64_t)kBufferBytes)
        hipLaunchKernelGGL(maxpool_bwd2x2_kernel, dim3((unsigned)(((int64_t)n * c * ho * wo + 255) / 256)), dim3(256), 0,
                           as_stream(stream), dout, argmax, din, h, w, ho, wo, (int64_t)n * c * ho * wo);
    else if (w % 2 == 0 && aligned(din, 8))
        hipLaunchKernelGGL(maxpool_bwd_kernel<true>, grid, dim3(256), 0, as_stream(stream), dout, argmax, din, h, w, ho, wo);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel<false>, grid, dim3(256), 0, as_stream(stream), dout, argmax, din, h, w, ho, wo);
    return check_launch("maxpool_bwd_kernel");
}

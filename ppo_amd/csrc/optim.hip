// Optimiser step over ONE flat parameter buffer (gfx950): global-norm gradient clipping
// (nn.utils.clip_grad_norm_, rl/rollout.py:1309-1310) fused with Adam
// (torch.optim.Adam as built at rl/rollout.py:126-141; eps 1e-5, no weight decay).
// HBM-bound: 16 B read + 12 B written per parameter (w, g, m, v in; w, m, v out).
//
//   1. grad_sumsq_kernel   per-workgroup partial sums of g^2 (fixed order => deterministic)
//   2. adam_kernel         every workgroup re-reduces the <= kPartials partials in the same
//                          order (so all agree bit-for-bit), forms
//                          clip = min(1, max_norm / (norm + 1e-6)) and applies Adam to its slice.
// No host synchronisation: the norm stays on the device (also written to norm_out for logging).
#include "common.h"

namespace ppo {
namespace {

constexpr int kPartials = 256;
__device__ __forceinline__ bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float *__restrict__ g, int64_t n, float *__restrict__ partials)
{
    __shared__ float s[256];
    // contiguous chunk per workgroup, grid-stride inside it, so the result does not depend on timing
    const int64_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * chunk;
    const int64_t hi = lo + chunk < n ? lo + chunk : n;
    float acc = 0.f;
    int64_t i = lo + threadIdx.x;
    for (; i + 768 < hi; i += 1024) {  // four independent loads in flight per thread
        const float x0 = g[i], x1 = g[i + 256], x2 = g[i + 512], x3 = g[i + 768];
        acc += x0 * x0;
        acc += x1 * x1;
        acc += x2 * x2;
        acc += x3 * x3;
    }
    for (; i < hi; i += 256) {
        const float x = g[i];
        acc += x * x;
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ w, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v, int64_t n,
                                                   const float *__restrict__ partials, int n_partials, float lr,
                                                   float one_minus_beta1, float beta2, float one_minus_beta2,
                                                   float eps, float bias_c1,
                                                   float bias_c2_sqrt, float max_norm, float grad_div,
                                                   float *__restrict__ norm_out, const int32_t *__restrict__ scatter,
                                                   int64_t n_scatter, float *__restrict__ packed)
{
    __shared__ float s_clip;
    __shared__ float s_part[256];
    // every workgroup re-reduces the partials the same way (one load per thread, then a fixed-order tree), so all
    // of them clip by the same factor without a grid-wide exchange
    s_part[threadIdx.x] = (int)threadIdx.x < n_partials ? partials[threadIdx.x] : 0.f;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s_part[threadIdx.x] += s_part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float tot = s_part[0];
        // gradients may still carry a 1/world_size (or micro-batch) factor: grad_div
        const float norm = sqrtf(tot) / grad_div;
        float clip = 1.f;
        if (max_norm > 0.f) {
            clip = max_norm / (norm + 1e-6f);
            clip = clip > 1.f ? 1.f : clip;
        }
        s_clip = clip / grad_div;
        if (norm_out && blockIdx.x == 0) *norm_out = norm;
    }
    __syncthreads();
    const float gscale = s_clip;
    const float step_size = lr / bias_c1;
    auto update = [&](float &wi, float gi, float &mi, float &vi) {
        gi *= gscale;
        mi = mi + (gi - mi) * one_minus_beta1;           // exp_avg.lerp_(grad, 1 - beta1)
        vi = vi * beta2 + one_minus_beta2 * gi * gi;     // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
        const float denom = sqrtf(vi) / bias_c2_sqrt + eps;
        wi = wi - step_size * (mi / denom);
    };
    // the flat buffers are 16-byte aligned and padded to a multiple of 4 (ppo_amd/models.py); otherwise scalar
    const bool vec = (n % 4 == 0) && aligned16(w) && aligned16(g) && aligned16(m) && aligned16(v);
    if (vec) {
        const int64_t n4 = n / 4;
        float4 *w4 = reinterpret_cast<float4 *>(w), *m4 = reinterpret_cast<float4 *>(m), *v4 = reinterpret_cast<float4 *>(v);
        const float4 *g4 = reinterpret_cast<const float4 *>(g);
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
            float4 ww = w4[i], gg = g4[i], mm = m4[i], vv = v4[i];
            update(ww.x, gg.x, mm.x, vv.x);
            update(ww.y, gg.y, mm.y, vv.y);
            update(ww.z, gg.z, mm.z, vv.z);
            update(ww.w, gg.w, mm.w, vv.w);
            w4[i] = ww;
            m4[i] = mm;
            v4[i] = vv;
            if (4 * i < n_scatter) {
                // the head of the flat buffer holds the convolution weights: their updated values also go to where the
                // convolution kernels read them, the per-lane MFMA operand layouts (forward and flipped / transposed),
                // so no re-pack launch sits between this step and the next forward
                const float nw[4] = {ww.x, ww.y, ww.z, ww.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int64_t j = 4 * i + e;
                    if (j < n_scatter) {
                        const int d0 = scatter[2 * j], d1 = scatter[2 * j + 1];
                        if (d0 >= 0) packed[d0] = nw[e];
                        if (d1 >= 0) packed[d1] = nw[e];
                    }
                }
            }
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            float wi = w[i], mi = m[i], vi = v[i];
            update(wi, g[i], mi, vi);
            w[i] = wi;
            m[i] = mi;
            v[i] = vi;
            if (i < n_scatter) {
                const int d0 = scatter[2 * i], d1 = scatter[2 * i + 1];
                if (d0 >= 0) packed[d0] = wi;
                if (d1 >= 0) packed[d1] = wi;
            }
        }
    }
}

}  // namespace
}  // namespace ppo

extern "C" size_t ppo_adam_workspace_bytes(void) { return ppo::kPartials * sizeof(float); }

static int adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, int64_t step, double lr,
                     double beta1, double beta2, double eps, float max_grad_norm, float grad_div, void *workspace,
                     float *grad_norm_out, const int32_t *scatter, int64_t n_scatter, float *packed, void *stream);

extern "C" int ppo_adam_step_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                                 int64_t step, double lr, double beta1, double beta2, double eps, float max_grad_norm,
                                 float grad_div, void *workspace, float *grad_norm_out, void *stream)
{
    return adam_step(params, grads, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, max_grad_norm, grad_div, workspace,
                     grad_norm_out, nullptr, 0, nullptr, stream);
}

extern "C" int ppo_adam_step_scatter_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                                         int64_t step, double lr, double beta1, double beta2, double eps,
                                         float max_grad_norm, float grad_div, void *workspace, float *grad_norm_out,
                                         const int32_t *scatter, int64_t n_scatter, float *packed, void *stream)
{
    if (n_scatter < 0 || n_scatter > n || (n_scatter > 0 && (!scatter || !packed)))
        return ppo::fail(PPO_E_INVALID, "ppo_adam_step_scatter_f32: bad scatter table");
    return adam_step(params, grads, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, max_grad_norm, grad_div, workspace,
                     grad_norm_out, scatter, n_scatter, packed, stream);
}

extern "C" int ppo_adam_step_presummed_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                                           int64_t step, double lr, double beta1, double beta2, double eps,
                                           float max_grad_norm, float grad_div, const float *partials, int n_partials,
                                           float *grad_norm_out, void *stream)
{
    using namespace ppo;
    if (n < 0 || step < 1) return fail(PPO_E_INVALID, "ppo_adam_step_presummed_f32: n < 0 or step < 1");
    if (n == 0) return PPO_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !partials || n_partials < 1 || n_partials > kPartials)
        return fail(PPO_E_INVALID, "ppo_adam_step_presummed_f32: null pointer or partial count outside [1, %d]", kPartials);
    if (!(grad_div > 0.f)) return fail(PPO_E_INVALID, "ppo_adam_step_presummed_f32: grad_div must be > 0");
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    int grid = (int)((n + 255) / 256);
    grid = grid > 1024 ? 1024 : grid;
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, as_stream(stream), params, grads, exp_avg, exp_avg_sq, n, partials,
                       n_partials, (float)lr, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)bc1,
                       (float)sqrt(bc2), max_grad_norm, grad_div, grad_norm_out, nullptr, (int64_t)0, nullptr);
    return check_launch("adam_kernel");
}

static int adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, int64_t step, double lr,
                     double beta1, double beta2, double eps, float max_grad_norm, float grad_div, void *workspace,
                     float *grad_norm_out, const int32_t *scatter, int64_t n_scatter, float *packed, void *stream)
{
    using namespace ppo;
    if (n < 0 || step < 1) return fail(PPO_E_INVALID, "ppo_adam_step_f32: n < 0 or step < 1");
    if (n == 0) return PPO_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !workspace)
        return fail(PPO_E_INVALID, "ppo_adam_step_f32: null pointer");
    if (!(grad_div > 0.f)) return fail(PPO_E_INVALID, "ppo_adam_step_f32: grad_div must be > 0");
    hipStream_t st = as_stream(stream);
    float *partials = static_cast<float *>(workspace);
    int np = (int)((n + 4095) / 4096);
    np = np > kPartials ? kPartials : np;
    hipLaunchKernelGGL(grad_sumsq_kernel, dim3(np), dim3(256), 0, st, grads, n, partials);
    int rc = check_launch("grad_sumsq_kernel");
    if (rc) return rc;
    // bias corrections in double on the host, as torch does with python floats
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    int grid = (int)((n + 255) / 256);
    grid = grid > 1024 ? 1024 : grid;
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, n, partials, np,
                       (float)lr, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)bc1,
                       (float)sqrt(bc2), max_grad_norm, grad_div, grad_norm_out, scatter, n_scatter, packed);
    return check_launch("adam_kernel");
}

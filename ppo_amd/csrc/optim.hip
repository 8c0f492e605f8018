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

__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float *__restrict__ g, int64_t n, float *__restrict__ partials)
{
    __shared__ float s[256];
    // contiguous chunk per workgroup, grid-stride inside it, so the result does not depend on timing
    const int64_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * chunk;
    const int64_t hi = lo + chunk < n ? lo + chunk : n;
    float acc = 0.f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
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
                                                   float *__restrict__ norm_out)
{
    __shared__ float s_clip;
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int i = 0; i < n_partials; ++i) tot += partials[i];
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
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        float mi = m[i], vi = v[i];
        mi = mi + (gi - mi) * one_minus_beta1;           // exp_avg.lerp_(grad, 1 - beta1)
        vi = vi * beta2 + one_minus_beta2 * gi * gi;     // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
        const float denom = sqrtf(vi) / bias_c2_sqrt + eps;
        w[i] = w[i] - step_size * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
}

}  // namespace
}  // namespace ppo

extern "C" size_t ppo_adam_workspace_bytes(void) { return ppo::kPartials * sizeof(float); }

extern "C" int ppo_adam_step_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                                 int64_t step, double lr, double beta1, double beta2, double eps, float max_grad_norm,
                                 float grad_div, void *workspace, float *grad_norm_out, void *stream)
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
                       (float)sqrt(bc2), max_grad_norm, grad_div, grad_norm_out);
    return check_launch("adam_kernel");
}

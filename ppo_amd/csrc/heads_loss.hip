// Head-side kernels beyond the discrete PPO policy loss (gfx950): everything between the fused head GEMM
// and the backward pass for the value phase, the distillation phase, TVF heads and gaussian policies.
//
//  * ppo_tanh_forward_f32 / ppo_tanh_backward_f32   encoder activation of the MLP / mujoco configs
//                                                    (rl/models.py:166, 456-457); HBM-bound elementwise.
//  * ppo_value_loss_f32     value phase of the dual architecture: vf_coef (V-R)^2 per value head
//                           (rl/rollout.py:1596-1608) + TVF loss coef 0.5 w_k (T_k - P_k)^2, sqrt(K) mean_k
//                           (rl/tvf.py:32-77), forward and gradient w.r.t. the head row.
//  * ppo_distil_loss_f32    distillation phase: 0.5 w_k (T_k - P_k)^2 [sqrt(K) mean_k when the targets are a
//                           vector] + beta KL(pi_new || pi_old) (rl/rollout.py:1331-1449, value_loss "mse",
//                           loss "kl_policy").
//  * ppo_gaussian_act_f32   a = mu + exp(log_std) * n, n ~ N(0,1) by Box-Muller from the counter-based uniform
//                           stream (rl/rollout.py:643-648), plus log N(a; mu, sigma) per dimension.
//  * ppo_gaussian_loss_f32  clipped surrogate per action dimension, mean over dimensions, minus the value
//                           loss (rl/rollout.py:1693-1704, 1744-1753) with gradients for mu, V and log_std.
//
// Layout: one head row per sample, leading dimension ldo; each loss kernel writes the WHOLE gradient row
// (zeros in columns its loss does not touch).  One wave per sample: lanes stride the row's columns, so the
// row is read and written coalesced, and per-sample reductions are wave shuffles.  `index` (nullable)
// maps a minibatch row to its row in the un-gathered batch arrays.
#include "common.h"
#include "loss_rows.h"

namespace ppo {
namespace {

__global__ __launch_bounds__(256) void tanh_forward_kernel(const float *__restrict__ x, float *__restrict__ y, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = tanhf(x[i]);
}

__global__ __launch_bounds__(256) void tanh_backward_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                            float *__restrict__ dx, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
#pragma clang fp contract(off)  // torch's tanh_backward rounds t*t before the subtraction
        const float t = y[i];
        const float tt = t * t;
        dx[i] = dy[i] * (1.f - tt);
    }
}

// 4 samples per 256-thread block, one wave each; the per-sample bodies live in loss_rows.h
__global__ __launch_bounds__(256) void value_loss_kernel(const float *__restrict__ heads, int B, ValueLossP p,
                                                         float *__restrict__ dheads, const int32_t *__restrict__ index)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    value_loss_row(p, heads + (size_t)b * p.ldo, dheads + (size_t)b * p.ldo, b, index ? index[b] : b, lane);
}

__global__ __launch_bounds__(256) void distil_loss_kernel(const float *__restrict__ heads, int B, DistilLossP p,
                                                          float *__restrict__ dheads, const int32_t *__restrict__ index)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    distil_loss_row(p, heads + (size_t)b * p.ldo, dheads + (size_t)b * p.ldo, b, index ? index[b] : b, lane);
}

__global__ __launch_bounds__(256) void gaussian_act_kernel(
    const float *__restrict__ heads, int B, int ldo, int nA, const float *__restrict__ log_std,
    const float *__restrict__ normal, uint64_t seed, uint64_t offset, int deterministic, float *__restrict__ actions,
    float *__restrict__ log_pac, float *__restrict__ raw_policy, float *__restrict__ values, int vh)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * nA) return;
    const int b = i / nA, a = i % nA;
    const float mu = heads[(size_t)b * ldo + a];
    float n = 0.f;
    if (!deterministic) {
        if (normal) {
            n = normal[i];
        } else {  // Box-Muller on two counters per element
            const float u1 = uniform01(seed, 2 * (offset + (uint64_t)i));
            const float u2 = uniform01(seed, 2 * (offset + (uint64_t)i) + 1);
            n = sqrtf(-2.f * logf(u1)) * cospif(2.f * u2);
        }
    }
    const float ls = log_std[a];
    const float act = deterministic ? mu : n * expf(ls) + mu;
    if (actions) actions[i] = act;
    if (raw_policy) raw_policy[i] = mu;
    if (log_pac) {  // Normal(mu, sigma).log_prob(act) = -(act-mu)^2 / (2 sigma^2) - log sigma - log sqrt(2 pi)
        const float var = expf(ls) * expf(ls);
        log_pac[i] = -((act - mu) * (act - mu)) / (2.f * var) - ls - kLogSqrt2Pi;
    }
    if (values && a == 0)
        for (int j = 0; j < vh; ++j) values[(size_t)b * vh + j] = heads[(size_t)b * ldo + nA + j];
}

__global__ __launch_bounds__(256) void gaussian_loss_kernel(const float *__restrict__ heads, int B, GaussLossP p,
                                                            float *__restrict__ dheads, float *__restrict__ dlog_std_rows,
                                                            const int32_t *__restrict__ index)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    gaussian_loss_row(p, heads + (size_t)b * p.ldo, dheads + (size_t)b * p.ldo,
                      dlog_std_rows ? dlog_std_rows + (size_t)b * p.nA : nullptr, b, index ? index[b] : b, lane);
}

int grid_1d(size_t n)
{
    size_t g = (n + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g ? g : 1));
}

}  // namespace
}  // namespace ppo

extern "C" int ppo_tanh_forward_f32(const float *x, float *y, size_t n, void *stream)
{
    using namespace ppo;
    if (n == 0) return PPO_OK;
    if (!x || !y) return fail(PPO_E_INVALID, "ppo_tanh_forward_f32: null pointer");
    hipLaunchKernelGGL(tanh_forward_kernel, dim3(grid_1d(n)), dim3(256), 0, as_stream(stream), x, y, n);
    return check_launch("tanh_forward_kernel");
}

extern "C" int ppo_tanh_backward_f32(const float *dy, const float *y, float *dx, size_t n, void *stream)
{
    using namespace ppo;
    if (n == 0) return PPO_OK;
    if (!dy || !y || !dx) return fail(PPO_E_INVALID, "ppo_tanh_backward_f32: null pointer");
    hipLaunchKernelGGL(tanh_backward_kernel, dim3(grid_1d(n)), dim3(256), 0, as_stream(stream), dy, y, dx, n);
    return check_launch("tanh_backward_kernel");
}

extern "C" int ppo_value_loss_f32(const float *heads, int B, int ldo, int value_col, int n_value_heads,
                                  const float *returns, float vf_coef, int tvf_col, int n_tvf, int tvf_stride,
                                  const float *tvf_returns, const float *tvf_weights, float tvf_coef, float grad_scale,
                                  float *dheads, float *stats, const int32_t *index, float tvf_keep_prob, uint64_t seed,
                                  uint64_t offset, void *stream)
{
    using namespace ppo;
    if (B < 0 || ldo <= 0 || n_value_heads < 0 || n_tvf < 0 || value_col < 0 || value_col + n_value_heads > ldo)
        return fail(PPO_E_INVALID, "ppo_value_loss_f32: bad shape");
    if (n_tvf > 0 && (tvf_stride <= 0 || tvf_col < 0 || tvf_col + (n_tvf - 1) * tvf_stride >= ldo))
        return fail(PPO_E_INVALID, "ppo_value_loss_f32: TVF columns [%d + k*%d, k < %d) exceed the row (%d)", tvf_col,
                    tvf_stride, n_tvf, ldo);
    if (!(tvf_keep_prob > 0.f)) return fail(PPO_E_INVALID, "ppo_value_loss_f32: tvf_keep_prob must be in (0, 1]");
    if (B == 0) return PPO_OK;
    if (!heads || !dheads) return fail(PPO_E_INVALID, "ppo_value_loss_f32: null pointer");
    if (n_tvf == 0) tvf_returns = nullptr;
    if (n_value_heads == 0) returns = nullptr;
    const ValueLossP p{ldo, value_col, n_value_heads, returns, vf_coef, tvf_col, n_tvf, n_tvf > 0 ? tvf_stride : 1,
                       tvf_returns, tvf_weights, tvf_coef, grad_scale, stats, tvf_keep_prob, seed, offset};
    hipLaunchKernelGGL(value_loss_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(stream), heads, B, p, dheads, index);
    return check_launch("value_loss_kernel");
}

extern "C" int ppo_distil_loss_f32(const float *heads, int B, int ldo, int n_actions, int pred_col, int n_pred,
                                   int pred_stride, int vector_targets, const float *targets, const float *weights,
                                   const float *old_policy, const float *log_std, float beta, float grad_scale,
                                   float *dheads, float *stats, const int32_t *index, void *stream)
{
    using namespace ppo;
    if (B < 0 || n_actions <= 0 || n_actions > kMaxActions || n_pred <= 0 || pred_stride <= 0 || pred_col < n_actions ||
        pred_col + (n_pred - 1) * pred_stride >= ldo)
        return fail(PPO_E_INVALID, "ppo_distil_loss_f32: bad shape");
    if (B == 0) return PPO_OK;
    if (!heads || !targets || !old_policy || !dheads) return fail(PPO_E_INVALID, "ppo_distil_loss_f32: null pointer");
    const DistilLossP p{ldo, n_actions, pred_col, n_pred, pred_stride, vector_targets, targets, weights, old_policy,
                        log_std, beta, grad_scale, stats};
    hipLaunchKernelGGL(distil_loss_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(stream), heads, B, p, dheads, index);
    return check_launch("distil_loss_kernel");
}

extern "C" int ppo_gaussian_act_f32(const float *heads, int B, int ldo, int n_actions, const float *log_std,
                                    const float *normal, uint64_t seed, uint64_t offset, int deterministic,
                                    float *actions, float *log_pac, float *raw_policy, float *values,
                                    int n_value_heads, void *stream)
{
    using namespace ppo;
    if (B < 0 || n_actions <= 0 || n_value_heads < 0 || ldo < n_actions + n_value_heads)
        return fail(PPO_E_INVALID, "ppo_gaussian_act_f32: bad shape");
    if (B == 0) return PPO_OK;
    if (!heads || !log_std) return fail(PPO_E_INVALID, "ppo_gaussian_act_f32: null pointer");
    const int n = B * n_actions;
    hipLaunchKernelGGL(gaussian_act_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), heads, B, ldo,
                       n_actions, log_std, normal, seed, offset, deterministic, actions, log_pac, raw_policy, values,
                       n_value_heads);
    return check_launch("gaussian_act_kernel");
}

extern "C" int ppo_gaussian_loss_f32(const float *heads, int B, int ldo, int n_actions, int n_value_heads,
                                     const float *actions, const float *old_log_pac, const float *advantages,
                                     const float *returns, const float *log_std, float eps_clip, float vf_coef,
                                     float grad_scale, float *dheads, float *dlog_std_rows, float *stats,
                                     const int32_t *index, void *stream)
{
    using namespace ppo;
    if (B < 0 || n_actions <= 0 || n_value_heads < 0 || ldo < n_actions + n_value_heads)
        return fail(PPO_E_INVALID, "ppo_gaussian_loss_f32: bad shape");
    if (B == 0) return PPO_OK;
    if (!heads || !actions || !old_log_pac || !advantages || !log_std || !dheads || (n_value_heads > 0 && !returns))
        return fail(PPO_E_INVALID, "ppo_gaussian_loss_f32: null pointer");
    const GaussLossP p{ldo, n_actions, n_value_heads, actions, old_log_pac, advantages, returns, log_std, eps_clip, vf_coef,
                       grad_scale, stats};
    hipLaunchKernelGGL(gaussian_loss_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(stream), heads, B, p, dheads,
                       dlog_std_rows, index);
    return check_launch("gaussian_loss_kernel");
}

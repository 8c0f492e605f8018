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

namespace ppo {
namespace {

constexpr int kMaxActions = 32;

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

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

enum { VS_VALUE = 0, VS_TVF, VS_TOTAL, VS_SPARE, VS_N };

__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t counter)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (counter + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return ((float)(z >> 40) + 0.5f) * (1.0f / 16777216.0f);
}

// 4 samples per 256-thread block, one wave each
__global__ __launch_bounds__(256) void value_loss_kernel(
    const float *__restrict__ heads, int B, int ldo, int value_col, int vh, const float *__restrict__ returns,
    float vf_coef, int tvf_col, int K, int tvf_stride, const float *__restrict__ tvf_returns,
    const float *__restrict__ tvf_weights, float tvf_coef, float grad_scale, float *__restrict__ dheads,
    float *__restrict__ stats, const int32_t *__restrict__ index, float keep_prob, uint64_t seed, uint64_t offset)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    const float *z = heads + (size_t)b * ldo;
    float *dz = dheads + (size_t)b * ldo;
    const int sb = index ? index[b] : b;
    const float tvf_scale = K > 0 ? tvf_coef / sqrtf((float)K) : 0.f;  // sqrt(K) * mean_k = sum_k / sqrt(K)
    float vloss = 0.f, tloss = 0.f;
    for (int c = lane; c < ldo; c += 64) {
        float g = 0.f;
        if (returns && c >= value_col && c < value_col + vh) {
            const float diff = z[c] - returns[(size_t)sb * vh + (c - value_col)];
            vloss += vf_coef * diff * diff;
            g = 2.f * vf_coef * diff;
        } else if (tvf_returns && c >= tvf_col && c < tvf_col + K * tvf_stride && (c - tvf_col) % tvf_stride == 0) {
            const int k = (c - tvf_col) / tvf_stride;
            float w = tvf_weights ? tvf_weights[k] : 1.f;
            // horizon dropout (rl/tvf.py:64-69): each (sample, head) term is kept with probability keep_prob and
            // weighted 1 / keep_prob; the draw is a counter-based uniform keyed by (seed, offset + b * K + k)
            if (keep_prob < 1.f) w = uniform01(seed, offset + (uint64_t)b * K + k) < keep_prob ? w / keep_prob : 0.f;
            const float diff = z[c] - tvf_returns[(size_t)sb * K + k];
            tloss += 0.5f * tvf_scale * w * diff * diff;
            g = tvf_scale * w * diff;
        }
        dz[c] = grad_scale * g;
    }
    if (stats) {
        vloss = wave_sum(vloss);
        tloss = wave_sum(tloss);
        if (lane == 0) {
            float *s = stats + (size_t)b * VS_N;
            s[VS_VALUE] = vloss;
            s[VS_TVF] = tloss;
            s[VS_TOTAL] = vloss + tloss;
            s[VS_SPARE] = 0.f;
        }
    }
}

enum { DS_VALUE = 0, DS_POLICY, DS_TOTAL, DS_SQERR, DS_N };

__global__ __launch_bounds__(256) void distil_loss_kernel(
    const float *__restrict__ heads, int B, int ldo, int nA, int pred_col, int n_pred, int pred_stride, int vector_targets,
    const float *__restrict__ targets, const float *__restrict__ weights, const float *__restrict__ old_policy,
    const float *__restrict__ log_std, float beta, float grad_scale, float *__restrict__ dheads,
    float *__restrict__ stats, const int32_t *__restrict__ index)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    const float *z = heads + (size_t)b * ldo;
    float *dz = dheads + (size_t)b * ldo;
    const int sb = index ? index[b] : b;
    // policy term over the nA policy outputs (nA <= 32 <= 64: one lane per action)
    const float logit = lane < nA ? z[lane] : -INFINITY;
    const float old = lane < nA ? old_policy[(size_t)sb * nA + lane] : 0.f;
    float kl, gpol;
    if (log_std) {
        // gaussian (rl/rollout.py:1401-1409): 0.5 mean_a (mu_old - mu)^2 / (1e-5 + 2 sigma^2), sigma detached
        const float sigma = lane < nA ? expf(log_std[lane]) : 1.f;
        const float den = 1e-5f + 2.f * sigma * sigma;
        const float d = lane < nA ? logit - old : 0.f;
        // the reference adds this term to the loss twice (rl/rollout.py:1409 and again :1419): keep its scale
        kl = 2.f * wave_sum(0.5f * d * d / den) / nA;
        gpol = 2.f * d / den / nA;
    } else {
        // discrete: KL(new || old);  d KL / d z_j = p_j (log p_j - log q_j - KL)
        const float mx = wave_max(logit);
        const float e = lane < nA ? expf(logit - mx) : 0.f;
        const float lse = mx + logf(wave_sum(e));
        const float lp = logit - lse;
        const float p = lane < nA ? expf(lp) : 0.f;
        kl = wave_sum(lane < nA ? p * (lp - old) : 0.f);
        gpol = p * (lp - old - kl);
    }
    const float vscale = vector_targets ? 1.f / sqrtf((float)n_pred) : 1.f;
    float vloss = 0.f, sq = 0.f;
    for (int c = lane; c < ldo; c += 64) {
        float g = 0.f;
        if (c < nA) {
            g = beta * gpol;  // c == lane here
        } else if (c >= pred_col && c < pred_col + n_pred * pred_stride && (c - pred_col) % pred_stride == 0) {
            const int k = (c - pred_col) / pred_stride;
            const float w = weights ? weights[k] : 1.f;
            const float diff = z[c] - targets[(size_t)sb * n_pred + k];
            vloss += 0.5f * vscale * w * diff * diff;
            sq += diff * diff * w * w;
            g = vscale * w * diff;
        }
        dz[c] = grad_scale * g;
    }
    if (stats) {
        vloss = wave_sum(vloss);
        sq = wave_sum(sq);
        if (lane == 0) {
            float *s = stats + (size_t)b * DS_N;
            s[DS_VALUE] = vloss;
            s[DS_POLICY] = beta * kl;
            s[DS_TOTAL] = vloss + beta * kl;
            s[DS_SQERR] = sq / n_pred;
        }
    }
}

constexpr float kLogSqrt2Pi = 0.91893853320467274178f;

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

enum { GS_LOSS_CLIP = 0, GS_SPARE, GS_VALUE_LOSS, GS_CLIPPED, GS_KL_APPROX, GS_SPARE2, GS_GAIN, GS_RATIO, GS_N };

__global__ __launch_bounds__(256) void gaussian_loss_kernel(
    const float *__restrict__ heads, int B, int ldo, int nA, int vh, const float *__restrict__ actions,
    const float *__restrict__ old_log_pac, const float *__restrict__ advantages, const float *__restrict__ returns,
    const float *__restrict__ log_std, float eps_clip, float vf_coef, float grad_scale, float *__restrict__ dheads,
    float *__restrict__ dlog_std_rows, float *__restrict__ stats, const int32_t *__restrict__ index)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    const float *z = heads + (size_t)b * ldo;
    float *dz = dheads + (size_t)b * ldo;
    const int sb = index ? index[b] : b;
    const float adv = advantages[sb];
    float loss_clip = 0.f, vloss = 0.f, clipped = 0.f, ratio_sum = 0.f, klap = 0.f;
    for (int c = lane; c < ldo; c += 64) {
        float g = 0.f;
        if (c < nA) {
            const float ls = log_std[c];
            const float sigma = expf(ls);
            const float act = actions[(size_t)sb * nA + c];
            const float d = act - z[c];
            const float var = sigma * sigma;
            const float logpac = -(d * d) / (2.f * var) - ls - kLogSqrt2Pi;
            const float old = old_log_pac[(size_t)sb * nA + c];
            const float ratio = expf(logpac - old);
            const float cr = fminf(fmaxf(ratio, 1.f - eps_clip), 1.f + eps_clip);
            const float s1 = ratio * adv, s2 = cr * adv;
            const bool inside = ratio >= 1.f - eps_clip && ratio <= 1.f + eps_clip;
            float dclip = inside ? adv : (s1 < s2 ? adv : (s1 == s2 ? 0.5f * adv : 0.f));
            // gain += mean_a min(s1, s2);  d logpac / d mu = d / var;  d logpac / d log_std = d^2 / var - 1
            const float w = dclip * ratio / nA;
            g = -w * (d / var);
            if (dlog_std_rows) dlog_std_rows[(size_t)b * nA + c] = -grad_scale * w * (d * d / var - 1.f);
            loss_clip += fminf(s1, s2) / nA;
            clipped += (fabsf(ratio - 1.f) > eps_clip ? 1.f : 0.f) / nA;
            ratio_sum += ratio / nA;
            klap += (old - logpac) / nA;
        } else if (c < nA + vh) {
            const float diff = z[c] - returns[(size_t)sb * vh + (c - nA)];
            vloss += vf_coef * diff * diff;
            g = 2.f * vf_coef * diff;
        }
        dz[c] = grad_scale * g;
    }
    if (stats) {
        loss_clip = wave_sum(loss_clip);
        vloss = wave_sum(vloss);
        clipped = wave_sum(clipped);
        ratio_sum = wave_sum(ratio_sum);
        klap = wave_sum(klap);
        if (lane == 0) {
            float *s = stats + (size_t)b * GS_N;
            s[GS_LOSS_CLIP] = loss_clip;
            s[GS_SPARE] = 0.f;
            s[GS_VALUE_LOSS] = vloss;
            s[GS_CLIPPED] = clipped;
            s[GS_KL_APPROX] = klap;
            s[GS_SPARE2] = 0.f;
            s[GS_GAIN] = loss_clip - vloss;
            s[GS_RATIO] = ratio_sum;
        }
    }
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
    hipLaunchKernelGGL(value_loss_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(stream), heads, B, ldo, value_col,
                       n_value_heads, returns, vf_coef, tvf_col, n_tvf, n_tvf > 0 ? tvf_stride : 1, tvf_returns,
                       tvf_weights, tvf_coef, grad_scale, dheads, stats, index, tvf_keep_prob, seed, offset);
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
    hipLaunchKernelGGL(distil_loss_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(stream), heads, B, ldo, n_actions,
                       pred_col, n_pred, pred_stride, vector_targets, targets, weights, old_policy, log_std, beta,
                       grad_scale, dheads, stats, index);
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
    hipLaunchKernelGGL(gaussian_loss_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(stream), heads, B, ldo,
                       n_actions, n_value_heads, actions, old_log_pac, advantages, returns, log_std, eps_clip, vf_coef,
                       grad_scale, dheads, dlog_std_rows, stats, index);
    return check_launch("gaussian_loss_kernel");
}

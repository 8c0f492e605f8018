// The per-sample bodies of the head-side loss kernels (heads_loss.hip, policy.hip), shared with the fused MLP training
// kernel (mlp_fused.hip), which runs them on head rows that sit in LDS: one body each, so every launch form gives the
// same bits.  z / dz are the sample's head row and its gradient row (any address space); b is the minibatch row (statistics,
// dropout counters), sb its row in the un-gathered batch arrays.  The wave-per-sample bodies are executed by all 64 lanes of
// one wave; the discrete PPO body by ONE thread.
#pragma once
#include "common.h"
#include "policy_act.h"

namespace ppo {

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

constexpr float kLogSqrt2Pi = 0.91893853320467274178f;

// ---------------------------------------------------------------------------------------------- value phase
enum { VS_VALUE = 0, VS_TVF, VS_TOTAL, VS_SPARE, VS_N };
struct ValueLossP {
    int ldo, value_col, vh;
    const float *returns;
    float vf_coef;
    int tvf_col, K, tvf_stride;
    const float *tvf_returns, *tvf_weights;
    float tvf_coef, grad_scale;
    float *stats;
    float keep_prob;
    uint64_t seed, offset;
};
__device__ __forceinline__ void value_loss_row(const ValueLossP &p, const float *z, float *dz, int b, int sb, int lane)
{
    const int ldo = p.ldo, value_col = p.value_col, vh = p.vh, tvf_col = p.tvf_col, K = p.K, tvf_stride = p.tvf_stride;
    const float *returns = p.returns, *tvf_returns = p.tvf_returns, *tvf_weights = p.tvf_weights;
    const float vf_coef = p.vf_coef, tvf_coef = p.tvf_coef, grad_scale = p.grad_scale, keep_prob = p.keep_prob;
    const uint64_t seed = p.seed, offset = p.offset;
    float *stats = p.stats;
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

// ---------------------------------------------------------------------------------------------- distillation phase
enum { DS_VALUE = 0, DS_POLICY, DS_TOTAL, DS_SQERR, DS_N };
struct DistilLossP {
    int ldo, nA, pred_col, n_pred, pred_stride, vector_targets;
    const float *targets, *weights, *old_policy, *log_std;
    float beta, grad_scale;
    float *stats;
};
__device__ __forceinline__ void distil_loss_row(const DistilLossP &p, const float *z, float *dz, int b, int sb, int lane)
{
    const int ldo = p.ldo, nA = p.nA, pred_col = p.pred_col, n_pred = p.n_pred, pred_stride = p.pred_stride,
              vector_targets = p.vector_targets;
    const float *targets = p.targets, *weights = p.weights, *old_policy = p.old_policy, *log_std = p.log_std;
    const float beta = p.beta, grad_scale = p.grad_scale;
    float *stats = p.stats;
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

// ---------------------------------------------------------------------------------------------- gaussian policy phase
enum { GS_LOSS_CLIP = 0, GS_SPARE, GS_VALUE_LOSS, GS_CLIPPED, GS_KL_APPROX, GS_SPARE2, GS_GAIN, GS_RATIO, GS_N };
struct GaussLossP {
    int ldo, nA, vh;
    const float *actions, *old_log_pac, *advantages, *returns, *log_std;
    float eps_clip, vf_coef, grad_scale;
    float *stats;
};
// dlog_std_row: this sample's [nA] row of log_std gradient terms (nullable)
__device__ __forceinline__ void gaussian_loss_row(const GaussLossP &p, const float *z, float *dz, float *dlog_std_row, int b,
                                                  int sb, int lane)
{
    const int ldo = p.ldo, nA = p.nA, vh = p.vh;
    const float *actions = p.actions, *old_log_pac = p.old_log_pac, *advantages = p.advantages, *returns = p.returns,
                *log_std = p.log_std;
    const float eps_clip = p.eps_clip, vf_coef = p.vf_coef, grad_scale = p.grad_scale;
    float *stats = p.stats;
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
            if (dlog_std_row) dlog_std_row[c] = -grad_scale * w * (d * d / var - 1.f);
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

// ---------------------------------------------------------------------------------------------- discrete PPO policy phase
// statistics row per sample (reduced on demand by the host side, one D2H per iteration)
enum { ST_LOSS_CLIP = 0, ST_ENTROPY, ST_VALUE_LOSS, ST_CLIPPED, ST_KL_APPROX, ST_KL_TRUE, ST_GAIN, ST_RATIO, ST_N };
struct PpoLossP {
    int ldo, nA, vh;
    const int32_t *actions;
    const float *old_log_pac, *old_log_policy, *advantages, *returns;
    float eps_clip, ent_coef, vf_coef, grad_scale;
    float *stats;
};
// one THREAD per sample (`store` true), or - in the dense + heads + loss launch, where the head row sits in a wave's lanes
// and z(i) is a shuffle - every lane of the wave on the same values with one lane storing.  NA: the action count at
// compile time (0 = any).  z(i): head output i of the sample.
template <int NA, class Z>
__device__ __forceinline__ void ppo_loss_row_z(const PpoLossP &q, Z z, float *dz, int b, int sb, bool store)
{
    const int nA = NA ? NA : q.nA;
    constexpr int kUnroll = NA ? 32 : 1;  // full unroll when the count is a constant
    const int ldo = q.ldo, vh = q.vh;
    const int32_t *actions = q.actions;
    const float *old_log_pac = q.old_log_pac, *old_log_policy = q.old_log_policy, *advantages = q.advantages,
                *returns = q.returns;
    const float eps_clip = q.eps_clip, ent_coef = q.ent_coef, vf_coef = q.vf_coef, grad_scale = q.grad_scale;
    float *stats = q.stats;
    const int act = actions[sb];
    const float adv = advantages[sb];
    float lp[NA ? NA : kMaxActions], oldp[NA ? NA : kMaxActions];
    float mx = -INFINITY;
#pragma unroll kUnroll
    for (int a = 0; a < nA; ++a) {
        lp[a] = z(a);
        oldp[a] = old_log_policy ? old_log_policy[(size_t)sb * nA + a] : 0.f;
        mx = fmaxf(mx, lp[a]);
    }
    float se = 0.f;
#pragma unroll kUnroll
    for (int a = 0; a < nA; ++a) se += expf(lp[a] - mx);
    const float lse = mx + logf(se);
    float entropy = 0.f, kl_true = 0.f, logpac = 0.f;
#pragma unroll kUnroll
    for (int a = 0; a < nA; ++a) {
        lp[a] -= lse;
        const float p = expf(lp[a]);
        entropy -= p * lp[a];
        if (old_log_policy) kl_true += p * (lp[a] - oldp[a]);
        logpac = a == act ? lp[a] : logpac;
    }
    const float ratio = expf(logpac - old_log_pac[sb]);
    const float clipped_ratio = fminf(fmaxf(ratio, 1.f - eps_clip), 1.f + eps_clip);
    const float s1 = ratio * adv, s2 = clipped_ratio * adv;
    const float loss_clip = fminf(s1, s2);
    // d loss_clip / d ratio, with torch.min's tie rule (half to each side) and clamp's
    // pass-through inside [1-eps, 1+eps]:  inside -> adv; outside -> adv only if s1 < s2
    const bool inside = ratio >= 1.f - eps_clip && ratio <= 1.f + eps_clip;
    float dclip_dratio;
    if (inside) dclip_dratio = adv;
    else dclip_dratio = s1 < s2 ? adv : (s1 == s2 ? 0.5f * adv : 0.f);

    // value heads: vf_coef * (V - R)^2 per head (rl/rollout.py:1596-1608)
    float vloss = 0.f;
    for (int i = 0; i < vh; ++i) {
        const float diff = z(nA + i) - returns[(size_t)sb * vh + i];
        vloss += vf_coef * diff * diff;
        if (store) dz[nA + i] = grad_scale * 2.f * vf_coef * diff;  // d(-gain)/dV
    }
    if (store)
        for (int i = nA + vh; i < ldo; ++i) dz[i] = 0.f;

    // d(-gain)/dlogit_j = -[ dclip_dratio * ratio * (1{j=act} - p_j) + ent_coef * (-p_j (logp_j + H)) ]
    const float w = dclip_dratio * ratio;
#pragma unroll kUnroll
    for (int a = 0; a < nA; ++a) {
        const float p = expf(lp[a]);
        const float dpg = w * ((a == act ? 1.f : 0.f) - p);
        const float dent = -p * (lp[a] + entropy);
        if (store) dz[a] = -grad_scale * (dpg + ent_coef * dent);
    }
    if (stats && store) {
        float *s = stats + (size_t)b * ST_N;
        s[ST_LOSS_CLIP] = loss_clip;
        s[ST_ENTROPY] = entropy;
        s[ST_VALUE_LOSS] = vloss;
        s[ST_CLIPPED] = fabsf(ratio - 1.f) > eps_clip ? 1.f : 0.f;
        s[ST_KL_APPROX] = old_log_pac[sb] - logpac;
        s[ST_KL_TRUE] = kl_true;
        s[ST_GAIN] = loss_clip + ent_coef * entropy - vloss;
        s[ST_RATIO] = ratio;
    }
}

template <int NA>
__device__ __forceinline__ void ppo_loss_row(const PpoLossP &q, const float *z, float *dz, int b, int sb)
{
    ppo_loss_row_z<NA>(q, [&](int i) { return z[i]; }, dz, b, sb, true);
}

}  // namespace ppo

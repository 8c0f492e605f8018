// Policy-head post-processing for the discrete-action PPO path (gfx950).  All of it is
// tiny per-sample work (n_actions <= 32): one thread per sample, fully fused, no host
// round trip (the reference syncs device->host 5 times per env step, rl/rollout.py:641,
// 809-813).
//
//  * ppo_policy_act_f32   log_softmax of the policy logits (rl/models.py:488), then either
//                         Gumbel-max sampling  argmax(logp - log(-log u))  (rl/utils.py:248-256)
//                         or the greedy argmax (rl/models.py:475-485, temperature 0).
//  * ppo_ppo_loss_f32     clipped-surrogate + entropy + value loss of
//                         Runner.train_policy_minibatch (rl/rollout.py:1640-1660,1682,1744-1753,
//                         1596-1608), forward AND the gradient w.r.t. the head outputs, plus the
//                         per-sample statistics the reference logs.
//
// Head outputs arrive as one row per sample: [ policy logits (n_actions) | value heads (vh) | ... ],
// leading dimension ldo; columns past n_actions + vh get zero gradient (the reference's
// advantage head is evaluated but never enters the loss, rl/models.py:506).
#include "common.h"
#include "policy_act.h"

namespace ppo {
namespace {

// NA: the action count at compile time (0 = any): the per-action loops unroll, the row's loads are issued together and
// the arrays stay in registers (the runtime-count form paid one L2 round trip per action and indexed a scratch array)
template <int NA>
__global__ __launch_bounds__(64) void policy_act_kernel(const float *__restrict__ heads, int B, int ldo, int nA_,
                                                        float temperature, const float *__restrict__ uniform,
                                                        uint64_t seed, uint64_t offset, int greedy, ActOut out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *z = heads + (size_t)b * ldo;
    policy_act_row<NA>([&](int i) { return z[i]; }, b, nA_, temperature, uniform, seed, offset, greedy, out, true);
}

// statistics row per sample (reduced on demand by the host side, one D2H per iteration)
enum { ST_LOSS_CLIP = 0, ST_ENTROPY, ST_VALUE_LOSS, ST_CLIPPED, ST_KL_APPROX, ST_KL_TRUE, ST_GAIN, ST_RATIO, ST_N };

template <int NA>
__global__ __launch_bounds__(64) void ppo_loss_kernel(
    const float *__restrict__ heads, int B, int ldo, int nA_, int vh, const int32_t *__restrict__ actions,
    const float *__restrict__ old_log_pac, const float *__restrict__ old_log_policy,
    const float *__restrict__ advantages, const float *__restrict__ returns, float eps_clip, float ent_coef,
    float vf_coef, float grad_scale, float *__restrict__ dheads, float *__restrict__ stats, const int32_t *__restrict__ index)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int nA = NA ? NA : nA_;
    constexpr int kUnroll = NA ? 32 : 1;  // full unroll when the count is a constant
    const float *z = heads + (size_t)b * ldo;
    const int sb = index ? index[b] : b;  // row of this sample in the (un-gathered) batch arrays
    const int act = actions[sb];
    const float adv = advantages[sb];
    float lp[NA ? NA : kMaxActions], oldp[NA ? NA : kMaxActions];
    float mx = -INFINITY;
#pragma unroll kUnroll
    for (int a = 0; a < nA; ++a) {
        lp[a] = z[a];
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
    float *dz = dheads + (size_t)b * ldo;
    for (int i = 0; i < vh; ++i) {
        const float diff = z[nA + i] - returns[(size_t)sb * vh + i];
        vloss += vf_coef * diff * diff;
        dz[nA + i] = grad_scale * 2.f * vf_coef * diff;  // d(-gain)/dV
    }
    for (int i = nA + vh; i < ldo; ++i) dz[i] = 0.f;

    // d(-gain)/dlogit_j = -[ dclip_dratio * ratio * (1{j=act} - p_j) + ent_coef * (-p_j (logp_j + H)) ]
    const float w = dclip_dratio * ratio;
#pragma unroll kUnroll
    for (int a = 0; a < nA; ++a) {
        const float p = expf(lp[a]);
        const float dpg = w * ((a == act ? 1.f : 0.f) - p);
        const float dent = -p * (lp[a] + entropy);
        dz[a] = -grad_scale * (dpg + ent_coef * dent);
    }
    if (stats) {
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

}  // namespace
}  // namespace ppo

extern "C" int ppo_policy_act_f32(const float *heads, int B, int ldo, int n_actions, float temperature,
                                  const float *uniform, uint64_t seed, uint64_t offset, int greedy, float *log_policy,
                                  int32_t *actions, float *log_pac, float *raw_policy, float *values,
                                  int n_value_heads, void *stream)
{
    using namespace ppo;
    if (B < 0 || n_actions <= 0 || n_actions > kMaxActions || n_value_heads < 0 || ldo < n_actions + n_value_heads)
        return fail(PPO_E_INVALID, "ppo_policy_act_f32: bad shape (B=%d n_actions=%d ldo=%d, max %d actions)", B,
                    n_actions, ldo, kMaxActions);
    if (B == 0) return PPO_OK;
    if (!heads) return fail(PPO_E_INVALID, "ppo_policy_act_f32: null heads");
    if (!(temperature > 0.f)) return fail(PPO_E_INVALID, "ppo_policy_act_f32: temperature must be > 0 (use greedy=1 for argmax)");
    const ActOut out{log_policy, actions, log_pac, raw_policy, values, n_value_heads};
#define PPO_ACT_CASE(NA)                                                                                              \
    case NA:                                                                                                          \
        hipLaunchKernelGGL((policy_act_kernel<NA>), dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), heads, B, ldo, \
                           n_actions, temperature, uniform, seed, offset, greedy, out);                               \
        break;
    switch (n_actions) {  // the action counts of the benchmark suites at compile time, anything else at run time
        PPO_ACT_CASE(4)
        PPO_ACT_CASE(6)
        PPO_ACT_CASE(15)
        PPO_ACT_CASE(18)
        default:
            hipLaunchKernelGGL((policy_act_kernel<0>), dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), heads, B, ldo,
                               n_actions, temperature, uniform, seed, offset, greedy, out);
    }
#undef PPO_ACT_CASE
    return check_launch("policy_act_kernel");
}

extern "C" int ppo_ppo_loss_f32(const float *heads, int B, int ldo, int n_actions, int n_value_heads,
                                const int32_t *actions, const float *old_log_pac, const float *old_log_policy,
                                const float *advantages, const float *returns, float eps_clip, float ent_coef,
                                float vf_coef, float grad_scale, float *dheads, float *stats, const int32_t *index, void *stream)
{
    using namespace ppo;
    if (B < 0 || n_actions <= 0 || n_actions > kMaxActions || n_value_heads < 0 || ldo < n_actions + n_value_heads)
        return fail(PPO_E_INVALID, "ppo_ppo_loss_f32: bad shape");
    if (B == 0) return PPO_OK;
    if (!heads || !actions || !old_log_pac || !advantages || !dheads || (n_value_heads > 0 && !returns))
        return fail(PPO_E_INVALID, "ppo_ppo_loss_f32: null pointer");
#define PPO_LOSS_CASE(NA)                                                                                             \
    case NA:                                                                                                          \
        hipLaunchKernelGGL((ppo_loss_kernel<NA>), dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), heads, B, ldo,   \
                           n_actions, n_value_heads, actions, old_log_pac, old_log_policy, advantages, returns,      \
                           eps_clip, ent_coef, vf_coef, grad_scale, dheads, stats, index);                           \
        break;
    switch (n_actions) {
        PPO_LOSS_CASE(4)
        PPO_LOSS_CASE(6)
        PPO_LOSS_CASE(15)
        PPO_LOSS_CASE(18)
        default:
            hipLaunchKernelGGL((ppo_loss_kernel<0>), dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), heads, B, ldo,
                               n_actions, n_value_heads, actions, old_log_pac, old_log_policy, advantages, returns,
                               eps_clip, ent_coef, vf_coef, grad_scale, dheads, stats, index);
    }
#undef PPO_LOSS_CASE
    return check_launch("ppo_loss_kernel");
}

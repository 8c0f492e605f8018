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
#include "loss_rows.h"
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

// one thread per sample; the body lives in loss_rows.h (shared with the fused MLP training kernel)
template <int NA>
__global__ __launch_bounds__(64) void ppo_loss_kernel(const float *__restrict__ heads, int B, PpoLossP p,
                                                      float *__restrict__ dheads, const int32_t *__restrict__ index)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    ppo_loss_row<NA>(p, heads + (size_t)b * p.ldo, dheads + (size_t)b * p.ldo, b, index ? index[b] : b);
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
    const PpoLossP p{ldo, n_actions, n_value_heads, actions, old_log_pac, old_log_policy, advantages, returns, eps_clip,
                     ent_coef, vf_coef, grad_scale, stats};
#define PPO_LOSS_CASE(NA)                                                                                             \
    case NA:                                                                                                          \
        hipLaunchKernelGGL((ppo_loss_kernel<NA>), dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), heads, B, p,    \
                           dheads, index);                                                                            \
        break;
    switch (n_actions) {
        PPO_LOSS_CASE(4)
        PPO_LOSS_CASE(6)
        PPO_LOSS_CASE(15)
        PPO_LOSS_CASE(18)
        default:
            hipLaunchKernelGGL((ppo_loss_kernel<0>), dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), heads, B, p, dheads,
                               index);
    }
#undef PPO_LOSS_CASE
    return check_launch("ppo_loss_kernel");
}

// One sample's action step, shared by policy_act_kernel (policy.hip: one thread per sample, logits from HBM) and the
// dense + heads + sampling launch (gemm.hip finalize_heads_kernel: one wave per sample, logits in lanes): log-softmax of
// the policy logits (rl/models.py:488), then Gumbel-max sampling argmax(logp - log(-log u)) (rl/utils.py:248-256) or the
// greedy argmax (rl/models.py:475-485).  One body, so both launches give the same bits.
#pragma once
#include "common.h"

namespace ppo {

constexpr int kMaxActions = 32;

// counter-based uniform in (0, 1): 2 rounds of a 64-bit mix (splitmix64 finaliser) of
// (seed, counter); 24 mantissa bits, never 0 or 1 so that log(-log u) is finite.
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t counter)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (counter + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return ((float)(z >> 40) + 0.5f) * (1.0f / 16777216.0f);
}

struct ActOut {
    float *log_policy;   // [B, nA] or null
    int32_t *actions;    // [B] or null
    float *log_pac;      // [B] or null
    float *raw_policy;   // [B, nA] or null
    float *values;       // [B, vh] or null
    int vh;
};

// NA: the action count at compile time (0 = any): the per-action loops unroll and the arrays stay in registers.
// z(i): head output i of sample b (policy logits, then the value heads); `store`: this thread writes the results (the
// wave-per-sample caller runs the body in every lane on the same values and lets one lane store).
template <int NA, class Z>
__device__ __forceinline__ void policy_act_row(Z z, int b, int nA_, float temperature, const float *__restrict__ uniform,
                                               uint64_t seed, uint64_t offset, int greedy, const ActOut &o, bool store)
{
    const int nA = NA ? NA : nA_;
    constexpr int kUnroll = NA ? 32 : 1;  // full unroll when the count is a constant
    if (o.raw_policy && store)
#pragma unroll kUnroll
        for (int a = 0; a < nA; ++a) o.raw_policy[(size_t)b * nA + a] = z(a);
    if (o.values && store)
        for (int i = 0; i < o.vh; ++i) o.values[(size_t)b * o.vh + i] = z(nA + i);
    float logits[NA ? NA : kMaxActions];
    float mx = -INFINITY;
#pragma unroll kUnroll
    for (int a = 0; a < nA; ++a) {
        logits[a] = z(a) / temperature;
        mx = fmaxf(mx, logits[a]);
    }
    float se = 0.f;
#pragma unroll kUnroll
    for (int a = 0; a < nA; ++a) se += expf(logits[a] - mx);
    const float lse = mx + logf(se);
    int best = 0;
    float best_score = -INFINITY, best_lp = 0.f;
    bool any = false;  // no score beat -inf (NaN logits): action 0, as the running-maximum form
#pragma unroll kUnroll
    for (int a = 0; a < nA; ++a) {
        const float lp = logits[a] - lse;
        logits[a] = lp;
        if (o.log_policy && store) o.log_policy[(size_t)b * nA + a] = lp;
        float score;
        if (greedy) {
            score = z(a);  // argmax of the unscaled logits (rl/models.py:479)
        } else {
            const float u = uniform ? uniform[(size_t)b * nA + a] : uniform01(seed, offset + (uint64_t)b * nA + a);
            score = lp - logf(-logf(u));
        }
        if (score > best_score) {  // first maximum wins, as np.argmax / torch.argmax
            best_score = score;
            best = a;
            best_lp = lp;
            any = true;
        }
    }
    if (!any) best_lp = logits[0];
    if (o.actions && store) o.actions[b] = best;
    if (o.log_pac && store) o.log_pac[b] = best_lp;
}

}  // namespace ppo

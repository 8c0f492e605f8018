// Truncated-horizon (TVF) return estimator on gfx950 — replaces the reference's
// rl/returns_truncated.py:623-693 (_calculate_sampled_return_multi_fast) + :558-620 (_n_step_estimate)
// + :142-174 (_interpolate), called from rl/tvf.py:250-262.
//
//   ret[t,a,k] = mean_c ( S_n[t,a] + boot_{k,c}[t,a] * D_n[t,a] )       n = min(sample[k,c], h_k)
//   D_i[t] = prod_{j<i, t+j<N} gamma*(1-done[t+j])     S_n[t] = sum_{i<n, t+i<N} r[t+i] * D_i[t]
//   boot = interp(V[t+n, a, :], h_k - n)  for t < N-n,   interp(V[N, a, :], h_k - (N-t))  otherwise
//
// The host resolves everything that does not depend on (t, a): which n are needed, and for every
// (k, c) and every (k, N-t) the interpolation plan (zero / exact column / two columns and their
// float32 weights) exactly as the reference's searchsorted logic decides it.  The device does
//   kernel 1  one thread per (t, a): the running S and D over i (the reference's dtype rules: float32
//             arrays, the step factor gamma*(1-done) formed in float64), written at the needed n;
//   kernel 2  one thread per (t, a, k), k fastest so the [N, A, K] output is written coalesced: gathers
//             S/D and one or two value-sample columns per sample and accumulates in the reference's
//             operation order (float32, no contraction) => bit-identical to the reference.
// HBM/L2-bound: 4*(N+1)*A*V read + 4*N*A*K written, plus the gathers served by L2.
#include "common.h"

namespace ppo {
namespace {

__global__ __launch_bounds__(256) void tvf_prefix_kernel(const float *__restrict__ rewards,
                                                         const uint8_t *__restrict__ dones, int N, int A, double gamma,
                                                         const int32_t *__restrict__ nd_of_n, int max_n, int ND,
                                                         float *__restrict__ cS, float *__restrict__ cD)
{
#pragma clang fp contract(off)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * A) return;
    const int t = idx / A;
    const int a = idx - t * A;
    float S = 0.f, D = 1.f;
    float *oS = cS + (size_t)idx * ND;
    float *oD = cD + (size_t)idx * ND;
    for (int i = 0; i < max_n; ++i) {
        if (t + i < N) {  // rows past the end of the rollout stop updating (s[:N-i] += ..., :667-669)
            const size_t j = (size_t)(t + i) * A + a;
            const float term = rewards[j] * D;
            S = S + term;
            const double step = dones[j] ? 0.0 : gamma;  // gamma * (1 - bool) is float64
            D = (float)((double)D * step);                // ... rounded back into the float32 array
        }
        const int nd = nd_of_n[i + 1];
        if (nd >= 0) {
            oS[nd] = S;
            oD[nd] = D;
        }
    }
}

// A plan row (mode, i0, i1) + weights (w0, w1) says how to evaluate interp(values[..., :], target):
// mode 0 -> zero, 1 -> values[i0], 2 -> values[i0]*w0 + values[i1]*w1.
// The two-column case is evaluated in float64 and rounded once: in the reference the interpolation factor
// is a NumPy float64 scalar (a quotient of NumPy integers), and under NumPy >= 2 promotion (NEP 50, the NumPy
// the golden vectors were produced with) float32_array * float64_scalar is a float64 array; the float32
// destination rounds it (rl/returns_truncated.py:171-174, 609-614).
__device__ __forceinline__ float apply_plan(const float *__restrict__ row, const int32_t *__restrict__ p,
                                            const double *__restrict__ w)
{
#pragma clang fp contract(off)
    const int mode = p[0];
    if (mode == 0) return 0.f;
    const float v0 = row[p[1]];
    if (mode == 1) return v0;
    const float v1 = row[p[2]];
    const double x0 = (double)v0 * w[0];
    const double x1 = (double)v1 * w[1];
    return (float)(x0 + x1);
}

__global__ __launch_bounds__(256) void tvf_gather_kernel(const float *__restrict__ values, int N, int A, int V, int K,
                                                         int C, int ND, const int32_t *__restrict__ n_eff,
                                                         const int32_t *__restrict__ nd_index,
                                                         const int32_t *__restrict__ main_plan,
                                                         const double *__restrict__ main_w,
                                                         const int32_t *__restrict__ tail_plan,
                                                         const double *__restrict__ tail_w,
                                                         const uint8_t *__restrict__ k_zero, float inv_c,
                                                         const float *__restrict__ cS, const float *__restrict__ cD,
                                                         float *__restrict__ out)
{
#pragma clang fp contract(off)
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * A * K) return;
    const int k = idx % K;
    const int64_t ta = idx / K;
    const int t = ta / A;
    const int a = ta - (int64_t)t * A;
    float total = 0.f;
    if (!k_zero[k]) {
        const float *last_row = values + ((size_t)N * A + a) * V;
        for (int c = 0; c < C; ++c) {
            const int kc = k * C + c;
            const int n = n_eff[kc];
            const int nd = nd_index[kc];
            const float S = cS[(size_t)ta * ND + nd];
            const float D = cD[(size_t)ta * ND + nd];
            float boot;
            if (t < N - n) {
                boot = apply_plan(values + ((size_t)(t + n) * A + a) * V, main_plan + 3 * kc, main_w + 2 * kc);
            } else {
                const int kj = k * (N + 1) + (N - t);
                boot = apply_plan(last_row, tail_plan + 3 * kj, tail_w + 2 * kj);
            }
            const float md = boot * D;
            const float term = S + md;
            total = total + term;
        }
        total = total * inv_c;
    }
    out[idx] = total;
}

}  // namespace
}  // namespace ppo

extern "C" size_t ppo_tvf_returns_workspace_bytes(int N, int A, int ND)
{
    return (size_t)2 * N * A * (ND < 1 ? 1 : ND) * sizeof(float);
}

extern "C" int ppo_tvf_returns_f32(const float *rewards, const uint8_t *dones, const float *value_samples, int N, int A,
                                   int V, int K, int C, double gamma, const int32_t *n_eff, const int32_t *nd_index,
                                   const int32_t *nd_of_n, int max_n, int ND, const int32_t *main_plan,
                                   const double *main_w, const int32_t *tail_plan, const double *tail_w,
                                   const uint8_t *k_zero, void *workspace, size_t workspace_bytes, float *out,
                                   void *stream)
{
    using namespace ppo;
    if (N <= 0 || A <= 0 || V <= 0 || K <= 0 || C <= 0 || max_n < 1 || max_n > N || ND < 1)
        return fail(PPO_E_INVALID, "ppo_tvf_returns_f32: bad shape (N=%d A=%d V=%d K=%d C=%d max_n=%d ND=%d)", N, A, V,
                    K, C, max_n, ND);
    if (!rewards || !dones || !value_samples || !n_eff || !nd_index || !nd_of_n || !main_plan || !main_w || !tail_plan ||
        !tail_w || !k_zero || !workspace || !out)
        return fail(PPO_E_INVALID, "ppo_tvf_returns_f32: null pointer");
    if (workspace_bytes < ppo_tvf_returns_workspace_bytes(N, A, ND))
        return fail(PPO_E_INVALID, "ppo_tvf_returns_f32: workspace too small");
    hipStream_t st = as_stream(stream);
    float *cS = static_cast<float *>(workspace);
    float *cD = cS + (size_t)N * A * ND;
    hipLaunchKernelGGL(tvf_prefix_kernel, dim3((N * A + 255) / 256), dim3(256), 0, st, rewards, dones, N, A, gamma,
                       nd_of_n, max_n, ND, cS, cD);
    int rc = check_launch("tvf_prefix_kernel");
    if (rc) return rc;
    const int64_t total = (int64_t)N * A * K;
    hipLaunchKernelGGL(tvf_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, value_samples, N, A, V,
                       K, C, ND, n_eff, nd_index, main_plan, main_w, tail_plan, tail_w, k_zero, (float)(1.0 / C), cS, cD,
                       out);
    return check_launch("tvf_gather_kernel");
}

// Observation normalisation (gfx950): the running per-feature mean / variance the reference keeps in
// TVFModel.obs_rms and the clamp((x - mu) / (std + eps), -5, 5) transform applied before both networks
// (rl/models.py:661-694, rl/utils.py:379-455).  Opt-in (`--observation_normalization`), HBM-bound.
//
//  * ppo_obs_moments_f64      per-feature sum and sum of squares of one batch of prepared observations
//                             (uint8 -> x/255 as in prep_for_model, float32 as is), accumulated in float64.
//                             One thread per feature, serial over the batch: loads are coalesced across
//                             features and the result is deterministic.  The two sums are additive, so
//                             data-parallel ranks all-reduce them before the update.
//  * ppo_obs_rms_update_f64   the parallel-variance update of rl/utils.py:379-394 on the device-resident
//                             float64 mean / var, and the float32 constants mu = mean, std = sqrt(var) the
//                             transform reads (rl/models.py:661-663).  `count` stays on the host (it is
//                             count0 + the number of observations seen, no device round trip needed).
//  * ppo_obs_normalize_f32    out = clamp((x - mu) / (std + eps), -5, 5), float32 out; bit-identical to the
//                             torch expression (IEEE subtract / add / divide, no contraction possible).
//
// The reference reduces each batch with float32 torch.mean / torch.var and only then promotes to float64;
// here the batch is reduced in float64 directly, so the running statistics agree to float32 rounding of one
// batch mean (~1e-7 relative), not bit for bit; the transform given equal constants is bit-exact.
#include "common.h"

namespace ppo {
namespace {

__device__ __forceinline__ float prep(const void *x, size_t i, int is_u8)
{
    return is_u8 ? (float)static_cast<const uint8_t *>(x)[i] / 255.0f : static_cast<const float *>(x)[i];
}

__global__ __launch_bounds__(256) void obs_moments_kernel(const void *__restrict__ x, int is_u8, int B, int F,
                                                          double *__restrict__ moments)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    double s = 0.0, q = 0.0;
    int b = 0;
    for (; b + 4 <= B; b += 4) {  // four independent loads in flight
        const float v0 = prep(x, (size_t)b * F + f, is_u8), v1 = prep(x, (size_t)(b + 1) * F + f, is_u8);
        const float v2 = prep(x, (size_t)(b + 2) * F + f, is_u8), v3 = prep(x, (size_t)(b + 3) * F + f, is_u8);
        s += (double)v0; q += (double)v0 * (double)v0;
        s += (double)v1; q += (double)v1 * (double)v1;
        s += (double)v2; q += (double)v2 * (double)v2;
        s += (double)v3; q += (double)v3 * (double)v3;
    }
    for (; b < B; ++b) {
        const float v = prep(x, (size_t)b * F + f, is_u8);
        s += (double)v; q += (double)v * (double)v;
    }
    moments[f] = s;
    moments[F + f] = q;
}

__global__ __launch_bounds__(256) void obs_rms_update_kernel(const double *__restrict__ moments, double batch_count,
                                                             double count, double *__restrict__ mean,
                                                             double *__restrict__ var, float *__restrict__ mu,
                                                             float *__restrict__ std, int F)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const double bm = moments[f] / batch_count;
    const double bv = fmax(moments[F + f] / batch_count - bm * bm, 0.0);
    const double m = mean[f], v = var[f];
    const double delta = bm - m, tot = count + batch_count;
    const double new_mean = m + delta * batch_count / tot;
    const double m2 = v * count + bv * batch_count + delta * delta * count * batch_count / tot;
    const double new_var = m2 / tot;
    mean[f] = new_mean;
    var[f] = new_var;
    mu[f] = (float)new_mean;
    std[f] = sqrtf((float)new_var);
}

__device__ __forceinline__ float normalise(float v, float m, float s, float eps)
{
    return fminf(fmaxf((v - m) / (s + eps), -5.0f), 5.0f);
}

// VEC = 4 when F % 4 == 0 (rows then keep 16-byte alignment), else 1.
template <int VEC, bool U8>
__global__ __launch_bounds__(256) void obs_normalize_kernel(const void *__restrict__ x, const float *__restrict__ mu,
                                                            const float *__restrict__ std, float eps,
                                                            float *__restrict__ out, size_t n_vec, int fv)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(i % (size_t)fv);
        if constexpr (VEC == 4) {
            float4 v;
            if constexpr (U8) {
                const uchar4 r = static_cast<const uchar4 *>(x)[i];
                v = make_float4((float)r.x / 255.0f, (float)r.y / 255.0f, (float)r.z / 255.0f, (float)r.w / 255.0f);
            } else {
                v = static_cast<const float4 *>(x)[i];
            }
            const float4 m = reinterpret_cast<const float4 *>(mu)[f], s = reinterpret_cast<const float4 *>(std)[f];
            reinterpret_cast<float4 *>(out)[i] = make_float4(normalise(v.x, m.x, s.x, eps), normalise(v.y, m.y, s.y, eps),
                                                             normalise(v.z, m.z, s.z, eps), normalise(v.w, m.w, s.w, eps));
        } else {
            out[i] = normalise(prep(x, i, U8), mu[f], std[f], eps);
        }
    }
}

inline int grid_for(size_t n) { return (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }

}  // namespace
}  // namespace ppo

extern "C" int ppo_obs_moments_f64(const void *x, int is_u8, int B, int F, double *moments, void *stream)
{
    using namespace ppo;
    if (B <= 0 || F <= 0) return fail(PPO_E_INVALID, "ppo_obs_moments_f64: bad shape [%d, %d]", B, F);
    if (!x || !moments) return fail(PPO_E_INVALID, "ppo_obs_moments_f64: null pointer");
    hipLaunchKernelGGL(obs_moments_kernel, dim3((F + 255) / 256), dim3(256), 0, as_stream(stream), x, is_u8 ? 1 : 0, B, F,
                       moments);
    return check_launch("obs_moments_kernel");
}

extern "C" int ppo_obs_rms_update_f64(const double *moments, double batch_count, double count, double *mean, double *var,
                                      float *mu, float *std, int F, void *stream)
{
    using namespace ppo;
    if (F <= 0 || !(batch_count > 0.0) || !(count >= 0.0))
        return fail(PPO_E_INVALID, "ppo_obs_rms_update_f64: bad shape / counts (F %d, batch %g, count %g)", F, batch_count,
                    count);
    if (!moments || !mean || !var || !mu || !std) return fail(PPO_E_INVALID, "ppo_obs_rms_update_f64: null pointer");
    hipLaunchKernelGGL(obs_rms_update_kernel, dim3((F + 255) / 256), dim3(256), 0, as_stream(stream), moments, batch_count,
                       count, mean, var, mu, std, F);
    return check_launch("obs_rms_update_kernel");
}

extern "C" int ppo_obs_normalize_f32(const void *x, int is_u8, const float *mu, const float *std, float eps, float *out,
                                     int B, int F, void *stream)
{
    using namespace ppo;
    if (B < 0 || F <= 0) return fail(PPO_E_INVALID, "ppo_obs_normalize_f32: bad shape [%d, %d]", B, F);
    if (B == 0) return PPO_OK;
    if (!x || !mu || !std || !out) return fail(PPO_E_INVALID, "ppo_obs_normalize_f32: null pointer");
    const size_t n = (size_t)B * F;
    const bool vec = F % 4 == 0 && aligned(x, is_u8 ? 4 : 16) && aligned(out, 16) && aligned(mu, 16) && aligned(std, 16);
    hipStream_t st = as_stream(stream);
    if (vec) {
        if (is_u8)
            hipLaunchKernelGGL((obs_normalize_kernel<4, true>), dim3(grid_for(n / 4)), dim3(256), 0, st, x, mu, std, eps, out,
                               n / 4, F / 4);
        else
            hipLaunchKernelGGL((obs_normalize_kernel<4, false>), dim3(grid_for(n / 4)), dim3(256), 0, st, x, mu, std, eps, out,
                               n / 4, F / 4);
    } else {
        if (is_u8)
            hipLaunchKernelGGL((obs_normalize_kernel<1, true>), dim3(grid_for(n)), dim3(256), 0, st, x, mu, std, eps, out, n, F);
        else
            hipLaunchKernelGGL((obs_normalize_kernel<1, false>), dim3(grid_for(n)), dim3(256), 0, st, x, mu, std, eps, out, n, F);
    }
    return check_launch("obs_normalize_kernel");
}

// Small batch-level device ops of the PPO update (gfx950), all HBM-bound:
//
//  * ppo_gather_rows       minibatch gather of observation rows (and any other per-sample
//                          row) by a permutation index (the reference fancy-indexes on the host and
//                          uploads per micro-batch, rl/rollout.py:2349-2372)
//  * ppo_moments_f64       sum / sum of squares / count of a float32 array in float64, as a fixed-order
//                          two-stage reduction (deterministic); the three numbers stay on the device so a
//                          data-parallel run can all-reduce them before normalising
//  * ppo_normalize_f32     (a - mean) / (std + eps) with population std, the batch-level advantage
//                          normalisation of Runner.train_policy (rl/rollout.py:1887-1900)
#include "common.h"

namespace ppo {
namespace {

// one workgroup per destination row; 16-byte copies when the row allows it
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint8_t *__restrict__ src, int64_t row_bytes,
                                                          const int32_t *__restrict__ index, int n_rows,
                                                          int64_t n_src_rows, uint8_t *__restrict__ dst)
{
    const int r = blockIdx.x;
    if (r >= n_rows) return;
    int64_t s = index[r];
    if (s < 0 || s >= n_src_rows) s = 0;  // never read out of bounds; host validates indices
    const uint8_t *sp = src + s * row_bytes;
    uint8_t *dp = dst + (int64_t)r * row_bytes;
    if ((row_bytes & 15) == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0) {
        const int64_t n16 = row_bytes >> 4;
        const uint4 *s4 = reinterpret_cast<const uint4 *>(sp);
        uint4 *d4 = reinterpret_cast<uint4 *>(dp);
        // eight 16-byte loads in flight per thread before the first store: the rolled copy paid one HBM round trip per
        // 4 KB of a row (a 28 KB observation row = 7 of them in a row, 9.5 us for 256 rows)
        for (int64_t i0 = threadIdx.x; i0 < n16; i0 += 256 * 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t i = i0 + 256 * u;
                v[u] = s4[i < n16 ? i : i0];  // clamped address, unconditional load
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t i = i0 + 256 * u;
                if (i < n16) d4[i] = v[u];
            }
        }
    } else {
        for (int64_t i = threadIdx.x; i < row_bytes; i += 256) dp[i] = sp[i];
    }
}

constexpr int kMomentBlocks = 128;

__global__ __launch_bounds__(256) void moments_partial_kernel(const float *__restrict__ x, int64_t n,
                                                              double *__restrict__ partials)
{
    __shared__ double s0[256], s1[256];
    const int64_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * chunk;
    const int64_t hi = lo + chunk < n ? lo + chunk : n;
    double a = 0.0, b = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const double v = (double)x[i];
        a += v;
        b += v * v;
    }
    s0[threadIdx.x] = a;
    s1[threadIdx.x] = b;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) {
            s0[threadIdx.x] += s0[threadIdx.x + w];
            s1[threadIdx.x] += s1[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = s0[0];
        partials[2 * blockIdx.x + 1] = s1[0];
    }
}

__global__ void moments_final_kernel(const double *__restrict__ partials, int n_partials, int64_t n,
                                     double *__restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < n_partials; ++i) {
            a += partials[2 * i];
            b += partials[2 * i + 1];
        }
        out[0] = a;
        out[1] = b;
        out[2] = (double)n;
    }
}

// moments: [sum, sumsq, count] (possibly all-reduced over ranks)
__global__ __launch_bounds__(256) void normalize_kernel(const float *__restrict__ x, int64_t n,
                                                        const double *__restrict__ moments, float eps,
                                                        float *__restrict__ out, float *__restrict__ mean_std_out)
{
    const double cnt = moments[2];
    const double mean = moments[0] / cnt;
    double var = moments[1] / cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float mean32 = (float)mean;
    const float denom = (float)sqrt(var) + eps;  // a.std() is an f32 scalar in the reference
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (x[i] - mean32) / denom;
    if (i == 0 && mean_std_out) {
        mean_std_out[0] = mean32;
        mean_std_out[1] = (float)sqrt(var);
    }
}

// dst += src, four floats per thread where the pair is 16-byte aligned (micro-batch gradient accumulation)
// w[i] *= mask[i] (mask is 0 / 1): the reference's `tvf_head.weight.data *= tvf_features_mask` (rl/models.py:425-427)
__global__ __launch_bounds__(256) void mask_mul_kernel(float *__restrict__ w, const uint8_t *__restrict__ mask, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = w[i] * (float)mask[i];
}

__global__ __launch_bounds__(256) void accumulate_kernel(float *__restrict__ dst, const float *__restrict__ src, int64_t n,
                                                         int vec)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        if (4 * i + 3 < n) {
            float4 d = reinterpret_cast<float4 *>(dst)[i];
            const float4 v = reinterpret_cast<const float4 *>(src)[i];
            d.x += v.x, d.y += v.y, d.z += v.z, d.w += v.w;
            reinterpret_cast<float4 *>(dst)[i] = d;
        } else {
            for (int64_t k = 4 * i; k < n; ++k) dst[k] += src[k];
        }
    } else if (i < n) {
        dst[i] += src[i];
    }
}

}  // namespace
}  // namespace ppo

extern "C" int ppo_gather_rows(const void *src, int64_t row_bytes, int64_t n_src_rows, const int32_t *index, int n_rows,
                               void *dst, void *stream)
{
    using namespace ppo;
    if (row_bytes <= 0 || n_rows < 0 || n_src_rows <= 0) return fail(PPO_E_INVALID, "ppo_gather_rows: bad shape");
    if (n_rows == 0) return PPO_OK;
    if (!src || !index || !dst) return fail(PPO_E_INVALID, "ppo_gather_rows: null pointer");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n_rows), dim3(256), 0, as_stream(stream),
                       static_cast<const uint8_t *>(src), row_bytes, index, n_rows, n_src_rows,
                       static_cast<uint8_t *>(dst));
    return check_launch("gather_rows_kernel");
}

extern "C" size_t ppo_moments_workspace_bytes(void) { return 2 * ppo::kMomentBlocks * sizeof(double); }

extern "C" int ppo_moments_f64(const float *x, int64_t n, double *moments, void *workspace, void *stream)
{
    using namespace ppo;
    if (n <= 0 || !x || !moments || !workspace) return fail(PPO_E_INVALID, "ppo_moments_f64: bad arguments");
    hipStream_t st = as_stream(stream);
    int nb = (int)((n + 4095) / 4096);
    nb = nb > kMomentBlocks ? kMomentBlocks : nb;
    double *partials = static_cast<double *>(workspace);
    hipLaunchKernelGGL(moments_partial_kernel, dim3(nb), dim3(256), 0, st, x, n, partials);
    int rc = check_launch("moments_partial_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(moments_final_kernel, dim3(1), dim3(64), 0, st, partials, nb, n, moments);
    return check_launch("moments_final_kernel");
}

extern "C" int ppo_normalize_f32(const float *x, int64_t n, const double *moments, float eps, float *out,
                                 float *mean_std_out, void *stream)
{
    using namespace ppo;
    if (n <= 0 || !x || !moments || !out) return fail(PPO_E_INVALID, "ppo_normalize_f32: bad arguments");
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, n,
                       moments, eps, out, mean_std_out);
    return check_launch("normalize_kernel");
}

extern "C" int ppo_mask_mul_f32(float *w, const uint8_t *mask, int64_t n, void *stream)
{
    using namespace ppo;
    if (n < 0) return fail(PPO_E_INVALID, "ppo_mask_mul_f32: n < 0");
    if (n == 0) return PPO_OK;
    if (!w || !mask) return fail(PPO_E_INVALID, "ppo_mask_mul_f32: null pointer");
    hipLaunchKernelGGL(mask_mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), w, mask, n);
    return check_launch("mask_mul_kernel");
}

extern "C" int ppo_accumulate_f32(float *dst, const float *src, int64_t n, void *stream)
{
    using namespace ppo;
    if (n < 0) return fail(PPO_E_INVALID, "ppo_accumulate_f32: n < 0");
    if (n == 0) return PPO_OK;
    if (!dst || !src) return fail(PPO_E_INVALID, "ppo_accumulate_f32: null pointer");
    const int vec = aligned(dst, 16) && aligned(src, 16);
    const int64_t threads = vec ? (n + 3) / 4 : n;
    hipLaunchKernelGGL(accumulate_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, as_stream(stream), dst, src,
                       n, vec);
    return check_launch("accumulate_kernel");
}

// GAE / lambda-return / bootstrapped-return scans over a time-major rollout
// ([N, A], env index contiguous) for gfx950.
//
// Replaces rl/returns.py:7-67 of the reference (NumPy, reverse python loop over
// t).  The recurrence is first order, A_t = delta_t + c_t * A_{t+1}, so it is a
// scan over time of affine maps; columns (envs) never interact.
//
// Two regimes, both HBM-bound streaming kernels (no MFMA: there is no
// contraction here):
//
//  * columns (wide batches): one thread owns 4 adjacent env columns and walks
//    time serially from t=N-1 to 0 with 16-B loads/stores (a wave reads 1 KiB
//    contiguous per row).  The loads do not depend on the carry, so U rows are
//    issued ahead of the dependent arithmetic.  Same operation order and the
//    same precision as the reference => bit-identical results.
//
//  * tiles (narrow batches, where A/4 threads cannot fill 256 CUs): a 1024-thread
//    workgroup owns CB columns and splits the time axis into 1024/CB segments;
//    every thread reduces its segment to an affine map (C, D) in float64, the
//    maps are composed through LDS, and a second pass (served by L1/L2) replays
//    the segment with the incoming carry.  Float64 throughout: results agree
//    with the reference's float64-carry path to ~1e-15 relative before the f32
//    store, and with its float32-carry paths to f32 rounding noise.
//
// Arithmetic order/precision follows NumPy's promotion in the reference; see
// include/ppo_amd.h (PPO_TERM_*) and oracle/returns_oracle.c.
#include "common.h"

namespace ppo {
namespace {

// ---------------------------------------------------------------------------
// One step of the reference recurrence, in the reference's operation order.
// ---------------------------------------------------------------------------
template <int TERM>
struct Rec;

template <>
struct Rec<PPO_TERM_U8> {  // bool terminals: `1.0 - bool` is float64 in NumPy
    using carry_t = double;
    using term_t = uint8_t;
    struct Coef {
        double gl_a, gl_r;
    };
    static __device__ __forceinline__ Coef coef(double gl_a, double gl_r) { return {gl_a, gl_r}; }
    static __device__ __forceinline__ void step(float r, float v, float vnext, uint8_t d, float gamma32,
                                                const Coef &k, double &pa, double &pr)
    {
#pragma clang fp contract(off)
        const float gv = gamma32 * vnext;  // f32 product (python float * f32 array)
        const double m = d ? 0.0 : 1.0;    // 1.0 - bool -> f64
        double delta = (double)gv * m;
        delta = (double)r + delta;
        delta = delta - (double)v;
        double ca = k.gl_a * m;
        ca = ca * pa;
        pa = delta + ca;
        double cr = k.gl_r * m;
        cr = cr * pr;
        pr = delta + cr;
    }
};

template <>
struct Rec<PPO_TERM_F32> {  // float32 terminals: everything stays float32
    using carry_t = float;
    using term_t = float;
    struct Coef {
        float gl_a, gl_r;
    };
    static __device__ __forceinline__ Coef coef(double gl_a, double gl_r) { return {(float)gl_a, (float)gl_r}; }
    static __device__ __forceinline__ void step(float r, float v, float vnext, float d, float gamma32,
                                                const Coef &k, float &pa, float &pr)
    {
#pragma clang fp contract(off)
        const float m = 1.0f - d;
        float gv = gamma32 * vnext;
        gv = gv * m;
        float delta = r + gv;
        delta = delta - v;
        float ca = k.gl_a * m;
        ca = ca * pa;
        pa = delta + ca;
        float cr = k.gl_r * m;
        cr = cr * pr;
        pr = delta + cr;
    }
};

template <>
struct Rec<PPO_TERM_NONE> {  // terminals=None: `1.0 - False` is the python float 1.0
    using carry_t = float;
    using term_t = uint8_t;  // unused
    struct Coef {
        float gl_a, gl_r;
    };
    static __device__ __forceinline__ Coef coef(double gl_a, double gl_r) { return {(float)gl_a, (float)gl_r}; }
    static __device__ __forceinline__ void step(float r, float v, float vnext, uint8_t, float gamma32,
                                                const Coef &k, float &pa, float &pr)
    {
#pragma clang fp contract(off)
        const float gv = gamma32 * vnext;
        float delta = r + gv;
        delta = delta - v;
        const float ca = k.gl_a * pa;
        pa = delta + ca;
        const float cr = k.gl_r * pr;
        pr = delta + cr;
    }
};

// VEC-wide packets (VEC = 4: 16-B global accesses; VEC = 1: ragged / unaligned A)
template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) Pack {
    T x[VEC];
};

template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> load_pack(const T *p)
{
    return *reinterpret_cast<const Pack<T, VEC> *>(p);
}
template <typename T, int VEC>
__device__ __forceinline__ void store_pack(T *p, const Pack<T, VEC> &v)
{
    *reinterpret_cast<Pack<T, VEC> *>(p) = v;
}

// ---------------------------------------------------------------------------
// columns regime
// ---------------------------------------------------------------------------
template <int TERM, int VEC, int U>
__global__ __launch_bounds__(256) void gae_columns_kernel(
    const float *__restrict__ rewards, const float *__restrict__ values,
    const float *__restrict__ final_value, const void *__restrict__ terminals,
    float *__restrict__ adv_out, float *__restrict__ ret_out, int N, int n_packs, int64_t ld,
    float gamma32, double gl_a, double gl_r)
{
#pragma clang fp contract(off)
    using R = Rec<TERM>;
    using carry_t = typename R::carry_t;
    using term_t = typename R::term_t;
    const int pack = blockIdx.x * blockDim.x + threadIdx.x;
    if (pack >= n_packs) return;
    const int64_t col = (int64_t)pack * VEC;
    const term_t *__restrict__ term = static_cast<const term_t *>(terminals);
    const typename R::Coef k = R::coef(gl_a, gl_r);

    carry_t pa[VEC], pr[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) pa[j] = pr[j] = 0;
    Pack<float, VEC> vnext = load_pack<float, VEC>(final_value + col);

    for (int t0 = N - 1; t0 >= 0; t0 -= U) {
        Pack<float, VEC> rr[U], vv[U];
        Pack<term_t, VEC> dd[U];
        // issue U rows of independent loads before the dependent chain
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 - u;
            if (t >= 0) {
                const int64_t i = (int64_t)t * ld + col;
                rr[u] = load_pack<float, VEC>(rewards + i);
                vv[u] = load_pack<float, VEC>(values + i);
                if (TERM != PPO_TERM_NONE) dd[u] = load_pack<term_t, VEC>(term + i);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 - u;
            if (t >= 0) {
                const int64_t i = (int64_t)t * ld + col;
                Pack<float, VEC> oa, orr;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    R::step(rr[u].x[j], vv[u].x[j], vnext.x[j],
                            TERM != PPO_TERM_NONE ? dd[u].x[j] : term_t(0), gamma32, k, pa[j], pr[j]);
                    oa.x[j] = (float)pa[j];
                    // td_lambda: f32(advantage) + value, an f32 add (rl/returns.py:66-67)
                    orr.x[j] = (float)pr[j] + vv[u].x[j];
                }
                if (adv_out) store_pack<float, VEC>(adv_out + i, oa);
                if (ret_out) store_pack<float, VEC>(ret_out + i, orr);
                vnext = vv[u];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// tiles regime: CB columns x (1024 / CB) time segments per workgroup
// ---------------------------------------------------------------------------
template <int TERM>
__device__ __forceinline__ void tile_step(float r, float v, float vnext, const void *term, int64_t i,
                                          float gamma32, double gl_a, double gl_r, double &delta,
                                          double &ca, double &cr)
{
#pragma clang fp contract(off)
    double m = 1.0;
    if (TERM == PPO_TERM_U8) m = static_cast<const uint8_t *>(term)[i] ? 0.0 : 1.0;
    if (TERM == PPO_TERM_F32) m = 1.0 - (double)static_cast<const float *>(term)[i];
    const float gv = gamma32 * vnext;
    delta = (double)gv * m;
    delta = (double)r + delta;
    delta = delta - (double)v;
    ca = gl_a * m;
    cr = gl_r * m;
}

template <int TERM, int CB>
__global__ __launch_bounds__(1024) void gae_tiles_kernel(
    const float *__restrict__ rewards, const float *__restrict__ values,
    const float *__restrict__ final_value, const void *__restrict__ terminals,
    float *__restrict__ adv_out, float *__restrict__ ret_out, int N, int A, int64_t ld,
    float gamma32, double gl_a, double gl_r)
{
#pragma clang fp contract(off)
    constexpr int S = 1024 / CB;  // time segments
    // affine maps of each (segment, column): X_start = D + C * X_end, for both chains
    __shared__ double sC[2][S][CB];
    __shared__ double sD[2][S][CB];

    const int tid = threadIdx.x;
    const int cl = tid % CB;
    const int seg = tid / CB;
    const int col = blockIdx.x * CB + cl;
    const int L = (N + S - 1) / S;
    const int t_lo = seg * L;
    const int t_hi = min(N, t_lo + L);  // exclusive
    const bool live = col < A && t_lo < t_hi;

    // pass 1: reduce the segment to (C, D)
    double Ca = 1.0, Da = 0.0, Cr = 1.0, Dr = 0.0;
    if (live) {
        float vnext = (t_hi == N) ? final_value[col] : values[(int64_t)t_hi * ld + col];
        for (int t = t_hi - 1; t >= t_lo; --t) {
            const int64_t i = (int64_t)t * ld + col;
            const float r = rewards[i];
            const float v = values[i];
            double delta, ca, cr;
            tile_step<TERM>(r, v, vnext, terminals, i, gamma32, gl_a, gl_r, delta, ca, cr);
            Da = delta + ca * Da;
            Ca = ca * Ca;
            Dr = delta + cr * Dr;
            Cr = cr * Cr;
            vnext = v;
        }
    }
    sC[0][seg][cl] = Ca;
    sD[0][seg][cl] = Da;
    sC[1][seg][cl] = Cr;
    sD[1][seg][cl] = Dr;
    __syncthreads();

    // carry into each segment: compose the maps of all later segments, latest first.
    // 2*CB threads, one per (chain, column); the LDS reads are carry-independent.
    if (tid < 2 * CB) {
        const int chain = tid / CB;
        const int c = tid % CB;
        double x = 0.0;  // value of the chain at t = N
        for (int s = S - 1; s >= 0; --s) {
            const double C = sC[chain][s][c];
            const double D = sD[chain][s][c];
            sD[chain][s][c] = x;  // carry entering segment s from above
            x = D + C * x;
        }
    }
    __syncthreads();

    // pass 2: replay the segment with its incoming carry (inputs come from L1/L2)
    if (live) {
        double pa = sD[0][seg][cl];
        double pr = sD[1][seg][cl];
        float vnext = (t_hi == N) ? final_value[col] : values[(int64_t)t_hi * ld + col];
        for (int t = t_hi - 1; t >= t_lo; --t) {
            const int64_t i = (int64_t)t * ld + col;
            const float r = rewards[i];
            const float v = values[i];
            double delta, ca, cr;
            tile_step<TERM>(r, v, vnext, terminals, i, gamma32, gl_a, gl_r, delta, ca, cr);
            pa = delta + ca * pa;
            pr = delta + cr * pr;
            if (adv_out) adv_out[i] = (float)pa;
            if (ret_out) ret_out[i] = (float)pr + v;
            vnext = v;
        }
    }
}

// ---------------------------------------------------------------------------
// bootstrapped returns (rl/returns.py:32-55): column-serial, same structure.
// ---------------------------------------------------------------------------
template <int DONE, int VEC, int U>
__global__ __launch_bounds__(256) void bootstrapped_columns_kernel(
    const float *__restrict__ rewards, const void *__restrict__ dones,
    const float *__restrict__ final_value, const float *__restrict__ gamma_arr, float gamma32,
    float *__restrict__ out, int N, int n_packs, int64_t ld)
{
#pragma clang fp contract(off)
    using term_t = typename Rec<DONE>::term_t;
    const int pack = blockIdx.x * blockDim.x + threadIdx.x;
    if (pack >= n_packs) return;
    const int64_t col = (int64_t)pack * VEC;
    const term_t *__restrict__ done = static_cast<const term_t *>(dones);

    // bool dones: the carry becomes float64 after the first step
    // (f32 carry * f32 gamma in f32, then * float64 mask): oracle/returns_oracle.c
    double cur64[VEC];
    float cur32[VEC];
    const Pack<float, VEC> vf = load_pack<float, VEC>(final_value + col);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        cur32[j] = vf.x[j];
        cur64[j] = 0.0;
    }

    for (int t0 = N - 1; t0 >= 0; t0 -= U) {
        Pack<float, VEC> rr[U], gg[U];
        Pack<term_t, VEC> dd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 - u;
            if (t >= 0) {
                const int64_t i = (int64_t)t * ld + col;
                rr[u] = load_pack<float, VEC>(rewards + i);
                dd[u] = load_pack<term_t, VEC>(done + i);
                if (gamma_arr) gg[u] = load_pack<float, VEC>(gamma_arr + i);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 - u;
            if (t >= 0) {
                const int64_t i = (int64_t)t * ld + col;
                Pack<float, VEC> o;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float g = gamma_arr ? gg[u].x[j] : gamma32;
                    if (DONE == PPO_TERM_U8) {
                        double x = (t == N - 1) ? (double)(cur32[j] * g) : cur64[j] * (double)g;
                        const double m = dd[u].x[j] ? 0.0 : 1.0;
                        x = x * m;
                        cur64[j] = (double)rr[u].x[j] + x;
                        o.x[j] = (float)cur64[j];
                    } else {
                        float x = cur32[j] * g;
                        x = x * (1.0f - (float)dd[u].x[j]);
                        cur32[j] = rr[u].x[j] + x;
                        o.x[j] = cur32[j];
                    }
                }
                store_pack<float, VEC>(out + i, o);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------
constexpr int kColumnsU = 8;

template <int TERM>
int launch_columns(const float *r, const float *v, const float *vf, const void *term, float *adv,
                   float *ret, int N, int A, int64_t ld, float g32, double gla, double glr,
                   hipStream_t st)
{
    const size_t tsz = TERM == PPO_TERM_F32 ? 4 : 1;
    const bool vec4 = (A % 4 == 0) && (ld % 4 == 0) && aligned(r, 16) && aligned(v, 16) &&
                      aligned(vf, 16) && (!adv || aligned(adv, 16)) && (!ret || aligned(ret, 16)) &&
                      (TERM == PPO_TERM_NONE || aligned(term, 4 * tsz));
    if (vec4) {
        const int packs = A / 4;
        // one wave per workgroup: waves drift apart instead of marching through the rows in
        // lockstep, which measured 5-10 % faster than 256-thread groups (tools/scan_tune.hip)
        const int block = 64;
        const int grid = (packs + block - 1) / block;
        hipLaunchKernelGGL((gae_columns_kernel<TERM, 4, kColumnsU>), dim3(grid), dim3(block), 0, st, r,
                           v, vf, term, adv, ret, N, packs, ld, g32, gla, glr);
    } else {
        const int block = 64;
        const int grid = (A + block - 1) / block;
        hipLaunchKernelGGL((gae_columns_kernel<TERM, 1, kColumnsU>), dim3(grid), dim3(block), 0, st, r,
                           v, vf, term, adv, ret, N, A, ld, g32, gla, glr);
    }
    return check_launch("gae_columns_kernel");
}

template <int TERM, int CB>
int launch_tiles_cb(const float *r, const float *v, const float *vf, const void *term, float *adv,
                    float *ret, int N, int A, int64_t ld, float g32, double gla, double glr,
                    hipStream_t st)
{
    const int grid = (A + CB - 1) / CB;
    hipLaunchKernelGGL((gae_tiles_kernel<TERM, CB>), dim3(grid), dim3(1024), 0, st, r, v, vf, term, adv,
                       ret, N, A, ld, g32, gla, glr);
    return check_launch("gae_tiles_kernel");
}

template <int TERM>
int launch_tiles(const float *r, const float *v, const float *vf, const void *term, float *adv,
                 float *ret, int N, int A, int64_t ld, float g32, double gla, double glr,
                 hipStream_t st)
{
    // keep the grid at or above ~256 workgroups where A allows, 64-B row chunks at least
    if (A <= 4096) return launch_tiles_cb<TERM, 16>(r, v, vf, term, adv, ret, N, A, ld, g32, gla, glr, st);
    if (A <= 8192) return launch_tiles_cb<TERM, 32>(r, v, vf, term, adv, ret, N, A, ld, g32, gla, glr, st);
    return launch_tiles_cb<TERM, 64>(r, v, vf, term, adv, ret, N, A, ld, g32, gla, glr, st);
}

// up to this many columns the tiles regime is faster (measured, tools/scan_sweep.py): the columns
// regime needs A/256 >= ~1024 waves to cover the HBM latency-bandwidth product of 256 CUs
constexpr int kTilesMaxA = 65536;

}  // namespace
}  // namespace ppo

extern "C" int ppo_gae_scan_f32(const float *rewards, const float *values, const float *final_value,
                                const void *terminals, int terminal_kind, float *adv_out,
                                float *ret_out, int N, int A, int64_t ld, double gamma, double lam_adv,
                                double lam_ret, int regime, void *stream)
{
    using namespace ppo;
    if (N < 0 || A < 0 || ld < A) return fail(PPO_E_INVALID, "ppo_gae_scan_f32: bad shape N=%d A=%d ld=%lld", N, A, (long long)ld);
    if (N == 0 || A == 0) return PPO_OK;
    if (!rewards || !values || !final_value) return fail(PPO_E_INVALID, "ppo_gae_scan_f32: null input");
    if (!adv_out && !ret_out) return fail(PPO_E_INVALID, "ppo_gae_scan_f32: no output requested");
    if (terminal_kind == PPO_TERM_NONE) terminals = nullptr;
    else if (terminal_kind == PPO_TERM_U8 || terminal_kind == PPO_TERM_F32) {
        if (!terminals) return fail(PPO_E_INVALID, "ppo_gae_scan_f32: terminals is null but terminal_kind=%d", terminal_kind);
    } else
        return fail(PPO_E_INVALID, "ppo_gae_scan_f32: unknown terminal_kind %d", terminal_kind);
    if (regime == PPO_SCAN_AUTO) regime = (A > kTilesMaxA) ? PPO_SCAN_COLUMNS : PPO_SCAN_TILES;
    if (regime != PPO_SCAN_COLUMNS && regime != PPO_SCAN_TILES)
        return fail(PPO_E_INVALID, "ppo_gae_scan_f32: unknown regime %d", regime);

    const float g32 = (float)gamma;     // python float * f32 array -> f32 scalar
    const double gla = gamma * lam_adv;  // python double product (rl/returns.py:27)
    const double glr = gamma * lam_ret;
    hipStream_t st = as_stream(stream);

#define PPO_DISPATCH(TERM)                                                                          \
    (regime == PPO_SCAN_COLUMNS                                                                     \
         ? launch_columns<TERM>(rewards, values, final_value, terminals, adv_out, ret_out, N, A, ld, \
                                g32, gla, glr, st)                                                  \
         : launch_tiles<TERM>(rewards, values, final_value, terminals, adv_out, ret_out, N, A, ld,   \
                              g32, gla, glr, st))
    switch (terminal_kind) {
        case PPO_TERM_U8: return PPO_DISPATCH(PPO_TERM_U8);
        case PPO_TERM_F32: return PPO_DISPATCH(PPO_TERM_F32);
        default: return PPO_DISPATCH(PPO_TERM_NONE);
    }
#undef PPO_DISPATCH
}

extern "C" int ppo_bootstrapped_returns_f32(const float *rewards, const void *dones, int done_kind,
                                            const float *final_value, const float *gamma_arr,
                                            double gamma, float *out, int N, int A, int64_t ld,
                                            void *stream)
{
    using namespace ppo;
    if (N < 0 || A < 0 || ld < A) return fail(PPO_E_INVALID, "ppo_bootstrapped_returns_f32: bad shape");
    if (N == 0 || A == 0) return PPO_OK;
    if (!rewards || !dones || !final_value || !out) return fail(PPO_E_INVALID, "ppo_bootstrapped_returns_f32: null pointer");
    if (done_kind != PPO_TERM_U8 && done_kind != PPO_TERM_F32)
        return fail(PPO_E_INVALID, "ppo_bootstrapped_returns_f32: done_kind must be U8 or F32");
    const size_t tsz = done_kind == PPO_TERM_F32 ? 4 : 1;
    const bool vec4 = (A % 4 == 0) && (ld % 4 == 0) && aligned(rewards, 16) && aligned(final_value, 16) &&
                      aligned(out, 16) && aligned(dones, 4 * tsz) && (!gamma_arr || aligned(gamma_arr, 16));
    const float g32 = (float)gamma;
    hipStream_t st = as_stream(stream);
    constexpr int U = 8;
#define PPO_BOOT(DONE)                                                                                   \
    if (vec4) {                                                                                          \
        const int packs = A / 4;                                                                         \
        hipLaunchKernelGGL((bootstrapped_columns_kernel<DONE, 4, U>), dim3((packs + 255) / 256),         \
                           dim3(256), 0, st, rewards, dones, final_value, gamma_arr, g32, out, N, packs, \
                           ld);                                                                          \
    } else {                                                                                             \
        hipLaunchKernelGGL((bootstrapped_columns_kernel<DONE, 1, U>), dim3((A + 63) / 64), dim3(64), 0,  \
                           st, rewards, dones, final_value, gamma_arr, g32, out, N, A, ld);              \
    }
    if (done_kind == PPO_TERM_U8) {
        PPO_BOOT(PPO_TERM_U8)
    } else {
        PPO_BOOT(PPO_TERM_F32)
    }
#undef PPO_BOOT
    return check_launch("bootstrapped_columns_kernel");
}

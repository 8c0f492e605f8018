// First-layer convolution of the IMPALA encoder on uint8 observations (<= 4 input channels, 16 output channels),
// fused with the 3x3 / stride 2 / pad 1 max-pool that follows it, POOLED OUT OF THE MFMA ACCUMULATORS
// (reference: rl/impala.py:96,104-105  x = firstconv(x); x = max_pool2d(x, 3, 2, 1); input scaling rl/models.py:842-848).
// OPT-IN (ppo_conv1_pool_form(0)): bit-identical to the LDS form and faster back to back, but an env step of the pipelined
// rollout is slower with it - see conv1_pool_supported() below and profiles/r04p_conv1_pool.md.
//
// Why a second form of conv3x3_pool_kernel (conv3x3.hip) for this layer: K = 9 * 4 = 36 is nine MFMAs per 16-pixel
// tile, so the LDS form spends most of an item outside the K loop - band staging, the pre-pool rows written to LDS,
// a barrier, the pooling phase reading them back (58 % of an item with the matrix pipe idle, stamps).  Here nothing
// goes through LDS and there is no barrier:
//
//  * A wave owns a strip: two 16-column tiles x a run of pooled rows of one image, and walks it top to bottom.
//    Tile k covers columns 14 k - 1 .. 14 k + 14: seven pooled outputs (centres at the odd lanes 1, 3, .. 13) whose
//    3-wide windows lie inside the tile, so the column direction of the pool is two DPP row shifts and the row
//    direction is a v_max3 over accumulators of consecutive convolution rows that sit in the same lane.
//    84 columns = 6 tiles exactly (42 = 6 x 7).
//  * The B operand (4 channels x 16 pixels per K step, lane = (pixel, channel)) of the nine taps is a 3 x 3 register
//    window per lane that slides down the image: a new convolution row costs ONE unaligned dword load per lane (the
//    bytes at columns c - 1, c, c + 1 of the new input row) through a range-checked buffer descriptor, three
//    v_cvt_f32_ubyteN and the exact x / 255 (u8_unit), issued two rows ahead.
//  * Same K order (tap-major), same v_mfma_f32_16x16x4_f32, bias added after the chain: bit-identical to
//    conv3x3_kernel + maxpool_fwd_kernel and to conv3x3_pool_kernel, ties and argmax included.  The argmax comes
//    from the decomposition "first maximum of each row, then first row holding the maximum" (= first in row-major
//    order); a pooled row with a zero maximum in one of its windows (whose sign is that of the FIRST zero) - or every
//    row of a strip whose weights / bias are not tame (|.| > 1e30 or NaN: outputs may be non-finite) - is redone by the
//    reference scan pool_window_scan on the nine values.
//  * One wave per workgroup, one strip per workgroup: the hardware dispatcher balances the strips over the SIMDs;
//    workgroup ids are read as (XCD, slot) so that the strips of one image share an L2.
#include "common.h"
#include "conv_stage.h"
#include "mfma.h"

#include <cstdlib>
#include <type_traits>

namespace ppo {
namespace {

// DPP row shifts inside the rows of 16 lanes (one pixel tile): the value of the lane to the left / right (zero where a
// row of lanes ends: those lanes hold no pooled output).  One v_mov_b32_dpp each; the inference form of columns_max
// folds the shift into v_max_f32_dpp by hand.
__device__ __forceinline__ float from_left(float v)   // row_shr:1: lane i <- lane i - 1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_right(float v)  // row_shl:1: lane i <- lane i + 1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}

// one convolution row of a tile after the column direction of the pool
struct HRow {
    f32x4 v;       // convolution + bias (column -1 of the image: -inf)
    float hm[4];   // max over columns c - 1, c, c + 1
    int hk[4];     // KB + the first of the three columns holding it (KB = 3 ky of the row's usual place in a window)
};

template <bool TRAIN, int KB>
__device__ __forceinline__ void columns_max(HRow &r)
{
    if constexpr (TRAIN) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float l = from_left(r.v[i]);
            const float m = __builtin_fmaxf(__builtin_fmaxf(l, r.v[i]), from_right(r.v[i]));
            r.hm[i] = m;
            r.hk[i] = l == m ? KB : (r.v[i] == m ? KB + 1 : KB + 2);
        }
    } else {
        // two v_max_f32 per value, each taking its shifted operand through DPP (the compiler emits two moves and a
        // v_max3_f32); s_nop 1: a DPP read of a VGPR needs two wait states after the VALU write, which the hazard
        // recogniser does not see inside an asm block
        float t0, t1, t2, t3;
        asm("s_nop 1\n\t"
            "v_max_f32_dpp %0, %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_max_f32_dpp %1, %9, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_max_f32_dpp %2, %10, %10 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_max_f32_dpp %3, %11, %11 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_max_f32_dpp %4, %8, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_max_f32_dpp %5, %9, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_max_f32_dpp %6, %10, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_max_f32_dpp %7, %11, %3 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(r.hm[0]), "=&v"(r.hm[1]), "=&v"(r.hm[2]), "=&v"(r.hm[3])
            : "v"(r.v[0]), "v"(r.v[1]), "v"(r.v[2]), "v"(r.v[3]));
    }
}

template <int H, int W, bool TRAIN, bool PACKED>
__global__ __launch_bounds__(64) void conv1_pool_kernel(const uint8_t *__restrict__ in, const int32_t *__restrict__ in_index,
                                                        const float *__restrict__ w, const float *__restrict__ bias,
                                                        float *__restrict__ out, uint8_t *__restrict__ argmax, int n_images,
                                                        int cin, int prs, int nstrip)
{
    static_assert(H % 2 == 0 && W % 2 == 0 && W % 4 == 0, "even maps: a window's last row / column always exists");
    constexpr int HO = H / 2, WO = W / 2;
    constexpr int NTILES = (WO + 6) / 7, NPAIR = (NTILES + 1) / 2;
    const int lane = threadIdx.x, l15 = lane & 15, g = lane >> 4;
    // workgroup id -> strip: ids b, b + 8, .. run on one XCD and take consecutive strips (one image = NPAIR * nstrip of them)
    const int n_tasks = n_images * NPAIR * nstrip;
    const int per_xcd = (n_tasks + 7) / 8;
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    const int task = xcd * per_xcd + slot;
    if (slot >= per_xcd || task >= n_tasks) return;
    const int img = task / (NPAIR * nstrip), rem = task % (NPAIR * nstrip);
    const int strip = rem / NPAIR, pair = rem % NPAIR;
    const int p0 = strip * prs, p1 = min(p0 + prs, HO);
    // (a map whose tiles do not pair up - 64 columns: 5 tiles - would need wave-uniform branches around the second tile's
    // MFMAs; that form measured no faster than the LDS kernel and is not built)
    static_assert(NTILES % 2 == 0 && WO % 7 == 0, "whole pairs of 16-column tiles");

    // A operand: lane (l15, g) holds w(co = l15, ci = g, tap s) for the nine K steps
    float wa[9];
    if constexpr (PACKED) {
        const float4 *pw = reinterpret_cast<const float4 *>(w);
        const float4 q0 = pw[lane], q1 = pw[64 + lane], q2 = pw[128 + lane];
        wa[0] = q0.x, wa[1] = q0.y, wa[2] = q0.z, wa[3] = q0.w, wa[4] = q1.x, wa[5] = q1.y, wa[6] = q1.z, wa[7] = q1.w, wa[8] = q2.x;
    } else {
#pragma unroll
        for (int s = 0; s < 9; ++s) wa[s] = g < cin ? w[(l15 * cin + g) * 9 + s] : 0.f;
    }
    float bias_r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bias_r[i] = bias ? bias[4 * g + i] : 0.f;
    // Tame parameters (every |w|, |b| <= 1e30: inputs are in [0, 1], 36 products per output) cannot produce a non-finite
    // output, so the walk does not look for one; otherwise every window of the strip goes through the reference scan.
    bool tame = true;
#pragma unroll
    for (int s = 0; s < 9; ++s) tame = tame && __builtin_fabsf(wa[s]) <= 1e30f;
#pragma unroll
    for (int i = 0; i < 4; ++i) tame = tame && __builtin_fabsf(bias_r[i]) <= 1e30f;
    const bool wild = __builtin_amdgcn_ballot_w64(!tame) != 0;
    // what is added to the accumulators: the bias; -inf in the lane that holds column -1 of the image (pool padding)
    float badd[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        badd[0][i] = (pair == 0 && l15 == 0) ? -INFINITY : bias_r[i];
        badd[1][i] = bias_r[i];
    }

    // input: the dword at column a = clamp(c - 1, 0, W - 4) of a row, shifted so that bytes 0..2 are columns c - 1, c, c + 1
    // (zeros shift in for the columns outside the image); channels >= cin read channel cin - 1 against zero weights
    const int img_src = in_index ? in_index[img] : img;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(in) + (size_t)img_src * cin * (H * W), 0, cin * (H * W), 0x00020000);
    int voff[2], shl[2], shr[2], xo[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int col = 14 * (2 * pair + t) - 1 + l15;
        const int a = min(max(col - 1, 0), W - 4);
        const int sh = col - 1 - a;
        shl[t] = sh < 0 ? -8 * sh : 0;
        shr[t] = sh > 0 ? min(8 * sh, 24) : 0;
        voff[t] = min(g, cin - 1) * (H * W) + a;
#ifdef PPO_TUNE_C1_ALIGNED  // timing aid (tools/build_variant.sh): dword-aligned loads, wrong columns
        voff[t] &= ~3;
#endif
        xo[t] = 7 * (2 * pair + t) + (l15 >> 1);
    }
    auto fetch = [&](int t, int y) -> uint32_t {  // y is wave-uniform; rows outside the image read as zeros
        const bool ok = y >= 0 && y < H;
        return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rin, ok ? voff[t] : kOutside, ok ? y * W : 0, 0);
    };
    auto unpack = [&](int t, uint32_t d, float (&x)[3]) {
        const uint32_t v = (d << shl[t]) >> shr[t];
        x[0] = u8_unit((float)(v & 0xffu));
        x[1] = u8_unit((float)((v >> 8) & 0xffu));
        x[2] = u8_unit((float)((v >> 16) & 0xffu));
    };

    // output: lane (odd l15 <= 13, g) stores channels 4 g .. 4 g + 3 of pooled column xo; the others are out of range
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)img * 16 * (HO * WO), 0,
                                                                          16 * HO * WO * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ramx = __builtin_amdgcn_make_buffer_rsrc(
        TRAIN ? argmax + (size_t)img * 16 * (HO * WO) : nullptr, 0, TRAIN ? 16 * HO * WO : 0, 0x00020000);
    int ooff_f[2], ooff_b[2];  // byte offsets into the pooled map / the argmax map
    uint64_t centres[2];       // the lanes of a tile that hold a pooled output (the others hold don't-care values)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const bool centre = (l15 & 1) && l15 <= 13;
        centres[t] = __builtin_amdgcn_ballot_w64(centre);
        ooff_b[t] = centre ? 4 * g * (HO * WO) + xo[t] : kOutside;
        ooff_f[t] = centre ? 4 * (4 * g * (HO * WO) + xo[t]) : kOutside;
    }

    // The 3 x 3 input window per tile - input row y lives in slot (y - ys + 1) % 3, so the j-th convolution row of the
    // strip reads slots j % 3, (j + 1) % 3, (j + 2) % 3 as ky = 0, 1, 2 and the row that slides in replaces slot j % 3 -
    // and the two rows in flight behind it.  A strip is 1 + 2 (p1 - p0) convolution rows starting at ys = 2 p0 - 1
    // (the strip of pooled row 0 computes a row -1 that is replaced by padding).
    const int ys = 2 * p0 - 1;
    float xw[2][3][3];
    uint32_t pend0[2], pend1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const uint32_t d0 = fetch(t, ys - 1), d1 = fetch(t, ys), d2 = fetch(t, ys + 1);
        pend0[t] = fetch(t, ys + 2);
        pend1[t] = fetch(t, ys + 3);
        unpack(t, d0, xw[t][0]);
        unpack(t, d1, xw[t][1]);
        unpack(t, d2, xw[t][2]);
    }
    int ynext = ys + 4;

    // one convolution row of both tiles (PH = j % 3): request the input row three below, nine MFMAs per tile, bias,
    // column maxima, slide the window
    auto conv_row = [&](auto ph, auto kb, HRow (&r)[2]) {
        constexpr int PH = decltype(ph)::value, KB = decltype(kb)::value;
        uint32_t newd[2];
#ifdef PPO_TUNE_C1_NOLOAD  // timing aid: no loads inside the walk
        newd[0] = newd[1] = 0x01020304u;
#else
        newd[0] = fetch(0, ynext);
        newd[1] = fetch(1, ynext);
#endif
        ++ynext;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 9; ++s) {
#ifdef PPO_TUNE_C1_NOMFMA  // timing aid: one multiply-add per lane instead of the MFMA
            acc0[s % 4] = fmaf(wa[s], xw[0][(PH + s / 3) % 3][s % 3], acc0[s % 4]);
            acc1[s % 4] = fmaf(wa[s], xw[1][(PH + s / 3) % 3][s % 3], acc1[s % 4]);
#else
            acc0 = mfma16(wa[s], xw[0][(PH + s / 3) % 3][s % 3], acc0);
            acc1 = mfma16(wa[s], xw[1][(PH + s / 3) % 3][s % 3], acc1);
#endif
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r[0].v[i] = acc0[i] + badd[0][i];
            r[1].v[i] = acc1[i] + badd[1][i];
        }
        columns_max<TRAIN, KB>(r[0]);
        columns_max<TRAIN, KB>(r[1]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            unpack(t, pend0[t], xw[t][PH]);
            pend0[t] = pend1[t];
            pend1[t] = newd[t];
        }
    };

    // one pooled row of both tiles from its three convolution rows (a: hk based at 6, b: at 3, c: at 6)
    auto pool_row = [&](const HRow (&ra)[2], const HRow (&rb)[2], const HRow (&rc)[2], int p) {
        float best[2][4];
        int tap[2][4];
        uint64_t zeros = 0;  // centre lanes with a zero maximum
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float tiny = INFINITY;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float m = __builtin_fmaxf(__builtin_fmaxf(ra[t].hm[i], rb[t].hm[i]), rc[t].hm[i]);
                best[t][i] = m;
                if (TRAIN) tap[t][i] = ra[t].hm[i] == m ? ra[t].hk[i] - 6 : (rb[t].hm[i] == m ? rb[t].hk[i] : rc[t].hk[i]);
                tiny = __builtin_fminf(tiny, __builtin_fabsf(m));
            }
            zeros |= __builtin_amdgcn_ballot_w64(tiny == 0.f) & centres[t];
        }
        if (wild || zeros != 0) {
            // rare: the reference scan on the nine values of every window of this pooled row
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float win[3][3];
                    win[0][0] = from_left(ra[t].v[i]), win[0][1] = ra[t].v[i], win[0][2] = from_right(ra[t].v[i]);
                    win[1][0] = from_left(rb[t].v[i]), win[1][1] = rb[t].v[i], win[1][2] = from_right(rb[t].v[i]);
                    win[2][0] = from_left(rc[t].v[i]), win[2][1] = rc[t].v[i], win[2][2] = from_right(rc[t].v[i]);
                    pool_window_scan<true, true>(win, p > 0, true, xo[t] > 0, true, best[t][i], tap[t][i]);
                }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int so = i * (HO * WO) + p * WO;
#ifdef PPO_TUNE_C1_NOSTORE  // timing aid: results stay in registers
                asm volatile("" ::"v"(best[t][i]), "v"(tap[t][i]));
                continue;
#endif
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, best[t][i]), rout, ooff_f[t], 4 * so, 0);
                if constexpr (TRAIN) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)tap[t][i], ramx, ooff_b[t], so, 0);
            }
        }
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;
    using K6 = std::integral_constant<int, 6>;
    HRow ra[2], rb[2], rc[2];
    conv_row(I0{}, K6{}, ra);
    if (p0 == 0) {  // pooled row 0: the window's first row is padding
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ra[t].v = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[t].hm[i] = -INFINITY, ra[t].hk[i] = 6;
        }
    }
    // three pooled rows per trip: the window slots repeat every three convolution rows, the roles of ra / rc swap
    // every pooled row
    int p = p0;
#pragma unroll 1
    for (;;) {
        conv_row(I1{}, K3{}, rb);
        conv_row(I2{}, K6{}, rc);
        pool_row(ra, rb, rc, p);
        if (++p == p1) break;
        conv_row(I0{}, K3{}, rb);
        conv_row(I1{}, K6{}, ra);
        pool_row(rc, rb, ra, p);
        if (++p == p1) break;
        conv_row(I2{}, K3{}, rb);
        conv_row(I0{}, K6{}, rc);
        pool_row(ra, rb, rc, p);
        if (++p == p1) break;
#pragma unroll
        for (int t = 0; t < 2; ++t) ra[t] = rc[t];
    }
}

template <int H, int W, bool TRAIN, bool PACKED>
int launch_conv1(const void *in, const int32_t *in_index, const float *w, const float *bias, float *out, uint8_t *argmax,
                 int n, int cin, hipStream_t st)
{
    constexpr int HO = H / 2, WO = W / 2, NPAIR = ((WO + 6) / 7 + 1) / 2;
    // pooled rows per strip: a strip computes 2 prs + 1 convolution rows for prs pooled ones; shorter strips for fewer
    // images, so that every SIMD still sees several
    static const int forced = getenv("PPO_AMD_CONV1_PRS") ? atoi(getenv("PPO_AMD_CONV1_PRS")) : 0;
    int prs = n >= 192 ? 7 : 6;  // n = 128: 20.1 us with strips of 6, 21.9 with 7, 22.3 with 3 (tools/conv1_speed.py)
    if (forced > 0) prs = forced;
    if (prs > HO) prs = HO;
    const int nstrip = (HO + prs - 1) / prs;
    const long n_tasks = (long)n * NPAIR * nstrip;
    const long grid = (n_tasks + 7) / 8 * 8;
    if (grid > 0x7fffffffL) return fail(PPO_E_INVALID, "conv1_pool: %d images are too many strips for one launch", n);
    hipLaunchKernelGGL((conv1_pool_kernel<H, W, TRAIN, PACKED>), dim3((unsigned)grid), dim3(64), 0, st,
                       static_cast<const uint8_t *>(in), in_index, w, bias, out, argmax, n, cin, prs, nstrip);
    return check_launch("conv1_pool_kernel");
}

}  // namespace

// Whether a first-layer launch takes this form: only when asked (ppo_conv1_pool_form(0) / PPO_AMD_CONV1_LDS=0).  Measured on one
// MI355X (profiles/r04p_conv1_pool.md): back-to-back launches on a hot input are faster in the inference form (84x84: 44.0 ->
// 35.4 us at 256 images, 22.9 -> 20.4 us at 128; training form 45.5 vs 46.4 us), but INSIDE the pipelined rollout, where the
// observations have just been uploaded and the other env group's kernels share the chip, an env step is slower with it
// (0.4933 vs 0.4727 ms, interleaved A/B tools/rollout_ab.py PPO_AB=conv1; bench.py's event brackets: 51.6 vs 30.6 us for the
// launch).  Cold input is not the reason (a ring of 690 MB of inputs: still 21.0 vs 23.3 us back to back); what differs there is
// that 4 608 one-wave workgroups share the chip with the other group's one-workgroup-per-CU kernels.  So the default is the LDS
// form everywhere (-1 and 1 are the same today); both kernels are bit-identical: a speed switch, not a result switch.
static int g_form = getenv("PPO_AMD_CONV1_LDS") ? atoi(getenv("PPO_AMD_CONV1_LDS")) : -1;

bool conv1_pool_supported(int cin, int cout, int h, int w, bool train)
{
    const bool can = cin == 4 && cout == 16 && h == 84 && w == 84;  // the Atari first layer (rl/atari.py: 4 stacked 84x84 frames)
    (void)train;
    return can && g_form == 0;
}

int conv1_pool_forward(const void *in, const int32_t *in_index, const float *w, bool packed, const float *bias, float *out,
                       uint8_t *argmax, int n, int cin, int h, int w_, hipStream_t st)
{
#define PPO_C1(HH)                                                                                                      \
    if (h == HH && w_ == HH) {                                                                                          \
        if (argmax)                                                                                                     \
            return packed ? launch_conv1<HH, HH, true, true>(in, in_index, w, bias, out, argmax, n, cin, st)            \
                          : launch_conv1<HH, HH, true, false>(in, in_index, w, bias, out, argmax, n, cin, st);          \
        return packed ? launch_conv1<HH, HH, false, true>(in, in_index, w, bias, out, argmax, n, cin, st)               \
                      : launch_conv1<HH, HH, false, false>(in, in_index, w, bias, out, argmax, n, cin, st);             \
    }
    PPO_C1(84)
#undef PPO_C1
    return fail(PPO_E_INVALID, "conv1_pool: unsupported geometry h=%d w=%d", h, w_);
}

}  // namespace ppo

extern "C" int ppo_conv1_pool_form(int form)
{
    const int before = ppo::g_form;
    ppo::g_form = form < 0 ? -1 : (form > 0 ? 1 : 0);
    return before;
}

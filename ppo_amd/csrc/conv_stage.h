// Global -> LDS staging shared by the convolution kernels (gfx950).
//
// A band of image rows (+halo, zero padded) of every input channel goes to LDS as planar
// [channel][row][col] with the load transform fused (identity / ReLU / uint8 -> x/255).
//
// Mapping: a wave-instruction covers 64/LG consecutive band rows with LG (16/32/64) lanes per row,
// lane = column; the (channel, row-block) a wave works on is wave-uniform, so the per-element work
// is a handful of VALU instructions (the first version decoded a flat index per element with three
// constant divisions: ~30 VALU per element, as much VALU time as the MFMAs themselves —
// profiles/: SQ_INSTS_VALU 8.2 per MFMA).  Loads are issued U at a time before the LDS stores so
// that U requests per thread are in flight.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ppo {

enum { IN_NONE = 0, IN_RELU = 1, IN_U8 = 2 };

// uint8 observation -> x / 255 (rl/models.py:842-848: x.float() / 255), bit for bit the IEEE quotient for every one of
// the 256 inputs, as a multiply and two fused multiply-adds instead of the ~10-instruction division sequence:
//   q = x * (1/255);  q += (x - 255 q) * (1/255)     (one Newton correction of the rounded reciprocal product;
// checked exhaustively on the host, tests/test_host_cpu.py, and against torch on the device, tests/test_conv_gpu.py)
__device__ __forceinline__ float u8_unit(float x)
{
    constexpr float r = 1.0f / 255.0f;
    const float q = x * r;
    return fmaf(fmaf(q, -255.0f, x), r, q);
}

// s_dst[c * PLANE + r * PW + col] for c < CP, r < ROWS, col < PW holds
//   f(src[img][c][y0 + r - HALO][col - HALO])   (0 outside the image or for c >= C)
template <int C, int CP, int H, int W, int ROWS, int PW, int PLANE, int HALO, int IN_MODE, int NWAVES>
__device__ __forceinline__ void stage_band(const void *__restrict__ src_, int img, int y0, float *__restrict__ s_dst,
                                           int tid)
{
    constexpr int LG = PW <= 16 ? 16 : (PW <= 32 ? 32 : 64);
    constexpr int RPI = 64 / LG;               // band rows per wave-instruction
    constexpr int NPASS = (PW + 63) / 64;      // column passes (> 1 only for PW > 64)
    constexpr int RSTEP = NWAVES * RPI;        // band rows per workgroup step
    constexpr int RB = (ROWS + RSTEP - 1) / RSTEP;
    constexpr int Q = CP * RB * NPASS;         // wave-uniform work items
    constexpr int U = 8;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int sub = lane / LG;
    const int col0 = lane % LG;
    const int rw = wave * RPI + sub;  // this lane's row inside a row block
#pragma unroll 1
    for (int q0 = 0; q0 < Q; q0 += U) {
        float v[U];
        int off[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + u;  // wave-uniform
            const int pass = q % NPASS;
            const int qq = q / NPASS;
            const int c = qq / RB;
            const int r = (qq % RB) * RSTEP + rw;
            const int col = col0 + 64 * pass;
            const bool in_lds = q < Q && r < ROWS && col < PW;
            const int gy = y0 + r - HALO;
            const int gx = col - HALO;
            off[u] = in_lds ? c * PLANE + r * PW + col : -1;
            v[u] = 0.f;
            if (in_lds && c < C && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                const size_t gi = ((size_t)(img * C + c) * H + gy) * W + gx;
                if (IN_MODE == IN_U8) v[u] = (float)static_cast<const uint8_t *>(src_)[gi];
                else v[u] = static_cast<const float *>(src_)[gi];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (off[u] >= 0) {
                float x = v[u];
                if (IN_MODE == IN_U8) x = u8_unit(x);
                if (IN_MODE == IN_RELU) x = fmaxf(x, 0.f);
                s_dst[off[u]] = x;
            }
        }
    }
}

// LDS-DMA variant for float32 sources (no fused transform: consumers apply ReLU when they read the
// operand).  One `global_load_lds_dword` per band row and wave: the data goes HBM -> LDS without passing
// through VGPRs, so a wave can have ALL its rows in flight at once (the register path is limited to U)
// and staging costs no VALU per element.  Requires the whole [CP][PLANE] region to have been zeroed once
// by the workgroup (halo columns and padded channels are never written here); rows outside the image are
// zeroed explicitly.  The caller's __syncthreads() drains the DMAs (hipcc emits vmcnt(0) before the barrier).
template <int C, int H, int W, int ROWS, int PW, int PLANE, int HALO, int NWAVES>
__device__ __forceinline__ void stage_band_dma(const float *__restrict__ src, int img, int y0,
                                               float *__restrict__ s_dst, int tid)
{
    using gptr_t = const __attribute__((address_space(1))) void *;
    using lptr_t = __attribute__((address_space(3))) void *;
    constexpr int NPASS = (W + 63) / 64;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // A wave owns band rows r = wave, wave + NWAVES, ... and walks the channels inside: between two
    // requests only a constant channel stride is added on both sides, so a request costs a few scalar
    // instructions (decoding a flat (channel,row) index per request cost more issue time than the
    // MFMAs the row feeds).
#pragma unroll 1
    for (int r = wave; r < ROWS; r += NWAVES) {
        const int gy = y0 + r - HALO;
        float *lrow = s_dst + r * PW + HALO;  // LDS address of column x = 0, channel 0
        if (gy >= 0 && gy < H) {
            const float *grow = src + ((size_t)img * C * H + gy) * W + lane;  // channel 0
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                if (lane + 64 * pass < W) {
#pragma unroll 8
                    for (int c = 0; c < C; ++c)
                        __builtin_amdgcn_global_load_lds((gptr_t)(grow + (size_t)c * H * W + 64 * pass),
                                                         (lptr_t)(lrow + c * PLANE + 64 * pass), 4, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int x = lane + 64 * pass;
                if (x < W)
                    for (int c = 0; c < C; ++c) lrow[c * PLANE + x] = 0.f;
            }
        }
    }
}

// The dy band of a stack's FIRST convolution computed from the pooled gradient instead of read from HBM:
//   dc = maxpool3x3s2_backward(g, argmax)   (even H, W: the 2x2-block form and summation order of maxpool_bwd2x2_kernel)
// written in stage_band_dma's HALO = 0 layout, s_dst[c * PLANE + r * PW + x].  One thread per 2x2 block of the band
// (4 (argmax, g) pairs -> 4 values).  The pre-pool gradient map (the largest tensor of the backward pass: 115 MB at
// 84x84x16x256) is then never written or read.
// Two phases, so that a caller can issue the loads of the NEXT item before its K loop and compute + write LDS after
// it: PooledRaw carries one task's loaded (g, argmax) neighbourhood untouched.
struct PooledRaw {
    float2 ga, gb;
    float ga2, gb2;
    unsigned aa, ab;
    int ta2, tb2;
};

template <int C, int H, int W, int TR, int NWAVES>
struct PooledMap {
    static_assert(H % 2 == 0 && W % 2 == 0 && TR % 2 == 0 && H % TR == 0, "even maps, whole bands");
    static constexpr int HO = H / 2, WO = W / 2, JB = TR / 2;
    static_assert(WO % 2 == 0 && (HO * WO) % 2 == 0, "8-byte aligned pairs");
    // a thread takes TWO neighbouring 2x2 blocks (pooled columns k0, k0 + 1): the windows that can select their
    // elements are (j, k0 .. k0 + 2) and (j + 1, k0 .. k0 + 2) -> per row one float2 + one float of g, one ushort + one
    // byte of argmax (8 loads for 8 outputs; one block per thread needs 8 loads for 4)
    static constexpr int KP = WO / 2, NTASK = C * JB * KP;
    static constexpr int Q = (NTASK + NWAVES * 64 - 1) / (NWAVES * 64);  // tasks per thread
};

template <int C, int H, int W, int TR, int NWAVES>
__device__ __forceinline__ void dy_pooled_load(const float *__restrict__ g, const uint8_t *__restrict__ argmax, int img,
                                               int y0, int tid, PooledRaw (&raw)[PooledMap<C, H, W, TR, NWAVES>::Q])
{
    using M = PooledMap<C, H, W, TR, NWAVES>;
    // Range-checked buffer reads (common.h) from this image's planes: a window beyond the map's right / bottom edge,
    // or a thread without a task, reads gradient 0 (and argmax 0, which then selects nothing but that 0).  Written
    // as `if (right) x = g[..]` every load was a branch with a full memory wait behind it, and the point of this
    // function - the gather of the NEXT item in flight during the K loop - was lost.
    const size_t img_base = (size_t)img * C * M::HO * M::WO;
    const __amdgpu_buffer_rsrc_t gb = buffer_of(g + img_base), ab = buffer_of(argmax + img_base);
#pragma unroll
    for (int q = 0; q < M::Q; ++q) {
        const int b = tid + q * NWAVES * 64;
        const bool task = b < M::NTASK;
        const int c = b / (M::JB * M::KP), rem = b % (M::JB * M::KP);
        const int jb = rem / M::KP, k0 = 2 * (rem % M::KP);
        const int j = y0 / 2 + jb;
        const int t = (c * M::HO + j) * M::WO + k0;
        const bool right = task && k0 + 2 < M::WO, down = task && j + 1 < M::HO;
        const int o_a = task ? t : -1, o_a2 = right ? t + 2 : -1, o_b = down ? t + M::WO : -1,
                  o_b2 = (down && right) ? t + M::WO + 2 : -1;
        PooledRaw r;
        r.ga = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(gb, o_a < 0 ? kOutside : o_a * 4, 0, 0));
        r.gb = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(gb, o_b < 0 ? kOutside : o_b * 4, 0, 0));
        r.ga2 = buffer_f32(gb, o_a2 < 0 ? kOutside : o_a2 * 4);
        r.gb2 = buffer_f32(gb, o_b2 < 0 ? kOutside : o_b2 * 4);
        r.aa = __builtin_amdgcn_raw_buffer_load_b16(ab, o_a < 0 ? kOutside : o_a, 0, 0);
        r.ab = __builtin_amdgcn_raw_buffer_load_b16(ab, o_b < 0 ? kOutside : o_b, 0, 0);
        r.ta2 = __builtin_amdgcn_raw_buffer_load_b8(ab, o_a2 < 0 ? kOutside : o_a2, 0, 0);
        r.tb2 = __builtin_amdgcn_raw_buffer_load_b8(ab, o_b2 < 0 ? kOutside : o_b2, 0, 0);
        raw[q] = r;
    }
}

template <int C, int H, int W, int TR, int PW, int PLANE, int NWAVES>
__device__ __forceinline__ void dy_pooled_store(const PooledRaw (&raw)[PooledMap<C, H, W, TR, NWAVES>::Q], int y0,
                                                float *__restrict__ s_dst, int tid)
{
    using M = PooledMap<C, H, W, TR, NWAVES>;
    static_assert(PLANE % 2 == 0 && PW % 2 == 0, "8-byte aligned pairs");
#pragma unroll
    for (int q = 0; q < M::Q; ++q) {
        const int b = tid + q * NWAVES * 64;
        if (b >= M::NTASK) continue;
        const int c = b / (M::JB * M::KP), rem = b % (M::JB * M::KP);
        const int jb = rem / M::KP, k0 = 2 * (rem % M::KP);
        const bool down = y0 / 2 + jb + 1 < M::HO;
        const float2 ga = raw[q].ga, gb = raw[q].gb;
        const float ga2 = raw[q].ga2, gb2 = raw[q].gb2;
        const unsigned aa = raw[q].aa, ab = raw[q].ab;
        const int ta2 = raw[q].ta2, tb2 = raw[q].tb2;
        const int ta0 = aa & 0xff, ta1 = aa >> 8;
        const int tb0 = down ? (int)(ab & 0xff) : -1, tb1 = down ? (int)(ab >> 8) : -1;
        // block k0: windows (j,k0) (j,k0+1) (j+1,k0) (j+1,k0+1); block k0+1: (j,k0+1) (j,k0+2) (j+1,k0+1) (j+1,k0+2);
        // summation order per element as maxpool_bwd2x2_kernel
        float2 top0, top1, bot0, bot1;
        top0.x = (ta0 == 4 ? ga.x : 0.f);
        top0.y = (ta0 == 5 ? ga.x : 0.f);
        top0.y += (ta1 == 3 ? ga.y : 0.f);
        bot0.x = (ta0 == 7 ? ga.x : 0.f);
        bot0.x += (tb0 == 1 ? gb.x : 0.f);
        bot0.y = (ta0 == 8 ? ga.x : 0.f);
        bot0.y += (ta1 == 6 ? ga.y : 0.f);
        bot0.y += (tb0 == 2 ? gb.x : 0.f);
        bot0.y += (tb1 == 0 ? gb.y : 0.f);
        top1.x = (ta1 == 4 ? ga.y : 0.f);
        top1.y = (ta1 == 5 ? ga.y : 0.f);
        top1.y += (ta2 == 3 ? ga2 : 0.f);
        bot1.x = (ta1 == 7 ? ga.y : 0.f);
        bot1.x += (tb1 == 1 ? gb.y : 0.f);
        bot1.y = (ta1 == 8 ? ga.y : 0.f);
        bot1.y += (ta2 == 6 ? ga2 : 0.f);
        bot1.y += (tb1 == 2 ? gb.y : 0.f);
        bot1.y += (tb2 == 0 ? gb2 : 0.f);
        float *d = s_dst + c * PLANE + (2 * jb) * PW + 2 * k0;
        *reinterpret_cast<float2 *>(d) = top0;
        *reinterpret_cast<float2 *>(d + 2) = top1;
        *reinterpret_cast<float2 *>(d + PW) = bot0;
        *reinterpret_cast<float2 *>(d + PW + 2) = bot1;
    }
}

template <int C, int H, int W, int TR, int PW, int PLANE, int NWAVES>
__device__ __forceinline__ void stage_dy_pooled(const float *__restrict__ g, const uint8_t *__restrict__ argmax, int img,
                                                int y0, float *__restrict__ s_dst, int tid)
{
    PooledRaw raw[PooledMap<C, H, W, TR, NWAVES>::Q];
    dy_pooled_load<C, H, W, TR, NWAVES>(g, argmax, img, y0, tid, raw);
    dy_pooled_store<C, H, W, TR, PW, PLANE, NWAVES>(raw, y0, s_dst, tid);
}

// ---------------------------------------------------------------------------------------------------------
// "Flat" band layout: a channel plane holds the band rows back to back with the IMAGE's row stride,
//   s_dst[c * PLANE + G + r * W + x] = f(src[img][c][y0 + r - 1][x]),  r < ROWS (one halo row above / below),
// i.e. no halo COLUMNS: the consumer masks the x-1 / x+1 taps at the image's left / right edge instead.
// Because global memory has the same row stride, the in-image part of a plane is ONE contiguous run, so the
// DMA moves it with 16-byte requests (64 lanes x 4 floats = 256 floats per request) instead of one 4-byte
// request per row: 4-8x fewer requests, and request issue was what bounded the convolutions
// (tools/dma16_probe.hip: the 16-byte form costs the same to issue and accepts 4-byte-aligned addresses on
// both sides).  G floats of guard before and after the rows keep the masked edge reads inside the plane.

// Register path (uint8 observations: /255 fused).  Same mapping idea as stage_band.
template <int C, int CP, int H, int W, int ROWS, int PLANE, int G, int IN_MODE, int NWAVES>
__device__ __forceinline__ void stage_band_flat(const void *__restrict__ src_, int img, int y0,
                                                float *__restrict__ s_dst, int tid)
{
    constexpr int LG = W <= 16 ? 16 : (W <= 32 ? 32 : 64);
    constexpr int RPI = 64 / LG;
    constexpr int NPASS = (W + 63) / 64;
    constexpr int RSTEP = NWAVES * RPI;
    constexpr int RB = (ROWS + RSTEP - 1) / RSTEP;
    constexpr int Q = CP * RB * NPASS;
    constexpr int U = 8;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int sub = lane / LG;
    const int col0 = lane % LG;
    const int rw = wave * RPI + sub;
#pragma unroll 1
    for (int q0 = 0; q0 < Q; q0 += U) {
        float v[U];
        int off[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + u;  // wave-uniform
            const int pass = q % NPASS;
            const int qq = q / NPASS;
            const int c = qq / RB;
            const int r = (qq % RB) * RSTEP + rw;
            const int col = col0 + 64 * pass;
            const bool in_lds = q < Q && r < ROWS && col < W;
            const int gy = y0 + r - 1;
            off[u] = in_lds ? c * PLANE + G + r * W + col : -1;
            v[u] = 0.f;
            if (in_lds && c < C && gy >= 0 && gy < H) {
                const size_t gi = ((size_t)(img * C + c) * H + gy) * W + col;
                if (IN_MODE == IN_U8) v[u] = (float)static_cast<const uint8_t *>(src_)[gi];
                else v[u] = static_cast<const float *>(src_)[gi];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (off[u] >= 0) {
                float x = v[u];
                if (IN_MODE == IN_U8) x = u8_unit(x);
                if (IN_MODE == IN_RELU) x = fmaxf(x, 0.f);
                s_dst[off[u]] = x;
            }
        }
    }
}

// The same register path in two phases, so that a caller can issue the loads of the NEXT item before its
// compute phase and write them to LDS after it (the loads' latency hides under the MFMAs): `raw` carries the
// loaded bits untouched (no conversion, so nothing waits on the loads until band_flat_store).
template <int CP, int W, int ROWS, int NWAVES>
struct FlatMap {
    static constexpr int LG = W <= 16 ? 16 : (W <= 32 ? 32 : 64);
    static constexpr int RPI = 64 / LG;
    static constexpr int NPASS = (W + 63) / 64;
    static constexpr int RSTEP = NWAVES * RPI;
    static constexpr int RB = (ROWS + RSTEP - 1) / RSTEP;
    static constexpr int Q = CP * RB * NPASS;  // wave-uniform work items = registers per thread
};

template <int C, int CP, int H, int W, int ROWS, int IN_MODE, int NWAVES>
__device__ __forceinline__ void band_flat_load(const void *__restrict__ src_, int img, int y0, int tid,
                                               uint32_t (&raw)[FlatMap<CP, W, ROWS, NWAVES>::Q])
{
    using M = FlatMap<CP, W, ROWS, NWAVES>;
    const int lane = tid & 63, wave = tid >> 6;
    const int rw = wave * M::RPI + lane / M::LG;
    const int col0 = lane % M::LG;
#pragma unroll
    for (int q = 0; q < M::Q; ++q) {
        const int pass = q % M::NPASS;
        const int qq = q / M::NPASS;
        const int c = qq / M::RB;
        const int r = (qq % M::RB) * M::RSTEP + rw;
        const int col = col0 + 64 * pass;
        const int gy = y0 + r - 1;
        raw[q] = 0;
        if (r < ROWS && col < W && c < C && gy >= 0 && gy < H) {
            const size_t gi = ((size_t)(img * C + c) * H + gy) * W + col;
            if (IN_MODE == IN_U8) raw[q] = static_cast<const uint8_t *>(src_)[gi];
            else raw[q] = static_cast<const uint32_t *>(src_)[gi];
        }
    }
}

template <int CP, int W, int ROWS, int PLANE, int G, int IN_MODE, int NWAVES>
__device__ __forceinline__ void band_flat_store(const uint32_t (&raw)[FlatMap<CP, W, ROWS, NWAVES>::Q],
                                                float *__restrict__ s_dst, int tid)
{
    using M = FlatMap<CP, W, ROWS, NWAVES>;
    const int lane = tid & 63, wave = tid >> 6;
    const int rw = wave * M::RPI + lane / M::LG;
    const int col0 = lane % M::LG;
#pragma unroll
    for (int q = 0; q < M::Q; ++q) {
        const int pass = q % M::NPASS;
        const int qq = q / M::NPASS;
        const int c = qq / M::RB;
        const int r = (qq % M::RB) * M::RSTEP + rw;
        const int col = col0 + 64 * pass;
        if (r < ROWS && col < W) {
            float x;
            if (IN_MODE == IN_U8) x = u8_unit((float)raw[q]);
            else x = __uint_as_float(raw[q]);
            if (IN_MODE == IN_RELU) x = fmaxf(x, 0.f);
            s_dst[c * PLANE + G + r * W + col] = x;
        }
    }
}

// uint8 observations, FOUR pixels per lane (W % 4 == 0): in the flat layout a channel's band is one contiguous run of
// ROWS * W bytes in global memory and of as many floats in LDS, so thread t of the workgroup takes dword t (then
// t + threads, ...) of the [channel][run] index space: one 4-byte load, four v_cvt_f32_ubyteN + u8_unit, one 16-byte
// LDS store.  The per-pixel form issued one byte load and one 4-byte LDS store per pixel (16 load slots per thread for
// the 84x84 band, a third of them masked off) — the first layer is bound by instruction issue, not by MFMAs (K = 36).
// Two phases like band_flat_load / band_flat_store, so the next item's loads fly during this item's compute.
template <int C, int W, int ROWS, int NTHREADS>
struct FlatU8Map {
    static_assert(W % 4 == 0, "a dword must not straddle two rows");
    static constexpr int DW = ROWS * W / 4;                           // dwords per channel run
    static constexpr int Q = (C * DW + NTHREADS - 1) / NTHREADS;      // dwords (registers) per thread
};

template <int C, int H, int W, int ROWS, int NTHREADS>
__device__ __forceinline__ void band_u8x4_load(const void *__restrict__ src_, int img, int y0, int tid,
                                               uint32_t (&raw)[FlatU8Map<C, W, ROWS, NTHREADS>::Q])
{
    using M = FlatU8Map<C, W, ROWS, NTHREADS>;
    const uint8_t *src = static_cast<const uint8_t *>(src_);
#pragma unroll
    for (int q = 0; q < M::Q; ++q) {
        const int idx = tid + q * NTHREADS;
        const int c = idx / M::DW, d = idx % M::DW;
        const int r = (4 * d) / W, col = (4 * d) % W;
        const int gy = y0 + r - 1;
        raw[q] = 0;  // rows outside the image read as zeros
        if (idx < C * M::DW && gy >= 0 && gy < H)
            raw[q] = *reinterpret_cast<const uint32_t *>(src + ((size_t)(img * C + c) * H + gy) * W + col);
    }
}

template <int C, int W, int ROWS, int PLANE, int G, int NTHREADS>
__device__ __forceinline__ void band_u8x4_store(const uint32_t (&raw)[FlatU8Map<C, W, ROWS, NTHREADS>::Q],
                                                float *__restrict__ s_dst, int tid)
{
    using M = FlatU8Map<C, W, ROWS, NTHREADS>;
    static_assert(PLANE % 4 == 0 && G % 4 == 0, "16-byte aligned LDS stores");
#pragma unroll
    for (int q = 0; q < M::Q; ++q) {
        const int idx = tid + q * NTHREADS;
        if (idx < C * M::DW) {
            const int c = idx / M::DW, d = idx % M::DW;
            const uint32_t v = raw[q];
            *reinterpret_cast<float4 *>(s_dst + c * PLANE + G + 4 * d) =
                make_float4(u8_unit((float)(v & 0xffu)), u8_unit((float)((v >> 8) & 0xffu)),
                            u8_unit((float)((v >> 16) & 0xffu)), u8_unit((float)(v >> 24)));
        }
    }
}

// LDS-DMA path.  Wave w moves channels w, w + NWAVES, ...: per channel the in-image rows are one run of
// n = rows * W floats -> n/4 lanes of 16-byte requests (256 floats each) plus one 4-byte request for the
// n % 4 leftover floats (never over-reads the source, never over-writes the rows below).  Band rows outside
// the image (first / last band of an image) are zeroed with plain LDS stores.  Padded channels (>= C) and
// the guards are never written: the caller zeroes the whole region once.
// AUX: cache-policy bits of the loads (0 = default; 16 = sc1: device-coherent reads that do not hit the CU's own L1,
// for data another workgroup has just written - stack_chain_split_kernel's exchange).
template <int C, int H, int W, int ROWS, int PLANE, int G, int NWAVES, int AUX = 0>
__device__ __forceinline__ void stage_band_chunk_dma(const float *__restrict__ src, int img, int y0,
                                                     float *__restrict__ s_dst, int tid)
{
    using gptr_t = const __attribute__((address_space(1))) void *;
    using lptr_t = __attribute__((address_space(3))) void *;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r_lo = y0 >= 1 ? 0 : 1 - y0;                          // first band row inside the image
    const int r_hi = (y0 - 1 + ROWS) <= H ? ROWS : H - (y0 - 1);    // one past the last
    constexpr int MAXREQ = (ROWS * W / 4 + 63) / 64;
    if (r_lo == 0 && r_hi == ROWS) {
        // Interior band (all but the first / last band of an image): every count below is a compile-time constant,
        // so a channel is MAXREQ requests and one add - the general form's per-request predicates, tail request and
        // zero-fill tests were ~100 instructions per channel, 17 % of a weight-gradient item (stamps).
        constexpr int N = ROWS * W, N4 = N / 4, TAIL = N % 4;
        const float *g0 = src + ((size_t)img * C * H + (size_t)(y0 - 1)) * W + 4 * lane;
        float *l0 = s_dst + G;
#pragma unroll 1
        for (int c = wave; c < C; c += NWAVES) {
            const float *gc = g0 + (size_t)c * (H * W);
            float *lc = l0 + c * PLANE;
#pragma unroll
            for (int q = 0; q < MAXREQ; ++q) {
                if ((q + 1) * 64 <= N4)  // whole request: no lane test at all
                    __builtin_amdgcn_global_load_lds((gptr_t)(gc + q * 256), (lptr_t)(lc + q * 256), 16, 0, AUX);
                else if (q * 64 + lane < N4)
                    __builtin_amdgcn_global_load_lds((gptr_t)(gc + q * 256), (lptr_t)(lc + q * 256), 16, 0, AUX);
            }
            if constexpr (TAIL > 0) {
                if (lane < TAIL)
                    __builtin_amdgcn_global_load_lds((gptr_t)(gc - 4 * lane + N4 * 4 + lane), (lptr_t)(lc + N4 * 4), 4, 0, AUX);
            }
        }
        return;
    }
    const int n = (r_hi - r_lo) * W;                                // floats to move per channel
    const int n4 = n >> 2;                                          // 16-byte lanes
    const int tail = n & 3;
#pragma unroll 1
    for (int c = wave; c < C; c += NWAVES) {
        const float *g0 = src + ((size_t)(img * C + c) * H + (y0 - 1 + r_lo)) * W;
        float *l0 = s_dst + c * PLANE + G + r_lo * W;
#pragma unroll
        for (int q = 0; q < MAXREQ; ++q) {
            if (q * 64 + lane < n4)
                __builtin_amdgcn_global_load_lds((gptr_t)(g0 + (q * 64 + lane) * 4), (lptr_t)(l0 + q * 256), 16, 0, AUX);
        }
        if (lane < tail)
            __builtin_amdgcn_global_load_lds((gptr_t)(g0 + n4 * 4 + lane), (lptr_t)(l0 + n4 * 4), 4, 0, AUX);
        if (r_lo > 0)
            for (int x = lane; x < r_lo * W; x += 64) s_dst[c * PLANE + G + x] = 0.f;
        if (r_hi < ROWS)
            for (int x = lane; x < (ROWS - r_hi) * W; x += 64) s_dst[c * PLANE + G + r_hi * W + x] = 0.f;
    }
}

template <int WORDS, int NTHREADS>
__device__ __forceinline__ void zero_lds(float *__restrict__ s, int tid)
{
    for (int i = tid; i < WORDS; i += NTHREADS) s[i] = 0.f;
}

// Running maximum over a 3x3 pooling window in row-major tap order; v[ky][kx] may hold anything where the tap lies outside
// the map (r0 / r2: the window's first / last row is inside, c0 / c2 likewise for columns; the middle row / column
// always is).  F.max_pool2d's rule: the first maximum wins, NaN propagates.  No "found" flag: the tap index starts at
// the first valid tap and the value at -inf, which gives the same result - a first tap of -inf leaves (value, index)
// as they are, anything larger or NaN replaces them.  R2 / C2 are compile-time when the map's height / width is even
// (the window's last row / column then always exists).
template <bool R2_ALWAYS, bool C2_ALWAYS>
__device__ __forceinline__ void pool_window_scan(const float (&v)[3][3], bool r0, bool r2, bool c0, bool c2, float &best,
                                                 int &best_tap)
{
    best = -INFINITY;
    best_tap = r0 ? (c0 ? 0 : 1) : (c0 ? 3 : 4);  // the first valid tap (taps 4, 5, 7, 8 need only r2 / c2)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float x = v[ky][kx];
            bool take = x > best || x != x;
            if (ky == 0) take = take && r0;
            if (ky == 2 && !R2_ALWAYS) take = take && r2;
            if (kx == 0) take = take && c0;
            if (kx == 2 && !C2_ALWAYS) take = take && c2;
            best = take ? x : best;
            best_tap = take ? ky * 3 + kx : best_tap;
        }
    }
}

// The same result with two thirds of the instructions when all nine values are finite (a sum tells: any NaN or
// infinity - or an overflow, which is no harm - sends the lane through the scan above): the maximum over the valid taps
// as a tree of three-operand maxima, then the FIRST tap holding it by an equality scan from the back.  The pooling
// phases of the conv + max-pool kernels are bound by VALU issue (stamps: 4.1 k of the 11.2 k cycles of a first-layer
// item), so instructions here are time.
template <bool R2_ALWAYS, bool C2_ALWAYS>
__device__ __forceinline__ void pool_window_max(const float (&v)[3][3], bool r0, bool r2, bool c0, bool c2, float &best,
                                                int &best_tap)
{
    const float sum = ((v[0][0] + v[0][1]) + (v[0][2] + v[1][0])) + ((v[1][1] + v[1][2]) + (v[2][0] + v[2][1])) + v[2][2];
    float w[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            bool ok = true;
            if (ky == 0) ok = ok && r0;
            if (ky == 2 && !R2_ALWAYS) ok = ok && r2;
            if (kx == 0) ok = ok && c0;
            if (kx == 2 && !C2_ALWAYS) ok = ok && c2;
            w[ky * 3 + kx] = ok ? v[ky][kx] : -INFINITY;
        }
    const float m0 = __builtin_fmaxf(__builtin_fmaxf(w[0], w[1]), w[2]);
    const float m1 = __builtin_fmaxf(__builtin_fmaxf(w[3], w[4]), w[5]);
    const float m2 = __builtin_fmaxf(__builtin_fmaxf(w[6], w[7]), w[8]);
    best = __builtin_fmaxf(__builtin_fmaxf(m0, m1), m2);
    int tap = 8;
#pragma unroll
    for (int k = 7; k >= 0; --k) tap = w[k] == best ? k : tap;
    best_tap = tap;
    // not finite somewhere, or a zero maximum (whose sign is that of the FIRST zero, which a maximum does not keep):
    // the reference's own scan decides
    if (!__builtin_isfinite(sum) || best == 0.f) pool_window_scan<R2_ALWAYS, C2_ALWAYS>(v, r0, r2, c0, c2, best, best_tap);
}

// The same from LDS: src -> tap (0, 0), rows `stride` floats apart.  All nine taps are read before the first compare,
// valid or not - the caller guarantees the addresses are inside the LDS allocation: `if (valid) v = src[..]` made every
// tap a branch with a full LDS wait behind it, nine round trips in a row per output.
template <bool R2_ALWAYS = false, bool C2_ALWAYS = false>
__device__ __forceinline__ void pool_window_lds(const float *src, int stride, bool r0, bool r2, bool c0, bool c2, float &best,
                                                int &best_tap)
{
    float v[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) v[ky][kx] = src[ky * stride + kx];
    pool_window_max<R2_ALWAYS, C2_ALWAYS>(v, r0, r2, c0, c2, best, best_tap);
}

}  // namespace ppo

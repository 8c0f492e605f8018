// OPT-IN reduced-precision form of the LDS-resident residual-block kernel (stack_fused.hip stack_tail_kernel, forward):
// every convolution as THREE bf16 MFMAs on (hi, lo) splits of both operands, float32 accumulation (gfx950).
//
//   x = hi + lo + r,  hi = bf16(x),  lo = bf16(x - hi),  |r| <= 2^-17 |x|
//   w * x  ~=  w_hi * x_hi  +  w_hi * x_lo  +  w_lo * x_hi          (dropped: w_lo * x_lo ~ 2^-16, the r terms ~ 2^-17)
//
// i.e. products carry ~16 bits - more than the TF32 (10-bit) products the reference's default `--precision=medium`
// lets cuDNN use on its own GPUs (/root/reference train.py:166-178), fewer than the exact f32 MFMA path that stays the
// default and the benchmark's arithmetic.  Why it can pay: v_mfma_f32_16x16x32_bf16 retires 16 x 16 x 32 MACs in 16
// cycles, v_mfma_f32_16x16x4_f32 16 x 16 x 4 in 32 - per tap and pixel tile of a 32-channel layer three bf16 MFMAs (48
// cycles) replace eight f32 ones (256 cycles), and the operand reads shrink from eight ds_read_b32 + eight v_med3 to two
// ds_read_b128 with no VALU instruction in the loop: the ReLU is applied, and the value split, ONCE by the epilogue
// that produces it.
//
// Layout.  A resident map is an array of per-pixel RECORDS over the zero-padded image ((H + 2) x (W + 2) pixels):
//   [ hi: 32 channels bf16 (64 B) | lo: 32 channels bf16 (64 B) | 16 B pad ]  = 144 B
// so the B fragment of v_mfma_f32_16x16x32_bf16 (lane = pixel l & 15, k = channels 8 (l >> 4) .. + 7) is ONE 16-byte
// read at a per-lane base + an immediate tap offset; 144 B = 36 banks puts the eight lanes of a ds_read_b128 pass on
// disjoint banks.  Two maps (block input / intermediate) ping-pong: 2 x 529 x 144 B = 149 KB at 21x21.  The residual
// stream itself (q = p + conv1(relu(conv0(relu(p))))) never touches LDS: each lane keeps the float32 values of the
// outputs it owns (pixel tile x 4 channels) in registers across the four layers, so nothing float32 has to be resident.
// The A fragments (weights, split on the host side of the launch by ppo_impala_stack_tail_pack_bf16x3) sit in registers
// per layer: 9 taps x (hi, lo) x 4 VGPRs = 72, as many as the f32 kernel's.
#include "common.h"

namespace ppo {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4b;

constexpr int kRec = 144;  // bytes per pixel record
#ifndef PPO_TUNE_W16_NW   // waves x pixel tiles per wave of the 16-channel window kernel (66 tiles at 25 x 42)
#define PPO_TUNE_W16_NW 12
#define PPO_TUNE_W16_MT 6
#endif

template <int C, int H, int W, int MT, int NW>
struct SplitCfg {
    static_assert(C == 32, "one K = 32 MFMA per tap");
    static constexpr int PW = W + 2, PH = H + 2;
    static constexpr int NPIX = H * W;
    static constexpr int MTILES = (NPIX + 15) / 16;
    static constexpr int WAVES = NW * 2;  // NW pixel-tile waves x 2 channel tiles
    static constexpr int MAP_BYTES = PW * PH * kRec;
    static constexpr size_t LDS_BYTES = (size_t)2 * MAP_BYTES;
    static_assert(MTILES <= MT * NW, "every pixel tile has a wave");
    static_assert(LDS_BYTES <= 160 * 1024, "two maps must fit the CU's LDS");
};

struct SplitTailArgs {
    const float *in;         // [n, C, H, W] forward: block input p; backward: g = d loss / d q1
    const bf16x8 *w;         // packed: [layer 4][channel tile 2][tap 9][hi, lo][lane 64] fragments of 8 bf16
                             // (backward: transposed + flipped, layers in processing order - see the pack entry point)
    const float *bias[4];    // forward only
    const float *mask[4];    // backward only: the forward pre-activation that gates each layer's output (a1, q0, a0, p)
    float *save[4];          // forward: a0, q0, a1, q1 (the last required, the others nullable: inference keeps q1 only)
                             // backward: da1, g1, da0, g0 (all required: the weight gradients read them)
    uint8_t *sign[4];        // SIGN MAPS [n, C / 4, H, W]: bit r of a byte = (channel 4 k + r of that pixel > 0), in the order of
                             // `mask` (a1, q0, a0, p).  Forward: written when non-null; backward: read INSTEAD of `mask` when
                             // non-null - the gates are 1 byte per 4 elements instead of 16 (the backward chain's HBM traffic
                             // is 9 maps otherwise, 4 of them read only for their sign)
    int n_images;
};

// round-to-nearest-even split (v_cvt_pk_bf16_f32 keeps NaNs NaNs)
__device__ __forceinline__ void split2(float x, __bf16 &hi, __bf16 &lo)
{
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

template <int C, int H, int W, int MT, int NW, bool BACKWARD>
__global__ __launch_bounds__(NW * 2 * 64) void stack_tail_bf16x3_kernel(SplitTailArgs a)
{
    using S = SplitCfg<C, H, W, MT, NW>;
    extern __shared__ __align__(16) unsigned char smem_b[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = tid >> 6, pw = wave % NW, ng = wave / NW;  // pixel-tile group, channel tile

    // zero both maps once: the halo records stay zero for every image and layer
    for (int i = tid * 16; i < 2 * S::MAP_BYTES; i += S::WAVES * 64 * 16) *reinterpret_cast<uint4 *>(smem_b + i) = uint4{0, 0, 0, 0};

    // per-lane constants of this wave's pixel tiles
    int rec0[MT];   // byte offset of the record of padded pixel (y, x) = the window origin of output pixel (y, x)
    int pix[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int p = (pw * MT + m) * 16 + l15;
        pix[m] = p;
        const int pc = p < S::NPIX ? p : 0;
        rec0[m] = ((pc / W) * S::PW + (pc % W)) * kRec;
    }
    const int ch0 = ng * 16 + 4 * g;  // this lane's four channels of every tile it owns
    __syncthreads();

    for (int img = blockIdx.x; img < a.n_images; img += gridDim.x) {
        const size_t img_off = (size_t)img * C * H * W;
        auto elem = [&](int m, int r) { return img_off + (size_t)(ch0 + r) * (H * W) + (pix[m] < S::NPIX ? pix[m] : 0); };
        auto sidx = [&](int m) { return ((size_t)img * (C / 4) + ch0 / 4) * (H * W) + (pix[m] < S::NPIX ? pix[m] : 0); };
        auto put_signs = [&](uint8_t *dst, const float (&v)[MT][4]) {
            if (!dst) return;
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (pix[m] < S::NPIX)
                    dst[sidx(m)] = (uint8_t)((v[m][0] > 0.f) | ((v[m][1] > 0.f) << 1) | ((v[m][2] > 0.f) << 2) | ((v[m][3] > 0.f) << 3));
        };
        // ---- the input map: float32 into the lanes that own it (pixel l15 of tile m, channels ch0 + r); it is the residual
        // stream from here on, its (ReLU'd: forward) split goes to map 0
        float xres[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) xres[m][r] = a.in[elem(m, r)];
        auto publish = [&](int map, const float (&v)[MT][4]) {
            // relu(v) (backward: v itself - there the ReLU sits behind the transposed convolution, as the gate of its output)
            // as (hi, lo) bf16 into the record of the (interior) padded pixel (y + 1, x + 1)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (pix[m] < S::NPIX) {
                    bf16x4 hi, lo;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        __bf16 h, l;
                        split2(BACKWARD ? v[m][r] : fmaxf(v[m][r], 0.f), h, l);
                        hi[r] = h, lo[r] = l;
                    }
                    unsigned char *rec = smem_b + map * S::MAP_BYTES + rec0[m] + (S::PW + 1) * kRec + ch0 * 2;
                    *reinterpret_cast<bf16x4 *>(rec) = hi;
                    *reinterpret_cast<bf16x4 *>(rec + 64) = lo;
                }
            }
        };
        auto store = [&](float *dst, const float (&v)[MT][4]) {
            if (!dst) return;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (pix[m] < S::NPIX) dst[elem(m, r)] = v[m][r];
        };
        __syncthreads();  // the previous image's last readers are done with map 0
        publish(0, xres);
        if constexpr (!BACKWARD) put_signs(a.sign[3], xres);

#pragma unroll 1
        for (int layer = 0; layer < 4; ++layer) {
            // this layer's A fragments: 9 taps x (hi, lo), 16 bytes per lane each
            bf16x8 whi[9], wlo[9];
            const bf16x8 *wl = a.w + ((size_t)(layer * 2 + ng) * 9 * 2) * 64 + lane;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                whi[t] = wl[(t * 2 + 0) * 64];
                wlo[t] = wl[(t * 2 + 1) * 64];
            }
            float bias_r[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) bias_r[r] = BACKWARD ? 0.f : a.bias[layer][ch0 + r];
            __syncthreads();  // the source map is complete
            const int odd = layer & 1;
            const unsigned char *src = smem_b + (odd ? S::MAP_BYTES : 0) + g * 16;
            f32x4b acc[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = f32x4b{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int toff = ((t / 3) * S::PW + (t % 3)) * kRec;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const bf16x8 bhi = *reinterpret_cast<const bf16x8 *>(src + rec0[m] + toff);
                    const bf16x8 blo = *reinterpret_cast<const bf16x8 *>(src + rec0[m] + toff + 64);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[t], bhi, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[t], blo, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[t], bhi, acc[m], 0, 0, 0);
                }
            }
            // ---- epilogue: lane holds pixel l15 x channels ch0 + r of each tile
            float y[MT][4];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) y[m][r] = acc[m][r] + bias_r[r];
            if constexpr (BACKWARD) {
                // the ReLU gate of this layer's output: its forward pre-activation (all loads first, then the selects), or its
                // sign map
                if (a.sign[layer]) {
                    unsigned bits[MT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) bits[m] = a.sign[layer][sidx(m)];
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[m][r] = (bits[m] >> r) & 1u ? y[m][r] : 0.f;
                } else {
                    float gate[MT][4];
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) gate[m][r] = a.mask[layer][elem(m, r)];
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[m][r] = gate[m][r] > 0.f ? y[m][r] : 0.f;
                }
            }
            if (!odd) {
                publish(1, y);  // the intermediate map is only ever a convolution input: its split is all that is kept
                store(a.save[layer], y);
                if constexpr (!BACKWARD) put_signs(a.sign[2 - layer], y);  // layer 0 -> a0 (sign[2]), layer 2 -> a1 (sign[0])
            } else {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xres[m][r] = y[m][r] + xres[m][r];
                if (layer == 1) publish(0, xres);  // (map 0's last readers - layer 0 - passed this layer's barrier)
                store(a.save[layer], xres);
                if constexpr (!BACKWARD)
                    if (layer == 1) put_signs(a.sign[1], xres);  // q0
            }
        }
    }
}

// ---- 16 channels (the 42x42 / 32x32 stack): K = 32 is TWO taps x 16 channels, and a map of per-pixel records no longer
// fits twice (42x42: 2 x 124 KB).  Every image is therefore cut into NWIN row windows, one workgroup each, that RECOMPUTE
// their halo: a window holds its OWN rows plus 4 rows towards the image's interior (one per layer of the chain), treats
// its cut edge as zero padding, and stores only rows it owns - a row k layers deep is wrong only within k rows of the cut,
// which the halo absorbs.  19 % more MFMA work at 42x42 (2 x 25 rows for 42), bought with MFMAs that are 5 x cheaper
// than the float32 ones; no exchange between workgroups, and a 128-image rollout group fills 256 CUs.
// Layout: hi and lo are separate planes of dense 32-byte records (16 channels), so the B fragment (lane = pixel l & 15;
// k = 8 (l >> 4) + j: tap 2 ks + (l >> 5), channels 8 ((l >> 4) & 1) + j) is one ds_read_b128 at a per-lane tap offset;
// eight consecutive pixels are 64 consecutive banks.  Maps ping-pong as above: 2 x 2 x 27 x 44 x 32 B = 149 KB.
template <int HI, int W, int NWIN, int R, int NW, int MT>
struct Split16Cfg {
    static constexpr int C = 16;
    static constexpr int OWN = HI / NWIN;                   // rows a window owns (stores)
    static_assert(HI % NWIN == 0 && (NWIN == 1 ? R == HI : R >= OWN + 4), "a cut needs four halo rows");
    static constexpr int PW = W + 2, PH = R + 2;
    static constexpr int NPIX = R * W;
    static constexpr int MTILES = (NPIX + 15) / 16;
    static constexpr int PLANE = PW * PH * 32;              // bytes: one of (hi, lo)
    static constexpr int MAP_BYTES = 2 * PLANE;
    static constexpr size_t LDS_BYTES = (size_t)2 * MAP_BYTES;
    static constexpr int KS = 5;                            // tap pairs (0,1) (2,3) (4,5) (6,7) (8,-)
    static_assert(MTILES <= MT * NW, "every pixel tile has a wave");
    static_assert(LDS_BYTES <= 160 * 1024, "two maps must fit the CU's LDS");
};

template <int HI, int W, int NWIN, int R, int NW, int MT, bool BACKWARD>
__global__ __launch_bounds__(NW * 64) void stack_win16_bf16x3_kernel(SplitTailArgs a)
{
    using S = Split16Cfg<HI, W, NWIN, R, NW, MT>;
    constexpr int C = 16;
    extern __shared__ __align__(16) unsigned char smem_b[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int i = tid * 16; i < 2 * S::MAP_BYTES; i += NW * 64 * 16) *reinterpret_cast<uint4 *>(smem_b + i) = uint4{0, 0, 0, 0};

    int rec0[MT], pix[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int p = (wave * MT + m) * 16 + l15;
        pix[m] = p;
        const int pc = p < S::NPIX ? p : 0;
        rec0[m] = ((pc / W) * S::PW + (pc % W)) * 32;
    }
    // this lane's tap of each pair and its channel half
    int tapoff[S::KS];
#pragma unroll
    for (int ks = 0; ks < S::KS; ++ks) {
        const int t = 2 * ks + (g >> 1) < 9 ? 2 * ks + (g >> 1) : 8;  // (the ninth pair's second tap has zero weights)
        tapoff[ks] = ((t / 3) * S::PW + (t % 3)) * 32 + (g & 1) * 16;
    }
    const int ch0 = 4 * g;  // the four output channels of this lane
    __syncthreads();

    for (int item = blockIdx.x; item < a.n_images * NWIN; item += gridDim.x) {
        const int img = item / NWIN, win = item % NWIN;
        const int own0 = win * S::OWN;
        const int r0 = NWIN == 1 ? 0 : (own0 - 4 < 0 ? 0 : (own0 - 4 > HI - R ? HI - R : own0 - 4));  // first image row of the window
        const size_t img_off = (size_t)img * C * HI * W + (size_t)r0 * W;
        auto elem = [&](int m, int r) { return img_off + (size_t)(ch0 + r) * (HI * W) + (pix[m] < S::NPIX ? pix[m] : 0); };
        auto owned = [&](int m) {
            const int y = r0 + pix[m] / W;
            return pix[m] < S::NPIX && y >= own0 && y < own0 + S::OWN;
        };
        // sign maps cover the image: a window reads its halo rows' bytes and writes the rows it owns
        auto sidx = [&](int m) { return ((size_t)img * (C / 4) + g) * (HI * W) + (size_t)r0 * W + (pix[m] < S::NPIX ? pix[m] : 0); };
        auto put_signs = [&](uint8_t *dst, const float (&v)[MT][4]) {
            if (!dst) return;
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (owned(m))
                    dst[sidx(m)] = (uint8_t)((v[m][0] > 0.f) | ((v[m][1] > 0.f) << 1) | ((v[m][2] > 0.f) << 2) | ((v[m][3] > 0.f) << 3));
        };
        float xres[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) xres[m][r] = a.in[elem(m, r)];
        auto publish = [&](int map, const float (&v)[MT][4]) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (pix[m] < S::NPIX) {
                    bf16x4 hi, lo;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        __bf16 h, l;
                        split2(BACKWARD ? v[m][r] : fmaxf(v[m][r], 0.f), h, l);
                        hi[r] = h, lo[r] = l;
                    }
                    unsigned char *rec = smem_b + map * S::MAP_BYTES + rec0[m] + (S::PW + 1) * 32 + ch0 * 2;
                    *reinterpret_cast<bf16x4 *>(rec) = hi;
                    *reinterpret_cast<bf16x4 *>(rec + S::PLANE) = lo;
                }
            }
        };
        auto store = [&](float *dst, const float (&v)[MT][4]) {
            if (!dst) return;
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (owned(m)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[elem(m, r)] = v[m][r];
                }
        };
        __syncthreads();  // the previous item's last readers are done with map 0
        publish(0, xres);
        if constexpr (!BACKWARD) put_signs(a.sign[3], xres);

#pragma unroll 1
        for (int layer = 0; layer < 4; ++layer) {
            bf16x8 whi[S::KS], wlo[S::KS];
            const bf16x8 *wl = a.w + (size_t)layer * S::KS * 2 * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < S::KS; ++ks) {
                whi[ks] = wl[(ks * 2 + 0) * 64];
                wlo[ks] = wl[(ks * 2 + 1) * 64];
            }
            float bias_r[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) bias_r[r] = BACKWARD ? 0.f : a.bias[layer][ch0 + r];
            __syncthreads();  // the source map is complete
            const int odd = layer & 1;
            const unsigned char *src = smem_b + (odd ? S::MAP_BYTES : 0);
            f32x4b acc[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = f32x4b{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < S::KS; ++ks) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const bf16x8 bhi = *reinterpret_cast<const bf16x8 *>(src + rec0[m] + tapoff[ks]);
                    const bf16x8 blo = *reinterpret_cast<const bf16x8 *>(src + rec0[m] + tapoff[ks] + S::PLANE);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[ks], bhi, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[ks], blo, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[ks], bhi, acc[m], 0, 0, 0);
                }
            }
            float y[MT][4];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) y[m][r] = acc[m][r] + bias_r[r];
            if constexpr (BACKWARD) {
                if (a.sign[layer]) {
                    unsigned bits[MT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) bits[m] = a.sign[layer][sidx(m)];
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[m][r] = (bits[m] >> r) & 1u ? y[m][r] : 0.f;
                } else {
                    float gate[MT][4];
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) gate[m][r] = a.mask[layer][elem(m, r)];
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[m][r] = gate[m][r] > 0.f ? y[m][r] : 0.f;
                }
            }
            if (!odd) {
                publish(1, y);
                store(a.save[layer], y);
                if constexpr (!BACKWARD) put_signs(a.sign[2 - layer], y);
            } else {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xres[m][r] = y[m][r] + xres[m][r];
                if (layer == 1) publish(0, xres);
                store(a.save[layer], xres);
                if constexpr (!BACKWARD)
                    if (layer == 1) put_signs(a.sign[1], xres);
            }
        }
    }
}

// weights [cout 32][cin 32][3][3] float32 -> A fragments: lane (row = l & 15, k = 8 (l >> 4) + j).  Forward: row = output
// channel, k = input channel, tap as stored.  Transposed (backward-data): row = INPUT channel, k = output channel, tap
// flipped - dX[i] = sum_{o, taps} W[o][i][2 - ky][2 - kx] dY[o] is a forward convolution with those weights.
constexpr int kMaxSplitJobs = 8;
struct SplitPackJobs {
    const float *w[kMaxSplitJobs][4];
    __bf16 *packed[kMaxSplitJobs];
    int transposed[kMaxSplitJobs];
    int channels[kMaxSplitJobs];
    int n;
};
__global__ __launch_bounds__(256) void pack_bf16x3_kernel(const SplitPackJobs jobs)
{
    const int job = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (layer, ng, tap, lane)
    if (job >= jobs.n) return;
    if (jobs.channels[job] == 16) {
        // [layer 4][tap pair 5][hi, lo][lane 64]: row = l & 15, k = 8 (l >> 4) + j = (tap 2 ks + (l >> 5), channel 8 ((l >> 4) & 1) + j)
        if (i >= 4 * 5 * 64) return;
        const int lane = i & 63, ks = (i >> 6) % 5, layer = i / (64 * 5);
        const float *w = jobs.w[job][layer];
        const int row = lane & 15, gq = lane >> 4, t = 2 * ks + (gq >> 1), c0 = 8 * (gq & 1);
        __bf16 *hi = jobs.packed[job] + (((size_t)layer * 5 + ks) * 2 + 0) * 64 * 8 + lane * 8;
        __bf16 *lo = jobs.packed[job] + (((size_t)layer * 5 + ks) * 2 + 1) * 64 * 8 + lane * 8;
        for (int j = 0; j < 8; ++j) {
            float v = 0.f;
            if (t < 9) v = jobs.transposed[job] ? w[((size_t)(c0 + j) * 16 + row) * 9 + (8 - t)] : w[((size_t)row * 16 + c0 + j) * 9 + t];
            split2(v, hi[j], lo[j]);
        }
        return;
    }
    if (i >= 4 * 2 * 9 * 64) return;
    const int lane = i & 63, t = (i >> 6) % 9, ng = (i / (64 * 9)) & 1, layer = i / (64 * 9 * 2);
    const float *w = jobs.w[job][layer];
    __bf16 *packed = jobs.packed[job];
    const int transposed = jobs.transposed[job];
    const int row = ng * 16 + (lane & 15), k0 = 8 * (lane >> 4);
    __bf16 *hi = packed + (((size_t)(layer * 2 + ng) * 9 + t) * 2 + 0) * 64 * 8 + lane * 8;
    __bf16 *lo = packed + (((size_t)(layer * 2 + ng) * 9 + t) * 2 + 1) * 64 * 8 + lane * 8;
    for (int j = 0; j < 8; ++j) {
        const float v = transposed ? w[((size_t)(k0 + j) * 32 + row) * 9 + (8 - t)] : w[((size_t)row * 32 + k0 + j) * 9 + t];
        split2(v, hi[j], lo[j]);
    }
}

template <int C, int H, int W, int MT, int NW, bool BACKWARD>
int launch_split_tail(const SplitTailArgs &args, hipStream_t st)
{
    using S = SplitCfg<C, H, W, MT, NW>;
    auto kern = stack_tail_bf16x3_kernel<C, H, W, MT, NW, BACKWARD>;
    static bool ready = false;
    if (!ready) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)S::LDS_BYTES);
        if (e != hipSuccess) return fail(PPO_E_HIP, "stack_tail_bf16x3: hipFuncSetAttribute: %s", hipGetErrorString(e));
        ready = true;
    }
    int grid = args.n_images < 256 ? args.n_images : 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(S::WAVES * 64), S::LDS_BYTES, st, args);
    return check_launch("stack_tail_bf16x3_kernel");
}

template <int HI, int W, int NWIN, int R, int NW, int MT, bool BACKWARD>
int launch_split_win16(const SplitTailArgs &args, hipStream_t st)
{
    using S = Split16Cfg<HI, W, NWIN, R, NW, MT>;
    auto kern = stack_win16_bf16x3_kernel<HI, W, NWIN, R, NW, MT, BACKWARD>;
    static bool ready = false;
    if (!ready) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)S::LDS_BYTES);
        if (e != hipSuccess) return fail(PPO_E_HIP, "stack_win16_bf16x3: hipFuncSetAttribute: %s", hipGetErrorString(e));
        ready = true;
    }
    const int items = args.n_images * NWIN;
    const int per_cu = S::LDS_BYTES <= 80 * 1024 ? 2 : 1;
    const int grid = items < 256 * per_cu ? items : 256 * per_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), S::LDS_BYTES, st, args);
    return check_launch("stack_win16_bf16x3_kernel");
}

int split_tail(const char *who, bool backward, const float *in, const void *packed, const float *const *bias_or_mask,
               float *s0, float *s1, float *s2, float *s3, int n_images, int channels, int h, int w, void *stream,
               uint8_t *const *signs = nullptr)
{
    if (n_images < 0) return fail(PPO_E_INVALID, "%s: negative batch", who);
    if (n_images == 0) return PPO_OK;
    if (!in || !packed || (!bias_or_mask && !(backward && signs)) || !s3 || !aligned(packed, 16))
        return fail(PPO_E_INVALID, "%s: null or misaligned pointer", who);
    if (backward && (!s0 || !s1 || !s2)) return fail(PPO_E_INVALID, "%s: the backward pass writes all four gradient maps", who);
    SplitTailArgs args;
    args.in = in, args.w = static_cast<const bf16x8 *>(packed), args.n_images = n_images;
    for (int l = 0; l < 4; ++l) {
        args.sign[l] = signs ? signs[l] : nullptr;
        if (backward && signs) {  // the sign maps stand in for the float gate maps
            if (!signs[l]) return fail(PPO_E_INVALID, "%s: null sign map of layer %d", who, l);
            args.bias[l] = args.mask[l] = nullptr;
            continue;
        }
        if (!bias_or_mask[l]) return fail(PPO_E_INVALID, "%s: null bias / mask of layer %d", who, l);
        args.bias[l] = backward ? nullptr : bias_or_mask[l];
        args.mask[l] = backward ? bias_or_mask[l] : nullptr;
    }
    args.save[0] = s0, args.save[1] = s1, args.save[2] = s2, args.save[3] = s3;
    hipStream_t st = as_stream(stream);
    if (channels == 32 && h == 21 && w == 21)
        return backward ? launch_split_tail<32, 21, 21, 7, 4, true>(args, st) : launch_split_tail<32, 21, 21, 7, 4, false>(args, st);
    if (channels == 32 && h == 11 && w == 11)
        return backward ? launch_split_tail<32, 11, 11, 2, 4, true>(args, st) : launch_split_tail<32, 11, 11, 2, 4, false>(args, st);
    // the procgen-shaped net (64x64 observations: 32x32, 16x16 and 8x8 maps)
    if (channels == 32 && h == 16 && w == 16)
        return backward ? launch_split_tail<32, 16, 16, 4, 4, true>(args, st) : launch_split_tail<32, 16, 16, 4, 4, false>(args, st);
    if (channels == 32 && h == 8 && w == 8)
        return backward ? launch_split_tail<32, 8, 8, 1, 4, true>(args, st) : launch_split_tail<32, 8, 8, 1, 4, false>(args, st);
    if (channels == 16 && h == 32 && w == 32)
        return backward ? launch_split_win16<32, 32, 2, 20, 10, 4, true>(args, st) : launch_split_win16<32, 32, 2, 20, 10, 4, false>(args, st);
    if (channels == 16 && h == 42 && w == 42)
        return backward ? launch_split_win16<42, 42, 2, 25, PPO_TUNE_W16_NW, PPO_TUNE_W16_MT, true>(args, st)
                        : launch_split_win16<42, 42, 2, 25, PPO_TUNE_W16_NW, PPO_TUNE_W16_MT, false>(args, st);
    return fail(PPO_E_INVALID, "%s: no kernel for %d channels at %dx%d", who, channels, h, w);
}

}  // namespace
}  // namespace ppo

extern "C" size_t ppo_impala_stack_tail_bf16x3_packed_bytes(void) { return (size_t)4 * 2 * 9 * 2 * 64 * 8 * sizeof(uint16_t); }

extern "C" int ppo_impala_stack_tail_bf16x3_supported(int channels, int h, int w)
{
    return (channels == 32 && ((h == 21 && w == 21) || (h == 11 && w == 11) || (h == 16 && w == 16) || (h == 8 && w == 8))) ||
           (channels == 16 && ((h == 42 && w == 42) || (h == 32 && w == 32)));
}

extern "C" int ppo_impala_stack_tail_pack_bf16x3_jobs(const ppo_split_pack_job *jobs, int n_jobs, void *stream)
{
    using namespace ppo;
    if (n_jobs < 0 || n_jobs > kMaxSplitJobs) return fail(PPO_E_INVALID, "ppo_impala_stack_tail_pack_bf16x3_jobs: 0 .. %d jobs", kMaxSplitJobs);
    if (n_jobs == 0) return PPO_OK;
    if (!jobs) return fail(PPO_E_INVALID, "ppo_impala_stack_tail_pack_bf16x3_jobs: null job table");
    SplitPackJobs t{};
    t.n = n_jobs;
    for (int j = 0; j < n_jobs; ++j) {
        if (jobs[j].channels != 32 && jobs[j].channels != 16) return fail(PPO_E_INVALID, "ppo_impala_stack_tail_pack_bf16x3_jobs: 16 or 32 channels");
        if (!jobs[j].packed || !aligned(jobs[j].packed, 16)) return fail(PPO_E_INVALID, "ppo_impala_stack_tail_pack_bf16x3_jobs: null or misaligned packed buffer");
        for (int l = 0; l < 4; ++l) {
            if (!jobs[j].weights[l]) return fail(PPO_E_INVALID, "ppo_impala_stack_tail_pack_bf16x3_jobs: null weights");
            t.w[j][l] = jobs[j].weights[l];
        }
        t.packed[j] = static_cast<__bf16 *>(jobs[j].packed);
        t.transposed[j] = jobs[j].transposed;
        t.channels[j] = jobs[j].channels;
    }
    hipLaunchKernelGGL(pack_bf16x3_kernel, dim3((4 * 2 * 9 * 64 + 255) / 256, n_jobs), dim3(256), 0, as_stream(stream), t);
    return check_launch("pack_bf16x3_kernel");
}

extern "C" int ppo_impala_stack_tail_pack_bf16x3(const float *const *weights, void *packed, int channels, int transposed,
                                                 void *stream)
{
    if (!weights) return ppo::fail(PPO_E_INVALID, "ppo_impala_stack_tail_pack_bf16x3: null pointer");
    ppo_split_pack_job job{{weights[0], weights[1], weights[2], weights[3]}, packed, channels, transposed};
    return ppo_impala_stack_tail_pack_bf16x3_jobs(&job, 1, stream);
}

extern "C" int ppo_impala_stack_tail_forward_bf16x3(const float *in, const void *packed, const float *const *biases, float *a0,
                                                    float *q0, float *a1, float *q1, int n_images, int channels, int h, int w,
                                                    void *stream)
{
    return ppo::split_tail("ppo_impala_stack_tail_forward_bf16x3", false, in, packed, biases, a0, q0, a1, q1, n_images, channels, h,
                           w, stream);
}

extern "C" int ppo_impala_stack_tail_backward_bf16x3(const float *g, const void *packed_t, const float *const *masks, float *da1,
                                                     float *g1, float *da0, float *g0, int n_images, int channels, int h, int w,
                                                     void *stream)
{
    return ppo::split_tail("ppo_impala_stack_tail_backward_bf16x3", true, g, packed_t, masks, da1, g1, da0, g0, n_images, channels,
                           h, w, stream);
}

extern "C" size_t ppo_impala_stack_tail_bf16x3_sign_bytes(int n_images, int channels, int h, int w)
{
    return (size_t)n_images * (channels / 4) * h * w;
}

extern "C" int ppo_impala_stack_tail_forward_signs_bf16x3(const float *in, const void *packed, const float *const *biases, float *a0,
                                                          float *q0, float *a1, float *q1, uint8_t *const *signs, int n_images,
                                                          int channels, int h, int w, void *stream)
{
    if (!signs) return ppo::fail(PPO_E_INVALID, "ppo_impala_stack_tail_forward_signs_bf16x3: null sign table");
    return ppo::split_tail("ppo_impala_stack_tail_forward_signs_bf16x3", false, in, packed, biases, a0, q0, a1, q1, n_images, channels,
                           h, w, stream, signs);
}

extern "C" int ppo_impala_stack_tail_backward_signs_bf16x3(const float *g, const void *packed_t, const uint8_t *const *signs, float *da1,
                                                           float *g1, float *da0, float *g0, int n_images, int channels, int h, int w,
                                                           void *stream)
{
    if (!signs) return ppo::fail(PPO_E_INVALID, "ppo_impala_stack_tail_backward_signs_bf16x3: null sign table");
    return ppo::split_tail("ppo_impala_stack_tail_backward_signs_bf16x3", true, g, packed_t, nullptr, da1, g1, da0, g0, n_images,
                           channels, h, w, stream, const_cast<uint8_t *const *>(signs));
}

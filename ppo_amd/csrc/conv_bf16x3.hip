// OPT-IN reduced-precision 3x3 convolution (`--precision=medium|low`): one layer, stride 1, zero padding 1, every product
// as THREE bf16 MFMAs on (hi, lo) splits of both operands with float32 accumulation (see stack_bf16x3.hip for the
// arithmetic and the error it leaves: ~16-bit products).  It stands in for conv3x3_kernel where a convolution is NOT part of
// an LDS-resident chain: the stack-first convolutions (rl/impala.py:96, forward: bias, raw input) and their backward-data
// form (the same operator on flipped, transposed weights: no bias).  The exact float32 kernels stay the default.
//
// One 256-thread workgroup walks (image, band of TR output rows) items.  The band's input rows (TR + 2) sit in LDS as dense
// 32-byte records of 16 channels, hi and lo in separate planes, one zero record either side of each row - the layout of
// stack_bf16x3.hip's 16-channel kernel, so the B fragment of v_mfma_f32_16x16x32_bf16 (lane = pixel l & 15, 8 channels
// of one tap) is one ds_read_b128 at a per-lane constant offset.  K = 32 is one tap x 32 channels, or two taps x 16.  The A
// fragments (weights, packed by ppo_conv3x3_pack_bf16x3_jobs) are loaded ONCE per workgroup and stay in registers over
// all its items; the next item's input is in flight in registers while this one is computed (the staging of
// wgrad_bf16x3.hip); outputs leave as float32 straight from the accumulators (16 consecutive pixels per channel and
// store instruction).  Up to three workgroups per CU, so staging, MFMAs and stores of different items overlap.
#include "common.h"

namespace ppo {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4c;

constexpr int kConvWaves = 4;
constexpr int kFarOutsideC = (int)0x80000000u;  // a byte offset beyond the descriptor that stays beyond it after a channel offset

constexpr int cmaxc(int a, int b) { return a > b ? a : b; }

// NS = 2: (hi, lo) splits, three products (the opt-in precision mode).  NS = 3: (hi, mid, lo) - 8 + 8 + 8 mantissa bits, i.e.
// the whole float32 significand - with six of the nine partial products (the dropped ones <= 2^-24 of a product): float32-
// ACCURATE convolution on the bf16 MFMA, a measured prototype for the exact-precision path, reached
// through ppo_conv3x3_bf16_split only.
template <int CI, int CO, int H, int W, int TR, bool POOL = false, int NS = 2>
struct SplitConvCfg {
    static_assert((CI == 16 || CI == 32) && (CO == 16 || CO == 32), "16 or 32 channels");
    static constexpr int NG = CI / 16, MT = CO / 16;
    static constexpr int KS = CI == 32 ? 9 : 5;               // K steps: a tap of 32 channels, or a pair of taps of 16
    static constexpr int RW = W + 2;
    // POOL (3x3 / stride 2 / pad 1 max-pool of the result, F.max_pool2d at rl/impala.py:105): a band is PR pooled rows =
    // the 2 PR + 1 pre-pool rows they cover (one row recomputed between neighbouring bands), parked in LDS as float32
    static_assert(!POOL || TR % 2 == 1, "a pooled band covers 2 PR + 1 convolution rows");
    static constexpr int PR = (TR - 1) / 2;
    static constexpr int HO = (H + 1) / 2, WO = (W + 1) / 2;
    static constexpr int NB = POOL ? (HO + PR - 1) / PR : (H + TR - 1) / TR;
    static constexpr int XREC = (TR + 2) * RW + 2;
    static constexpr int PLANE = XREC * 32;                   // bytes of one (group, hi | lo) plane
    static constexpr int X_BYTES = NG * NS * PLANE;
    static constexpr int PSTR = CO + 4;                        // floats per pre-pool pixel: 36 / 20 banks apart, 16-byte accesses conflict-free
    static constexpr int PRE_BYTES = POOL ? TR * W * PSTR * 4 : 0;
    static constexpr int LDS_BYTES = X_BYTES + PRE_BYTES;
    static constexpr int XPIX = (TR + 2) * W;                 // pixels staged per channel and band
    static constexpr int XIT = (XPIX + 63) / 64;
    static constexpr int NT = (TR * W + 15) / 16;             // output pixel tiles of a band
    static constexpr int NPW = kConvWaves / MT;               // waves that share the pixel tiles of one output-channel tile
    static constexpr int TPW = (NT + NPW - 1) / NPW;          // tiles per wave
};

struct SplitConvArgs {
    const float *in;     // [n, CI, H, W]
    const bf16x8 *w;     // [MT][KS][hi, lo][lane 64] fragments of 8 bf16
    const float *bias;   // [CO] or null
    float *out;          // [n, CO, H, W]; POOL: [n, CO, HO, WO]
    uint8_t *argmax;     // POOL: [n, CO, HO, WO] the winning tap ky * 3 + kx (ppo_maxpool3x3s2_forward_f32's record), nullable
    float floor;         // lower clamp of the input on load: 0 = ReLU, -inf = raw
    int n_images;
};

template <int NS>
__device__ __forceinline__ void store_split4c(unsigned char *p, int plane, const float (&v)[4])
{
    bf16x4 hi, mid, lo;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        hi[r] = (__bf16)v[r];
        const float r1 = v[r] - (float)hi[r];
        mid[r] = (__bf16)r1;
        lo[r] = (__bf16)(r1 - (float)mid[r]);
    }
    *reinterpret_cast<bf16x4 *>(p) = hi;
    *reinterpret_cast<bf16x4 *>(p + plane) = mid;  // NS = 2: this IS the low part (bf16 of the first remainder)
    if (NS == 3) *reinterpret_cast<bf16x4 *>(p + 2 * plane) = lo;
}

template <int CI, int CO, int H, int W, int TR, bool POOL, int NS>
__global__ __launch_bounds__(kConvWaves * 64) void conv3x3_bf16x3_kernel(SplitConvArgs a)
{
    using C = SplitConvCfg<CI, CO, H, W, TR, POOL, NS>;
    constexpr int HW = H * W;
    extern __shared__ __align__(16) unsigned char smem_c[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = wave % C::MT, pw = wave / C::MT;

    for (int i = tid * 16; i < C::X_BYTES; i += kConvWaves * 64 * 16) *reinterpret_cast<uint4 *>(smem_c + i) = uint4{0, 0, 0, 0};

    // ---- this wave's A fragments, once
    bf16x8 wsp[NS][C::KS];  // [part: 0 = hi ...][K step]
    {
        const bf16x8 *wl = a.w + (size_t)mt * C::KS * NS * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
            for (int q = 0; q < NS; ++q) wsp[q][ks] = wl[(ks * NS + q) * 64];
    }
    const int ch0 = mt * 16 + 4 * g;  // the four output channels of this lane
    float bias_r[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias_r[r] = a.bias ? a.bias[ch0 + r] : 0.f;

    // ---- K-loop constants: byte offset of this lane's tap / channel half inside the planes, window origin of its pixel
    int tapoff[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
        int t = ks, grp = g >> 1;
        if (CI == 16) {
            t = 2 * ks + (g >> 1) < 9 ? 2 * ks + (g >> 1) : 8;  // (the ninth pair's second tap has zero weights)
            grp = 0;
        }
        tapoff[ks] = grp * NS * C::PLANE + ((t / 3) * C::RW + (t % 3)) * 32 + (g & 1) * 16;
    }
    int rec0[C::TPW], pix[C::TPW];
#pragma unroll
    for (int t = 0; t < C::TPW; ++t) {
        const int p = ((pw * C::TPW + t) * 16) + l15;
        pix[t] = p;
        const int pc = p < TR * W ? p : 0;
        rec0[t] = ((pc / W) * C::RW + (pc % W)) * 32;
    }

    // ---- staging constants (as wgrad_bf16x3.hip): 16 consecutive pixels of one channel per 16-lane group
    const int quad = lane >> 4, pl = lane & 15;
    int x_rec[C::XIT];
#pragma unroll
    for (int i = 0; i < C::XIT; ++i) {
        const int p = 64 * i + 16 * wave + pl;
        x_rec[i] = ((p / W) * C::RW + p % W + 1) * 32 + quad * 8;
    }
    float xv[C::XIT][C::NG][4] = {};
    const int n_items = a.n_images * C::NB;
    auto issue = [&](int item) {
        const int img = item / C::NB, band = item % C::NB;
        const __amdgpu_buffer_rsrc_t xb = buffer_of(a.in + (size_t)img * CI * HW);
        const int x0 = ((POOL ? 2 * C::PR * band - 1 : band * TR) - 1) * W;  // first input row = first output row - 1
#pragma unroll
        for (int i = 0; i < C::XIT; ++i) {
            const int p = 64 * i + 16 * wave + pl;
            const int gp = x0 + p;
            int base = (((unsigned)gp < (unsigned)HW) & (p < C::XPIX)) ? (quad * 4 * HW + gp) * 4 : kFarOutsideC;
            asm volatile("" : "+v"(base));
#pragma unroll
            for (int gi = 0; gi < C::NG; ++gi)
#pragma unroll
                for (int r = 0; r < 4; ++r) xv[i][gi][r] = buffer_f32(xb, base + (gi * 16 + r) * HW * 4);
        }
    };
    auto publish = [&]() {
#pragma unroll
        for (int i = 0; i < C::XIT; ++i) {
            if (64 * i + 64 <= C::XPIX || 64 * i + 16 * wave + pl < C::XPIX) {
#pragma unroll
                for (int gi = 0; gi < C::NG; ++gi) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_fmed3f(xv[i][gi][r], a.floor, __builtin_inff());
                    store_split4c<NS>(smem_c + gi * NS * C::PLANE + x_rec[i], C::PLANE, v);
                }
            }
        }
    };

    int item = blockIdx.x;
    if (item < n_items) issue(item);
    __syncthreads();  // the zero fill is complete
    for (; item < n_items; item += gridDim.x) {
        publish();
        __syncthreads();  // the band is complete
        if (item + (int)gridDim.x < n_items) issue(item + gridDim.x);
        f32x4c acc[C::TPW];
#pragma unroll
        for (int t = 0; t < C::TPW; ++t) acc[t] = f32x4c{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
#pragma unroll
            for (int t = 0; t < C::TPW; ++t) {
                bf16x8 b[NS];
#pragma unroll
                for (int q = 0; q < NS; ++q) b[q] = *reinterpret_cast<const bf16x8 *>(smem_c + rec0[t] + tapoff[ks] + q * C::PLANE);
                // partial products w_i x_j with i + j < NS, smallest first
#ifndef PPO_TUNE_SPLIT3_ALL  // (timing / error aid: all nine products of the three-part form)
#define PPO_TUNE_SPLIT3_ALL 0
#endif
                constexpr int TOP = (NS == 3 && PPO_TUNE_SPLIT3_ALL) ? 2 * (NS - 1) : NS - 1;
#pragma unroll
                for (int sum = TOP; sum >= 0; --sum)
#pragma unroll
                    for (int i = NS - 1; i >= 0; --i)  // (NS = 2: w_lo x_hi, w_hi x_lo, w_hi x_hi)
                        if (i <= sum && sum - i < NS)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wsp[i][ks], b[sum - i], acc[t], 0, 0, 0);
            }
        }
        const int img = item / C::NB, band = item % C::NB;
        if constexpr (!POOL) {
            // ---- outputs: lane = pixel l15 of the tile x channels ch0 + r
            float *__restrict__ dst = a.out + (size_t)img * CO * HW + (size_t)ch0 * HW + band * TR * W;
            const int left = HW - band * TR * W;  // pixels of the image from this band's first one on (a last, shorter band)
#pragma unroll
            for (int t = 0; t < C::TPW; ++t) {
                if (pix[t] < TR * W && pix[t] < left) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[(size_t)r * HW + pix[t]] = acc[t][r] + bias_r[r];
                }
            }
            __syncthreads();  // the band's readers are done
        } else {
            // ---- the band's pre-pool rows to LDS, pixel-major [row][x][channel] (a lane's four channels are one 16-byte store),
            // then the 3x3 / stride 2 windows over them, four channels per thread
            float *s_pre = reinterpret_cast<float *>(smem_c + C::X_BYTES);
#pragma unroll
            for (int t = 0; t < C::TPW; ++t) {
                if (pix[t] < TR * W)
                    *reinterpret_cast<float4 *>(s_pre + pix[t] * C::PSTR + ch0) =
                        make_float4(acc[t][0] + bias_r[0], acc[t][1] + bias_r[1], acc[t][2] + bias_r[2], acc[t][3] + bias_r[3]);
            }
            __syncthreads();  // pre-pool rows complete; every wave is through with the input planes
            const int yf = 2 * C::PR * band - 1;  // image row of the band's first pre-pool row
            constexpr int Q = CO / 4;
            for (int o = tid; o < Q * C::PR * C::WO; o += kConvWaves * 64) {
                const int c4 = o % Q, pos = o / Q;
                const int oyl = pos / C::WO, ox = pos % C::WO, oy = C::PR * band + oyl;
                if (oy >= C::HO) continue;
                float best[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
                int best_tap[4] = {0, 0, 0, 0};
                bool found = false;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int ly = 2 * oyl + ky, iy = yf + ly;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int ix = 2 * ox - 1 + kx;
                        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                            const float4 v4 = *reinterpret_cast<const float4 *>(s_pre + (ly * W + ix) * C::PSTR + c4 * 4);
                            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                if (!found || v[r] > best[r] || v[r] != v[r]) {  // strict '>' scan: ties go to the first tap, as pool.hip / PyTorch
                                    best[r] = v[r];
                                    best_tap[r] = ky * 3 + kx;
                                }
                            }
                            found = true;
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t at = ((size_t)img * CO + c4 * 4 + r) * (C::HO * C::WO) + (size_t)oy * C::WO + ox;
                    a.out[at] = best[r];
                    if (a.argmax) a.argmax[at] = (uint8_t)best_tap[r];
                }
            }
            // (no barrier here: the next item's pre-pool stores come after its own "band is complete" barrier)
        }
    }
}

// weights [cout][cin][3][3] float32 -> A fragments [output tile][K step][hi, lo][lane][8]: row = l & 15, k = 8 (l >> 4) + j.
// Forward: row = output channel, k = (tap, input channel).  Transposed (backward-data): row = INPUT channel of the layer,
// k = (flipped tap, output channel of the layer): dX[i] = sum_{o, taps} W[o][i][2 - ky][2 - kx] dY[o].
constexpr int kMaxConvPackJobs = 8;
struct ConvPackJobs {
    const float *w[kMaxConvPackJobs];
    __bf16 *packed[kMaxConvPackJobs];
    int cin[kMaxConvPackJobs], cout[kMaxConvPackJobs], transposed[kMaxConvPackJobs], ns[kMaxConvPackJobs];
    int n;
};
__global__ __launch_bounds__(256) void conv_pack_bf16x3_kernel(const ConvPackJobs jobs)
{
    const int job = blockIdx.y;
    if (job >= jobs.n) return;
    const int cin = jobs.cin[job], cout = jobs.cout[job], tr = jobs.transposed[job];
    const int ci_op = tr ? cout : cin, co_op = tr ? cin : cout;  // channels of the OPERATOR the fragments serve
    const int ks_n = ci_op == 32 ? 9 : 5, mt_n = co_op / 16;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (output tile, K step, lane)
    if (i >= mt_n * ks_n * 64) return;
    const int lane = i & 63, ks = (i >> 6) % ks_n, mt = i / (64 * ks_n);
    const int row = mt * 16 + (lane & 15), gq = lane >> 4;
    const int t = ci_op == 32 ? ks : 2 * ks + (gq >> 1);
    const int c0 = ci_op == 32 ? 8 * gq : 8 * (gq & 1);
    const int ns = jobs.ns[job];
    __bf16 *part = jobs.packed[job] + (((size_t)mt * ks_n + ks) * ns) * 64 * 8 + lane * 8;  // parts 64 * 8 elements apart
    const float *w = jobs.w[job];
    for (int j = 0; j < 8; ++j) {
        float v = 0.f;
        if (t < 9) v = tr ? w[((size_t)(c0 + j) * cin + row) * 9 + (8 - t)] : w[((size_t)row * cin + c0 + j) * 9 + t];
        for (int q = 0; q < ns; ++q) {  // hi, then bf16 of each successive remainder
            const __bf16 h = (__bf16)v;
            part[q * 64 * 8 + j] = h;
            v -= (float)h;
        }
    }
}

template <int CI, int CO, int H, int W, int TR, bool POOL = false, int NS = 2>
int launch_split_conv(const SplitConvArgs &args, hipStream_t st)
{
    using C = SplitConvCfg<CI, CO, H, W, TR, POOL, NS>;
    auto kern = conv3x3_bf16x3_kernel<CI, CO, H, W, TR, POOL, NS>;
    static int per_cu = 0;
    if (!per_cu) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)C::LDS_BYTES);
        if (e != hipSuccess) return fail(PPO_E_HIP, "conv3x3_bf16x3: hipFuncSetAttribute: %s", hipGetErrorString(e));
        int occ = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(kern), kConvWaves * 64, C::LDS_BYTES);
        if (e != hipSuccess || occ < 1) return fail(PPO_E_HIP, "conv3x3_bf16x3: occupancy query: %s", hipGetErrorString(e));
        per_cu = occ > 4 ? 4 : occ;
    }
    const int n_items = args.n_images * C::NB;
    const int grid = n_items < 256 * per_cu ? n_items : 256 * per_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kConvWaves * 64), C::LDS_BYTES, st, args);
    return check_launch("conv3x3_bf16x3_kernel");
}

// (operator input channels, operator output channels, h, w, band rows)
#define PPO_SPLIT_CONV_GEOMETRIES(X) \
    X(16, 32, 42, 42, 7)             \
    X(32, 16, 42, 42, 7)             \
    X(32, 32, 21, 21, 7)             \
    X(16, 32, 32, 32, 8)             \
    X(32, 16, 32, 32, 8)             \
    X(32, 32, 16, 16, 8)
// with the max-pool: (input channels, output channels, h, w, 2 x pooled rows per band + 1)
#define PPO_SPLIT_CONV_POOL_GEOMETRIES(X) \
    X(16, 32, 42, 42, 7)                  \
    X(32, 32, 21, 21, 7)                  \
    X(16, 32, 32, 32, 9)                  \
    X(32, 32, 16, 16, 9)

}  // namespace
}  // namespace ppo

extern "C" int ppo_conv3x3_bf16x3_supported(int cin, int cout, int h, int w)
{
#define X(CI, CO, HH, WW, TR) \
    if (cin == CI && cout == CO && h == HH && w == WW) return 1;
    PPO_SPLIT_CONV_GEOMETRIES(X)
#undef X
    return 0;
}

extern "C" size_t ppo_conv3x3_bf16x3_packed_bytes(int cin, int cout)
{
    // the larger of the forward and the transposed packing: [tiles of the operator's outputs][K steps][hi, lo][64 lanes][16 B]
    const size_t fwd = (size_t)(cout / 16) * (cin == 32 ? 9 : 5) * 2 * 64 * 16;
    const size_t bwd = (size_t)(cin / 16) * (cout == 32 ? 9 : 5) * 2 * 64 * 16;
    return fwd > bwd ? fwd : bwd;
}

extern "C" int ppo_conv3x3_pack_bf16x3_jobs(const ppo_conv_pack_job *jobs, int n_jobs, void *stream)
{
    using namespace ppo;
    const char *who = "ppo_conv3x3_pack_bf16x3_jobs";
    if (n_jobs < 0 || n_jobs > kMaxConvPackJobs) return fail(PPO_E_INVALID, "%s: 0 .. %d jobs", who, kMaxConvPackJobs);
    if (n_jobs == 0) return PPO_OK;
    if (!jobs) return fail(PPO_E_INVALID, "%s: null job table", who);
    ConvPackJobs t{};
    t.n = n_jobs;
    for (int j = 0; j < n_jobs; ++j) {
        const ppo_conv_pack_job &q = jobs[j];
        if ((q.cin != 16 && q.cin != 32) || (q.cout != 16 && q.cout != 32)) return fail(PPO_E_INVALID, "%s: 16 or 32 channels", who);
        if (!q.weight || !q.packed || !aligned(q.packed, 16)) return fail(PPO_E_INVALID, "%s: null or misaligned pointer", who);
        t.w[j] = q.weight, t.packed[j] = static_cast<__bf16 *>(q.packed);
        t.cin[j] = q.cin, t.cout[j] = q.cout, t.transposed[j] = q.transposed, t.ns[j] = 2;
    }
    hipLaunchKernelGGL(conv_pack_bf16x3_kernel, dim3((2 * 9 * 64 + 255) / 256, n_jobs), dim3(256), 0, as_stream(stream), t);
    return check_launch("conv_pack_bf16x3_kernel");
}

extern "C" int ppo_conv3x3_bf16x3(const float *in, int relu_in, const void *packed, const float *bias, float *out, int n, int cin,
                                  int cout, int h, int w, void *stream)
{
    using namespace ppo;
    const char *who = "ppo_conv3x3_bf16x3";
    if (n < 0) return fail(PPO_E_INVALID, "%s: negative batch", who);
    if (n == 0) return PPO_OK;
    if (!in || !packed || !out || !aligned(packed, 16)) return fail(PPO_E_INVALID, "%s: null or misaligned pointer", who);
    if ((size_t)n * (cin > cout ? cin : cout) * h * w * sizeof(float) >= kBufferBytes)
        return fail(PPO_E_INVALID, "%s: tensor beyond the 2 GB a buffer descriptor spans", who);
    SplitConvArgs args{in, static_cast<const bf16x8 *>(packed), bias, out, nullptr, relu_in ? 0.f : -__builtin_inff(), n};
#define X(CI, CO, HH, WW, TR)                          \
    if (cin == CI && cout == CO && h == HH && w == WW) \
        return launch_split_conv<CI, CO, HH, WW, TR>(args, as_stream(stream));
    PPO_SPLIT_CONV_GEOMETRIES(X)
#undef X
    return fail(PPO_E_INVALID, "%s: no kernel for %d -> %d channels at %dx%d", who, cin, cout, h, w);
}

extern "C" int ppo_conv3x3_pool_bf16x3_supported(int cin, int cout, int h, int w)
{
#define X(CI, CO, HH, WW, TR) \
    if (cin == CI && cout == CO && h == HH && w == WW) return 1;
    PPO_SPLIT_CONV_POOL_GEOMETRIES(X)
#undef X
    return 0;
}

extern "C" int ppo_conv3x3_pool_bf16x3(const float *in, int relu_in, const void *packed, const float *bias, float *out, uint8_t *argmax,
                                       int n, int cin, int cout, int h, int w, void *stream)
{
    using namespace ppo;
    const char *who = "ppo_conv3x3_pool_bf16x3";
    if (n < 0) return fail(PPO_E_INVALID, "%s: negative batch", who);
    if (n == 0) return PPO_OK;
    if (!in || !packed || !out || !aligned(packed, 16)) return fail(PPO_E_INVALID, "%s: null or misaligned pointer", who);
    if ((size_t)n * (cin > cout ? cin : cout) * h * w * sizeof(float) >= kBufferBytes)
        return fail(PPO_E_INVALID, "%s: tensor beyond the 2 GB a buffer descriptor spans", who);
    SplitConvArgs args{in, static_cast<const bf16x8 *>(packed), bias, out, argmax, relu_in ? 0.f : -__builtin_inff(), n};
#define X(CI, CO, HH, WW, TR)                          \
    if (cin == CI && cout == CO && h == HH && w == WW) \
        return launch_split_conv<CI, CO, HH, WW, TR, true>(args, as_stream(stream));
    PPO_SPLIT_CONV_POOL_GEOMETRIES(X)
#undef X
    return fail(PPO_E_INVALID, "%s: no kernel for %d -> %d channels at %dx%d", who, cin, cout, h, w);
}

// ---- prototype: n_split = 3 is the float32-accurate form (six bf16 MFMAs per product block); n_split = 2 the shipped one
extern "C" size_t ppo_conv3x3_bf16_split_packed_bytes(int cin, int cout, int n_split)
{
    return ppo_conv3x3_bf16x3_packed_bytes(cin, cout) / 2 * (size_t)n_split;
}

extern "C" int ppo_conv3x3_pack_bf16_split(const float *weight, void *packed, int cin, int cout, int transposed, int n_split,
                                           void *stream)
{
    using namespace ppo;
    const char *who = "ppo_conv3x3_pack_bf16_split";
    if (n_split != 2 && n_split != 3) return fail(PPO_E_INVALID, "%s: 2 or 3 parts", who);
    if ((cin != 16 && cin != 32) || (cout != 16 && cout != 32)) return fail(PPO_E_INVALID, "%s: 16 or 32 channels", who);
    if (!weight || !packed || !aligned(packed, 16)) return fail(PPO_E_INVALID, "%s: null or misaligned pointer", who);
    ConvPackJobs t{};
    t.n = 1, t.w[0] = weight, t.packed[0] = static_cast<__bf16 *>(packed);
    t.cin[0] = cin, t.cout[0] = cout, t.transposed[0] = transposed, t.ns[0] = n_split;
    hipLaunchKernelGGL(conv_pack_bf16x3_kernel, dim3((2 * 9 * 64 + 255) / 256, 1), dim3(256), 0, as_stream(stream), t);
    return check_launch("conv_pack_bf16x3_kernel");
}

extern "C" int ppo_conv3x3_bf16_split(const float *in, int relu_in, const void *packed, const float *bias, float *out, int n, int cin,
                                      int cout, int h, int w, int n_split, void *stream)
{
    using namespace ppo;
    const char *who = "ppo_conv3x3_bf16_split";
    if (n_split == 2) return ppo_conv3x3_bf16x3(in, relu_in, packed, bias, out, n, cin, cout, h, w, stream);
    if (n_split != 3) return fail(PPO_E_INVALID, "%s: 2 or 3 parts", who);
    if (n < 0) return fail(PPO_E_INVALID, "%s: negative batch", who);
    if (n == 0) return PPO_OK;
    if (!in || !packed || !out || !aligned(packed, 16)) return fail(PPO_E_INVALID, "%s: null or misaligned pointer", who);
    if ((size_t)n * (cin > cout ? cin : cout) * h * w * sizeof(float) >= kBufferBytes)
        return fail(PPO_E_INVALID, "%s: tensor beyond the 2 GB a buffer descriptor spans", who);
    SplitConvArgs args{in, static_cast<const bf16x8 *>(packed), bias, out, nullptr, relu_in ? 0.f : -__builtin_inff(), n};
    if (cin == 16 && cout == 32 && h == 42 && w == 42) return launch_split_conv<16, 32, 42, 42, 7, false, 3>(args, as_stream(stream));
    if (cin == 32 && cout == 16 && h == 42 && w == 42) return launch_split_conv<32, 16, 42, 42, 7, false, 3>(args, as_stream(stream));
    if (cin == 32 && cout == 32 && h == 21 && w == 21) return launch_split_conv<32, 32, 21, 21, 7, false, 3>(args, as_stream(stream));
    return fail(PPO_E_INVALID, "%s: no three-part kernel for %d -> %d channels at %dx%d", who, cin, cout, h, w);
}

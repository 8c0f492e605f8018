// OPT-IN reduced-precision weight gradient of the 3x3 convolution (`--precision=medium|low`): every product as THREE bf16
// MFMAs on (hi, lo) splits of both operands with float32 accumulation, as stack_bf16x3.hip does for the residual blocks.
//
//   dW[o,i,ky,kx] = sum_{n,y,x} dy[n,o,y,x] * f(in[n,i,y+ky-1,x+kx-1])      db[o] = sum_{n,y,x} dy[n,o,y,x]
//
// (what autograd computes for torch.nn.Conv2d at rl/impala.py:61-62,96; f = max(., floor): the forward's ReLU, or the raw
// input with floor = -inf).  The exact float32 kernel (conv3x3_wgrad.hip) stays the default; this one writes the SAME
// per-workgroup slabs, so ppo_conv3x3_wgrad_reduce_f32 folds them in its fixed order whichever kernel produced them.
//
// GEMM view: M = output channel (A = dy), N = (tap, input channel) (B = the shifted input), K = pixels - the contraction
// runs over the index that is CONTIGUOUS in memory for both operands, the opposite of what v_mfma_f32_16x16x32_bf16 wants
// (lane = row / column, 8 consecutive K per lane).  gfx950's transposing LDS read does the turn: a band of the image sits
// in LDS as dense per-pixel records of 16 channels,
//     image(group of 16 channels, hi | lo)[record][16 x bf16]        (32 B per record)
// over the band's rows with one zero record either side of each row (so a tap is a constant record offset and the K loop
// has no select, mask or branch), and ds_read_b64_tr_b16 hands lane i of a 16-lane group channel i of four records.  One
// read instruction covers 16 consecutive records (8 per 32-lane half = 64 consecutive banks: conflict-free at any tap
// shift); K slot (g, j) of the MFMA is record 16 (j >> 2) + 4 g + (j & 3) of the step's 32 - A and B use the same map.
// Staging: each lane loads four channels of one pixel (dword buffer loads, rows outside the image read 0 through the
// range check), clamps, splits with v_cvt_pk_bf16_f32 and stores 8 + 8 bytes.  A 16-lane group takes 16 consecutive pixels
// of one channel quad: one cache line per group on the load side at the price of a 4-way bank conflict on the store side
// (measured 55.2 -> 51.5 us against the conflict-free 4 pixels x 4 quads, which touches four lines per group).  The next item's loads are in flight in registers during the K loop.  256-thread workgroups, up to
// three per CU (LDS <= 53 KB), so one workgroup's staging runs under another's MFMAs; accumulators stay in registers over
// all items of a workgroup; db is summed from the float32 dy values while they are staged.
// Per 32-pixel step and tap: 3 MFMAs (48 cycles) where the float32 kernel issues 8 (256 cycles); what bounds the launch
// is then HBM (x + dy once, 4 bytes per element), not the matrix pipe.
#include "common.h"

#include <cstdlib>

namespace ppo {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4w;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

constexpr int kRecBytes = 32;       // 16 channels x bf16
constexpr int kSplitWgradBatch = 5; // as conv3x3_wgrad.hip: a stack's four block convolutions + a riding first convolution
constexpr int kSplitWaves = 4;
// byte offset of a lane that must read 0: beyond the descriptor's 2 GB and still beyond it after a channel offset is added
constexpr int kFarOutside = (int)0x80000000u;

struct SplitWgradBatch {
    const float *in[kSplitWgradBatch];
    const float *dy[kSplitWgradBatch];
    float *partial[kSplitWgradBatch];
    float floor[kSplitWgradBatch];  // 0 = the forward's ReLU on load, -inf = raw input
};

constexpr int cmax(int a, int b) { return a > b ? a : b; }
// timing aids of tools/wgrad_bf16x3_phases.sh (results are void by design; never set in the shipped library):
// bit 0 no global loads, bit 1 no LDS stores, bit 2 no K loop
#ifndef PPO_TUNE_W3_SKIP
#define PPO_TUNE_W3_SKIP 0
#endif
#ifndef PPO_TUNE_W3_MAP
#define PPO_TUNE_W3_MAP 1
#endif
#ifndef PPO_TUNE_W3X3_TR42  // band heights of the three-part form (LDS is half as large again per band row)
#define PPO_TUNE_W3X3_TR42 7
#endif
#ifndef PPO_TUNE_W3X3_TR42B
#define PPO_TUNE_W3X3_TR42B 5
#endif
#ifndef PPO_TUNE_W3X3_TR21
#define PPO_TUNE_W3X3_TR21 5
#endif
#ifndef PPO_TUNE_W3_TR21   // band heights of 32 -> 32 at 21x21 and 16 -> 32 at 42x42
#define PPO_TUNE_W3_TR21 7
#endif
#ifndef PPO_TUNE_W3_TR42B
#define PPO_TUNE_W3_TR42B 7
#endif

// NS = 2: (hi, lo) parts, three products (`--precision=medium`).  NS = 3: (hi, mid, lo) = the whole float32 significand, six of
// the nine partial products (float32-accurate: profiles/r04_bf16_three_part_prototype.md; `--precision=high_bf16x6`).
template <int CIN, int COUT, int H, int W, int TR, int NS = 2>
struct SplitWgradCfg {
    static_assert(CIN % 16 == 0 && COUT % 16 == 0 && CIN <= 32 && COUT <= 32, "16 or 32 channels");
    static constexpr int NGI = CIN / 16, MT = COUT / 16;
    static constexpr int RW = W + 2;                          // records per band row: zero halo column either side
    static constexpr int NB = (H + TR - 1) / TR;              // bands per image
    static constexpr int STEPS = (TR * RW + 31) / 32;         // K steps of 32 records per band
    static constexpr int DREC = 32 * STEPS;                   // dy records (zero beyond TR * RW)
    static constexpr int XREC = 32 * STEPS + 2 * RW + 2;      // x records: one guard in front, rows -1 .. TR, the step padding
    static constexpr int OWN = NGI;                           // waves per K group: one per input-channel group
    static constexpr int KG = kSplitWaves / OWN;              // K groups
    static constexpr int SPG = (STEPS + KG - 1) / KG;         // steps per K group
    static constexpr int X_IMG = XREC * kRecBytes, D_IMG = DREC * kRecBytes;
    static constexpr int X_BYTES = NGI * NS * X_IMG, D_BYTES = MT * NS * D_IMG;
    static constexpr int JP = ((9 * CIN + 1) + 15) / 16 * 16; // slab row (conv3x3_wgrad.hip): j = tap * CIN + i, then db
    static constexpr int SLAB_BYTES = COUT * JP * 4 + kSplitWaves * COUT * 4;
    static constexpr int LDS_BYTES = cmax(X_BYTES + D_BYTES, SLAB_BYTES);
    static constexpr int XPIX = (TR + 2) * W, DPIX = TR * W;  // pixels staged per band and channel
    static constexpr int XIT = (XPIX + 63) / 64, DIT = (DPIX + 63) / 64;  // 16 pixels x 4 channel quads per wave-load
    static_assert(LDS_BYTES <= 160 * 1024, "band too tall");
};

__device__ __forceinline__ bf16x8 tr_read2(const unsigned char *p)
{
    // K slots 0..3 and 4..7 of this lane's group: records +0 and +16
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(p));
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(p + 16 * kRecBytes));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// four float32 -> NS bf16 quads (hi, then bf16 of each successive remainder) at p, p + plane, ...
template <int NS>
__device__ __forceinline__ void store_split4(unsigned char *p, int plane, const float (&v)[4])
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
    *reinterpret_cast<bf16x4 *>(p + plane) = mid;
    if (NS == 3) *reinterpret_cast<bf16x4 *>(p + 2 * plane) = lo;
}

template <int CIN, int COUT, int H, int W, int TR, int NS>
__global__ __launch_bounds__(kSplitWaves * 64) void conv3x3_wgrad_bf16x3_kernel(SplitWgradBatch batch, int n_images)
{
    using C = SplitWgradCfg<CIN, COUT, H, W, TR, NS>;
    constexpr int HW = H * W;
    extern __shared__ __align__(16) unsigned char smem_w[];
    unsigned char *const s_x = smem_w;               // [NGI][hi, lo][XREC] records
    unsigned char *const s_d = smem_w + C::X_BYTES;  // [MT][hi, lo][DREC] records
    const float *__restrict__ in = batch.in[blockIdx.y];
    const float *__restrict__ dy = batch.dy[blockIdx.y];
    float *__restrict__ partial = batch.partial[blockIdx.y];
    const float floor = batch.floor[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int i = tid * 16; i < C::X_BYTES + C::D_BYTES; i += kSplitWaves * 64 * 16)  // halo / guard / padding records stay zero
        *reinterpret_cast<uint4 *>(smem_w + i) = uint4{0, 0, 0, 0};

    // ---- staging constants: lane = (pixel lane >> 2 of 16, channel quad lane & 3); wave-load i covers band pixels
    // 64 i + 16 wave + (0..15)
#if PPO_TUNE_W3_MAP == 1  // 16 consecutive pixels of one channel per 16-lane group (one cache line), chunk = lane >> 4
    const int quad = lane >> 4, pl = lane & 15;
#else                    // 4 pixels x 4 channel quads per 16-lane group: four whole records per LDS store group
    const int quad = lane & 3, pl = lane >> 2;
#endif
    int x_rec[C::XIT], d_rec[C::DIT];  // LDS byte offset of the quad inside an image
#pragma unroll
    for (int i = 0; i < C::XIT; ++i) {
        const int p = 64 * i + 16 * wave + pl;
        x_rec[i] = ((p / W) * C::RW + p % W + 2) * kRecBytes + quad * 8;  // + guard record + halo column
    }
#pragma unroll
    for (int i = 0; i < C::DIT; ++i) {
        const int p = 64 * i + 16 * wave + pl;
        d_rec[i] = ((p / W) * C::RW + p % W + 1) * kRecBytes + quad * 8;
    }

    // ---- K-loop constants: this lane's address inside a 16-record block (row q = (l & 15) >> 2, quad p = l & 3)
    const int own = wave % C::OWN, kg = wave / C::OWN;
    const int blk = ((4 * (lane >> 4) + ((lane & 15) >> 2)) * kRecBytes) + (lane & 3) * 8;
    const int s_begin = kg * C::SPG, s_end = s_begin + C::SPG < C::STEPS ? s_begin + C::SPG : C::STEPS;

    f32x4w acc[C::MT][9];
#pragma unroll
    for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[m][t] = f32x4w{0.f, 0.f, 0.f, 0.f};
    float bsum[C::MT][4];
#pragma unroll
    for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) bsum[m][r] = 0.f;

    float xv[C::XIT][C::NGI][4] = {}, dv[C::DIT][C::MT][4] = {};
    const int n_items = n_images * C::NB;
    auto issue = [&](int item) {
        if (PPO_TUNE_W3_SKIP & 1) return;
        const int img = item / C::NB, band = item % C::NB;
        const __amdgpu_buffer_rsrc_t xb = buffer_of(in + (size_t)img * CIN * HW), db = buffer_of(dy + (size_t)img * COUT * HW);
        const int x0 = (band * TR - 1) * W, d0 = band * TR * W;  // plane offset of the band's first staged pixel
#pragma unroll
        for (int i = 0; i < C::XIT; ++i) {
            const int p = 64 * i + 16 * wave + pl;
            const int gp = x0 + p;
            // (bitwise &, one select, and an opaque result: a short-circuit condition becomes control flow around the loads
            // and every join of it a full s_waitcnt vmcnt(0), common.h)
            int base = (((unsigned)gp < (unsigned)HW) & (p < C::XPIX)) ? (quad * 4 * HW + gp) * 4 : kFarOutside;
            asm volatile("" : "+v"(base));
#pragma unroll
            for (int gi = 0; gi < C::NGI; ++gi)
#pragma unroll
                for (int r = 0; r < 4; ++r) xv[i][gi][r] = buffer_f32(xb, base + (gi * 16 + r) * HW * 4);
        }
#pragma unroll
        for (int i = 0; i < C::DIT; ++i) {
            const int p = 64 * i + 16 * wave + pl;
            const int gp = d0 + p;
            int base = ((gp < HW) & (p < C::DPIX)) ? (quad * 4 * HW + gp) * 4 : kFarOutside;
            asm volatile("" : "+v"(base));
#pragma unroll
            for (int m = 0; m < C::MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) dv[i][m][r] = buffer_f32(db, base + (m * 16 + r) * HW * 4);
        }
    };
    auto publish = [&]() {
        if (PPO_TUNE_W3_SKIP & 2) return;
#pragma unroll
        for (int i = 0; i < C::XIT; ++i) {
            if (64 * i + 64 <= C::XPIX || 64 * i + 16 * wave + pl < C::XPIX) {
#pragma unroll
                for (int gi = 0; gi < C::NGI; ++gi) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_fmed3f(xv[i][gi][r], floor, __builtin_inff());
                    store_split4<NS>(s_x + gi * NS * C::X_IMG + x_rec[i], C::X_IMG, v);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < C::DIT; ++i) {
            if (64 * i + 64 <= C::DPIX || 64 * i + 16 * wave + pl < C::DPIX) {
#pragma unroll
                for (int m = 0; m < C::MT; ++m) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bsum[m][r] += dv[i][m][r];
                    store_split4<NS>(s_d + m * NS * C::D_IMG + d_rec[i], C::D_IMG, dv[i][m]);
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
        // ---- K loop: steps of this wave's K group, all nine taps of its input-channel group, every output tile
        const unsigned char *xs = s_x + own * NS * C::X_IMG + blk + s_begin * 32 * kRecBytes;
        const unsigned char *ds = s_d + blk + s_begin * 32 * kRecBytes;
#pragma unroll 1
        for (int s = s_begin; s < ((PPO_TUNE_W3_SKIP & 4) ? s_begin : s_end); ++s) {
            bf16x8 ap[NS][C::MT];  // dy parts: 0 = hi
#pragma unroll
            for (int m = 0; m < C::MT; ++m)
#pragma unroll
                for (int q = 0; q < NS; ++q) ap[q][m] = tr_read2(ds + (m * NS + q) * C::D_IMG);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int toff = ((t / 3) * C::RW + (t % 3)) * kRecBytes;  // dy record d <-> x record d + ky RW + kx (guard + halo included)
                bf16x8 bp[NS];
#pragma unroll
                for (int q = 0; q < NS; ++q) bp[q] = tr_read2(xs + toff + q * C::X_IMG);
#pragma unroll
                for (int m = 0; m < C::MT; ++m) {
                    // partial products a_i b_j with i + j < NS, smallest first (NS = 2: lo hi, hi lo, hi hi)
#pragma unroll
                    for (int sum = NS - 1; sum >= 0; --sum)
#pragma unroll
                        for (int i = sum; i >= 0; --i)
                            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[i][m], bp[sum - i], acc[m][t], 0, 0, 0);
                }
            }
            xs += 32 * kRecBytes;
            ds += 32 * kRecBytes;
        }
        __syncthreads();  // the band's readers are done
    }

    // ---- the workgroup's slab: K groups fold into one LDS image in turn, then [COUT][JP] floats leave as whole rows
    float *s_slab = reinterpret_cast<float *>(smem_w);
    float *s_bias = s_slab + COUT * C::JP;  // [wave][COUT]
    for (int i = tid; i < COUT * C::JP; i += kSplitWaves * 64) s_slab[i] = 0.f;
    // accumulator element r of lane l: output channel 16 m + 4 (l >> 4) + r, input channel 16 own + (l & 15)
#pragma unroll 1
    for (int k = 0; k < C::KG; ++k) {
        __syncthreads();
        if (kg == k) {
#pragma unroll
            for (int m = 0; m < C::MT; ++m)
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float *d = s_slab + (m * 16 + 4 * (lane >> 4) + r) * C::JP + t * CIN + own * 16 + (lane & 15);
                        *d += acc[m][t][r];
                    }
        }
    }
    // db: lanes with the same channel quad (lane & 3) hold partial sums of the same four channels
#pragma unroll
    for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = bsum[m][r];
#if PPO_TUNE_W3_MAP == 1
#pragma unroll
            for (int sh = 1; sh < 16; sh <<= 1) v += __shfl_xor(v, sh);
            if (pl == 0) s_bias[wave * COUT + m * 16 + quad * 4 + r] = v;
#else
#pragma unroll
            for (int sh = 4; sh < 64; sh <<= 1) v += __shfl_xor(v, sh);
            if (lane < 4) s_bias[wave * COUT + m * 16 + lane * 4 + r] = v;
#endif
        }
    __syncthreads();
    if (tid < COUT) {
        float v = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < kSplitWaves; ++w2) v += s_bias[w2 * COUT + tid];
        s_slab[tid * C::JP + 9 * CIN] = v;
    }
    __syncthreads();
    float4 *out = reinterpret_cast<float4 *>(partial + (size_t)blockIdx.x * COUT * C::JP);
    for (int i = tid; i < COUT * C::JP / 4; i += kSplitWaves * 64) out[i] = reinterpret_cast<const float4 *>(s_slab)[i];
}

template <int CIN, int COUT, int H, int W, int TR, int NS = 2>
int launch_split_wgrad(const SplitWgradBatch &b, int count, int n_images, size_t workspace_bytes, int *n_slabs, hipStream_t st)
{
    using C = SplitWgradCfg<CIN, COUT, H, W, TR, NS>;
    auto kern = conv3x3_wgrad_bf16x3_kernel<CIN, COUT, H, W, TR, NS>;
    static int per_cu = 0;
    if (!per_cu) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)C::LDS_BYTES);
        if (e != hipSuccess) return fail(PPO_E_HIP, "conv3x3_wgrad_bf16x3: hipFuncSetAttribute: %s", hipGetErrorString(e));
        int occ = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(kern), kSplitWaves * 64, C::LDS_BYTES);
        if (e != hipSuccess || occ < 1) return fail(PPO_E_HIP, "conv3x3_wgrad_bf16x3: occupancy query: %s", hipGetErrorString(e));
        per_cu = occ > 4 ? 4 : occ;
        if (getenv("PPO_AMD_DEBUG_OCCUPANCY")) {
            hipFuncAttributes fa{};
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern));
            fprintf(stderr, "conv3x3_wgrad_bf16x3<%d,%d,%d,%d,%d>: occupancy query %d, LDS %d B, %d regs, static LDS %zu\n", CIN, COUT, H, W, TR,
                    occ, (int)C::LDS_BYTES, fa.numRegs, fa.sharedSizeBytes);
        }
    }
    // one resident wave of workgroups over all problems of the launch; every workgroup writes one slab
    const int n_items = n_images * C::NB;
    int grid = (256 * per_cu) / count;
    if (grid > n_items) grid = n_items;
    const size_t slab = (size_t)COUT * C::JP * sizeof(float);
    if ((size_t)grid * slab > workspace_bytes) grid = (int)(workspace_bytes / slab);
    if (grid < 1) return fail(PPO_E_INVALID, "conv3x3_wgrad_bf16x3: workspace too small (%zu bytes)", workspace_bytes);
    *n_slabs = grid;
    hipLaunchKernelGGL(kern, dim3(grid, count), dim3(kSplitWaves * 64), C::LDS_BYTES, st, b, n_images);
    return check_launch("conv3x3_wgrad_bf16x3_kernel");
}

#define PPO_SPLIT_WGRAD_GEOMETRIES(X) \
    X(16, 16, 42, 42, 7)              \
    X(16, 32, 42, 42, PPO_TUNE_W3_TR42B) \
    X(32, 32, 21, 21, PPO_TUNE_W3_TR21)  \
    X(32, 32, 11, 11, 11)             \
    X(16, 16, 32, 32, 8)              \
    X(16, 32, 32, 32, 8)              \
    X(32, 32, 16, 16, 8)              \
    X(32, 32, 8, 8, 8)

}  // namespace
}  // namespace ppo

extern "C" int ppo_conv3x3_backward_weight_bf16x3_supported(int cin, int cout, int h, int w)
{
#define X(CI, CO, HH, WW, TR) \
    if (cin == CI && cout == CO && h == HH && w == WW) return 1;
    PPO_SPLIT_WGRAD_GEOMETRIES(X)
#undef X
    return 0;
}

static int split_wgrad_entry(const char *who, int n_split, const float *const *ins, const int *relu, const float *const *dys,
                             void *const *workspaces, size_t workspace_bytes, int count, int n, int cin, int cout, int h, int w,
                             int *n_slabs, void *stream)
{
    using namespace ppo;
    if (count < 1 || count > kSplitWgradBatch) return fail(PPO_E_INVALID, "%s: 1 .. %d problems per launch", who, kSplitWgradBatch);
    if (!ins || !relu || !dys || !workspaces || !n_slabs) return fail(PPO_E_INVALID, "%s: null pointer", who);
    if (n <= 0) return fail(PPO_E_INVALID, "%s: empty batch", who);
    if ((size_t)n * (cin > cout ? cin : cout) * h * w * sizeof(float) >= kBufferBytes)
        return fail(PPO_E_INVALID, "%s: tensor beyond the 2 GB a buffer descriptor spans", who);
    SplitWgradBatch b{};
    for (int i = 0; i < count; ++i) {
        if (!ins[i] || !dys[i] || !workspaces[i] || !aligned(workspaces[i], 16)) return fail(PPO_E_INVALID, "%s: null or misaligned pointer (problem %d)", who, i);
        b.in[i] = ins[i], b.dy[i] = dys[i], b.partial[i] = static_cast<float *>(workspaces[i]);
        b.floor[i] = relu[i] ? 0.f : -__builtin_inff();
    }
    if (n_split == 3) {  // the Atari-shaped net's layers only
#define X3(CI, CO, HH, WW, TR)                         \
    if (cin == CI && cout == CO && h == HH && w == WW) \
        return launch_split_wgrad<CI, CO, HH, WW, TR, 3>(b, count, n, workspace_bytes, n_slabs, as_stream(stream));
        X3(16, 16, 42, 42, PPO_TUNE_W3X3_TR42)
        X3(16, 32, 42, 42, PPO_TUNE_W3X3_TR42B)
        X3(32, 32, 21, 21, PPO_TUNE_W3X3_TR21)
        X3(32, 32, 11, 11, 11)
#undef X3
        return fail(PPO_E_INVALID, "%s: no three-part kernel for %d -> %d channels at %dx%d", who, cin, cout, h, w);
    }
#define X(CI, CO, HH, WW, TR)                                 \
    if (cin == CI && cout == CO && h == HH && w == WW)        \
        return launch_split_wgrad<CI, CO, HH, WW, TR>(b, count, n, workspace_bytes, n_slabs, as_stream(stream));
    PPO_SPLIT_WGRAD_GEOMETRIES(X)
#undef X
    return fail(PPO_E_INVALID, "%s: no kernel for %d -> %d channels at %dx%d", who, cin, cout, h, w);
}

extern "C" int ppo_conv3x3_backward_weight_slabs_batch_bf16x3(const float *const *ins, const int *relu, const float *const *dys,
                                                              void *const *workspaces, size_t workspace_bytes, int count, int n,
                                                              int cin, int cout, int h, int w, int *n_slabs, void *stream)
{
    return split_wgrad_entry("ppo_conv3x3_backward_weight_slabs_batch_bf16x3", 2, ins, relu, dys, workspaces, workspace_bytes, count, n,
                             cin, cout, h, w, n_slabs, stream);
}

/* n_split = 3: the float32-accurate three-part form (six products); n_split = 2 is the entry point above */
extern "C" int ppo_conv3x3_backward_weight_slabs_batch_bf16_split(const float *const *ins, const int *relu, const float *const *dys,
                                                                  void *const *workspaces, size_t workspace_bytes, int count, int n,
                                                                  int cin, int cout, int h, int w, int n_split, int *n_slabs,
                                                                  void *stream)
{
    if (n_split != 2 && n_split != 3) return ppo::fail(PPO_E_INVALID, "ppo_conv3x3_backward_weight_slabs_batch_bf16_split: 2 or 3 parts");
    return split_wgrad_entry("ppo_conv3x3_backward_weight_slabs_batch_bf16_split", n_split, ins, relu, dys, workspaces, workspace_bytes,
                             count, n, cin, cout, h, w, n_slabs, stream);
}

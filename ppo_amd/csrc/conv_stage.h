// Global -> LDS staging shared by the convolution kernels (gfx950).
//
// A band of image rows (+1-pixel halo, zero padded) of every input channel goes to LDS as planar
// [channel][row][col] with the load transform fused (identity / ReLU / uint8 -> x/255).
// The loop is written as "U independent global loads, then U LDS stores" so that each thread keeps U
// requests in flight: a plain load->store loop waits for every load (s_waitcnt vmcnt(0) per element)
// and made the whole convolution latency-bound (81 us -> see profiles/ for the effect).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ppo {

enum { IN_NONE = 0, IN_RELU = 1, IN_U8 = 2 };

// s_dst[c * PLANE + r * PW + col] for c < CP, r < ROWS, col < PW holds
//   f(src[img][c][y0 + r - HALO][col - HALO])   (0 outside the image or for c >= C)
template <int C, int CP, int H, int W, int ROWS, int PW, int PLANE, int HALO, int IN_MODE, int NTHREADS>
__device__ __forceinline__ void stage_band(const void *__restrict__ src_, int img, int y0, float *__restrict__ s_dst,
                                           int tid)
{
    constexpr int TOTAL = CP * ROWS * PW;
    constexpr int U = 8;
#pragma unroll 1
    for (int base = 0; base < TOTAL; base += NTHREADS * U) {
        float v[U];
        int off[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * NTHREADS + tid;
            const int c = idx / (ROWS * PW);
            const int rem = idx - c * (ROWS * PW);
            const int r = rem / PW;
            const int col = rem - r * PW;
            const int gy = y0 + r - HALO;
            const int gx = col - HALO;
            off[u] = idx < TOTAL ? c * PLANE + r * PW + col : -1;
            v[u] = 0.f;
            if (idx < TOTAL && c < C && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                const size_t gi = (((size_t)img * C + c) * H + gy) * W + gx;
                if (IN_MODE == IN_U8) v[u] = (float)static_cast<const uint8_t *>(src_)[gi];
                else v[u] = static_cast<const float *>(src_)[gi];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (off[u] >= 0) {
                float x = v[u];
                if (IN_MODE == IN_U8) x = x / 255.0f;
                if (IN_MODE == IN_RELU) x = fmaxf(x, 0.f);
                s_dst[off[u]] = x;
            }
        }
    }
}

}  // namespace ppo

// f32-input MFMA helpers for gfx950 (exact f32: bitwise a k-ordered fmaf chain,
// cdna_hip_programming.md §3 "FP32-input MFMA").
#pragma once
#include <hip/hip_runtime.h>

namespace ppo {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] * B[4x16].
//   lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15];
//   lane l receives D[i = (l >> 4) * 4 + r][j = l & 15] in element r of the accumulator.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ReLU as ONE VALU instruction (v_med3_f32 x, 0, +inf).  fmaxf(x, 0) costs two: hipcc first
// canonicalises the operand with v_max x, x (IEEE quieting), which doubles the VALU work that sits
// between MFMAs when ReLU is applied at operand-read time.
__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_inff()); }

}  // namespace ppo

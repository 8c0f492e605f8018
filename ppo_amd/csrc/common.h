// Shared host-side helpers for libppo_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/ppo_amd.h"

namespace ppo {

// thread-local message behind ppo_last_error()
char *error_buffer();
int fail(int code, const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PPO_E_HIP, "%s: %s", what, hipGetErrorString(e));
    return PPO_OK;
}

inline bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Range-checked reads through a buffer descriptor.  Why they exist: a predicated global load - `ok ? p[i] : 0`,
// however it is spelled - is compiled to a branch around the load, and at every such join the wait-count
// bookkeeping falls back to s_waitcnt vmcnt(0): the wave stalls for a full memory round trip per load and, with it,
// for every LDS-DMA request in flight.  A buffer load is unconditional; a lane that must not read passes kOutside as
// its byte offset and the hardware range check returns zeros.  An absent tensor is a descriptor of zero bytes:
// every read of it is zero, no branch on the pointer.  Offsets are 32-bit byte offsets from the base.
constexpr uint32_t kBufferBytes = 0x80000000u;
constexpr int kOutside = -16;  // 0xfffffff0 as an unsigned offset: beyond kBufferBytes
typedef int buf_i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_of(const void *base, bool present = true)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, present ? kBufferBytes : 0u, 0x00020000);
}
__device__ __forceinline__ float4 buffer_f32x4(__amdgpu_buffer_rsrc_t b, int byte_off)
{
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(b, byte_off, 0, 0));
}
__device__ __forceinline__ float buffer_f32(__amdgpu_buffer_rsrc_t b, int byte_off)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b, byte_off, 0, 0));
}

constexpr int kWave = 64;  // CDNA wavefront

}  // namespace ppo

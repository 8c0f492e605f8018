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

constexpr int kWave = 64;  // CDNA wavefront

}  // namespace ppo

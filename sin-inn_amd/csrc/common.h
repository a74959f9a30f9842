// Shared helpers for libsininn (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/sininn.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace sininn {

void set_error(const char* fmt, ...);

#define SININN_CHECK(cond, ...)                \
  do {                                         \
    if (!(cond)) {                             \
      ::sininn::set_error(__VA_ARGS__);        \
      return 1;                                \
    }                                          \
  } while (0)

#define SININN_LAUNCH_CHECK(name)                                                   \
  do {                                                                              \
    hipError_t e__ = hipGetLastError();                                             \
    if (e__ != hipSuccess) {                                                        \
      ::sininn::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));   \
      return (int)e__;                                                              \
    }                                                                               \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// GLOW soft clamp (FrEIA GLOWCouplingBlock.log_e): literal 0.636, not 2/pi.
__device__ __forceinline__ float glow_log_e(float s, float clamp) { return clamp * 0.636f * atanf(s / clamp); }
__device__ __forceinline__ float glow_dlog_e(float s, float clamp) {
  float u = s / clamp;
  return 0.636f / (1.0f + u * u);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

}  // namespace sininn

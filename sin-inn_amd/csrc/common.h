// Shared helpers for libsininn (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <type_traits>

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

// The ONE way library code creates a weight-gradient descriptor: every field zero, the size tag set.  (Round 2's abort came
// from a stack item that was filled field by field after the struct had grown -- DESIGN 8; wgrad_group refuses untagged items.)
static inline sininn_wgrad_item wgrad_item_init() {
  sininn_wgrad_item it = {};
  it.struct_bytes = sizeof(sininn_wgrad_item);
  return it;
}

// GLOW soft clamp (FrEIA GLOWCouplingBlock.log_e): literal 0.636, not 2/pi.
__device__ __forceinline__ float glow_log_e(float s, float clamp) { return clamp * 0.636f * atanf(s / clamp); }
__device__ __forceinline__ float glow_dlog_e(float s, float clamp) {
  float u = s / clamp;
  return 0.636f / (1.0f + u * u);
}

// ---- explicit LDS reads ---------------------------------------------------------------------------------------------
// The compiler fuses neighbouring 8-byte LDS loads into ds_read2_b64, which runs at half the rate of ds_read_b64 and banks
// per 16 contiguous lanes modulo 32 (MI355X_MICROARCH.md, LDS table): layouts that are conflict-free for ds_read_b64 then
// conflict 2-4 ways (SQ_LDS_BANK_CONFLICT was 60 % of SQ_LDS_IDX_ACTIVE in the Winograd kernel).  These helpers emit the
// instruction we designed the layout for; the data is usable after lds_wait<N>(regs...) (N = newer reads still in flight).
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned lds_addr(const void* p) { return static_cast<unsigned>(reinterpret_cast<uintptr_t>(p)); }

template <int OFF>
__device__ __forceinline__ f32x2 lds_read_b64(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536 && OFF % 8 == 0, "ds_read_b64 immediate offset");
  f32x2 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

template <int N>
__device__ __forceinline__ void lds_wait(f32x2& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N>
__device__ __forceinline__ void lds_wait(f32x2& a, f32x2& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }

// ---- raw buffer loads ---------------------------------------------------------------------------------------------------
// A load whose byte offset is >= num_records returns 0.  Staging loops give every lane that lies outside the image (or beyond
// the packed columns) the offset BUF_OOB, so zero fill costs no select, no zeroed registers and no exec-mask branch, and the
// per-step advance is the SCALAR offset operand: a staged 16-byte unit is ONE instruction (vector-ALU / branch instructions in
// a K loop are issue time taken from the matrix pipe, tools/mfma_valu.hip).  Real offsets must stay below 2^31.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned BUF_OOB = 0xffffffffu;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0x80000000u, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned uniform_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, (int)uniform_off, 0));
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

}  // namespace sininn

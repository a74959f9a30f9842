// Diagnostic (not part of the product): what f32 MFMA rate does this box sustain?
//   variant 0: v_mfma_f32_16x16x4_f32 from registers, NACC independent accumulators, W waves per SIMD
//   variant 1: same + one ds_read_b64 pair per 2 MFMAs (operands re-read from LDS)
//   variant 2: v_mfma_f32_32x32x2_f32 from registers
//   variant 3 / 4: 16x16x4 with the A and B operand taken from component j / j and j + 1 of two float4 register quads that
//                  change on every instruction (the pattern of a GEMM loop fed by 16-byte fragment loads): same / different
//                  register index modulo 4 for A and B
// Reports TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VAR, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)(i % 7) * 0.25f - 0.5f;
  __syncthreads();
  float a = threadIdx.x * 0.001f + 0.5f, b = 1.0f - threadIdx.x * 0.002f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (VAR == 2) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0; for (int j = 0; j < NACC; ++j) for (int q = 0; q < 16; ++q) s += acc[j][q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = (f32x4){0, 0, 0, 0};
    const float2* lp = reinterpret_cast<const float2*>(lds) + (threadIdx.x & 63) * 9;
    for (int it = 0; it < iters; ++it) {
      if constexpr (VAR == 3 || VAR == 4) {
        const f32x4 fa = *reinterpret_cast<const f32x4*>(lds + (threadIdx.x & 63) * 4 + ((it & 3) << 8));
        const f32x4 fb = *reinterpret_cast<const f32x4*>(lds + 2048 + (threadIdx.x & 63) * 4 + ((it & 3) << 8));
#pragma unroll
        for (int j = 0; j < NACC; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j & 3], fb[(j + (VAR == 4 ? 1 : 0)) & 3], acc[j], 0, 0, 0);
      } else if constexpr (VAR == 1) {
#pragma unroll
        for (int j = 0; j < NACC; j += 2) {
          float2 fa = lp[(it * 2 + j) & 255], fb = lp[(it * 2 + j + 1) & 255];
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.x, fb.x, acc[j], 0, 0, 0);
          acc[j + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.y, fb.y, acc[j + 1], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
      }
    }
    float s = 0; for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int VAR, int NACC>
void run(const char* name, int blocks_per_cu, double flop_per_mfma) {
  float* out; unsigned long long* clk;
  int blocks = 256 * blocks_per_cu, iters = getenv("PEAK_ITERS") ? atoi(getenv("PEAK_ITERS")) : 20000;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<VAR, NACC>), dim3(blocks), dim3(256), 0, 0, out, 100, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<VAR, NACC>), dim3(blocks), dim3(256), 0, 0, out, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  double flops = (double)blocks * 4 * iters * NACC * flop_per_mfma;
  printf("%-44s blocks/CU=%d  %7.1f TFLOP/s  clock %.2f GHz  (%.2f ms)\n", name, blocks_per_cu, flops / ms / 1e9,
         (double)h[0] / (double)h[1] * 0.1, ms);
  hipFree(out); hipFree(clk);
}

int main() {
  run<0, 4>("16x16x4 regs, 4 acc", 1, 2048);
  run<0, 8>("16x16x4 regs, 8 acc", 1, 2048);
  run<0, 8>("16x16x4 regs, 8 acc", 2, 2048);
  run<1, 8>("16x16x4 + ds_read_b64 per 2 mfma, 8 acc", 1, 2048);
  run<1, 8>("16x16x4 + ds_read_b64 per 2 mfma, 8 acc", 2, 2048);
  run<3, 16>("16x16x4, A = fa[j], B = fb[j], 16 acc", 1, 2048);
  run<4, 16>("16x16x4, A = fa[j], B = fb[j+1], 16 acc", 1, 2048);
  run<3, 4>("16x16x4, A = fa[j], B = fb[j], 4 acc", 1, 2048);
  run<3, 16>("16x16x4, A = fa[j], B = fb[j], 16 acc", 2, 2048);
  run<2, 4>("32x32x2 regs, 4 acc", 1, 4096);
  run<2, 4>("32x32x2 regs, 4 acc", 2, 4096);
  return 0;
}

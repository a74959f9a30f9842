// Diagnostic (not part of the product): how much vector-ALU work fits in the shadow of the f32 matrix pipe?
// Every wave runs   repeat { v_mfma ; NV x v_fma_f32 (independent registers) }   with 8 rotating accumulators, W waves per SIMD.
// DEP = 1: the first v_fma of a group produces the A operand of the NEXT v_mfma (the pattern of a per-lane Winograd transform
// feeding the matrix pipe).  LDS = 1: one ds_read_b32 per MFMA as well.  Prints clocks per MFMA per SIMD (32 = the pipe's floor
// for 16x16x4, 64 for 32x32x2).   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu.hip -o tools/mfma_valu && tools/mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BIG, int NV, int DEP, int LDS, int PK = 0>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
  __shared__ float lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = (float)(i % 7) * 0.25f - 0.5f;
  __syncthreads();
  float a = threadIdx.x * 0.001f + 0.5f, b = 1.0f - threadIdx.x * 0.002f;
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = 0.25f * j + threadIdx.x;
  const float c = 0.999f, d = 0.001f;
  float l = 0.f;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 xp[8], cp = {0.5f, 0.25f};
  for (int j = 0; j < 8; ++j) xp[j] = (f32x2){0.25f * j, 1.f + threadIdx.x};
  const float* lp = lds + (threadIdx.x & 63);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if constexpr (BIG) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[j & 3]) : "v"(DEP ? x[0] : a), "v"(b));
        if constexpr (LDS) asm volatile("ds_read_b32 %0, %1" : "=v"(l) : "v"((unsigned)(size_t)(lp + ((it * 8 + j) & 15) * 64)));
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          if constexpr (PK) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xp[v & 7]) : "v"(cp));
          else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(c), "v"(d));
        }
        if constexpr (DEP) asm volatile("s_nop 1");
      }
      if constexpr (LDS) { asm volatile("s_waitcnt lgkmcnt(0)"); x[7] += l; }
    }
    float s = 0; for (int j = 0; j < 4; ++j) for (int q = 0; q < 16; ++q) s += acc[j][q];
    for (int j = 0; j < 8; ++j) s += x[j] + xp[j].x + xp[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f32x4 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(DEP ? x[0] : a), "v"(b));
        if constexpr (LDS) asm volatile("ds_read_b32 %0, %1" : "=v"(l) : "v"((unsigned)(size_t)(lp + ((it * 8 + j) & 15) * 64)));
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          if constexpr (PK) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xp[v & 7]) : "v"(cp));
          else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(c), "v"(d));
        }
        if constexpr (DEP) asm volatile("s_nop 1");
      }
      if constexpr (LDS) { asm volatile("s_waitcnt lgkmcnt(0)"); x[7] += l; }
    }
    float s = 0; for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3] + x[j] + xp[j].x + xp[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int BIG, int NV, int DEP, int LDS, int PK = 0>
void run(int blocks_per_cu) {
  float* out; unsigned long long* clk;
  const int blocks = 256 * blocks_per_cu, iters = 4000;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, 16);
  hipLaunchKernelGGL((k<BIG, NV, DEP, LDS, PK>), dim3(blocks), dim3(256), 0, 0, out, 100, clk);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<BIG, NV, DEP, LDS, PK>), dim3(blocks), dim3(256), 0, 0, out, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double mfma_per_simd = (double)iters * 8 * blocks_per_cu;
  const double flops = (double)blocks * 4 * iters * 8 * (BIG ? 4096.0 : 2048.0);
  printf("%s  %s/mfma %d  dep %d  lds %d  waves/SIMD %d : %6.1f shader clocks per MFMA per SIMD, %6.1f TFLOP/s\n",
         BIG ? "32x32x2 " : "16x16x4 ", PK ? "v_pk_add_f32" : "v_fma_f32", NV, DEP, LDS, blocks_per_cu, (double)h[0] / mfma_per_simd, flops / ms / 1e9);
  hipFree(out); hipFree(clk);
}

template <int BIG>
void sweep() {
  for (int w = 1; w <= 2; ++w) {
    run<BIG, 0, 0, 0>(w); run<BIG, 1, 0, 0>(w); run<BIG, 2, 0, 0>(w); run<BIG, 3, 0, 0>(w); run<BIG, 4, 0, 0>(w);
    run<BIG, 6, 0, 0>(w); run<BIG, 8, 0, 0>(w); run<BIG, 12, 0, 0>(w); run<BIG, 16, 0, 0>(w);
    run<BIG, 3, 1, 0>(w); run<BIG, 3, 0, 1>(w); run<BIG, 3, 1, 1>(w);
    run<BIG, 1, 0, 0, 1>(w); run<BIG, 2, 0, 0, 1>(w); run<BIG, 4, 0, 0, 1>(w); run<BIG, 8, 0, 0, 1>(w);
  }
}

int main() {
  sweep<0>();
  sweep<1>();
  return 0;
}

// Diagnostic: 32x32x2 MFMA loop fed from LDS like conv32 (MT=2, NT=2 per wave, ds_read_b64 fragments), with optional
// per-iteration barrier / LDS writes / global loads, to find which ingredient costs MFMA rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int FLAGS>   // 1 barrier per iteration, 2 ds_write of staged regs, 4 global loads, 8 sched_barrier prefetch
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ g, float* out, int iters) {
  extern __shared__ float lds[];
  constexpr int S = 34;
  for (int i = threadIdx.x; i < 436 * S; i += 256) lds[i] = (float)((i * 7) % 13) * 0.125f - 0.75f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  int a_base[2], b_base[2];
  for (int m = 0; m < 2; ++m) a_base[m] = (((wm * 2 + m) * 2 + (li >> 4)) * 18 + (li & 15)) * S + 2 * kh;
  for (int n = 0; n < 2; ++n) b_base[n] = (180 + (wn * 2 + n) * 32 + li) * S + 2 * kh;
  f32x16 acc[2][2];
  for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int q = 0; q < 16; ++q) acc[m][n][q] = 0.f;
  f32x4 wreg[4];
  for (int r = 0; r < 4; ++r) wreg[r] = (f32x4){0, 0, 0, 0};
  const float* gp = g + (size_t)blockIdx.x * 4096 + threadIdx.x * 4;
  for (int it = 0; it < iters; ++it) {
    if (FLAGS & 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* d = lds + (180 + 128 * ((it + 1) & 1) * 0 + (threadIdx.x >> 3) + 32 * r) * S + (threadIdx.x & 7) * 4;
        *reinterpret_cast<float2*>(d) = make_float2(wreg[r][0], wreg[r][1]);
        *reinterpret_cast<float2*>(d + 2) = make_float2(wreg[r][2], wreg[r][3]);
      }
    }
    if (FLAGS & 4) {
#pragma unroll
      for (int r = 0; r < 4; ++r) wreg[r] = *reinterpret_cast<const f32x4*>(gp + ((it * 4 + r) & 1023) * 1024);
    }
    const float* A = lds + ((it % 9) / 3 * 18 + (it % 3)) * S;
    float2 af[2][2], bf[2][2];
    for (int m = 0; m < 2; ++m) af[0][m] = *reinterpret_cast<const float2*>(A + a_base[m]);
    for (int n = 0; n < 2; ++n) bf[0][n] = *reinterpret_cast<const float2*>(lds + b_base[n]);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < 8) {
        for (int m = 0; m < 2; ++m) af[nxt][m] = *reinterpret_cast<const float2*>(A + a_base[m] + (ks + 1) * 4);
        for (int n = 0; n < 2; ++n) bf[nxt][n] = *reinterpret_cast<const float2*>(lds + b_base[n] + (ks + 1) * 4);
      }
      if (FLAGS & 8) __builtin_amdgcn_sched_barrier(0);
      for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n)
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][m].x, bf[cur][n].x, acc[m][n], 0, 0, 0);
      for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n)
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][m].y, bf[cur][n].y, acc[m][n], 0, 0, 0);
    }
    if (FLAGS & 1) __syncthreads();
  }
  float s = 0;
  for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int q = 0; q < 16; ++q) s += acc[m][n][q];
  out[blockIdx.x * 256 + threadIdx.x] = s + wreg[0][0];
}

template <int FLAGS>
void run(const char* name, const float* g, float* out) {
  const int blocks = 512, iters = 2000;
  const size_t lds = 436 * 34 * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k<FLAGS>), dim3(blocks), dim3(256), lds, 0, g, out, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<FLAGS>), dim3(blocks), dim3(256), lds, 0, g, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 8 * 8 * 4096.0;
  printf("%-58s %7.1f TFLOP/s (%.2f ms)\n", name, flops / ms / 1e9, ms);
}

int main() {
  float *g, *out;
  hipMalloc(&g, (size_t)512 * 4096 * 4 + 1024 * 1024 * 4 * 4); hipMalloc(&out, 512 * 256 * 4);
  hipMemset(g, 0, (size_t)512 * 4096 * 4 + 1024 * 1024 * 4 * 4);
  run<0>("LDS-fed 32x32x2, 2x2 tiles/wave, compiler schedule", g, out);
  run<8>("  + sched_barrier prefetch", g, out);
  run<9>("  + prefetch + barrier/iter", g, out);
  run<11>("  + prefetch + barrier + ds_write staging", g, out);
  run<15>("  + prefetch + barrier + ds_write + global loads", g, out);
  run<7>("  compiler schedule + barrier + ds_write + global loads", g, out);
  return 0;
}

// Implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (the f32 MFMA shape that sustains ~99 % of the 157 TFLOP/s
// peak on MI355X; the 16x16x4 shape tops out near 125 TFLOP/s) for column counts that are multiples of 32.
// Same structure as conv_mfma_impl.h: TH x 16 pixel tile per block, halo tile staged once per channel chunk and
// shared by all taps, weights double-buffered through VGPRs -> LDS, one barrier per (chunk, tap).
//   * a 32-row MFMA tile = 2 image rows x 16 pixels; lane (i = lane&31, kh = lane>>5) holds A[i][k=kh]
//   * LDS rows are k-contiguous with stride CK+2 floats: 32 consecutive rows then map to 32 distinct bank pairs
//     for ds_read_b64 (one 8-byte read feeds two MFMAs: k and k+1 of this lane's k-half)
//   * coupling epilogue: 32-column tile = [ s 16ch | t 16ch ], partner lane = lane ^ 16
#pragma once
#include "conv_mfma_impl.h"

namespace sininn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

extern int g_conv_dma;   // weights by LDS-DMA (test hook / default set in conv_mfma.hip)

template <int KS, int TH, int WM, int WN, int MT, int NT, int CK>
__global__ __launch_bounds__(256, 2) void conv32_kernel(ConvDev p) {
  constexpr int HALO = KS / 2;
  constexpr int IW = 16 + 2 * HALO;
  constexpr int IH = TH + 2 * HALO;
  constexpr int NPIX_IN = IH * IW;
  constexpr int TAPS = KS * KS;
  constexpr int BN = WN * NT * 32;
  constexpr int S = CK + 2;
  constexpr int C4N = CK / 4;
  constexpr int KSTEPS = CK / 4;
  constexpr int IN_F4 = (NPIX_IN * C4N + 255) / 256;
  constexpr int W_F4 = (BN * C4N + 255) / 256;
  static_assert(WM * MT * 2 == TH && WM * WN == 4, "bad wave layout");
  static_assert(CK % 8 == 0 && CK <= 32, "bad channel chunk");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const in_lds = smem;
  float* const w_lds0 = smem + NPIX_IN * S;
  float* const w_lds1 = w_lds0 + BN * S;

  const int tid = threadIdx.x;
  stamp_begin(p);
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, kh = lane >> 5;

  int bid = blockIdx.x;
  const int tx = bid % p.tiles_x; bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * 16;
  const int n0 = blockIdx.y * BN;

  int in_goff[IN_F4], in_loff[IN_F4];
#pragma unroll
  for (int r = 0; r < IN_F4; ++r) {
    const int f = tid + 256 * r;
    const int pix = f / C4N, c4 = f - pix * C4N;
    const int py = pix / IW, px = pix - py * IW;
    const int gy = y0 + py - HALO, gx = x0 + px - HALO;
    const bool inside = (pix < NPIX_IN);
    const bool inimg = inside && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    in_loff[r] = inside ? (pix * S + c4 * 4) : -1;
    in_goff[r] = inimg ? (((b * p.H + gy) * p.W + gx) * p.in_stride + c4 * 4) : -1;
  }
  int w_goff[W_F4], w_loff[W_F4];
#pragma unroll
  for (int r = 0; r < W_F4; ++r) {
    const int f = tid + 256 * r;
    const int row = f / C4N, c4 = f - row * C4N;
    const bool inside = row < BN;
    w_loff[r] = inside ? (row * S + c4 * 4) : -1;
    w_goff[r] = (inside && (n0 + row) < p.Np) ? ((n0 + row) * p.Cin + c4 * 4) : -1;
  }

  const int nchunks = p.Cin / CK;
  const int nit = nchunks * TAPS;

  f32x4 in_reg[IN_F4], w_reg[W_F4];
  auto load_in = [&](int chunk) {
#pragma unroll
    for (int r = 0; r < IN_F4; ++r) {
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      in_reg[r] = (in_goff[r] >= 0) ? *reinterpret_cast<const f32x4*>(p.in + in_goff[r] + chunk * CK) : z;
    }
  };
  auto st8 = [](float* dst, const f32x4& v) {      // rows are only 8-byte aligned (stride CK+2)
    *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
    *reinterpret_cast<float2*>(dst + 2) = make_float2(v[2], v[3]);
  };
  auto store_in = [&]() {
#pragma unroll
    for (int r = 0; r < IN_F4; ++r)
      if (in_loff[r] >= 0) st8(in_lds + in_loff[r], in_reg[r]);
  };
  auto load_w = [&](int it) {
    const int chunk = it / TAPS, tap = it - chunk * TAPS;
    const float* base = p.w + (size_t)tap * p.Np * p.Cin + chunk * CK;
#pragma unroll
    for (int r = 0; r < W_F4; ++r) {
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      w_reg[r] = (w_goff[r] >= 0) ? *reinterpret_cast<const f32x4*>(base + w_goff[r]) : z;
    }
  };
  auto store_w = [&](float* dst) {
#pragma unroll
    for (int r = 0; r < W_F4; ++r)
      if (w_loff[r] >= 0) st8(dst + w_loff[r], w_reg[r]);
  };

  int a_base[MT], b_base[NT];
#pragma unroll
  for (int m = 0; m < MT; ++m) a_base[m] = (((wm * MT + m) * 2 + (li >> 4)) * IW + (li & 15)) * S + 2 * kh;
#pragma unroll
  for (int n = 0; n < NT; ++n) b_base[n] = ((wn * NT + n) * 32 + li) * S + 2 * kh;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[m][n][q] = 0.f;

  load_in(0);
  load_w(0);
  store_in();
  store_w(w_lds0);
  if (nit > 1) load_w(1);
  __syncthreads();

  int chunk = 0, tap = 0;
  for (int it = 0; it < nit; ++it) {
    if (!(p.ablate & 1)) {
    if (it + 1 < nit) store_w(((it + 1) & 1) ? w_lds1 : w_lds0);
    if (it + 2 < nit) load_w(it + 2);
    }
    const bool last_tap = (tap == TAPS - 1);
    const bool next_in = last_tap && (chunk + 1 < nchunks) && !(p.ablate & 1);
    if (next_in) load_in(chunk + 1);

    const float* Bw = (it & 1) ? w_lds1 : w_lds0;
    const int dy = tap / KS, dx = tap - dy * KS;
    const float* A = in_lds + (dy * IW + dx) * S;
    float2 af[2][MT], bf[2][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) af[0][m] = *reinterpret_cast<const float2*>(A + a_base[m]);
#pragma unroll
    for (int n = 0; n < NT; ++n) bf[0][n] = *reinterpret_cast<const float2*>(Bw + b_base[n]);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < KSTEPS && !(p.ablate & 4)) {
#pragma unroll
        for (int m = 0; m < MT; ++m) af[nxt][m] = *reinterpret_cast<const float2*>(A + a_base[m] + (ks + 1) * 4);
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[nxt][n] = *reinterpret_cast<const float2*>(Bw + b_base[n] + (ks + 1) * 4);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][m].x, bf[cur][n].x, acc[m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][m].y, bf[cur][n].y, acc[m][n], 0, 0, 0);
    }
    if (!(p.ablate & 2)) __syncthreads();
    if (next_in) {
      store_in();
      __syncthreads();
    }
    if (last_tap) { tap = 0; ++chunk; } else { ++tap; }
  }

  // ---- epilogue phase 1: accumulators -> LDS tile T[pixel][BN+4]
  //      (lane holds D[row = (q&3) + 8*(q>>2) + 4*kh][col = li], q = 0..15; row = 16*(image row in pair) + px)
  {
    constexpr int TS = BN + 4;
    float* const T = smem;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = (q & 3) + 8 * (q >> 2) + 4 * kh;
          T[((wm * MT + m) * 32 + row) * TS + (wn * NT + n) * 32 + li] = acc[m][n][q];
        }
    __syncthreads();
    __shared__ float red[4];
    conv_epilogue_tile<TH, BN, 16>(p, T, b, y0, x0, n0, tid, red);
    stamp_end(p);
  }
}

// Same kernel with the weight tiles moved global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no
// ds_write).  The DMA destination is lane-linear, so weight rows are UNPADDED (128 B for CK = 32) and bank conflicts
// are avoided by an XOR swizzle of the 16-byte chunks applied on the per-lane global SOURCE address and on the read
// (chunk' = chunk ^ ((row >> 1) & 7)); requires CK == 32 and Np % BN == 0 (no zero-filled rows).
template <int KS, int TH, int WM, int WN, int MT, int NT, int CK>
__global__ __launch_bounds__(256, 2) void conv32_dma_kernel(ConvDev p) {
  constexpr int HALO = KS / 2;
  constexpr int IW = 16 + 2 * HALO;
  constexpr int IH = TH + 2 * HALO;
  constexpr int NPIX_IN = IH * IW;
  constexpr int TAPS = KS * KS;
  constexpr int BN = WN * NT * 32;
  constexpr int S = CK + 2;
  constexpr int C4N = CK / 4;
  constexpr int KSTEPS = CK / 4;
  constexpr int IN_F4 = (NPIX_IN * C4N + 255) / 256;
  constexpr int W_DMA = BN * 8 / 256;               // 16-byte chunks per thread per weight tile
  static_assert(CK == 32 && (BN * 8) % 256 == 0, "DMA variant: CK == 32, BN multiple of 32");
  static_assert(WM * MT * 2 == TH && WM * WN == 4, "bad wave layout");
  static_assert(CK % 8 == 0 && CK <= 32, "bad channel chunk");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const in_lds = smem;
  float* const w_lds0 = smem + NPIX_IN * S;
  float* const w_lds1 = w_lds0 + BN * 32;          // unpadded, swizzled

  const int tid = threadIdx.x;
  stamp_begin(p);
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, kh = lane >> 5;

  int bid = blockIdx.x;
  const int tx = bid % p.tiles_x; bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * 16;
  const int n0 = blockIdx.y * BN;

  int in_goff[IN_F4], in_loff[IN_F4];
#pragma unroll
  for (int r = 0; r < IN_F4; ++r) {
    const int f = tid + 256 * r;
    const int pix = f / C4N, c4 = f - pix * C4N;
    const int py = pix / IW, px = pix - py * IW;
    const int gy = y0 + py - HALO, gx = x0 + px - HALO;
    const bool inside = (pix < NPIX_IN);
    const bool inimg = inside && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    in_loff[r] = inside ? (pix * S + c4 * 4) : -1;
    in_goff[r] = inimg ? (((b * p.H + gy) * p.W + gx) * p.in_stride + c4 * 4) : -1;
  }
  int w_goff[W_DMA];
#pragma unroll
  for (int r = 0; r < W_DMA; ++r) {
    const int P = (r * 4 + wave) * 64 + lane;        // linear 16-byte chunk of the LDS tile this lane fills
    const int row = P >> 3, qp = P & 7;
    const int q = qp ^ ((row >> 1) & 7);             // logical k-chunk stored at physical chunk qp
    w_goff[r] = (n0 + row) * p.Cin + q * 4;
  }

  const int nchunks = p.Cin / CK;
  const int nit = nchunks * TAPS;

  f32x4 in_reg[IN_F4];
  auto load_in = [&](int chunk) {
#pragma unroll
    for (int r = 0; r < IN_F4; ++r) {
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      in_reg[r] = (in_goff[r] >= 0) ? *reinterpret_cast<const f32x4*>(p.in + in_goff[r] + chunk * CK) : z;
    }
  };
  auto st8 = [](float* dst, const f32x4& v) {      // rows are only 8-byte aligned (stride CK+2)
    *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
    *reinterpret_cast<float2*>(dst + 2) = make_float2(v[2], v[3]);
  };
  auto store_in = [&]() {
#pragma unroll
    for (int r = 0; r < IN_F4; ++r)
      if (in_loff[r] >= 0) st8(in_lds + in_loff[r], in_reg[r]);
  };
  auto dma_w = [&](int it, float* dst) {
    const int chunk = it / TAPS, tap = it - chunk * TAPS;
    const float* base = p.w + (size_t)tap * p.Np * p.Cin + chunk * CK;
#pragma unroll
    for (int r = 0; r < W_DMA; ++r)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + w_goff[r]),
                                       (__attribute__((address_space(3))) void*)(dst + (r * 4 + wave) * 256), 16, 0, 0);
  };

  int a_base[MT], b_base[NT], b_swz[NT];
#pragma unroll
  for (int m = 0; m < MT; ++m) a_base[m] = (((wm * MT + m) * 2 + (li >> 4)) * IW + (li & 15)) * S + 2 * kh;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int row = (wn * NT + n) * 32 + li;
    b_base[n] = row * 32 + 2 * kh;
    b_swz[n] = (row >> 1) & 7;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[m][n][q] = 0.f;

  load_in(0);
  dma_w(0, w_lds0);
  store_in();
  __syncthreads();                                   // vmcnt(0) + barrier: DMA'd tile visible to every wave

  int chunk = 0, tap = 0;
  for (int it = 0; it < nit; ++it) {
    if (!(p.ablate & 1) && it + 1 < nit) dma_w(it + 1, ((it + 1) & 1) ? w_lds1 : w_lds0);
    const bool last_tap = (tap == TAPS - 1);
    const bool next_in = last_tap && (chunk + 1 < nchunks) && !(p.ablate & 1);
    if (next_in) load_in(chunk + 1);

    const float* Bw = (it & 1) ? w_lds1 : w_lds0;
    const int dy = tap / KS, dx = tap - dy * KS;
    const float* A = in_lds + (dy * IW + dx) * S;
    float2 af[2][MT], bf[2][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) af[0][m] = *reinterpret_cast<const float2*>(A + a_base[m]);
#pragma unroll
    for (int n = 0; n < NT; ++n) bf[0][n] = *reinterpret_cast<const float2*>(Bw + b_base[n] + ((0 ^ b_swz[n]) << 2));
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < KSTEPS && !(p.ablate & 4)) {
#pragma unroll
        for (int m = 0; m < MT; ++m) af[nxt][m] = *reinterpret_cast<const float2*>(A + a_base[m] + (ks + 1) * 4);
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[nxt][n] = *reinterpret_cast<const float2*>(Bw + b_base[n] + (((ks + 1) ^ b_swz[n]) << 2));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][m].x, bf[cur][n].x, acc[m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][m].y, bf[cur][n].y, acc[m][n], 0, 0, 0);
    }
    if (!(p.ablate & 2)) __syncthreads();
    if (next_in) {
      store_in();
      __syncthreads();
    }
    if (last_tap) { tap = 0; ++chunk; } else { ++tap; }
  }

  // ---- epilogue phase 1: accumulators -> LDS tile T[pixel][BN+4]
  //      (lane holds D[row = (q&3) + 8*(q>>2) + 4*kh][col = li], q = 0..15; row = 16*(image row in pair) + px)
  {
    constexpr int TS = BN + 4;
    float* const T = smem;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = (q & 3) + 8 * (q >> 2) + 4 * kh;
          T[((wm * MT + m) * 32 + row) * TS + (wn * NT + n) * 32 + li] = acc[m][n][q];
        }
    __syncthreads();
    __shared__ float red[4];
    conv_epilogue_tile<TH, BN, 16>(p, T, b, y0, x0, n0, tid, red);
    stamp_end(p);
  }
}

template <int KS, int TH, int WM, int WN, int MT, int NT, int CK>
static int launch32_ck(const ConvDev& d, hipStream_t st) {
  constexpr int HALO = KS / 2;
  constexpr int NPIX_IN = (TH + 2 * HALO) * (16 + 2 * HALO);
  constexpr int BN = WN * NT * 32;
  constexpr int S = CK + 2;
  constexpr size_t lds_main = (size_t)(NPIX_IN + 2 * BN) * S * sizeof(float);
  constexpr size_t lds_epi = (size_t)TH * 16 * (BN + 4) * sizeof(float);
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  static_assert(lds <= 160 * 1024, "LDS tile too large");
  dim3 grid(d.tiles_x * d.tiles_y * d.B, (d.Np + BN - 1) / BN);
  if constexpr (CK == 32) {
    if (g_conv_dma && d.Np % BN == 0) {
      auto kd = conv32_dma_kernel<KS, TH, WM, WN, MT, NT, CK>;
      if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("conv32: cannot raise LDS limit to %zu", lds); return 1; }
      }
      hipLaunchKernelGGL(kd, grid, dim3(256), lds, st, d);
      SININN_LAUNCH_CHECK("conv32_dma");
      return 0;
    }
  }
  auto k = conv32_kernel<KS, TH, WM, WN, MT, NT, CK>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("conv32: cannot raise LDS limit to %zu", lds); return 1; }
  }
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, d);
  SININN_LAUNCH_CHECK("conv32");
  return 0;
}

template <int KS, int TH, int WM, int WN, int MT, int NT>
static int launch32(const ConvDev& d, hipStream_t st) {
  switch (d.CK) {
    case 32: return launch32_ck<KS, TH, WM, WN, MT, NT, 32>(d, st);
    case 24: return launch32_ck<KS, TH, WM, WN, MT, NT, 24>(d, st);
    case 16: return launch32_ck<KS, TH, WM, WN, MT, NT, 16>(d, st);
    case 8: return launch32_ck<KS, TH, WM, WN, MT, NT, 8>(d, st);
    default: set_error("conv32: unsupported channel chunk %d", d.CK); return 1;
  }
}

// returns -1 when no 32-wide configuration fits the shape well (caller falls back to the 16-wide kernel)
template <int KS>
static int dispatch32(ConvDev& d, hipStream_t st, int force_cfg, bool must) {
  const int nt32 = d.Np / 32;
  const long tiles8 = (long)d.B * ((d.H + 7) / 8) * ((d.W + 15) / 16);
  auto set_tiles = [&](int th) { d.tiles_x = (d.W + 15) / 16; d.tiles_y = (d.H + th - 1) / th; };
  if (force_cfg == 7 && nt32 == 1) { d.tiles_x = (d.W + 15) / 16; d.tiles_y = (d.H + 15) / 16; return launch32<KS, 16, 4, 1, 2, 1>(d, st); }
  if (force_cfg == 5) { set_tiles(8); return launch32<KS, 8, 4, 1, 1, 1>(d, st); }
  if (nt32 % 2 == 0 && force_cfg == 6) { set_tiles(4); return launch32<KS, 4, 2, 2, 1, 1>(d, st); }
  if (KS == 3 && nt32 % 2 == 0 && force_cfg == 4) { d.tiles_x = (d.W + 15) / 16; d.tiles_y = (d.H + 15) / 16; return launch32<KS, 16, 4, 1, 2, 2>(d, st); }
  if ((KS == 1 && nt32 % 2 == 0 && force_cfg == 0) || (nt32 % 8 == 0 && force_cfg == 0)) {
    // narrow column blocks (64): ~35-42 KB of LDS -> 3-4 blocks per CU and several rounds of blocks per launch, so
    // the load / MFMA / store phases of different blocks overlap (1x1: one K iteration per block, latency / HBM
    // bound, 23 -> 37 TF/s; 3x3 with 256 columns: +5..9 % measured on one box against 128-column blocks)
    set_tiles(8); return launch32<KS, 8, 2, 2, 2, 1>(d, st);
  }
  if (nt32 % 4 == 0) {
    bool small = tiles8 * (nt32 / 4) < 512;
    if (force_cfg == 1) small = false;
    if (force_cfg == 2) small = true;
    if (!small) { set_tiles(8); return launch32<KS, 8, 2, 2, 2, 2>(d, st); }
    set_tiles(4); return launch32<KS, 4, 2, 2, 1, 2>(d, st);
  }
  if (nt32 % 3 == 0) {
    // small images (level-1 tensors): 32-column blocks when that gives >= 3 balanced blocks per CU (+11 % measured)
    if (force_cfg == 0 && tiles8 * (nt32 / 3) < 512 && tiles8 * nt32 >= 768 && (tiles8 * nt32) % 256 == 0) {
      set_tiles(8); return launch32<KS, 8, 4, 1, 1, 1>(d, st);
    }
    if (!must && force_cfg == 0 && tiles8 * (nt32 / 3) < 256) return -1;
    set_tiles(8); return launch32<KS, 8, 4, 1, 1, 3>(d, st);
  }
  if (nt32 == 2) { set_tiles(8); return launch32<KS, 8, 2, 2, 2, 1>(d, st); }
  if (nt32 == 1) { set_tiles(8); return launch32<KS, 8, 4, 1, 1, 1>(d, st); }
  if (must) { set_error("conv32: no configuration for Np=%d", d.Np); return 1; }
  return -1;
}

}  // namespace sininn

// Winograd F(2x2, 3x3) convolution on v_mfma_f32_32x32x2_f32 (the 16x16x4 form tops out ~25 % lower on MI355X,
// tools/mfma_peak.hip).  Same maths, block shape, LDS staging and epilogue as wino_impl.h; what changes is the wave tile:
//
//   * wave = 32 Winograd tiles (4 tile rows x 8) x 32 columns x 8 of the 16 transform positions: the two "position
//     halves" (rows a = 0,1 / a = 2,3 of V = B^T d B) of a tile group live in two waves.  16 accumulator registers per
//     position -> 128, as before, but each wave does HALF of the input transform (3 of the 4 patch rows, 2 of the 4
//     transformed rows), and the LDS traffic moves as 16-byte reads.
//   * MFMA operand mapping: lane (i = lane & 31, kq = lane >> 5).  A = V_pos[tile i][channel 4*kq + j], B =
//     U_pos[column i][channel 4*kq + j] for the j-th of the 4 MFMAs of a position (each MFMA sums channels j and 4 + j).
//   * output transform: Y = A^T M A splits over the position rows: each wave forms the partial 2x2 result of its two rows
//     and writes it to the LDS tile of its half (T0 / T1); the shared float4 epilogue adds the two tiles on the fly.
#pragma once
#include "conv_mfma_impl.h"

namespace sininn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OFF>
__device__ __forceinline__ f32x4 lds_read_b128(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536 && OFF % 16 == 0, "ds_read_b128 immediate offset");
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void lds_wait(f32x4& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }

template <int HT, int CG>
__global__ __launch_bounds__(256 * CG, 2 / CG) void wino32_kernel(ConvDev p) {
  constexpr int CK = 8;
  constexpr int NTHR = 256 * CG;
  constexpr int IW = 18, NPIX_IN = 18 * 18;
  constexpr int BG = 32;                                // columns of one wave group
  constexpr int BN = CG * BG;                           // columns of the block
  constexpr int SI = 12, SU = 12;                       // LDS pixel / column strides (floats): 8 channels + 4 pad
  // row pitch 224 + a 4-float skew on every second row pair: conflict-free ds_read_b128 for every patch offset
  // (brute-forced over the four 16-lane groups of the instruction, MI355X_MICROARCH.md LDS table)
  constexpr int PITCH = 224;
  constexpr int IN_F4 = (NPIX_IN * 2 + NTHR - 1) / NTHR;   // 2 float4 per pixel
  constexpr int U_F4 = (16 * BN * 2 + NTHR - 1) / NTHR;
  constexpr int IN_BUF = IW * PITCH + 4, U_BUF = 16 * BN * SU;
  constexpr bool U_EXACT = (16 * BN * 2) % NTHR == 0;      // every thread slot of the U staging loop is a real quad

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const in_lds0 = smem;
  float* const in_lds1 = smem + IN_BUF;
  float* const u_lds0 = smem + 2 * IN_BUF;
  float* const u_lds1 = u_lds0 + U_BUF;

  const int tid = threadIdx.x;
  stamp_begin(p);
  const int wv = (tid >> 6) & 3, grp = tid >> 8, lane = tid & 63;
  const int tg = wv >> 1, ph = wv & 1;                  // tile group (32 tiles), position half
  const int li = lane & 31, kq = lane >> 5;

  int bid = blockIdx.x;
  const int tx = bid % p.tiles_x; bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int y0 = ty * 16, x0 = tx * 16;
  const int n0 = blockIdx.y * BN;

  // ---- staging descriptors ---------------------------------------------------------------------
  // global side: byte offsets for raw buffer loads (BUF_OOB outside the image / beyond the packed columns -> the load returns
  // zeros), relative to this block's image; the channel chunk advances through the scalar offset operand
  unsigned in_goff[IN_F4]; int in_loff[IN_F4];
#pragma unroll
  for (int r = 0; r < IN_F4; ++r) {
    const int f = tid + NTHR * r;
    const int pix = f >> 1, c4 = f & 1;
    const int py = pix / IW, px = pix - py * IW;
    const int gy = y0 + py - 1, gx = x0 + px - 1;
    const bool inside = pix < NPIX_IN;
    const bool inimg = inside && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    // slots beyond the halo tile store their (zero) quad into the four spare floats behind the tile: no exec-mask branch per store
    in_loff[r] = inside ? (py * PITCH + ((py >> 1) & 1) * 4 + px * SI + c4 * 4) : IW * PITCH;
    in_goff[r] = inimg ? (unsigned)(((gy * p.W + gx) * p.in_stride + c4 * 4) * 4) : BUF_OOB;
  }
  unsigned u_goff[U_F4]; int u_loff[U_F4];
#pragma unroll
  for (int r = 0; r < U_F4; ++r) {
    const int f = tid + NTHR * r;
    const int c4 = f & 1, col = (f >> 1) % BN, pos = (f >> 1) / BN;
    const bool inside = pos < 16;
    u_loff[r] = (U_EXACT || inside) ? ((pos * BN + col) * SU + c4 * 4) : -1;
    // U pack = [16 positions][Cin / 8 chunks][Np columns][8 channels]
    u_goff[r] = (inside && (n0 + col) < p.Np) ? (unsigned)(((pos * (p.Cin / CK) * p.Np + n0 + col) * CK + c4 * 4) * 4) : BUF_OOB;
  }
  const int nchunks = p.Cin / CK;

  const __amdgpu_buffer_rsrc_t in_rs = buf_rsrc(p.in + (size_t)b * p.H * p.W * p.in_stride), u_rs = buf_rsrc(p.w);
  const unsigned in_step = (unsigned)p.in_chunk * 4u, u_step = (unsigned)(p.Np * CK) * 4u;      // bytes per channel chunk
  f32x4 in_reg[IN_F4], u_reg[U_F4];
  auto load_chunk = [&](int chunk) {
#pragma unroll
    for (int r = 0; r < IN_F4; ++r) in_reg[r] = buf_load4(in_rs, in_goff[r], (unsigned)chunk * in_step);
#pragma unroll
    for (int r = 0; r < U_F4; ++r) u_reg[r] = buf_load4(u_rs, u_goff[r], (unsigned)chunk * u_step);
  };
  auto store_chunk = [&](float* idst, float* udst) {
#pragma unroll
    for (int r = 0; r < IN_F4; ++r)
      *reinterpret_cast<f32x4*>(idst + in_loff[r]) = in_reg[r];
#pragma unroll
    for (int r = 0; r < U_F4; ++r)
      if (U_EXACT || u_loff[r] >= 0) *reinterpret_cast<f32x4*>(udst + u_loff[r]) = u_reg[r];
  };

  // this lane's Winograd tile (A operand) and the three patch rows (X, Y, Z) its position half needs, ordered so that
  // both halves run the same arithmetic:  t0 = X - Z,  t1 = Z + sgn * Y
  //   rows a = 0,1 (ph 0):  X = d0, Y = d1, Z = d2, sgn = +1      rows a = 2,3 (ph 1):  X = d2, Y = d3, Z = d1, sgn = -1
  const int a_ty = 4 * tg + (li >> 3), a_tx = li & 7;
  const float sgn = ph ? -1.f : 1.f;
  const f32x2 sgn2 = {sgn, sgn};
  int a_row[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int prow = ph ? (r == 0 ? 2 : (r == 1 ? 3 : 1)) : r;
    const int py = 2 * a_ty + prow;
    a_row[r] = py * PITCH + ((py >> 1) & 1) * 4 + (2 * a_tx) * SI + 4 * kq;
  }
  const int b_base = ((2 * ph) * 4 * BN + grp * BG + li) * SU + 4 * kq;   // first position of this half: q = 8 * ph

  f32x16 acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  load_chunk(0);
  store_chunk(in_lds0, u_lds0);
  if (nchunks > 1) load_chunk(1);
  __syncthreads();

  for (int it = 0; it < nchunks; ++it) {
    if (!(p.ablate & 1)) {
      if (it + 1 < nchunks) store_chunk(((it + 1) & 1) ? in_lds1 : in_lds0, ((it + 1) & 1) ? u_lds1 : u_lds0);
      if (it + 2 < nchunks) load_chunk(it + 2);
    }
    const float* ibuf = (it & 1) ? in_lds1 : in_lds0;
    const unsigned R0 = lds_addr(ibuf + a_row[0]), R1 = lds_addr(ibuf + a_row[1]), R2 = lds_addr(ibuf + a_row[2]);
    const unsigned Uc = lds_addr(((it & 1) ? u_lds1 : u_lds0) + b_base);

    // ---- half input transform; 4 channels per lane, as two packed pairs (v_pk_add_f32) --------------------------
    f32x2 v[2][4][2];                                  // [transformed row][column][channel pair]
    {
      f32x4 x[4], y[4], z[4];
      static_for<0, 4>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        x[c] = lds_read_b128<c * SI * 4>(R0);
        z[c] = lds_read_b128<c * SI * 4>(R2);
      });
      static_for<0, 4>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        y[c] = lds_read_b128<c * SI * 4>(R1);
      });
      asm volatile("s_waitcnt lgkmcnt(4)"
                   : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]));
      f32x2 t[2][4][2];
#pragma unroll
      for (int c = 0; c < 4; ++c) { t[0][c][0] = x[c].xy - z[c].xy; t[0][c][1] = x[c].zw - z[c].zw; }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
#pragma unroll
      for (int c = 0; c < 4; ++c) { t[1][c][0] = z[c].xy + sgn2 * y[c].xy; t[1][c][1] = z[c].zw + sgn2 * y[c].zw; }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          v[h][0][e] = t[h][0][e] - t[h][2][e];
          v[h][1][e] = t[h][1][e] + t[h][2][e];
          v[h][2][e] = t[h][2][e] - t[h][1][e];
          v[h][3][e] = t[h][1][e] - t[h][3][e];
        }
    }
    // ---- 8 positions x 4 k-steps; the U fragment of position q + PF is requested before the MFMAs of q ---------
    // positions are processed in PAIRS with their MFMAs interleaved (q, q+1, q, q+1, ...): four back-to-back MFMAs on
    // one accumulator would each wait for the previous one's 64-cycle result latency
    f32x4 bf[4];
    bf[0] = lds_read_b128<(0 * BN * SU) * 4>(Uc);
    bf[1] = lds_read_b128<(1 * BN * SU) * 4>(Uc);
    static_for<0, 4>([&](auto pp) {
      constexpr int q = 2 * decltype(pp)::value;       // pair (q, q+1); fragments in bf[(q & 2)], bf[(q & 2) + 1]
      if constexpr (q + 2 < 8) {
        bf[((q + 2) & 2) + 0] = lds_read_b128<((q + 2) * BN * SU) * 4>(Uc);
        bf[((q + 2) & 2) + 1] = lds_read_b128<((q + 3) * BN * SU) * 4>(Uc);
      }
      constexpr int newer = (q + 2 < 8) ? 2 : 0;
      asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bf[(q & 2) + 0]), "+v"(bf[(q & 2) + 1]) : "n"(newer));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[q >> 2][q & 3][j >> 1][j & 1], bf[(q & 2) + 0][j], acc[q], 0, 0, 0);
        acc[q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[(q + 1) >> 2][(q + 1) & 3][j >> 1][j & 1], bf[(q & 2) + 1][j], acc[q + 1], 0, 0, 0);
      }
    });
    if (!(p.ablate & 2)) __syncthreads();
  }

  // ---- partial output transform of this wave's two position rows -> LDS tile T0 / T1 [pixel][BN+4] -------------
  // acc[h*4 + c][r] = M[a = 2*ph + h][c] of (tile i = (r & 3) + 8 * (r >> 2) + 4 * kq, column li)
  {
    constexpr int TS = BN + 4;
    float* const T0 = smem;
    float* const T1 = smem + 256 * TS;
    float* const T = ph ? T1 : T0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float R[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        R[h][0] = acc[h * 4 + 0][r] + acc[h * 4 + 1][r] + acc[h * 4 + 2][r];
        R[h][1] = acc[h * 4 + 1][r] - acc[h * 4 + 2][r] - acc[h * 4 + 3][r];
      }
      // rows a = 0,1:  Y0 = R0 + R1, Y1 = R1       rows a = 2,3:  Y0 = R0, Y1 = -R0 - R1
      const float y00 = ph ? R[0][0] : R[0][0] + R[1][0];
      const float y01 = ph ? R[0][1] : R[0][1] + R[1][1];
      const float y10 = ph ? -R[0][0] - R[1][0] : R[1][0];
      const float y11 = ph ? -R[0][1] - R[1][1] : R[1][1];
      const int i = (r & 3) + 8 * (r >> 2) + 4 * kq;
      const int oy = 2 * (4 * tg + (i >> 3)), ox = 2 * (i & 7);
      float* t00 = T + (oy * 16 + ox) * TS + grp * BG + li;
      t00[0] = y00; t00[TS] = y01; t00[16 * TS] = y10; t00[17 * TS] = y11;
    }
    __syncthreads();
    __shared__ float red[4 * CG];
    conv_epilogue_tile<16, BN, HT, NTHR>(p, T0, b, y0, x0, n0, tid, red, T1);
    stamp_end(p);
  }
}

// Only the 4-wave / 32-column variant with the 16-column coupling interleave is instantiated: this kernel is dispatched for
// long-K layers that leave one wave per SIMD (few blocks), where it measured 7-8 % faster than the 16x16x4 kernel; with two
// waves per SIMD the two are equal or the 16x16x4 one is 1-3 % ahead (tools/bench_kernels.py --wino 1 --cfg 4 vs 0).
static int wino32_launch(ConvDev& d, hipStream_t st) {
  constexpr int CG = 1, BN = CG * 32;
  constexpr size_t lds_main = (size_t)(2 * (18 * 224 + 4) + 2 * 16 * BN * 12) * sizeof(float);
  constexpr size_t lds_epi = (size_t)2 * 256 * (BN + 4) * sizeof(float);
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  static_assert(lds <= 160 * 1024, "LDS tile too large");
  d.tiles_x = (d.W + 15) / 16;
  d.tiles_y = (d.H + 15) / 16;
  dim3 grid(d.tiles_x * d.tiles_y * d.B, (d.Np + BN - 1) / BN);
  auto k = wino32_kernel<8, CG>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("wino32: cannot raise LDS limit to %zu", lds); return 1; }
  hipLaunchKernelGGL(k, grid, dim3(256 * CG), lds, st, d);
  SININN_LAUNCH_CHECK("wino32");
  return 0;
}

}  // namespace sininn

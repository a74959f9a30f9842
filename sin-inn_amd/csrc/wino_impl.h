// Winograd F(2x2, 3x3) convolution on v_mfma_f32_16x16x4_f32: 2.25x fewer MFMA FLOPs than the direct
// implicit GEMM for the 3x3 subnet convs (archs.py:11-13) and their data gradients.
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
//   * block  = 16x16 output pixels (8x8 Winograd tiles) x (CG * 32) packed columns; 4 * CG waves: wave = (column
//     group, 16 tiles).  With CG = 2 the two column groups share ONE staged halo tile: a 256-channel input is then
//     read once per 64 columns instead of once per 32 (PMC: the Np=64 coupling conv fetched 198 MB for a 67 MB input).
//   * the filter transform U = G g G^T is done once per weight update by the pack kernel:
//     U[pos 16][column][cin] (k-contiguous rows), so per position the kernel runs a plain
//     [16 tiles x K] x [K x 16 columns] MFMA product, accumulating all 16 positions in registers
//     (16 pos x 2 column tiles x 4 regs = 128 accumulator VGPRs per lane).
//   * the input transform V = B^T d B is done PER LANE in registers: the MFMA A operand of lane (tile i, k-lane kq)
//     for position p is V_p[tile i][channel kq] -- exactly what the lane obtains by transforming the 4x4 patch of its
//     own (tile, channel pair) read from the LDS halo tile; no transformed tile is ever stored.
//   * the output transform Y = A^T M A is also per lane: the D layout keeps all 16 positions of a (tile, column) pair
//     in the same lane.  The 2x2 results go through the LDS tile T[pixel][32+4] and the shared float4 epilogue
//     (ReLU / coupling / mask / add), exactly like the direct kernels.
//   * K loop: channel chunks of 8 (one ds_read_b64 = channels 2kq, 2kq+1 -> two MFMAs), halo tile and U chunk
//     double-buffered through VGPRs -> LDS, one barrier per chunk.
#pragma once
#include "conv_mfma_impl.h"
#include "wino32_impl.h"

namespace sininn {

template <int NT, int HT, int CG>
__global__ __launch_bounds__(256 * CG, 2 / CG) void wino_kernel(ConvDev p) {
  constexpr int CK = 8;
  constexpr int NTHR = 256 * CG;
  constexpr int IW = 18, NPIX_IN = 18 * 18;
  constexpr int BG = NT * 16;                           // columns of one wave group
  constexpr int BN = CG * BG;                           // columns of the block
  constexpr int SI = 12, SU = 12;                       // LDS pixel / column strides (floats): 8 channels + 4 pad
  // halo-tile row pitch + a 4-float skew on every second row pair: the two tile rows a wave reads with one
  // ds_read_b64 then fall on disjoint banks (tile step 24 floats = multiples of 8 banks; row-pair step 440 +- 4)
  constexpr int PITCH = IW * SI + 4;
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
  const int wave = (tid >> 6) & 3, grp = tid >> 8, lane = tid & 63;   // wave: tile group, grp: column group
  const int li = lane & 15, kq = lane >> 4;

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

  // this lane's Winograd tile for the A operand (tile = 16*wave + li) and its patch origin in the halo tile
  const int a_ty = 2 * wave + (li >> 3), a_tx = li & 7;
  // patch rows a = 0,1 carry the skew of row pair a_ty, rows a = 2,3 that of row pair a_ty + 1
  const int a_base01 = (2 * a_ty) * PITCH + (a_ty & 1) * 4 + (2 * a_tx) * SI + 2 * kq;
  const int a_base23 = (2 * a_ty) * PITCH + ((a_ty + 1) & 1) * 4 + (2 * a_tx) * SI + 2 * kq;
  const int b_base = (grp * BG + li) * SU + 2 * kq;

  f32x4 acc[16][NT];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[q][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // a wave whose second column tile lies entirely beyond the packed columns (Np = 48 in a 64-column block: the level-0
  // coupling conv) skips that tile's MFMAs: its SIMD partner of the other column group then has the matrix pipe to itself
  // for a quarter of the time (wave-uniform branch; the U fragments are still read so the counted waits stay valid)
  const bool tile1_live = (n0 + grp * BG + 16) < p.Np;

  load_chunk(0);
  store_chunk(in_lds0, u_lds0);
  if (nchunks > 1) load_chunk(1);
  __syncthreads();

  for (int it = 0; it < nchunks; ++it) {
    if (!(p.ablate & 1)) {
      if (it + 1 < nchunks) store_chunk(((it + 1) & 1) ? in_lds1 : in_lds0, ((it + 1) & 1) ? u_lds1 : u_lds0);
      if (it + 2 < nchunks) load_chunk(it + 2);
    }
    const unsigned A01 = lds_addr(((it & 1) ? in_lds1 : in_lds0) + a_base01);
    const unsigned A23 = lds_addr(((it & 1) ? in_lds1 : in_lds0) + a_base23);
    const unsigned Uc = lds_addr(((it & 1) ? u_lds1 : u_lds0) + b_base);

    // ---- input transform V = B^T d B for (this lane's tile, channels 2kq / 2kq+1) -----------------
    f32x2 v[4][4];
    {
      f32x2 d[4][4];
      static_for<0, 16>([&](auto i) {
        constexpr int a = decltype(i)::value >> 2, c = decltype(i)::value & 3;
        d[a][c] = lds_read_b64<(a * PITCH + c * SI) * 4>(a < 2 ? A01 : A23);
      });
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[0][2]), "+v"(d[0][3]), "+v"(d[1][0]), "+v"(d[1][1]), "+v"(d[1][2]),
                     "+v"(d[1][3]), "+v"(d[2][0]), "+v"(d[2][1]), "+v"(d[2][2]), "+v"(d[2][3]), "+v"(d[3][0]), "+v"(d[3][1]),
                     "+v"(d[3][2]), "+v"(d[3][3]));
      f32x2 t[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        t[0][c] = d[0][c] - d[2][c];
        t[1][c] = d[1][c] + d[2][c];
        t[2][c] = d[2][c] - d[1][c];
        t[3][c] = d[1][c] - d[3][c];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        v[a][0] = t[a][0] - t[a][2];
        v[a][1] = t[a][1] + t[a][2];
        v[a][2] = t[a][2] - t[a][1];
        v[a][3] = t[a][1] - t[a][3];
      }
    }
    // ---- 16 positions x NT column tiles x 2 k-steps -----------------------------------------------
    // the U fragments of position q + PF are requested before the MFMAs of position q issue, so the matrix pipe
    // does not idle for the LDS latency at every position
    constexpr int PF = 2;
    static_assert(NT == 2, "the wait counts below assume two column tiles per wave");
    f32x2 bf[PF + 1][NT];
    auto positions = [&](auto live1_tag) {
      constexpr bool LIVE1 = decltype(live1_tag)::value;
      static_for<0, PF>([&](auto qq) {
        constexpr int q = decltype(qq)::value;
        bf[q][0] = lds_read_b64<((q * BN) * SU) * 4>(Uc);
        if constexpr (LIVE1) bf[q][1] = lds_read_b64<((q * BN + 16) * SU) * 4>(Uc);
      });
      static_for<0, 16>([&](auto qq) {
        constexpr int q = decltype(qq)::value;
        if constexpr (q + PF < 16) {
          bf[(q + PF) % (PF + 1)][0] = lds_read_b64<(((q + PF) * BN) * SU) * 4>(Uc);
          if constexpr (LIVE1) bf[(q + PF) % (PF + 1)][1] = lds_read_b64<(((q + PF) * BN + 16) * SU) * 4>(Uc);
        }
        constexpr int newer = (q + PF < 16 ? PF : 15 - q) * (LIVE1 ? 2 : 1);     // reads issued after this position's
        if constexpr (LIVE1) lds_wait<newer>(bf[q % (PF + 1)][0], bf[q % (PF + 1)][1]);
        else lds_wait<newer>(bf[q % (PF + 1)][0]);
        acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[q >> 2][q & 3].x, bf[q % (PF + 1)][0].x, acc[q][0], 0, 0, 0);
        if constexpr (LIVE1) acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[q >> 2][q & 3].x, bf[q % (PF + 1)][1].x, acc[q][1], 0, 0, 0);
        acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[q >> 2][q & 3].y, bf[q % (PF + 1)][0].y, acc[q][0], 0, 0, 0);
        if constexpr (LIVE1) acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[q >> 2][q & 3].y, bf[q % (PF + 1)][1].y, acc[q][1], 0, 0, 0);
      });
    };
    if (tile1_live) positions(std::true_type{}); else positions(std::false_type{});
    if (!(p.ablate & 2)) __syncthreads();
  }

  // ---- output transform Y = A^T M A per lane, 2x2 results -> LDS tile T[pixel][BN+4] ---------------
  // lane holds M_q[tile = 16*wave + 4*kq + r][col = li] in acc[q][n][r]
  {
    constexpr int TS = BN + 4;
    float* const T = smem;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float m[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) m[q] = acc[q][n][r];
        float r0[4], r1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          r0[c] = m[0 * 4 + c] + m[1 * 4 + c] + m[2 * 4 + c];
          r1[c] = m[1 * 4 + c] - m[2 * 4 + c] - m[3 * 4 + c];
        }
        const int tl = 4 * kq + r;
        const int oy = 2 * (2 * wave + (tl >> 3)), ox = 2 * (tl & 7);
        float* t00 = T + (oy * 16 + ox) * TS + grp * BG + n * 16 + li;
        t00[0] = r0[0] + r0[1] + r0[2];
        t00[TS] = r0[1] - r0[2] - r0[3];
        t00[16 * TS] = r1[0] + r1[1] + r1[2];
        t00[17 * TS] = r1[1] - r1[2] - r1[3];
      }
    __syncthreads();
    __shared__ float red[4 * CG];
    conv_epilogue_tile<16, BN, HT, NTHR>(p, T, b, y0, x0, n0, tid, red);
    stamp_end(p);
  }
}

template <int CG>
static int wino_launch(ConvDev& d, hipStream_t st) {
  constexpr int NT = 2, BN = CG * NT * 16;
  constexpr size_t lds_main = (size_t)(2 * (18 * (18 * 12 + 4) + 4) + 2 * 16 * BN * 12) * sizeof(float);
  constexpr size_t lds_epi = (size_t)256 * (BN + 4) * sizeof(float);
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  static_assert(lds <= 160 * 1024, "LDS tile too large");
  d.tiles_x = (d.W + 15) / 16;
  d.tiles_y = (d.H + 15) / 16;
  dim3 grid(d.tiles_x * d.tiles_y * d.B, (d.Np + BN - 1) / BN);
  // HT = half-width of the coupling (s|t) column interleave the weights were packed with (16 or 8)
  auto k = (d.col_tile == 16) ? wino_kernel<NT, 8, CG> : wino_kernel<NT, 16, CG>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("wino: cannot raise LDS limit to %zu", lds); return 1; }
  hipLaunchKernelGGL(k, grid, dim3(256 * CG), lds, st, d);
  SININN_LAUNCH_CHECK("wino");
  return 0;
}

// cg2 (test hook): 0 auto, 1 force 32-column blocks, 2 force 64-column blocks; +4: never the 32x32x2 kernel
// (wino32_impl.h); +8: the 32x32x2 kernel wherever it applies (32-column blocks, 16-column coupling interleave)
static int wino_dispatch(ConvDev& d, hipStream_t st, int cg2 = 0) {
  const bool never32 = (cg2 & 4) != 0, always32 = (cg2 & 8) != 0;
  cg2 &= 3;
  // 64-column blocks (8 waves sharing one halo tile) when the columns split evenly and the input is wide enough
  // for its re-reads to matter
  // (an even number of 32-column groups; the last group may be partial, as with 32-column blocks)
  const bool even_groups = ((d.Np + 31) / 32) % 2 == 0;
  bool wide = even_groups && d.Cin >= 64;
  if (cg2 == 1) wide = false;
  if (cg2 == 2) wide = even_groups;
  if (wide) return wino_launch<2>(d, st);
  // long K and few blocks (at most one 4-wave block per CU -> one wave per SIMD): the 32x32x2 kernel
  const long blocks = (long)d.B * ((d.H + 15) / 16) * ((d.W + 15) / 16) * ((d.Np + 31) / 32);
  const bool use32 = d.col_tile == 16 && !never32 && (always32 || (d.Cin >= 128 && blocks <= 256));
  return use32 ? wino32_launch(d, st) : wino_launch<1>(d, st);
}

}  // namespace sininn

// Weight gradient of the subnet convolutions (backward of nn.Conv2d in archs.py:11-17) on
// v_mfma_f32_16x16x4_f32.
//
// GEMM view (reduction over pixels):
//     dW[tap][n][c] = sum_pix dout[pix][n] * in[pix + off(tap)][c]        db[n] = sum_pix dout[pix][n]
//   * MFMA rows = output channels n, MFMA cols = input channels c, k = 4 consecutive pixels of a row
//   * a block owns (RT*16 rows) x (CT*16 cols) x ALL taps of the output and walks a contiguous range
//     of 8x16-pixel tiles (split-K over pixels); the `in` halo tile is staged once per pixel tile
//     and shared by the 9 taps; the dout A-fragment is shared by all taps / column tiles of a wave
//   * partial sums go to a slab per split; a second kernel reduces the slabs in a fixed order
//     (bitwise reproducible, no float atomics) straight into the OIHW gradient (+=).
#include <stdlib.h>

#include "common.h"

namespace sininn {

struct WgradDev {
  const float* in; int in_stride; int Cin;
  const float* dout; int dout_stride; int N;
  int B, H, W;
  int tiles_x, tiles_y, ntiles, tiles_per_split;
  float* partial;   // [S][taps][Nr][Cc]
  float* bpartial;  // [S][Nr]
  int Nr, Cc;
  size_t in_gs, dout_gs;    // > 0: channel-group-major fp32 operand [C/8][pixel][8], floats between groups (Winograd kernels)
  int in_bf16, dout_bf16;   // mixed-precision path: the operand lives in HBM as bf16 (strides in elements); converted to
                            // fp32 while it is staged, the gradient itself accumulates on the f32 matrix pipe
};

typedef __bf16 wg_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 wg_load4(const float* base, size_t off, int is_bf16) {
  if (is_bf16) {
    const wg_bf16x4 v = *reinterpret_cast<const wg_bf16x4*>(reinterpret_cast<const __bf16*>(base) + off);
    return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  }
  return *reinterpret_cast<const f32x4*>(base + off);
}

// ------------------------------------------------------------------------------------------------
// Staging of one operand tile (pixel-major, 16 bytes per slot) of a weight-gradient block: global -> registers.
// The per-tile address arithmetic is issue time taken from the matrix pipe (tools/mfma_valu.hip), so every thread slot keeps,
// constant over the tiles, its BYTE offset from the tile's origin pixel -- the halo corner (y0 - HALO, x0 - HALO), so offsets
// are never negative -- and its tile-local (row, column).  Per tile a slot then costs two adds, two unsigned compares and ONE
// load with a scalar (wave-uniform) base and a 32-bit lane offset.  A statically dead slot (beyond the tile or the channel
// range) carries a position that fails every bounds check.  QPP = float4 slots per pixel, TW = tile width in pixels (with
// halo), NPX = pixels of the tile (with halo).
// ------------------------------------------------------------------------------------------------
template <int F4, int NTHR, int QPP, int TW, int NPX, int HALO>
struct WgStage {
  unsigned off[F4], pos[F4];
  unsigned pix_bytes;

  // stride: elements per pixel (pixel-major) -- or gs > 0: channel-group-major [C/8][pixel][8], gs elements between groups;
  // ch0: first channel of the block, chmax: channels of the tensor
  __device__ __forceinline__ void init(int tid, int W, int stride, size_t gs, int is_bf16, int ch0, int chmax) {
    const unsigned es = is_bf16 ? 2u : 4u, ps = gs ? 8u : (unsigned)stride;
    pix_bytes = ps * es;
#pragma unroll
    for (int r = 0; r < F4; ++r) {
      const int f = tid + NTHR * r;
      const int pix = f / QPP, ch = ch0 + (f % QPP) * 4;
      const int py = pix / TW, px = pix - py * TW;
      const unsigned col = gs ? (unsigned)((size_t)(ch >> 3) * gs) + (unsigned)(ch & 7) : (unsigned)ch;
      off[r] = (((unsigned)py * (unsigned)W + (unsigned)px) * ps + col) * es;
      pos[r] = (pix < NPX && ch < chmax) ? (unsigned)(py << 16 | px) : 0xffffffffu;     // fails both checks while H, W < 65535
    }
  }
  // tile origin (image b, pixel row y0, column x0: WITHOUT the halo) in an image of H x W pixels
  template <bool BF16>
  __device__ __forceinline__ void load(f32x4 (&reg)[F4], const void* tensor, int b, int y0, int x0, int H, int W) const {
    // wave-uniform 64-bit base, forced into SGPRs (the halo corner of the first tile row / column lies before the image: only
    // slots that pass the bounds check are dereferenced)
    const long long o = (((long long)b * H + (y0 - HALO)) * W + (x0 - HALO)) * (long long)pix_bytes;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)o), hi = __builtin_amdgcn_readfirstlane((unsigned)(o >> 32));
    const char* const base = reinterpret_cast<const char*>(tensor) + (long long)(((unsigned long long)hi << 32) | lo);
    const unsigned yo = (unsigned)(y0 - HALO), xo = (unsigned)(x0 - HALO);
#pragma unroll
    for (int r = 0; r < F4; ++r) {
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if ((yo + (pos[r] >> 16)) < (unsigned)H && (xo + (pos[r] & 0xffffu)) < (unsigned)W) {
        if constexpr (BF16) {
          const wg_bf16x4 h = *reinterpret_cast<const wg_bf16x4*>(base + off[r]);
          val = (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        } else {
          val = *reinterpret_cast<const f32x4*>(base + off[r]);
        }
      }
      reg[r] = val;
    }
  }
};

// both operand tiles of a block for one pixel tile; the precision branch is taken once per tile, not once per slot
template <class SD, class SI, int DF, int IF>
__device__ __forceinline__ void wg_load_tile(const WgradDev& p, const SD& ds, const SI& is, f32x4 (&d_reg)[DF], f32x4 (&i_reg)[IF],
                                             int b, int y0, int x0) {
  if (!(p.dout_bf16 | p.in_bf16)) {
    ds.template load<false>(d_reg, p.dout, b, y0, x0, p.H, p.W);
    is.template load<false>(i_reg, p.in, b, y0, x0, p.H, p.W);
  } else {                                              // bf16 operands in HBM (mixed-precision path)
    if (p.dout_bf16) ds.template load<true>(d_reg, p.dout, b, y0, x0, p.H, p.W); else ds.template load<false>(d_reg, p.dout, b, y0, x0, p.H, p.W);
    if (p.in_bf16) is.template load<true>(i_reg, p.in, b, y0, x0, p.H, p.W); else is.template load<false>(i_reg, p.in, b, y0, x0, p.H, p.W);
  }
}

constexpr int WG_TH = 8;   // pixel tile 8 x 16
// Diagnostic build variants of the Winograd weight gradient (tools/build_variant.sh ... "-DWG_ABL=n"):
// bit 0 no transforms, bit 1 one staged tile reused (no global loads / barriers), bit 2 no MFMAs.
#ifndef WG_ABL
#define WG_ABL 0
#endif

template <int KS, int RT, int CT, int WR, int WC>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgradDev p) {
  constexpr int HALO = KS / 2;
  constexpr int IW = 16 + 2 * HALO;
  constexpr int IH = WG_TH + 2 * HALO;
  constexpr int NPIX_IN = IH * IW;
  constexpr int NPIX = WG_TH * 16;
  constexpr int TAPS = KS * KS;
  constexpr int BNW = RT * 16, BCW = CT * 16;
  constexpr int SD = BNW + 16;     // LDS row strides (floats): == 16 mod 32 -> the 4 k-lanes of a
  constexpr int SI = BCW + 16;     // ds_read_b32 hit disjoint bank groups
  constexpr int GR = RT / WR, GC = CT / WC;      // wave grid over the block's (row-tile, col-tile) plane
  constexpr int D_F4 = (NPIX * BNW / 4 + 255) / 256;
  constexpr int I_F4 = (NPIX_IN * BCW / 4 + 255) / 256;
  static_assert(GR * GC == 4 && GR * WR == RT && GC * WC == CT, "tiles must split over 4 waves");

  __shared__ __attribute__((aligned(16))) float d_lds[NPIX * SD];
  __shared__ __attribute__((aligned(16))) float i_lds[NPIX_IN * SI];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, kq = lane >> 4;
  const int split = blockIdx.x;
  const int n0 = blockIdx.y * BNW, c0 = blockIdx.z * BCW;

  const int rt0 = (wave % GR) * WR, ct0 = (wave / GR) * WC;    // first row / col tile of this wave

  f32x4 acc[WR][WC][TAPS];
#pragma unroll
  for (int a = 0; a < WR; ++a)
#pragma unroll
    for (int c = 0; c < WC; ++c)
#pragma unroll
      for (int t = 0; t < TAPS; ++t) acc[a][c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[WR];
#pragma unroll
  for (int a = 0; a < WR; ++a) bsum[a] = 0.f;

  const int t_begin = split * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.ntiles);

  WgStage<D_F4, 256, BNW / 4, 16, NPIX, 0> d_st;
  WgStage<I_F4, 256, BCW / 4, IW, NPIX_IN, HALO> i_st;
  d_st.init(tid, p.W, p.dout_stride, 0, p.dout_bf16, n0, p.N);
  i_st.init(tid, p.W, p.in_stride, 0, p.in_bf16, c0, p.Cin);
  f32x4 d_reg[D_F4], i_reg[I_F4];
  auto load_tile = [&](int tile) {
    int tt = tile;
    const int tx = tt % p.tiles_x; tt /= p.tiles_x;
    const int ty = tt % p.tiles_y;
    const int b = tt / p.tiles_y;
    wg_load_tile(p, d_st, i_st, d_reg, i_reg, b, ty * WG_TH, tx * 16);
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int r = 0; r < D_F4; ++r) {
      const int f = tid + 256 * r;
      const int pix = f / (BNW / 4), n4 = f % (BNW / 4);
      if (pix < NPIX) *reinterpret_cast<f32x4*>(d_lds + pix * SD + n4 * 4) = d_reg[r];
    }
#pragma unroll
    for (int r = 0; r < I_F4; ++r) {
      const int f = tid + 256 * r;
      const int pix = f / (BCW / 4), c4 = f % (BCW / 4);
      if (pix < NPIX_IN) *reinterpret_cast<f32x4*>(i_lds + pix * SI + c4 * 4) = i_reg[r];
    }
  };

  if (t_begin < t_end) load_tile(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();            // previous tile's reads are done
    store_tile();
    __syncthreads();
    if (tile + 1 < t_end) load_tile(tile + 1);   // in flight under the MFMAs below

    for (int ks = 0; ks < NPIX / 4; ++ks) {
      const int prow = ks >> 2, pcol = (ks & 3) * 4 + kq;     // this lane's pixel of the k-step
      const float* drow = d_lds + (prow * 16 + pcol) * SD + li;
      const float* irow = i_lds + (prow * IW + pcol) * SI + li;
      float afr[WR];
#pragma unroll
      for (int a = 0; a < WR; ++a) {
        afr[a] = drow[(rt0 + a) * 16];
        bsum[a] += afr[a];
      }
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int dy = t / KS, dx = t % KS;
#pragma unroll
        for (int c = 0; c < WC; ++c) {
          const float bfr = irow[(dy * IW + dx) * SI + (ct0 + c) * 16];   // shared by the wave's WR row tiles
#pragma unroll
          for (int a = 0; a < WR; ++a)
            acc[a][c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[a], bfr, acc[a][c][t], 0, 0, 0);
        }
      }
    }
  }

  // ---- write the slab: D[row = n (4*kq + r)][col = c (li)] --------------------------------------
#pragma unroll
  for (int a = 0; a < WR; ++a) {
#pragma unroll
    for (int c = 0; c < WC; ++c)
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + (rt0 + a) * 16 + 4 * kq + r;
          const int cc = c0 + (ct0 + c) * 16 + li;
          if (n < p.Nr && cc < p.Cc)
            p.partial[(((size_t)split * TAPS + t) * p.Nr + n) * p.Cc + cc] = acc[a][c][t][r];
        }
    // bias partial: lanes (li, kq) hold the sum over their pixels of dout[.][n = rt*16 + li]
    float bs = bsum[a];
    bs += __shfl_xor(bs, 16);
    bs += __shfl_xor(bs, 32);
    if (blockIdx.z == 0 && ct0 == 0 && kq == 0) {
      const int n = n0 + (rt0 + a) * 16 + li;
      if (n < p.Nr) p.bpartial[(size_t)split * p.Nr + n] = bs;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// 32x32x2 variant (v_mfma_f32_32x32x2_f32 sustains ~155 TFLOP/s, the 16x16x4 shape ~125) for N % 32 == 0.
// A block owns ONE 32(n) x 32(c) x all-taps output tile and the four waves split the PIXELS of every staged 8x16
// pixel tile (wave w takes image rows 2w, 2w+1), so the number of distinct output tiles is large, the split-K
// factor S small and the slab traffic about half of the 16x16x4 kernel's; the four per-wave accumulators are
// summed through LDS in a fixed order before the slab is written.
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KS>
__device__ __forceinline__ void wgrad32_body(const WgradDev& p, const int split, const int n0, const int c0, const int zblock) {
  constexpr int HALO = KS / 2;
  constexpr int IW = 16 + 2 * HALO;
  constexpr int IH = WG_TH + 2 * HALO;
  constexpr int NPIX_IN = IH * IW;
  constexpr int NPIX = WG_TH * 16;
  constexpr int TAPS = KS * KS;
  constexpr int SP = 36;                               // LDS row stride (floats)
  constexpr int D_F4 = (NPIX * 8 + 255) / 256;         // 32 columns = 8 float4 per pixel
  constexpr int I_F4 = (NPIX_IN * 8 + 255) / 256;
  constexpr int STAGE = (NPIX + NPIX_IN) * SP;
  constexpr int REDF = 4 * 16 * 64;                    // one tap of all four waves
  __shared__ __attribute__((aligned(16))) float lds[STAGE > REDF ? STAGE : REDF];
  float* const d_lds = lds;
  float* const i_lds = lds + NPIX * SP;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, kh = lane >> 5;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
  float bsum = 0.f;

  const int t_begin = split * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.ntiles);

  WgStage<D_F4, 256, 8, 16, NPIX, 0> d_st;
  WgStage<I_F4, 256, 8, IW, NPIX_IN, HALO> i_st;
  d_st.init(tid, p.W, p.dout_stride, 0, p.dout_bf16, n0, p.N);
  i_st.init(tid, p.W, p.in_stride, 0, p.in_bf16, c0, p.Cin);
  f32x4 d_reg[D_F4], i_reg[I_F4];
  auto load_tile = [&](int tile) {
    int tt = tile;
    const int tx = tt % p.tiles_x; tt /= p.tiles_x;
    const int ty = tt % p.tiles_y;
    const int b = tt / p.tiles_y;
    wg_load_tile(p, d_st, i_st, d_reg, i_reg, b, ty * WG_TH, tx * 16);
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int r = 0; r < D_F4; ++r) {
      const int f = tid + 256 * r;
      const int pix = f >> 3, n4 = f & 7;
      if (pix < NPIX) *reinterpret_cast<f32x4*>(d_lds + pix * SP + n4 * 4) = d_reg[r];
    }
#pragma unroll
    for (int r = 0; r < I_F4; ++r) {
      const int f = tid + 256 * r;
      const int pix = f >> 3, c4 = f & 7;
      if (pix < NPIX_IN) *reinterpret_cast<f32x4*>(i_lds + pix * SP + c4 * 4) = i_reg[r];
    }
  };

  if (t_begin < t_end) load_tile(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (tile + 1 < t_end) load_tile(tile + 1);
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {
      const int prow = 2 * wave + (ks >> 3), pcol = (ks & 7) * 2 + kh;   // this lane's pixel of the k-step
      const float a = d_lds[(prow * 16 + pcol) * SP + li];
      bsum += a;
      const float* irow = i_lds + (prow * IW + pcol) * SP + li;
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int dy = t / KS, dx = t % KS;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, irow[(dy * IW + dx) * SP], acc[t], 0, 0, 0);
      }
    }
  }

  // ---- in-block reduction over the four waves (fixed order), one tap at a time, then the slab write ----
  // lane holds D[row n = (q&3) + 8*(q>>2) + 4*kh][col c = li] in register q
  __syncthreads();
#pragma unroll                       // must stay fully unrolled: a runtime index would push acc[] to scratch
  for (int t = 0; t < TAPS; ++t) {
#pragma unroll
    for (int q = 0; q < 16; ++q) lds[(wave * 16 + q) * 64 + lane] = acc[t][q];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = wave * 4 + j;                       // this wave finishes registers 4w .. 4w+3
      const float v = (lds[(0 * 16 + q) * 64 + lane] + lds[(1 * 16 + q) * 64 + lane]) +
                      (lds[(2 * 16 + q) * 64 + lane] + lds[(3 * 16 + q) * 64 + lane]);
      const int n = n0 + (q & 3) + 8 * (q >> 2) + 4 * kh;
      const int c = c0 + li;
      if (n < p.Nr && c < p.Cc) p.partial[(((size_t)split * TAPS + t) * p.Nr + n) * p.Cc + c] = v;
    }
    __syncthreads();
  }
  // bias partial: lane (li, kh) holds the sum over its pixels of dout[.][n0 + li]
  bsum += __shfl_xor(bsum, 32);
  if (lane < 32) lds[wave * 32 + lane] = bsum;
  __syncthreads();
  if (zblock == 0 && tid < 32) {
    const int n = n0 + tid;
    if (n < p.Nr) p.bpartial[(size_t)split * p.Nr + n] = (lds[tid] + lds[32 + tid]) + (lds[64 + tid] + lds[96 + tid]);
  }
}

template <int KS>
__global__ __launch_bounds__(256, 2) void wgrad32_kernel(WgradDev p) {
  wgrad32_body<KS>(p, blockIdx.x, blockIdx.y * 32, blockIdx.z * 32, blockIdx.z);
}


// ------------------------------------------------------------------------------------------------
// Winograd weight gradient for 3x3 convs:   dU[pos][n][c] = sum_tiles W_pos[tile][n] * V_pos[tile][c]
//   W = A dY A^T (4x4 from the 2x2 output-gradient patch of a Winograd tile),  V = B^T d B (4x4 input patch),
//   dg = G^T dU G is applied per lane before the slab write.  16 positions x (tiles x N x C) instead of 9 taps x (pixels x N x C):
//   2.25x fewer MFMA FLOPs.  Both transforms are done per lane in registers (lane = (n or c, tile)); MFMA rows = n,
//   cols = c, k = 4 Winograd tiles.  Block = 64 n x 32 c, waves 2 x 2, wave tile 32 n x 16 c x 16 positions
//   (128 accumulator VGPRs); pixel tiles of TH x 16 (4 x 16 in the shipped kernels: 16 Winograd tiles = 4 k-steps, walked two
//   k-steps -- one Winograd tile row -- at a time, see the loop) are walked split-K style.
// ------------------------------------------------------------------------------------------------
// WIDE_C: block = 32 n x 64 c (waves 1 x 4) instead of 64 n x 32 c (waves 2 x 2), for problems with at most 32 output
// channels (the four growth convs of an IRN DenseBlock, N = 32: in a 64-row block the two waves of the upper row half have
// no live row tile at all and only transform inputs for nothing -- half the block's matrix-pipe slots)
// (UNR: the k-loop's unroll factor until round 3; the loop now walks k-step pairs without unrolling.  The parameter is kept so
// that kernel names stay comparable with the committed profiles.)
template <int TH, int UNR, int KH, bool WIDE_C = false>
__device__ __forceinline__ void wgrad_wino_body(const WgradDev& p, const int split, const int n0, const int c0, const int zblock) {
  // KH = 2: eight waves; the two wave quads take alternate halves of every pixel tile's k-steps and are summed through
  // LDS at the end -> one slab per CU instead of two (half the split-K slab traffic), one staged tile per 8 waves.
  constexpr int NTHR = 256 * KH;
  constexpr int IW = 18, IH = TH + 2, NPIX_IN = IH * IW, NPIX = TH * 16;
  constexpr int BNW = WIDE_C ? 32 : 64, BCW = WIDE_C ? 64 : 32;
  // the four k-lanes of a wave read pixels 4 kq (+ 2 e): 4*SD == 4*SI == 16 (mod 32), so k-lanes 0 / 2 and 1 / 3 fall on the two
  // halves of the banks (two passes for 64 lanes, the minimum); rows stay 16-byte aligned for the float4 staging stores
  constexpr int SD = BNW + 4, SI = BCW + 4;
  static_assert((4 * SD) % 32 == 16 && (4 * SI) % 32 == 16, "LDS pitch against the k-lane pixel stride");
  static_assert(!(WIDE_C && KH != 1), "the 32 x 64 block shape exists for four-wave blocks only");
  constexpr int D_F4 = (NPIX * BNW / 4 + NTHR - 1) / NTHR;
  constexpr int I_F4 = (NPIX_IN * BCW / 4 + NTHR - 1) / NTHR;
  constexpr int KSTEPS = TH / KH;                      // k-steps of a tile per wave
  constexpr int LDS_TILE = NPIX * SD + NPIX_IN * SI;
  constexpr int LDS_RED = (KH == 2) ? 16 * 256 * 4 : 0;   // 8 positions x 2 row tiles x 256 lanes x float4
  constexpr int LDS_FLOATS = LDS_TILE > LDS_RED ? LDS_TILE : LDS_RED;
  static_assert(TH % (2 * KH) == 0, "a wave walks whole Winograd tile rows (two k-steps each)");
  __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];
  float* const d_lds = smem;
  float* const i_lds = smem + NPIX * SD;

  const int tid = threadIdx.x;
  const int wave = (tid >> 6) & 3, kh = tid >> 8, lane = tid & 63;
  const int li = lane & 15, kq = lane >> 4;
  const int wr = WIDE_C ? 0 : (wave & 1), wc = WIDE_C ? wave : (wave >> 1);

  f32x4 acc[16][2];
#pragma unroll
  for (int q = 0; q < 16; ++q) { acc[q][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[q][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  float bsum[2] = {0.f, 0.f};
  const bool a1_live = (n0 + 32 * wr + 16) < p.N;       // wave-uniform

  const int t_begin = split * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.ntiles);

  WgStage<D_F4, NTHR, BNW / 4, 16, NPIX, 0> d_st;
  WgStage<I_F4, NTHR, BCW / 4, IW, NPIX_IN, 1> i_st;
  d_st.init(tid, p.W, p.dout_stride, p.dout_gs, p.dout_bf16, n0, p.N);
  i_st.init(tid, p.W, p.in_stride, p.in_gs, p.in_bf16, c0, p.Cin);
  f32x4 d_reg[D_F4], i_reg[I_F4];
  auto load_tile = [&](int tile) {
    int tt = tile;
    const int tx = tt % p.tiles_x; tt /= p.tiles_x;
    const int ty = tt % p.tiles_y;
    const int b = tt / p.tiles_y;
    wg_load_tile(p, d_st, i_st, d_reg, i_reg, b, ty * TH, tx * 16);
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int r = 0; r < D_F4; ++r) {
      const int f = tid + NTHR * r;
      const int pix = f / (BNW / 4), n4 = f % (BNW / 4);
      if (pix < NPIX) *reinterpret_cast<f32x4*>(d_lds + pix * SD + n4 * 4) = d_reg[r];
    }
#pragma unroll
    for (int r = 0; r < I_F4; ++r) {
      const int f = tid + NTHR * r;
      const int pix = f / (BCW / 4), c4 = f % (BCW / 4);
      if (pix < NPIX_IN) *reinterpret_cast<f32x4*>(i_lds + pix * SI + c4 * 4) = i_reg[r];
    }
  };

  if (t_begin < t_end) load_tile(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
#if (WG_ABL & 2)
    if (tile == t_begin) { __syncthreads(); store_tile(); __syncthreads(); }
#else
    __syncthreads();
    store_tile();
    __syncthreads();
    if (tile + 1 < t_end) load_tile(tile + 1);
#endif
    // Vector-ALU instructions do NOT hide behind the matrix pipe on this chip (tools/mfma_valu.hip: with two waves per SIMD every
    // VALU instruction per v_mfma_f32_16x16x4_f32 adds ~2.5 clocks to its 32, every LDS read ~4.6), so the transforms are written
    // for the fewest instructions.  The k-steps of a tile are walked two at a time -- Winograd tile row `trow`, this lane's tiles
    // (trow, 2 kq) [even k-step] and (trow, 2 kq + 1) [odd k-step]:
    //   * V = B^T d B: the 4x4 patches of the two x-adjacent tiles are columns 0-3 and 2-5 of one 4x6 patch, so the row pass
    //     (24 values read, 24 adds) is shared; only the column pass runs per tile;
    //   * row 3 and column 3 of BOTH transforms are computed negated (d3 - d1 for d1 - d3, +y1 for -y1): free in V, saves every
    //     negation in W, and each product w * v is unchanged;
    //   * the bias-gradient partial of a tile is W's (1, 1) entry, (y00 + y10) + (y01 + y11).
#pragma unroll 1
    for (int mm = 0; mm < KSTEPS / 2; ++mm) {
      const int trow = kh * (KSTEPS / 2) + mm;
      float v[2][16];
      {
        const float* ip = i_lds + ((2 * trow) * IW + 4 * kq) * SI + 16 * wc + li;
#if (WG_ABL & 1)
        const float v0 = ip[0], v1 = ip[SI];
#pragma unroll
        for (int c = 0; c < 16; ++c) { v[0][c] = (c & 1) ? v1 : v0; v[1][c] = (c & 1) ? v0 : v1; }
#else
        float t[4][6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const float d0 = ip[(0 * IW + c) * SI], d1 = ip[(1 * IW + c) * SI], d2 = ip[(2 * IW + c) * SI], d3 = ip[(3 * IW + c) * SI];
          t[0][c] = d0 - d2; t[1][c] = d1 + d2; t[2][c] = d2 - d1; t[3][c] = d3 - d1;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            v[e][a * 4 + 0] = t[a][2 * e + 0] - t[a][2 * e + 2];
            v[e][a * 4 + 1] = t[a][2 * e + 1] + t[a][2 * e + 2];
            v[e][a * 4 + 2] = t[a][2 * e + 2] - t[a][2 * e + 1];
            v[e][a * 4 + 3] = t[a][2 * e + 3] - t[a][2 * e + 1];
          }
#endif
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          if (a == 1 && !a1_live) continue;     // row tile entirely beyond N (N = 48 in the 64-row block): no transform, no MFMAs
          const float* dp = d_lds + ((2 * trow) * 16 + 4 * kq + 2 * e) * SD + 32 * wr + 16 * a + li;
          const float y00 = dp[0], y01 = dp[SD], y10 = dp[16 * SD], y11 = dp[17 * SD];
          const float r10 = y00 + y10, r11 = y01 + y11, r20 = y00 - y10, r21 = y01 - y11;
          float w[16];
          w[0] = y00; w[1] = y00 + y01; w[2] = y00 - y01; w[3] = y01;
          w[4] = r10; w[5] = r10 + r11; w[6] = r10 - r11; w[7] = r11;
          w[8] = r20; w[9] = r20 + r21; w[10] = r20 - r21; w[11] = r21;
          w[12] = y10; w[13] = y10 + y11; w[14] = y10 - y11; w[15] = y11;
          bsum[a] += w[5];
#if (WG_ABL & 1)
#pragma unroll
          for (int q = 0; q < 16; ++q) w[q] = (q & 1) ? y01 : y00;
#endif
#if (WG_ABL & 4)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[q][a][0] += w[q] * v[e][q];
#else
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[q][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q], v[e][q], acc[q][a], 0, 0, 0);
#endif
        }
      }
    }
  }

  // ---- KH = 2: add the second wave quad's accumulators (and bias sums) to the first through LDS, 8 positions at a time
  if constexpr (KH == 2) {
    f32x4* const red = reinterpret_cast<f32x4*>(smem);
    const int slot = wave * 64 + lane;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();                                 // tile buffers (first round) / previous round consumed
      if (kh == 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int a = 0; a < 2; ++a) red[(q * 2 + a) * 256 + slot] = acc[half * 8 + q][a];
      }
      __syncthreads();
      if (kh == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int a = 0; a < 2; ++a) acc[half * 8 + q][a] += red[(q * 2 + a) * 256 + slot];
      }
    }
    __syncthreads();
    float* const bred = smem;
    if (kh == 1) { bred[slot * 2 + 0] = bsum[0]; bred[slot * 2 + 1] = bsum[1]; }
    __syncthreads();
    if (kh == 1) return;
    bsum[0] += bred[slot * 2 + 0];
    bsum[1] += bred[slot * 2 + 1];
  }

  // ---- dg = G^T dU G per lane (all 16 positions of an (n, c) pair live in one lane), then the 9-tap slab write:
  // D[row n = 4*kq + r][col c = li]; 9 instead of 16 values per pair cuts the slab traffic by 44 % and the ordered
  // reduction is the plain tap-major one
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x[3][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float u0 = acc[q][a][r], u1 = acc[4 + q][a][r], u2 = acc[8 + q][a][r], u3 = acc[12 + q][a][r];
        x[0][q] = u0 + 0.5f * (u1 + u2);
        x[1][q] = 0.5f * (u1 - u2);
        x[2][q] = 0.5f * (u1 + u2) + u3;
      }
      const int n = n0 + 32 * wr + 16 * a + 4 * kq + r;
      const int c = c0 + 16 * wc + li;
      if (n < p.Nr && c < p.Cc) {
        float* dst = p.partial + ((size_t)split * 9 * p.Nr + n) * p.Cc + c;
        const size_t tap_stride = (size_t)p.Nr * p.Cc;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          dst[(i * 3 + 0) * tap_stride] = x[i][0] + 0.5f * (x[i][1] + x[i][2]);
          dst[(i * 3 + 1) * tap_stride] = 0.5f * (x[i][1] - x[i][2]);
          dst[(i * 3 + 2) * tap_stride] = 0.5f * (x[i][1] + x[i][2]) + x[i][3];
        }
      }
    }
    float bs = bsum[a];
    bs += __shfl_xor(bs, 16);
    bs += __shfl_xor(bs, 32);
    if (zblock == 0 && wc == 0 && kq == 0) {
      const int n = n0 + 32 * wr + 16 * a + li;
      if (n < p.Nr) p.bpartial[(size_t)split * p.Nr + n] = bs;
    }
  }
}

template <int TH, int UNR, int KH>
__global__ __launch_bounds__(256 * KH, 2 / KH) void wgrad_wino_kernel(WgradDev p) {
  wgrad_wino_body<TH, UNR, KH>(p, blockIdx.x, blockIdx.y * 64, blockIdx.z * 32, blockIdx.z);
}

// Reduce S slabs in a fixed order and accumulate into the OIHW gradient:  gw[n][c][tap] += sum_s partial[s][tap][n][c]
// A thread owns 4 consecutive c of one (tap, n) row (16-byte loads); the S slabs are split over the block's 4 waves,
// 4 loads in flight each; the 4 partial sums are combined through LDS in a fixed order -> bitwise reproducible.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bpartial,
                                    int S, int taps, int Nr, int Cc, int N, int Cin,
                                    float* __restrict__ gw, float* __restrict__ gb) {
  __shared__ f32x4 red[4][64];
  const int c4n = Cc >> 2;
  const int total4 = taps * Nr * c4n;                 // float4 columns of one slab
  const int wblocks = (total4 + 63) / 64;
  const int grp = threadIdx.x >> 6;
  if ((int)blockIdx.x >= wblocks) {
    // ---- bias: db[n] += sum_s bpartial[s][n]; same 4-way slab split and fixed-order combine ----
    if (gb == nullptr) return;
    const int i = ((int)blockIdx.x - wblocks) * 64 + (threadIdx.x & 63);
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    if (i < N) {
      int k = grp;
      for (; k + 12 < S; k += 16) {
        b0 += bpartial[(size_t)k * Nr + i];
        b1 += bpartial[(size_t)(k + 4) * Nr + i];
        b2 += bpartial[(size_t)(k + 8) * Nr + i];
        b3 += bpartial[(size_t)(k + 12) * Nr + i];
      }
      for (; k < S; k += 4) b0 += bpartial[(size_t)k * Nr + i];
    }
    red[grp][threadIdx.x & 63][0] = (b0 + b1) + (b2 + b3);
    __syncthreads();
    if (grp == 0 && i < N) {
      const int l = threadIdx.x;
      gb[i] += (red[0][l][0] + red[1][l][0]) + (red[2][l][0] + red[3][l][0]);
    }
    return;
  }
  const int o = blockIdx.x * 64 + (threadIdx.x & 63);
  const bool live = o < total4;
  const size_t stride4 = (size_t)total4;
  const f32x4* p4 = reinterpret_cast<const f32x4*>(partial);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  if (live) {
    int k = grp;
    for (; k + 12 < S; k += 16) {
      s0 += p4[o + (size_t)k * stride4];
      s1 += p4[o + (size_t)(k + 4) * stride4];
      s2 += p4[o + (size_t)(k + 8) * stride4];
      s3 += p4[o + (size_t)(k + 12) * stride4];
    }
    for (; k < S; k += 4) s0 += p4[o + (size_t)k * stride4];
  }
  red[grp][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && live) {
    const int l = threadIdx.x;
    const f32x4 tot = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
    const int c0 = (o % c4n) * 4, n = (o / c4n) % Nr, t = o / (c4n * Nr);
    if (n < N) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c0 + j < Cin) gw[((size_t)n * Cin + c0 + j) * taps + t] += tot[j];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Grouped weight gradients: the four convs of a GLOW block (two subnets x two convs) in ONE launch.
// A single conv offers only N*C / (64*32) = 4 .. 24 output tiles, so filling 512 blocks took 21 .. 128 pixel splits
// per conv, i.e. ~38 MB of split-K slabs written and re-read per conv (1.3 GB per training step) and a reduce launch
// each.  Grouped, the block's 24 (level 0) / 72 (level 1) tiles share the grid: 7 .. 21 splits, slabs 3-6x smaller,
// two launches (gradient + ordered reduce) per block instead of eight.  Results stay bitwise reproducible: every slab
// is summed in split order and each gradient element is touched by exactly one thread.
// ------------------------------------------------------------------------------------------------
constexpr int WG_MAXP = 8;
struct WgradProb {
  WgradDev d;
  float* gw; float* gb;
  int gap_begin, gap_len;  // operand channels [gap_begin, gap_begin + gap_len) are padding with no counterpart in the weight:
                           // they are skipped and later channels shift down (IRN DenseBlock feature buffer, cin padded to 8)
  int S, nblk, cblk;
  int wide_c;            // Winograd kernel: this problem's blocks are 32 n x 64 c (N <= 32) instead of 64 n x 32 c
  int narrow_c;          // bf16 kernel: this problem's blocks are 128 n x 32 c (Cin <= 32) instead of 64 n x 64 c
  int block_begin;       // first block of this problem in the grouped gradient grid
  int red_begin;         // first block of this problem in the grouped reduce grid
};
struct WgradGroup { WgradProb p[WG_MAXP]; int n; int taps; };

__device__ __forceinline__ int group_problem(const WgradGroup& g, int bid, bool reduce) {
  int pi = 0;
#pragma unroll
  for (int i = 1; i < WG_MAXP; ++i)
    if (i < g.n && bid >= (reduce ? g.p[i].red_begin : g.p[i].block_begin)) pi = i;
  return pi;
}

template <int TH, int UNR>
__global__ __launch_bounds__(256, 2) void wgrad_wino_group_kernel(WgradGroup g) {
  const int pi = group_problem(g, blockIdx.x, false);
  const WgradProb& q = g.p[pi];
  const int lb = blockIdx.x - q.block_begin;
  const int split = lb % q.S, nb = (lb / q.S) % q.nblk, cb = lb / (q.S * q.nblk);
  if (q.wide_c) wgrad_wino_body<TH, UNR, 1, true>(q.d, split, nb * 32, cb * 64, cb);
  else wgrad_wino_body<TH, UNR, 1>(q.d, split, nb * 64, cb * 32, cb);
}

template <int KS>
__global__ __launch_bounds__(256, 2) void wgrad32_group_kernel(WgradGroup g) {
  const int pi = group_problem(g, blockIdx.x, false);
  const WgradProb& q = g.p[pi];
  const int lb = blockIdx.x - q.block_begin;
  const int split = lb % q.S, nb = (lb / q.S) % q.nblk, cb = lb / (q.S * q.nblk);
  wgrad32_body<KS>(q.d, split, nb * 32, cb * 32, cb);
}

// Ordered slab reduce of a whole group:  gw[n][c][tap] += sum_s partial[s][tap][n][c]  (+ the bias sums).
// Block = RC float4 columns (4 consecutive c of one n) x taps: thread (tap, column) sums its S slab entries in split
// order with coalesced 16-byte loads, the block then transposes through LDS so that the OIHW gradient (taps innermost)
// is updated with contiguous runs instead of a 4-byte scatter at a 36-byte stride.
template <int TAPS>
__global__ __launch_bounds__(256) void wgrad_reduce_group_kernel(WgradGroup g) {
  constexpr int RC = 256 / TAPS;                     // columns per block (28 for 3x3, 256 for 1x1)
  __shared__ float tile[RC * 4 * TAPS];
  const int pi = group_problem(g, blockIdx.x, true);
  const WgradProb& q = g.p[pi];
  const WgradDev& d = q.d;
  const int c4n = d.Cc >> 2;
  const int ncol = d.Nr * c4n;                       // float4 columns of one tap plane
  const int wblocks = (ncol + RC - 1) / RC;
  const int lb = blockIdx.x - q.red_begin;
  const int tid = threadIdx.x;
  if (lb >= wblocks) {                               // ---- bias rows: db[n] += sum_s bpartial[s][n] ----
    const int n = (lb - wblocks) * 256 + tid;
    if (q.gb != nullptr && n < d.N) {
      float acc = 0.f;
      for (int k = 0; k < q.S; ++k) acc += d.bpartial[(size_t)k * d.Nr + n];
      q.gb[n] += acc;
    }
    return;
  }
  const int t = tid / RC, cl = tid - t * RC;         // tap, block-local column
  const int col = lb * RC + cl;
  if (t < TAPS && col < ncol) {
    const f32x4* src = reinterpret_cast<const f32x4*>(d.partial) + (size_t)t * ncol + col;
    const size_t slab = (size_t)TAPS * ncol;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    int k = 0;
#pragma unroll 4                                     // eight slab loads in flight; the association below is unchanged
    for (; k + 1 < q.S; k += 2) { a0 += src[(size_t)k * slab]; a1 += src[(size_t)(k + 1) * slab]; }
    if (k < q.S) a0 += src[(size_t)k * slab];
    a0 += a1;                                        // fixed association: (even splits) + (odd splits)
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[(cl * 4 + j) * TAPS + t] = a0[j];
  }
  __syncthreads();
  // element e of the block = (column cl, channel j, tap t), consecutive in the OIHW gradient for consecutive e
  for (int e = tid; e < RC * 4 * TAPS; e += 256) {
    const int cl2 = e / (4 * TAPS), r = e - cl2 * 4 * TAPS;
    const int j = r / TAPS, t2 = r - j * TAPS;
    const int col2 = lb * RC + cl2;
    if (col2 < ncol) {
      const int n = col2 / c4n, c = (col2 - n * c4n) * 4 + j;
      if (n < d.N && c < d.Cin && !(c >= q.gap_begin && c < q.gap_begin + q.gap_len)) {
        const int cw = c < q.gap_begin ? c : c - q.gap_len;
        q.gw[((size_t)n * (d.Cin - q.gap_len) + cw) * TAPS + t2] += tile[e];
      }
    }
  }
}

struct WgradPlan { int RT, CT, Nr, Cc, nblk, cblk, S, tiles_per_split, ntiles, tiles_x, tiles_y; size_t bytes; bool use32; bool wino; int th; int kh; };

void wgrad_set_bf16_mfma(int on);
static bool g_wgrad_force16 = false;   // test hook
static bool g_wgrad_wino = true;       // Winograd weight gradient for 3x3 (test hook bit 1 disables)
static bool g_wgrad_wino_th8 = false;  // test hook bit 2: 8-row pixel tiles in the Winograd weight gradient
static bool g_wgrad_wino_kh2 = false;  // test hook bit 3: 8-wave blocks with an in-block k split (half the slabs; same kernel
                                       // time, but 3 % slower end to end when other streams' kernels co-run) -- off
void wgrad_set_wide_c(int on);
static int g_wgrad_group_mode = 1;     // 1: whole block (4 convs), 2: per half-coupling (2 convs); test hook bit 5 selects 2
static bool g_wgrad_grouped = true;    // test hook bit 4 clears: the block executor issues one launch pair per conv (round-1 path)
void wgrad_set_force16(int on) { g_wgrad_force16 = (on & 1) != 0; g_wgrad_wino = (on & 2) == 0; g_wgrad_wino_th8 = (on & 4) != 0; g_wgrad_wino_kh2 = (on & 8) != 0; g_wgrad_grouped = (on & 16) == 0; g_wgrad_group_mode = (on & 32) ? 2 : 1; wgrad_set_bf16_mfma((on & 64) == 0); wgrad_set_wide_c((on & 128) == 0); }
bool wgrad_grouping_enabled() { return g_wgrad_grouped && g_wgrad_wino && !g_wgrad_force16; }
int wgrad_group_mode() { return g_wgrad_group_mode; }

static WgradPlan make_plan(int N, int Cin, int ksize, int B, int H, int W) {
  WgradPlan pl;
  pl.wino = (ksize == 3) && g_wgrad_wino;
  pl.use32 = (N % 32 == 0) && !g_wgrad_force16 && !pl.wino;
  // row tiles: 3 when N is a multiple of 48 but not of 64 (N=48), else 4; col tiles 4 with RT=3, else 2 or 4
  if (N % 64 != 0 && N % 48 == 0) { pl.RT = 3; pl.CT = 4; }
  else { pl.RT = 4; pl.CT = (Cin >= 64 && ksize == 1) ? 4 : 2; }
  if (pl.use32) { pl.RT = 2; pl.CT = 2; }              // 32 x 32 output tile per block
  if (pl.wino) { pl.RT = 4; pl.CT = 2; }               // 64 x 32 output tile, 16 transform positions
  const int bnw = pl.RT * 16, bcw = pl.CT * 16;
  pl.nblk = (N + bnw - 1) / bnw;
  pl.cblk = (Cin + bcw - 1) / bcw;
  pl.Nr = pl.nblk * bnw;
  pl.Cc = pl.cblk * bcw;
  pl.tiles_x = (W + 15) / 16;
  pl.kh = (pl.wino && g_wgrad_wino_kh2) ? 2 : 1;      // 8-wave blocks, in-block k split
  pl.th = (pl.wino && !g_wgrad_wino_th8 && pl.kh == 1) ? 4 : WG_TH;
  pl.tiles_y = (H + pl.th - 1) / pl.th;
  pl.ntiles = B * pl.tiles_x * pl.tiles_y;
  // the RT=3 tile (N=48) needs ~90 KB of LDS and >256 registers: one block per CU -> aim for one round of 256
  // blocks (half the slab traffic of 512); the other tiles run two blocks per CU
  int S = (((pl.RT == 3 && !pl.wino) || pl.kh == 2) ? 256 : 512) / (pl.nblk * pl.cblk);
  if (S < 1) S = 1;
  if (S > pl.ntiles) S = pl.ntiles;
  pl.tiles_per_split = (pl.ntiles + S - 1) / S;
  pl.S = (pl.ntiles + pl.tiles_per_split - 1) / pl.tiles_per_split;
  const int taps = ksize * ksize;
  pl.bytes = ((size_t)pl.S * taps * pl.Nr * pl.Cc + (size_t)pl.S * pl.Nr) * sizeof(float);
  return pl;
}

size_t wgrad_workspace_bytes(int N, int Cin, int ksize, int B, int H, int W) {
  return make_plan(N, Cin, ksize, B, H, W).bytes;
}

template <int KS>
static void launch_wgrad(const WgradPlan& pl, const WgradDev& d, hipStream_t st) {
  dim3 grid(pl.S, pl.nblk, pl.cblk);
  if (pl.use32) hipLaunchKernelGGL((wgrad32_kernel<KS>), grid, dim3(256), 0, st, d);
  else if (pl.RT == 3) hipLaunchKernelGGL((wgrad_mfma_kernel<KS, 3, 4, 3, 1>), grid, dim3(256), 0, st, d);
  else if (pl.CT == 4) hipLaunchKernelGGL((wgrad_mfma_kernel<KS, 4, 4, 2, 2>), grid, dim3(256), 0, st, d);
  else hipLaunchKernelGGL((wgrad_mfma_kernel<KS, 4, 2, 2, 1>), grid, dim3(256), 0, st, d);
}

int wgrad_launch(const float* in, int in_stride, int Cin, const float* dout, int dout_stride, int N,
                 int B, int H, int W, int ksize, float* gw, float* gb, void* ws, size_t ws_bytes, hipStream_t st) {
  SININN_CHECK(ksize == 1 || ksize == 3, "wgrad: ksize %d not in {1,3}", ksize);
  SININN_CHECK(in && dout && gw && ws, "wgrad: null pointer");
  SININN_CHECK(Cin > 0 && Cin % 4 == 0 && N > 0 && N % 4 == 0, "wgrad: Cin=%d and N=%d must be multiples of 4", Cin, N);
  SININN_CHECK(in_stride >= Cin && in_stride % 4 == 0 && aligned16(in), "wgrad: in must be 16-byte aligned, stride %% 4 == 0");
  SININN_CHECK(dout_stride >= N && dout_stride % 4 == 0 && aligned16(dout), "wgrad: dout must be 16-byte aligned, stride %% 4 == 0");
  SININN_CHECK(B > 0 && H > 0 && W > 0, "wgrad: bad shape");
  // the staging descriptors (WgStage) hold 32-bit byte offsets from a tile's origin pixel
  SININN_CHECK(H < 65535 && W < 65535 && 64ull * W * (unsigned)in_stride + 4ull * Cin < (1ull << 32) &&
               64ull * W * (unsigned)dout_stride + 4ull * N < (1ull << 32),
               "wgrad: a pixel tile spans more than 4 GB (32-bit staging offsets)");
  const WgradPlan pl = make_plan(N, Cin, ksize, B, H, W);
  SININN_CHECK(ws_bytes >= pl.bytes, "wgrad: workspace too small (%zu < %zu)", ws_bytes, pl.bytes);
  WgradDev d;
  d.in = in; d.in_stride = in_stride; d.Cin = Cin; d.dout = dout; d.dout_stride = dout_stride; d.N = N;
  d.B = B; d.H = H; d.W = W; d.tiles_x = pl.tiles_x; d.tiles_y = pl.tiles_y; d.ntiles = pl.ntiles;
  d.tiles_per_split = pl.tiles_per_split; d.Nr = pl.Nr; d.Cc = pl.Cc;
  d.in_bf16 = 0; d.dout_bf16 = 0; d.in_gs = 0; d.dout_gs = 0;
  const int taps = ksize * ksize;
  d.partial = static_cast<float*>(ws);
  d.bpartial = d.partial + (size_t)pl.S * taps * pl.Nr * pl.Cc;
  if (pl.wino) {
    if (pl.kh == 2) hipLaunchKernelGGL((wgrad_wino_kernel<8, 2, 2>), dim3(pl.S, pl.nblk, pl.cblk), dim3(512), 0, st, d);
    else if (pl.th == 4) hipLaunchKernelGGL((wgrad_wino_kernel<4, 2, 1>), dim3(pl.S, pl.nblk, pl.cblk), dim3(256), 0, st, d);
    else hipLaunchKernelGGL((wgrad_wino_kernel<8, 1, 1>), dim3(pl.S, pl.nblk, pl.cblk), dim3(256), 0, st, d);
    SININN_LAUNCH_CHECK("wgrad_wino");
  } else {
    if (ksize == 3) launch_wgrad<3>(pl, d, st); else launch_wgrad<1>(pl, d, st);
    SININN_LAUNCH_CHECK("wgrad_mfma");
  }
  const int total4 = taps * pl.Nr * (pl.Cc / 4);
  const int bias_blocks = gb ? (N + 63) / 64 : 0;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total4 + 63) / 64 + bias_blocks), dim3(256), 0, st,
                     d.partial, d.bpartial, pl.S, taps, pl.Nr, pl.Cc, N, Cin, gw, gb);
  SININN_LAUNCH_CHECK("wgrad_reduce");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// bf16 weight gradient (mixed-precision path): dW[tap][n][c] = sum_pix dout[pix][n] * in[pix + tap][c] on
// v_mfma_f32_32x32x16_bf16, fp32 accumulation.  MFMA rows = n, columns = c, k = 16 consecutive pixels of an image row.
//   * both operands need k-contiguous fragments (8 consecutive PIXELS of one channel per lane), i.e. the transpose of the
//     pixel-major tensors: the tile is transposed while it is staged -- a lane loads the same channel group of two
//     neighbouring pixels (16-byte global loads), packs (pixel, pixel + 1) pairs and writes one dword per channel into
//     dT[n][pixel] / iT[c][halo row][x]; the 64 lanes of a wave write 64 consecutive pixel pairs of one channel row.
//   * the nine taps of a 3x3 conv share the staged halo tile: for tile row y and tap row dy the lane reads the 16 aligned
//     pixels x = 8h .. 8h + 15 of halo row y + dy (two ds_read_b128) and forms the dx = 0 / 1 / 2 fragments in registers
//     (dx = 2 is a register re-selection, dx = 1 four v_alignbyte): 7 LDS reads feed 9 MFMAs.
//   * block = 64 n x 64 c x all taps, waves 2 x 2, wave tile 32 x 32 x 9 taps = 144 accumulator registers; pixel tiles of
//     8 x 16 are walked split-K style and the fp32 slabs go through the same ordered group reduce as the fp32 kernels.
// ------------------------------------------------------------------------------------------------
typedef __bf16 wgb_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned wgb_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16_pair(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 v = {(__bf16)lo, (__bf16)hi};
  return *reinterpret_cast<unsigned*>(&v);
}

// Staging is split into a load phase (global -> registers, issued for the NEXT pixel tile before the MFMA loop of the
// current one) and a store phase (registers -> transposed LDS tile): the first build loaded and stored in one loop between
// the two barriers, which left every tile's global-load latency exposed (only the CU's second block covered it).
// IN_BF16 / DOUT_BF16 = storage type of the two operands (an fp32 operand is rounded while it is stored).  A 3x3 problem
// holds 144 accumulator registers, so only its bf16 operand (the 256-channel hidden tensor / hidden gradient: the HBM
// stream) is prefetched across the MFMA loop (16-32 VGPRs); the fp32 operand's loads (32-64 VGPRs) are issued after the
// loop, not live across it -- prefetching both spilled 98 VGPRs.  1x1 problems (16 accumulators) prefetch both.
// Diagnostic build variants of the bf16 weight gradient (tools/build_variant.sh ... "-DWGB_ABL=n"; timing only, results are wrong):
// 1 no MFMA loop, 2 no transposing LDS stores, 4 no global loads, 8 no operand shifts / bias sums in the loop, 16 no barriers
#ifndef WGB_ABL
#define WGB_ABL 0
#endif
// staging item f of a thread -> (pixel-pair slot, channel group).  WGB_COALESCE: four consecutive lanes take four consecutive channel
// groups of ONE pixel pair -- 64 contiguous bytes per pixel, whole sectors per wave instruction -- instead of 64 lanes taking one
// 16-byte group of 64 different pixels (a quarter of every sector used; the ablation builds put 40 % of the kernel on its loads)
#ifndef WGB_COALESCE
#define WGB_COALESCE 1
#endif
#if WGB_COALESCE
#define WGB_SLOT(f) (((f) >> 2) & 63)
#define WGB_GROUP(f) ((((f) >> 8) << 2) | ((f) & 3))
#else
#define WGB_SLOT(f) ((f) & 63)
#define WGB_GROUP(f) ((f) >> 6)
#endif
constexpr int WGB_PD = 128 * 2 + 16;                                  // bytes per dout row (channel n): 4-bank step
constexpr int wgb_pi(int ks) { return (((8 + 2 * (ks / 2)) * 24 * 2 + 255) / 256) * 256 + 16; }   // bytes per in row (channel c)

// NARROW: the block is 128 n x 32 c (four waves side by side along n) instead of 64 x 64: a conv with Cin <= 32 (the 3x3 conv1 of a
// level-0 subnet: 24 input channels) then pads its channel axis to 32, not 64 -- half the MFMAs and half the staging of that operand
template <int KS, bool IN_BF16, bool DOUT_BF16, bool NARROW = false>
__device__ __forceinline__ void wgrad_bf16_body(const WgradDev& p, const int split, const int n0, const int c0, const int zblock,
                                                unsigned char* const lds) {
  constexpr int ND = NARROW ? 128 : 64, NI = NARROW ? 32 : 64;        // staged channels of dout / in
  constexpr int HALO = KS / 2, TAPS = KS * KS, TH = 8;
  constexpr int IROWS = TH + 2 * HALO, IWV = 16 + 2 * HALO;          // staged halo rows, valid pixels per halo row
  constexpr int IROWP = 24;                                           // pixels per halo row in LDS (x = 16 .. 23 readable)
  constexpr int PD = WGB_PD, PI = wgb_pi(KS);
  constexpr int IPAIRS = IROWS * (IWV / 2);                           // pixel pairs of the halo tile
  constexpr int DG = ND / (DOUT_BF16 ? 8 : 4), IG = NI / (IN_BF16 ? 8 : 4);   // channel groups of the staged channels (8 bf16 / 4 fp32 per load)
  constexpr int D_ITEMS = 64 * DG / 256;                              // (pixel pair, group) items per thread: 2 / 4
  constexpr int I_GRPS = 64 * IG / 256, I_RND = (IPAIRS + 63) / 64;   // groups per thread, rounds of 64 pixel pairs
  constexpr int I_ITEMS = I_GRPS * I_RND;
  constexpr bool PF_I = IN_BF16 || KS == 1, PF_D = (DOUT_BF16 && !(IN_BF16 && KS == 3)) || KS == 1;
  unsigned char* const dT = lds;
  unsigned char* const iT = lds + ND * PD;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int wr = NARROW ? wave : (wave & 1), wc = NARROW ? 0 : (wave >> 1);

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  float bsum = 0.f;

  const int t_begin = split * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.ntiles);

  wgb_u32x4 d_reg[PF_D ? D_ITEMS : 1][2], i_reg[PF_I ? I_ITEMS : 1][2];   // (pixel, pixel + 1) of one channel group, raw 16 bytes each
  const wgb_u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto tile_origin = [&](int tile, int& b, int& y0, int& x0) {
    int tt = tile;
    const int tx = tt % p.tiles_x; tt /= p.tiles_x;
    const int ty = tt % p.tiles_y;
    b = tt / p.tiles_y; y0 = ty * TH; x0 = tx * 16;
  };
  auto load16 = [&](const float* base, size_t elem, bool is_bf16) -> wgb_u32x4 {
#if (WGB_ABL & 4)
    return (wgb_u32x4){(unsigned)elem, 0u, 0u, 0u};
#endif
    return is_bf16 ? *reinterpret_cast<const wgb_u32x4*>(reinterpret_cast<const __bf16*>(base) + elem)
                   : *reinterpret_cast<const wgb_u32x4*>(base + elem);
  };
  auto load_d = [&](int tile) {
    int b, y0, x0; tile_origin(tile, b, y0, x0);
#pragma unroll
    for (int i = 0; i < D_ITEMS; ++i) {
      const int f = tid + 256 * i;
      const int pp = WGB_SLOT(f), g = WGB_GROUP(f);
      const int gy = y0 + (pp >> 3), gx = x0 + (pp & 7) * 2;
      const int n = n0 + g * (DOUT_BF16 ? 8 : 4);
      const bool rowok = gy < p.H && n < p.N;                        // N % 8 (bf16) / % 4 (fp32) == 0: host check
      const size_t base = ((size_t)(b * p.H + gy) * p.W + gx) * p.dout_stride + n;
      d_reg[i][0] = (rowok && gx < p.W) ? load16(p.dout, base, DOUT_BF16) : zero4;
      d_reg[i][1] = (rowok && gx + 1 < p.W) ? load16(p.dout, base + p.dout_stride, DOUT_BF16) : zero4;
    }
  };
  auto load_i = [&](int tile) {
    int b, y0, x0; tile_origin(tile, b, y0, x0);
#pragma unroll
    for (int gi = 0; gi < I_GRPS; ++gi) {
      const int f = tid + 256 * gi;
      const int g = WGB_GROUP(f), l = WGB_SLOT(f);
      const int c = c0 + g * (IN_BF16 ? 8 : 4);
#pragma unroll
      for (int rnd = 0; rnd < I_RND; ++rnd) {
        const int q = l + 64 * rnd;
        const int row = q / (IWV / 2), pr = q - row * (IWV / 2);
        const int gy = y0 + row - HALO, gx = x0 + pr * 2 - HALO;
        const bool rowok = q < IPAIRS && gy >= 0 && gy < p.H && c < p.Cin;
        const size_t base = ((size_t)(b * p.H + gy) * p.W + gx) * p.in_stride + c;       // used only when in range
        i_reg[gi * I_RND + rnd][0] = (rowok && gx >= 0 && gx < p.W) ? load16(p.in, base, IN_BF16) : zero4;
        i_reg[gi * I_RND + rnd][1] = (rowok && gx + 1 >= 0 && gx + 1 < p.W) ? load16(p.in, base + p.in_stride, IN_BF16) : zero4;
      }
    }
  };
  // (pixel, pixel + 1) pairs of channel j of a group -> one dword; rows of the transposed tile are `pitch` bytes apart
  auto store_pairs = [&](unsigned char* dst, int pitch, const wgb_u32x4& v0, const wgb_u32x4& v1, bool is_bf16) {
#if (WGB_ABL & 2)
    asm volatile("" :: "v"(v0), "v"(v1), "v"(dst));
    return;
#endif
    if (is_bf16) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        *reinterpret_cast<unsigned*>(dst + (2 * k) * pitch) = (v0[k] & 0xffffu) | (v1[k] << 16);
        *reinterpret_cast<unsigned*>(dst + (2 * k + 1) * pitch) = (v0[k] >> 16) | (v1[k] & 0xffff0000u);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<unsigned*>(dst + j * pitch) = pack_bf16_pair(__uint_as_float(v0[j]), __uint_as_float(v1[j]));
    }
  };
  auto store_d = [&]() {
#pragma unroll
    for (int i = 0; i < D_ITEMS; ++i) {
      const int f = tid + 256 * i;
      const int pp = WGB_SLOT(f), g = WGB_GROUP(f);
      store_pairs(dT + (g * (DOUT_BF16 ? 8 : 4)) * PD + pp * 4, PD, d_reg[i][0], d_reg[i][1], DOUT_BF16);
    }
  };
  auto store_i = [&]() {
#pragma unroll
    for (int gi = 0; gi < I_GRPS; ++gi) {
      const int f = tid + 256 * gi;
      const int g = WGB_GROUP(f), l = WGB_SLOT(f);
#pragma unroll
      for (int rnd = 0; rnd < I_RND; ++rnd) {
        const int q = l + 64 * rnd;
        if (q < IPAIRS) {
          const int row = q / (IWV / 2), pr = q - row * (IWV / 2);
          store_pairs(iT + (g * (IN_BF16 ? 8 : 4)) * PI + row * (IROWP * 2) + pr * 4, PI,
                      i_reg[gi * I_RND + rnd][0], i_reg[gi * I_RND + rnd][1], IN_BF16);
        }
      }
    }
  };

  // operand that is not prefetched across the MFMA loop: staged between the barriers in batches of NB items -- all 2 NB loads of
  // a batch are issued before the first is stored (8 NB transient VGPRs; the MFMA operands are dead here).  One item at a time
  // exposed a global-load round trip per item: 8 per tile for the fp32 input of conv1's gradient.
  constexpr int NB = 4;                               // (8: no faster, and the 3x3 kernel spills 8 VGPRs)
  auto stage_d_direct = [&](int tile) {
    int b, y0, x0; tile_origin(tile, b, y0, x0);
#pragma unroll 1
    for (int i0 = 0; i0 < D_ITEMS; i0 += NB) {
      wgb_u32x4 v0[NB], v1[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int i = i0 + u;
        v0[u] = zero4; v1[u] = zero4;
        if (i < D_ITEMS) {
          const int f = tid + 256 * i;
          const int pp = WGB_SLOT(f), g = WGB_GROUP(f);
          const int gy = y0 + (pp >> 3), gx = x0 + (pp & 7) * 2;
          const int n = n0 + g * (DOUT_BF16 ? 8 : 4);
          const bool rowok = gy < p.H && n < p.N;
          const size_t base = ((size_t)(b * p.H + gy) * p.W + gx) * p.dout_stride + n;
          if (rowok && gx < p.W) v0[u] = load16(p.dout, base, DOUT_BF16);
          if (rowok && gx + 1 < p.W) v1[u] = load16(p.dout, base + p.dout_stride, DOUT_BF16);
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int i = i0 + u;
        if (i < D_ITEMS) {
          const int f = tid + 256 * i;
          const int pp = WGB_SLOT(f), g = WGB_GROUP(f);
          store_pairs(dT + (g * (DOUT_BF16 ? 8 : 4)) * PD + pp * 4, PD, v0[u], v1[u], DOUT_BF16);
        }
      }
    }
  };
  auto stage_i_direct = [&](int tile) {
    int b, y0, x0; tile_origin(tile, b, y0, x0);
#pragma unroll 1
    for (int it0 = 0; it0 < I_ITEMS; it0 += NB) {
      wgb_u32x4 v0[NB], v1[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int it = it0 + u;
        v0[u] = zero4; v1[u] = zero4;
        if (it < I_ITEMS) {
          const int gi = it / I_RND, rnd = it - gi * I_RND;
          const int f = tid + 256 * gi;
          const int g = WGB_GROUP(f), q = WGB_SLOT(f) + 64 * rnd;
          const int c = c0 + g * (IN_BF16 ? 8 : 4);
          if (q < IPAIRS) {
            const int row = q / (IWV / 2), pr = q - row * (IWV / 2);
            const int gy = y0 + row - HALO, gx = x0 + pr * 2 - HALO;
            const bool rowok = gy >= 0 && gy < p.H && c < p.Cin;
            const size_t base = ((size_t)(b * p.H + gy) * p.W + gx) * p.in_stride + c;
            if (rowok && gx >= 0 && gx < p.W) v0[u] = load16(p.in, base, IN_BF16);
            if (rowok && gx + 1 >= 0 && gx + 1 < p.W) v1[u] = load16(p.in, base + p.in_stride, IN_BF16);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int it = it0 + u;
        if (it < I_ITEMS) {
          const int gi = it / I_RND, rnd = it - gi * I_RND;
          const int f = tid + 256 * gi;
          const int g = WGB_GROUP(f), q = WGB_SLOT(f) + 64 * rnd;
          if (q < IPAIRS) {
            const int row = q / (IWV / 2), pr = q - row * (IWV / 2);
            store_pairs(iT + (g * (IN_BF16 ? 8 : 4)) * PI + row * (IROWP * 2) + pr * 4, PI, v0[u], v1[u], IN_BF16);
          }
        }
      }
    }
  };

  if (t_begin < t_end) {
    if constexpr (PF_D) load_d(t_begin);
    if constexpr (PF_I) load_i(t_begin);
  }
  for (int tile = t_begin; tile < t_end; ++tile) {
#if !(WGB_ABL & 16)
    __syncthreads();                              // the previous tile's fragment reads are done
#endif
    if constexpr (PF_D) store_d(); else stage_d_direct(tile);
    if constexpr (PF_I) store_i(); else stage_i_direct(tile);
#if !(WGB_ABL & 16)
    __syncthreads();
#endif
    if (tile + 1 < t_end) {                       // in flight under the MFMAs below
      if constexpr (PF_D) load_d(tile + 1);
      if constexpr (PF_I) load_i(tile + 1);
    }
    // ---- TH k-steps of 16 pixels (one tile row each) -----------------------------------------------------------------
    const unsigned char* arow = dT + (wr * 32 + r) * PD + h * 16;
    const unsigned char* brow = iT + (wc * 32 + r) * PI + h * 16;
#pragma unroll 2
    for (int y = 0; y < ((WGB_ABL & 1) ? 0 : TH); ++y) {
      const wgb_bf16x8 af = *reinterpret_cast<const wgb_bf16x8*>(arow + y * 32);
#if !(WGB_ABL & 8)
      {
        const __bf16* av = reinterpret_cast<const __bf16*>(&af);
        float sacc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sacc += (float)av[j];
        bsum += sacc;
      }
#endif
#pragma unroll
      for (int dy = 0; dy < KS; ++dy) {
        const wgb_u32x4 lo = *reinterpret_cast<const wgb_u32x4*>(brow + (y + dy) * (IROWP * 2));
        if constexpr (KS == 1) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, *reinterpret_cast<const wgb_bf16x8*>(&lo), acc[0], 0, 0, 0);
        } else {
          const wgb_u32x4 hi = *reinterpret_cast<const wgb_u32x4*>(brow + (y + dy) * (IROWP * 2) + 16);
          wgb_u32x4 s1, s2;
#if (WGB_ABL & 8)
          s1 = lo; s2 = hi;
#else
          s1[0] = __builtin_amdgcn_alignbyte(lo[1], lo[0], 2);
          s1[1] = __builtin_amdgcn_alignbyte(lo[2], lo[1], 2);
          s1[2] = __builtin_amdgcn_alignbyte(lo[3], lo[2], 2);
          s1[3] = __builtin_amdgcn_alignbyte(hi[0], lo[3], 2);
          s2[0] = lo[1]; s2[1] = lo[2]; s2[2] = lo[3]; s2[3] = hi[0];
#endif
          acc[dy * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, *reinterpret_cast<const wgb_bf16x8*>(&lo), acc[dy * 3 + 0], 0, 0, 0);
          acc[dy * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, *reinterpret_cast<const wgb_bf16x8*>(&s1), acc[dy * 3 + 1], 0, 0, 0);
          acc[dy * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, *reinterpret_cast<const wgb_bf16x8*>(&s2), acc[dy * 3 + 2], 0, 0, 0);
        }
      }
    }
  }

  // ---- slab write: D[row n = (e & 3) + 8 (e >> 2) + 4 h][col c = r] -------------------------------------------------
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int n = n0 + wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      const int c = c0 + wc * 32 + r;
      if (n < p.Nr && c < p.Cc) p.partial[(((size_t)split * TAPS + t) * p.Nr + n) * p.Cc + c] = acc[t][e];
    }
  // bias partial: lane (r, h) holds the sum over its pixels of dout[.][n0 + 32 wr + r]
  bsum += __shfl_xor(bsum, 32);
  if (zblock == 0 && wc == 0 && h == 0) {
    const int n = n0 + wr * 32 + r;
    if (n < p.Nr) p.bpartial[(size_t)split * p.Nr + n] = bsum;
  }
}

template <int KS>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_group_kernel(WgradGroup g) {
  constexpr int LDS_WIDE = 64 * WGB_PD + 64 * wgb_pi(KS), LDS_NARROW = 128 * WGB_PD + 32 * wgb_pi(KS);
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_WIDE > LDS_NARROW ? LDS_WIDE : LDS_NARROW];
  const int pi = group_problem(g, blockIdx.x, false);
  const WgradProb& q = g.p[pi];
  const int lb = blockIdx.x - q.block_begin;
  const int split = lb % q.S, nb = (lb / q.S) % q.nblk, cb = lb / (q.S * q.nblk);
  // operand storage types are per problem (block-uniform): conv2's gradient reads a bf16 hidden tensor and the fp32 tail
  // gradient, conv1's the fp32 input and the bf16 hidden gradient
  if (q.narrow_c) {
    if (q.d.in_bf16 && !q.d.dout_bf16) wgrad_bf16_body<KS, true, false, true>(q.d, split, nb * 128, cb * 32, cb, lds);
    else if (!q.d.in_bf16 && q.d.dout_bf16) wgrad_bf16_body<KS, false, true, true>(q.d, split, nb * 128, cb * 32, cb, lds);
    else wgrad_bf16_body<KS, true, true, true>(q.d, split, nb * 128, cb * 32, cb, lds);
    return;
  }
  if (q.d.in_bf16 && !q.d.dout_bf16) wgrad_bf16_body<KS, true, false>(q.d, split, nb * 64, cb * 64, cb, lds);
  else if (!q.d.in_bf16 && q.d.dout_bf16) wgrad_bf16_body<KS, false, true>(q.d, split, nb * 64, cb * 64, cb, lds);
  else wgrad_bf16_body<KS, true, true>(q.d, split, nb * 64, cb * 64, cb, lds);
}

// ---- grouped launch (host) ---------------------------------------------------------------------------------------
struct WgradGroupPlan { WgradGroup g; int grad_blocks, red_blocks; size_t bytes; bool wino; int th; bool mfma_bf16; };

static bool g_wgrad_wide_c = true;      // test hook bit 7 clears: N <= 32 problems use the 64 x 32 block shape like everything else
                                        // (and Cin <= 32 problems of the bf16 kernel its 64 x 64 blocks instead of 128 x 32)
static bool g_wgrad_bf16_mfma = true;   // test hook bit 6 clears: bf16-operand problems accumulate on the f32 pipe (Winograd)
void wgrad_set_bf16_mfma(int on) { g_wgrad_bf16_mfma = on != 0; }
void wgrad_set_wide_c(int on) { g_wgrad_wide_c = on != 0; }

static int plan_group(const sininn_wgrad_item* items, int n, int B, int H, int W, int ksize, float* ws, WgradGroupPlan& pl) {
  SININN_CHECK(items && n >= 1 && n <= WG_MAXP, "wgrad_group: 1..%d problems per group", WG_MAXP);
  SININN_CHECK(ksize == 1 || ksize == 3, "wgrad_group: ksize %d not in {1,3}", ksize);
  SININN_CHECK(B > 0 && H > 0 && W > 0, "wgrad_group: bad shape");
  for (int i = 0; i < n; ++i) {
    const sininn_wgrad_item& it = items[i];
    SININN_CHECK(it.struct_bytes == sizeof(sininn_wgrad_item),
                 "wgrad_group: problem %d: struct_bytes = %zu, this library's sininn_wgrad_item has %zu (ABI %d): zero the "
                 "descriptor and set struct_bytes = sizeof(sininn_wgrad_item)", i, it.struct_bytes, sizeof(sininn_wgrad_item),
                 SININN_ABI_VERSION);
    SININN_CHECK(it.gap_len == 0 || (it.gap_len > 0 && it.gap_begin >= 0 && it.gap_begin + it.gap_len <= it.Cin),
                 "wgrad_group: bad channel gap [%d, +%d) for Cin=%d in problem %d", it.gap_begin, it.gap_len, it.Cin, i);
    SININN_CHECK((it.in_bf16 == 0 || it.in_bf16 == 1) && (it.dout_bf16 == 0 || it.dout_bf16 == 1),
                 "wgrad_group: dtype flags must be 0 / 1 (problem %d)", i);
  }
  // a group whose every problem has a bf16 operand runs on the bf16 matrix pipe (one operand is stored as bf16 already,
  // the other is rounded while it is staged); anything else accumulates exact fp32 products on the f32 pipe
  bool all_mixed = g_wgrad_bf16_mfma;
  for (int i = 0; i < n; ++i)
    all_mixed = all_mixed && (items[i].in_bf16 || items[i].dout_bf16) && items[i].Cin % 8 == 0 && items[i].N % 8 == 0;
  pl.mfma_bf16 = all_mixed;
  pl.wino = (ksize == 3) && !pl.mfma_bf16;
  pl.th = pl.mfma_bf16 ? 8 : (pl.wino ? 4 : WG_TH);
  const int bnw = (pl.wino || pl.mfma_bf16) ? 64 : 32, bcw = pl.mfma_bf16 ? 64 : 32;
  const int taps = ksize * ksize;
  const int tiles_x = (W + 15) / 16, tiles_y = (H + pl.th - 1) / pl.th;
  const int ntiles = B * tiles_x * tiles_y;
  int out_tiles = 0;
  for (int i = 0; i < n; ++i) {
    const sininn_wgrad_item& it = items[i];
    SININN_CHECK(it.Cin > 0 && it.Cin % 4 == 0 && it.N > 0 && it.N % 4 == 0, "wgrad_group: Cin=%d and N=%d must be multiples of 4", it.Cin, it.N);
    const bool wide_c = pl.wino && it.N <= 32 && g_wgrad_wide_c;
    const bool narrow_c = pl.mfma_bf16 && it.Cin <= 32 && it.N >= 128 && g_wgrad_wide_c;
    const int bn_i = wide_c ? 32 : (narrow_c ? 128 : bnw), bc_i = wide_c ? 64 : (narrow_c ? 32 : bcw);
    out_tiles += ((it.N + bn_i - 1) / bn_i) * ((it.Cin + bc_i - 1) / bc_i);
  }
  // two blocks per CU; all problems of a group see the same pixels, so one split count serves them all
  // (SININN_WGRAD_BLOCKS: diagnostic -- target block count of the grouped gradient launch, e.g. 256 = one block per CU, which
  // leaves half of every CU's registers / LDS to the pass chains' kernels that run beside it)
  static const int target_blocks = getenv("SININN_WGRAD_BLOCKS") ? atoi(getenv("SININN_WGRAD_BLOCKS")) : 512;
  int S = (target_blocks > 0 ? target_blocks : 512) / out_tiles;
  if (S < 1) S = 1;
  if (S > ntiles) S = ntiles;
  const int tps = (ntiles + S - 1) / S;
  S = (ntiles + tps - 1) / tps;
  pl.g.n = n; pl.g.taps = taps;
  size_t off = 0;                                    // floats
  int gb = 0, rb = 0;
  constexpr int RC3 = 256 / 9, RC1 = 256;
  for (int i = 0; i < n; ++i) {
    const sininn_wgrad_item& it = items[i];
    WgradProb& q = pl.g.p[i];
    q.wide_c = (pl.wino && it.N <= 32 && g_wgrad_wide_c) ? 1 : 0;
    q.narrow_c = (pl.mfma_bf16 && it.Cin <= 32 && it.N >= 128 && g_wgrad_wide_c) ? 1 : 0;
    const int bn_i = q.wide_c ? 32 : (q.narrow_c ? 128 : bnw), bc_i = q.wide_c ? 64 : (q.narrow_c ? 32 : bcw);
    q.nblk = (it.N + bn_i - 1) / bn_i; q.cblk = (it.Cin + bc_i - 1) / bc_i; q.S = S;
    WgradDev& d = q.d;
    d.in = it.in; d.in_stride = it.in_stride; d.Cin = it.Cin; d.dout = it.dout; d.dout_stride = it.dout_stride; d.N = it.N;
    d.B = B; d.H = H; d.W = W; d.tiles_x = tiles_x; d.tiles_y = tiles_y; d.ntiles = ntiles; d.tiles_per_split = tps;
    d.Nr = q.nblk * bn_i; d.Cc = q.cblk * bc_i;
    d.in_bf16 = it.in_bf16; d.dout_bf16 = it.dout_bf16;
    d.in_gs = it.in_group_stride > 0 ? (size_t)it.in_group_stride : 0;
    d.dout_gs = it.dout_group_stride > 0 ? (size_t)it.dout_group_stride : 0;
    // the staging descriptors (WgStage) hold 32-bit byte offsets from a tile's origin pixel
    SININN_CHECK(H < 65535 && W < 65535 &&
                 (unsigned long long)((it.Cin + 7) / 8) * (d.in_gs ? d.in_gs : 8) * 4ull + 64ull * W * (unsigned)it.in_stride < (1ull << 32) &&
                 (unsigned long long)((it.N + 7) / 8) * (d.dout_gs ? d.dout_gs : 8) * 4ull + 64ull * W * (unsigned)it.dout_stride < (1ull << 32),
                 "wgrad group: operand %d spans more than 4 GB from a tile origin (32-bit staging offsets)", i);
    d.partial = ws ? ws + off : nullptr; off += (size_t)S * taps * d.Nr * d.Cc;
    d.bpartial = ws ? ws + off : nullptr; off += ((size_t)S * d.Nr + 3) / 4 * 4;
    q.gw = it.gw; q.gb = it.gb;
    q.gap_begin = it.gap_len > 0 ? it.gap_begin : 0; q.gap_len = it.gap_len > 0 ? it.gap_len : 0;
    q.block_begin = gb; gb += S * q.nblk * q.cblk;
    q.red_begin = rb;
    const int ncol = d.Nr * (d.Cc / 4), rc = (taps == 9) ? RC3 : RC1;
    rb += (ncol + rc - 1) / rc + (it.gb ? (it.N + 255) / 256 : 0);
  }
  pl.grad_blocks = gb; pl.red_blocks = rb; pl.bytes = off * sizeof(float);
  return 0;
}

size_t wgrad_group_workspace_bytes(const sininn_wgrad_item* items, int n, int B, int H, int W, int ksize) {
  WgradGroupPlan pl;
  if (plan_group(items, n, B, H, W, ksize, nullptr, pl)) return 0;
  return pl.bytes;
}

int wgrad_group_launch(const sininn_wgrad_item* items, int n, int B, int H, int W, int ksize, void* ws, size_t ws_bytes,
                       hipStream_t st) {
  SININN_CHECK(ws != nullptr && aligned16(ws), "wgrad_group: workspace must be 16-byte aligned");
  WgradGroupPlan pl;
  if (int rc = plan_group(items, n, B, H, W, ksize, static_cast<float*>(ws), pl)) return rc;
  SININN_CHECK(ws_bytes >= pl.bytes, "wgrad_group: workspace too small (%zu < %zu)", ws_bytes, pl.bytes);
  for (int i = 0; i < n; ++i) {
    const sininn_wgrad_item& it = items[i];
    SININN_CHECK(it.in && it.dout && it.gw, "wgrad_group: null pointer in problem %d", i);
    SININN_CHECK((it.in_stride >= it.Cin || it.in_group_stride > 0) && it.in_stride % 4 == 0 && aligned16(it.in), "wgrad_group: in must be 16-byte aligned, stride %% 4 == 0");
    SININN_CHECK((it.dout_stride >= it.N || it.dout_group_stride > 0) && it.dout_stride % 4 == 0 && aligned16(it.dout), "wgrad_group: dout must be 16-byte aligned, stride %% 4 == 0");
    SININN_CHECK((it.in_group_stride <= 0 && it.dout_group_stride <= 0) || (pl.wino && !it.in_bf16 && !it.dout_bf16),
                 "wgrad_group: channel-group-major operands need the fp32 3x3 Winograd kernel");
    SININN_CHECK((it.in_bf16 == 0 || it.in_bf16 == 1) && (it.dout_bf16 == 0 || it.dout_bf16 == 1), "wgrad_group: dtype flags must be 0 / 1");
    SININN_CHECK(it.gap_len == 0 || (it.gap_len > 0 && it.gap_begin >= 0 && it.gap_begin + it.gap_len <= it.Cin),
                 "wgrad_group: bad channel gap [%d, +%d) for Cin=%d in problem %d", it.gap_begin, it.gap_len, it.Cin, i);
  }
  if (pl.mfma_bf16) {
    if (ksize == 3) hipLaunchKernelGGL((wgrad_bf16_group_kernel<3>), dim3(pl.grad_blocks), dim3(256), 0, st, pl.g);
    else hipLaunchKernelGGL((wgrad_bf16_group_kernel<1>), dim3(pl.grad_blocks), dim3(256), 0, st, pl.g);
    SININN_LAUNCH_CHECK("wgrad_bf16_group");
    if (ksize == 3) hipLaunchKernelGGL((wgrad_reduce_group_kernel<9>), dim3(pl.red_blocks), dim3(256), 0, st, pl.g);
    else hipLaunchKernelGGL((wgrad_reduce_group_kernel<1>), dim3(pl.red_blocks), dim3(256), 0, st, pl.g);
  } else if (pl.wino) {
    hipLaunchKernelGGL((wgrad_wino_group_kernel<4, 2>), dim3(pl.grad_blocks), dim3(256), 0, st, pl.g);
    SININN_LAUNCH_CHECK("wgrad_wino_group");
    hipLaunchKernelGGL((wgrad_reduce_group_kernel<9>), dim3(pl.red_blocks), dim3(256), 0, st, pl.g);
  } else {
    hipLaunchKernelGGL((wgrad32_group_kernel<1>), dim3(pl.grad_blocks), dim3(256), 0, st, pl.g);
    SININN_LAUNCH_CHECK("wgrad32_group");
    hipLaunchKernelGGL((wgrad_reduce_group_kernel<1>), dim3(pl.red_blocks), dim3(256), 0, st, pl.g);
  }
  SININN_LAUNCH_CHECK("wgrad_reduce_group");
  return 0;
}

}  // namespace sininn

// Implicit-GEMM convolution on v_mfma_f32_16x16x4_f32 with LDS-staged NHWC halo tiles and fused
// epilogues (ReLU / affine coupling forward + inverse + log-det / ReLU-mask / add).
//
// Replaces, for the sin-inn hot path: nn.Conv2d in subnet_conv / subnet_conv_1x1 (archs.py:11-17)
// and the elementwise tail of FrEIA's GLOWCouplingBlock (archs.py:61-64).
//
// GEMM view:  D[pixel][col] = sum_{tap, c} in[pixel + off(tap)][c] * w[tap][col][c]
//   * block tile  = TH x 16 output pixels of one image  x  (WN*NT*16) packed columns
//   * K loop      = (channel chunk of CK) x (tap); the input halo tile of a chunk is staged ONCE in
//                   LDS and re-used by all 9 taps (the A operand is just a shifted LDS address)
//   * operands    = k-contiguous rows in LDS (stride CK+4 floats => conflict-free ds_read_b64),
//                   each 8-byte read feeds two MFMAs (k and k+1)
//   * pipeline    = next iteration's weights (+ next chunk's halo tile) are fetched global->VGPR
//                   while the current iteration's MFMAs run, then written to the other LDS buffer;
//                   one barrier per iteration.
#include "common.h"

namespace sininn {

struct ConvDev {
  const float* in; int in_stride; int Cin;
  const float* w; const float* bias; int Np;
  int B, H, W;
  int CK;
  float* out; int out_stride; int N; const int* out_map;
  const float* v; int v_stride;
  float* out2; int out2_stride;
  float* sbuf; float* logdet;
  int Co; float clamp;
  const float* mask; int mask_stride;
  const float* addend; int addend_stride; const int* addend_map;
  int tiles_x, tiles_y;
  int mode;
};

template <int KS, int TH, int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvDev p) {
  constexpr int HALO = KS / 2;
  constexpr int IW = 16 + 2 * HALO;
  constexpr int IH = TH + 2 * HALO;
  constexpr int NPIX_IN = IH * IW;
  constexpr int TAPS = KS * KS;
  constexpr int BN = WN * NT * 16;
  constexpr int IN_F4 = (NPIX_IN * 8 + 255) / 256;  // float4 per thread for a CK<=32 halo tile
  constexpr int W_F4 = (BN * 8 + 255) / 256;        // float4 per thread for a CK<=32 weight tile
  static_assert(WM * MT == TH && WM * WN == 4, "bad wave layout");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int CK = p.CK;
  const int S = CK + 4;
  const int c4n = CK >> 2;
  float* const in_lds0 = smem;
  float* const in_lds1 = smem + NPIX_IN * S;
  float* const w_lds0 = smem + 2 * NPIX_IN * S;
  float* const w_lds1 = w_lds0 + BN * S;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 15, kq = lane >> 4;

  int bid = blockIdx.x;
  const int tx = bid % p.tiles_x; bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * 16;
  const int n0 = blockIdx.y * BN;

  // ---- per-thread staging descriptors (constant over the K loop) --------------------------------
  int in_goff[IN_F4], in_loff[IN_F4];
#pragma unroll
  for (int r = 0; r < IN_F4; ++r) {
    const int f = tid + 256 * r;
    const int pix = f / c4n, c4 = f - pix * c4n;
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
    const int row = f / c4n, c4 = f - row * c4n;
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
  auto store_in = [&](float* dst) {
#pragma unroll
    for (int r = 0; r < IN_F4; ++r)
      if (in_loff[r] >= 0) *reinterpret_cast<f32x4*>(dst + in_loff[r]) = in_reg[r];
  };
  auto load_w = [&](int chunk, int tap) {
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
      if (w_loff[r] >= 0) *reinterpret_cast<f32x4*>(dst + w_loff[r]) = w_reg[r];
  };

  // ---- fragment base offsets (floats) -----------------------------------------------------------
  int a_base[MT], b_base[NT];
#pragma unroll
  for (int m = 0; m < MT; ++m) a_base[m] = ((wm * MT + m) * IW + li) * S + 2 * kq;
#pragma unroll
  for (int n = 0; n < NT; ++n) b_base[n] = ((wn * NT + n) * 16 + li) * S + 2 * kq;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- prologue ---------------------------------------------------------------------------------
  load_in(0);
  load_w(0, 0);
  store_in(in_lds0);
  store_w(w_lds0);
  __syncthreads();

  const int ksteps = CK >> 3;
  int chunk = 0, tap = 0;
  for (int it = 0; it < nit; ++it) {
    int ntap = tap + 1, nchunk = chunk;
    if (ntap == TAPS) { ntap = 0; nchunk = chunk + 1; }
    const bool has_next = (it + 1) < nit;
    const bool next_in = has_next && (ntap == 0);
    if (has_next) load_w(nchunk, ntap);
    if (next_in) load_in(nchunk);

    const float* A = (chunk & 1) ? in_lds1 : in_lds0;
    const float* Bw = (it & 1) ? w_lds1 : w_lds0;
    const int dy = tap / KS, dx = tap - dy * KS;
    const int a_off = (dy * IW + dx) * S;
    for (int ks = 0; ks < ksteps; ++ks) {
      float2 af[MT], bf[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const float2*>(A + a_base[m] + a_off + ks * 8);
#pragma unroll
      for (int n = 0; n < NT; ++n) bf[n] = *reinterpret_cast<const float2*>(Bw + b_base[n] + ks * 8);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].x, bf[n].x, acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].y, bf[n].y, acc[m][n], 0, 0, 0);
        }
    }

    if (has_next) store_w(((it + 1) & 1) ? w_lds1 : w_lds0);
    if (next_in) store_in((nchunk & 1) ? in_lds1 : in_lds0);
    __syncthreads();
    tap = ntap; chunk = nchunk;
  }

  const int MODE = p.mode;
  // ---- epilogue: lane holds D[row = 4*kq + r][col = li] of every 16x16 tile ---------------------
  float ld_acc = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int gy = y0 + wm * MT + m;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int tile = blockIdx.y * (WN * NT) + wn * NT + n;   // global 16-column tile index
      const int col = tile * 16 + li;
      if (MODE == SININN_CONV_COUPLE_FWD || MODE == SININN_CONV_COUPLE_INV) {
        // tile = [ s[8*tile .. +7] | t[8*tile .. +7] ] : lanes li<8 hold s, their partner li+8 holds t.
        const bool colok = col < p.Np;
        const float bia = (colok && p.bias) ? p.bias[col] : 0.f;
        float mine[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[r] = acc[m][n][r] + bia;
        const bool lo = li < 8;
        // low lanes finish rows r=0,1 ; high lanes finish rows r=2,3 (two shuffles instead of four)
        const float send0 = lo ? mine[2] : mine[0];
        const float send1 = lo ? mine[3] : mine[1];
        const float recv0 = __shfl_xor(send0, 8);
        const float recv1 = __shfl_xor(send1, 8);
        const int c = tile * 8 + (li & 7);
        const int rbase = lo ? 0 : 2;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float s = lo ? mine[q] : (q == 0 ? recv0 : recv1);
          const float t = lo ? (q == 0 ? recv0 : recv1) : mine[2 + q];
          const int gx = x0 + 4 * kq + rbase + q;
          if (c < p.Co && gy < p.H && gx < p.W) {
            const size_t pix = (size_t)(b * p.H + gy) * p.W + gx;
            const float vv = p.v[pix * p.v_stride + c];
            const float L = glow_log_e(s, p.clamp);
            const float e = expf(L);
            float yv;
            if (MODE == SININN_CONV_COUPLE_FWD) { yv = e * vv + t; ld_acc += L; }
            else { yv = (vv - t) / e; ld_acc -= L; }
            const int oc = p.out_map ? p.out_map[c] : c;
            p.out[pix * p.out_stride + oc] = yv;
            if (p.out2) p.out2[pix * p.out2_stride + c] = yv;
            if (p.sbuf) p.sbuf[pix * p.Co + c] = s;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gx = x0 + 4 * kq + r;
          if (col < p.N && gy < p.H && gx < p.W) {
            const size_t pix = (size_t)(b * p.H + gy) * p.W + gx;
            float val = acc[m][n][r];
            if (MODE == SININN_CONV_RELU) {
              val = fmaxf(val + p.bias[col], 0.f);
            } else if (MODE == SININN_CONV_LINEAR) {
              val = val + (p.bias ? p.bias[col] : 0.f);
            } else if (MODE == SININN_CONV_MASK) {
              val = (p.mask[pix * p.mask_stride + col] > 0.f) ? val : 0.f;
            } else if (MODE == SININN_CONV_ADD) {
              const int ac = p.addend_map ? p.addend_map[col] : col;
              val += p.addend[pix * p.addend_stride + ac];
            }
            p.out[pix * p.out_stride + col] = val;
          }
        }
      }
    }
  }
  if (MODE == SININN_CONV_COUPLE_FWD || MODE == SININN_CONV_COUPLE_INV) {
    if (p.logdet) {
      const float tot = wave_sum(ld_acc);
      if (lane == 0) atomicAdd(p.logdet + b, tot);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------
template <int KS, int TH, int WM, int WN, int MT, int NT>
static int launch_cfg(const ConvDev& d, int mode, hipStream_t st) {
  constexpr int HALO = KS / 2;
  constexpr int NPIX_IN = (TH + 2 * HALO) * (16 + 2 * HALO);
  constexpr int BN = WN * NT * 16;
  const int S = d.CK + 4;
  const size_t lds = (size_t)2 * (NPIX_IN + BN) * S * sizeof(float);
  dim3 grid(d.tiles_x * d.tiles_y * d.B, (d.Np + BN - 1) / BN);
  (void)mode;
  auto k = conv_mfma_kernel<KS, TH, WM, WN, MT, NT>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("conv: cannot raise LDS limit to %zu", lds); return 1; }
  }
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, d);
  SININN_LAUNCH_CHECK("conv_mfma");
  return 0;
}

// Tile-shape choice.  ntile16 = packed column tiles; pick the widest block that divides the columns
// well, and the 4-row spatial tile when the 8-row one would leave the 256 CUs under-filled.
template <int KS>
static int dispatch(ConvDev& d, int mode, hipStream_t st, int force_cfg) {
  const int nt16 = d.Np / 16;
  auto set_tiles = [&](int th) { d.tiles_x = (d.W + 15) / 16; d.tiles_y = (d.H + th - 1) / th; };
  const long pix_tiles8 = (long)d.B * ((d.H + 7) / 8) * ((d.W + 15) / 16);
  // column-block width in 16-col tiles
  int bnt;
  if (nt16 % 8 == 0) bnt = 8;
  else if (nt16 % 6 == 0) bnt = 6;
  else if (nt16 % 3 == 0) bnt = 3;
  else if (nt16 <= 2) bnt = 2;
  else bnt = 4;
  const long blocks8 = pix_tiles8 * ((nt16 + bnt - 1) / bnt);
  bool small = blocks8 < 512;            // < 2 blocks per CU with the 8-row tile -> use 4-row tiles
  if (force_cfg == 1) small = false;
  if (force_cfg == 2) small = true;
  if (!small) {
    set_tiles(8);
    switch (bnt) {
      case 8: return launch_cfg<KS, 8, 2, 2, 4, 4>(d, mode, st);
      case 6: return launch_cfg<KS, 8, 2, 2, 4, 3>(d, mode, st);
      case 4: return launch_cfg<KS, 8, 2, 2, 4, 2>(d, mode, st);
      case 3: return launch_cfg<KS, 8, 4, 1, 2, 3>(d, mode, st);
      default: return launch_cfg<KS, 8, 4, 1, 2, 2>(d, mode, st);
    }
  } else {
    set_tiles(4);
    switch (bnt) {
      case 8: return launch_cfg<KS, 4, 1, 4, 4, 2>(d, mode, st);
      case 6: return launch_cfg<KS, 4, 2, 2, 2, 3>(d, mode, st);
      case 4: return launch_cfg<KS, 4, 2, 2, 2, 2>(d, mode, st);
      case 3: return launch_cfg<KS, 4, 4, 1, 1, 3>(d, mode, st);
      default: return launch_cfg<KS, 4, 4, 1, 1, 2>(d, mode, st);
    }
  }
}

static int g_force_cfg = 0;   // test hook: 0 auto, 1 force 8-row tiles, 2 force 4-row tiles
static int g_force_ck = 0;    // test hook: override the channel chunk

int conv_launch(const sininn_conv_args* a, hipStream_t st) {
  SININN_CHECK(a != nullptr, "conv: null args");
  SININN_CHECK(a->ksize == 1 || a->ksize == 3, "conv: ksize %d not in {1,3}", a->ksize);
  SININN_CHECK(a->Cin > 0 && a->Cin % 8 == 0, "conv: Cin=%d must be a positive multiple of 8", a->Cin);
  SININN_CHECK(a->Np > 0 && a->Np % 16 == 0, "conv: Np=%d must be a positive multiple of 16", a->Np);
  SININN_CHECK(a->B > 0 && a->H > 0 && a->W > 0, "conv: bad image shape %dx%dx%d", a->B, a->H, a->W);
  SININN_CHECK((long)a->B * a->H * a->W * (long)(a->in_stride > a->out_stride ? a->in_stride : a->out_stride) < (1l << 31),
               "conv: tensor too large for 32-bit pixel offsets");
  SININN_CHECK(a->in && a->w && a->out, "conv: null tensor pointer");
  SININN_CHECK(a->in_stride >= a->Cin && a->in_stride % 4 == 0 && aligned16(a->in),
               "conv: input must be 16-byte aligned with stride %% 4 == 0 (stride=%d)", a->in_stride);
  SININN_CHECK(aligned16(a->w), "conv: packed weights must be 16-byte aligned");
  const bool couple = a->mode == SININN_CONV_COUPLE_FWD || a->mode == SININN_CONV_COUPLE_INV;
  if (couple) {
    SININN_CHECK(a->Co > 0 && a->Co % 8 == 0 && a->Np == 2 * a->Co, "conv: coupling needs Np == 2*Co, Co %% 8 == 0 (Np=%d Co=%d)", a->Np, a->Co);
    SININN_CHECK(a->v != nullptr && a->v_stride >= a->Co, "conv: coupling needs v");
    SININN_CHECK(a->clamp > 0.f, "conv: clamp must be > 0");
    SININN_CHECK(a->out_stride >= a->Co, "conv: out_stride < Co");
  } else {
    SININN_CHECK(a->N > 0 && a->N <= a->Np && a->out_stride >= a->N, "conv: bad N=%d (Np=%d, out_stride=%d)", a->N, a->Np, a->out_stride);
    if (a->mode == SININN_CONV_RELU) SININN_CHECK(a->bias != nullptr, "conv: RELU mode needs bias");
    if (a->mode == SININN_CONV_MASK) SININN_CHECK(a->mask != nullptr && a->mask_stride >= a->N, "conv: MASK mode needs mask");
    if (a->mode == SININN_CONV_ADD) SININN_CHECK(a->addend != nullptr, "conv: ADD mode needs addend");
  }
  ConvDev d;
  d.in = a->in; d.in_stride = a->in_stride; d.Cin = a->Cin;
  d.w = a->w; d.bias = a->bias; d.Np = a->Np;
  d.B = a->B; d.H = a->H; d.W = a->W;
  d.out = a->out; d.out_stride = a->out_stride; d.N = a->N; d.out_map = a->out_map;
  d.v = a->v; d.v_stride = a->v_stride; d.out2 = a->out2; d.out2_stride = a->out2_stride;
  d.sbuf = a->sbuf; d.logdet = a->logdet; d.Co = a->Co; d.clamp = a->clamp;
  d.mask = a->mask; d.mask_stride = a->mask_stride;
  d.addend = a->addend; d.addend_stride = a->addend_stride; d.addend_map = a->addend_map;
  d.mode = a->mode;
  SININN_CHECK(a->mode >= 0 && a->mode <= SININN_CONV_LINEAR, "conv: unknown mode %d", a->mode);
  // channel chunk: the largest of 32/24/16/8 that divides Cin
  int ck = 8;
  for (int c : {32, 24, 16, 8}) if (a->Cin % c == 0) { ck = c; break; }
  if (g_force_ck && a->Cin % g_force_ck == 0 && g_force_ck % 8 == 0 && g_force_ck <= 32) ck = g_force_ck;
  d.CK = ck;
  if (a->ksize == 3) return dispatch<3>(d, a->mode, st, g_force_cfg);
  return dispatch<1>(d, a->mode, st, g_force_cfg);
}

void conv_set_test_hooks(int force_cfg, int force_ck) { g_force_cfg = force_cfg; g_force_ck = force_ck; }

}  // namespace sininn

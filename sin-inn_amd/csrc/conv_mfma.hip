// Host-side validation + dispatch of the implicit-GEMM conv engine (kernel: conv_mfma_impl.h; the 3x3 and 1x1
// instantiations are compiled in conv_mfma_k3.hip / conv_mfma_k1.hip so the two translation units build in parallel).
#include "conv_mfma_impl.h"

namespace sininn {

int conv_dispatch_k3(ConvDev& d, hipStream_t st, int force_cfg);
int conv_dispatch_k1(ConvDev& d, hipStream_t st, int force_cfg);
int conv32_dispatch_k3(ConvDev& d, hipStream_t st, int force_cfg, bool must);
int conv32_dispatch_k1(ConvDev& d, hipStream_t st, int force_cfg, bool must);
int wino_dispatch_k3(ConvDev& d, hipStream_t st, int cg2);

int g_conv_dma = 0;
static int g_force_cfg = 0;   // test hook: 0 auto, 1 force 8-row tiles, 2 force 4-row tiles
static int g_force_ck = 0;
static int g_ablate = 0;     // diagnostic: see ConvDev::ablate (results are wrong when set)    // test hook: override the channel chunk

int conv_bf16_launch(const sininn_conv_args* a, hipStream_t st);

// argument validation + device-side descriptor of one fp32 conv (shared by conv_launch and the fused 1x1 pair)
int conv_prepare(const sininn_conv_args* a, ConvDev& d) {
  SININN_CHECK(a != nullptr, "conv: null args");
  SININN_CHECK(a->ksize == 1 || a->ksize == 3, "conv: ksize %d not in {1,3}", a->ksize);
  SININN_CHECK(a->Cin > 0 && a->Cin % 8 == 0, "conv: Cin=%d must be a positive multiple of 8", a->Cin);
  SININN_CHECK(a->Np > 0 && a->Np % 16 == 0, "conv: Np=%d must be a positive multiple of 16", a->Np);
  SININN_CHECK(a->B > 0 && a->H > 0 && a->W > 0, "conv: bad image shape %dx%dx%d", a->B, a->H, a->W);
  SININN_CHECK((long)a->B * a->H * a->W * (long)(a->in_stride > a->out_stride ? a->in_stride : a->out_stride) < (1l << 31),
               "conv: tensor too large for 32-bit pixel offsets");
  SININN_CHECK(a->in && a->w && a->out, "conv: null tensor pointer");
  SININN_CHECK((a->in_stride >= a->Cin || a->in_group_stride > 0) && a->in_stride % 4 == 0 && aligned16(a->in),
               "conv: input must be 16-byte aligned with stride %% 4 == 0 (stride=%d)", a->in_stride);
  SININN_CHECK(aligned16(a->w), "conv: packed weights must be 16-byte aligned");
  const bool couple = a->mode == SININN_CONV_COUPLE_FWD || a->mode == SININN_CONV_COUPLE_INV;
  if (couple) {
    SININN_CHECK(a->Co > 0 && a->Co % 8 == 0 && a->Np == 2 * a->Co, "conv: coupling needs Np == 2*Co, Co %% 8 == 0 (Np=%d Co=%d)", a->Np, a->Co);
    SININN_CHECK(a->v != nullptr && a->v_stride >= a->Co, "conv: coupling needs v");
    SININN_CHECK(a->clamp > 0.f, "conv: clamp must be > 0");
    SININN_CHECK(a->out_stride >= a->Co, "conv: out_stride < Co");
  } else {
    SININN_CHECK(a->N > 0 && a->N <= a->Np && (a->out_stride >= a->N || a->out_group_stride > 0), "conv: bad N=%d (Np=%d, out_stride=%d)", a->N, a->Np, a->out_stride);
    if (a->mode == SININN_CONV_RELU) SININN_CHECK(a->bias != nullptr, "conv: RELU mode needs bias");
    if (a->mode == SININN_CONV_MASK) SININN_CHECK(a->mask != nullptr && (a->mask_stride >= a->N || a->mask_group_stride > 0), "conv: MASK mode needs mask");
    const bool cbwd = a->mode == SININN_CONV_ADD_CBWD_FWD || a->mode == SININN_CONV_ADD_CBWD_INV;
    if (a->mode == SININN_CONV_ADD || cbwd) SININN_CHECK(a->addend != nullptr, "conv: ADD mode needs addend");
    if (cbwd) SININN_CHECK(a->v && a->sbuf && a->out2 && a->Co == a->N && a->out_stride >= 2 * a->Co && a->clamp > 0.f, "conv: ADD_CBWD needs v, sbuf, out2, Co == N, out_stride >= 2*Co");
    if ((a->mode == SININN_CONV_LINEAR || a->mode == SININN_CONV_ADD) && a->mask)
      SININN_CHECK(a->Co >= 0 && a->Co % 4 == 0 && a->mask_stride >= a->N && a->mask_stride % 4 == 0 && aligned16(a->mask) && a->mask_group_stride <= 0,
                   "conv: LeakyReLU tail needs Co %% 4 == 0 and a 16-byte aligned pixel-major mask (Co=%d, mask_stride=%d)", a->Co, a->mask_stride);
    if (a->mode == SININN_CONV_IRN_FWD || a->mode == SININN_CONV_IRN_INV)
      SININN_CHECK(a->mask && a->v && a->mask_stride >= a->N && a->v_stride >= a->N && a->clamp > 0.f, "conv: IRN modes need aux (mask), v and clamp");
  }
  // the epilogue moves 16-byte vectors along the channel axis
  SININN_CHECK(aligned16(a->out) && a->out_stride % 4 == 0, "conv: out must be 16-byte aligned with stride %% 4 == 0");
  SININN_CHECK(!a->bias || aligned16(a->bias), "conv: bias must be 16-byte aligned");
  if (couple) {
    SININN_CHECK(aligned16(a->v) && a->v_stride % 4 == 0, "conv: v must be 16-byte aligned with stride %% 4 == 0");
    SININN_CHECK(!a->out2 || (aligned16(a->out2) && a->out2_stride % 4 == 0), "conv: out2 must be 16-byte aligned");
    SININN_CHECK(!a->sbuf || aligned16(a->sbuf), "conv: sbuf must be 16-byte aligned");
  }
  if (a->mode == SININN_CONV_MASK) SININN_CHECK(aligned16(a->mask) && a->mask_stride % 4 == 0, "conv: mask must be 16-byte aligned");
  SININN_CHECK(a->N % 4 == 0 || couple, "conv: N must be a multiple of 4");
  if ((a->mode == SININN_CONV_ADD || a->mode == SININN_CONV_ADD_CBWD_FWD || a->mode == SININN_CONV_ADD_CBWD_INV) && !a->addend_map)
    SININN_CHECK(aligned16(a->addend) && a->addend_stride % 4 == 0, "conv: addend must be 16-byte aligned");
  d.in = a->in; d.in_stride = a->in_stride; d.Cin = a->Cin;
  d.w = a->w; d.bias = a->bias; d.Np = a->Np;
  d.B = a->B; d.H = a->H; d.W = a->W;
  d.out = a->out; d.out_stride = a->out_stride; d.N = a->N; d.out_map = a->out_map;
  d.v = a->v; d.v_stride = a->v_stride; d.out2 = a->out2; d.out2_stride = a->out2_stride;
  d.sbuf = a->sbuf; d.logdet = a->logdet; d.Co = a->Co; d.clamp = a->clamp;
  d.mask = a->mask; d.mask_stride = a->mask_stride;
  d.addend = a->addend; d.addend_stride = a->addend_stride; d.addend_map = a->addend_map;
  d.mode = a->mode;
  d.col_tile = a->col_tile;
  d.stamp = a->stamp;
  d.ablate = g_ablate;
  d.in_chunk = a->in_group_stride > 0 ? a->in_group_stride : 8;
  d.out_gs = a->out_group_stride > 0 ? (size_t)a->out_group_stride : 0;
  d.mask_gs = a->mask_group_stride > 0 ? (size_t)a->mask_group_stride : 0;
  if (d.out_gs || d.mask_gs) {
    // group-major hidden tensors go through the epilogue's full-quad fast path only
    SININN_CHECK(a->winograd && (a->mode == SININN_CONV_RELU || a->mode == SININN_CONV_MASK) && a->N % 64 == 0 && a->N == a->Np,
                 "conv: channel-group-major out / mask needs a Winograd RELU / MASK conv with N == Np, N %% 64 == 0");
    SININN_CHECK(!d.mask_gs || a->mode == SININN_CONV_MASK, "conv: mask_group_stride without MASK mode");
  }
  if (a->in_group_stride > 0) SININN_CHECK(a->winograd && a->in_stride == 8, "conv: the channel-group-major input layout needs the Winograd kernels and in_stride == 8");
  if (a->winograd) {
    // the Winograd kernels stage through raw buffer loads: 32-bit byte offsets inside one image, 32-bit chunk advance
    SININN_CHECK((unsigned long long)a->H * a->W * a->in_stride * 4ull < (1ull << 31),
                 "conv: one image of the input (%d x %d x stride %d floats) exceeds the 2 GB a Winograd block addresses", a->H, a->W, a->in_stride);
    SININN_CHECK((unsigned long long)(a->Cin / 8) * (unsigned long long)d.in_chunk * 4ull < (1ull << 32), "conv: input channel-group stride too large");
  }
  SININN_CHECK(a->mode >= 0 && a->mode <= SININN_CONV_ADD_CBWD_INV, "conv: unknown mode %d", a->mode);
  // channel chunk: the largest of 32/24/16/8 that divides Cin
  int ck = 8;
  for (int c : {32, 24, 16, 8}) if (a->Cin % c == 0) { ck = c; break; }
  if (g_force_ck && a->Cin % g_force_ck == 0 && g_force_ck % 8 == 0 && g_force_ck <= 32) ck = g_force_ck;
  d.CK = ck;
  d.tiles_x = d.tiles_y = 0;
  return 0;
}

int conv_launch(const sininn_conv_args* a, hipStream_t st) {
  SININN_CHECK(a != nullptr, "conv: null args");
  if (a->w_bf16) return conv_bf16_launch(a, st);
  ConvDev d;
  if (int rc = conv_prepare(a, d)) return rc;
  const bool couple = a->mode == SININN_CONV_COUPLE_FWD || a->mode == SININN_CONV_COUPLE_INV;
  if (a->winograd) {
    SININN_CHECK(a->ksize == 3, "conv: the Winograd pack needs ksize 3");
    d.col_tile = (couple && a->col_tile == 32) ? 32 : 16;
    if (couple && d.col_tile == 32) SININN_CHECK(a->Co % 16 == 0, "conv: col_tile 32 needs Co %% 16 == 0");
    return wino_dispatch_k3(d, st, g_force_cfg % 16);   // test hook: see wino_dispatch
  }
  // column-tile width: coupling weights are packed for one specific width (col_tile); other modes take the
  // 32-wide MFMA shape whenever the column count allows (g_force_cfg >= 10 pins the 16-wide kernel for tests)
  const int tile = couple ? (a->col_tile == 32 ? 32 : 16) : ((a->Np % 32 == 0 && g_force_cfg < 10) ? 32 : 16);
  if (couple && tile == 32) SININN_CHECK(a->Co % 16 == 0, "conv: col_tile 32 needs Co %% 16 == 0 (Co=%d)", a->Co);
  const int fc = g_force_cfg % 10;
  if (tile == 32) {
    const int rc = (a->ksize == 3) ? conv32_dispatch_k3(d, st, fc, couple) : conv32_dispatch_k1(d, st, fc, couple);
    if (rc >= 0) return rc;
  }
  if (a->ksize == 3) return conv_dispatch_k3(d, st, fc);
  return conv_dispatch_k1(d, st, fc);
}

void conv_set_test_hooks(int force_cfg, int force_ck) {
  // force_cfg = ablate*1000 + dma*100 + cfg   (cfg >= 10 pins the 16-wide kernel; dma: 1 on, 2 off, 0 default)
  g_force_cfg = force_cfg % 100; g_force_ck = force_ck; g_ablate = force_cfg / 1000;
  const int dma = (force_cfg / 100) % 10;
  if (dma) g_conv_dma = (dma == 1);
}

}  // namespace sininn

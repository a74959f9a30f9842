// Two chained 1x1 convs of a GLOW subnet in ONE launch, the 256-channel hidden tile kept in LDS between them
// (reference: subnet_conv_1x1, archs.py:15-17, inside FrEIA's GLOWCouplingBlock, archs.py:56-64):
//   forward :  h  = relu(x W1^T + b1)          ->  (s,t) = h W2^T + b2   -> affine coupling epilogue
//   backward:  dh = (dr W2) . [h > 0]          ->  dx    = dh W1         -> skip-add / fused coupling-backward epilogue
// The two-kernel path moves the hidden tensor (67 MB at BASELINE configs[1], level 0) through HBM / the Infinity Cache
// twice per pair -- once written by the first conv at the CU's store-drain rate, once re-read by the second with two
// blocks per CU in flight ("1x1 subnets").  Here the first conv's tile stays on chip for the second; it is still
// written to HBM once (training needs h for the ReLU mask / weight gradient, dh for the weight gradient) but that store
// now overlaps the second GEMM, and it is skipped entirely when `first->out` is NULL (no-grad passes).
//
// Block = TH x 16 pixels (P = 16 TH; 64 pixels when the second conv has fewer than four 16-column tiles, else 32 so that two
// blocks fit a CU's LDS and one block's HBM pass overlaps the other's MFMAs), 256 threads.  Both GEMMs run on v_mfma_f32_16x16x4_f32, rows = pixels:
//   stage 1: wave w owns hidden columns [64 w, 64 w + 64); A = the staged input tile (LDS), B = rows of the packed weights
//            [256][K1] read straight from L2 (16 bytes per lane per 16 channels: the k order inside a 16-channel step is
//            permuted identically for A and B, so one float4 feeds four MFMAs);
//   pass   : bias + ReLU (or the ReLU mask read from HBM) over the LDS tile with 16-byte accesses, the same pass writes
//            the tile to HBM;
//   stage 2: the N2 / 16 column tiles are dealt round-robin to the waves, all TH row tiles each; A = the hidden tile (LDS),
//            B = rows of the second conv's packed weights [N2][256] from L2; accumulators -> T[pixel][N2 + 4] in LDS ->
//            the shared epilogue (conv_mfma_impl.h), so every mode of the two-kernel path behaves identically.
#include <cstdlib>
#include "conv_mfma_impl.h"

namespace sininn {

int conv_prepare(const sininn_conv_args* a, ConvDev& d);
int conv_pair_bf16_supported(const sininn_conv_args* f, const sininn_conv_args* s);      // conv_pair_bf16.hip
int conv_pair_bf16_launch(const sininn_conv_args* f, const sininn_conv_args* s, hipStream_t st);

struct PairDev { ConvDev a, b; };

// build tunables (A/B'd with tools/build_variant.sh + SININN_LIB on one box): how many 16-channel steps ahead the weight
// fragments are requested from L2 in stage 1 (ring size - 1) and in stage 2 (row-split / column-split wave layouts)
#ifndef PAIR_RING1
#define PAIR_RING1 2
#endif
#ifndef PAIR_DEPTH_M
#define PAIR_DEPTH_M 2
#endif
#ifndef PAIR_DEPTH_N
#define PAIR_DEPTH_N 1
#endif

constexpr int PK_HID = 256;           // hidden channels (SININN_HIDDEN)
constexpr int PK_HS = PK_HID + 4;     // floats per pixel row of the hidden tile in LDS

template <int TH, int BN2, int HT>
__global__ __launch_bounds__(256) void conv_pair_k1_kernel(PairDev q) {
  constexpr int P = TH * 16, MT = TH;
  constexpr int NT2 = BN2 / 16;
  const ConvDev& pa = q.a;
  const ConvDev& pb = q.b;
  extern __shared__ __attribute__((aligned(16))) float smem_pair[];
  const int K1 = pa.Cin, K1R = (K1 + 15) / 16 * 16, XS = K1R + 4;
  float* const hs = smem_pair;                     // [P][PK_HS]; later T[P][BN2 + 4]
  float* const xs = smem_pair + P * PK_HS;         // [P][XS]

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, kq = lane >> 4;
  int bid = blockIdx.x;
  const int tx = bid % pa.tiles_x; bid /= pa.tiles_x;
  const int ty = bid % pa.tiles_y;
  const int b = bid / pa.tiles_y;
  const int y0 = ty * TH, x0 = tx * 16;

  // phase stamps (diagnostic, second->stamp -> 8 words: input staging, GEMM 1, activation pass, GEMM 2, epilogue, -, total,
  // blocks): thread 0 of every block adds its shader-clock deltas (tools/bench_pair.py --phases)
  const bool stamping = pb.stamp != nullptr && tid == 0;
  unsigned long long tprev = stamping ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long tstart = tprev;
  auto mark = [&](int k) {
    if (stamping) { const unsigned long long t = __builtin_amdgcn_s_memtime(); atomicAdd(pb.stamp + k, t - tprev); tprev = t; }
  };

  // ---- weight fragments of BOTH GEMMs are requested before the input tile is even staged: they depend on nothing this block
  // computes, and every phase of this kernel used to open with an exposed L2 / HBM round trip (phase stamps at level 0: 11.6 k
  // clocks of input staging for one load, 23.6 k for a GEMM whose MFMAs need 2 k)
  // stage 1: raw buffer loads, one lane offset per column tile, the 16-channel step is the scalar offset; the k-quad beyond K1
  // of the last step (K1 % 16 != 0) carries BUF_OOB
  const __amdgpu_buffer_rsrc_t wa_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pa.w), 0, PK_HID * K1 * 4, 0x00020000);
  unsigned woff1[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) woff1[n] = (unsigned)(((wave * 64 + n * 16 + li) * K1 + 4 * kq) * 4);
  const int nsteps = K1R / 16;
  auto load_b = [&](int s, f32x4 (&bf)[4]) {
    const bool live = 16 * s + 4 * kq < K1;         // K1 % 4 == 0 (host check)
#pragma unroll
    for (int n = 0; n < 4; ++n) bf[n] = buf_load4(wa_rs, live ? woff1[n] : BUF_OOB, (pa.ablate & 1) ? 0u : (unsigned)(64 * s));
  };
  // weights requested RING1 - 1 sixteen-channel steps ahead through a register ring (an L2 round trip is longer than the
  // 16 MT MFMAs of a step)
  constexpr int RING1 = PAIR_RING1;
  f32x4 bfr[RING1][4];
#pragma unroll
  for (int s = 0; s < RING1 - 1; ++s)
    if (s < nsteps) load_b(s, bfr[s]);

  // ---- stage-2 wave layout and its first weight fragments (requested here, used after the activation pass) ----------------
  constexpr bool MSPLIT = NT2 < 4;
  // wave grid of stage 2: WM x WN = 4; row-split: one row tile per wave (P = 64: WM = 4) or per wave pair (P = 32: WM = 2,
  // the pair's two waves take alternate column tiles)
  constexpr int WM = MSPLIT ? MT : 1, WN = 4 / WM;
  static_assert(!MSPLIT || MT == 4 || MT == 2, "row-tile split needs 2 or 4 row tiles");
  constexpr int NI = (NT2 + WN - 1) / WN, MI = MSPLIT ? 1 : MT, DEPTH = MSPLIT ? PAIR_DEPTH_M : PAIR_DEPTH_N;
  constexpr int KS2 = (NI * MI <= 4) ? 4 : ((NI * MI <= 8) ? 2 : 1);     // accumulator sets, see stage 1
  auto nt_of = [&](int i) -> int { return (wave / WM) + WN * i; };
  auto mt_of = [&](int m) -> int { return MSPLIT ? (wave % WM) : m; };
  const __amdgpu_buffer_rsrc_t wb_rs = buf_rsrc(pb.w);
  unsigned woff2[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) woff2[i] = nt_of(i) < NT2 ? (unsigned)((((pa.ablate & 1) ? 0 : nt_of(i)) * 16 + li) * PK_HID + 4 * kq) * 4u : BUF_OOB;
  constexpr int NSTEPS = PK_HID / 16;
  f32x4 bf2[DEPTH + 1][NI];
  auto load_b2 = [&](int s, f32x4 (&dst)[NI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i) dst[i] = buf_load4(wb_rs, woff2[i], (pa.ablate & 1) ? 0u : (unsigned)(64 * s));
  };
#pragma unroll
  for (int s = 0; s < DEPTH; ++s) load_b2(s, bf2[s]);

  // ---- stage 0: input tile -> LDS (zero beyond the image and beyond K1) --------------------------------------------
  {
    // raw buffer loads relative to this block's image: a slot outside the image or beyond K1 carries BUF_OOB and reads zeros
    const int q4 = K1R / 4;
    const __amdgpu_buffer_rsrc_t in_rs = buf_rsrc(pa.in + (size_t)b * pa.H * pa.W * pa.in_stride);
    // four slots per thread in flight (K1 = 192 at level 1 is six slots per thread: one at a time was six round trips per block)
    for (int f0 = tid; f0 < P * q4; f0 += 4 * 256) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int f = f0 + 256 * u;
        const int pl = f / q4, c = (f - pl * q4) * 4;
        const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
        const unsigned off = (f < P * q4 && gy < pa.H && gx < pa.W && c < K1) ? (unsigned)(((gy * pa.W + gx) * pa.in_stride + c) * 4) : BUF_OOB;
        v[u] = buf_load4(in_rs, off, 0u);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int f = f0 + 256 * u;
        const int pl = f / q4, c = (f - pl * q4) * 4;
        if (f < P * q4) *reinterpret_cast<f32x4*>(xs + pl * XS + c) = v[u];
      }
    }
  }
  __syncthreads();
  mark(0);

  // ---- stage 1: hidden[P][256] = in[P][K1] . Wa[256][K1]^T ; wave w -> columns 64 w .. 64 w + 63 -----------------------
  {
    // with few tiles per wave an accumulator would come round again after 2-8 MFMAs (40-clock dependent latency against a
    // 32-clock issue interval, plus the waits in between): the k-steps of a 16-channel step alternate between KS accumulator
    // sets that are summed at the end (per-block GEMM time -25 % at level 0)
    constexpr int KS1 = (MT * 4 >= 16) ? 1 : 2;
    f32x4 accs[KS1][MT][4];
#pragma unroll
    for (int k = 0; k < KS1; ++k)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) accs[k][m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < nsteps; s0 += RING1) {
#pragma unroll
      for (int u = 0; u < RING1; ++u) {
        const int s = s0 + u;
        if (s < nsteps) {
          if (s + RING1 - 1 < nsteps) load_b(s + RING1 - 1, bfr[(u + RING1 - 1) % RING1]);
          f32x4 af[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const f32x4*>(xs + (m * 16 + li) * XS + 16 * s + 4 * kq);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int n = 0; n < 4; ++n)
                accs[j % KS1][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][j], bfr[u][n][j], accs[j % KS1][m][n], 0, 0, 0);
        }
      }
    }
    // D[row = 4 kq + r][col = li] -> hs[pixel][column] (raw sums; bias / activation in the pass below)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = accs[0][m][n][r];
#pragma unroll
          for (int k = 1; k < KS1; ++k) v += accs[k][m][n][r];
          hs[(m * 16 + 4 * kq + r) * PK_HS + wave * 64 + n * 16 + li] = v;
        }
  }
  __syncthreads();
  mark(1);

  // ---- pass: bias + ReLU (forward) / ReLU mask (backward) on the tile, 16 bytes per access; the tile goes to HBM ---------
  {
    constexpr int Q = PK_HID / 4;                    // 64 quads per pixel: a thread keeps its quad, 4 pixels per iteration
    const int cq = (tid & 63) * 4;
    f32x4 bq = {0.f, 0.f, 0.f, 0.f};
    if (pa.bias) bq = *reinterpret_cast<const f32x4*>(pa.bias + cq);
    const bool masked = pa.mode == SININN_CONV_MASK;
    // backward: the ReLU masks of all the thread's pixels are requested before the first is used (one global-load latency
    // for the pass instead of one per group of iterations; the stage-1 accumulators are dead, registers are free)
    f32x4 m_all[P / 4];
    if (masked) {
#pragma unroll
      for (int i = 0; i < P / 4; ++i) {
        const int pl = (tid >> 6) + 4 * i;
        const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
        m_all[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (gy < pa.H && gx < pa.W)
          m_all[i] = *reinterpret_cast<const f32x4*>(pa.mask + ((size_t)(b * pa.H + gy) * pa.W + gx) * pa.mask_stride + cq);
      }
    }
#pragma unroll
    for (int i = 0; i < P / 4; ++i) {
      const int pl = (tid >> 6) + 4 * i;
      const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
      const bool inimg = gy < pa.H && gx < pa.W;
      const size_t pix = (size_t)(b * pa.H + gy) * pa.W + gx;
      f32x4 v = *reinterpret_cast<const f32x4*>(hs + pl * PK_HS + cq) + bq;
      if (masked) {
        const f32x4 m = m_all[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = m[j] > 0.f ? v[j] : 0.f;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (!inimg) v = (f32x4){0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(hs + pl * PK_HS + cq) = v;
      if (inimg && pa.out && !(pa.ablate & 2)) *reinterpret_cast<f32x4*>(pa.out + pix * pa.out_stride + cq) = v;
    }
    (void)Q;
  }
  __syncthreads();
  mark(2);

  // ---- stage 2: out[P][BN2] = hidden[P][256] . Wb[BN2][256]^T -------------------------------------------------------------
  // >= 4 column tiles: they are dealt round-robin to the waves (nt = wave, wave + 4, ...), every wave takes all row tiles;
  // fewer (level-0 shapes: N2 = 48 / 32): every wave takes ONE row tile and all column tiles, so no wave idles -- its
  // 4 * NT2 MFMAs per 16-channel step are shorter than an L2 round trip, hence the weights are requested two steps ahead
  f32x4 acc2[KS2][NI][MI];
#pragma unroll
  for (int k = 0; k < KS2; ++k)
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int m = 0; m < MI; ++m) acc2[k][i][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
#pragma unroll
    for (int s = 0; s < NSTEPS; ++s) {
      if (s + DEPTH < NSTEPS) load_b2(s + DEPTH, bf2[(s + DEPTH) % (DEPTH + 1)]);
      f32x4 af[MI];
#pragma unroll
      for (int m = 0; m < MI; ++m) af[m] = *reinterpret_cast<const f32x4*>(hs + (mt_of(m) * 16 + li) * PK_HS + 16 * s + 4 * kq);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int m = 0; m < MI; ++m)
            acc2[j % KS2][i][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][j], bf2[s % (DEPTH + 1)][i][j], acc2[j % KS2][i][m], 0, 0, 0);
    }
  }
  __syncthreads();                                   // every wave is done reading the hidden tile
  mark(3);
  constexpr int TS = BN2 + 4;
  float* const T = smem_pair;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int nt = nt_of(i);
    if (nt < NT2) {
#pragma unroll
      for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc2[0][i][m][r];
#pragma unroll
          for (int k = 1; k < KS2; ++k) v += acc2[k][i][m][r];
          T[(mt_of(m) * 16 + 4 * kq + r) * TS + nt * 16 + li] = v;
        }
    }
  }
  __syncthreads();
  __shared__ float red[4];
  conv_epilogue_tile<TH, BN2, HT, 256>(pb, T, b, y0, x0, 0, tid, red);
  if (stamping) {
    mark(4);
    atomicAdd(pb.stamp + 6, tprev - tstart);
    atomicAdd(pb.stamp + 7, 1ull);
  }
}

template <int TH, int BN2, int HT>
static int pair_launch(PairDev& q, hipStream_t st) {
  constexpr int P = TH * 16;
  q.a.tiles_x = q.b.tiles_x = (q.a.W + 15) / 16;
  q.a.tiles_y = q.b.tiles_y = (q.a.H + TH - 1) / TH;
  const int K1R = (q.a.Cin + 15) / 16 * 16;
  const size_t lds = (size_t)(P * PK_HS + P * (K1R + 4)) * sizeof(float);
  SININN_CHECK(lds <= 160 * 1024 && (size_t)P * (BN2 + 4) <= (size_t)P * PK_HS, "conv_pair: LDS tile too large");
  auto k = conv_pair_k1_kernel<TH, BN2, HT>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("conv_pair: cannot raise LDS limit to %zu", lds); return 1; }
  }
  hipLaunchKernelGGL(k, dim3(q.a.tiles_x * q.a.tiles_y * q.a.B), dim3(256), lds, st, q);
  SININN_LAUNCH_CHECK("conv_pair_k1");
  return 0;
}

void conv_pair_k1_enable(int on);
#ifndef PAIR_TH
#define PAIR_TH 2
#endif
static bool g_pair_enabled = getenv("SININN_PAIR_K1") == nullptr || atoi(getenv("SININN_PAIR_K1")) != 0;   // A/B switch

void conv_pair_k1_enable(int on) { g_pair_enabled = on != 0; }     // test hook: A/B against the two-launch path

// 1 when the pair (first: RELU or MASK into the 256-channel hidden tensor; second: any mode reading it) can run fused
int conv_pair_k1_supported(const sininn_conv_args* f, const sininn_conv_args* s) {
  if (!g_pair_enabled || !f || !s) return 0;
  if (f->w_bf16 || s->w_bf16) return conv_pair_bf16_supported(f, s);          // mixed-precision twin
  if (f->ksize != 1 || s->ksize != 1 || f->w_bf16 || s->w_bf16 || f->winograd || s->winograd) return 0;
  if (f->in_bf16 || f->out_bf16 || f->mask_bf16 || s->in_bf16 || s->out_bf16 || s->mask_bf16) return 0;
  if (f->in_group_stride > 0 || f->out_group_stride > 0 || f->mask_group_stride > 0 || s->in_group_stride > 0 ||
      s->out_group_stride > 0 || s->mask_group_stride > 0) return 0;
  if (!(f->mode == SININN_CONV_RELU || f->mode == SININN_CONV_MASK)) return 0;
  if (f->Np != PK_HID || f->N != PK_HID || s->Cin != PK_HID || s->in_stride != PK_HID) return 0;
  if (f->out && (s->in != f->out || f->out_stride != PK_HID)) return 0;
  if (f->Cin % 8 != 0 || f->Cin > 192) return 0;
  if (f->B != s->B || f->H != s->H || f->W != s->W) return 0;
  if (s->mode == SININN_CONV_IRN_FWD || s->mode == SININN_CONV_IRN_INV || s->mode == SININN_CONV_LRELU) return 0;
  const bool couple = s->mode == SININN_CONV_COUPLE_FWD || s->mode == SININN_CONV_COUPLE_INV;
  if (couple && s->col_tile == 32) return s->Np == 64 || s->Np == 192 || s->Np == 96 || s->Np == 32;
  return s->Np == 16 || s->Np == 32 || s->Np == 48 || s->Np == 64 || s->Np == 96 || s->Np == 192;
}

// Policy of the block executor, not capability: TRAINING forward subnets (hidden tensor stored) with fewer than 32 input
// channels -- level 0 of the headline config -- take the two-launch path: 54.5 us against 64.9 us for the pair there
// (tools/bench_pair.py), 8.56 against 8.60 ms per step in an A/B on one box (SININN_PAIR_FWD_MINK=0 / 32 / 128: 8.60 / 8.56 /
// 8.63 ms).  The no-grad pair (no hidden store), the mixed-precision pair and every backward pair stay fused.
int conv_pair_k1_preferred(const sininn_conv_args* f) {
  static const int fwd_mink = getenv("SININN_PAIR_FWD_MINK") ? atoi(getenv("SININN_PAIR_FWD_MINK")) : 32;
  static const int fwd_mink_bf16 = getenv("SININN_PAIR_FWD_MINK_BF16") ? atoi(getenv("SININN_PAIR_FWD_MINK_BF16")) : 0;   // diagnostic
  return !(f->mode == SININN_CONV_RELU && f->out && f->Cin < (f->w_bf16 ? fwd_mink_bf16 : fwd_mink));
}

int conv_pair_k1_launch(const sininn_conv_args* f, const sininn_conv_args* s, hipStream_t st) {
  SININN_CHECK(conv_pair_k1_supported(f, s), "conv_pair: unsupported pair (check sininn_conv_pair_k1_supported first)");
  if (f->w_bf16) return conv_pair_bf16_launch(f, s, st);
  SININN_CHECK((unsigned long long)f->H * f->W * f->in_stride * 4ull < (1ull << 31),
               "conv_pair: one image of the input exceeds the 2 GB a block addresses (raw buffer staging)");
  PairDev q;
  sininn_conv_args fa = *f;
  alignas(16) float dummy_out[4] = {0.f, 0.f, 0.f, 0.f};   // conv_prepare insists on an output pointer; NULL = "do not store h"
  if (!fa.out) { fa.out = dummy_out; fa.out_stride = PK_HID; }
  if (int rc = conv_prepare(&fa, q.a)) return rc;
  if (!f->out) q.a.out = nullptr;
  sininn_conv_args sa = *s;
  if (!f->out) sa.in = q.a.in;                        // never dereferenced: the second conv reads the LDS tile
  if (int rc = conv_prepare(&sa, q.b)) return rc;
  const bool couple = s->mode == SININN_CONV_COUPLE_FWD || s->mode == SININN_CONV_COUPLE_INV;
  const bool ht16 = couple && s->col_tile == 32;
#define PAIR_CASE(BN) case BN: return ht16 ? pair_launch<PAIR_TH, BN, 16>(q, st) : pair_launch<PAIR_TH, BN, 8>(q, st)
  switch (s->Np) {
    PAIR_CASE(16); PAIR_CASE(32); PAIR_CASE(48); PAIR_CASE(64); PAIR_CASE(96); PAIR_CASE(192);
    default: set_error("conv_pair: unsupported Np=%d", s->Np); return 1;
  }
#undef PAIR_CASE
}

}  // namespace sininn

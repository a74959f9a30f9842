// Mixed-precision twin of conv_pair_k1.hip: the two chained 1x1 convs of a GLOW subnet (subnet_conv_1x1, archs.py:15-17)
// in ONE launch on v_mfma_f32_32x32x16_bf16, the 256-channel hidden tile kept in LDS as bf16 between them.
//   forward :  h  = bf16(relu(x W1^T + b1))   ->  (s,t) = h W2^T + b2  -> fp32 affine coupling epilogue
//   backward:  dh = bf16((dr W2) . [h > 0])   ->  dx    = dh W1        -> fp32 skip-add / fused coupling-backward epilogue
// In the bf16 path these convs are bound by the hidden tensor's HBM round trips (134 MB as bf16 at 512 x 512, level 0):
// written by the first conv, re-read by the second.  Here it is written once (training) or never (first->out == NULL), and
// never re-read.  The values are the same as in the two-launch path: fp32 accumulation, bias / ReLU / mask in fp32, ONE
// rounding to bf16 before the second GEMM.
//
// Block = 4 x 16 pixels (two 32-pixel row tiles), 256 threads.  Stage 1: wave w owns hidden columns [64 w, 64 w + 64) as two
// 32-column tiles x two row tiles; A = the input tile (fp32 in HBM, rounded while it is staged), B = 16-byte rows of the
// packed weights [256][Kp] straight from L2.  A pass over the tile writes it to HBM with 16-byte stores (and applies the
// ReLU mask, read as bf16, in the backward pair).  Stage 2: the (row tile, 32-column tile) pairs are dealt round-robin to the
// waves; accumulators -> T[pixel][N2 + 4] fp32 -> the shared epilogue of every conv kernel (conv_mfma_impl.h).
#include "conv_bf16_types.h"

namespace sininn {

struct PairDevB { ConvDevB a, b; };

constexpr int PB_HID = 256;
constexpr int PB_HSB = PB_HID * 2 + 16;      // bytes per pixel row of the hidden tile: 16 (mod 256) -> conflict-free ds_read_b128

template <int BN2, int HT>
__global__ __launch_bounds__(256, 4) void conv_pair_bf16_kernel(PairDevB q) {
  constexpr int TH = 4, P = 64, MT = 2;
  constexpr int NT2 = (BN2 + 31) / 32, TILES2 = MT * NT2, NI = (TILES2 + 3) / 4, TS = BN2 + 4;
  const ConvDev& pa = q.a.c;
  const ConvDev& pb = q.b.c;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_pb[];
  const int Kp1 = q.a.Kp, XSB = Kp1 * 2 + 16;
  constexpr int HS_BYTES = P * PB_HSB, T_BYTES = P * TS * 4;
  unsigned char* const hs = smem_pb;                                     // [P][PB_HSB] bf16; later T[P][TS] fp32
  unsigned char* const xs = smem_pb + (HS_BYTES > T_BYTES ? HS_BYTES : T_BYTES);   // [P][XSB] bf16

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  int bid = blockIdx.x;
  const int tx = bid % pa.tiles_x; bid /= pa.tiles_x;
  const int ty = bid % pa.tiles_y;
  const int b = bid / pa.tiles_y;
  const int y0 = ty * TH, x0 = tx * 16;

  // ---- stage 0: fp32 input tile -> bf16 in LDS (zero beyond the image and beyond K1) -------------------------------------
  {
    const float* in = static_cast<const float*>(q.a.in);
    const int q4 = Kp1 / 4;
    // eight slots per thread in flight (raw buffer loads: a slot outside the image / beyond Cin reads zeros): K1 = 192 is twelve
    // slots per thread, and one at a time was twelve global-load round trips at the head of every block
    const __amdgpu_buffer_rsrc_t in_rs = buf_rsrc(in + (size_t)b * pa.H * pa.W * pa.in_stride);
    constexpr int NB = 8;
    for (int f0 = tid; f0 < P * q4; f0 += NB * 256) {
      f32x4 v[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int f = f0 + 256 * u;
        const int pl = f / q4, c = (f - pl * q4) * 4;
        const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
        const unsigned off = (f < P * q4 && gy < pa.H && gx < pa.W && c < pa.Cin)            // Cin % 4 == 0 (host check)
                                 ? (unsigned)(((gy * pa.W + gx) * pa.in_stride + c) * 4) : BUF_OOB;
        v[u] = buf_load4(in_rs, off, 0u);
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int f = f0 + 256 * u;
        if (f < P * q4) {
          const int pl = f / q4, c = (f - pl * q4) * 4;
          bf16x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (__bf16)v[u][j];
          *reinterpret_cast<bf16x4*>(xs + pl * XSB + c * 2) = o;
        }
      }
    }
  }
  __syncthreads();

  // ---- stage 1: hidden[P][256] = in[P][Kp1] . Wa[256][Kp1]^T --------------------------------------------------------------
  {
    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
    const __bf16* wrow = q.a.w + (size_t)(wave * 64 + r) * Kp1 + 8 * h;
    const int nsteps = Kp1 / 16;
    constexpr int RING = 4;                                                // weights requested three 16-channel steps ahead
    bf16x8 bfr[RING][2];
    auto load_b = [&](int s, bf16x8 (&dst)[2]) {
#pragma unroll
      for (int n = 0; n < 2; ++n) dst[n] = *reinterpret_cast<const bf16x8*>(wrow + (size_t)n * 32 * Kp1 + 16 * s);
    };
#pragma unroll
    for (int s = 0; s < RING - 1; ++s)
      if (s < nsteps) load_b(s, bfr[s]);
    for (int s0 = 0; s0 < nsteps; s0 += RING) {
#pragma unroll
      for (int u = 0; u < RING; ++u) {
        const int s = s0 + u;
        if (s < nsteps) {
          if (s + RING - 1 < nsteps) load_b(s + RING - 1, bfr[(u + RING - 1) % RING]);
          bf16x8 af[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const bf16x8*>(xs + (m * 32 + r) * XSB + (16 * s + 8 * h) * 2);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bfr[u][n], acc[m][n], 0, 0, 0);
        }
      }
    }
    // D: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5); bias + ReLU in fp32 (forward), one rounding to bf16
    const bool relu = pa.mode == SININN_CONV_RELU;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int col = wave * 64 + n * 32 + r;
      const float bias = pa.bias ? pa.bias[col] : 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
          float v = acc[m][n][e] + bias;
          if (relu) v = fmaxf(v, 0.f);
          *reinterpret_cast<__bf16*>(hs + (m * 32 + row) * PB_HSB + col * 2) = (__bf16)v;
        }
    }
  }
  __syncthreads();

  // ---- pass: (backward) ReLU mask on the tile; the tile goes to HBM with 16-byte stores ------------------------------------
  {
    const bool masked = pa.mode == SININN_CONV_MASK;
    const int c8 = (tid & 31) * 8;
    if (masked || q.a.out_b) {
      // the ReLU masks of all the thread's pixels are requested before the first is used: one global-load latency per pass
      bf16x8 m_all[P / 8];
      if (masked) {
#pragma unroll
        for (int i = 0; i < P / 8; ++i) {
          const int pl = (tid >> 5) + 8 * i;
          const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
          m_all[i] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
          if (gy < pa.H && gx < pa.W)
            m_all[i] = *reinterpret_cast<const bf16x8*>(q.a.mask_b + ((size_t)(b * pa.H + gy) * pa.W + gx) * pa.mask_stride + c8);
        }
      }
#pragma unroll
      for (int i = 0; i < P / 8; ++i) {
        const int pl = (tid >> 5) + 8 * i;
        const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
        const bool inimg = gy < pa.H && gx < pa.W;
        const size_t pix = (size_t)(b * pa.H + gy) * pa.W + gx;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(hs + pl * PB_HSB + c8 * 2);
        if (masked) {
          const bf16x8 m = m_all[i];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = ((float)m[j] > 0.f) ? v[j] : (__bf16)0.f;
          *reinterpret_cast<bf16x8*>(hs + pl * PB_HSB + c8 * 2) = v;
        }
        if (inimg && q.a.out_b) *reinterpret_cast<bf16x8*>(q.a.out_b + pix * pa.out_stride + c8) = v;
      }
    }
  }
  __syncthreads();

  // ---- stage 2: out[P][BN2] = hidden[P][256] . Wb[Np][256]^T ; (row tile, 32-column tile) pairs round-robin over the waves ----
  f32x16 acc2[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;
  {
    constexpr int NSTEPS = PB_HID / 16, RING = 4;
    bf16x8 bfr[RING][NI];
    auto load_b = [&](int s, bf16x8 (&dst)[NI]) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int t = wave + 4 * i, nt = t / MT;
        const int colr = nt * 32 + r;
        bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        dst[i] = (t < TILES2 && colr < pb.Np) ? *reinterpret_cast<const bf16x8*>(q.b.w + (size_t)colr * PB_HID + 16 * s + 8 * h) : z;
      }
    };
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) load_b(s, bfr[s]);
#pragma unroll 1
    for (int s0 = 0; s0 < NSTEPS; s0 += RING) {
#pragma unroll
      for (int u = 0; u < RING; ++u) {
        const int s = s0 + u;
        if (s + RING - 1 < NSTEPS) load_b(s + RING - 1, bfr[(u + RING - 1) % RING]);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int t = wave + 4 * i, mt = t % MT;
          if (t < TILES2) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(hs + (mt * 32 + r) * PB_HSB + (16 * s + 8 * h) * 2);
            acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[u][i], acc2[i], 0, 0, 0);
          }
        }
      }
    }
  }
  __syncthreads();                                   // every wave is done reading the hidden tile
  float* const T = reinterpret_cast<float*>(smem_pb);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int t = wave + 4 * i, mt = t % MT, nt = t / MT;
    const int col = nt * 32 + r;
    if (t < TILES2 && col < BN2) {
#pragma unroll
      for (int e = 0; e < 16; ++e) T[(mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * TS + col] = acc2[i][e];
    }
  }
  __syncthreads();
  __shared__ float red[4];
  conv_epilogue_tile<TH, BN2, HT, 256>(pb, T, b, y0, x0, 0, tid, red);
}

template <int BN2, int HT>
static int pair_bf16_launch(PairDevB& q, hipStream_t st) {
  constexpr int P = 64, TH = 4;
  q.a.c.tiles_x = q.b.c.tiles_x = (q.a.c.W + 15) / 16;
  q.a.c.tiles_y = q.b.c.tiles_y = (q.a.c.H + TH - 1) / TH;
  const size_t hs_bytes = (size_t)P * PB_HSB, t_bytes = (size_t)P * (BN2 + 4) * 4;
  const size_t lds = (hs_bytes > t_bytes ? hs_bytes : t_bytes) + (size_t)P * (q.a.Kp * 2 + 16);
  SININN_CHECK(lds <= 160 * 1024, "conv_pair_bf16: LDS tiles too large");
  auto k = conv_pair_bf16_kernel<BN2, HT>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("conv_pair_bf16: cannot raise LDS limit to %zu", lds); return 1; }
  }
  hipLaunchKernelGGL(k, dim3(q.a.c.tiles_x * q.a.c.tiles_y * q.a.c.B), dim3(256), lds, st, q);
  SININN_LAUNCH_CHECK("conv_pair_bf16");
  return 0;
}

// 1 when the pair can run fused on the bf16 path: first = fp32 input -> bf16 hidden tensor (RELU or MASK), second = bf16
// hidden tensor -> fp32 epilogue (coupling / ADD / ADD_CBWD)
int conv_pair_bf16_supported(const sininn_conv_args* f, const sininn_conv_args* s) {
  if (!f || !s || !f->w_bf16 || !s->w_bf16) return 0;
  if (f->ksize != 1 || s->ksize != 1 || f->winograd || s->winograd) return 0;
  if (f->in_bf16 || !f->out_bf16 || !s->in_bf16 || s->out_bf16) return 0;
  if (f->mode == SININN_CONV_MASK ? !f->mask_bf16 : f->mode != SININN_CONV_RELU) return 0;
  if (f->in_group_stride > 0 || f->out_group_stride > 0 || f->mask_group_stride > 0 || s->in_group_stride > 0 ||
      s->out_group_stride > 0 || s->mask_group_stride > 0) return 0;
  if (f->Np != PB_HID || f->N != PB_HID || s->Cin != PB_HID || s->in_stride != PB_HID) return 0;
  if (f->out && (s->in != f->out || f->out_stride != PB_HID)) return 0;
  if (f->Cin % 8 != 0 || f->Cin > 192) return 0;
  if (f->B != s->B || f->H != s->H || f->W != s->W) return 0;
  const bool couple = s->mode == SININN_CONV_COUPLE_FWD || s->mode == SININN_CONV_COUPLE_INV;
  const bool cbwd = s->mode == SININN_CONV_ADD_CBWD_FWD || s->mode == SININN_CONV_ADD_CBWD_INV;
  if (!(couple || cbwd || s->mode == SININN_CONV_ADD)) return 0;
  if (couple && s->col_tile == 32) return s->Np == 64 || s->Np == 192 || s->Np == 96 || s->Np == 32;
  return s->Np == 16 || s->Np == 32 || s->Np == 48 || s->Np == 64 || s->Np == 96 || s->Np == 192;
}

// conv_sub1_bf16.hip: the persistent forward for the wide subnets (96 -> 256 -> 192), conv2's pack resident in LDS
int conv_sub1_bf16_wide_fwd_supported(const sininn_conv_args* f, const sininn_conv_args* s);
int conv_sub1_bf16_wide_fwd_launch(const sininn_conv_args* f, const sininn_conv_args* s, hipStream_t st);
int conv_sub1_bf16_wide_bwd_supported(const sininn_conv_args* d2, const sininn_conv_args* d1);
int conv_sub1_bf16_wide_bwd_launch(const sininn_conv_args* d2, const sininn_conv_args* d1, hipStream_t st);

int conv_pair_bf16_launch(const sininn_conv_args* f, const sininn_conv_args* s, hipStream_t st) {
  SININN_CHECK(conv_pair_bf16_supported(f, s), "conv_pair_bf16: unsupported pair");
  if (conv_sub1_bf16_wide_fwd_supported(f, s)) return conv_sub1_bf16_wide_fwd_launch(f, s, st);
  if (conv_sub1_bf16_wide_bwd_supported(f, s)) return conv_sub1_bf16_wide_bwd_launch(f, s, st);
  SININN_CHECK((unsigned long long)f->H * f->W * f->in_stride * 4ull < (1ull << 31),
               "conv_pair_bf16: one image of the input exceeds the 2 GB a block addresses (raw buffer staging)");
  PairDevB q;
  sininn_conv_args fa = *f;
  alignas(16) __bf16 dummy_out[8] = {};               // conv_bf16_prepare insists on an output pointer; NULL = "do not store h"
  if (!fa.out) { fa.out = reinterpret_cast<float*>(dummy_out); fa.out_stride = PB_HID; }
  if (int rc = conv_bf16_prepare(&fa, q.a)) return rc;
  if (!f->out) q.a.out_b = nullptr;
  sininn_conv_args sa = *s;
  if (!f->out) sa.in = f->in;                         // never dereferenced: the second conv reads the LDS tile
  if (int rc = conv_bf16_prepare(&sa, q.b)) return rc;
  const bool couple = s->mode == SININN_CONV_COUPLE_FWD || s->mode == SININN_CONV_COUPLE_INV;
  const bool ht16 = couple && s->col_tile == 32;
#define PAIRB_CASE(BN) case BN: return ht16 ? pair_bf16_launch<BN, 16>(q, st) : pair_bf16_launch<BN, 8>(q, st)
  switch (s->Np) {
    PAIRB_CASE(16); PAIRB_CASE(32); PAIRB_CASE(48); PAIRB_CASE(64); PAIRB_CASE(96); PAIRB_CASE(192);
    default: set_error("conv_pair_bf16: unsupported Np=%d", s->Np); return 1;
  }
#undef PAIRB_CASE
}

}  // namespace sininn

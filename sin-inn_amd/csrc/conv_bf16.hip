// bf16 implicit-GEMM convolution on v_mfma_f32_32x32x16_bf16 (fp32 accumulation) for the mixed-precision path of
// BASELINE configs[3] / [4]: the conv subnets (archs.py:11-17) compute in bf16, the invertible flow itself stays fp32.
//
//   * operands: the input is either fp32 (a half of the flow tensor / the coupling-tail gradient: converted to bf16
//     with round-to-nearest-even while it is staged) or bf16 (the hidden tensor h / its gradient dh, which only the
//     executor's own kernels touch and which therefore live in HBM as bf16: half the bytes of the 67..600 MB round
//     trip); weights are packed bf16 [tap][column][K] once per optimiser step; accumulation and every epilogue
//     (bias, ReLU, affine coupling + log-det, ReLU mask, skip-gradient add) run in fp32.
//   * block = 16 x 16 output pixels x 64 columns (8 x 16 for 1x1); a 3x3 wave owns two 32-pixel row tiles x both 32-column
//     tiles (four accumulator tiles of 32 x 32, lane = column, 16 registers = pixel rows): the weights staged per chunk (37 KB,
//     the larger part of the staging traffic) serve 256 pixels and 4 LDS fragment reads feed 4 MFMAs.  (Four row tiles x one
//     column tile per wave -- 5 reads per 4 MFMAs -- measured the same: the loop is not bound by the fragment reads.)
//   * K loop = channel chunks (CK = 32 or 16): the halo tile of a chunk AND the weights of all nine taps of that chunk are
//     staged together (one LDS buffer, two barriers per chunk, 36 MFMAs per wave between them; the next chunk's global
//     loads are in flight under the MFMAs, and the CU's second block computes while this one stages); shifted LDS
//     addresses serve the taps.  (The first version staged one tap per barrier -- 4 MFMAs per wave per barrier, 15 % of
//     the bf16 peak.)  All fragment reads are 16-byte ds_read_b128 of 8 consecutive k: pixel / column stride = 2 CK + 16
//     bytes and an image-row pitch that is a multiple of 256 bytes make every hardware lane group of the read
//     ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS table) hit 16 distinct 4-bank slots.
//   * epilogue: accumulators -> fp32 LDS tile T[pixel][BN + 4] -> the shared float4 epilogue of the fp32 kernels
//     (coupling / add / fused coupling backward), or the bf16-output epilogues here (ReLU -> h, mask -> dh).
#include "conv_bf16_types.h"

namespace sininn {

template <int CK> struct BfGeom {
  static constexpr int PIXB = CK * 2 + 16;                               // bytes per staged pixel / weight column
  static constexpr int pitch(int iw) { return (iw * PIXB + 255) / 256 * 256; }
};

// bf16-output epilogues: T[pixel][BN+4] fp32 -> 8 channels (16 bytes) per store
template <int BN, int NPIX>
__device__ __forceinline__ void epilogue_bf16(const ConvDevB& q, const float* T, int b, int y0, int x0, int n0, int tid) {
  const ConvDev& p = q.c;
  constexpr int TS = BN + 4, Q = BN / 8;
  static_assert(256 % Q == 0, "a thread keeps its column group over the pixel loop");
  constexpr int ITERS = (NPIX * Q + 255) / 256;
  const int q8 = tid % Q;
  const int col = n0 + q8 * 8;
  if (col >= p.N) return;                                    // N % 8 == 0 is checked on the host
  const bool masked = p.mode == SININN_CONV_MASK;
  // loop invariants (bias) hoisted, the per-pixel masks of ALL iterations requested up front: one global-load latency per block
  f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
  if (!masked && p.bias) {
    b0 = *reinterpret_cast<const f32x4*>(p.bias + col);
    b1 = *reinterpret_cast<const f32x4*>(p.bias + col + 4);
  }
  auto load_mask = [&](int it) -> bf16x8 {
    bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!masked) return z;
    const int pl = (tid + it * 256) / Q;
    const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
    if (!(pl < NPIX && gy < p.H && gx < p.W)) return z;
    return *reinterpret_cast<const bf16x8*>(q.mask_b + ((size_t)(b * p.H + gy) * p.W + gx) * p.mask_stride + col);
  };
  bf16x8 mk_all[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) mk_all[it] = load_mask(it);
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const bf16x8 mk = mk_all[it];
    const int pl = (tid + it * 256) / Q;
    const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
    if (pl < NPIX && gy < p.H && gx < p.W) {
      const size_t pix = (size_t)(b * p.H + gy) * p.W + gx;
      f32x4 a = *reinterpret_cast<const f32x4*>(T + pl * TS + q8 * 8);
      f32x4 c = *reinterpret_cast<const f32x4*>(T + pl * TS + q8 * 8 + 4);
      if (!masked) {
        a += b0; c += b1;
        if (p.mode == SININN_CONV_RELU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { a[j] = fmaxf(a[j], 0.f); c[j] = fmaxf(c[j], 0.f); }
        }
      } else {                                             // gradient through the ReLU of h
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a[j] = ((float)mk[j] > 0.f) ? a[j] : 0.f;
          c[j] = ((float)mk[4 + j] > 0.f) ? c[j] : 0.f;
        }
      }
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) { o[j] = (__bf16)a[j]; o[4 + j] = (__bf16)c[j]; }
      *reinterpret_cast<bf16x8*>(q.out_b + pix * p.out_stride + col) = o;
    }
  }
}

template <int KS, int CK, int HT, bool IN_BF16, int BN = 64>
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel(ConvDevB q) {
  const ConvDev& p = q.c;
  // 3x3: 16 x 16 output pixels per block (the nine taps' weights staged per chunk then serve 256 pixels), 4 row tiles of two
  // pixel rows per wave; 1x1: 8 x 16 pixels (HBM-bound on the hidden tensor: more, smaller blocks stream better)
  constexpr int TH = (KS == 3) ? 16 : 8;
  constexpr int HALO = KS / 2, IW = 16 + 2 * HALO, IH = TH + 2 * HALO, NPIX_IN = IH * IW, TAPS = KS * KS;
  static_assert(BN == 64 || (BN == 32 && KS == 3), "32-column blocks: 3x3 convs with at most 32 packed columns (the level-0 data gradient of conv1)");
  constexpr int PIXB = BfGeom<CK>::PIXB, PITCH = BfGeom<CK>::pitch(IW);
  constexpr int IN_BYTES = IH * PITCH, W_TAP = BN * PIXB;
  // staging work items: input = (pixel, 4 fp32 channels -> 8 bytes) or (pixel, 8 bf16 channels -> 16 bytes)
  constexpr int IN_PER_PIX = IN_BF16 ? CK / 8 : CK / 4;
  constexpr int IN_ITEMS = (NPIX_IN * IN_PER_PIX + 255) / 256;
  constexpr int W_PER_COL = CK / 8;                                      // 16-byte pieces per weight column
  constexpr int W_ITEMS = (TAPS * BN * W_PER_COL + 255) / 256;           // all taps of a chunk are staged together
  constexpr bool W_PREFETCH = (KS == 1);                                 // register prefetch of the next chunk's weights

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  unsigned char* const in_lds = smem_b;
  unsigned char* const w_lds = smem_b + IN_BYTES;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  // wave tile: 3x3 -> each wave two row tiles x BOTH 32-column tiles (4 fragment reads per 4 MFMAs instead of 5: the MFMA loop
  // is LDS-bound); 1x1 (four row tiles per block) -> 2 x 2 waves, two row tiles x one column tile each
  constexpr int WN = (KS == 3) ? 1 : 2, WM = 4 / WN, MW = (TH / 2) / WM, NW = (BN / 32) / WN;   // row / column tiles per wave
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  int bid = blockIdx.x;
  const int tx = bid % p.tiles_x; bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * 16;
  const int n0 = blockIdx.y * BN;

  // ---- staging descriptors (kept small: the kernel lives at the 256-VGPR edge) -----------------------------------------
  // input: item f = tid + 256 i -> (pixel f / IN_PER_PIX, part f % IN_PER_PIX); only the global offset is kept, the LDS
  // offset is recomputed when the chunk is stored.  weights: item -> (tap, column, part) is affine in i for a fixed thread.
  auto in_offset = [&](int i) -> int {
    const int f = tid + 256 * i;
    const int pix = f / IN_PER_PIX, part = f - pix * IN_PER_PIX;
    const int py = pix / IW, px = pix - py * IW;
    const int gy = y0 + py - HALO, gx = x0 + px - HALO;
    const bool inimg = pix < NPIX_IN && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    return inimg ? (((b * p.H + gy) * p.W + gx) * p.in_stride + part * (IN_BF16 ? 8 : 4)) : -1;          // elements
  };
  // bf16 inputs (K = 256: eight chunks) keep the offsets in registers; fp32 inputs (few chunks, twice the items) recompute them
  int in_goff[IN_BF16 ? IN_ITEMS : 1];
  if constexpr (IN_BF16) {
#pragma unroll
    for (int i = 0; i < IN_ITEMS; ++i) in_goff[i] = in_offset(i);
  }
  // weight items: (tap, column, 16-byte part) flattened as ct = tap * BN + column; item i of a thread is ct0 + i * CT_STEP
  constexpr int CT_STEP = 256 / W_PER_COL;                               // 128 (CK 16), 64 (CK 32), 16 (CK 128)
  static_assert(CT_STEP % BN == 0 || (TAPS == 1 && BN % CT_STEP == 0), "weight staging: affine item -> (tap, column)");
  const int w_ct0 = tid / W_PER_COL, w_part = tid % W_PER_COL;
  const int w_tap0 = w_ct0 / BN, w_col0 = w_ct0 % BN;
  constexpr int W_TAP_STEP = (CT_STEP >= BN) ? CT_STEP / BN : 0;         // taps advanced per item
  constexpr int W_COL_STEP = (CT_STEP >= BN) ? 0 : CT_STEP;              // columns advanced per item (1x1 only)
  const int w_l0 = w_tap0 * W_TAP + w_col0 * PIXB + w_part * 16;
  const int w_g0 = (w_tap0 * p.Np + n0 + w_col0) * q.Kp + w_part * 8;   // elements
  const int w_gstep = (W_TAP_STEP * p.Np + W_COL_STEP) * q.Kp;
  constexpr int W_LSTEP = W_TAP_STEP * W_TAP + W_COL_STEP * PIXB;
  auto w_live = [&](int i) -> bool {
    return (w_tap0 + i * W_TAP_STEP) < TAPS && (w_col0 + i * W_COL_STEP) < BN && (n0 + w_col0 + i * W_COL_STEP) < p.Np;
  };
  const int nchunks = q.Kp / CK;

  typedef typename std::conditional<IN_BF16, bf16x8, bf16x4>::type in_reg_t;     // fp32 inputs are rounded at load time
  in_reg_t in_reg[IN_ITEMS];
  bf16x8 w_reg[W_ITEMS];
  auto load_chunk = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < IN_ITEMS; ++i) {
      const int part = (tid + 256 * i) % IN_PER_PIX;
      if constexpr (IN_BF16) {
        bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        const bool live = in_goff[i] >= 0 && chunk * CK + part * 8 < p.Cin;     // Cin % 8 == 0 (host check)
        in_reg[i] = live ? *reinterpret_cast<const bf16x8*>(static_cast<const __bf16*>(q.in) + in_goff[i] + chunk * CK) : z;
      }
    }
    if constexpr (W_PREFETCH) {
#pragma unroll
      for (int i = 0; i < W_ITEMS; ++i) {
        bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        w_reg[i] = w_live(i) ? *reinterpret_cast<const bf16x8*>(q.w + w_g0 + i * w_gstep + chunk * CK) : z;
      }
    }
  };
  auto store_chunk = [&](int chunk) {
    // fp32 inputs (cond, dr: at most six chunks) are not prefetched across the MFMA loop -- eleven float4 per thread would
    // push the kernel into scratch; they are loaded, rounded and stored here while the CU's other block computes.  All
    // global loads of the chunk are ISSUED first, inputs then weights (returns are in order: the inputs can be converted
    // and stored while the weights are still in flight); waiting for each group in turn cost two exposed round trips per
    // chunk, 46-61 % of the block time of the small-K convs that produce the hidden tensor (tools/bf16_phases.py --hidden-out)
    f32x4 in_raw[IN_BF16 ? 1 : IN_ITEMS];
    if constexpr (!IN_BF16) {
#pragma unroll
      for (int i = 0; i < IN_ITEMS; ++i) {
        const int part = (tid + 256 * i) % IN_PER_PIX;
        in_raw[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int go = in_offset(i);
        const bool live = go >= 0 && chunk * CK + part * 4 < p.Cin;             // Cin % 4 == 0 (host check)
        if (live) in_raw[i] = *reinterpret_cast<const f32x4*>(static_cast<const float*>(q.in) + go + chunk * CK);
      }
    }
    if constexpr (!W_PREFETCH) {
      // 3x3: the nine taps' weights (36 VGPRs per thread) are NOT held across the MFMA loop -- with them the kernel sat at
      // 254 VGPRs and the compiler serialised every fragment read behind its MFMA (ds_read -> lgkmcnt(0) -> v_mfma chains).
      // They are L2-resident; the load latency is covered by the CU's second block.
#pragma unroll
      for (int i = 0; i < W_ITEMS; ++i) {
        bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        w_reg[i] = w_live(i) ? *reinterpret_cast<const bf16x8*>(q.w + w_g0 + i * w_gstep + chunk * CK) : z;
      }
    }
    if constexpr (!IN_BF16) {
#pragma unroll
      for (int i = 0; i < IN_ITEMS; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) in_reg[i][j] = (__bf16)in_raw[i][j];
    }
#pragma unroll
    for (int i = 0; i < IN_ITEMS; ++i) {
      const int f = tid + 256 * i;
      const int pix = f / IN_PER_PIX, part = f - pix * IN_PER_PIX;
      const int py = pix / IW, px = pix - py * IW;
      if (pix < NPIX_IN)
        *reinterpret_cast<in_reg_t*>(in_lds + py * PITCH + px * PIXB + part * (IN_BF16 ? 16 : 8)) = in_reg[i];
    }
#pragma unroll
    for (int i = 0; i < W_ITEMS; ++i)
      if ((w_tap0 + i * W_TAP_STEP) < TAPS && (w_col0 + i * W_COL_STEP) < BN)
        *reinterpret_cast<bf16x8*>(w_lds + w_l0 + i * W_LSTEP) = w_reg[i];
  };

  f32x16 acc[MW][NW];
#pragma unroll
  for (int m = 0; m < MW; ++m)
#pragma unroll
    for (int n = 0; n < NW; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

  // A fragment of lane (r, h) for row tile m: pixel (row 8 wm + 2 m + (r >> 4), column r & 15), channels 8 h .. 8 h + 7
  const int a_off0 = (2 * MW * wm + (r >> 4)) * PITCH + (r & 15) * PIXB + h * 16;
  const int b_off = (wn * NW * 32 + r) * PIXB + h * 16;

  // phase stamps (diagnostic, sininn_conv_args.stamp -> 8 words: barrier A, staging, barrier B, load issue, MFMA loop,
  // epilogue, total, blocks): wave 0 / lane 0 of every block adds its shader-clock deltas
  const bool stamping = p.stamp != nullptr && tid == 0;
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long tprev = stamping ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long tstart = tprev;
  auto mark = [&](int k) {
    if (stamping) { const unsigned long long t = __builtin_amdgcn_s_memtime(); ph[k] += t - tprev; tprev = t; }
  };

  load_chunk(0);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    mark(3);
    __syncthreads();                       // the previous chunk's fragment reads are done (single LDS buffer; the second
    mark(0);
    store_chunk(chunk);                    // block of the CU computes meanwhile)
    mark(1);
    __syncthreads();
    mark(2);
    if (chunk + 1 < nchunks) load_chunk(chunk + 1);          // in flight under the MFMAs below
    mark(3);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const unsigned char* A = in_lds + (tap / KS) * PITCH + (tap % KS) * PIXB;
      const unsigned char* B = w_lds + tap * W_TAP + b_off;
#pragma unroll
      for (int ks = 0; ks < CK / 16; ++ks) {
        bf16x8 bfr[NW], af[MW];
#pragma unroll
        for (int n = 0; n < NW; ++n) bfr[n] = *reinterpret_cast<const bf16x8*>(B + n * 32 * PIXB + ks * 32);
#pragma unroll
        for (int m = 0; m < MW; ++m) af[m] = *reinterpret_cast<const bf16x8*>(A + a_off0 + m * 2 * PITCH + ks * 32);
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
          for (int n = 0; n < NW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bfr[n], acc[m][n], 0, 0, 0);
      }
    }
    mark(4);
  }
  __syncthreads();

  // ---- accumulators -> T[pixel][BN + 4]; D layout: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) ----------
  constexpr int TS = BN + 4;
  float* const T = reinterpret_cast<float*>(smem_b);
#pragma unroll
  for (int m = 0; m < MW; ++m)
#pragma unroll
    for (int n = 0; n < NW; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        const int pl = wm * (MW * 32) + m * 32 + row;
        T[pl * TS + (wn * NW + n) * 32 + r] = acc[m][n][e];
      }
  __syncthreads();
  if (q.out_bf16) {
    epilogue_bf16<BN, TH * 16>(q, T, b, y0, x0, n0, tid);
  } else {
    __shared__ float red[4];
    conv_epilogue_tile<TH, BN, HT, 256>(p, T, b, y0, x0, n0, tid, red);
  }
  if (stamping) {
    mark(5);
#pragma unroll
    for (int k = 0; k < 6; ++k) atomicAdd(p.stamp + k, ph[k]);
    atomicAdd(p.stamp + 6, tprev - tstart);
    atomicAdd(p.stamp + 7, 1ull);
  }
}

template <int KS, int CK, int HT, bool IN_BF16, int BN = 64>
static int launch_one(const ConvDevB& q, hipStream_t st) {
  constexpr int TH = (KS == 3) ? 16 : 8;
  constexpr int HALO = KS / 2, IW = 16 + 2 * HALO, IH = TH + 2 * HALO, TAPS = KS * KS;
  constexpr size_t lds_main = (size_t)IH * BfGeom<CK>::pitch(IW) + (size_t)TAPS * BN * BfGeom<CK>::PIXB;
  constexpr size_t lds_epi = (size_t)TH * 16 * (BN + 4) * sizeof(float);
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  static_assert(lds <= 80 * 1024, "two blocks per CU must fit");
  static_assert(CK != 16 || KS != 3 || lds_main <= 53 * 1024, "CK 16: three blocks per CU by the main-loop footprint");
  auto k = conv_bf16_kernel<KS, CK, HT, IN_BF16, BN>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("conv_bf16: cannot raise LDS limit to %zu", lds); return 1; }
  dim3 grid(q.c.tiles_x * q.c.tiles_y * q.c.B, (q.c.Np + BN - 1) / BN);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, q);
  SININN_LAUNCH_CHECK("conv_bf16");
  return 0;
}

template <int KS, int CK>
static int launch_ht(const ConvDevB& q, hipStream_t st) {
  const bool ht16 = q.c.col_tile == 32;            // coupling interleave half-width (irrelevant for the other modes)
  if constexpr (KS == 3 && CK == 32) {
    // at most 32 packed columns (level-0 data gradient of conv1, N = 24): 32-column blocks -- half the weight staging and
    // MFMAs of a 64-column block that is 62 % padding, 46 KB of LDS (three blocks per CU)
    if (q.in_bf16 && q.c.Np <= 32 && !ht16) return launch_one<KS, CK, 8, true, 32>(q, st);
  }
  if (q.in_bf16) return ht16 ? launch_one<KS, CK, 16, true>(q, st) : launch_one<KS, CK, 8, true>(q, st);
  // fp32 inputs only feed the bf16-output convs of the path (cond -> h, dr -> dh): the coupling half-width is irrelevant
  SININN_CHECK(q.out_bf16, "conv_bf16: an fp32-input conv must have a bf16 output");
  return launch_one<KS, CK, 8, false>(q, st);
}

int g_bf16_force_ck16 = 0;     // diagnostic (SININN_BF16_CK16=1): 16-channel chunks everywhere (3 blocks per CU instead of 2)

template <int KS>
static int launch_ks(const ConvDevB& q, hipStream_t st) {
  static const bool env16 = getenv("SININN_BF16_CK16") != nullptr;
  if constexpr (KS == 1) {
    // 1x1 convs are HBM-bound on the hidden tensor: 128-channel chunks (two for K = 256) keep 35 KB of loads in flight
    // per block and 16 MFMAs per wave between barriers instead of 4
    if (q.Kp % 128 == 0 && !env16) return launch_ht<1, 128>(q, st);
  }
  return (q.Kp % 32 == 0 && !env16 && !g_bf16_force_ck16) ? launch_ht<KS, 32>(q, st) : launch_ht<KS, 16>(q, st);
}

// a->w: bf16 pack [taps][Np][Kp], Kp = Cin rounded up to 16; a->in: fp32 or (in_bf16) bf16; a->out: fp32 or (out_bf16) bf16
// argument validation + device-side descriptor (shared by conv_bf16_launch and the fused 1x1 pair, conv_pair_bf16.hip)
int conv_bf16_prepare(const sininn_conv_args* a, ConvDevB& q) {
  SININN_CHECK(a->ksize == 1 || a->ksize == 3, "conv_bf16: ksize %d not in {1,3}", a->ksize);
  SININN_CHECK(a->Cin > 0 && a->Cin % 8 == 0, "conv_bf16: Cin=%d must be a positive multiple of 8", a->Cin);
  SININN_CHECK(a->Np > 0 && a->Np % 16 == 0, "conv_bf16: Np=%d must be a positive multiple of 16", a->Np);
  SININN_CHECK(a->B > 0 && a->H > 0 && a->W > 0, "conv_bf16: bad image shape");
  SININN_CHECK((long)a->B * a->H * a->W * (long)(a->in_stride > a->out_stride ? a->in_stride : a->out_stride) < (1l << 31),
               "conv_bf16: tensor too large for 32-bit pixel offsets");
  SININN_CHECK(a->in && a->w && a->out && aligned16(a->in) && aligned16(a->w) && aligned16(a->out), "conv_bf16: null / unaligned tensor");
  SININN_CHECK(a->in_stride >= a->Cin && a->in_stride % (a->in_bf16 ? 8 : 4) == 0, "conv_bf16: bad input stride %d", a->in_stride);
  const bool couple = a->mode == SININN_CONV_COUPLE_FWD || a->mode == SININN_CONV_COUPLE_INV;
  const bool cbwd = a->mode == SININN_CONV_ADD_CBWD_FWD || a->mode == SININN_CONV_ADD_CBWD_INV;
  if (a->out_bf16) {
    SININN_CHECK(a->mode == SININN_CONV_RELU || a->mode == SININN_CONV_LINEAR || a->mode == SININN_CONV_MASK,
                 "conv_bf16: bf16 output supports RELU / LINEAR / MASK only");
    SININN_CHECK(a->N > 0 && a->N % 8 == 0 && a->N <= a->Np && a->out_stride >= a->N && a->out_stride % 8 == 0, "conv_bf16: bad N / out_stride");
    if (a->mode == SININN_CONV_MASK) SININN_CHECK(a->mask && a->mask_bf16 && a->mask_stride % 8 == 0 && aligned16(a->mask), "conv_bf16: MASK needs a bf16 mask");
    if (a->mode == SININN_CONV_RELU) SININN_CHECK(a->bias != nullptr, "conv_bf16: RELU mode needs bias");
  } else {
    SININN_CHECK(couple || a->mode == SININN_CONV_ADD || cbwd || a->mode == SININN_CONV_LINEAR || a->mode == SININN_CONV_RELU,
                 "conv_bf16: fp32 output supports COUPLE / ADD / ADD_CBWD / LINEAR / RELU");
    SININN_CHECK(a->out_stride % 4 == 0, "conv_bf16: out_stride %% 4");
    if (couple) {
      SININN_CHECK(a->Co > 0 && a->Co % 8 == 0 && a->Np == 2 * a->Co && a->v && a->v_stride >= a->Co && a->clamp > 0.f && a->out_stride >= a->Co,
                   "conv_bf16: coupling needs Np == 2*Co, v, clamp");
      SININN_CHECK(aligned16(a->v) && a->v_stride % 4 == 0 && (!a->out2 || (aligned16(a->out2) && a->out2_stride % 4 == 0)) && (!a->sbuf || aligned16(a->sbuf)),
                   "conv_bf16: coupling operands must be 16-byte aligned");
      SININN_CHECK(a->col_tile == 16 || (a->col_tile == 32 && a->Co % 16 == 0), "conv_bf16: col_tile must be 16 or 32 (Co %% 16 == 0)");
    } else {
      SININN_CHECK(a->N > 0 && a->N % 4 == 0 && a->N <= a->Np && a->out_stride >= a->N, "conv_bf16: bad N");
      if (a->mode == SININN_CONV_ADD || cbwd) SININN_CHECK(a->addend != nullptr, "conv_bf16: ADD mode needs addend");
      if ((a->mode == SININN_CONV_ADD || cbwd) && !a->addend_map) SININN_CHECK(aligned16(a->addend) && a->addend_stride % 4 == 0, "conv_bf16: addend alignment");
      if (cbwd) SININN_CHECK(a->v && a->sbuf && a->out2 && a->Co == a->N && a->out_stride >= 2 * a->Co && a->clamp > 0.f, "conv_bf16: ADD_CBWD needs v, sbuf, out2");
    }
  }
  SININN_CHECK(!a->bias || aligned16(a->bias), "conv_bf16: bias must be 16-byte aligned");
  q = ConvDevB{};
  ConvDev& d = q.c;
  d.in = nullptr; d.in_stride = a->in_stride; d.Cin = a->Cin;
  d.w = nullptr; d.bias = a->bias; d.Np = a->Np;
  d.B = a->B; d.H = a->H; d.W = a->W;
  d.out = a->out_bf16 ? nullptr : a->out; d.out_stride = a->out_stride; d.N = a->N; d.out_map = a->out_map;
  d.v = a->v; d.v_stride = a->v_stride; d.out2 = a->out2; d.out2_stride = a->out2_stride;
  d.sbuf = a->sbuf; d.logdet = a->logdet; d.Co = a->Co; d.clamp = a->clamp;
  d.mask = nullptr; d.mask_stride = a->mask_stride;
  d.addend = a->addend; d.addend_stride = a->addend_stride; d.addend_map = a->addend_map;
  d.mode = a->mode; d.col_tile = couple ? a->col_tile : 16; d.stamp = a->stamp; d.ablate = 0; d.CK = 0; d.in_chunk = 8; d.out_gs = 0; d.mask_gs = 0;
  const int th = a->ksize == 3 ? 16 : 8;
  d.tiles_x = (a->W + 15) / 16; d.tiles_y = (a->H + th - 1) / th;
  q.in = a->in; q.w = reinterpret_cast<const __bf16*>(a->w);
  q.out_b = a->out_bf16 ? reinterpret_cast<__bf16*>(a->out) : nullptr;
  q.mask_b = reinterpret_cast<const __bf16*>(a->mask);
  q.Kp = (a->Cin + 15) / 16 * 16;
  q.in_bf16 = a->in_bf16; q.out_bf16 = a->out_bf16;
  return 0;
}

int conv3_smallk_bf16_supported(const sininn_conv_args* a);      // conv3_smallk_bf16.hip: Cin <= 32 -> 256, ReLU, bf16 out
int conv3_smallk_bf16_launch(const sininn_conv_args* a, hipStream_t st);

int conv_bf16_launch(const sininn_conv_args* a, hipStream_t st) {
  if (conv3_smallk_bf16_supported(a)) return conv3_smallk_bf16_launch(a, st);
  ConvDevB q;
  if (int rc = conv_bf16_prepare(a, q)) return rc;
  return a->ksize == 3 ? launch_ks<3>(q, st) : launch_ks<1>(q, st);
}

// ------------------------------------------------------------------------------------------------
// bf16 weight packs (one launch per conv):  wb_fwd [taps][Np][Kp]  (row q = output channel colmap[q], k = input channel,
// zero beyond Cin), b_fwd [Np] fp32 packed bias, wb_dgrad [taps][Cdp][Kd] (row c = input channel of the conv, k = output
// channel n, flipped taps: the B operand of the data-gradient conv; Kd = N rounded up to 16).
// ------------------------------------------------------------------------------------------------
__global__ void pack_bf16_kernel(const float* __restrict__ w, const float* __restrict__ bias, int N, int Cin, int taps,
                                 const int* __restrict__ colmap, int Np, int Kp, __bf16* __restrict__ wf,
                                 float* __restrict__ bf, int Cdp, int Kd, __bf16* __restrict__ wd) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int nf = wf ? taps * Np * Kp : 0;
  const int nd = wd ? taps * Cdp * Kd : 0;
  if (idx < nf) {
    const int c = idx % Kp, qq = (idx / Kp) % Np, t = idx / (Kp * Np);
    const int n = colmap ? colmap[qq] : qq;
    wf[idx] = (__bf16)((n >= 0 && n < N && c < Cin) ? w[((size_t)n * Cin + c) * taps + t] : 0.f);
  } else if (idx < nf + nd) {
    const int k = idx - nf;
    const int n = k % Kd, c = (k / Kd) % Cdp, t = k / (Kd * Cdp);
    wd[k] = (__bf16)((c < Cin && n < N) ? w[((size_t)n * Cin + c) * taps + (taps - 1 - t)] : 0.f);
  }
  if (bf && idx < Np) {
    const int n = colmap ? colmap[idx] : idx;
    bf[idx] = (bias && n >= 0 && n < N) ? bias[n] : 0.f;
  }
}

int pack_bf16_launch(const float* w, const float* bias, int N, int Cin, int ksize, const int* colmap, int Np, void* wb_fwd,
                     float* b_fwd, int Cdp, void* wb_dgrad, hipStream_t st) {
  SININN_CHECK(w != nullptr && N > 0 && Cin > 0 && (ksize == 1 || ksize == 3), "pack_bf16: bad arguments");
  SININN_CHECK(!wb_fwd || Np >= 1, "pack_bf16: bad Np");
  SININN_CHECK(!wb_dgrad || Cdp >= Cin, "pack_bf16: Cdp < Cin");
  const int taps = ksize * ksize;
  const int Kp = (Cin + 15) / 16 * 16, Kd = (N + 15) / 16 * 16;
  int total = (wb_fwd ? taps * Np * Kp : 0) + (wb_dgrad ? taps * Cdp * Kd : 0);
  if (total < Np) total = Np;
  hipLaunchKernelGGL(pack_bf16_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, bias, N, Cin, taps, colmap, Np, Kp,
                     static_cast<__bf16*>(wb_fwd), b_fwd, Cdp, Kd, static_cast<__bf16*>(wb_dgrad));
  SININN_LAUNCH_CHECK("pack_bf16");
  return 0;
}

}  // namespace sininn

// Mixed-precision twins of the persistent fused 1x1-subnet kernels of conv_sub1.hip (reference: subnet_conv_1x1, archs.py:15-17,
// inside FrEIA's GLOWCouplingBlock, archs.py:56-64): bf16 operands on v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16, fp32
// accumulation, bias / ReLU / mask / coupling arithmetic / gradients in fp32 -- the arithmetic of conv_pair_bf16.hip and the
// bf16 weight-gradient kernel, without their HBM traffic.  On the mixed-precision path the 1x1 subnets are bound by the hidden
// tensors' round trips (134 MB each as bf16 at BASELINE configs[3], level 0): h written by the forward pass, re-read as the
// ReLU mask and by conv2's weight gradient, dh written and re-read by conv1's weight gradient -- 670 MB per half-coupling
// against ~100 MB of x / dr / dx / side inputs, at 0.04 of the bf16 matrix pipe.  Here h and dh never leave the chip.
//
// Backward, per 64-pixel tile (block = 512 threads = 8 waves, persistent, wave w owns hidden columns [32 w, 32 w + 32)):
//   stage 0  x / dr tiles (fp32 in HBM, requested a tile ahead) -> bf16 in LDS, each in two images: [pixel][channel] (row reads:
//            A operand of stage R / stage 2) and [channel][pixel] (A operand of the weight-gradient stages)
//   stage R  h = relu(x W1^T + b1): two 32 x 32 accumulator tiles per wave; column on the lane, pixels in the registers
//   stage W2 dW2[n][c] += sum_p dr[p][n] h[p][c]: the accumulator tile of stage R, rounded to bf16, IS the B operand (a product
//            that sums over the tile's row index takes it with no lane movement and no LDS); A = dr^T from the transposed
//            image, read in the k order the accumulator registers dictate; accumulators live in registers for the whole block
//   stage 2  dh = (dr W2) . [h > 0]: same tile shape as stage R, so the mask is applied register by register
//   stage W1 dW1^T[k][c] += sum_p [x | 1][p][k] dh[p][c]: as stage W2 (db1 rides on a row of ones of the transposed x image)
//   stage 3  dx = dh W1 sums over the hidden index, i.e. over lanes and waves: dh goes through LDS once ([pixel][256] bf16,
//            two lanes' values merged into one 4-byte write), each wave takes one 16 x 16 tile of dx over K = 256 on
//            v_mfma_f32_16x16x32_bf16 with W1's data-gradient pack resident in LDS
//   epilogue the quad epilogue of conv_sub1_bwd_kernel (ADD / ADD_CBWD_*, side inputs requested mid-tile)
// One slab of partial gradients per block in the layout of conv_sub1.hip: sub1_reduce_kernel sums them.
//
// Kernels of this file (all persistent, 512 threads, one block per CU):
//   conv_sub1b_fwd_kernel / conv_sub1b_bwd_kernel      level-0 shapes ((Cin, 2 Co) in {(8, 16), (16, 32), (24, 48)}), 64-pixel tiles:
//                                                      forward without storing h; the whole backward (above)
//   conv_sub1b_wide_fwd_kernel                         level 1 (96 -> 256 -> 192), 32-pixel tiles, conv2's pack resident in LDS
//   conv_sub1b_wide_bwd_kernel (+ wide_reduce_kernel)  level 1: both data gradients (+ conv1's weight gradient, WG1)
//   conv_sub1b_wide_wg2_kernel (+ wide_reduce2_kernel) level 1: conv2's weight gradient, h read from HBM into operand registers
#include <cstdlib>
#include "conv_bf16_types.h"
#include "conv_sub1_types.h"

namespace sininn {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct Sub1DevB {
  ConvDev r, a, b;       // as Sub1Dev: fp32 tensors and the epilogue descriptor; r.w / a.w / b.w are not used
  const __bf16* w1f;     // conv1 forward pack [256][K1R]
  const __bf16* w2;      // backward: conv2 data-gradient pack [256][K2]; forward: conv2 forward pack [N2][256]
  const __bf16* w1d;     // conv1 data-gradient pack [K1R][256]
  float* slab;
  int ntiles, no_dx;
};

template <int K1, int K2>
struct Sub1BShape {
  using SH = Sub1Shape<K1, K2>;
  static constexpr int K1R = SH::K1R, NS1 = K1R / 16, NS2 = K2 / 16, NT2M = (K2 + 31) / 32;
  static constexpr int XSB = K1R * 2 + 16;          // bytes per pixel row of the x image (16 mod 32: conflict-free 16-byte row reads)
  static constexpr int DSB = K2 * 2 + 16;           // ... of the dr image
  static constexpr int TRB = S1_P * 2 + 8;          // bytes per channel row of a transposed image (8 mod 256: conflict-free 8-byte reads)
  static constexpr int HSB = S1_HID * 2 + 16;       // bytes per row of a [.][256] bf16 image
  static constexpr int XT_ROWS = 32, DT_ROWS = 32 * NT2M;
  static constexpr int TS = K1R + 4;
  static constexpr int X_BYTES = S1_P * XSB, XT_BYTES = XT_ROWS * TRB, D_BYTES = S1_P * DSB, DT_BYTES = DT_ROWS * TRB;
  static constexpr int STAGE_BYTES = X_BYTES + XT_BYTES + D_BYTES + DT_BYTES;     // the four images of one tile
  static constexpr int DH_OFF = 2 * STAGE_BYTES, DH_BYTES = S1_P * HSB;
  static constexpr int WD_OFF = DH_OFF + DH_BYTES, WD_BYTES = K1R * HSB;
  static constexpr int T_OFF = WD_OFF + WD_BYTES, T_BYTES = S1_P * TS * 4;
  static constexpr size_t LDS = T_OFF + T_BYTES;
  static_assert(STAGE_BYTES % 16 == 0 && XT_BYTES % 16 == 0 && DT_BYTES % 16 == 0, "conv_sub1_bf16: image alignment");
};

// accumulator tile (32 x 32, column on the lane) -> the two 16-row operand fragments of a following MFMA that sums over the
// tile's rows: element j of lane half h of fragment s is row 16 s + 8 (j >> 2) + 4 h + (j & 3)
__device__ __forceinline__ void acc_to_frags(const f32x16& v, bf16x8 (&f)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) f[s][j] = (__bf16)v[8 * s + j];
}

// [channel][pixel] image -> the A fragment matching acc_to_frags' k order: pixels p0 .. p0 + 3 and p0 + 8 .. p0 + 11
__device__ __forceinline__ bf16x8 read_tr_frag(const unsigned char* row_px) {
  const bf16x4 lo = *reinterpret_cast<const bf16x4*>(row_px);
  const bf16x4 hi = *reinterpret_cast<const bf16x4*>(row_px + 16);
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// fragments of a 32 x 32 tile (columns cw .. cw + 31 of pixels pbase .. pbase + 31) -> the [pixel][256] image.  A fragment
// dword holds two consecutive pixels of the lane's column; lanes 2 i / 2 i + 1 exchange one half each (DPP quad_perm [1,0,3,2])
// so that every lane writes one dword: even lanes (pixel, columns c, c + 1), odd lanes (pixel + 1, columns c - 1, c)
__device__ __forceinline__ void frags_to_image(const bf16x8 (&f)[2], unsigned char* img, int row_bytes, int pbase, int cw, int r, int hh) {
  const bool odd = (r & 1) != 0;
  const unsigned sel = odd ? 0x03020706u : 0x05040100u;
  unsigned char* const base = img + (pbase + 4 * hh + (odd ? 1 : 0)) * row_bytes + (cw + (r & ~1)) * 2;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const u32x4 own = __builtin_bit_cast(u32x4, f[s]);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)own[d], 0xB1, 0xF, 0xF, true);
      const unsigned out = __builtin_amdgcn_perm(nb, own[d], sel);
      const int p = 16 * s + 8 * (d >> 1) + 2 * (d & 1);
      *reinterpret_cast<unsigned*>(base + p * row_bytes) = out;
    }
  }
}

template <int K1, int K2>
__global__ __launch_bounds__(S1_NTHR) void conv_sub1b_bwd_kernel(Sub1DevB q) {
  using SH = Sub1Shape<K1, K2>;
  using SB = Sub1BShape<K1, K2>;
  constexpr int P = S1_P, NTHR = S1_NTHR, K1R = SB::K1R, NS1 = SB::NS1, NS2 = SB::NS2, NT2M = SB::NT2M;
  constexpr int XSB = SB::XSB, DSB = SB::DSB, TRB = SB::TRB, HSB = SB::HSB, TS = SB::TS, NT3 = K1R / 16;
  static_assert(K1 % 8 == 0 && K1 <= 24 && K2 % 16 == 0 && K2 <= 48, "conv_sub1b_bwd: shape");
  const ConvDev& pr = q.r;
  const ConvDev& pa = q.a;
  const ConvDev& pb = q.b;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_s1b[];
  unsigned char* const dhs = smem_s1b + SB::DH_OFF;               // [P][HSB]: dh as bf16
  unsigned char* const wd = smem_s1b + SB::WD_OFF;                // [K1R][HSB]: W1 data-gradient pack
  float* const T = reinterpret_cast<float*>(smem_s1b + SB::T_OFF);   // [P][TS]

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, hh = lane >> 5;
  const int cw = wave * 32;

  // ---- once per block: zero the staging images (pad columns / rows are never written again), W1's data-gradient pack -> LDS,
  // the wave's weight fragments -> registers ---------------------------------------------------------------------------------
  for (int f = tid; f < 2 * SB::STAGE_BYTES / 16; f += NTHR) reinterpret_cast<u32x4*>(smem_s1b)[f] = (u32x4){0u, 0u, 0u, 0u};
  if (!q.no_dx) {
    for (int f = tid; f < K1R * (S1_HID / 8); f += NTHR) {
      const int n = f / (S1_HID / 8), c8 = f - n * (S1_HID / 8);
      *reinterpret_cast<u32x4*>(wd + n * HSB + c8 * 16) = *reinterpret_cast<const u32x4*>(q.w1d + (size_t)n * S1_HID + c8 * 8);
    }
  }
  const float b1v = pr.bias ? pr.bias[cw + r] : 0.f;
  bf16x8 w1f[NS1], w2f[NS2];                        // B operands: W[column cw + r][16 s + 8 hh .. + 7]
#pragma unroll
  for (int s = 0; s < NS1; ++s) w1f[s] = *reinterpret_cast<const bf16x8*>(q.w1f + (size_t)(cw + r) * K1R + 16 * s + 8 * hh);
#pragma unroll
  for (int s = 0; s < NS2; ++s) w2f[s] = *reinterpret_cast<const bf16x8*>(q.w2 + (size_t)(cw + r) * K2 + 16 * s + 8 * hh);

  f32x16 accW2[NT2M], accW1;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    accW1[e] = 0.f;
#pragma unroll
    for (int t = 0; t < NT2M; ++t) accW2[t][e] = 0.f;
  }

  // ---- staging slots of a thread (the same for every tile) and the global loads of a tile: conv_sub1_bwd_kernel's ----------
  constexpr int QX = K1 / 4, QD = K2 / 4;
  constexpr int FX = (P * QX + NTHR - 1) / NTHR, FD = (P * QD + NTHR - 1) / NTHR;
  f32x4 accb2[FD];                                  // db2 in fp32: a staging slot is the same (pixel, channel quad) of every tile
#pragma unroll
  for (int u = 0; u < FD; ++u) accb2[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int tiles_img = pr.tiles_x * pr.tiles_y;
  auto issue_tile = [&](int tile, f32x4 (&vx)[FX], f32x4 (&vd)[FD]) {
    const bool live = tile < q.ntiles;
    const int b = live ? tile / tiles_img : 0;
    const int trem = tile - b * tiles_img;
    const int ty = trem / pr.tiles_x, tx = trem - ty * pr.tiles_x;
    const int y0 = ty * 4, x0 = tx * 16;
    const size_t img = (size_t)b * pr.H * pr.W;
    const __amdgpu_buffer_rsrc_t x_rs = buf_rsrc(pr.in + img * pr.in_stride);
    const __amdgpu_buffer_rsrc_t d_rs = buf_rsrc(pa.in + img * pa.in_stride);
#pragma unroll
    for (int u = 0; u < FX; ++u) {
      const int f = tid + NTHR * u;
      const int pl = f / QX, c = (f - pl * QX) * 4;
      const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
      const unsigned off = (live && f < P * QX && gy < pr.H && gx < pr.W) ? (unsigned)(((gy * pr.W + gx) * pr.in_stride + c) * 4) : BUF_OOB;
      vx[u] = buf_load4(x_rs, off, 0u);
    }
#pragma unroll
    for (int u = 0; u < FD; ++u) {
      const int f = tid + NTHR * u;
      const int pl = f / QD, c = (f - pl * QD) * 4;
      const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
      const unsigned off = (live && f < P * QD && gy < pa.H && gx < pa.W) ? (unsigned)(((gy * pa.W + gx) * pa.in_stride + c) * 4) : BUF_OOB;
      vd[u] = buf_load4(d_rs, off, 0u);
    }
  };
  // regs -> the four bf16 images of a staged tile (+ the row of ones of the transposed x image: 1 for pixels inside the image)
  auto store_tile = [&](int tile, int buf, const f32x4 (&vx)[FX], const f32x4 (&vd)[FD]) {
    unsigned char* const xs = smem_s1b + buf * SB::STAGE_BYTES;
    unsigned char* const xt = xs + SB::X_BYTES;
    unsigned char* const ds = xt + SB::XT_BYTES;
    unsigned char* const dt = ds + SB::D_BYTES;
#pragma unroll
    for (int u = 0; u < FX; ++u) {
      const int f = tid + NTHR * u;
      const int pl = f / QX, c = (f - pl * QX) * 4;
      if (f < P * QX) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (__bf16)vx[u][j];
        *reinterpret_cast<bf16x4*>(xs + pl * XSB + c * 2) = o;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<__bf16*>(xt + (c + j) * TRB + pl * 2) = o[j];
      }
    }
#pragma unroll
    for (int u = 0; u < FD; ++u) {
      const int f = tid + NTHR * u;
      const int pl = f / QD, c = (f - pl * QD) * 4;
      if (f < P * QD) {
        accb2[u] += vd[u];                            // pixels outside the image were loaded as zeros
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (__bf16)vd[u][j];
        *reinterpret_cast<bf16x4*>(ds + pl * DSB + c * 2) = o;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<__bf16*>(dt + (c + j) * TRB + pl * 2) = o[j];
      }
    }
    if (tid < P) {
      const int b = tile / tiles_img;
      const int trem = tile - b * tiles_img;
      const int ty = trem / pr.tiles_x, tx = trem - ty * pr.tiles_x;
      const int gy = ty * 4 + (tid >> 4), gx = tx * 16 + (tid & 15);
      *reinterpret_cast<__bf16*>(xt + K1 * TRB + tid * 2) = (__bf16)((gy < pr.H && gx < pr.W) ? 1.f : 0.f);
    }
  };

  // ---- the epilogue of stage 3 on one quad per thread: conv_sub1_bwd_kernel's --------------------------------------------------
  const int e_pl = tid >> 3, e_q = tid & 7, e_col = 4 * e_q;
  const int emode = pb.mode;
  const bool e_cbwd = emode == SININN_CONV_ADD_CBWD_FWD || emode == SININN_CONV_ADD_CBWD_INV;
  const bool e_fast = !q.no_dx;
  int amap[4] = {e_col, e_col + 1, e_col + 2, e_col + 3};
  f32x4 e_bq = {0.f, 0.f, 0.f, 0.f};
  if (e_fast && e_col < pb.N) {
    if (pb.addend_map) {
#pragma unroll
      for (int j = 0; j < 4; ++j) amap[j] = pb.addend_map[e_col + j];
    }
    if (pb.bias) e_bq = *reinterpret_cast<const f32x4*>(pb.bias + e_col);
  }

  // side inputs of a tile's epilogue quad (raw buffer loads relative to the image: a quad outside the image / beyond N / of a tile
  // beyond the last carries BUF_OOB and reads zeros; no branch around a load)
  struct Side { f32x4 ad, u, s; };
  auto load_side = [&](int tile, Side& sd) {
    const bool live = tile < q.ntiles;
    const int b = live ? tile / tiles_img : 0;
    const int trem = tile - b * tiles_img;
    const int ty = trem / pr.tiles_x, tx = trem - ty * pr.tiles_x;
    const int gy = ty * 4 + (e_pl >> 4), gx = tx * 16 + (e_pl & 15);
    const bool ok = live && e_fast && e_col < pb.N && gy < pb.H && gx < pb.W;
    const size_t img = (size_t)b * pb.H * pb.W;
    const unsigned ip = (unsigned)(gy * pb.W + gx);
    const __amdgpu_buffer_rsrc_t ad_rs = buf_rsrc(pb.addend + img * pb.addend_stride);
    if (pb.addend_map) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        sd.ad[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ad_rs, (int)(ok ? (ip * pb.addend_stride + amap[j]) * 4u : BUF_OOB), 0, 0));
    } else {
      sd.ad = buf_load4(ad_rs, ok ? (ip * pb.addend_stride + e_col) * 4u : BUF_OOB, 0u);
    }
    const bool cb = ok && e_cbwd;
    sd.u = buf_load4(buf_rsrc((e_cbwd ? pb.v : pb.addend) + img * (e_cbwd ? pb.v_stride : 0)), cb ? (ip * pb.v_stride + e_col) * 4u : BUF_OOB, 0u);
    sd.s = buf_load4(buf_rsrc((e_cbwd ? pb.sbuf : pb.addend) + img * (e_cbwd ? pb.Co : 0)), cb ? (ip * pb.Co + e_col) * 4u : BUF_OOB, 0u);
  };
  const int G = gridDim.x;

  // One tile.  On entry LDS buffer `buf` holds the tile's images, `vxn` / `vdn` the x / dr of tile + G and `sd` the side inputs of
  // this tile (all requested two tiles ago); on exit buffer buf ^ 1 holds tile + G, `vxn` / `vdn` are in flight for tile + 3 G and
  // `sd` for tile + 2 G: a tile is ~1.5 us of work and a global round trip ~2 us, so every global operand is two tiles ahead
  auto body = [&](int tile, int buf, f32x4 (&vxn)[FX], f32x4 (&vdn)[FD], Side& sd) {
    const int b = tile / tiles_img;
    const int trem = tile - b * tiles_img;
    const int ty = trem / pr.tiles_x, tx = trem - ty * pr.tiles_x;
    const int y0 = ty * 4, x0 = tx * 16;
    const unsigned char* const xs = smem_s1b + buf * SB::STAGE_BYTES;
    const unsigned char* const xt = xs + SB::X_BYTES;
    const unsigned char* const ds = xt + SB::XT_BYTES;
    const unsigned char* const dt = ds + SB::D_BYTES;

    bf16x8 hf[2][2];                                 // h, then dh, of the wave's 64 x 32 tile as operand fragments [row tile][k-step]
    // ---- stage R + stage W2 ---------------------------------------------------------------------------------------------------
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int s = 0; s < NS1; ++s) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(xs + (32 * m + r) * XSB + (16 * s + 8 * hh) * 2);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, w1f[s], acc, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = fmaxf(acc[e] + b1v, 0.f);
      acc_to_frags(acc, hf[m]);
    }
#pragma unroll
    for (int t = 0; t < NT2M; ++t)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8 af = read_tr_frag(dt + (32 * t + r) * TRB + (32 * m + 16 * s + 4 * hh) * 2);
          accW2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, hf[m][s], accW2[t], 0, 0, 0);
        }

    // ---- stage 2: dh = (dr W2) . [h > 0], register by register ------------------------------------------------------------------
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int s = 0; s < NS2; ++s) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(ds + (32 * m + r) * DSB + (16 * s + 8 * hh) * 2);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, w2f[s], acc, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = ((float)hf[m][e >> 3][e & 7] > 0.f) ? acc[e] : 0.f;
      acc_to_frags(acc, hf[m]);
    }
    // ---- stage W1: dW1^T[k][c] += sum_p [x | 1][p][k] dh[p][c] ---------------------------------------------------------------------
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 af = read_tr_frag(xt + r * TRB + (32 * m + 16 * s + 4 * hh) * 2);
        accW1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, hf[m][s], accW1, 0, 0, 0);
      }
    // ---- dh -> LDS for stage 3 (the previous tile's stage 3 is behind barrier E) ---------------------------------------------------
    if (!q.no_dx) {
#pragma unroll
      for (int m = 0; m < 2; ++m) frags_to_image(hf[m], dhs, HSB, 32 * m, cw, r, hh);
    }

    if (tile + G < q.ntiles) store_tile(tile + G, buf ^ 1, vxn, vdn);
    issue_tile(tile + 3 * G, vxn, vdn);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                 // (C) the whole dh tile is in LDS (and the next tile's images)

    // ---- stage 3: dx tile = dh W1 (K = 256): wave -> 16-pixel tile wave % 4, 16-column tile wave / 4 ----------------------------
    if (!q.no_dx) {
      const int pt = wave & 3, nt = wave >> 2;
      const int row = lane & 15, kg = lane >> 4;
      if (nt < NT3) {
        f32x4 acc3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S1_HID / 32; ++s) {
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(dhs + (16 * pt + row) * HSB + (32 * s + 8 * kg) * 2);
          const bf16x8 bf = *reinterpret_cast<const bf16x8*>(wd + (16 * nt + row) * HSB + (32 * s + 8 * kg) * 2);
          acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc3, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) T[(16 * pt + 4 * kg + e) * TS + 16 * nt + row] = acc3[e];
      }
    }
    __syncthreads();                                 // (E) the dx tile is in T; every wave is done with dhs and this tile's images
    const int e_gy = y0 + (e_pl >> 4), e_gx = x0 + (e_pl & 15);
    const bool e_live = e_fast && e_col < pb.N && e_gy < pb.H && e_gx < pb.W;
    const size_t e_pix = (size_t)b * pb.H * pb.W + (unsigned)(e_gy * pb.W + e_gx);
    asm volatile("" :: "v"(sd.ad), "v"(sd.u), "v"(sd.s));   // waited for by every wave, before any store of this tile is issued
    const f32x4 e_ad = sd.ad, e_u = sd.u, e_s = sd.s;
    if (!q.no_dx) {
      if (e_live) {
        f32x4 val = *reinterpret_cast<const f32x4*>(T + e_pl * TS + e_col);
        val += e_bq;
        if (pb.addend_map) {
#pragma unroll
          for (int j = 0; j < 4; ++j) val[j] += e_ad[j];
        } else {
          val += e_ad;
        }
        if (!e_cbwd) {
          *reinterpret_cast<f32x4*>(pb.out + e_pix * pb.out_stride + e_col) = val;
        } else {
          const float gl = pb.logdet ? pb.logdet[b] : 0.f;
          f32x4 o_a, o_b, o_c;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float g = val[j], u = e_u[j], sv = e_s[j];
            const float L = glow_log_e(sv, pb.clamp), dL = glow_dlog_e(sv, pb.clamp);
            const float e = expf(L);
            if (emode == SININN_CONV_ADD_CBWD_FWD) { o_c[j] = g * e; o_b[j] = g; o_a[j] = (g * u * e + gl) * dL; }
            else { o_c[j] = g / e; o_b[j] = -o_c[j]; o_a[j] = -(g * u + gl) * dL; }
          }
          *reinterpret_cast<f32x4*>(pb.out + e_pix * pb.out_stride + e_col) = o_a;
          *reinterpret_cast<f32x4*>(pb.out + e_pix * pb.out_stride + pb.Co + e_col) = o_b;
          *reinterpret_cast<f32x4*>(pb.out2 + e_pix * pb.out2_stride + e_col) = o_c;
        }
      }
    }
    load_side(tile + 2 * G, sd);
  };

  f32x4 vxa[FX], vda[FD], vxb[FX], vdb[FD];
  Side sda, sdb;
  issue_tile(blockIdx.x, vxa, vda);
  __syncthreads();                                   // the zero fill is complete before the first tile lands on it
  store_tile(blockIdx.x, 0, vxa, vda);
  issue_tile(blockIdx.x + G, vxa, vda);
  issue_tile(blockIdx.x + 2 * G, vxb, vdb);
  load_side(blockIdx.x, sda);
  load_side(blockIdx.x + G, sdb);
  __syncthreads();
  for (int tile = blockIdx.x; tile < q.ntiles; tile += 2 * G) {
    body(tile, 0, vxa, vda, sda);
    if (tile + G < q.ntiles) body(tile + G, 1, vxb, vdb, sdb);
  }

  // ---- the block's partial gradients -> its slab (layout of conv_sub1_bwd_kernel) ------------------------------------------------
  float* const slab = q.slab + (size_t)blockIdx.x * SH::SLAB;
#pragma unroll
  for (int t = 0; t < NT2M; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int n = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * hh;
      if (n < K2) slab[n * S1_HID + cw + r] = accW2[t][e];
    }
  float* const slab1 = slab + K2 * S1_HID;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int k = (e & 3) + 8 * (e >> 2) + 4 * hh;
    if (k < SH::W1S) slab1[k * S1_HID + cw + r] = accW1[e];
  }
  // db2[n] = sum over the 64 staging slots (pixels of a tile) of channel n, in a fixed order
  __syncthreads();
  float* const red = reinterpret_cast<float*>(dhs);  // [P * QD] float4 (12 KB of the dh image's 33 KB)
#pragma unroll
  for (int u = 0; u < FD; ++u) {
    const int f = tid + NTHR * u;
    if (f < P * QD) *reinterpret_cast<f32x4*>(red + 4 * f) = accb2[u];
  }
  __syncthreads();
  if (tid < 64) {
    float v = 0.f;
    if (tid < K2) {
      float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int pl = 0; pl < P; pl += 4)
#pragma unroll
        for (int i = 0; i < 4; ++i) part[i] += red[(pl + i) * K2 + tid];
      v = (part[0] + part[1]) + (part[2] + part[3]);
    }
    slab[K2 * S1_HID + S1_HID * SH::W1S + tid] = v;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Forward: h = relu(x W1^T + b1) -> (s | t) = h W2^T + b2 -> coupling + log-det, persistent; the structure of
// conv_sub1_fwd_kernel.  Stage 1 as stage R above; the h tile goes to LDS as bf16 ([pixel][256]); stage 2 on
// v_mfma_f32_16x16x32_bf16: wave (g, m) = K half g of the 16-pixel tile m, all column tiles, W2's pack resident in LDS; the two
// partial tiles are summed by the coupling epilogue.
template <int K1, int N2, int HT>
__global__ __launch_bounds__(S1_NTHR) void conv_sub1b_fwd_kernel(Sub1DevB q) {
  constexpr int P = S1_P, NTHR = S1_NTHR;
  constexpr int K1R = (K1 + 15) / 16 * 16, NS1 = K1R / 16, XSB = K1R * 2 + 16, HSB = S1_HID * 2 + 16, NU2 = N2 / 16, TS = N2 + 4;
  static_assert(K1 % 8 == 0 && K1 <= 24 && N2 % 16 == 0 && N2 <= 48, "conv_sub1b_fwd: shape");
  const ConvDev& pr = q.r;
  const ConvDev& pb = q.b;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_s1bf[];
  unsigned char* const xs0 = smem_s1bf;                           // 2 x [P][XSB]
  unsigned char* const hs = xs0 + 2 * P * XSB;                    // [P][HSB]
  unsigned char* const w2s = hs + P * HSB;                        // [N2][HSB]
  float* const T0 = reinterpret_cast<float*>(w2s + N2 * HSB);     // 2 x [P][TS]
  float* const T1 = T0 + P * TS;
  __shared__ float ldw[2][NTHR / 64];
  int ld_b[2] = {0, 0};
  int ld_pending = -1;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, hh = lane >> 5;
  const int cw = wave * 32;

  for (int f = tid; f < 2 * P * XSB / 16; f += NTHR) reinterpret_cast<u32x4*>(xs0)[f] = (u32x4){0u, 0u, 0u, 0u};
  for (int f = tid; f < N2 * (S1_HID / 8); f += NTHR) {
    const int n = f / (S1_HID / 8), c8 = f - n * (S1_HID / 8);
    *reinterpret_cast<u32x4*>(w2s + n * HSB + c8 * 16) = *reinterpret_cast<const u32x4*>(q.w2 + (size_t)n * S1_HID + c8 * 8);
  }
  const float b1v = pr.bias ? pr.bias[cw + r] : 0.f;
  bf16x8 w1f[NS1];
#pragma unroll
  for (int s = 0; s < NS1; ++s) w1f[s] = *reinterpret_cast<const bf16x8*>(q.w1f + (size_t)(cw + r) * K1R + 16 * s + 8 * hh);

  // the coupling epilogue on one channel quad per thread: conv_sub1_fwd_kernel's
  constexpr int CO = N2 / 2, NQ = CO / 4;
  const int e_pl = tid / NQ, e_cl = 4 * (tid - e_pl * NQ);
  const bool e_thread = tid < P * NQ;
  const int e_tcol = (e_cl / HT) * (2 * HT) + (e_cl % HT);
  const bool e_inv = pb.mode == SININN_CONV_COUPLE_INV;
  f32x4 e_bs = {0.f, 0.f, 0.f, 0.f}, e_bt = e_bs;
  int e_omap[4] = {e_cl, e_cl + 1, e_cl + 2, e_cl + 3};
  if (e_thread) {
    if (pb.bias) {
      e_bs = *reinterpret_cast<const f32x4*>(pb.bias + e_tcol);
      e_bt = *reinterpret_cast<const f32x4*>(pb.bias + e_tcol + HT);
    }
    if (pb.out_map) {
#pragma unroll
      for (int j = 0; j < 4; ++j) e_omap[j] = pb.out_map[e_cl + j];
    }
  }

  constexpr int QX = K1 / 4, FX = (P * QX + NTHR - 1) / NTHR;
  const int tiles_img = pr.tiles_x * pr.tiles_y;
  auto issue_tile = [&](int tile, f32x4 (&vx)[FX]) {
    const bool live = tile < q.ntiles;
    const int b = live ? tile / tiles_img : 0;
    const int trem = tile - b * tiles_img;
    const int ty = trem / pr.tiles_x, tx = trem - ty * pr.tiles_x;
    const __amdgpu_buffer_rsrc_t x_rs = buf_rsrc(pr.in + (size_t)b * pr.H * pr.W * pr.in_stride);
#pragma unroll
    for (int u = 0; u < FX; ++u) {
      const int f = tid + NTHR * u;
      const int pl = f / QX, c = (f - pl * QX) * 4;
      const int gy = ty * 4 + (pl >> 4), gx = tx * 16 + (pl & 15);
      const unsigned off = (live && f < P * QX && gy < pr.H && gx < pr.W) ? (unsigned)(((gy * pr.W + gx) * pr.in_stride + c) * 4) : BUF_OOB;
      vx[u] = buf_load4(x_rs, off, 0u);
    }
  };
  auto store_tile = [&](int buf, const f32x4 (&vx)[FX]) {
    unsigned char* const xs = xs0 + buf * (P * XSB);
#pragma unroll
    for (int u = 0; u < FX; ++u) {
      const int f = tid + NTHR * u;
      const int pl = f / QX, c = (f - pl * QX) * 4;
      if (f < P * QX) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (__bf16)vx[u][j];
        *reinterpret_cast<bf16x4*>(xs + pl * XSB + c * 2) = o;
      }
    }
  };

  // v of a tile's epilogue quad (raw buffer load relative to the image: a quad outside the image / a tile beyond the last reads zeros)
  auto load_v = [&](int tile) -> f32x4 {
    const bool live = tile < q.ntiles;
    const int b = live ? tile / tiles_img : 0;
    const int trem = tile - b * tiles_img;
    const int ty = trem / pr.tiles_x, tx = trem - ty * pr.tiles_x;
    const int gy = ty * 4 + (e_pl >> 4), gx = tx * 16 + (e_pl & 15);
    const bool ok = live && e_thread && gy < pb.H && gx < pb.W;
    return buf_load4(buf_rsrc(pb.v + (size_t)b * pb.H * pb.W * pb.v_stride), ok ? ((unsigned)(gy * pb.W + gx) * pb.v_stride + e_cl) * 4u : BUF_OOB, 0u);
  };
  const int G = gridDim.x;
  const int g2 = wave >> 2, m2 = wave & 3;          // stage 2: K half, 16-pixel tile
  const int row = lane & 15, kg = lane >> 4;

  // One tile.  On entry LDS buffer `buf` holds the tile's x image, `vxn` the x of tile + G (requested two tiles ago) and `ev` the v
  // of this tile (requested two tiles ago); on exit buffer buf ^ 1 holds tile + G, `vxn` is in flight for tile + 3 G, `ev` for
  // tile + 2 G.  A tile is ~1 us of work and a global round trip ~2 us: every global operand is requested two tiles ahead (one tile
  // ahead, the wait for v was half of the tile time)
  auto body = [&](int tile, int buf, f32x4 (&vxn)[FX], f32x4& ev) {
    const int b = tile / tiles_img;
    const int trem = tile - b * tiles_img;
    const int ty = trem / pr.tiles_x, tx = trem - ty * pr.tiles_x;
    const int y0 = ty * 4, x0 = tx * 16;
    const unsigned char* const xs = xs0 + buf * (P * XSB);

    // ---- stage 1 (== stage R of the backward kernel: bitwise the h it recomputes) -> hs ------------------------------------------
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int s = 0; s < NS1; ++s) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(xs + (32 * m + r) * XSB + (16 * s + 8 * hh) * 2);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, w1f[s], acc, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = fmaxf(acc[e] + b1v, 0.f);
      bf16x8 hf[2];
      acc_to_frags(acc, hf);
      frags_to_image(hf, hs, HSB, 32 * m, cw, r, hh);
    }
    if (tile + G < q.ntiles) store_tile(buf ^ 1, vxn);
    issue_tile(tile + 3 * G, vxn);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                 // (B) the whole h tile is in LDS (and the next tile's x)
    if (pb.logdet) {
      if (tid == 0 && ld_pending >= 0) {
        const float* w8 = ldw[ld_pending];
        atomicAdd(pb.logdet + ld_b[ld_pending], ((w8[0] + w8[1]) + (w8[2] + w8[3])) + ((w8[4] + w8[5]) + (w8[6] + w8[7])));
      }
      ld_pending = buf;
    }

    // ---- stage 2: this wave's K half of out[16-pixel tile m2][all N2 columns] ---------------------------------------------------
    f32x4 acc[NU2];
#pragma unroll
    for (int u = 0; u < NU2; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S1_HID / 64; ++s) {
      const int k0 = 128 * g2 + 32 * s + 8 * kg;
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(hs + (16 * m2 + row) * HSB + k0 * 2);
#pragma unroll
      for (int u = 0; u < NU2; ++u) {
        const bf16x8 bf = *reinterpret_cast<const bf16x8*>(w2s + (16 * u + row) * HSB + k0 * 2);
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[u], 0, 0, 0);
      }
    }
    {
      float* const T = g2 ? T1 : T0;
#pragma unroll
      for (int u = 0; u < NU2; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) T[(16 * m2 + 4 * kg + e) * TS + 16 * u + row] = acc[u][e];
    }
    __syncthreads();                                 // (E) both partial tiles are in LDS; every wave is done with hs
    const int e_gy = y0 + (e_pl >> 4), e_gx = x0 + (e_pl & 15);
    const bool e_live = e_thread && e_gy < pb.H && e_gx < pb.W;
    const size_t e_img = (size_t)b * pb.H * pb.W;
    const unsigned e_ip = (unsigned)(e_gy * pb.W + e_gx);
    asm volatile("" :: "v"(ev));                     // waited for by every wave, before any store of this tile is issued
    const f32x4 e_v = ev;
    float ld_acc = 0.f;
    if (e_live) {
      f32x4 s4 = *reinterpret_cast<const f32x4*>(T0 + e_pl * TS + e_tcol) + e_bs;
      f32x4 t4 = *reinterpret_cast<const f32x4*>(T0 + e_pl * TS + e_tcol + HT) + e_bt;
      s4 += *reinterpret_cast<const f32x4*>(T1 + e_pl * TS + e_tcol);
      t4 += *reinterpret_cast<const f32x4*>(T1 + e_pl * TS + e_tcol + HT);
      const size_t pix = e_img + e_ip;
      f32x4 y4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float L = glow_log_e(s4[j], pb.clamp);
        const float e = expf(L);
        if (!e_inv) { y4[j] = e * e_v[j] + t4[j]; ld_acc += L; }
        else { y4[j] = (e_v[j] - t4[j]) / e; ld_acc -= L; }
      }
      if (pb.out_map) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pb.out[pix * pb.out_stride + e_omap[j]] = y4[j];
      } else {
        *reinterpret_cast<f32x4*>(pb.out + pix * pb.out_stride + e_cl) = y4;
      }
      if (pb.out2) *reinterpret_cast<f32x4*>(pb.out2 + pix * pb.out2_stride + e_cl) = y4;
      if (pb.sbuf) *reinterpret_cast<f32x4*>(pb.sbuf + pix * pb.Co + e_cl) = s4;
    }
    ev = load_v(tile + 2 * G);
    if (pb.logdet) {
      const float wsum = wave_sum(ld_acc);
      if (lane == 0) ldw[buf][wave] = wsum;
      ld_b[buf] = b;
    }
  };

  f32x4 vxa[FX], vxb[FX], eva, evb;
  issue_tile(blockIdx.x, vxa);
  __syncthreads();                                   // zero fill of the pad columns before the first tile lands
  store_tile(0, vxa);
  issue_tile(blockIdx.x + G, vxa);
  issue_tile(blockIdx.x + 2 * G, vxb);
  eva = load_v(blockIdx.x);
  evb = load_v(blockIdx.x + G);
  __syncthreads();
  for (int tile = blockIdx.x; tile < q.ntiles; tile += 2 * G) {
    body(tile, 0, vxa, eva);
    if (tile + G < q.ntiles) body(tile + G, 1, vxb, evb);
  }
  if (pb.logdet) {
    __syncthreads();
    if (tid == 0 && ld_pending >= 0) {
      const float* w8 = ldw[ld_pending];
      atomicAdd(pb.logdet + ld_b[ld_pending], ((w8[0] + w8[1]) + (w8[2] + w8[3])) + ((w8[4] + w8[5]) + (w8[6] + w8[7])));
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Forward of the WIDE 1x1 subnets (level 1: 96 -> 256 -> 2 x 96), persistent: the structure of conv_sub1b_fwd_kernel on 32-pixel
// tiles (2 x 16), so that conv2's whole pack (192 x 256 bf16 = 99 KB) stays in LDS for the lifetime of the block beside the h
// tile, the x double buffer and the output tile (153 KB in all).  The pair kernel it replaces (conv_pair_bf16_kernel, 1024 - 3600
// blocks of 64 pixels) walks K = 256 of the second GEMM through a chain of L2 round trips for its weight fragments in every
// block: 76 us at BASELINE configs[3] for 4 us of matrix work.  Stage 1 = one 32 x 32 accumulator tile per wave over K1 / 16
// steps (bitwise the pair kernel's h); h goes to LDS as bf16 and -- training passes: the backward of a wide subnet is not fused,
// it reads h -- to HBM straight from the registers (4-byte stores, 64-byte runs); stage 2: 2 x N2 / 16 tiles of 16 x 16 on
// v_mfma_f32_16x16x32_bf16, three per wave, K = 256 in one piece; the coupling epilogue takes up to two channel quads per thread.
template <int K1, int N2, int HT>
__global__ __launch_bounds__(S1_NTHR) void conv_sub1b_wide_fwd_kernel(Sub1DevB q, __bf16* hout, int hout_stride) {
  constexpr int P = 32, NTHR = S1_NTHR;
  constexpr int K1R = (K1 + 15) / 16 * 16, NS1 = K1R / 16, XSB = K1R * 2 + 16, HSB = S1_HID * 2 + 16, NU2 = N2 / 16, TS = N2 + 4;
  constexpr int NT2 = 2 * NU2, TPW = (NT2 + 7) / 8;               // 16 x 16 output tiles of a pixel tile; per wave
  static_assert(K1 % 16 == 0 && K1 <= 96 && N2 % 32 == 0 && N2 <= 192, "conv_sub1b_wide_fwd: shape");
  const ConvDev& pr = q.r;
  const ConvDev& pb = q.b;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_s1bw[];
  unsigned char* const xs0 = smem_s1bw;                           // 2 x [P][XSB]
  unsigned char* const hs = xs0 + 2 * P * XSB;                    // [P][HSB]
  unsigned char* const w2s = hs + P * HSB;                        // [N2][HSB]
  float* const T = reinterpret_cast<float*>(w2s + N2 * HSB);      // [P][TS]
  __shared__ float ldw[2][NTHR / 64];
  int ld_b[2] = {0, 0};
  int ld_pending = -1;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, hh = lane >> 5;
  const int cw = wave * 32;
  const bool odd = (r & 1) != 0;

  for (int f = tid; f < N2 * (S1_HID / 8); f += NTHR) {
    const int n = f / (S1_HID / 8), c8 = f - n * (S1_HID / 8);
    *reinterpret_cast<u32x4*>(w2s + n * HSB + c8 * 16) = *reinterpret_cast<const u32x4*>(q.w2 + (size_t)n * S1_HID + c8 * 8);
  }
  const float b1v = pr.bias ? pr.bias[cw + r] : 0.f;
  bf16x8 w1f[NS1];
#pragma unroll
  for (int s = 0; s < NS1; ++s) w1f[s] = *reinterpret_cast<const bf16x8*>(q.w1f + (size_t)(cw + r) * K1R + 16 * s + 8 * hh);

  // coupling epilogue: channel quads (pixel, 4 channels) dealt to the threads, NQE per thread
  constexpr int CO = N2 / 2, NQ = CO / 4, NQE = (P * NQ + NTHR - 1) / NTHR;
  const bool e_inv = pb.mode == SININN_CONV_COUPLE_INV;
  const int tiles_img = pr.tiles_x * pr.tiles_y;
  auto tile_origin = [&](int tile, int& b, int& y0, int& x0) {
    b = tile / tiles_img;
    const int trem = tile - b * tiles_img;
    const int ty = trem / pr.tiles_x;
    y0 = ty * 2; x0 = (trem - ty * pr.tiles_x) * 16;
  };

  constexpr int QX = K1 / 4, FX = (P * QX + NTHR - 1) / NTHR;
  auto issue_tile = [&](int tile, f32x4 (&vx)[FX]) {
    const bool live = tile < q.ntiles;
    int b, y0, x0; tile_origin(live ? tile : 0, b, y0, x0);
    const __amdgpu_buffer_rsrc_t x_rs = buf_rsrc(pr.in + (size_t)b * pr.H * pr.W * pr.in_stride);
#pragma unroll
    for (int u = 0; u < FX; ++u) {
      const int f = tid + NTHR * u;
      const int pl = f / QX, c = (f - pl * QX) * 4;
      const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
      const unsigned off = (live && f < P * QX && gy < pr.H && gx < pr.W) ? (unsigned)(((gy * pr.W + gx) * pr.in_stride + c) * 4) : BUF_OOB;
      vx[u] = buf_load4(x_rs, off, 0u);
    }
  };
  auto store_tile = [&](int buf, const f32x4 (&vx)[FX]) {
    unsigned char* const xs = xs0 + buf * (P * XSB);
#pragma unroll
    for (int u = 0; u < FX; ++u) {
      const int f = tid + NTHR * u;
      const int pl = f / QX, c = (f - pl * QX) * 4;
      if (f < P * QX) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (__bf16)vx[u][j];
        *reinterpret_cast<bf16x4*>(xs + pl * XSB + c * 2) = o;
      }
    }
  };
  // v of a tile's epilogue quads (raw buffer loads relative to the image: a quad outside the image / a tile beyond the last reads zeros)
  auto load_v = [&](int tile, f32x4 (&ev)[NQE]) {
    const bool live = tile < q.ntiles;
    int b, y0, x0; tile_origin(live ? tile : 0, b, y0, x0);
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(pb.v + (size_t)b * pb.H * pb.W * pb.v_stride);
#pragma unroll
    for (int i = 0; i < NQE; ++i) {
      const int e = tid + NTHR * i;
      const int pl = e / NQ, cl = 4 * (e - pl * NQ);
      const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
      const bool ok = live && e < P * NQ && gy < pb.H && gx < pb.W;
      ev[i] = buf_load4(rs, ok ? ((unsigned)(gy * pb.W + gx) * pb.v_stride + cl) * 4u : BUF_OOB, 0u);
    }
  };
  const int G = gridDim.x;
  const int row = lane & 15, kg = lane >> 4;
  const unsigned sel = odd ? 0x03020706u : 0x05040100u;
  // per-thread constants of the epilogue quads (the same for every tile): packed bias of s / t, output channel map
  f32x4 e_bs[NQE], e_bt[NQE];
  int e_om[NQE][4];
#pragma unroll
  for (int i = 0; i < NQE; ++i) {
    const int e = tid + NTHR * i;
    const int pl = e / NQ, cl = 4 * (e - pl * NQ);
    const int tcol = (cl / HT) * (2 * HT) + (cl % HT);
    e_bs[i] = e_bt[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) e_om[i][j] = cl + j;
    if (e < P * NQ) {
      if (pb.bias) {
        e_bs[i] = *reinterpret_cast<const f32x4*>(pb.bias + tcol);
        e_bt[i] = *reinterpret_cast<const f32x4*>(pb.bias + tcol + HT);
      }
      if (pb.out_map) {
#pragma unroll
        for (int j = 0; j < 4; ++j) e_om[i][j] = pb.out_map[cl + j];
      }
    }
  }

  auto body = [&](int tile, int buf, f32x4 (&vxn)[FX], f32x4 (&ev)[NQE]) {
    int b, y0, x0; tile_origin(tile, b, y0, x0);
    const unsigned char* const xs = xs0 + buf * (P * XSB);
    // ---- stage 1: h[32][32 of this wave] -> LDS (and HBM) ---------------------------------------------------------------------
    {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int s = 0; s < NS1; ++s) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(xs + r * XSB + (16 * s + 8 * hh) * 2);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, w1f[s], acc, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = fmaxf(acc[e] + b1v, 0.f);
      bf16x8 hf[2];
      acc_to_frags(acc, hf);
      const __amdgpu_buffer_rsrc_t h_rs = buf_rsrc(hout + (size_t)b * pr.H * pr.W * hout_stride);
      unsigned char* const base = hs + (4 * hh + (odd ? 1 : 0)) * HSB + (cw + (r & ~1)) * 2;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const u32x4 own = __builtin_bit_cast(u32x4, hf[s]);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)own[d], 0xB1, 0xF, 0xF, true);
          const unsigned word = __builtin_amdgcn_perm(nb, own[d], sel);
          const int p = 16 * s + 8 * (d >> 1) + 2 * (d & 1);     // + 4 hh + odd: this lane's pixel of the tile
          *reinterpret_cast<unsigned*>(base + p * HSB) = word;
          if (hout) {
            const int pp = p + 4 * hh + (odd ? 1 : 0);
            const int gy = y0 + (pp >> 4), gx = x0 + (pp & 15);
            const unsigned off = (gy < pr.H && gx < pr.W) ? (unsigned)(((gy * pr.W + gx) * hout_stride + cw + (r & ~1)) * 2) : BUF_OOB;
            __builtin_amdgcn_raw_buffer_store_b32(word, h_rs, (int)off, 0, 0);
          }
        }
      }
    }
    if (tile + G < q.ntiles) store_tile(buf ^ 1, vxn);
    issue_tile(tile + 3 * G, vxn);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                 // (B) the whole h tile is in LDS (and the next tile's x)
    if (pb.logdet) {
      if (tid == 0 && ld_pending >= 0) {
        const float* w8 = ldw[ld_pending];
        atomicAdd(pb.logdet + ld_b[ld_pending], ((w8[0] + w8[1]) + (w8[2] + w8[3])) + ((w8[4] + w8[5]) + (w8[6] + w8[7])));
      }
      ld_pending = buf;
    }
    // ---- stage 2: 16 x 16 tiles (pixel tile t & 1, column tile t >> 1), t = wave, wave + 8, ... ------------------------------------
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int t = wave + 8 * i;
      if (t < NT2) {
        const int pt = t & 1, nt = t >> 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S1_HID / 32; ++s) {
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(hs + (16 * pt + row) * HSB + (32 * s + 8 * kg) * 2);
          const bf16x8 bf = *reinterpret_cast<const bf16x8*>(w2s + (16 * nt + row) * HSB + (32 * s + 8 * kg) * 2);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) T[(16 * pt + 4 * kg + e) * TS + 16 * nt + row] = acc[e];
      }
    }
    __syncthreads();                                 // (E) the output tile is in LDS; every wave is done with hs
    asm volatile("" :: "v"(ev[0]));
    float ld_acc = 0.f;
    const size_t e_img = (size_t)b * pb.H * pb.W;
#pragma unroll
    for (int i = 0; i < NQE; ++i) {
      const int e = tid + NTHR * i;
      const int pl = e / NQ, cl = 4 * (e - pl * NQ);
      const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
      if (e < P * NQ && gy < pb.H && gx < pb.W) {
        const int tcol = (cl / HT) * (2 * HT) + (cl % HT);
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(T + pl * TS + tcol) + e_bs[i];
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(T + pl * TS + tcol + HT) + e_bt[i];
        const size_t pix = e_img + (unsigned)(gy * pb.W + gx);
        f32x4 y4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float L = glow_log_e(s4[j], pb.clamp);
          const float ex = expf(L);
          if (!e_inv) { y4[j] = ex * ev[i][j] + t4[j]; ld_acc += L; }
          else { y4[j] = (ev[i][j] - t4[j]) / ex; ld_acc -= L; }
        }
        if (pb.out_map) {
#pragma unroll
          for (int j = 0; j < 4; ++j) pb.out[pix * pb.out_stride + e_om[i][j]] = y4[j];
        } else {
          *reinterpret_cast<f32x4*>(pb.out + pix * pb.out_stride + cl) = y4;
        }
        if (pb.out2) *reinterpret_cast<f32x4*>(pb.out2 + pix * pb.out2_stride + cl) = y4;
        if (pb.sbuf) *reinterpret_cast<f32x4*>(pb.sbuf + pix * pb.Co + cl) = s4;
      }
    }
    load_v(tile + 2 * G, ev);
    if (pb.logdet) {
      const float wsum = wave_sum(ld_acc);
      if (lane == 0) ldw[buf][wave] = wsum;
      ld_b[buf] = b;
    }
  };

  f32x4 vxa[FX], vxb[FX], eva[NQE], evb[NQE];
  issue_tile(blockIdx.x, vxa);
  store_tile(0, vxa);
  issue_tile(blockIdx.x + G, vxa);
  issue_tile(blockIdx.x + 2 * G, vxb);
  load_v(blockIdx.x, eva);
  load_v(blockIdx.x + G, evb);
  __syncthreads();
  for (int tile = blockIdx.x; tile < q.ntiles; tile += 2 * G) {
    body(tile, 0, vxa, eva);
    if (tile + G < q.ntiles) body(tile + G, 1, vxb, evb);
  }
  if (pb.logdet) {
    __syncthreads();
    if (tid == 0 && ld_pending >= 0) {
      const float* w8 = ldw[ld_pending];
      atomicAdd(pb.logdet + ld_b[ld_pending], ((w8[0] + w8[1]) + (w8[2] + w8[3])) + ((w8[4] + w8[5]) + (w8[6] + w8[7])));
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Data gradients of the WIDE 1x1 subnets (level 1: dr[192] -> dh[256] . [h > 0] -> dx[96]), persistent, 32-pixel tiles: the twin of
// conv_sub1b_wide_fwd_kernel for the backward pair (conv_pair_bf16_kernel<96, .>: 90 us at BASELINE configs[3]).  The wave's
// fragments of conv2's data-gradient pack (K = 192: 12 fragments) live in registers, conv1's data-gradient pack (96 x 256) in LDS.
//   stage 2  dh[32][32 of this wave] = dr W2 on v_mfma_f32_32x32x16_bf16, rounded to bf16; the ReLU mask is applied to the packed
//            word (pixel, columns c, c + 1) after the lane swap, with the word of h loaded from the same address one tile ahead;
//            the word goes to the LDS image for stage 3 and to HBM (conv1's weight gradient reads dh)
//   stage 3  dx[32][96] = dh W1: 2 x 6 tiles of 16 x 16, K = 256 split in two halves over the waves (three half-tiles each), the
//            epilogue (ADD / ADD_CBWD_*, up to two channel quads per thread, side inputs requested a tile ahead) sums the halves
//   WG1      (training): dW1^T[k][c] += sum_p x[p][k] dh[p][c] and db1 += sum_p dh ride along -- the masked dh tile in its accumulator
//            layout is the B operand (as in conv_sub1b_bwd_kernel), x^T comes from a [channel][pixel] image; one slab per block,
//            wide_reduce_kernel sums them.  dh then never reaches HBM and conv1's problem leaves the grouped weight-gradient launch
template <int K1, int K2, bool WG1>
__global__ __launch_bounds__(S1_NTHR) void conv_sub1b_wide_bwd_kernel(Sub1DevB q, const __bf16* hmask, int hmask_stride, __bf16* dhout, int dhout_stride,
                                                                      const float* xin, int xin_stride) {
  constexpr int P = 32, NTHR = S1_NTHR;
  constexpr int XTB = P * 2 + 8, NM1 = K1 / 32;                  // bytes per channel row of the transposed x image; 32-row tiles of dW1^T
  constexpr int NS2 = K2 / 16, DSB = K2 * 2 + 16, HSB = S1_HID * 2 + 16, NP1 = K1, NU1 = NP1 / 16, TS = NP1 + 4;
  constexpr int NHT = 2 * NU1 * 2, TPW = (NHT + 7) / 8;          // (pixel tile, column tile, K half) pieces of stage 3; per wave
  static_assert(K1 % 16 == 0 && K1 <= 96 && K2 % 16 == 0 && K2 <= 192, "conv_sub1b_wide_bwd: shape");
  const ConvDev& pa = q.a;
  const ConvDev& pb = q.b;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_s1bwb[];
  unsigned char* const ds0 = smem_s1bwb;                          // 2 x [P][DSB]: dr as bf16
  unsigned char* const dhs = ds0 + 2 * P * DSB;                   // [P][HSB]: dh as bf16
  unsigned char* const wd = dhs + P * HSB;                        // [NP1][HSB]: W1 data-gradient pack
  float* const T0 = reinterpret_cast<float*>(wd + NP1 * HSB);     // 2 x [P][TS]: the two K halves of the dx tile
  float* const T1 = T0 + P * TS;
  unsigned char* const xt0 = reinterpret_cast<unsigned char*>(T1 + P * TS);   // WG1: 2 x [K1][XTB]: x^T as bf16

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, hh = lane >> 5;
  const int cw = wave * 32;
  const bool odd = (r & 1) != 0;
  const unsigned sel = odd ? 0x03020706u : 0x05040100u;

  f32x16 accW1[WG1 ? NM1 : 1];
  float accb1 = 0.f;
#pragma unroll
  for (int t = 0; t < (WG1 ? NM1 : 1); ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) accW1[t][e] = 0.f;

  for (int f = tid; f < NP1 * (S1_HID / 8); f += NTHR) {
    const int n = f / (S1_HID / 8), c8 = f - n * (S1_HID / 8);
    *reinterpret_cast<u32x4*>(wd + n * HSB + c8 * 16) = *reinterpret_cast<const u32x4*>(q.w1d + (size_t)n * S1_HID + c8 * 8);
  }
  bf16x8 w2f[NS2];
#pragma unroll
  for (int s = 0; s < NS2; ++s) w2f[s] = *reinterpret_cast<const bf16x8*>(q.w2 + (size_t)(cw + r) * K2 + 16 * s + 8 * hh);

  const int tiles_img = pa.tiles_x * pa.tiles_y;
  auto tile_origin = [&](int tile, int& b, int& y0, int& x0) {
    b = tile / tiles_img;
    const int trem = tile - b * tiles_img;
    const int ty = trem / pa.tiles_x;
    y0 = ty * 2; x0 = (trem - ty * pa.tiles_x) * 16;
  };
  constexpr int QD = K2 / 4, FD = (P * QD + NTHR - 1) / NTHR;
  auto issue_tile = [&](int tile, f32x4 (&vd)[FD]) {
    const bool live = tile < q.ntiles;
    int b, y0, x0; tile_origin(live ? tile : 0, b, y0, x0);
    const __amdgpu_buffer_rsrc_t d_rs = buf_rsrc(pa.in + (size_t)b * pa.H * pa.W * pa.in_stride);
#pragma unroll
    for (int u = 0; u < FD; ++u) {
      const int f = tid + NTHR * u;
      const int pl = f / QD, c = (f - pl * QD) * 4;
      const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
      const unsigned off = (live && f < P * QD && gy < pa.H && gx < pa.W) ? (unsigned)(((gy * pa.W + gx) * pa.in_stride + c) * 4) : BUF_OOB;
      vd[u] = buf_load4(d_rs, off, 0u);
    }
  };
  auto store_tile = [&](int buf, const f32x4 (&vd)[FD]) {
    unsigned char* const ds = ds0 + buf * (P * DSB);
#pragma unroll
    for (int u = 0; u < FD; ++u) {
      const int f = tid + NTHR * u;
      const int pl = f / QD, c = (f - pl * QD) * 4;
      if (f < P * QD) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (__bf16)vd[u][j];
        *reinterpret_cast<bf16x4*>(ds + pl * DSB + c * 2) = o;
      }
    }
  };
  constexpr int QXW = K1 / 4, FXW = (P * QXW + NTHR - 1) / NTHR;
  auto issue_x = [&](int tile, f32x4 (&vx)[FXW]) {
    const bool live = tile < q.ntiles;
    int b, y0, x0; tile_origin(live ? tile : 0, b, y0, x0);
    const __amdgpu_buffer_rsrc_t x_rs = buf_rsrc(xin + (size_t)b * pa.H * pa.W * xin_stride);
#pragma unroll
    for (int u = 0; u < FXW; ++u) {
      const int f = tid + NTHR * u;
      const int pl = f / QXW, c = (f - pl * QXW) * 4;
      const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
      const unsigned off = (live && f < P * QXW && gy < pa.H && gx < pa.W) ? (unsigned)(((gy * pa.W + gx) * xin_stride + c) * 4) : BUF_OOB;
      vx[u] = buf_load4(x_rs, off, 0u);
    }
  };
  auto store_x = [&](int buf, const f32x4 (&vx)[FXW]) {
    unsigned char* const xt = xt0 + buf * (K1 * XTB);
#pragma unroll
    for (int u = 0; u < FXW; ++u) {
      const int f = tid + NTHR * u;
      const int pl = f / QXW, c = (f - pl * QXW) * 4;
      if (f < P * QXW) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<__bf16*>(xt + (c + j) * XTB + pl * 2) = (__bf16)vx[u][j];
      }
    }
  };
  // byte offset (relative to the image, row stride in elements) of this lane's packed word d of fragment s: even lanes pixel pp,
  // columns (c, c + 1); odd lanes pixel pp + 1, columns (c - 1, c)
  auto word_off = [&](int y0, int x0, int s, int d, int stride) -> unsigned {
    const int pp = 16 * s + 8 * (d >> 1) + 2 * (d & 1) + 4 * hh + (odd ? 1 : 0);
    const int gy = y0 + (pp >> 4), gx = x0 + (pp & 15);
    return (gy < pa.H && gx < pa.W) ? (unsigned)(((gy * pa.W + gx) * stride + cw + (r & ~1)) * 2) : BUF_OOB;
  };
  auto issue_mask = [&](int tile, unsigned (&mk)[8]) {
    const bool live = tile < q.ntiles;
    int b, y0, x0; tile_origin(live ? tile : 0, b, y0, x0);
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(hmask + (size_t)b * pa.H * pa.W * hmask_stride);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int d = 0; d < 4; ++d)
        mk[4 * s + d] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(live ? word_off(y0, x0, s, d, hmask_stride) : BUF_OOB), 0, 0);
  };

  // epilogue quads (pixel, 4 columns of dx) dealt to the threads; per-thread constants
  constexpr int NQ = NP1 / 4, NQE = (P * NQ + NTHR - 1) / NTHR;
  const int emode = pb.mode;
  const bool e_cbwd = emode == SININN_CONV_ADD_CBWD_FWD || emode == SININN_CONV_ADD_CBWD_INV;
  const bool e_fast = !q.no_dx;
  int amap[NQE][4];
  f32x4 e_bq[NQE];
#pragma unroll
  for (int i = 0; i < NQE; ++i) {
    const int e = tid + NTHR * i;
    const int pl = e / NQ, col = 4 * (e - pl * NQ);
    e_bq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) amap[i][j] = col + j;
    if (e_fast && e < P * NQ && col < pb.N) {
      if (pb.addend_map) {
#pragma unroll
        for (int j = 0; j < 4; ++j) amap[i][j] = pb.addend_map[col + j];
      }
      if (pb.bias) e_bq[i] = *reinterpret_cast<const f32x4*>(pb.bias + col);
    }
  }
  struct Side { f32x4 ad, u, s; };
  auto load_side = [&](int tile, Side (&sd)[NQE]) {
    const bool live = tile < q.ntiles;
    int b, y0, x0; tile_origin(live ? tile : 0, b, y0, x0);
    const size_t img = (size_t)b * pb.H * pb.W;
    const __amdgpu_buffer_rsrc_t ad_rs = buf_rsrc(pb.addend + img * pb.addend_stride);
    const __amdgpu_buffer_rsrc_t u_rs = buf_rsrc((e_cbwd ? pb.v : pb.addend) + img * (e_cbwd ? pb.v_stride : 0));
    const __amdgpu_buffer_rsrc_t s_rs = buf_rsrc((e_cbwd ? pb.sbuf : pb.addend) + img * (e_cbwd ? pb.Co : 0));
#pragma unroll
    for (int i = 0; i < NQE; ++i) {
      const int e = tid + NTHR * i;
      const int pl = e / NQ, col = 4 * (e - pl * NQ);
      const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
      const bool ok = live && e_fast && e < P * NQ && col < pb.N && gy < pb.H && gx < pb.W;
      const unsigned ip = (unsigned)(gy * pb.W + gx);
      if (pb.addend_map) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          sd[i].ad[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ad_rs, (int)(ok ? (ip * pb.addend_stride + amap[i][j]) * 4u : BUF_OOB), 0, 0));
      } else {
        sd[i].ad = buf_load4(ad_rs, ok ? (ip * pb.addend_stride + col) * 4u : BUF_OOB, 0u);
      }
      const bool cb = ok && e_cbwd;
      sd[i].u = buf_load4(u_rs, cb ? (ip * pb.v_stride + col) * 4u : BUF_OOB, 0u);
      sd[i].s = buf_load4(s_rs, cb ? (ip * pb.Co + col) * 4u : BUF_OOB, 0u);
    }
  };

  const int G = gridDim.x;
  const int row = lane & 15, kg = lane >> 4;
  f32x4 vd[FD], vxw[FXW];
  unsigned mk[8];
  Side sd[NQE];
  issue_tile(blockIdx.x, vd);
  if constexpr (WG1) issue_x(blockIdx.x, vxw);
  store_tile(0, vd);
  if constexpr (WG1) store_x(0, vxw);
  issue_tile(blockIdx.x + G, vd);
  if constexpr (WG1) issue_x(blockIdx.x + G, vxw);
  issue_mask(blockIdx.x, mk);
  load_side(blockIdx.x, sd);
  __syncthreads();
  int buf = 0;
  for (int tile = blockIdx.x; tile < q.ntiles; tile += G, buf ^= 1) {
    int b, y0, x0; tile_origin(tile, b, y0, x0);
    const unsigned char* const ds = ds0 + buf * (P * DSB);
    // ---- stage 2: dh = (dr W2) . [h > 0] -> LDS + HBM ---------------------------------------------------------------------------------
    {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int s = 0; s < NS2; ++s) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(ds + r * DSB + (16 * s + 8 * hh) * 2);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, w2f[s], acc, 0, 0, 0);
      }
      bf16x8 hf[2];
      acc_to_frags(acc, hf);
      const __amdgpu_buffer_rsrc_t o_rs = buf_rsrc(dhout + (size_t)b * pa.H * pa.W * dhout_stride);
      unsigned char* const base = dhs + (4 * hh + (odd ? 1 : 0)) * HSB + (cw + (r & ~1)) * 2;
      auto keep_of = [](unsigned mw) -> unsigned {   // h > 0 of the two bf16 halves: not zero and not negative
        return (((mw & 0x7fffu) != 0u && (mw & 0x8000u) == 0u) ? 0x0000ffffu : 0u) |
               (((mw & 0x7fff0000u) != 0u && (mw & 0x80000000u) == 0u) ? 0xffff0000u : 0u);
      };
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        u32x4 own = __builtin_bit_cast(u32x4, hf[s]);
        if constexpr (WG1) {
          // the mask words arrive in the swapped layout (pixel, columns c, c + 1): swapped back (the swap is an involution) they
          // mask this lane's own (pixel, pixel + 1) pairs, and the masked fragment is the B operand of the weight-gradient MFMAs
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const unsigned mw = mk[4 * s + d];
            const unsigned mnb = (unsigned)__builtin_amdgcn_mov_dpp((int)mw, 0xB1, 0xF, 0xF, true);
            own[d] &= keep_of(__builtin_amdgcn_perm(mnb, mw, sel));
          }
          hf[s] = __builtin_bit_cast(bf16x8, own);
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)own[d], 0xB1, 0xF, 0xF, true);
          unsigned word = __builtin_amdgcn_perm(nb, own[d], sel);
          if constexpr (!WG1) word &= keep_of(mk[4 * s + d]);
          *reinterpret_cast<unsigned*>(base + (16 * s + 8 * (d >> 1) + 2 * (d & 1)) * HSB) = word;
          if (dhout) __builtin_amdgcn_raw_buffer_store_b32(word, o_rs, (int)word_off(y0, x0, s, d, dhout_stride), 0, 0);
        }
      }
      if constexpr (WG1) {
        const unsigned char* const xt = xt0 + buf * (K1 * XTB);
#pragma unroll
        for (int t = 0; t < NM1; ++t)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const bf16x8 af = read_tr_frag(xt + (32 * t + r) * XTB + (16 * s + 4 * hh) * 2);
            accW1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, hf[s], accW1[t], 0, 0, 0);
          }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < 8; ++j) accb1 += (float)hf[s][j];
      }
    }
    issue_mask(tile + G, mk);                        // the next tile's mask words: a tile ahead of their use
    if (tile + G < q.ntiles) {
      store_tile(buf ^ 1, vd);
      if constexpr (WG1) store_x(buf ^ 1, vxw);
    }
    issue_tile(tile + 2 * G, vd);
    if constexpr (WG1) issue_x(tile + 2 * G, vxw);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                 // (C) the whole dh tile is in LDS (and the next tile's dr / x^T)

    // ---- stage 3: dx[32][NP1] = dh W1, K halves: piece t = wave + 8 i -> (K half t & 1, pixel tile (t >> 1) & 1, column tile t >> 2) ----
    if (!q.no_dx) {
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int t = wave + 8 * i;
        if (t < NHT) {
          const int g2 = t & 1, pt = (t >> 1) & 1, nt = t >> 2;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < S1_HID / 64; ++s) {
            const int k0 = 128 * g2 + 32 * s + 8 * kg;
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(dhs + (16 * pt + row) * HSB + k0 * 2);
            const bf16x8 bf = *reinterpret_cast<const bf16x8*>(wd + (16 * nt + row) * HSB + k0 * 2);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc, 0, 0, 0);
          }
          float* const T = g2 ? T1 : T0;
#pragma unroll
          for (int e = 0; e < 4; ++e) T[(16 * pt + 4 * kg + e) * TS + 16 * nt + row] = acc[e];
        }
      }
    }
    __syncthreads();                                 // (E) both halves of the dx tile are in LDS; every wave is done with dhs
    asm volatile("" :: "v"(sd[0].ad), "v"(sd[0].u), "v"(sd[0].s));
    if (!q.no_dx) {
#pragma unroll
      for (int i = 0; i < NQE; ++i) {
        const int e = tid + NTHR * i;
        const int pl = e / NQ, col = 4 * (e - pl * NQ);
        const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
        if (e < P * NQ && col < pb.N && gy < pb.H && gx < pb.W) {
          const size_t pix = (size_t)b * pb.H * pb.W + (unsigned)(gy * pb.W + gx);
          f32x4 val = *reinterpret_cast<const f32x4*>(T0 + pl * TS + col) + *reinterpret_cast<const f32x4*>(T1 + pl * TS + col);
          val += e_bq[i];
          val += sd[i].ad;
          if (!e_cbwd) {
            *reinterpret_cast<f32x4*>(pb.out + pix * pb.out_stride + col) = val;
          } else {
            const float gl = pb.logdet ? pb.logdet[b] : 0.f;
            f32x4 o_a, o_b, o_c;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float g = val[j], u = sd[i].u[j], sv = sd[i].s[j];
              const float L = glow_log_e(sv, pb.clamp), dL = glow_dlog_e(sv, pb.clamp);
              const float ex = expf(L);
              if (emode == SININN_CONV_ADD_CBWD_FWD) { o_c[j] = g * ex; o_b[j] = g; o_a[j] = (g * u * ex + gl) * dL; }
              else { o_c[j] = g / ex; o_b[j] = -o_c[j]; o_a[j] = -(g * u + gl) * dL; }
            }
            *reinterpret_cast<f32x4*>(pb.out + pix * pb.out_stride + col) = o_a;
            *reinterpret_cast<f32x4*>(pb.out + pix * pb.out_stride + pb.Co + col) = o_b;
            *reinterpret_cast<f32x4*>(pb.out2 + pix * pb.out2_stride + col) = o_c;
          }
        }
      }
    }
    load_side(tile + G, sd);                         // the next tile's side inputs: a tile ahead of their use
  }
  if constexpr (WG1) {                               // slab of this block: dW1^T [K1][256] | db1 [256]
    float* const slab = q.slab + (size_t)blockIdx.x * ((K1 + 1) * S1_HID);
#pragma unroll
    for (int t = 0; t < NM1; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) slab[(32 * t + (e & 3) + 8 * (e >> 2) + 4 * hh) * S1_HID + cw + r] = accW1[t][e];
    const float bsum = accb1 + __shfl_xor(accb1, 32);
    if (hh == 0) slab[K1 * S1_HID + cw + r] = bsum;
  }
}

// gw1[n][k] += sum over the slabs of dW1^T[k][n], gb1[n] += sum of db1[n]: eight interleaved groups of slabs, each summed in
// ascending order by one thread, the eight partial sums added as a balanced tree (the association of sub1_reduce_kernel)
template <int K1>
__global__ __launch_bounds__(256) void wide_reduce_kernel(const float* __restrict__ slabs, int S, float* __restrict__ gw1, float* __restrict__ gb1) {
  constexpr int ET = (K1 + 1) * S1_HID / 4;
  constexpr size_t slab4 = (size_t)(K1 + 1) * S1_HID / 4;
  __shared__ f32x4 part[8][32];
  const int tid = threadIdx.x, g = tid >> 5, el = tid & 31;
  const int e = blockIdx.x * 32 + el;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
  if (e < ET) {
    const f32x4* src = reinterpret_cast<const f32x4*>(slabs) + e;
    int s = g;
#pragma unroll 4
    for (; s + 8 < S; s += 16) { a0 += src[(size_t)s * slab4]; a1 += src[(size_t)(s + 8) * slab4]; }
    if (s < S) a0 += src[(size_t)s * slab4];
    a0 += a1;
  }
  part[g][el] = a0;
  __syncthreads();
  if (tid < 32 && e < ET) {
    const f32x4 tot = ((part[0][el] + part[1][el]) + (part[2][el] + part[3][el])) + ((part[4][el] + part[5][el]) + (part[6][el] + part[7][el]));
    const int idx = e * 4, k = idx / S1_HID, n = idx - k * S1_HID;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (k < K1) { if (gw1) gw1[(n + j) * K1 + k] += tot[j]; }
      else if (gb1) gb1[n + j] += tot[j];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// conv2's weight gradient of a WIDE 1x1 subnet (level 1: dW2[n][c] += sum_p dr[p][n] h[p][c], db2 += sum_p dr), persistent, 32-pixel
// tiles: the third kernel of the level-1 backward.  h is read from HBM (bf16 [pixel][256]) with 4-byte loads in the lane-swapped
// layout the forward kernel stored it in; swapped back it is the B operand as it stands (column on the lane, pixel pairs in the
// registers) -- no LDS for the 256-channel operand at all; dr^T comes from a [channel][pixel] image (fp32 dr rounded while staged).
// Six 32 x 32 accumulators per wave for the lifetime of the block, one slab per block (wide_reduce2_kernel).
template <int K2>
__global__ __launch_bounds__(S1_NTHR) void conv_sub1b_wide_wg2_kernel(const float* dr, int dr_stride, const __bf16* hin, int h_stride, int B, int H, int W,
                                                                      int tiles_x, int tiles_y, int ntiles, float* slabs) {
  constexpr int P = 32, NTHR = S1_NTHR, NM2 = K2 / 32, DTB = P * 2 + 8;
  static_assert(K2 % 32 == 0 && K2 <= 192, "conv_sub1b_wide_wg2: shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_s1bw2[];    // 2 x [K2][DTB]: dr^T as bf16
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, hh = lane >> 5;
  const int cw = wave * 32;
  const bool odd = (r & 1) != 0;
  const unsigned sel = odd ? 0x03020706u : 0x05040100u;
  (void)B;

  f32x16 acc[NM2];
#pragma unroll
  for (int t = 0; t < NM2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  const int tiles_img = tiles_x * tiles_y;
  auto tile_origin = [&](int tile, int& b, int& y0, int& x0) {
    b = tile / tiles_img;
    const int trem = tile - b * tiles_img;
    const int ty = trem / tiles_x;
    y0 = ty * 2; x0 = (trem - ty * tiles_x) * 16;
  };
  constexpr int QD = K2 / 4, FD = (P * QD + NTHR - 1) / NTHR;
  f32x4 accb2[FD];                                  // db2 in fp32: a staging slot is the same (pixel, channel quad) of every tile
#pragma unroll
  for (int u = 0; u < FD; ++u) accb2[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto issue_tile = [&](int tile, f32x4 (&vd)[FD]) {
    const bool live = tile < ntiles;
    int b, y0, x0; tile_origin(live ? tile : 0, b, y0, x0);
    const __amdgpu_buffer_rsrc_t d_rs = buf_rsrc(dr + (size_t)b * H * W * dr_stride);
#pragma unroll
    for (int u = 0; u < FD; ++u) {
      const int f = tid + NTHR * u;
      const int pl = f / QD, c = (f - pl * QD) * 4;
      const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
      const unsigned off = (live && f < P * QD && gy < H && gx < W) ? (unsigned)(((gy * W + gx) * dr_stride + c) * 4) : BUF_OOB;
      vd[u] = buf_load4(d_rs, off, 0u);
    }
  };
  auto store_tile = [&](int buf, const f32x4 (&vd)[FD]) {
    unsigned char* const dt = smem_s1bw2 + buf * (K2 * DTB);
#pragma unroll
    for (int u = 0; u < FD; ++u) {
      const int f = tid + NTHR * u;
      const int pl = f / QD, c = (f - pl * QD) * 4;
      if (f < P * QD) {
        accb2[u] += vd[u];                            // pixels outside the image were loaded as zeros
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<__bf16*>(dt + (c + j) * DTB + pl * 2) = (__bf16)vd[u][j];
      }
    }
  };
  auto issue_h = [&](int tile, unsigned (&hw)[8]) {
    const bool live = tile < ntiles;
    int b, y0, x0; tile_origin(live ? tile : 0, b, y0, x0);
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(hin + (size_t)b * H * W * h_stride);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int pp = 16 * s + 8 * (d >> 1) + 2 * (d & 1) + 4 * hh + (odd ? 1 : 0);
        const int gy = y0 + (pp >> 4), gx = x0 + (pp & 15);
        const unsigned off = (live && gy < H && gx < W) ? (unsigned)(((gy * W + gx) * h_stride + cw + (r & ~1)) * 2) : BUF_OOB;
        hw[4 * s + d] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0);
      }
  };

  const int G = gridDim.x;
  f32x4 vd[FD];
  unsigned hw[8];
  issue_tile(blockIdx.x, vd);
  store_tile(0, vd);
  issue_tile(blockIdx.x + G, vd);
  issue_h(blockIdx.x, hw);
  __syncthreads();
  int buf = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += G, buf ^= 1) {
    const unsigned char* const dt = smem_s1bw2 + buf * (K2 * DTB);
    bf16x8 hf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 own;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const unsigned w_ = hw[4 * s + d];
        const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)w_, 0xB1, 0xF, 0xF, true);
        own[d] = __builtin_amdgcn_perm(nb, w_, sel);                 // back to (pixel, pixel + 1) of this lane's column
      }
      hf[s] = __builtin_bit_cast(bf16x8, own);
    }
    issue_h(tile + G, hw);
#pragma unroll
    for (int t = 0; t < NM2; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 af = read_tr_frag(dt + (32 * t + r) * DTB + (16 * s + 4 * hh) * 2);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, hf[s], acc[t], 0, 0, 0);
      }
    if (tile + G < ntiles) store_tile(buf ^ 1, vd);
    issue_tile(tile + 2 * G, vd);
    __syncthreads();
  }
  // slab: dW2 [K2][256] | db2 [256: first K2 used]
  float* const slab = slabs + (size_t)blockIdx.x * ((K2 + 1) * S1_HID);
#pragma unroll
  for (int t = 0; t < NM2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) slab[(32 * t + (e & 3) + 8 * (e >> 2) + 4 * hh) * S1_HID + cw + r] = acc[t][e];
  __syncthreads();
  float* const red = reinterpret_cast<float*>(smem_s1bw2);         // [P * QD] float4 = 24 KB of the 27 KB
#pragma unroll
  for (int u = 0; u < FD; ++u) {
    const int f = tid + NTHR * u;
    if (f < P * QD) *reinterpret_cast<f32x4*>(red + 4 * f) = accb2[u];
  }
  __syncthreads();
  if (tid < S1_HID) {
    float v = 0.f;
    if (tid < K2) {
      float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int pl = 0; pl < P; pl += 4)
#pragma unroll
        for (int i = 0; i < 4; ++i) part[i] += red[(pl + i) * K2 + tid];
      v = (part[0] + part[1]) + (part[2] + part[3]);
    }
    slab[K2 * S1_HID + tid] = v;
  }
}

// gw2[n][c] += sum over the slabs of dW2[n][c] (the slab row IS the OIHW row of a 1x1 conv), gb2[n] += sum of db2[n]
template <int K2>
__global__ __launch_bounds__(256) void wide_reduce2_kernel(const float* __restrict__ slabs, int S, float* __restrict__ gw2, float* __restrict__ gb2) {
  constexpr int ET = (K2 + 1) * S1_HID / 4;
  constexpr size_t slab4 = (size_t)(K2 + 1) * S1_HID / 4;
  __shared__ f32x4 part[8][32];
  const int tid = threadIdx.x, g = tid >> 5, el = tid & 31;
  const int e = blockIdx.x * 32 + el;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
  if (e < ET) {
    const f32x4* src = reinterpret_cast<const f32x4*>(slabs) + e;
    int s = g;
#pragma unroll 4
    for (; s + 8 < S; s += 16) { a0 += src[(size_t)s * slab4]; a1 += src[(size_t)(s + 8) * slab4]; }
    if (s < S) a0 += src[(size_t)s * slab4];
    a0 += a1;
  }
  part[g][el] = a0;
  __syncthreads();
  if (tid < 32 && e < ET) {
    const f32x4 tot = ((part[0][el] + part[1][el]) + (part[2][el] + part[3][el])) + ((part[4][el] + part[5][el]) + (part[6][el] + part[7][el]));
    const int idx = e * 4;
    if (idx < K2 * S1_HID) {
      if (gw2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) gw2[idx + j] += tot[j];
      }
    } else if (gb2) {
      const int n = idx - K2 * S1_HID;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n + j < K2) gb2[n + j] += tot[j];
    }
  }
}

// ---- host -----------------------------------------------------------------------------------------------------------------------
size_t conv_sub1_bwd_workspace_bytes(int cond_cin, int co);      // conv_sub1.hip: the slab layout is shared
static bool g_sub1b_enabled = getenv("SININN_SUB1_BF16") == nullptr || atoi(getenv("SININN_SUB1_BF16")) != 0;   // A/B switch
void conv_sub1_bf16_enable(int on) { g_sub1b_enabled = on != 0; }
bool conv_sub1_bf16_enabled() { return g_sub1b_enabled; }

static int fill_common(Sub1DevB& q, int B, int H, int W) {
  q.r.tiles_x = q.a.tiles_x = q.b.tiles_x = (W + 15) / 16;
  q.r.tiles_y = q.a.tiles_y = q.b.tiles_y = (H + 3) / 4;
  q.ntiles = q.r.tiles_x * q.r.tiles_y * B;
  q.r.stamp = q.a.stamp = q.b.stamp = nullptr;
  return 0;
}

template <int K1, int K2>
static int sub1b_launch(Sub1DevB& q, int* blocks_out, hipStream_t st) {
  using SB = Sub1BShape<K1, K2>;
  auto k = conv_sub1b_bwd_kernel<K1, K2>;
  static_assert(SB::LDS <= 160 * 1024, "conv_sub1b_bwd: LDS");
  if (SB::LDS > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SB::LDS);
    if (e != hipSuccess) { set_error("conv_sub1_bf16_bwd: cannot raise the LDS limit to %zu", (size_t)SB::LDS); return 1; }
  }
  const int blocks = q.ntiles < S1_MAX_BLOCKS ? q.ntiles : S1_MAX_BLOCKS;
  hipLaunchKernelGGL(k, dim3(blocks), dim3(S1_NTHR), SB::LDS, st, q);
  SININN_LAUNCH_CHECK("conv_sub1_bf16_bwd");
  *blocks_out = blocks;
  return 0;
}

// rc / d2 / d1 as conv_sub1_bwd_launch takes them, on the mixed-precision path: bf16 weight packs (w_bf16), fp32 x / dr / dx.
int conv_sub1_bf16_bwd_launch(const sininn_conv_args* rc, const sininn_conv_args* d2, const sininn_conv_args* d1, int no_dx, void* ws,
                              size_t ws_bytes, int* slabs_out, hipStream_t st) {
  SININN_CHECK(rc && d2 && d1 && ws && slabs_out, "conv_sub1_bf16_bwd: null argument");
  const int K1 = rc->Cin, K2 = d2->Cin;
  SININN_CHECK(sub1_shape_ok(K1, K2), "conv_sub1_bf16_bwd: unsupported subnet shape (Cin %d, 2 Co %d)", K1, K2);
  SININN_CHECK(rc->ksize == 1 && d2->ksize == 1 && d1->ksize == 1 && rc->w_bf16 && d2->w_bf16 && d1->w_bf16 && !rc->in_bf16 && !d2->in_bf16,
               "conv_sub1_bf16_bwd: 1x1 convs with bf16 weight packs and fp32 x / dr only");
  SININN_CHECK(rc->Np == S1_HID && d2->Np == S1_HID && d1->Cin == S1_HID && d1->N == K1 && d1->Np == (K1 + 15) / 16 * 16,
               "conv_sub1_bf16_bwd: the three convs do not form a subnet backward");
  SININN_CHECK(rc->B == d2->B && rc->H == d2->H && rc->W == d2->W && rc->B == d1->B && rc->H == d1->H && rc->W == d1->W, "conv_sub1_bf16_bwd: shapes differ");
  SININN_CHECK(rc->in_stride % 4 == 0 && d2->in_stride % 4 == 0 && aligned16(rc->in) && aligned16(d2->in) && aligned16(rc->w) && aligned16(d2->w) &&
               aligned16(d1->w) && aligned16(ws), "conv_sub1_bf16_bwd: operands must be 16-byte aligned with strides that are multiples of 4 floats");
  SININN_CHECK((unsigned long long)rc->H * rc->W * (rc->in_stride > d2->in_stride ? rc->in_stride : d2->in_stride) * 4ull < (1ull << 31),
               "conv_sub1_bf16_bwd: one image of an operand exceeds the 2 GB a block addresses (raw buffer staging)");
  SININN_CHECK(ws_bytes >= conv_sub1_bwd_workspace_bytes(K1, K2 / 2), "conv_sub1_bf16_bwd: workspace too small (%zu < %zu)", ws_bytes,
               conv_sub1_bwd_workspace_bytes(K1, K2 / 2));
  if (!no_dx) {
    const bool cbwd = d1->mode == SININN_CONV_ADD_CBWD_FWD || d1->mode == SININN_CONV_ADD_CBWD_INV;
    SININN_CHECK(cbwd || d1->mode == SININN_CONV_ADD, "conv_sub1_bf16_bwd: d1->mode must be ADD or ADD_CBWD_* (got %d)", d1->mode);
    SININN_CHECK(d1->mask == nullptr && d1->out_map == nullptr && d1->out && d1->addend && d1->out_stride % 4 == 0 && aligned16(d1->out) &&
                 (d1->addend_map != nullptr || (d1->addend_stride % 4 == 0 && aligned16(d1->addend))) && (!d1->bias || aligned16(d1->bias)),
                 "conv_sub1_bf16_bwd: d1 needs 16-byte aligned out / addend with strides that are multiples of 4, no mask, no out_map");
    if (cbwd)
      SININN_CHECK(d1->Co % 4 == 0 && d1->v_stride % 4 == 0 && d1->out2_stride % 4 == 0 && d1->v && d1->sbuf && d1->out2 && aligned16(d1->v) &&
                   aligned16(d1->sbuf) && aligned16(d1->out2), "conv_sub1_bf16_bwd: ADD_CBWD needs 16-byte aligned v / sbuf / out2");
  }
  Sub1DevB q = {};
  alignas(16) static float dummy[8] = {};          // conv_bf16_prepare insists on pointers the kernel never follows
  ConvDevB t;
  sininn_conv_args ra = *rc;
  ra.mode = SININN_CONV_RELU; ra.out = dummy; ra.out_stride = S1_HID; ra.N = S1_HID; ra.out_bf16 = 1; ra.mask = nullptr;
  if (int e = conv_bf16_prepare(&ra, t)) return e;
  q.r = t.c; q.r.in = rc->in; q.w1f = t.w;
  sininn_conv_args da = *d2;
  da.mode = SININN_CONV_LINEAR; da.mask = nullptr; da.mask_bf16 = 0; da.out = dummy; da.out_stride = S1_HID; da.N = S1_HID; da.out_bf16 = 1;
  if (int e = conv_bf16_prepare(&da, t)) return e;
  q.a = t.c; q.a.in = d2->in; q.w2 = t.w;
  sininn_conv_args db = *d1;
  db.in = dummy; db.in_stride = S1_HID; db.in_bf16 = 1;
  if (no_dx) { db.mode = SININN_CONV_LINEAR; db.out = dummy; db.out_stride = (K1 + 3) / 4 * 4; db.addend = nullptr; db.addend_map = nullptr; }
  if (int e = conv_bf16_prepare(&db, t)) return e;
  q.b = t.c; q.w1d = t.w;
  fill_common(q, rc->B, rc->H, rc->W);
  q.no_dx = no_dx ? 1 : 0;
  q.slab = static_cast<float*>(ws);
  if (K1 == 8) return sub1b_launch<8, 16>(q, slabs_out, st);
  if (K1 == 16) return sub1b_launch<16, 32>(q, slabs_out, st);
  return sub1b_launch<24, 48>(q, slabs_out, st);
}

template <int K1, int N2, int HT>
static int sub1b_fwd_launch(Sub1DevB& q, hipStream_t st) {
  constexpr int K1R = (K1 + 15) / 16 * 16;
  constexpr size_t lds = (size_t)2 * S1_P * (K1R * 2 + 16) + (size_t)(S1_P + N2) * (S1_HID * 2 + 16) + (size_t)2 * S1_P * (N2 + 4) * 4;
  static_assert(lds + 64 <= 160 * 1024, "conv_sub1b_fwd: LDS");
  auto k = conv_sub1b_fwd_kernel<K1, N2, HT>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("conv_sub1_bf16_fwd: cannot raise the LDS limit to %zu", lds); return 1; }
  }
  const int blocks = q.ntiles < S1_MAX_BLOCKS ? q.ntiles : S1_MAX_BLOCKS;
  hipLaunchKernelGGL(k, dim3(blocks), dim3(S1_NTHR), lds, st, q);
  SININN_LAUNCH_CHECK("conv_sub1_bf16_fwd");
  return 0;
}

// first / second as conv_pair_bf16_launch takes them (conv1: fp32 in, bf16 hidden; conv2: coupling epilogue); h is never stored
int conv_sub1_bf16_fwd_supported(const sininn_conv_args* f, const sininn_conv_args* s) {
  if (!g_sub1b_enabled || !f || !s) return 0;
  if (f->ksize != 1 || s->ksize != 1 || !f->w_bf16 || !s->w_bf16 || f->winograd || s->winograd || f->in_bf16 || !s->in_bf16 || s->out_bf16) return 0;
  if (f->in_group_stride > 0 || f->out_group_stride > 0 || s->in_group_stride > 0) return 0;
  if (f->mode != SININN_CONV_RELU || !(s->mode == SININN_CONV_COUPLE_FWD || s->mode == SININN_CONV_COUPLE_INV)) return 0;
  if (f->Np != S1_HID || f->N != S1_HID || s->Cin != S1_HID) return 0;
  if (f->B != s->B || f->H != s->H || f->W != s->W) return 0;
  return sub1_shape_ok(f->Cin, s->Np) ? 1 : 0;
}

int conv_sub1_bf16_fwd_launch(const sininn_conv_args* f, const sininn_conv_args* s, hipStream_t st) {
  SININN_CHECK(conv_sub1_bf16_fwd_supported(f, s), "conv_sub1_bf16_fwd: unsupported subnet");
  SININN_CHECK((unsigned long long)f->H * f->W * f->in_stride * 4ull < (1ull << 31),
               "conv_sub1_bf16_fwd: one image of the input exceeds the 2 GB a block addresses (raw buffer staging)");
  Sub1DevB q = {};
  alignas(16) static float dummy[8] = {};
  ConvDevB t;
  sininn_conv_args fa = *f;
  fa.out = dummy; fa.out_stride = S1_HID; fa.out_bf16 = 1;
  if (int e = conv_bf16_prepare(&fa, t)) return e;
  q.r = t.c; q.r.in = f->in; q.w1f = t.w;
  sininn_conv_args sa = *s;
  sa.in = dummy; sa.in_stride = S1_HID;
  if (int e = conv_bf16_prepare(&sa, t)) return e;
  q.b = t.c; q.a = t.c; q.w2 = t.w; q.w1d = nullptr;
  fill_common(q, f->B, f->H, f->W);
  q.no_dx = 0; q.slab = nullptr;
  const bool ht16 = s->col_tile == 32;
  SININN_CHECK(!ht16 || s->Co % 16 == 0, "conv_sub1_bf16_fwd: col_tile 32 needs Co %% 16 == 0");
  if (f->Cin == 8) return ht16 ? sub1b_fwd_launch<8, 16, 16>(q, st) : sub1b_fwd_launch<8, 16, 8>(q, st);
  if (f->Cin == 16) return ht16 ? sub1b_fwd_launch<16, 32, 16>(q, st) : sub1b_fwd_launch<16, 32, 8>(q, st);
  return ht16 ? sub1b_fwd_launch<24, 48, 16>(q, st) : sub1b_fwd_launch<24, 48, 8>(q, st);
}

// ---- wide subnets (level 1): forward only ------------------------------------------------------------------------------------------
static bool g_sub1b_wide = getenv("SININN_SUB1_BF16_WIDE") == nullptr || atoi(getenv("SININN_SUB1_BF16_WIDE")) != 0;   // A/B switch

template <int K1, int N2, int HT>
static int sub1b_wide_launch(Sub1DevB& q, __bf16* hout, int hout_stride, hipStream_t st) {
  constexpr int K1R = (K1 + 15) / 16 * 16;
  constexpr size_t lds = (size_t)2 * 32 * (K1R * 2 + 16) + (size_t)(32 + N2) * (S1_HID * 2 + 16) + (size_t)32 * (N2 + 4) * 4;
  static_assert(lds + 64 <= 160 * 1024, "conv_sub1b_wide_fwd: LDS");
  auto k = conv_sub1b_wide_fwd_kernel<K1, N2, HT>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("conv_sub1_bf16_wide_fwd: cannot raise the LDS limit to %zu", lds); return 1; }
  const int blocks = q.ntiles < S1_MAX_BLOCKS ? q.ntiles : S1_MAX_BLOCKS;
  hipLaunchKernelGGL(k, dim3(blocks), dim3(S1_NTHR), lds, st, q, hout, hout_stride);
  SININN_LAUNCH_CHECK("conv_sub1_bf16_wide_fwd");
  return 0;
}

// first / second as conv_pair_bf16_launch takes them; first->out (bf16 [pixel][256]) may be NULL: h is then not stored
int conv_sub1_bf16_wide_fwd_supported(const sininn_conv_args* f, const sininn_conv_args* s) {
  if (!g_sub1b_enabled || !g_sub1b_wide || !f || !s) return 0;
  if (f->ksize != 1 || s->ksize != 1 || !f->w_bf16 || !s->w_bf16 || f->winograd || s->winograd || f->in_bf16 || !s->in_bf16 || s->out_bf16) return 0;
  if (f->in_group_stride > 0 || f->out_group_stride > 0 || s->in_group_stride > 0) return 0;
  if (f->mode != SININN_CONV_RELU || !(s->mode == SININN_CONV_COUPLE_FWD || s->mode == SININN_CONV_COUPLE_INV)) return 0;
  if (f->Np != S1_HID || f->N != S1_HID || s->Cin != S1_HID) return 0;
  if (f->B != s->B || f->H != s->H || f->W != s->W) return 0;
  if (f->out && (!f->out_bf16 || f->out_stride % 8 != 0 || (unsigned long long)f->H * f->W * f->out_stride * 2ull >= (1ull << 31))) return 0;
  if ((unsigned long long)f->H * f->W * f->in_stride * 4ull >= (1ull << 31) || (unsigned long long)s->H * s->W * s->v_stride * 4ull >= (1ull << 31)) return 0;
  if (s->col_tile != 32 || s->Co % 16 != 0) return 0;
  return (f->Cin == 96 && s->Np == 192) ? 1 : 0;
}

int conv_sub1_bf16_wide_fwd_launch(const sininn_conv_args* f, const sininn_conv_args* s, hipStream_t st) {
  SININN_CHECK(conv_sub1_bf16_wide_fwd_supported(f, s), "conv_sub1_bf16_wide_fwd: unsupported subnet");
  Sub1DevB q = {};
  alignas(16) static float dummy[8] = {};
  ConvDevB t;
  sininn_conv_args fa = *f;
  fa.out = dummy; fa.out_stride = S1_HID; fa.out_bf16 = 1;
  if (int e = conv_bf16_prepare(&fa, t)) return e;
  q.r = t.c; q.r.in = f->in; q.w1f = t.w;
  sininn_conv_args sa = *s;
  sa.in = dummy; sa.in_stride = S1_HID;
  if (int e = conv_bf16_prepare(&sa, t)) return e;
  q.b = t.c; q.a = t.c; q.w2 = t.w; q.w1d = nullptr;
  q.r.tiles_x = q.a.tiles_x = q.b.tiles_x = (f->W + 15) / 16;
  q.r.tiles_y = q.a.tiles_y = q.b.tiles_y = (f->H + 1) / 2;
  q.ntiles = q.r.tiles_x * q.r.tiles_y * f->B;
  q.r.stamp = q.a.stamp = q.b.stamp = nullptr;
  q.no_dx = 0; q.slab = nullptr;
  return sub1b_wide_launch<96, 192, 16>(q, reinterpret_cast<__bf16*>(f->out), f->out ? f->out_stride : 0, st);
}

// the backward pair of a wide subnet (d2: dr fp32 -> dh bf16 with a bf16 ReLU mask; d1: dh -> dx, ADD / ADD_CBWD epilogue)
int conv_sub1_bf16_wide_bwd_supported(const sininn_conv_args* d2, const sininn_conv_args* d1) {
  if (!g_sub1b_enabled || !g_sub1b_wide || !d2 || !d1) return 0;
  if (d2->ksize != 1 || d1->ksize != 1 || !d2->w_bf16 || !d1->w_bf16 || d2->winograd || d1->winograd) return 0;
  if (d2->in_bf16 || !d2->out_bf16 || !d1->in_bf16 || d1->out_bf16) return 0;
  if (d2->mode != SININN_CONV_MASK || !d2->mask || !d2->mask_bf16 || d2->mask_stride % 8 != 0) return 0;
  if (d2->in_group_stride > 0 || d2->out_group_stride > 0 || d2->mask_group_stride > 0 || d1->in_group_stride > 0) return 0;
  if (d2->Np != S1_HID || d2->N != S1_HID || d1->Cin != S1_HID) return 0;
  if (d2->out && d2->out_stride % 8 != 0) return 0;
  if (d2->B != d1->B || d2->H != d1->H || d2->W != d1->W) return 0;
  const bool cbwd = d1->mode == SININN_CONV_ADD_CBWD_FWD || d1->mode == SININN_CONV_ADD_CBWD_INV;
  if (!(cbwd || d1->mode == SININN_CONV_ADD)) return 0;
  if (d1->mask || d1->out_map || !d1->out || !d1->addend || d1->out_stride % 4 != 0 || !aligned16(d1->out)) return 0;
  if (!d1->addend_map && (d1->addend_stride % 4 != 0 || !aligned16(d1->addend))) return 0;
  if (d1->bias && !aligned16(d1->bias)) return 0;
  if (cbwd && !(d1->Co % 4 == 0 && d1->v_stride % 4 == 0 && d1->out2_stride % 4 == 0 && d1->v && d1->sbuf && d1->out2 && aligned16(d1->v) &&
                aligned16(d1->sbuf) && aligned16(d1->out2))) return 0;
  const unsigned long long px = (unsigned long long)d2->H * d2->W;
  if (px * d2->in_stride * 4ull >= (1ull << 31) || px * d2->mask_stride * 2ull >= (1ull << 31) || px * (d2->out ? d2->out_stride : 0) * 2ull >= (1ull << 31)) return 0;
  if (px * (unsigned long long)(d1->addend_stride > d1->out_stride ? d1->addend_stride : d1->out_stride) * 4ull >= (1ull << 31)) return 0;
  return (d2->Cin == 192 && d1->N == 96 && d1->Np == 96) ? 1 : 0;
}

// x / x_stride / ws: non-null -> the weight gradient of conv1 rides along (slabs into ws, summed by conv_sub1_bf16_wide_reduce) and
// d2->out may be NULL (dh is then not stored)
size_t conv_sub1_bf16_wide_ws_bytes(int ksize, int dtype, int cond_cin, int co) {
  if (!g_sub1b_enabled || !g_sub1b_wide || ksize != 1 || dtype != 1 || cond_cin != 96 || co != 96) return 0;
  return (size_t)S1_MAX_BLOCKS * (96 + 1) * S1_HID * sizeof(float);
}

static int wide_bwd_launch(const sininn_conv_args* d2, const sininn_conv_args* d1, const float* x, int x_stride, void* ws, size_t ws_bytes,
                           int* slabs_out, hipStream_t st) {
  SININN_CHECK(conv_sub1_bf16_wide_bwd_supported(d2, d1), "conv_sub1_bf16_wide_bwd: unsupported pair");
  const bool wg1 = x != nullptr;
  if (wg1) {
    SININN_CHECK(ws && aligned16(ws) && ws_bytes >= conv_sub1_bf16_wide_ws_bytes(1, 1, 96, 96) && slabs_out, "conv_sub1_bf16_wide_bwd: workspace too small");
    SININN_CHECK(aligned16(x) && x_stride % 4 == 0 && (unsigned long long)d2->H * d2->W * x_stride * 4ull < (1ull << 31), "conv_sub1_bf16_wide_bwd: bad x");
  }
  Sub1DevB q = {};
  alignas(16) static float dummy[8] = {};
  ConvDevB t;
  sininn_conv_args da = *d2;
  da.mode = SININN_CONV_LINEAR; da.mask = nullptr; da.mask_bf16 = 0; da.out = dummy; da.out_stride = S1_HID; da.N = S1_HID; da.out_bf16 = 1;
  if (int e = conv_bf16_prepare(&da, t)) return e;
  q.a = t.c; q.a.in = static_cast<const float*>(d2->in); q.w2 = t.w;
  sininn_conv_args db = *d1;
  db.in = dummy; db.in_stride = S1_HID; db.in_bf16 = 1;
  if (int e = conv_bf16_prepare(&db, t)) return e;
  q.b = t.c; q.r = t.c; q.w1d = t.w; q.w1f = nullptr;
  q.r.tiles_x = q.a.tiles_x = q.b.tiles_x = (d2->W + 15) / 16;
  q.r.tiles_y = q.a.tiles_y = q.b.tiles_y = (d2->H + 1) / 2;
  q.ntiles = q.a.tiles_x * q.a.tiles_y * d2->B;
  q.r.stamp = q.a.stamp = q.b.stamp = nullptr;
  q.no_dx = 0; q.slab = static_cast<float*>(ws);
  constexpr size_t lds0 = (size_t)2 * 32 * (192 * 2 + 16) + (size_t)(32 + 96) * (S1_HID * 2 + 16) + (size_t)2 * 32 * (96 + 4) * 4;
  constexpr size_t lds1 = lds0 + (size_t)2 * 96 * (32 * 2 + 8);
  static_assert(lds1 <= 160 * 1024, "conv_sub1b_wide_bwd: LDS");
  const int blocks = q.ntiles < S1_MAX_BLOCKS ? q.ntiles : S1_MAX_BLOCKS;
  const __bf16* hm = reinterpret_cast<const __bf16*>(d2->mask);
  __bf16* dho = reinterpret_cast<__bf16*>(d2->out);
  const int dhs_ = d2->out ? d2->out_stride : 0;
  if (wg1) {
    auto k = conv_sub1b_wide_bwd_kernel<96, 192, true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    if (e != hipSuccess) { set_error("conv_sub1_bf16_wide_bwd: cannot raise the LDS limit to %zu", lds1); return 1; }
    hipLaunchKernelGGL(k, dim3(blocks), dim3(S1_NTHR), lds1, st, q, hm, d2->mask_stride, dho, dhs_, x, x_stride);
    *slabs_out = blocks;
  } else {
    auto k = conv_sub1b_wide_bwd_kernel<96, 192, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0);
    if (e != hipSuccess) { set_error("conv_sub1_bf16_wide_bwd: cannot raise the LDS limit to %zu", lds0); return 1; }
    hipLaunchKernelGGL(k, dim3(blocks), dim3(S1_NTHR), lds0, st, q, hm, d2->mask_stride, dho, dhs_, static_cast<const float*>(nullptr), 0);
  }
  SININN_LAUNCH_CHECK("conv_sub1_bf16_wide_bwd");
  return 0;
}

int conv_sub1_bf16_wide_bwd_launch(const sininn_conv_args* d2, const sininn_conv_args* d1, hipStream_t st) {
  return wide_bwd_launch(d2, d1, nullptr, 0, nullptr, 0, nullptr, st);
}

int conv_sub1_bf16_wide_bwd_wg1_launch(const sininn_conv_args* d2, const sininn_conv_args* d1, const float* x, int x_stride, void* ws, size_t ws_bytes,
                                       int* slabs_out, hipStream_t st) {
  SININN_CHECK(x != nullptr, "conv_sub1_bf16_wide_bwd_wg1: x is NULL");
  return wide_bwd_launch(d2, d1, x, x_stride, ws, ws_bytes, slabs_out, st);
}

int conv_sub1_bf16_wide_reduce(const void* ws, int slabs, float* gw1, float* gb1, hipStream_t st) {
  SININN_CHECK(ws && slabs > 0 && slabs <= S1_MAX_BLOCKS, "conv_sub1_bf16_wide_reduce: bad arguments");
  constexpr int ET = (96 + 1) * S1_HID / 4;
  hipLaunchKernelGGL((wide_reduce_kernel<96>), dim3((ET + 31) / 32), dim3(256), 0, st, static_cast<const float*>(ws), slabs, gw1, gb1);
  SININN_LAUNCH_CHECK("conv_sub1_bf16_wide_reduce");
  return 0;
}

// conv2's weight gradient of a wide subnet: dr fp32 [pixel][192], h bf16 [pixel][256] -> slabs in ws; then the reduce
size_t conv_sub1_bf16_wide_wg2_ws_bytes(int ksize, int dtype, int cond_cin, int co) {
  if (!g_sub1b_enabled || !g_sub1b_wide || ksize != 1 || dtype != 1 || cond_cin != 96 || co != 96) return 0;
  return (size_t)S1_MAX_BLOCKS * (192 + 1) * S1_HID * sizeof(float);
}

int conv_sub1_bf16_wide_wg2_launch(const float* dr, int dr_stride, const void* h, int h_stride, int B, int H, int W, void* ws, size_t ws_bytes,
                                   float* gw2, float* gb2, hipStream_t st) {
  SININN_CHECK(dr && h && ws && aligned16(dr) && aligned16(h) && aligned16(ws) && dr_stride % 4 == 0 && h_stride % 8 == 0 && B > 0 && H > 0 && W > 0,
               "conv_sub1_bf16_wide_wg2: bad arguments");
  SININN_CHECK(ws_bytes >= (size_t)S1_MAX_BLOCKS * (192 + 1) * S1_HID * sizeof(float), "conv_sub1_bf16_wide_wg2: workspace too small");
  SININN_CHECK((unsigned long long)H * W * dr_stride * 4ull < (1ull << 31) && (unsigned long long)H * W * h_stride * 2ull < (1ull << 31),
               "conv_sub1_bf16_wide_wg2: one image of an operand exceeds the 2 GB a block addresses");
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 1) / 2, ntiles = tiles_x * tiles_y * B;
  const int blocks = ntiles < S1_MAX_BLOCKS ? ntiles : S1_MAX_BLOCKS;
  constexpr size_t lds = (size_t)2 * 192 * (32 * 2 + 8);
  auto k = conv_sub1b_wide_wg2_kernel<192>;
  hipLaunchKernelGGL(k, dim3(blocks), dim3(S1_NTHR), lds, st, dr, dr_stride, reinterpret_cast<const __bf16*>(h), h_stride, B, H, W, tiles_x, tiles_y,
                     ntiles, static_cast<float*>(ws));
  SININN_LAUNCH_CHECK("conv_sub1_bf16_wide_wg2");
  constexpr int ET = (192 + 1) * S1_HID / 4;
  hipLaunchKernelGGL((wide_reduce2_kernel<192>), dim3((ET + 31) / 32), dim3(256), 0, st, static_cast<const float*>(ws), blocks, gw2, gb2);
  SININN_LAUNCH_CHECK("conv_sub1_bf16_wide_reduce2");
  return 0;
}

}  // namespace sininn

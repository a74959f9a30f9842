// The whole backward pass of a 1x1 conv subnet of a GLOW half-coupling in ONE persistent launch (reference: subnet_conv_1x1,
// archs.py:15-17, differentiated inside FrEIA's GLOWCouplingBlock, archs.py:56-64):
//     h   = relu(x W1^T + b1)                   recomputed from the subnet's input x -- the forward pass did not store it
//     dW2 += h^T dr      db2 += sum_p dr        weight / bias gradient of conv2
//     dh  = (dr W2) . [h > 0]                   data gradient of conv2 with the ReLU mask, never leaves the chip
//     dx  = dh W1  (+ skip gradient / + fused coupling backward of the other half: the shared epilogue)
//     dW1 += dh^T x      db1 += sum_p dh        weight / bias gradient of conv1
// The five-launch path it replaces (fused data-gradient pair + grouped weight-gradient kernel + slab reduce, with the forward
// pass storing h) moves the 256-channel hidden tensors through HBM five times per half-coupling -- h written by the forward
// pass, re-read as the ReLU mask and by conv2's weight gradient, dh written and re-read by conv1's weight gradient: 335 MB at
// BASELINE configs[1], level 0, against 31 MB of x / dr / dx -- and its 1x1 classes sat at 0.17 - 0.32 of the f32 matrix pipe.
//
// Block = 512 threads (8 waves, one block per CU), persistent over 64-pixel tiles (4 x 16 pixels); the wave's weight fragments of
// stage R / stage 2 live in registers for the lifetime of the block, a tile's x / dr are requested while the previous tile is being
// processed, the side inputs of its epilogue at its start.  Per tile, on v_mfma_f32_16x16x4_f32:
//   stage 0  x tile [32][K1] and dr tile [32][K2] -> LDS (raw buffer loads, zero fill outside the image)
//   stage R  h  = relu(x W1^T + b1): wave w owns hidden columns [32 w, 32 w + 32) -- the SAME k order and accumulator split as
//            stage 1 of conv_pair_k1_kernel, so the recomputed h is bitwise the h the forward pass used
//   stage W2 dW2[n][c] += sum_p dr[p][n] h[p][c]: rows = the wave's 32 hidden channels, one 8-byte LDS read per lane supplies the
//            A operands of two row tiles (row i of tile t <-> channel 32 w + 2 i + t), k = 4 pixels per MFMA; accumulators stay
//            in registers for the lifetime of the block
//   stage 2  dh = (dr W2) . [h > 0] written IN PLACE over h (an element's mask is the element it overwrites, and a wave only ever
//            touches its own 32 columns of the tile in stages R / W2 / 2: no block barrier between them)
//   stage 3  dx tile = dh W1 (K = 256; W1's data-gradient pack lives in LDS for the lifetime of the block) -> the shared epilogue
//            (conv_epilogue_tile: ADD / ADD_CBWD_*), identical arithmetic to stage 2 of conv_pair_k1_kernel
//   stage W1 dW1[n][c] += sum_p dh[p][n] x[p][c]; the x tile carries a column of ones behind its K1 channels, so db1 falls out of
//            the same MFMAs
// At the end a block writes its partial gradients as ONE slab; sub1_reduce_kernel sums the slabs in a fixed order into the OIHW
// gradients (bitwise reproducible, no float atomics) -- on the weight-gradient stream, like every other += into a gradient.
#include <cstdlib>
#include "conv_sub1_types.h"

namespace sininn {

int conv_prepare(const sininn_conv_args* a, ConvDev& d);

template <int K1, int K2, bool STAMP>
__global__ __launch_bounds__(S1_NTHR) void conv_sub1_bwd_kernel(Sub1Dev q) {
  using SH = Sub1Shape<K1, K2>;
  constexpr int P = S1_P, MT = 4, HS = S1_HS, DS = SH::DS, XS = SH::XS, NU1 = SH::NU1, NU2 = SH::NU2, NP1 = SH::NP1, NT2 = SH::NT2;
  constexpr int NS1 = SH::K1R / 16, NS2 = K2 / 16, NTHR = S1_NTHR;
  static_assert(K1 % 8 == 0 && K1 <= 24 && K2 % 16 == 0 && K2 <= 48, "conv_sub1_bwd: shape");
  const ConvDev& pr = q.r;
  const ConvDev& pa = q.a;
  const ConvDev& pb = q.b;
  extern __shared__ __attribute__((aligned(16))) float smem_sub1[];
  float* const hs = smem_sub1;                      // [P][HS]: h, then dh
  float* const drs0 = hs + P * HS;                  // 2 x [P][DS]  (double-buffered: the next tile is staged while this one computes)
  float* const xs0 = drs0 + 2 * P * DS;             // 2 x [P][XS]: x | 1 | 0...
  float* const wd = xs0 + 2 * P * XS;               // [NP1][HS]: W1 data-gradient pack
  float* const T = wd + NP1 * HS;                   // [P][NP1 + 4]: the dx tile on its way to the epilogue

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, kq = lane >> 4;
  const int cw = wave * 32;                         // this wave's hidden columns

  // phase stamps (diagnostic build of the kernel, STAMP: d1->stamp -> 2 x 16 words from thread 0 / thread 448 -- wave 0 / wave 7:
  // 0 top of tile, 1 stage R, 2 stage W2, 3 stage 2, 4 stage 3, 5 stage W1, 6 epilogue, 7 slab write, 8 total, 9 tiles, 10 next
  // tile -> LDS (incl. the wait for its loads), 11 requests, 12 barrier C, 13 T write, 14 barrier E).  Deltas are summed in
  // registers and written once at the end of the block: an atomic per phase sits in the same in-order counter as the loads whose
  // latency is being measured (tools/bench_sub1.py)
  const bool stamping = STAMP && pb.stamp != nullptr && (tid == 0 || tid == 448);
  unsigned ph[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) ph[k] = 0u;
  unsigned long long tprev = stamping ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long tstart = tprev;
  auto mark = [&](int k) {
    if constexpr (STAMP) {
      if (stamping) { const unsigned long long t = __builtin_amdgcn_s_memtime(); ph[k] += (unsigned)(t - tprev); tprev = t; }
    }
  };

  // ---- once per block: W1's data-gradient pack -> LDS; the wave's weight fragments of stage R and stage 2 -> registers -------
  if (!q.no_dx) {
    for (int f = tid; f < NP1 * (S1_HID / 4); f += NTHR) {
      const int n = f / (S1_HID / 4), c = (f - n * (S1_HID / 4)) * 4;
      *reinterpret_cast<f32x4*>(wd + n * HS + c) = *reinterpret_cast<const f32x4*>(pb.w + (size_t)n * S1_HID + c);
    }
  }
  float b1q[2];                                     // bias of the wave's columns cw + 16 n + li
#pragma unroll
  for (int n = 0; n < 2; ++n) b1q[n] = pr.bias ? pr.bias[cw + 16 * n + li] : 0.f;
  f32x4 bf1[NS1][2], bf2[NS2][2];                   // B operands: W[column cw + 16 n + li][16 s + 4 kq .. + 3]
  {
    const __amdgpu_buffer_rsrc_t w1_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pr.w), 0, S1_HID * K1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pa.w), 0, S1_HID * K2 * 4, 0x00020000);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const unsigned o1 = (unsigned)(((cw + n * 16 + li) * K1 + 4 * kq) * 4), o2 = (unsigned)(((cw + n * 16 + li) * K2 + 4 * kq) * 4);
#pragma unroll
      for (int s = 0; s < NS1; ++s) bf1[s][n] = buf_load4(w1_rs, 16 * s + 4 * kq < K1 ? o1 : BUF_OOB, (unsigned)(64 * s));
#pragma unroll
      for (int s = 0; s < NS2; ++s) bf2[s][n] = buf_load4(w2_rs, o2, (unsigned)(64 * s));
    }
  }

  // ---- gradient accumulators of the block (registers) ----------------------------------------------------------------------
  f32x4 accW2[2][NU2], accW1[2][NU1];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int u = 0; u < NU2; ++u) accW2[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NU1; ++u) accW1[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float accb2 = 0.f;

  // ---- staging slots of a thread (the same for every tile) and the global loads of a tile ------------------------------------
  constexpr int QX = K1 / 4, QD = K2 / 4;
  constexpr int FX = (P * QX + NTHR - 1) / NTHR, FD = (P * QD + NTHR - 1) / NTHR;
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

  // ---- the epilogue of stage 3 on one quad per thread (pixel tid / 8, columns 4 (tid % 8) ..): the modes the block executor
  // uses, with every global side input requested at the START of the tile.  Arithmetic and stores are conv_epilogue_tile's.
  const int e_pl = tid >> 3, e_q = tid & 7, e_col = 4 * e_q;
  const int emode = pb.mode;
  const bool e_cbwd = emode == SININN_CONV_ADD_CBWD_FWD || emode == SININN_CONV_ADD_CBWD_INV;
  const bool e_fast = !q.no_dx;                     // alignment / mode requirements of the quad epilogue: checked on the host
  int amap[4] = {e_col, e_col + 1, e_col + 2, e_col + 3};
  f32x4 e_bq = {0.f, 0.f, 0.f, 0.f};
  if (e_fast && e_col < pb.N) {
    if (pb.addend_map) {
#pragma unroll
      for (int j = 0; j < 4; ++j) amap[j] = pb.addend_map[e_col + j];
    }
    if (pb.bias) e_bq = *reinterpret_cast<const f32x4*>(pb.bias + e_col);
  }

  // regs -> LDS of a staged tile (+ the column of ones / zeros behind the K1 channels: a 1 for pixels inside the image, so that
  // db1 = sum_p dh rides on the MFMAs of stage W1)
  auto store_tile = [&](int tile, int buf, const f32x4 (&vx)[FX], const f32x4 (&vd)[FD]) {
    float* const xs = xs0 + buf * (P * XS);
    float* const drs = drs0 + buf * (P * DS);
#pragma unroll
    for (int u = 0; u < FX; ++u) {
      const int f = tid + NTHR * u;
      const int pl = f / QX, c = (f - pl * QX) * 4;
      if (f < P * QX) *reinterpret_cast<f32x4*>(xs + pl * XS + c) = vx[u];
    }
#pragma unroll
    for (int u = 0; u < FD; ++u) {
      const int f = tid + NTHR * u;
      const int pl = f / QD, c = (f - pl * QD) * 4;
      if (f < P * QD) *reinterpret_cast<f32x4*>(drs + pl * DS + c) = vd[u];
    }
    constexpr int QP = (SH::XD - K1) / 4;
    if (tid < P * QP) {
      const int b = tile / tiles_img;
      const int trem = tile - b * tiles_img;
      const int ty = trem / pr.tiles_x, tx = trem - ty * pr.tiles_x;
      const int pl = tid / QP, c = K1 + (tid - pl * QP) * 4;
      const int gy = ty * 4 + (pl >> 4), gx = tx * 16 + (pl & 15);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c == K1 && gy < pr.H && gx < pr.W) v[0] = 1.f;
      *reinterpret_cast<f32x4*>(xs + pl * XS + c) = v;
    }
  };

  // ---- software pipeline over the block's tiles: tile t computes from LDS buffer t & 1 while tile t + 1 (requested during tile
  // t - 1) is written into the other buffer and tile t + 2 is requested; two block barriers per tile ------------------------------
  f32x4 vx[FX], vd[FD];
  issue_tile(blockIdx.x, vx, vd);
  store_tile(blockIdx.x, 0, vx, vd);
  issue_tile(blockIdx.x + gridDim.x, vx, vd);
  __syncthreads();
  int buf = 0;
  for (int tile = blockIdx.x; tile < q.ntiles; tile += gridDim.x, buf ^= 1) {
    const int b = tile / tiles_img;
    const int trem = tile - b * tiles_img;
    const int ty = trem / pr.tiles_x, tx = trem - ty * pr.tiles_x;
    const int y0 = ty * 4, x0 = tx * 16;
    const float* const xs = xs0 + buf * (P * XS);
    const float* const drs = drs0 + buf * (P * DS);
    mark(0);

    // ---- stage R: h[P][32 of this wave] = relu(x W1^T + b1) -> hs (one row tile at a time; every accumulator sees the k-steps in
    // the order conv_pair_k1_kernel feeds them, so this h is bitwise the h of the forward pass) ---------------------------------
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      f32x4 accs[2][2];
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int n = 0; n < 2; ++n) accs[k][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NS1; ++s) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(xs + (m * 16 + li) * XS + 16 * s + 4 * kq);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            accs[j % 2][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf1[s][n][j], accs[j % 2][n], 0, 0, 0);
      }
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = (accs[0][n][r] + accs[1][n][r]) + b1q[n];
          hs[(m * 16 + 4 * kq + r) * HS + cw + n * 16 + li] = fmaxf(v, 0.f);
        }
    }
    mark(1);

    // ---- stage W2: dW2[n][c] += sum_p dr[p][n] h[p][c] over this wave's 32 channels (wave-private columns of hs): one 8-byte LDS
    // read per lane supplies the A operands of two row tiles (row i of tile t <-> channel cw + 2 i + t), k = 4 pixels per MFMA ----
#pragma unroll 4
    for (int ks = 0; ks < P / 4; ++ks) {
      const f32x2 a2 = *reinterpret_cast<const f32x2*>(hs + (4 * ks + kq) * HS + cw + 2 * li);
      float bu[NU2];
#pragma unroll
      for (int u = 0; u < NU2; ++u) bu[u] = drs[(4 * ks + kq) * DS + 16 * u + li];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < NU2; ++u)
          accW2[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[t], bu[u], accW2[t][u], 0, 0, 0);
    }
    {   // db2 partial: thread (column tid % 64, pixel group tid / 64)
      const int col = tid & 63;
      if (col < K2) {
#pragma unroll
        for (int i = 0; i < P / 8; ++i) accb2 += drs[(wave * (P / 8) + i) * DS + col];
      }
    }
    mark(2);

    // ---- stage 2: dh = (dr W2) . [h > 0], in place over this wave's columns of hs ----------------------------------------
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      f32x4 accs[2][2];
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int n = 0; n < 2; ++n) accs[k][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NS2; ++s) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(drs + (m * 16 + li) * DS + 16 * s + 4 * kq);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            accs[j % 2][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf2[s][n][j], accs[j % 2][n], 0, 0, 0);
      }
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* const e = hs + (m * 16 + 4 * kq + r) * HS + cw + n * 16 + li;
          const float v = accs[0][n][r] + accs[1][n][r];
          *e = *e > 0.f ? v : 0.f;
        }
    }
    mark(3);
    // side inputs of this tile's epilogue: requested HERE -- half a tile after the previous tile's epilogue stores and half a tile
    // before they are used.  One counter covers vector loads and stores on this chip and a register that is the data or address
    // of a store in flight may not be overwritten before the store has completed: requested at the top of the tile, these loads
    // waited for the previous epilogue's stores (9 k clocks per tile in the phase stamps).  Raw buffer loads relative to the
    // image (a quad outside the image / beyond N carries BUF_OOB and reads zeros): no branch around a load.
    const int e_gy = y0 + (e_pl >> 4), e_gx = x0 + (e_pl & 15);
    const bool e_live = e_fast && e_col < pb.N && e_gy < pb.H && e_gx < pb.W;
    const size_t e_img = (size_t)b * pb.H * pb.W;
    const unsigned e_ip = (unsigned)(e_gy * pb.W + e_gx);            // pixel inside the image
    const size_t e_pix = e_img + e_ip;
    f32x4 e_ad, e_u, e_s;
    {
      const __amdgpu_buffer_rsrc_t ad_rs = buf_rsrc(pb.addend + e_img * pb.addend_stride);
      if (pb.addend_map) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          e_ad[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ad_rs, (int)(e_live ? (e_ip * pb.addend_stride + amap[j]) * 4u : BUF_OOB), 0, 0));
      } else {
        e_ad = buf_load4(ad_rs, e_live ? (e_ip * pb.addend_stride + e_col) * 4u : BUF_OOB, 0u);
      }
      const bool cb = e_live && e_cbwd;
      e_u = buf_load4(buf_rsrc((e_cbwd ? pb.v : pb.addend) + e_img * (e_cbwd ? pb.v_stride : 0)), cb ? (e_ip * pb.v_stride + e_col) * 4u : BUF_OOB, 0u);
      e_s = buf_load4(buf_rsrc((e_cbwd ? pb.sbuf : pb.addend) + e_img * (e_cbwd ? pb.Co : 0)), cb ? (e_ip * pb.Co + e_col) * 4u : BUF_OOB, 0u);
    }
    __builtin_amdgcn_sched_barrier(0);               // ... and before the next tiles' x / dr: the counter is in order, so waiting for the
                                                     // side inputs does not wait for those loads
    // the next tile (requested one tile ago) -> the other LDS buffer; the tile after it is requested
    if (tile + (int)gridDim.x < q.ntiles) {
      store_tile(tile + gridDim.x, buf ^ 1, vx, vd);
      mark(10);
      issue_tile(tile + 2 * gridDim.x, vx, vd);
    }
    __builtin_amdgcn_sched_barrier(0);               // the requests above stay above: their latency is what stage 3 / W1 hide
    mark(11);
    __syncthreads();                                 // (C) the whole dh tile is in LDS (and the next tile's x / dr)
    mark(12);

    // ---- stage 3: dx tile = dh W1 (K = 256): wave -> row tile wave % 4, column tile wave / 4; the k order of conv_pair_k1_kernel ---
    const int mt3 = wave & 3, nt3 = wave >> 2;
    f32x4 acc3[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc3[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!q.no_dx && nt3 < NT2) {
#pragma unroll 4
      for (int s = 0; s < S1_HID / 16; ++s) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(hs + (mt3 * 16 + li) * HS + 16 * s + 4 * kq);
        const f32x4 bf = *reinterpret_cast<const f32x4*>(wd + (nt3 * 16 + li) * HS + 16 * s + 4 * kq);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc3[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], acc3[j], 0, 0, 0);
      }
    }
    mark(4);
    // ---- stage W1: dW1[n][c] += sum_p dh[p][n] [x | 1][p][c] over this wave's 32 hidden channels ---------------------------
#pragma unroll 4
    for (int ks = 0; ks < P / 4; ++ks) {
      const f32x2 a2 = *reinterpret_cast<const f32x2*>(hs + (4 * ks + kq) * HS + cw + 2 * li);
      float bu[NU1];
#pragma unroll
      for (int u = 0; u < NU1; ++u) bu[u] = xs[(4 * ks + kq) * XS + 16 * u + li];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < NU1; ++u)
          accW1[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[t], bu[u], accW1[t][u], 0, 0, 0);
    }
    mark(5);
    if (!q.no_dx) {
      constexpr int TS = NP1 + 4;
      if (nt3 < NT2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc3[0][r];
#pragma unroll
          for (int k = 1; k < 4; ++k) v += acc3[k][r];
          T[(mt3 * 16 + 4 * kq + r) * TS + nt3 * 16 + li] = v;
        }
      }
    }
    mark(13);
    __syncthreads();                                 // (E) the dx tile is in T; every wave is done with hs and this tile's x / dr
    mark(14);
    // the side inputs are waited for HERE, by every wave and before any store is issued: a wave that skipped the epilogue below
    // (no live quad) would otherwise carry the pending loads to the top of the next tile, where the first instruction that reuses
    // one of their registers has to wait for them -- and, the counter being shared and in order, for this tile's stores
    asm volatile("" :: "v"(e_ad), "v"(e_u), "v"(e_s));
    if (!q.no_dx) {
      constexpr int TS = NP1 + 4;
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
    mark(6);
    if (stamping) ph[9] += 1u;
  }

  // ---- the block's partial gradients -> its slab ------------------------------------------------------------------------------
  float* const slab = q.slab + (size_t)blockIdx.x * SH::SLAB;
  // lane (li, kq), row tile t, register r of dW2's tile u holds dW2[n = 16 u + li][c = cw + 8 kq + 2 r + t]: the two t are two
  // consecutive channels of one OIHW row
#pragma unroll
  for (int u = 0; u < NU2; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x2 v = {accW2[0][u][r], accW2[1][u][r]};
      *reinterpret_cast<f32x2*>(slab + (16 * u + li) * S1_HID + cw + 8 * kq + 2 * r) = v;
    }
  float* const slab1 = slab + K2 * S1_HID;
  // ... and of dW1's tile u: dW1[n = cw + 8 kq + 2 r + t][c = 16 u + li], written TRANSPOSED as [c][n] so that the two t are again
  // two consecutive floats (8-byte stores like dW2's; the reduce kernel puts the 256 x K1 matrix back into OIHW order)
#pragma unroll
  for (int u = 0; u < NU1; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x2 v = {accW1[0][u][r], accW1[1][u][r]};
      *reinterpret_cast<f32x2*>(slab1 + (16 * u + li) * S1_HID + cw + 8 * kq + 2 * r) = v;
    }
  // db2: the eight pixel groups of a column in a fixed order
  __syncthreads();
  hs[tid] = accb2;
  __syncthreads();
  if (tid < 64)
    slab[K2 * S1_HID + S1_HID * SH::W1S + tid] =
        tid < K2 ? (((hs[tid] + hs[64 + tid]) + (hs[128 + tid] + hs[192 + tid])) + ((hs[256 + tid] + hs[320 + tid]) + (hs[384 + tid] + hs[448 + tid]))) : 0.f;
  if constexpr (STAMP) {
    if (stamping) {
      mark(7);
      ph[8] = (unsigned)(tprev - tstart);
      unsigned long long* const w = pb.stamp + (tid == 0 ? 0 : 16);
#pragma unroll
      for (int k = 0; k < 16; ++k) atomicAdd(w + k, (unsigned long long)ph[k]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The FORWARD of the same subnet + affine coupling + log-det, persistent (north_star's fused coupling kernel for the 1x1 subnets):
//     h = relu(x W1^T + b1)  ->  (s | t) = h W2^T + b2  ->  y = e(s) v + t  (inverse: (v - t) / e(s)),  log-det += sum log e(s)
// The pair kernel it replaces at these shapes (conv_pair_k1_kernel, 2048 blocks of 32 pixels) spends 58 - 62 us per launch on
// 15 us of MFMA work: every block walks K = 256 of the second GEMM through a chain of L2 round trips for the weight fragments.
// Here a block (512 threads, one per CU) keeps W2's pack in LDS and its W1 fragments in registers for all of its 64-pixel tiles:
//   stage 1  h[64][32 of this wave] -> hs: the code of stage R above (bitwise the h the backward recomputes)
//   stage 2  K = 256 split in two halves over the two wave quads; wave (g, m): row tile m, all column tiles, K half g;
//            A = hs (16-byte reads), B = W2 pack rows from LDS; the two partial tiles T0 / T1 are summed by the shared
//            coupling epilogue (conv_epilogue_tile's T2 operand): same fixed order every run
// The next tile's x is requested one tile ahead and written into the other half of a double buffer; two block barriers per tile.
template <int K1, int N2, int HT>
__global__ __launch_bounds__(S1_NTHR) void conv_sub1_fwd_kernel(Sub1Dev q) {
  constexpr int P = S1_P, MT = 4, HS = S1_HS, NTHR = S1_NTHR;
  constexpr int K1R = (K1 + 15) / 16 * 16, XS = K1R + 4, NS1 = K1R / 16, NU2 = N2 / 16, TS = N2 + 4;
  static_assert(K1 % 8 == 0 && K1 <= 24 && N2 % 16 == 0 && N2 <= 48, "conv_sub1_fwd: shape");
  const ConvDev& pr = q.r;                          // conv1: in = x, w = forward pack [256][K1], bias
  const ConvDev& pb = q.b;                          // conv2: w = forward pack [N2][256] (s | t interleaved) + the coupling epilogue
  extern __shared__ __attribute__((aligned(16))) float smem_sub1f[];
  float* const hs = smem_sub1f;                     // [P][HS]
  float* const xs0 = hs + P * HS;                   // 2 x [P][XS]
  float* const w2s = xs0 + 2 * P * XS;              // [N2][HS]
  float* const T0 = w2s + N2 * HS;                  // 2 x [P][TS]: the two K halves of the output tile
  float* const T1 = T0 + P * TS;
  __shared__ float ldw[2][NTHR / 64];               // per-wave log-det sums of the last two tiles
  int ld_b[2] = {0, 0};                             // ... and the image they belong to
  int ld_pending = -1;                              // buffer whose sums have not been added to logdet yet

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, kq = lane >> 4;
  const int cw = wave * 32;

  for (int f = tid; f < N2 * (S1_HID / 4); f += NTHR) {
    const int n = f / (S1_HID / 4), c = (f - n * (S1_HID / 4)) * 4;
    *reinterpret_cast<f32x4*>(w2s + n * HS + c) = *reinterpret_cast<const f32x4*>(pb.w + (size_t)n * S1_HID + c);
  }
  float b1q[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) b1q[n] = pr.bias ? pr.bias[cw + 16 * n + li] : 0.f;
  f32x4 bf1[NS1][2];
  {
    const __amdgpu_buffer_rsrc_t w1_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pr.w), 0, S1_HID * K1 * 4, 0x00020000);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const unsigned o1 = (unsigned)(((cw + n * 16 + li) * K1 + 4 * kq) * 4);
#pragma unroll
      for (int s = 0; s < NS1; ++s) bf1[s][n] = buf_load4(w1_rs, 16 * s + 4 * kq < K1 ? o1 : BUF_OOB, (unsigned)(64 * s));
    }
  }

  // ---- the coupling epilogue on one channel quad per thread (pixel tid / NQ, channels 4 (tid % NQ) ..): conv_epilogue_tile's
  // arithmetic and stores, with the one global side input (v) requested half a tile before it is used (the shared epilogue asks
  // for it after the barrier and waits out the round trip: 11 of this kernel's 41 us at BASELINE configs[1], level 0)
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
    float* const xs = xs0 + buf * (P * XS);
#pragma unroll
    for (int u = 0; u < FX; ++u) {
      const int f = tid + NTHR * u;
      const int pl = f / QX, c = (f - pl * QX) * 4;
      if (f < P * QX) *reinterpret_cast<f32x4*>(xs + pl * XS + c) = vx[u];
    }
    if constexpr (K1R > K1) {                        // the k-quads of stage 1 beyond K1 meet zero weights: they must hold finite values
      constexpr int QP = (K1R - K1) / 4;
      if (tid < P * QP) {
        const int pl = tid / QP, c = K1 + (tid - pl * QP) * 4;
        *reinterpret_cast<f32x4*>(xs + pl * XS + c) = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  f32x4 vx[FX];
  issue_tile(blockIdx.x, vx);
  store_tile(0, vx);
  issue_tile(blockIdx.x + gridDim.x, vx);
  __syncthreads();
  int buf = 0;
  const int g2 = wave >> 2, m2 = wave & 3;          // stage 2: K half, row tile
  for (int tile = blockIdx.x; tile < q.ntiles; tile += gridDim.x, buf ^= 1) {
    const int b = tile / tiles_img;
    const int trem = tile - b * tiles_img;
    const int ty = trem / pr.tiles_x, tx = trem - ty * pr.tiles_x;
    const int y0 = ty * 4, x0 = tx * 16;
    const float* const xs = xs0 + buf * (P * XS);

    // ---- stage 1 (== stage R of the backward kernel) ---------------------------------------------------------------------------
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      f32x4 accs[2][2];
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int n = 0; n < 2; ++n) accs[k][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NS1; ++s) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(xs + (m * 16 + li) * XS + 16 * s + 4 * kq);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            accs[j % 2][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf1[s][n][j], accs[j % 2][n], 0, 0, 0);
      }
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = (accs[0][n][r] + accs[1][n][r]) + b1q[n];
          hs[(m * 16 + 4 * kq + r) * HS + cw + n * 16 + li] = fmaxf(v, 0.f);
        }
    }
    // the next tile (requested one tile ago) -> the other buffer; the tile after it is requested
    // v of this tile's epilogue quad (raw buffer load relative to the image: a quad outside the image reads zeros and is not
    // stored).  Requested BEFORE the next tiles' x: the counter is in order, so waiting for v then does not wait for those loads
    const int e_gy = y0 + (e_pl >> 4), e_gx = x0 + (e_pl & 15);
    const bool e_live = e_thread && e_gy < pb.H && e_gx < pb.W;
    const size_t e_img = (size_t)b * pb.H * pb.W;
    const unsigned e_ip = (unsigned)(e_gy * pb.W + e_gx);
    const f32x4 e_v = buf_load4(buf_rsrc(pb.v + e_img * pb.v_stride), e_live ? (e_ip * pb.v_stride + e_cl) * 4u : BUF_OOB, 0u);
    __builtin_amdgcn_sched_barrier(0);
    if (tile + (int)gridDim.x < q.ntiles) {
      store_tile(buf ^ 1, vx);
      issue_tile(tile + 2 * gridDim.x, vx);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                 // (B) the whole h tile is in LDS (and the next tile's x)
    if (pb.logdet) {
      if (tid == 0 && ld_pending >= 0) {              // the previous tile's log-det: every wave has parked its sum before (B)
        const float* w8 = ldw[ld_pending];
        atomicAdd(pb.logdet + ld_b[ld_pending], ((w8[0] + w8[1]) + (w8[2] + w8[3])) + ((w8[4] + w8[5]) + (w8[6] + w8[7])));
      }
      ld_pending = buf;
    }

    // ---- stage 2: this wave's K half of out[row tile m2][all N2 columns] ---------------------------------------------------------
    f32x4 acc[NU2];
#pragma unroll
    for (int u = 0; u < NU2; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int s = 0; s < S1_HID / 32; ++s) {
      const int k0 = 128 * g2 + 16 * s + 4 * kq;
      const f32x4 af = *reinterpret_cast<const f32x4*>(hs + (m2 * 16 + li) * HS + k0);
      f32x4 bf[NU2];
#pragma unroll
      for (int u = 0; u < NU2; ++u) bf[u] = *reinterpret_cast<const f32x4*>(w2s + (u * 16 + li) * HS + k0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < NU2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[u][j], acc[u], 0, 0, 0);
    }
    {
      float* const T = g2 ? T1 : T0;
#pragma unroll
      for (int u = 0; u < NU2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(m2 * 16 + 4 * kq + r) * TS + u * 16 + li] = acc[u][r];
    }
    __syncthreads();                                 // (E) both partial tiles are in LDS; every wave is done with hs
    asm volatile("" :: "v"(e_v));                    // waited for by every wave, before any store of this tile is issued
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
    // log-det of the tile: the waves' sums are parked in LDS and added (fixed order) by ONE thread after the NEXT block barrier --
    // one global atomic per block and tile, as in the shared epilogue, without a barrier of its own.  (An atomic per wave was tried:
    // 8 192 same-address float atomics per launch serialise in L2 and doubled the kernel's duration.)
    if (pb.logdet) {
      const float wsum = wave_sum(ld_acc);
      if (lane == 0) ldw[buf][wave] = wsum;
      ld_b[buf] = b;
    }
  }
  if (pb.logdet) {
    __syncthreads();
    if (tid == 0 && ld_pending >= 0) {
      const float* w8 = ldw[ld_pending];
      atomicAdd(pb.logdet + ld_b[ld_pending], ((w8[0] + w8[1]) + (w8[2] + w8[3])) + ((w8[4] + w8[5]) + (w8[6] + w8[7])));
    }
  }
}

// gw2 / gb2 / gw1 / gb1 += sum over the slabs, in a fixed association: eight interleaved groups of slabs, each summed in
// ascending order by one thread, the eight partial sums added as a balanced tree.
template <int K1, int K2>
__global__ __launch_bounds__(256) void sub1_reduce_kernel(const float* __restrict__ slabs, int S, float* __restrict__ gw2,
                                                          float* __restrict__ gb2, float* __restrict__ gw1, float* __restrict__ gb1) {
  using SH = Sub1Shape<K1, K2>;
  constexpr int E2 = K2 * S1_HID / 4, E1 = S1_HID * SH::W1S / 4, EB = 16, ET = E2 + E1 + EB;
  __shared__ f32x4 part[8][32];
  const int tid = threadIdx.x, g = tid >> 5, el = tid & 31;
  const int e = blockIdx.x * 32 + el;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
  if (e < ET) {
    const f32x4* src = reinterpret_cast<const f32x4*>(slabs) + e;
    constexpr size_t slab4 = SH::SLAB / 4;
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
    if (e < E2) {
      if (gw2) {
        if ((reinterpret_cast<uintptr_t>(gw2) & 15) == 0) *reinterpret_cast<f32x4*>(gw2 + (size_t)e * 4) += tot;
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) gw2[(size_t)e * 4 + j] += tot[j];
        }
      }
    } else if (e < E2 + E1) {
      const int idx = (e - E2) * 4, c = idx / S1_HID, n = idx - c * S1_HID;     // slab row c = input channel (row K1: db1), 4 outputs n
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (c < K1) { if (gw1) gw1[(n + j) * K1 + c] += tot[j]; }
        else if (c == K1) { if (gb1) gb1[n + j] += tot[j]; }
      }
    } else if (gb2) {
      const int idx = (e - E2 - E1) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (idx + j < K2) gb2[idx + j] += tot[j];
    }
  }
}

// conv_sub1_bf16.hip: the mixed-precision twins (bf16 weight packs, fp32 tensors)
bool conv_sub1_bf16_enabled();
void conv_sub1_bf16_enable(int on);
int conv_sub1_bf16_bwd_launch(const sininn_conv_args* rc, const sininn_conv_args* d2, const sininn_conv_args* d1, int no_dx, void* ws,
                              size_t ws_bytes, int* slabs_out, hipStream_t st);
int conv_sub1_bf16_fwd_supported(const sininn_conv_args* f, const sininn_conv_args* s);
int conv_sub1_bf16_fwd_launch(const sininn_conv_args* f, const sininn_conv_args* s, hipStream_t st);

static bool g_sub1_enabled = getenv("SININN_SUB1_BWD") == nullptr || atoi(getenv("SININN_SUB1_BWD")) != 0;   // A/B switch
void conv_sub1_bwd_enable(int on) { g_sub1_enabled = on != 0; conv_sub1_bf16_enable(on); }
bool conv_sub1_bwd_enabled() { return g_sub1_enabled; }

static bool shape_ok(int k1, int k2) { return sub1_shape_ok(k1, k2); }


// Shapes the fused backward serves: a fp32 1x1 subnet with K1 = Cin of conv1 and K2 = 2 * Co columns of conv2 in {(8, 16),
// (16, 32), (24, 48)} -- level 0 of the SRF network (C = 48: 24 | 24) and the small networks of the tests.  Wider subnets (level 1: 96 / 192) would need
// 300 accumulator registers per lane for the two weight gradients and stay on the data-gradient pair + grouped weight gradients.
int conv_sub1_bwd_shape_supported(int ksize, int dtype, int cond_cin, int co) {
  const bool on = dtype == 0 ? g_sub1_enabled : (dtype == 1 && conv_sub1_bf16_enabled());
  return on && ksize == 1 && shape_ok(cond_cin, 2 * co);
}

size_t conv_sub1_bwd_workspace_bytes(int cond_cin, int co) {
  if (!shape_ok(cond_cin, 2 * co)) return 0;
  const int nu1 = (cond_cin + 16) / 16;
  return (size_t)S1_MAX_BLOCKS * ((size_t)2 * co * S1_HID + (size_t)S1_HID * 16 * nu1 + 64) * sizeof(float);
}

template <int K1, int K2>
static int sub1_launch(Sub1Dev& q, int* blocks_out, hipStream_t st) {
  using SH = Sub1Shape<K1, K2>;
  auto k = q.b.stamp ? conv_sub1_bwd_kernel<K1, K2, (K1 == 24)> : conv_sub1_bwd_kernel<K1, K2, false>;
  static_assert(SH::LDS <= 160 * 1024, "conv_sub1_bwd: LDS");
  if (SH::LDS > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SH::LDS);
    if (e != hipSuccess) { set_error("conv_sub1_bwd: cannot raise the LDS limit to %zu", SH::LDS); return 1; }
  }
  const int blocks = q.ntiles < S1_MAX_BLOCKS ? q.ntiles : S1_MAX_BLOCKS;
  hipLaunchKernelGGL(k, dim3(blocks), dim3(S1_NTHR), SH::LDS, st, q);
  SININN_LAUNCH_CHECK("conv_sub1_bwd");
  *blocks_out = blocks;
  return 0;
}

template <int K1, int K2>
static int sub1_reduce(const float* slabs, int S, float* gw2, float* gb2, float* gw1, float* gb1, hipStream_t st) {
  using SH = Sub1Shape<K1, K2>;
  constexpr int ET = K2 * S1_HID / 4 + S1_HID * SH::W1S / 4 + 16;
  hipLaunchKernelGGL((sub1_reduce_kernel<K1, K2>), dim3((ET + 31) / 32), dim3(256), 0, st, slabs, S, gw2, gb2, gw1, gb1);
  SININN_LAUNCH_CHECK("sub1_reduce");
  return 0;
}

// rc: the recompute conv (x -> h: in, in_stride, Cin, w = forward pack [256][Cin], bias); d2 / d1: the two data-gradient convs as
// the block executor describes them for the pair kernel.  Writes `*slabs_out` slabs into ws.
int conv_sub1_bwd_launch(const sininn_conv_args* rc, const sininn_conv_args* d2, const sininn_conv_args* d1, int no_dx, void* ws,
                         size_t ws_bytes, int* slabs_out, hipStream_t st) {
  SININN_CHECK(rc && d2 && d1 && ws && slabs_out, "conv_sub1_bwd: null argument");
  if (rc->w_bf16) return conv_sub1_bf16_bwd_launch(rc, d2, d1, no_dx, ws, ws_bytes, slabs_out, st);
  const int K1 = rc->Cin, K2 = d2->Cin;
  SININN_CHECK(shape_ok(K1, K2), "conv_sub1_bwd: unsupported subnet shape (Cin %d, 2 Co %d)", K1, K2);
  SININN_CHECK(rc->ksize == 1 && d2->ksize == 1 && d1->ksize == 1 && !rc->w_bf16 && !d2->w_bf16 && !d1->w_bf16 && !rc->winograd &&
               !d2->winograd && !d1->winograd && !rc->in_bf16 && !d2->in_bf16, "conv_sub1_bwd: fp32 1x1 convs only");
  SININN_CHECK(rc->Np == S1_HID && d2->Np == S1_HID && d1->Cin == S1_HID && d1->N == K1 && d1->Np == (K1 + 15) / 16 * 16,
               "conv_sub1_bwd: the three convs do not form a subnet backward");
  SININN_CHECK(rc->B == d2->B && rc->H == d2->H && rc->W == d2->W && rc->B == d1->B && rc->H == d1->H && rc->W == d1->W, "conv_sub1_bwd: shapes differ");
  SININN_CHECK(rc->in_stride % 4 == 0 && d2->in_stride % 4 == 0 && aligned16(rc->in) && aligned16(d2->in) && aligned16(d1->w) && aligned16(ws),
               "conv_sub1_bwd: operands must be 16-byte aligned with strides that are multiples of 4 floats");
  SININN_CHECK((unsigned long long)rc->H * rc->W * (rc->in_stride > d2->in_stride ? rc->in_stride : d2->in_stride) * 4ull < (1ull << 31),
               "conv_sub1_bwd: one image of an operand exceeds the 2 GB a block addresses (raw buffer staging)");
  SININN_CHECK(ws_bytes >= conv_sub1_bwd_workspace_bytes(K1, K2 / 2), "conv_sub1_bwd: workspace too small (%zu < %zu)", ws_bytes,
               conv_sub1_bwd_workspace_bytes(K1, K2 / 2));
  if (!no_dx) {
    // the kernel's own epilogue (one quad per thread, side inputs requested a tile ahead) serves the modes the block executor uses
    const bool cbwd = d1->mode == SININN_CONV_ADD_CBWD_FWD || d1->mode == SININN_CONV_ADD_CBWD_INV;
    SININN_CHECK(cbwd || d1->mode == SININN_CONV_ADD, "conv_sub1_bwd: d1->mode must be ADD or ADD_CBWD_* (got %d)", d1->mode);
    SININN_CHECK(d1->mask == nullptr && d1->out_map == nullptr && d1->out && d1->addend && d1->out_stride % 4 == 0 && aligned16(d1->out) &&
                 (d1->addend_map != nullptr || (d1->addend_stride % 4 == 0 && aligned16(d1->addend))) && (!d1->bias || aligned16(d1->bias)),
                 "conv_sub1_bwd: d1 needs 16-byte aligned out / addend with strides that are multiples of 4, no mask, no out_map");
    if (cbwd)
      SININN_CHECK(d1->Co % 4 == 0 && d1->v_stride % 4 == 0 && d1->out2_stride % 4 == 0 && d1->v && d1->sbuf && d1->out2 && aligned16(d1->v) &&
                   aligned16(d1->sbuf) && aligned16(d1->out2), "conv_sub1_bwd: ADD_CBWD needs 16-byte aligned v / sbuf / out2");
  }
  Sub1Dev q;
  alignas(16) static float dummy[4] = {0.f, 0.f, 0.f, 0.f};   // conv_prepare insists on pointers the kernel never follows
  sininn_conv_args ra = *rc;
  ra.mode = SININN_CONV_RELU; ra.out = dummy; ra.out_stride = S1_HID; ra.N = S1_HID;
  if (int e = conv_prepare(&ra, q.r)) return e;
  sininn_conv_args da = *d2;
  da.mode = SININN_CONV_LINEAR; da.mask = nullptr; da.out = dummy; da.out_stride = S1_HID; da.N = S1_HID;
  if (int e = conv_prepare(&da, q.a)) return e;
  sininn_conv_args db = *d1;
  db.in = dummy; db.in_stride = S1_HID;
  if (no_dx) { db.mode = SININN_CONV_LINEAR; db.out = dummy; db.out_stride = K1; db.addend = nullptr; db.addend_map = nullptr; }
  if (int e = conv_prepare(&db, q.b)) return e;
  q.r.tiles_x = q.a.tiles_x = q.b.tiles_x = (rc->W + 15) / 16;
  q.r.tiles_y = q.a.tiles_y = q.b.tiles_y = (rc->H + 3) / 4;
  q.ntiles = q.r.tiles_x * q.r.tiles_y * rc->B;
  q.no_dx = no_dx ? 1 : 0;
  q.slab = static_cast<float*>(ws);
  q.r.stamp = q.a.stamp = nullptr;
  q.b.stamp = d1->stamp;                          // diagnostic phase stamps (10 words)
  if (K1 == 8) return sub1_launch<8, 16>(q, slabs_out, st);
  if (K1 == 16) return sub1_launch<16, 32>(q, slabs_out, st);
  return sub1_launch<24, 48>(q, slabs_out, st);
}

int conv_sub1_bwd_reduce(int cond_cin, int co, const void* ws, int slabs, float* gw2, float* gb2, float* gw1, float* gb1, hipStream_t st) {
  SININN_CHECK(shape_ok(cond_cin, 2 * co) && ws && slabs > 0 && slabs <= S1_MAX_BLOCKS, "conv_sub1_bwd_reduce: bad arguments");
  const float* s = static_cast<const float*>(ws);
  if (cond_cin == 8) return sub1_reduce<8, 16>(s, slabs, gw2, gb2, gw1, gb1, st);
  if (cond_cin == 16) return sub1_reduce<16, 32>(s, slabs, gw2, gb2, gw1, gb1, st);
  return sub1_reduce<24, 48>(s, slabs, gw2, gb2, gw1, gb1, st);
}

// ---- forward --------------------------------------------------------------------------------------------------------------------
template <int K1, int N2, int HT>
static int sub1_fwd_launch(Sub1Dev& q, hipStream_t st) {
  constexpr int K1R = (K1 + 15) / 16 * 16;
  constexpr size_t lds = (size_t)(S1_P * S1_HS + 2 * S1_P * (K1R + 4) + N2 * S1_HS + 2 * S1_P * (N2 + 4)) * sizeof(float);
  static_assert(lds + 64 <= 160 * 1024, "conv_sub1_fwd: LDS");
  auto k = conv_sub1_fwd_kernel<K1, N2, HT>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("conv_sub1_fwd: cannot raise the LDS limit to %zu", lds); return 1; }
  }
  const int blocks = q.ntiles < S1_MAX_BLOCKS ? q.ntiles : S1_MAX_BLOCKS;
  hipLaunchKernelGGL(k, dim3(blocks), dim3(S1_NTHR), lds, st, q);
  SININN_LAUNCH_CHECK("conv_sub1_fwd");
  return 0;
}

// first: conv1 of the subnet (mode RELU; `out` is ignored: the hidden tensor is never stored), second: conv2 with a coupling
// epilogue (COUPLE_FWD / COUPLE_INV), described as for sininn_conv_pair_k1.
int conv_sub1_fwd_supported(const sininn_conv_args* f, const sininn_conv_args* s) {
  if (f && s && f->w_bf16) return conv_sub1_bf16_fwd_supported(f, s);
  if (!g_sub1_enabled || !f || !s) return 0;
  if (f->ksize != 1 || s->ksize != 1 || f->w_bf16 || s->w_bf16 || f->winograd || s->winograd || f->in_bf16 || s->in_bf16 || f->out_bf16) return 0;
  if (f->in_group_stride > 0 || f->out_group_stride > 0 || s->in_group_stride > 0) return 0;
  if (f->mode != SININN_CONV_RELU || !(s->mode == SININN_CONV_COUPLE_FWD || s->mode == SININN_CONV_COUPLE_INV)) return 0;
  if (f->Np != S1_HID || f->N != S1_HID || s->Cin != S1_HID) return 0;
  if (f->B != s->B || f->H != s->H || f->W != s->W) return 0;
  return shape_ok(f->Cin, s->Np) ? 1 : 0;
}

int conv_sub1_fwd_launch(const sininn_conv_args* f, const sininn_conv_args* s, hipStream_t st) {
  SININN_CHECK(conv_sub1_fwd_supported(f, s), "conv_sub1_fwd: unsupported subnet (check sininn_conv_sub1_fwd_supported first)");
  if (f->w_bf16) return conv_sub1_bf16_fwd_launch(f, s, st);
  SININN_CHECK((unsigned long long)f->H * f->W * f->in_stride * 4ull < (1ull << 31),
               "conv_sub1_fwd: one image of the input exceeds the 2 GB a block addresses (raw buffer staging)");
  Sub1Dev q;
  alignas(16) static float dummy[4] = {0.f, 0.f, 0.f, 0.f};
  sininn_conv_args fa = *f;
  fa.out = dummy; fa.out_stride = S1_HID;
  if (int e = conv_prepare(&fa, q.r)) return e;
  sininn_conv_args sa = *s;
  sa.in = dummy; sa.in_stride = S1_HID;
  if (int e = conv_prepare(&sa, q.b)) return e;
  q.a = q.b;
  q.r.tiles_x = q.b.tiles_x = (f->W + 15) / 16;
  q.r.tiles_y = q.b.tiles_y = (f->H + 3) / 4;
  q.ntiles = q.r.tiles_x * q.r.tiles_y * f->B;
  q.no_dx = 0; q.slab = nullptr;
  q.r.stamp = q.b.stamp = nullptr;
  const bool ht16 = s->col_tile == 32;
  SININN_CHECK(!ht16 || s->Co % 16 == 0, "conv_sub1_fwd: col_tile 32 needs Co %% 16 == 0");
  if (f->Cin == 8) return ht16 ? sub1_fwd_launch<8, 16, 16>(q, st) : sub1_fwd_launch<8, 16, 8>(q, st);
  if (f->Cin == 16) return ht16 ? sub1_fwd_launch<16, 32, 16>(q, st) : sub1_fwd_launch<16, 32, 8>(q, st);
  return ht16 ? sub1_fwd_launch<24, 48, 16>(q, st) : sub1_fwd_launch<24, 48, 8>(q, st);
}

}  // namespace sininn

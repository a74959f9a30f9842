// 3x3 conv with few input channels into the 256-channel hidden tensor, mixed precision: conv1 of a 3x3 subnet at level 0
// (subnet_conv, archs.py:11-13: 24 -> 256 + ReLU; fp32 input rounded to bf16 while staged, bf16 output).  The general bf16 conv
// (conv_bf16.hip) runs this layer as 4096 blocks of 256 pixels x 64 columns, each staging its input tile and 9 x 64 x 16 weights
// per channel chunk and then draining a 32 KB output tile: 124 us at BASELINE configs[3] for 15 us of matrix work and a 134 MB
// store (~25 us).  This kernel is the "fat output" shape taken at face value:
//   * a block computes ALL 256 output columns of its pixels (8 waves x 32 columns), so a pixel tile is staged once, not four times;
//   * K = 9 taps x Kp (16 / 32) is small enough for a wave's weight fragments to live in registers for the whole launch
//     (18 fragments = 72 VGPRs): the MFMA loop reads only the input image from LDS, one 16-byte read per MFMA;
//   * blocks are persistent over 16 x 16 pixel tiles; the next tile's halo (18 x 18 pixels) is requested and written into the other
//     half of a double buffer while this tile computes (two batches, three 32-pixel steps between request and use): one block
//     barrier per tile;
//   * the epilogue never touches LDS: a 32 x 32 accumulator tile has its column on the lane and pixels in the registers; bias + ReLU
//     + rounding to bf16 happen there, two lanes swap halves (DPP + v_perm_b32) and every lane stores 4 bytes = two adjacent
//     channels of one pixel; a wave's store instruction covers 64-byte runs, the eight waves of a block complete the 512-byte row.
// MASK = the data gradient of conv2 at the same level (dr: 2 Co <= 48 channels -> dh: 256 channels, masked by [h > 0], bf16 out;
// 127 us in the general kernel at configs[3]): the same kernel with 27 weight fragments per wave and the mask applied to the
// packed output word -- the mask word of (pixel, columns c, c + 1) is loaded from h with the address the output word is stored
// to, two 32-pixel steps ahead of its use.
// `bits` (block executor only): the ReLU gates as a bit mask, one 32-bit word per (32-pixel step of a tile, hidden column): bit p =
// gate of pixel p of the step.  A lane of conv1 holds 16 of a column's 32 pixels in its accumulator registers: it packs their
// gates, ORs with the lane 32 apart (the other 16 pixels) and the wave stores 128 contiguous bytes per step; the masked data
// gradient loads the words of its two columns once per step (8 bytes per lane) instead of eight words of h: 32 B per pixel
// instead of 512 B (134 MB -> 8 MB at BASELINE configs[3], level 0).  The layout is private to these two kernels.
#include <cstdlib>
#include "conv_bf16_types.h"

namespace sininn {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int C3K_NTHR = 512, C3K_HALO = 18, C3K_HPIX = C3K_HALO * C3K_HALO, C3K_MAX_BLOCKS = 256;

template <int CIN, bool MASK, bool BITS>
__global__ __launch_bounds__(C3K_NTHR) void conv3_smallk_bf16_kernel(ConvDevB q, int ntiles, unsigned* bits) {
  constexpr int KP = (CIN + 15) / 16 * 16;
  constexpr int NS = KP / 16, XSB = KP * 2 + 16;                  // bytes per halo pixel: 16 (mod 32) -> conflict-free 16-byte reads
  constexpr int IMG = (C3K_HPIX * XSB + 15) / 16 * 16;
  constexpr int QX = CIN / 4, FX = (C3K_HPIX * QX + C3K_NTHR - 1) / C3K_NTHR;
  const ConvDev& p = q.c;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_c3k[];   // 2 x IMG

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, hh = lane >> 5;
  const int cw = wave * 32;

  for (int f = tid; f < 2 * IMG / 16; f += C3K_NTHR) reinterpret_cast<u32x4*>(smem_c3k)[f] = (u32x4){0u, 0u, 0u, 0u};

  // the wave's weight fragments: B operand of k-step (tap t, half s): W[t][column cw + r][16 s + 8 hh .. + 7]
  bf16x8 wf[9][NS];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int s = 0; s < NS; ++s) wf[t][s] = *reinterpret_cast<const bf16x8*>(q.w + ((size_t)t * p.Np + cw + r) * KP + 16 * s + 8 * hh);
  const float bv = p.bias ? p.bias[cw + r] : 0.f;

  // staging slot u of a thread: halo pixel (tid + 512 u) / QX, channel quad (tid + 512 u) % QX -- the same for every tile
  const int tiles_img = p.tiles_x * p.tiles_y;
  const float* const in = static_cast<const float*>(q.in);
  constexpr int FXH = (FX + 1) / 2;                               // the slots are requested / written in two batches per tile (registers)
  auto issue_half = [&](int tile, int half, f32x4 (&vx)[FXH]) {
    const bool live = tile < ntiles;
    const int b = live ? tile / tiles_img : 0;
    const int trem = tile - b * tiles_img;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(in + (size_t)b * p.H * p.W * p.in_stride);
#pragma unroll
    for (int u = 0; u < FXH; ++u) {
      const int f = tid + C3K_NTHR * (half * FXH + u);
      const int pl = f / QX, c = (f - pl * QX) * 4;
      const int hy = pl / C3K_HALO, hx = pl - hy * C3K_HALO;
      const int gy = ty * 16 - 1 + hy, gx = tx * 16 - 1 + hx;
      const bool ok = live && pl < C3K_HPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      vx[u] = buf_load4(rs, ok ? (unsigned)(((gy * p.W + gx) * p.in_stride + c) * 4) : BUF_OOB, 0u);
    }
  };
  auto store_half = [&](int buf, int half, const f32x4 (&vx)[FXH]) {
    unsigned char* const xs = smem_c3k + buf * IMG;
#pragma unroll
    for (int u = 0; u < FXH; ++u) {
      const int f = tid + C3K_NTHR * (half * FXH + u);
      const int pl = f / QX, c = (f - pl * QX) * 4;
      if (pl < C3K_HPIX) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (__bf16)vx[u][j];
        *reinterpret_cast<bf16x4*>(xs + pl * XSB + c * 2) = o;
      }
    }
  };

  const bool odd = (r & 1) != 0;
  const unsigned sel = odd ? 0x03020706u : 0x05040100u;
  const int G = gridDim.x;
  // byte offset of this lane's output word d of fragment s of 32-pixel step m of a tile (relative to the image; `stride` in
  // elements): even lanes pixel pp, columns (c, c + 1); odd lanes pixel pp + 1, columns (c - 1, c).  BUF_OOB outside the image
  auto word_off = [&](int y0, int x0, int m, int s, int d, int stride) -> unsigned {
    const int pp = 16 * s + 8 * (d >> 1) + 2 * (d & 1) + 4 * hh + (odd ? 1 : 0);
    const int gy = y0 + 2 * m + (pp >> 4), gx = x0 + (pp & 15);
    return (gy < p.H && gx < p.W) ? (unsigned)(((gy * p.W + gx) * stride + cw + (r & ~1)) * 2) : BUF_OOB;
  };
  // ReLU-mask words of a 32-pixel step, requested two steps ahead of their use (a step is 27 MFMAs ~ 0.4 us per wave)
  unsigned mk[2][BITS ? 2 : 8];
  auto issue_mask = [&](int tile, int m, unsigned (&dst)[BITS ? 2 : 8]) {
    if constexpr (MASK) {
      const bool live = tile < ntiles;
      const int b = live ? tile / tiles_img : 0;
      const int trem = tile - b * tiles_img;
      const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
      if constexpr (BITS) {                          // gate words of this lane's two columns for step m of the tile
        const unsigned step = (unsigned)(((b * p.tiles_y + ty) * p.tiles_x + tx) * 8 + m);
        const u32x2 g2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(buf_rsrc(bits), (int)(live ? (step * 256u + cw + (r & ~1)) * 4u : BUF_OOB), 0, 0));
        dst[0] = g2[0]; dst[1] = g2[1];
      } else {
        const __amdgpu_buffer_rsrc_t rs = buf_rsrc(q.mask_b + (size_t)b * p.H * p.W * p.mask_stride);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int d = 0; d < 4; ++d)
            dst[4 * s + d] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(live ? word_off(ty * 16, tx * 16, m, s, d, p.mask_stride) : BUF_OOB), 0, 0);
      }
    }
  };

  f32x4 vx[FXH];
  issue_half(blockIdx.x, 0, vx);
  __syncthreads();                                   // zero fill complete
  store_half(0, 0, vx);
  issue_half(blockIdx.x, 1, vx);
  store_half(0, 1, vx);
  issue_mask(blockIdx.x, 0, mk[0]);
  issue_mask(blockIdx.x, 1, mk[1]);
  __syncthreads();
  int buf = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += G, buf ^= 1) {
    const int b = tile / tiles_img;
    const int trem = tile - b * tiles_img;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int y0 = ty * 16, x0 = tx * 16;
    const unsigned char* const xs = smem_c3k + buf * IMG;
    const __amdgpu_buffer_rsrc_t out_rs = buf_rsrc(q.out_b + (size_t)b * p.H * p.W * p.out_stride);
#pragma unroll
    for (int m = 0; m < 8; ++m) {                   // 32 pixels per step: tile rows 2 m, 2 m + 1
      const unsigned char* const arow = xs + ((2 * m + (r >> 4)) * C3K_HALO + (r & 15)) * XSB + 16 * hh;
      f32x16 acc;
      unsigned gate_word = 0u;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(arow + ((t / 3) * C3K_HALO + (t % 3)) * XSB + 32 * s);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wf[t][s], acc, 0, 0, 0);
        }
      // bias + ReLU (or the mask) + one rounding; element j of fragment s is pixel 16 s + 8 (j >> 2) + 4 hh + (j & 3) of the 32
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = MASK ? (__bf16)acc[8 * s + j] : (__bf16)fmaxf(acc[8 * s + j] + bv, 0.f);
        const u32x4 own = __builtin_bit_cast(u32x4, f);
        if constexpr (!MASK) {
          if constexpr (BITS) {                      // gates of this lane's 8 pixels of fragment s: pixel 16 s + 8 (j >> 2) + (j & 3) (+ 4 hh below)
#pragma unroll
            for (int dq = 0; dq < 4; ++dq) {
              const unsigned t = own[dq] & 0x7fff7fffu;          // after the ReLU a value is +0, -0 or positive
              const int p0 = 16 * s + 8 * (dq >> 1) + 2 * (dq & 1);
              gate_word |= ((t & 0xffffu) ? 1u : 0u) << p0;
              gate_word |= ((t >> 16) ? 1u : 0u) << (p0 + 1);
            }
          }
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)own[d], 0xB1, 0xF, 0xF, true);
          unsigned outw = __builtin_amdgcn_perm(nb, own[d], sel);
          if constexpr (MASK) {                      // h > 0 of the two bf16 halves: not zero and not negative
            const unsigned mw = BITS ? 0u : mk[m & 1][(BITS ? 0 : 4 * s + d)];
            unsigned keep;
            if constexpr (BITS) {                    // bit pp of the gate words of columns c, c + 1 (pp = this word's pixel of the step)
              const int pp = 16 * s + 8 * (d >> 1) + 2 * (d & 1) + 4 * hh + (odd ? 1 : 0);
              keep = (((mk[m & 1][0] >> pp) & 1u) ? 0x0000ffffu : 0u) | (((mk[m & 1][1] >> pp) & 1u) ? 0xffff0000u : 0u);
            } else {
              keep = (((mw & 0x7fffu) != 0u && (mw & 0x8000u) == 0u) ? 0x0000ffffu : 0u) |
                     (((mw & 0x7fff0000u) != 0u && (mw & 0x80000000u) == 0u) ? 0xffff0000u : 0u);
            }
            outw &= keep;
          }
          __builtin_amdgcn_raw_buffer_store_b32(outw, out_rs, (int)word_off(y0, x0, m, s, d, p.out_stride), 0, 0);
        }
      }
      if constexpr (!MASK && BITS) {                 // this lane's 16 pixels | the other 16 (lane 32 apart); lanes 0 .. 31 store 128 bytes
        unsigned g = gate_word << (4 * hh);
        g |= (unsigned)__shfl_xor((int)g, 32);
        const size_t step = ((size_t)(b * p.tiles_y + ty) * p.tiles_x + tx) * 8 + m;
        if (hh == 0) bits[step * 256 + cw + r] = g;
      }
      if (m < 6) issue_mask(tile, m + 2, mk[m & 1]);  // the set just consumed: two steps ahead (the next tile's first two at the end)
      else issue_mask(tile + G, m - 6, mk[m & 1]);
      // the next tile's halo image -> the other buffer, in two batches: requested at steps 0 / 4, written at steps 3 / 7 (a
      // tile beyond the last reads zeros into an image nobody uses)
      if (m == 0) issue_half(tile + G, 0, vx);
      if (m == 3) { store_half(buf ^ 1, 0, vx); issue_half(tile + G, 1, vx); }
      if (m == 7) store_half(buf ^ 1, 1, vx);
    }
    __syncthreads();                                 // every wave is done with this buffer; the next tile's image is complete
  }
}

static bool g_c3k_enabled = getenv("SININN_CONV3_SMALLK") == nullptr || atoi(getenv("SININN_CONV3_SMALLK")) != 0;   // A/B switch
void conv3_smallk_enable(int on) { g_c3k_enabled = on != 0; }

// conv1 of a 3x3 subnet on the mixed-precision path: fp32 input with at most 32 channels -> ReLU -> 256 bf16 channels
int conv3_smallk_bf16_supported(const sininn_conv_args* a) {
  if (!g_c3k_enabled || !a) return 0;
  if (a->ksize != 3 || !a->w_bf16 || a->in_bf16 || !a->out_bf16 || a->winograd) return 0;
  if (a->mode == SININN_CONV_RELU) { if (!a->bias) return 0; }
  else if (a->mode == SININN_CONV_MASK) { if (!a->mask || !a->mask_bf16 || a->mask_stride % 8 != 0 || a->mask_group_stride > 0) return 0; }
  else return 0;
  if (a->Np != 256 || a->N != 256) return 0;
  if (a->mode == SININN_CONV_RELU ? !(a->Cin == 8 || a->Cin == 16 || a->Cin == 24 || a->Cin == 32) : !(a->Cin == 16 || a->Cin == 32 || a->Cin == 48)) return 0;
  if (a->in_group_stride > 0 || a->out_group_stride > 0) return 0;
  if (a->out_stride % 8 != 0 || a->in_stride % 4 != 0) return 0;
  const unsigned long long px = (unsigned long long)a->H * a->W;
  if (px * a->in_stride * 4ull >= (1ull << 31) || px * a->out_stride * 2ull >= (1ull << 31)) return 0;   // raw buffer offsets per image
  if (a->mode == SININN_CONV_MASK && px * a->mask_stride * 2ull >= (1ull << 31)) return 0;
  return 1;
}

template <int CIN, bool MASK, bool BITS>
static int c3k_launch_b(const ConvDevB& q, int ntiles, unsigned* bits, hipStream_t st) {
  constexpr int KP = (CIN + 15) / 16 * 16;
  constexpr int IMG = (C3K_HPIX * (KP * 2 + 16) + 15) / 16 * 16;
  constexpr size_t lds = 2 * (size_t)IMG;
  auto k = conv3_smallk_bf16_kernel<CIN, MASK, BITS>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("conv3_smallk_bf16: cannot raise the LDS limit to %zu", lds); return 1; }
  }
  const int blocks = ntiles < C3K_MAX_BLOCKS ? ntiles : C3K_MAX_BLOCKS;
  hipLaunchKernelGGL(k, dim3(blocks), dim3(C3K_NTHR), lds, st, q, ntiles, bits);
  SININN_LAUNCH_CHECK("conv3_smallk_bf16");
  return 0;
}

template <int CIN, bool MASK>
static int c3k_launch(const ConvDevB& q, int ntiles, unsigned* bits, hipStream_t st) {
  return bits ? c3k_launch_b<CIN, MASK, true>(q, ntiles, bits, st) : c3k_launch_b<CIN, MASK, false>(q, ntiles, bits, st);
}

// bits: NULL, or the gate bit mask [B*H*W][8] words that conv1 (mode RELU) writes and the masked data gradient (mode MASK) reads
// instead of a->mask (block executor: the region lives in `saved`)
int conv3_smallk_bf16_launch_bits(const sininn_conv_args* a, unsigned* bits, hipStream_t st) {
  SININN_CHECK(conv3_smallk_bf16_supported(a), "conv3_smallk_bf16: unsupported conv");
  ConvDevB q;
  if (int rc = conv_bf16_prepare(a, q)) return rc;
  q.c.tiles_x = (a->W + 15) / 16; q.c.tiles_y = (a->H + 15) / 16;
  const int ntiles = q.c.tiles_x * q.c.tiles_y * a->B;
  SININN_CHECK(!bits || (unsigned long long)ntiles * 8ull * 256ull * 4ull < (1ull << 31), "conv3_smallk_bf16: gate bit mask exceeds 2 GB");
  if (a->mode == SININN_CONV_MASK) {
    switch (a->Cin) {
      case 16: return c3k_launch<16, true>(q, ntiles, bits, st);
      case 32: return c3k_launch<32, true>(q, ntiles, bits, st);
      default: return c3k_launch<48, true>(q, ntiles, bits, st);
    }
  }
  switch (a->Cin) {
    case 8: return c3k_launch<8, false>(q, ntiles, bits, st);
    case 16: return c3k_launch<16, false>(q, ntiles, bits, st);
    case 24: return c3k_launch<24, false>(q, ntiles, bits, st);
    default: return c3k_launch<32, false>(q, ntiles, bits, st);
  }
}

int conv3_smallk_bf16_launch(const sininn_conv_args* a, hipStream_t st) { return conv3_smallk_bf16_launch_bits(a, nullptr, st); }

}  // namespace sininn

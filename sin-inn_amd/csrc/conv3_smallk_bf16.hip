// 3x3 conv with few input channels into the 256-channel hidden tensor, mixed precision: conv1 of a 3x3 subnet at level 0
// (subnet_conv, archs.py:11-13: 24 -> 256 + ReLU; fp32 input rounded to bf16 while staged, bf16 output).  The general bf16 conv
// (conv_bf16.hip) runs this layer as 4096 blocks of 256 pixels x 64 columns, each staging its input tile and 9 x 64 x 16 weights
// per channel chunk and then draining a 32 KB output tile: 124 us at BASELINE configs[3] for 15 us of matrix work and a 134 MB
// store (~25 us).  This kernel is the "fat output" shape taken at face value:
//   * a block computes ALL 256 output columns of its pixels (8 waves x 32 columns), so a pixel tile is staged once, not four times;
//   * K = 9 taps x Kp (16 / 32) is small enough for a wave's weight fragments to live in registers for the whole launch
//     (18 fragments = 72 VGPRs): the MFMA loop reads only the input image from LDS, one 16-byte read per MFMA;
//   * blocks are persistent over 16 x 16 pixel tiles; the next tile's halo (18 x 18 pixels) is requested a tile ahead and written
//     into the other half of a double buffer mid-tile: one block barrier per tile;
//   * the epilogue never touches LDS: a 32 x 32 accumulator tile has its column on the lane and pixels in the registers; bias + ReLU
//     + rounding to bf16 happen there, two lanes swap halves (DPP + v_perm_b32) and every lane stores 4 bytes = two adjacent
//     channels of one pixel; a wave's store instruction covers 64-byte runs, the eight waves of a block complete the 512-byte row.
#include <cstdlib>
#include "conv_bf16_types.h"

namespace sininn {

constexpr int C3K_NTHR = 512, C3K_HALO = 18, C3K_HPIX = C3K_HALO * C3K_HALO, C3K_MAX_BLOCKS = 256;

template <int KP>
__global__ __launch_bounds__(C3K_NTHR) void conv3_smallk_bf16_kernel(ConvDevB q, int ntiles) {
  constexpr int NS = KP / 16, XSB = KP * 2 + 16;                  // bytes per halo pixel: 16 (mod 32) -> conflict-free 16-byte reads
  constexpr int IMG = (C3K_HPIX * XSB + 15) / 16 * 16;
  constexpr int QMAX = KP / 4, FX = (C3K_HPIX * QMAX + C3K_NTHR - 1) / C3K_NTHR;
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

  // staging slots of a thread: (halo pixel, channel quad), the same for every tile
  const int QX = p.Cin / 4;
  int s_pk[FX];                                                   // halo row << 16 | halo column << 8 | first channel; -1: no slot
#pragma unroll
  for (int u = 0; u < FX; ++u) {
    const int f = tid + C3K_NTHR * u;
    const int pl = f / QX, c = (f - pl * QX) * 4;
    const int hy = pl / C3K_HALO, hx = pl - hy * C3K_HALO;
    s_pk[u] = pl < C3K_HPIX ? (hy << 16) | (hx << 8) | c : -1;
  }
  const int tiles_img = p.tiles_x * p.tiles_y;
  const float* const in = static_cast<const float*>(q.in);
  auto issue_tile = [&](int tile, f32x4 (&vx)[FX]) {
    const bool live = tile < ntiles;
    const int b = live ? tile / tiles_img : 0;
    const int trem = tile - b * tiles_img;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(in + (size_t)b * p.H * p.W * p.in_stride);
#pragma unroll
    for (int u = 0; u < FX; ++u) {
      const int gy = ty * 16 - 1 + (s_pk[u] >> 16), gx = tx * 16 - 1 + ((s_pk[u] >> 8) & 255);
      const bool ok = live && s_pk[u] >= 0 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      vx[u] = buf_load4(rs, ok ? (unsigned)(((gy * p.W + gx) * p.in_stride + (s_pk[u] & 255)) * 4) : BUF_OOB, 0u);
    }
  };
  auto store_tile = [&](int buf, const f32x4 (&vx)[FX]) {
    unsigned char* const xs = smem_c3k + buf * IMG;
#pragma unroll
    for (int u = 0; u < FX; ++u) {
      if (s_pk[u] >= 0) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (__bf16)vx[u][j];
        *reinterpret_cast<bf16x4*>(xs + ((s_pk[u] >> 16) * C3K_HALO + ((s_pk[u] >> 8) & 255)) * XSB + (s_pk[u] & 255) * 2) = o;
      }
    }
  };

  const bool odd = (r & 1) != 0;
  const unsigned sel = odd ? 0x03020706u : 0x05040100u;
  const int G = gridDim.x;
  f32x4 vx[FX];
  issue_tile(blockIdx.x, vx);
  __syncthreads();                                   // zero fill complete
  store_tile(0, vx);
  issue_tile(blockIdx.x + G, vx);
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
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(arow + ((t / 3) * C3K_HALO + (t % 3)) * XSB + 32 * s);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wf[t][s], acc, 0, 0, 0);
        }
      // bias + ReLU + one rounding; element j of fragment s is pixel 16 s + 8 (j >> 2) + 4 hh + (j & 3) of the 32
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (__bf16)fmaxf(acc[8 * s + j] + bv, 0.f);
        const u32x4 own = __builtin_bit_cast(u32x4, f);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)own[d], 0xB1, 0xF, 0xF, true);
          const unsigned outw = __builtin_amdgcn_perm(nb, own[d], sel);
          const int pp = 16 * s + 8 * (d >> 1) + 2 * (d & 1) + 4 * hh + (odd ? 1 : 0);      // even lanes: pixel pp (columns c, c + 1)
          const int gy = y0 + 2 * m + (pp >> 4), gx = x0 + (pp & 15);
          const unsigned off = (gy < p.H && gx < p.W) ? (unsigned)(((gy * p.W + gx) * p.out_stride + cw + (r & ~1)) * 2) : BUF_OOB;
          __builtin_amdgcn_raw_buffer_store_b32(outw, out_rs, (int)off, 0, 0);
        }
      }
      if (m == 3) {                                  // mid-tile: the next tile (requested a tile ago) -> the other buffer; the one after it is requested
        if (tile + G < ntiles) store_tile(buf ^ 1, vx);
        issue_tile(tile + 2 * G, vx);
      }
    }
    __syncthreads();                                 // every wave is done with this buffer; the next tile's image is complete
  }
}

static bool g_c3k_enabled = getenv("SININN_CONV3_SMALLK") == nullptr || atoi(getenv("SININN_CONV3_SMALLK")) != 0;   // A/B switch
void conv3_smallk_enable(int on) { g_c3k_enabled = on != 0; }

// conv1 of a 3x3 subnet on the mixed-precision path: fp32 input with at most 32 channels -> ReLU -> 256 bf16 channels
int conv3_smallk_bf16_supported(const sininn_conv_args* a) {
  if (!g_c3k_enabled || !a) return 0;
  if (a->ksize != 3 || !a->w_bf16 || a->in_bf16 || !a->out_bf16 || a->winograd || a->mode != SININN_CONV_RELU) return 0;
  if (a->Np != 256 || a->N != 256 || a->Cin % 8 != 0 || a->Cin > 32 || !a->bias) return 0;
  if (a->in_group_stride > 0 || a->out_group_stride > 0) return 0;
  if (a->out_stride % 8 != 0 || a->in_stride % 4 != 0) return 0;
  const unsigned long long px = (unsigned long long)a->H * a->W;
  if (px * a->in_stride * 4ull >= (1ull << 31) || px * a->out_stride * 2ull >= (1ull << 31)) return 0;   // raw buffer offsets per image
  return 1;
}

template <int KP>
static int c3k_launch(const ConvDevB& q, int ntiles, hipStream_t st) {
  constexpr int IMG = (C3K_HPIX * (KP * 2 + 16) + 15) / 16 * 16;
  constexpr size_t lds = 2 * (size_t)IMG;
  auto k = conv3_smallk_bf16_kernel<KP>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("conv3_smallk_bf16: cannot raise the LDS limit to %zu", lds); return 1; }
  }
  const int blocks = ntiles < C3K_MAX_BLOCKS ? ntiles : C3K_MAX_BLOCKS;
  hipLaunchKernelGGL(k, dim3(blocks), dim3(C3K_NTHR), lds, st, q, ntiles);
  SININN_LAUNCH_CHECK("conv3_smallk_bf16");
  return 0;
}

int conv3_smallk_bf16_launch(const sininn_conv_args* a, hipStream_t st) {
  SININN_CHECK(conv3_smallk_bf16_supported(a), "conv3_smallk_bf16: unsupported conv");
  ConvDevB q;
  if (int rc = conv_bf16_prepare(a, q)) return rc;
  q.c.tiles_x = (a->W + 15) / 16; q.c.tiles_y = (a->H + 15) / 16;
  const int ntiles = q.c.tiles_x * q.c.tiles_y * a->B;
  return q.Kp == 16 ? c3k_launch<16>(q, ntiles, st) : c3k_launch<32>(q, ntiles, st);
}

}  // namespace sininn

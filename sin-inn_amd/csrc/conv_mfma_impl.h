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
#pragma once
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
  int col_tile;   // coupling (s|t) interleave width of the packed weights (16 or 32)
  unsigned long long* stamp;   // optional {start, end} wall-clock words (sininn_conv_args.stamp)
  size_t out_gs, mask_gs;   // > 0: `out` / `mask` are channel-group-major [N/8][pixel][8] with this many floats between groups
                            // (RELU / MASK modes of the Winograd kernels: the hidden tensors h / dh of a 3x3 GLOW block)
  int in_chunk;   // floats between consecutive 8-channel groups of one input pixel: 8 for pixel-major tensors; B*H*W*8 for
                  // the channel-group-major layout [C/8][pixel][8] (then in_stride == 8) -- Winograd kernels only
  int ablate;   // diagnostic only (tools/bench_kernels.py --ablate): 1 skip staging, 2 skip loop barrier, 4 skip LDS reads
};


// kernel-side execution window for the live roofline measurement (block-uniform branch, one atomic per block)
__device__ __forceinline__ void stamp_begin(const ConvDev& p) {
  if (p.stamp && threadIdx.x == 0) atomicMin(p.stamp, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}
__device__ __forceinline__ void stamp_end(const ConvDev& p) {
  if (p.stamp && threadIdx.x == 0) atomicMax(p.stamp + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

// ------------------------------------------------------------------------------------------------
// Epilogue, phase 2 (shared by the 16- and 32-wide kernels): the block's accumulator tile has been
// staged in LDS as T[pixel][BN+4] (raw GEMM sums, block-local columns); all 256 threads now walk it
// in 16-byte units along the channel axis, so every global access of the epilogue (bias, mask, v,
// addend, y, s) is a coalesced float4 instead of a per-lane 4-byte scatter.
//   HT = half-tile of the coupling interleave: tile of 2*HT columns = [ s HT ch | t HT ch ].
// ------------------------------------------------------------------------------------------------
// T2 (optional): a second tile of partial sums added to T on the fly (position-split Winograd kernel).
template <int TH, int BN, int HT, int NTHR = 256>
__device__ __forceinline__ void conv_epilogue_tile(const ConvDev& p, const float* T, int b, int y0, int x0, int n0,
                                                   int tid, float* red, const float* T2 = nullptr) {
  constexpr int TS = BN + 4;
  constexpr int NPIX = TH * 16;
  const int MODE = p.mode;
  if (p.ablate & 8) return;
  if (MODE == SININN_CONV_COUPLE_FWD || MODE == SININN_CONV_COUPLE_INV) {
    constexpr int CB = BN / 2;                       // channels of this block
    constexpr int Q = CB / 4;                        // channel quads per pixel
    const int c_block0 = n0 / 2;
    float ld_acc = 0.f;
    if constexpr (NTHR % Q == 0) {
      // a thread's channel quad is the same in every iteration: channel offsets and the packed bias are loop invariants,
      // and the only per-pixel global input (v) is requested for all iterations up front, so the loop pays one
      // global-load latency, not one per iteration (it was a third of the bf16 kernel's block time)
      const int q4 = tid % Q;
      const int cl = q4 * 4;                         // block-local channel
      const int c = c_block0 + cl;
      const bool cok = c < p.Co;
      const int tcol = (cl / HT) * (2 * HT) + (cl % HT);
      f32x4 bs = {0.f, 0.f, 0.f, 0.f}, bt = bs;
      if (p.bias && cok) {
        bs = *reinterpret_cast<const f32x4*>(p.bias + n0 + tcol);
        bt = *reinterpret_cast<const f32x4*>(p.bias + n0 + tcol + HT);
      }
      constexpr int ITERS = (NPIX * Q + NTHR - 1) / NTHR;
      auto load_v = [&](int it) -> f32x4 {
        const int pl = (tid + it * NTHR) / Q;
        const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (!(cok && pl < NPIX && gy < p.H && gx < p.W)) return z;
        return *reinterpret_cast<const f32x4*>(p.v + ((size_t)(b * p.H + gy) * p.W + gx) * p.v_stride + c);
      };
      // every iteration's v is requested before the first one is used (the accumulators are dead, registers are free): the
      // loop pays ONE global-load latency instead of one per iteration
      f32x4 v_all[ITERS];
#pragma unroll
      for (int it = 0; it < ITERS; ++it) v_all[it] = load_v(it);
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const f32x4 v4 = v_all[it];
        const int pl = (tid + it * NTHR) / Q;
        const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
        if (cok && pl < NPIX && gy < p.H && gx < p.W) {
          f32x4 s4 = *reinterpret_cast<const f32x4*>(T + pl * TS + tcol) + bs;
          f32x4 t4 = *reinterpret_cast<const f32x4*>(T + pl * TS + tcol + HT) + bt;
          if (T2) {
            s4 += *reinterpret_cast<const f32x4*>(T2 + pl * TS + tcol);
            t4 += *reinterpret_cast<const f32x4*>(T2 + pl * TS + tcol + HT);
          }
          const size_t pix = (size_t)(b * p.H + gy) * p.W + gx;
          f32x4 y4;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float L = glow_log_e(s4[j], p.clamp);
            const float e = expf(L);
            if (MODE == SININN_CONV_COUPLE_FWD) { y4[j] = e * v4[j] + t4[j]; ld_acc += L; }
            else { y4[j] = (v4[j] - t4[j]) / e; ld_acc -= L; }
          }
          if (p.out_map) {
#pragma unroll
            for (int j = 0; j < 4; ++j) p.out[pix * p.out_stride + p.out_map[c + j]] = y4[j];
          } else {
            *reinterpret_cast<f32x4*>(p.out + pix * p.out_stride + c) = y4;
          }
          if (p.out2) *reinterpret_cast<f32x4*>(p.out2 + pix * p.out2_stride + c) = y4;
          if (p.sbuf) *reinterpret_cast<f32x4*>(p.sbuf + pix * p.Co + c) = s4;
        }
      }
    } else
    for (int idx = tid; idx < NPIX * Q; idx += NTHR) {
      const int pl = idx / Q, q4 = idx - pl * Q;
      const int cl = q4 * 4;                         // block-local channel
      const int c = c_block0 + cl;
      const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
      if (c < p.Co && gy < p.H && gx < p.W) {
        const int tcol = (cl / HT) * (2 * HT) + (cl % HT);
        f32x4 s4 = *reinterpret_cast<const f32x4*>(T + pl * TS + tcol);
        f32x4 t4 = *reinterpret_cast<const f32x4*>(T + pl * TS + tcol + HT);
        if (T2) {
          s4 += *reinterpret_cast<const f32x4*>(T2 + pl * TS + tcol);
          t4 += *reinterpret_cast<const f32x4*>(T2 + pl * TS + tcol + HT);
        }
        if (p.bias) {
          s4 += *reinterpret_cast<const f32x4*>(p.bias + n0 + tcol);
          t4 += *reinterpret_cast<const f32x4*>(p.bias + n0 + tcol + HT);
        }
        const size_t pix = (size_t)(b * p.H + gy) * p.W + gx;
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(p.v + pix * p.v_stride + c);
        f32x4 y4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float L = glow_log_e(s4[j], p.clamp);
          const float e = expf(L);
          if (MODE == SININN_CONV_COUPLE_FWD) { y4[j] = e * v4[j] + t4[j]; ld_acc += L; }
          else { y4[j] = (v4[j] - t4[j]) / e; ld_acc -= L; }
        }
        if (p.out_map) {
#pragma unroll
          for (int j = 0; j < 4; ++j) p.out[pix * p.out_stride + p.out_map[c + j]] = y4[j];
        } else {
          *reinterpret_cast<f32x4*>(p.out + pix * p.out_stride + c) = y4;
        }
        if (p.out2) *reinterpret_cast<f32x4*>(p.out2 + pix * p.out2_stride + c) = y4;
        if (p.sbuf) *reinterpret_cast<f32x4*>(p.sbuf + pix * p.Co + c) = s4;
      }
    }
    if (p.logdet) {                                  // block-uniform branch
      const float w = wave_sum(ld_acc);
      if ((tid & 63) == 0) red[tid >> 6] = w;
      __syncthreads();
      if (tid == 0) {
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < NTHR / 64; ++i) tot += red[i];
        atomicAdd(p.logdet + b, tot);
      }
    }
  } else {
    constexpr int Q = BN / 4;
    if constexpr (NTHR % Q == 0) {
      constexpr int ITERS = (NPIX * Q + NTHR - 1) / NTHR;
      const bool act = MODE == SININN_CONV_RELU || MODE == SININN_CONV_LINEAR || MODE == SININN_CONV_LRELU;
      const bool plain_add = MODE == SININN_CONV_ADD && p.addend_map == nullptr;
      const int colq = n0 + (tid % Q) * 4;           // this thread's column quad in every iteration
      // LINEAR / ADD with a mask: LeakyReLU-backward tail on the columns >= Co (IRN DenseBlock data gradients, see sininn.h)
      const bool lrelu_tail = (MODE == SININN_CONV_LINEAR || MODE == SININN_CONV_ADD) && p.mask != nullptr && colq >= p.Co;
      if ((act || MODE == SININN_CONV_MASK || plain_add) && colq + 3 < p.N) {
        // hot modes, full quads: the bias is a loop invariant; the per-pixel side input (mask / addend) is requested one
        // iteration ahead.  Partial quads and the other modes take the general loop below.
        f32x4 bq = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && (act || plain_add)) bq = *reinterpret_cast<const f32x4*>(p.bias + colq);
        auto side = [&](int it) -> f32x4 {
          f32x4 z = {0.f, 0.f, 0.f, 0.f};
          if (act) return z;
          const int pl = (tid + it * NTHR) / Q;
          const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
          if (!(pl < NPIX && gy < p.H && gx < p.W)) return z;
          const size_t pix = (size_t)(b * p.H + gy) * p.W + gx;
          if (MODE == SININN_CONV_MASK)
            return p.mask_gs ? *reinterpret_cast<const f32x4*>(p.mask + (size_t)(colq >> 3) * p.mask_gs + pix * 8 + (colq & 7))
                             : *reinterpret_cast<const f32x4*>(p.mask + pix * p.mask_stride + colq);
          return *reinterpret_cast<const f32x4*>(p.addend + pix * p.addend_stride + colq);
        };
        // all side inputs of the thread are requested up front (see the coupling path above): with one request in flight the
        // eight iterations of a 32-column block each waited out a global-load round trip -- 24 us of the 105 us of the level-0
        // data gradient of conv2 (profiles/r03_wino_fwd_ablation.log)
        f32x4 side_all[ITERS], fk_all[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) side_all[it] = side(it);
        if (lrelu_tail) {                                // the LeakyReLU gates of the tail columns, likewise
#pragma unroll
          for (int it = 0; it < ITERS; ++it) {
            const int pl = (tid + it * NTHR) / Q;
            const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
            fk_all[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (pl < NPIX && gy < p.H && gx < p.W)
              fk_all[it] = *reinterpret_cast<const f32x4*>(p.mask + ((size_t)(b * p.H + gy) * p.W + gx) * p.mask_stride + colq);
          }
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          const f32x4 sd = side_all[it];
          const int pl = (tid + it * NTHR) / Q;
          const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
          if (pl < NPIX && gy < p.H && gx < p.W) {
            const size_t pix = (size_t)(b * p.H + gy) * p.W + gx;
            f32x4 val = *reinterpret_cast<const f32x4*>(T + pl * TS + (tid % Q) * 4);
            if (T2) val += *reinterpret_cast<const f32x4*>(T2 + pl * TS + (tid % Q) * 4);
            if (act) {
              val += bq;
              if (MODE == SININN_CONV_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) val[j] = fmaxf(val[j], 0.f);
              } else if (MODE == SININN_CONV_LRELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) val[j] = val[j] > 0.f ? val[j] : val[j] * p.clamp;
              }
            } else if (MODE == SININN_CONV_MASK) {
#pragma unroll
              for (int j = 0; j < 4; ++j) val[j] = (sd[j] > 0.f) ? val[j] : 0.f;
            } else {
              val += bq;
              val += sd;
            }
            if (lrelu_tail) {
              const f32x4 fk = fk_all[it];
#pragma unroll
              for (int j = 0; j < 4; ++j) val[j] = fk[j] > 0.f ? val[j] : val[j] * p.clamp;
            }
            if (p.out_gs) *reinterpret_cast<f32x4*>(p.out + (size_t)(colq >> 3) * p.out_gs + pix * 8 + (colq & 7)) = val;
            else *reinterpret_cast<f32x4*>(p.out + pix * p.out_stride + colq) = val;
          }
        }
        return;
      }
      // data gradient of conv1 with the first half's coupling backward fused in (ADD_CBWD_*): full quads with 16-byte aligned
      // side tensors take the same shape of loop -- every global input of the thread (skip gradient, u, s) is requested up front,
      // (ds | dt) and dv leave as float4.  (The general loop below did these as 4-byte accesses with one round trip per
      // iteration: ~15 us of the ~72 us launches of this class.)
      const bool cbwd = MODE == SININN_CONV_ADD_CBWD_FWD || MODE == SININN_CONV_ADD_CBWD_INV;
      if (cbwd && colq + 3 < p.N && p.Co % 4 == 0 && p.v_stride % 4 == 0 && p.out_stride % 4 == 0 && p.out2_stride % 4 == 0 &&
          ((reinterpret_cast<uintptr_t>(p.v) | reinterpret_cast<uintptr_t>(p.sbuf) | reinterpret_cast<uintptr_t>(p.out) |
            reinterpret_cast<uintptr_t>(p.out2)) & 15) == 0 &&
          (p.addend_map != nullptr || (p.addend_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(p.addend) & 15) == 0))) {
        f32x4 bq = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bq = *reinterpret_cast<const f32x4*>(p.bias + colq);
        const float gl = p.logdet ? p.logdet[b] : 0.f;
        int amap[4] = {colq, colq + 1, colq + 2, colq + 3};
        if (p.addend_map) {
#pragma unroll
          for (int j = 0; j < 4; ++j) amap[j] = p.addend_map[colq + j];
        }
        f32x4 ad_all[ITERS], u_all[ITERS], s_all[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          const int pl = (tid + it * NTHR) / Q;
          const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
          f32x4 z = {0.f, 0.f, 0.f, 0.f};
          ad_all[it] = z; u_all[it] = z; s_all[it] = z;
          if (pl < NPIX && gy < p.H && gx < p.W) {
            const size_t pix = (size_t)(b * p.H + gy) * p.W + gx;
            if (p.addend_map) {
#pragma unroll
              for (int j = 0; j < 4; ++j) ad_all[it][j] = p.addend[pix * p.addend_stride + amap[j]];
            } else {
              ad_all[it] = *reinterpret_cast<const f32x4*>(p.addend + pix * p.addend_stride + colq);
            }
            u_all[it] = *reinterpret_cast<const f32x4*>(p.v + pix * p.v_stride + colq);
            s_all[it] = *reinterpret_cast<const f32x4*>(p.sbuf + pix * p.Co + colq);
          }
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          const int pl = (tid + it * NTHR) / Q;
          const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
          if (pl < NPIX && gy < p.H && gx < p.W) {
            const size_t pix = (size_t)(b * p.H + gy) * p.W + gx;
            f32x4 val = *reinterpret_cast<const f32x4*>(T + pl * TS + (tid % Q) * 4);
            if (T2) val += *reinterpret_cast<const f32x4*>(T2 + pl * TS + (tid % Q) * 4);
            val += bq;
            val += ad_all[it];
            f32x4 ds4, dt4, dv4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float g = val[j], u = u_all[it][j], sv = s_all[it][j];
              const float L = glow_log_e(sv, p.clamp), dL = glow_dlog_e(sv, p.clamp);
              const float e = expf(L);
              if (MODE == SININN_CONV_ADD_CBWD_FWD) { dv4[j] = g * e; dt4[j] = g; ds4[j] = (g * u * e + gl) * dL; }
              else { dv4[j] = g / e; dt4[j] = -dv4[j]; ds4[j] = -(g * u + gl) * dL; }
            }
            *reinterpret_cast<f32x4*>(p.out + pix * p.out_stride + colq) = ds4;
            *reinterpret_cast<f32x4*>(p.out + pix * p.out_stride + p.Co + colq) = dt4;
            *reinterpret_cast<f32x4*>(p.out2 + pix * p.out2_stride + colq) = dv4;
          }
        }
        return;
      }
    }
    for (int idx = tid; idx < NPIX * Q; idx += NTHR) {
      const int pl = idx / Q, q4 = idx - pl * Q;
      const int col = n0 + q4 * 4;
      const int gy = y0 + (pl >> 4), gx = x0 + (pl & 15);
      if (col < p.N && gy < p.H && gx < p.W) {
        const size_t pix = (size_t)(b * p.H + gy) * p.W + gx;
        f32x4 val = *reinterpret_cast<const f32x4*>(T + pl * TS + q4 * 4);
        if (T2) val += *reinterpret_cast<const f32x4*>(T2 + pl * TS + q4 * 4);
        const bool full = (col + 3 < p.N);
        if (MODE == SININN_CONV_RELU || MODE == SININN_CONV_LINEAR || MODE == SININN_CONV_LRELU) {
          if (p.bias) val += *reinterpret_cast<const f32x4*>(p.bias + col);   // packed bias has Np >= col+4 entries
          if (MODE == SININN_CONV_RELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) val[j] = fmaxf(val[j], 0.f);
          } else if (MODE == SININN_CONV_LRELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) val[j] = val[j] > 0.f ? val[j] : val[j] * p.clamp;
          }
        } else if (MODE == SININN_CONV_IRN_FWD || MODE == SININN_CONV_IRN_INV) {
          if (p.bias) val += *reinterpret_cast<const f32x4*>(p.bias + col);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (col + j < p.N) {
              const float hv = p.mask[pix * p.mask_stride + col + j];
              const float sv = p.clamp * (2.f / (1.f + expf(-hv)) - 1.f);
              const float vv = p.v[pix * p.v_stride + col + j];
              val[j] = (MODE == SININN_CONV_IRN_FWD) ? vv * expf(sv) + val[j] : (vv - val[j]) / expf(sv);
            }
        } else if (MODE == SININN_CONV_MASK) {
          if (full) {
            const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + pix * p.mask_stride + col);
#pragma unroll
            for (int j = 0; j < 4; ++j) val[j] = (mk[j] > 0.f) ? val[j] : 0.f;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (col + j < p.N) val[j] = (p.mask[pix * p.mask_stride + col + j] > 0.f) ? val[j] : 0.f;
          }
        } else if (MODE == SININN_CONV_ADD || MODE == SININN_CONV_ADD_CBWD_FWD || MODE == SININN_CONV_ADD_CBWD_INV) {
          if (p.bias) val += *reinterpret_cast<const f32x4*>(p.bias + col);
          if (p.addend_map == nullptr && full) {
            val += *reinterpret_cast<const f32x4*>(p.addend + pix * p.addend_stride + col);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (col + j < p.N) {
                const int ac = p.addend_map ? p.addend_map[col + j] : (col + j);
                val[j] += p.addend[pix * p.addend_stride + ac];
              }
          }
        }
        if ((MODE == SININN_CONV_LINEAR || MODE == SININN_CONV_ADD) && p.mask != nullptr) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (col + j < p.N && col + j >= p.Co && !(p.mask[pix * p.mask_stride + col + j] > 0.f)) val[j] *= p.clamp;
        }
        if (MODE == SININN_CONV_ADD_CBWD_FWD || MODE == SININN_CONV_ADD_CBWD_INV) {
          // val = gradient w.r.t. the first half's output y; emit (ds | dt) and dv of that half's coupling tail
          const float gl = p.logdet ? p.logdet[b] : 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (col + j < p.N) {
              const int c = col + j;
              const float g = val[j];
              const float u = p.v[pix * p.v_stride + c];
              const float sv = p.sbuf[pix * p.Co + c];
              const float L = glow_log_e(sv, p.clamp), dL = glow_dlog_e(sv, p.clamp);
              const float e = expf(L);
              float ds, dt, dvv;
              if (MODE == SININN_CONV_ADD_CBWD_FWD) { dvv = g * e; dt = g; ds = (g * u * e + gl) * dL; }
              else { dvv = g / e; dt = -dvv; ds = -(g * u + gl) * dL; }
              p.out[pix * p.out_stride + c] = ds;
              p.out[pix * p.out_stride + p.Co + c] = dt;
              p.out2[pix * p.out2_stride + c] = dvv;
            }
        } else if (full) {
          *reinterpret_cast<f32x4*>(p.out + pix * p.out_stride + col) = val;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (col + j < p.N) p.out[pix * p.out_stride + col + j] = val[j];
        }
      }
    }
  }
}

template <int KS, int TH, int WM, int WN, int MT, int NT, int CK>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvDev p) {
  constexpr int HALO = KS / 2;
  constexpr int IW = 16 + 2 * HALO;
  constexpr int IH = TH + 2 * HALO;
  constexpr int NPIX_IN = IH * IW;
  constexpr int TAPS = KS * KS;
  constexpr int BN = WN * NT * 16;
  constexpr int S = CK + 4;                        // LDS row stride (floats): 4*odd -> conflict-free b64 reads
  constexpr int C4N = CK / 4;
  constexpr int KSTEPS = CK / 8;
  constexpr int IN_F4 = (NPIX_IN * C4N + 255) / 256;
  constexpr int W_F4 = (BN * C4N + 255) / 256;
  static_assert(WM * MT == TH && WM * WN == 4, "bad wave layout");
  static_assert(CK % 8 == 0 && CK <= 32, "bad channel chunk");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const in_lds = smem;                      // single buffer (changes once per TAPS iterations)
  float* const w_lds0 = smem + NPIX_IN * S;        // weights: double buffer
  float* const w_lds1 = w_lds0 + BN * S;

  const int tid = threadIdx.x;
  stamp_begin(p);
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
    const int pix = f / C4N, c4 = f - pix * C4N;
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
    const int row = f / C4N, c4 = f - row * C4N;
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
  auto store_in = [&]() {
#pragma unroll
    for (int r = 0; r < IN_F4; ++r)
      if (in_loff[r] >= 0) *reinterpret_cast<f32x4*>(in_lds + in_loff[r]) = in_reg[r];
  };
  auto load_w = [&](int it) {
    const int chunk = it / TAPS, tap = it - chunk * TAPS;
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

  // ---- prologue: tile 0 into LDS, tile 1's weights in flight in registers -------------------------
  load_in(0);
  load_w(0);
  store_in();
  store_w(w_lds0);
  if (nit > 1) load_w(1);
  __syncthreads();

  int chunk = 0, tap = 0;
  for (int it = 0; it < nit; ++it) {
    // (1) weights of iteration it+1 (fetched during it-1): registers -> the other LDS buffer; its last
    //     readers finished before the barrier that ended iteration it-1
    if (it + 1 < nit) store_w(((it + 1) & 1) ? w_lds1 : w_lds0);
    // (2) start fetching the weights of it+2 and, on the last tap of a chunk, the next chunk's halo tile
    if (it + 2 < nit) load_w(it + 2);
    const bool last_tap = (tap == TAPS - 1);
    const bool next_in = last_tap && (chunk + 1 < nchunks);
    if (next_in) load_in(chunk + 1);

    // (3) MFMAs of this (chunk, tap): fragments double-buffered in registers across k-steps
    const float* Bw = (it & 1) ? w_lds1 : w_lds0;
    const int dy = tap / KS, dx = tap - dy * KS;
    const float* A = in_lds + (dy * IW + dx) * S;
    float2 af[2][MT], bf[2][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) af[0][m] = *reinterpret_cast<const float2*>(A + a_base[m]);
#pragma unroll
    for (int n = 0; n < NT; ++n) bf[0][n] = *reinterpret_cast<const float2*>(Bw + b_base[n]);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < KSTEPS) {
#pragma unroll
        for (int m = 0; m < MT; ++m) af[nxt][m] = *reinterpret_cast<const float2*>(A + a_base[m] + (ks + 1) * 8);
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[nxt][n] = *reinterpret_cast<const float2*>(Bw + b_base[n] + (ks + 1) * 8);
      }
      // keep the next k-step's LDS reads ahead of this k-step's MFMAs (hipcc otherwise sinks them next to their
      // first use and the wave eats the LDS latency on every k-step)
      __builtin_amdgcn_sched_barrier(0);
      // two passes so that consecutive MFMAs never hit the same accumulator (16x16x4 f32: 32-cycle issue,
      // 40-cycle dependent latency)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][m].x, bf[cur][n].x, acc[m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][m].y, bf[cur][n].y, acc[m][n], 0, 0, 0);
    }
    __syncthreads();
    if (next_in) {            // every wave is done with the old halo tile: replace it
      store_in();
      __syncthreads();
    }
    if (last_tap) { tap = 0; ++chunk; } else { ++tap; }
  }

  // ---- epilogue phase 1: accumulators -> LDS tile T[pixel][BN+4] (lane holds D[row = 4*kq + r][col = li]) ----
  // (the last main-loop barrier guarantees nobody still reads the operand tiles that T overlays)
  {
    constexpr int TS = BN + 4;
    float* const T = smem;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          T[((wm * MT + m) * 16 + 4 * kq + r) * TS + (wn * NT + n) * 16 + li] = acc[m][n][r];
    __syncthreads();
    __shared__ float red[4];
    conv_epilogue_tile<TH, BN, 8>(p, T, b, y0, x0, n0, tid, red);
    stamp_end(p);
  }
}

template <int KS, int TH, int WM, int WN, int MT, int NT, int CK>
static int launch_cfg_ck(const ConvDev& d, hipStream_t st) {
  constexpr int HALO = KS / 2;
  constexpr int NPIX_IN = (TH + 2 * HALO) * (16 + 2 * HALO);
  constexpr int BN = WN * NT * 16;
  constexpr int S = CK + 4;
  constexpr size_t lds_main = (size_t)(NPIX_IN + 2 * BN) * S * sizeof(float);
  constexpr size_t lds_epi = (size_t)TH * 16 * (BN + 4) * sizeof(float);
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  static_assert(lds <= 160 * 1024, "LDS tile too large");
  dim3 grid(d.tiles_x * d.tiles_y * d.B, (d.Np + BN - 1) / BN);
  auto k = conv_mfma_kernel<KS, TH, WM, WN, MT, NT, CK>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("conv: cannot raise LDS limit to %zu", lds); return 1; }
  }
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, d);
  SININN_LAUNCH_CHECK("conv_mfma");
  return 0;
}

template <int KS, int TH, int WM, int WN, int MT, int NT>
static int launch_cfg(const ConvDev& d, hipStream_t st) {
  switch (d.CK) {
    case 32: return launch_cfg_ck<KS, TH, WM, WN, MT, NT, 32>(d, st);
    case 24: return launch_cfg_ck<KS, TH, WM, WN, MT, NT, 24>(d, st);
    case 16: return launch_cfg_ck<KS, TH, WM, WN, MT, NT, 16>(d, st);
    case 8: return launch_cfg_ck<KS, TH, WM, WN, MT, NT, 8>(d, st);
    default: set_error("conv: unsupported channel chunk %d", d.CK); return 1;
  }
}

// Tile-shape choice.  nt16 = packed column tiles; pick the widest block that divides the columns well, and the
// 4-row spatial tile when the 8-row one would leave the 256 CUs under-filled.
template <int KS>
static int dispatch(ConvDev& d, hipStream_t st, int force_cfg) {
  const int nt16 = d.Np / 16;
  auto set_tiles = [&](int th) { d.tiles_x = (d.W + 15) / 16; d.tiles_y = (d.H + th - 1) / th; };
  const long pix_tiles8 = (long)d.B * ((d.H + 7) / 8) * ((d.W + 15) / 16);
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
      case 8: return launch_cfg<KS, 8, 2, 2, 4, 4>(d, st);
      case 6: return launch_cfg<KS, 8, 2, 2, 4, 3>(d, st);
      case 4: return launch_cfg<KS, 8, 2, 2, 4, 2>(d, st);
      case 3: return launch_cfg<KS, 8, 4, 1, 2, 3>(d, st);
      default: return launch_cfg<KS, 8, 4, 1, 2, 2>(d, st);
    }
  } else {
    set_tiles(4);
    switch (bnt) {
      case 8: return launch_cfg<KS, 4, 1, 4, 4, 2>(d, st);
      case 6: return launch_cfg<KS, 4, 2, 2, 2, 3>(d, st);
      case 4: return launch_cfg<KS, 4, 2, 2, 2, 2>(d, st);
      case 3: return launch_cfg<KS, 4, 4, 1, 1, 3>(d, st);
      default: return launch_cfg<KS, 4, 4, 1, 1, 2>(d, st);
    }
  }
}

}  // namespace sininn

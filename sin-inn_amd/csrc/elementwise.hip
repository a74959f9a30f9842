// HBM-bound kernels of the sin-inn training path: weight packing, coupling backward tail, index maps
// (squeeze / permute / layout import-export), losses, warps, frame-window sampler, Adam.
#include "common.h"

namespace sininn {

struct Str4 { int64_t b, c, h, w; };
static inline Str4 mk(const int64_t s[4]) { return Str4{s[0], s[1], s[2], s[3]}; }

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ w, const float* __restrict__ bias, int N, int Cin,
                                    int taps, const int* __restrict__ colmap, int Np, float* __restrict__ w_fwd,
                                    float* __restrict__ b_fwd, int Cdp, float* __restrict__ w_dgrad) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int nf = w_fwd ? taps * Np * Cin : 0;
  const int nd = w_dgrad ? taps * Cdp * N : 0;
  if (idx < nf) {
    const int c = idx % Cin, q = (idx / Cin) % Np, t = idx / (Cin * Np);
    const int n = colmap ? colmap[q] : q;
    w_fwd[idx] = (n >= 0 && n < N) ? w[((size_t)n * Cin + c) * taps + t] : 0.f;
  } else if (idx < nf + nd) {
    const int k = idx - nf;
    const int n = k % N, c = (k / N) % Cdp, t = k / (N * Cdp);
    w_dgrad[k] = (c < Cin) ? w[((size_t)n * Cin + c) * taps + (taps - 1 - t)] : 0.f;
  }
  if (b_fwd && idx < Np) {
    const int n = colmap ? colmap[idx] : idx;
    b_fwd[idx] = (bias && n >= 0 && n < N) ? bias[n] : 0.f;
  }
}

int pack_launch(const float* w, const float* bias, int N, int Cin, int ksize, const int* colmap, int Np,
                float* w_fwd, float* b_fwd, int Cdp, float* w_dgrad, hipStream_t st) {
  SININN_CHECK(w != nullptr && N > 0 && Cin > 0 && (ksize == 1 || ksize == 3), "pack: bad arguments");
  SININN_CHECK(!w_fwd || Np >= 1, "pack: bad Np");
  SININN_CHECK(!w_dgrad || Cdp >= Cin, "pack: Cdp < Cin");
  const int taps = ksize * ksize;
  int total = (w_fwd ? taps * Np * Cin : 0) + (w_dgrad ? taps * Cdp * N : 0);
  if (total < Np) total = Np;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, bias, N, Cin, taps, colmap,
                     Np, w_fwd, b_fwd, Cdp, w_dgrad);
  SININN_LAUNCH_CHECK("pack_weights");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// coupling backward tail
// ------------------------------------------------------------------------------------------------
// 4 channels per thread: s / dr / dv (and dy / vy when they are not gathered through a channel map) move as float4
__global__ void coupling_bwd_kernel(const float* __restrict__ dy, int dy_stride, const int* __restrict__ dy_map,
                                    const float* __restrict__ vy, int vy_stride, const int* __restrict__ vy_map,
                                    const float* __restrict__ s, const float* __restrict__ gld, int64_t total4,
                                    int HW, int Co, float clamp, int inverse, float* __restrict__ dr,
                                    float* __restrict__ dv, int dv_stride) {
  const int q = Co >> 2;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % q) * 4;
    const int64_t pix = idx / q;
    f32x4 g, u;
    if (dy_map) {
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = dy[pix * dy_stride + dy_map[c + j]];
    } else {
      g = *reinterpret_cast<const f32x4*>(dy + pix * dy_stride + c);
    }
    if (vy_map) {
#pragma unroll
      for (int j = 0; j < 4; ++j) u[j] = vy[pix * vy_stride + vy_map[c + j]];
    } else {
      u = *reinterpret_cast<const f32x4*>(vy + pix * vy_stride + c);
    }
    const f32x4 sv = *reinterpret_cast<const f32x4*>(s + pix * Co + c);
    const float gl = gld ? gld[pix / HW] : 0.f;
    f32x4 ds, dt, dvv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float L = glow_log_e(sv[j], clamp);
      const float dL = glow_dlog_e(sv[j], clamp);
      const float e = expf(L);
      if (!inverse) { dvv[j] = g[j] * e; dt[j] = g[j]; ds[j] = (g[j] * u[j] * e + gl) * dL; }
      else { dvv[j] = g[j] / e; dt[j] = -dvv[j]; ds[j] = -(g[j] * u[j] + gl) * dL; }
    }
    *reinterpret_cast<f32x4*>(dr + pix * (2 * Co) + c) = ds;
    *reinterpret_cast<f32x4*>(dr + pix * (2 * Co) + Co + c) = dt;
    *reinterpret_cast<f32x4*>(dv + pix * dv_stride + c) = dvv;
  }
}

int coupling_bwd_launch(const float* dy, int dy_stride, const int* dy_map, const float* vy, int vy_stride,
                        const int* vy_map, const float* s, const float* gld, int B, int HW, int Co, float clamp,
                        int inverse, float* dr, float* dv, int dv_stride, hipStream_t st) {
  SININN_CHECK(dy && vy && s && dr && dv, "coupling_bwd: null pointer");
  SININN_CHECK(B > 0 && HW > 0 && Co > 0 && clamp > 0.f, "coupling_bwd: bad shape");
  SININN_CHECK(dy_stride >= Co && vy_stride >= Co && dv_stride >= Co, "coupling_bwd: stride < Co");
  SININN_CHECK(Co % 4 == 0 && dv_stride % 4 == 0 && aligned16(s) && aligned16(dr) && aligned16(dv), "coupling_bwd: Co %% 4 and 16-byte alignment required");
  SININN_CHECK(dy_map || (dy_stride % 4 == 0 && aligned16(dy)), "coupling_bwd: dy must be 16-byte aligned");
  SININN_CHECK(vy_map || (vy_stride % 4 == 0 && aligned16(vy)), "coupling_bwd: vy must be 16-byte aligned");
  const int64_t total = (int64_t)B * HW * (Co / 4);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(coupling_bwd_kernel, dim3(blocks), dim3(256), 0, st, dy, dy_stride, dy_map, vy, vy_stride,
                     vy_map, s, gld, total, HW, Co, clamp, inverse, dr, dv, dv_stride);
  SININN_LAUNCH_CHECK("coupling_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// squeeze / unsqueeze / layout change as one strided gather
// ------------------------------------------------------------------------------------------------
__global__ void squeeze_kernel(const float* __restrict__ in, Str4 is, float* __restrict__ out, Str4 os, int B, int C,
                               int H, int W, int levels, int inverse, const int* __restrict__ chan_map,
                               int map_on_out, int x_fastest) {
  const int64_t total = (int64_t)B * C * H * W;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int b, c, y, x;
    int64_t r = idx;
    if (x_fastest) { x = r % W; r /= W; y = r % H; r /= H; c = r % C; b = (int)(r / C); }
    else { c = r % C; r /= C; x = r % W; r /= W; y = r % H; b = (int)(r / H); }
    // coarse coordinates after `levels` squeezes
    int cc = c, cy = y, cx = x, cch = C;
    for (int l = 0; l < levels; ++l) {
      cc = ((cy & 1) * 2 + (cx & 1)) * cch + cc;
      cy >>= 1; cx >>= 1; cch *= 4;
    }
    int fc = c, qc = cc;              // fine-side / coarse-side channel
    // forward: in = fine, out = coarse ; inverse: in = coarse, out = fine
    if (chan_map) {
      if (!inverse) { if (map_on_out) qc = chan_map[qc]; else fc = chan_map[fc]; }
      else { if (map_on_out) fc = chan_map[fc]; else qc = chan_map[qc]; }
    }
    const int64_t fine = b * (inverse ? os.b : is.b) + fc * (inverse ? os.c : is.c) + y * (inverse ? os.h : is.h) +
                         x * (inverse ? os.w : is.w);
    const int64_t coarse = b * (inverse ? is.b : os.b) + qc * (inverse ? is.c : os.c) +
                           cy * (inverse ? is.h : os.h) + cx * (inverse ? is.w : os.w);
    if (!inverse) out[coarse] = in[fine]; else out[fine] = in[coarse];
  }
}

int squeeze_launch(const float* in, const int64_t is[4], float* out, const int64_t os[4], int B, int C, int H, int W,
                   int levels, int inverse, const int* chan_map, int map_on_out, hipStream_t st) {
  SININN_CHECK(in && out && is && os, "squeeze: null pointer");
  SININN_CHECK(B > 0 && C > 0 && H > 0 && W > 0 && levels >= 0 && levels <= 4, "squeeze: bad shape");
  SININN_CHECK((H % (1 << levels)) == 0 && (W % (1 << levels)) == 0, "squeeze: H=%d W=%d not divisible by 2^%d", H, W, levels);
  const int64_t total = (int64_t)B * C * H * W;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  const int64_t fine_sw = inverse ? os[3] : is[3];
  hipLaunchKernelGGL(squeeze_kernel, dim3(blocks), dim3(256), 0, st, in, mk(is), out, mk(os), B, C, H, W, levels,
                     inverse, chan_map, map_on_out, fine_sw == 1 ? 1 : 0);
  SININN_LAUNCH_CHECK("squeeze");
  return 0;
}

__global__ void permute_kernel(const float* __restrict__ in, int in_stride, float* __restrict__ out, int out_stride,
                               int64_t M, int C, const int* __restrict__ idx) {
  const int64_t total = M * C;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(k % C);
    const int64_t m = k / C;
    out[m * out_stride + j] = in[m * in_stride + idx[j]];
  }
}

int permute_launch(const float* in, int in_stride, float* out, int out_stride, int64_t M, int C, const int* idx,
                   hipStream_t st) {
  SININN_CHECK(in && out && idx && M > 0 && C > 0 && in_stride >= C && out_stride >= C, "permute: bad arguments");
  const int64_t total = M * C;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(permute_kernel, dim3(blocks), dim3(256), 0, st, in, in_stride, out, out_stride, M, C, idx);
  SININN_LAUNCH_CHECK("permute");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// losses
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void decode4(int64_t r, int C, int H, int W, int x_fastest, int& b, int& c, int& y, int& x) {
  if (x_fastest) { x = r % W; r /= W; y = r % H; r /= H; c = r % C; b = (int)(r / C); }
  else { c = r % C; r /= C; x = r % W; r /= W; y = r % H; b = (int)(r / H); }
}

__global__ void sqdiff_sum_kernel(const float* __restrict__ x, Str4 xs, const float* __restrict__ y, Str4 ys, int B,
                                  int C, int H, int W, int x_fastest, float* __restrict__ out) {
  const int64_t total = (int64_t)B * C * H * W;
  float acc = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int b, c, yy, xx;
    decode4(idx, C, H, W, x_fastest, b, c, yy, xx);
    float d = x[b * xs.b + c * xs.c + yy * xs.h + xx * xs.w];
    if (y) d -= y[b * ys.b + c * ys.c + yy * ys.h + xx * ys.w];
    acc += d * d;
  }
  __shared__ float red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

int sqdiff_sum_launch(const float* x, const int64_t xs[4], const float* y, const int64_t ys[4], int B, int C, int H,
                      int W, float* out, hipStream_t st) {
  SININN_CHECK(x && xs && out && B > 0 && C > 0 && H > 0 && W > 0, "sqdiff_sum: bad arguments");
  SININN_CHECK(!y || ys, "sqdiff_sum: y without strides");
  const int64_t total = (int64_t)B * C * H * W;
  // one same-address atomic per block: keep the block count low (2048 of them serialise for ~25 us in L2)
  const int blocks = (int)((total + 1023) / 1024 < 512 ? (total + 1023) / 1024 : 512);
  Str4 yss = y ? mk(ys) : Str4{0, 0, 0, 0};
  hipLaunchKernelGGL(sqdiff_sum_kernel, dim3(blocks), dim3(256), 0, st, x, mk(xs), y, yss, B, C, H, W,
                     xs[3] == 1 ? 1 : 0, out);
  SININN_LAUNCH_CHECK("sqdiff_sum");
  return 0;
}

__global__ void sqdiff_bwd_kernel(const float* __restrict__ x, Str4 xs, const float* __restrict__ y, Str4 ys, int B,
                                  int C, int H, int W, int x_fastest, const float* __restrict__ scale, float gscale,
                                  float* __restrict__ gx, Str4 gxs, float* __restrict__ gy, Str4 gys) {
  const int64_t total = (int64_t)B * C * H * W;
  const float k = scale[0] * gscale;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int b, c, yy, xx;
    decode4(idx, C, H, W, x_fastest, b, c, yy, xx);
    float d = x[b * xs.b + c * xs.c + yy * xs.h + xx * xs.w];
    if (y) d -= y[b * ys.b + c * ys.c + yy * ys.h + xx * ys.w];
    const float g = k * d;
    if (gx) gx[b * gxs.b + c * gxs.c + yy * gxs.h + xx * gxs.w] = g;
    if (gy) gy[b * gys.b + c * gys.c + yy * gys.h + xx * gys.w] = -g;
  }
}

int sqdiff_bwd_launch(const float* x, const int64_t xs[4], const float* y, const int64_t ys[4], int B, int C, int H,
                      int W, const float* scale, float gscale, float* gx, const int64_t gxs[4], float* gy,
                      const int64_t gys[4], hipStream_t st) {
  SININN_CHECK(x && xs && scale && (gx || gy), "sqdiff_bwd: bad arguments");
  SININN_CHECK((!gx || gxs) && (!gy || gys) && (!y || ys), "sqdiff_bwd: missing strides");
  const int64_t total = (int64_t)B * C * H * W;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  Str4 z{0, 0, 0, 0};
  hipLaunchKernelGGL(sqdiff_bwd_kernel, dim3(blocks), dim3(256), 0, st, x, mk(xs), y, y ? mk(ys) : z, B, C, H, W,
                     xs[3] == 1 ? 1 : 0, scale, gscale, gx, gx ? mk(gxs) : z, gy, gy ? mk(gys) : z);
  SININN_LAUNCH_CHECK("sqdiff_bwd");
  return 0;
}

// Gram matrices of loss.mmd: each block stages a chunk of KC logical positions of all B samples
constexpr int MMD_KC = 128;
constexpr int MMD_SLOTS = 16;          // == SININN_MMD_SLOTS
__global__ void mmd_gram_kernel(const float* __restrict__ x, Str4 xs, const float* __restrict__ y, Str4 ys, int B,
                                int C, int H, int W, int x_fastest, float* __restrict__ g) {
  extern __shared__ float sm[];
  float* xl = sm;                   // [B][KC+1]
  float* yl = sm + B * (MMD_KC + 1);
  const int64_t K = (int64_t)C * H * W;
  const int npair = B * B;
  for (int64_t k0 = (int64_t)blockIdx.x * MMD_KC; k0 < K; k0 += (int64_t)gridDim.x * MMD_KC) {
    __syncthreads();
    for (int e = threadIdx.x; e < B * MMD_KC; e += blockDim.x) {
      const int i = e / MMD_KC, kk = e % MMD_KC;
      const int64_t k = k0 + kk;
      float xv = 0.f, yv = 0.f;
      if (k < K) {
        int c, yy, xx; int64_t r = k;
        if (x_fastest) { xx = r % W; r /= W; yy = r % H; c = (int)(r / H); }
        else { c = r % C; r /= C; xx = r % W; yy = (int)(r / W); }
        xv = x[i * xs.b + c * xs.c + yy * xs.h + xx * xs.w];
        yv = y[i * ys.b + c * ys.c + yy * ys.h + xx * ys.w];
      }
      xl[i * (MMD_KC + 1) + kk] = xv;
      yl[i * (MMD_KC + 1) + kk] = yv;
    }
    __syncthreads();
    for (int pr = threadIdx.x; pr < npair; pr += blockDim.x) {
      const int i = pr / B, j = pr % B;
      float sxx = 0.f, syy = 0.f, sxy = 0.f;
      for (int kk = 0; kk < MMD_KC; ++kk) {
        const float xi = xl[i * (MMD_KC + 1) + kk], xj = xl[j * (MMD_KC + 1) + kk];
        const float yi = yl[i * (MMD_KC + 1) + kk], yj = yl[j * (MMD_KC + 1) + kk];
        sxx += xi * xj; syy += yi * yj; sxy += xi * yj;
      }
      // 16 slot copies of the three Grams behind the result: 1024 blocks adding to ONE copy serialise in L2
      float* gs = g + 3 * npair * (1 + (blockIdx.x & (MMD_SLOTS - 1)));
      atomicAdd(gs + pr, sxx);
      atomicAdd(gs + npair + pr, syy);
      atomicAdd(gs + 2 * npair + pr, sxy);
    }
  }
}

__global__ void mmd_slots_kernel(float* __restrict__ g, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 1; k <= MMD_SLOTS; ++k) s += g[(size_t)k * n + i];      // fixed order
  g[i] = s;
}

int mmd_gram_launch(const float* x, const int64_t xs[4], const float* y, const int64_t ys[4], int B, int C, int H,
                    int W, float* g, hipStream_t st) {
  SININN_CHECK(x && y && xs && ys && g, "mmd_gram: null pointer");
  SININN_CHECK(B > 0 && B <= 64 && C > 0 && H > 0 && W > 0, "mmd_gram: batch must be in 1..64 (got %d)", B);
  const int64_t K = (int64_t)C * H * W;
  const int blocks = (int)((K + MMD_KC - 1) / MMD_KC < 1024 ? (K + MMD_KC - 1) / MMD_KC : 1024);
  const size_t lds = (size_t)2 * B * (MMD_KC + 1) * sizeof(float);
  hipLaunchKernelGGL(mmd_gram_kernel, dim3(blocks), dim3(256), lds, st, x, mk(xs), y, mk(ys), B, C, H, W,
                     xs[3] == 1 ? 1 : 0, g);
  SININN_LAUNCH_CHECK("mmd_gram");
  hipLaunchKernelGGL(mmd_slots_kernel, dim3((3 * B * B + 255) / 256), dim3(256), 0, st, g, 3 * B * B);
  SININN_LAUNCH_CHECK("mmd_slots");
  return 0;
}

// loss.py:20-36 on the Grams; also the four BxB matrices the backward needs:
//   gx = AX x + BX y ,  gy = AY x + BY y   (rows = samples)
__global__ void mmd_finish_kernel(const float* __restrict__ g, int B, int rev, float* __restrict__ out,
                                  float* __restrict__ coef) {
  extern __shared__ float sm[];
  float* gxx = sm;            // dL/d dxx (masked), [B][B]
  float* gyy = sm + B * B;
  float* gxy = sm + 2 * B * B;
  __shared__ float red[4];
  const float Cs[2][3] = {{0.2f, 1.5f, 3.0f}, {0.2f, 0.2f, 0.2f}};
  const float As[2][3] = {{2.f, 2.f, 2.f}, {0.1f, 0.5f, 2.f}};
  const int npair = B * B;
  const float inv = 1.f / (float)npair;
  float acc = 0.f;
  for (int pr = threadIdx.x; pr < npair; pr += blockDim.x) {
    const int i = pr / B, j = pr % B;
    const float xx = g[pr], yy = g[npair + pr], xy = g[2 * npair + pr];
    const float xii = g[i * B + i], xjj = g[j * B + j], yii = g[npair + i * B + i], yjj = g[npair + j * B + j];
    const float rxx = xii + xjj - 2.f * xx, ryy = yii + yjj - 2.f * yy, rxy = xii + yjj - 2.f * xy;
    const float dxx = fmaxf(rxx, 0.f), dyy = fmaxf(ryy, 0.f), dxy = fmaxf(rxy, 0.f);
    float kxx = 0.f, kyy = 0.f, kxy = 0.f, pxx = 0.f, pyy = 0.f, pxy = 0.f;
    for (int q = 0; q < 3; ++q) {
      const float Cq = Cs[rev][q], a = As[rev][q];
      const float ca = powf(Cq, a);
      const float bxx = (Cq + dxx) / a, byy = (Cq + dyy) / a, bxy = (Cq + dxy) / a;
      kxx += ca * powf(bxx, -a); kyy += ca * powf(byy, -a); kxy += ca * powf(bxy, -a);
      pxx -= ca * powf(bxx, -a - 1.f); pyy -= ca * powf(byy, -a - 1.f); pxy -= ca * powf(bxy, -a - 1.f);
    }
    acc += kxx + kyy - 2.f * kxy;
    gxx[pr] = (rxx >= 0.f) ? pxx * inv : 0.f;
    gyy[pr] = (ryy >= 0.f) ? pyy * inv : 0.f;
    gxy[pr] = (rxy >= 0.f) ? -2.f * pxy * inv : 0.f;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) * inv;
  if (!coef) return;
  float* AX = coef; float* BX = coef + npair; float* AY = coef + 2 * npair; float* BY = coef + 3 * npair;
  for (int pr = threadIdx.x; pr < npair; pr += blockDim.x) {
    const int i = pr / B, j = pr % B;
    const float sx = gxx[i * B + j] + gxx[j * B + i];
    const float sy = gyy[i * B + j] + gyy[j * B + i];
    float ax = -2.f * sx, by = -2.f * sy;
    if (i == j) {
      float rowx = 0.f, rowy = 0.f;
      for (int k = 0; k < B; ++k) {
        rowx += gxx[i * B + k] + gxx[k * B + i] + gxy[i * B + k];   // d/dx_i of |x_i|^2 terms
        rowy += gyy[i * B + k] + gyy[k * B + i] + gxy[k * B + i];   // d/dy_i of |y_i|^2 terms
      }
      ax += 2.f * rowx; by += 2.f * rowy;
    }
    AX[pr] = ax;
    BX[pr] = -2.f * gxy[i * B + j];   // gx_i <- y_j
    AY[pr] = -2.f * gxy[j * B + i];   // gy_i <- x_j
    BY[pr] = by;
  }
}

int mmd_finish_launch(const float* g, int B, int rev, float* out, float* coef, hipStream_t st) {
  SININN_CHECK(g && out && B > 0 && B <= 64, "mmd_finish: bad arguments");
  hipLaunchKernelGGL(mmd_finish_kernel, dim3(1), dim3(256), (size_t)3 * B * B * sizeof(float), st, g, B, rev ? 1 : 0,
                     out, coef);
  SININN_LAUNCH_CHECK("mmd_finish");
  return 0;
}

__global__ void mmd_bwd_kernel(const float* __restrict__ x, Str4 xs, const float* __restrict__ y, Str4 ys, int B, int C,
                               int H, int W, int x_fastest, const float* __restrict__ coef,
                               const float* __restrict__ scale, float* __restrict__ gx, Str4 gxs,
                               float* __restrict__ gy, Str4 gys) {
  const int64_t K = (int64_t)C * H * W;
  const int64_t total = K * B;
  const int npair = B * B;
  const float sc = scale[0];
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = idx % K;
    const int i = (int)(idx / K);
    int c, yy, xx; int64_t r = k;
    if (x_fastest) { xx = r % W; r /= W; yy = r % H; c = (int)(r / H); }
    else { c = r % C; r /= C; xx = r % W; yy = (int)(r / W); }
    const int64_t ox = c * xs.c + yy * xs.h + xx * xs.w, oy = c * ys.c + yy * ys.h + xx * ys.w;
    float ax = 0.f, ay = 0.f;
    for (int j = 0; j < B; ++j) {
      const float xv = x[j * xs.b + ox], yv = y[j * ys.b + oy];
      ax += coef[i * B + j] * xv + coef[npair + i * B + j] * yv;
      ay += coef[2 * npair + i * B + j] * xv + coef[3 * npair + i * B + j] * yv;
    }
    if (gx) gx[i * gxs.b + c * gxs.c + yy * gxs.h + xx * gxs.w] = sc * ax;
    if (gy) gy[i * gys.b + c * gys.c + yy * gys.h + xx * gys.w] = sc * ay;
  }
}

int mmd_bwd_launch(const float* x, const int64_t xs[4], const float* y, const int64_t ys[4], int B, int C, int H, int W,
                   const float* coef, const float* scale, float* gx, const int64_t gxs[4], float* gy,
                   const int64_t gys[4], hipStream_t st) {
  SININN_CHECK(x && y && xs && ys && coef && scale && (gx || gy), "mmd_bwd: bad arguments");
  SININN_CHECK((!gx || gxs) && (!gy || gys), "mmd_bwd: missing strides");
  const int64_t total = (int64_t)B * C * H * W;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  Str4 z{0, 0, 0, 0};
  hipLaunchKernelGGL(mmd_bwd_kernel, dim3(blocks), dim3(256), 0, st, x, mk(xs), y, mk(ys), B, C, H, W,
                     xs[3] == 1 ? 1 : 0, coef, scale, gx, gx ? mk(gxs) : z, gy, gy ? mk(gys) : z);
  SININN_LAUNCH_CHECK("mmd_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// affine warp (kornia.warp_affine == affine_grid + grid_sample, align_corners=False, zeros padding)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void affine_src(const float* th, int x, int y, int H, int W, float& ix, float& iy) {
  const float xn = (2.f * x + 1.f) / W - 1.f, yn = (2.f * y + 1.f) / H - 1.f;
  const float xs = th[0] * xn + th[1] * yn + th[2];
  const float ys = th[3] * xn + th[4] * yn + th[5];
  ix = ((xs + 1.f) * W - 1.f) * 0.5f;
  iy = ((ys + 1.f) * H - 1.f) * 0.5f;
}

__global__ void affine_warp_kernel(const float* __restrict__ img, Str4 is, const float* __restrict__ theta, int B,
                                   int C, int H, int W, int x_fastest, float* __restrict__ out, Str4 os,
                                   const float* __restrict__ ref, Str4 rs, float* __restrict__ sse) {
  const int64_t total = (int64_t)B * C * H * W;
  float acc = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int b, c, y, x;
    decode4(idx, C, H, W, x_fastest, b, c, y, x);
    float ix, iy;
    affine_src(theta + b * 6, x, y, H, W, ix, iy);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const float* base = img + b * is.b + c * is.c;
    auto tap = [&](int yy, int xx) -> float {
      return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? base[yy * is.h + xx * is.w] : 0.f;
    };
    const float val = tap(y0, x0) * wy0 * wx0 + tap(y0, x0 + 1) * wy0 * wx1 + tap(y0 + 1, x0) * wy1 * wx0 +
                      tap(y0 + 1, x0 + 1) * wy1 * wx1;
    out[b * os.b + c * os.c + y * os.h + x * os.w] = val;
    if (ref) { const float d = val - ref[b * rs.b + c * rs.c + y * rs.h + x * rs.w]; acc += d * d; }
  }
  if (sse) {
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sse, red[0] + red[1] + red[2] + red[3]);
  }
}

int affine_warp_launch(const float* img, const int64_t is[4], const float* theta, int B, int C, int H, int W, float* out,
                       const int64_t os[4], const float* ref, const int64_t rs[4], float* sse, hipStream_t st) {
  SININN_CHECK(img && is && theta && out && os && B > 0 && C > 0 && H > 0 && W > 0, "affine_warp: bad arguments");
  SININN_CHECK((!ref || (rs && sse)), "affine_warp: ref needs strides and sse");
  const int64_t total = (int64_t)B * C * H * W;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  Str4 z{0, 0, 0, 0};
  hipLaunchKernelGGL(affine_warp_kernel, dim3(blocks), dim3(256), 0, st, img, mk(is), theta, B, C, H, W,
                     os[3] == 1 ? 1 : 0, out, mk(os), ref, ref ? mk(rs) : z, ref ? sse : nullptr);
  SININN_LAUNCH_CHECK("affine_warp");
  return 0;
}

__global__ void affine_warp_bwd_kernel(const float* __restrict__ gout, Str4 gs, const float* __restrict__ theta, int B,
                                       int C, int H, int W, int x_fastest, float* __restrict__ gimg, Str4 gis) {
  const int64_t total = (int64_t)B * C * H * W;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int b, c, y, x;
    decode4(idx, C, H, W, x_fastest, b, c, y, x);
    float ix, iy;
    affine_src(theta + b * 6, x, y, H, W, ix, iy);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const float g = gout[b * gs.b + c * gs.c + y * gs.h + x * gs.w];
    float* base = gimg + b * gis.b + c * gis.c;
    auto put = [&](int yy, int xx, float wgt) {
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) atomicAdd(base + yy * gis.h + xx * gis.w, g * wgt);
    };
    put(y0, x0, wy0 * wx0); put(y0, x0 + 1, wy0 * wx1); put(y0 + 1, x0, wy1 * wx0); put(y0 + 1, x0 + 1, wy1 * wx1);
  }
}

int affine_warp_bwd_launch(const float* gout, const int64_t gs[4], const float* theta, int B, int C, int H, int W,
                           float* gimg, const int64_t gis[4], hipStream_t st) {
  SININN_CHECK(gout && gs && theta && gimg && gis && B > 0 && C > 0 && H > 0 && W > 0, "affine_warp_bwd: bad arguments");
  const int64_t total = (int64_t)B * C * H * W;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(affine_warp_bwd_kernel, dim3(blocks), dim3(256), 0, st, gout, mk(gs), theta, B, C, H, W,
                     gs[3] == 1 ? 1 : 0, gimg, mk(gis));
  SININN_LAUNCH_CHECK("affine_warp_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// optical-flow backward warp + photometric L1 (planar NCHW, C <= 8)
//   one lane per output pixel; the two right-hand taps of the 4-tap gather are taken from the next
//   lane's left-hand taps with a wavefront shuffle whenever the neighbour samples the adjacent
//   source column of the same rows (true almost everywhere for a smooth flow), else loaded.
// ------------------------------------------------------------------------------------------------
constexpr int FW_MAXC = 8;
// image-like operands (img / target / warped / gwarped) are fp32 or bf16 (BASELINE configs[3]: the warp runs in the
// bf16 arithmetic of that config: bf16 in HBM, fp32 interpolation, weights, metric and every gradient accumulator);
// flow and metric stay fp32 (geometry / a 1-channel reduction)
__device__ __forceinline__ float fw_ld(const float* p, int64_t i) { return p[i]; }
__device__ __forceinline__ float fw_ld(const __bf16* p, int64_t i) { return (float)p[i]; }
__device__ __forceinline__ void fw_st(float* p, int64_t i, float v) { p[i] = v; }
__device__ __forceinline__ void fw_st(__bf16* p, int64_t i, float v) { p[i] = (__bf16)v; }

template <typename T>
__global__ void flow_warp_l1_kernel(const T* __restrict__ img, const float* __restrict__ flow,
                                    const T* __restrict__ target, int B, int C, int H, int W,
                                    T* __restrict__ warped, float* __restrict__ metric) {
  const int64_t HW = (int64_t)H * W;
  const int64_t total = (int64_t)B * HW;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // grid covers total rounded up to 64
  const bool live = idx < total;
  const int64_t pidx = live ? idx : total - 1;
  const int b = (int)(pidx / HW);
  const int64_t r = pidx % HW;
  const int y = (int)(r / W), x = (int)(r % W);
  const float fx = flow[(b * 2 + 0) * HW + r], fy = flow[(b * 2 + 1) * HW + r];
  // grid = (coords+flow)/(W-1,H-1)*2-1, sampled with align_corners=False  (quirk C-18 kept)
  const float gxn = (x + fx) / (float)(W - 1) * 2.f - 1.f, gyn = (y + fy) / (float)(H - 1) * 2.f - 1.f;
  const float ix = ((gxn + 1.f) * W - 1.f) * 0.5f, iy = ((gyn + 1.f) * H - 1.f) * 0.5f;
  const float flx = floorf(ix), fly = floorf(iy);
  const int x0 = (int)flx, y0 = (int)fly;
  const float wx1 = ix - flx, wy1 = iy - fly, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
  const int lane = threadIdx.x & 63;
  const int nx0 = __shfl_down(x0, 1), ny0 = __shfl_down(y0, 1), nb = __shfl_down(b, 1);
  const bool share = (lane < 63) && (nb == b) && (nx0 == x0 + 1) && (ny0 == y0);
  float l1 = 0.f;
  for (int c = 0; c < C; ++c) {
    const T* base = img + ((int64_t)b * C + c) * HW;
    auto tap = [&](int yy, int xx) -> float {
      return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? fw_ld(base, (int64_t)yy * W + xx) : 0.f;
    };
    const float t00 = tap(y0, x0), t10 = tap(y0 + 1, x0);
    const float n00 = __shfl_down(t00, 1), n10 = __shfl_down(t10, 1);
    const float t01 = share ? n00 : tap(y0, x0 + 1);
    const float t11 = share ? n10 : tap(y0 + 1, x0 + 1);
    float val = t00 * wy0 * wx0 + t01 * wy0 * wx1 + t10 * wy1 * wx0 + t11 * wy1 * wx1;
    if (live) {
      if (warped) {
        fw_st(warped, ((int64_t)b * C + c) * HW + r, val);
        if constexpr (!std::is_same<T, float>::value) val = (float)(__bf16)val;   // the metric sees what was stored
      }
      if (target) l1 += fabsf(fw_ld(target, ((int64_t)b * C + c) * HW + r) - val);
    }
  }
  if (live && metric) metric[(int64_t)b * HW + r] = l1 / (float)C;
}

template <typename T>
static int flow_warp_l1_launch_t(const T* img, const float* flow, const T* target, int B, int C, int H, int W,
                                 T* warped, float* metric, hipStream_t st) {
  SININN_CHECK(img && flow && (warped || metric), "flow_warp_l1: null pointer");
  SININN_CHECK(!metric || target, "flow_warp_l1: metric needs target");
  SININN_CHECK(B > 0 && C > 0 && C <= FW_MAXC && H > 1 && W > 1, "flow_warp_l1: bad shape (C<=8, H,W>1)");
  const int64_t total = (int64_t)B * H * W;
  hipLaunchKernelGGL(flow_warp_l1_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, img, flow, target, B,
                     C, H, W, warped, metric);
  SININN_LAUNCH_CHECK("flow_warp_l1");
  return 0;
}
int flow_warp_l1_launch(const float* img, const float* flow, const float* target, int B, int C, int H, int W,
                        float* warped, float* metric, hipStream_t st) {
  return flow_warp_l1_launch_t<float>(img, flow, target, B, C, H, W, warped, metric, st);
}
int flow_warp_l1_bf16_launch(const void* img, const float* flow, const void* target, int B, int C, int H, int W,
                             void* warped, float* metric, hipStream_t st) {
  return flow_warp_l1_launch_t<__bf16>(static_cast<const __bf16*>(img), flow, static_cast<const __bf16*>(target), B, C, H, W,
                                       static_cast<__bf16*>(warped), metric, st);
}

template <typename T>
__global__ void flow_warp_l1_bwd_kernel(const T* __restrict__ img, const float* __restrict__ flow,
                                        const T* __restrict__ target, const T* __restrict__ warped,
                                        const T* __restrict__ gwarped, const float* __restrict__ gmetric, int B,
                                        int C, int H, int W, float* __restrict__ gimg, float* __restrict__ gflow) {
  const int64_t HW = (int64_t)H * W;
  const int64_t total = (int64_t)B * HW;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int b = (int)(idx / HW);
  const int64_t r = idx % HW;
  const int y = (int)(r / W), x = (int)(r % W);
  const float fx = flow[(b * 2 + 0) * HW + r], fy = flow[(b * 2 + 1) * HW + r];
  const float gxn = (x + fx) / (float)(W - 1) * 2.f - 1.f, gyn = (y + fy) / (float)(H - 1) * 2.f - 1.f;
  const float ix = ((gxn + 1.f) * W - 1.f) * 0.5f, iy = ((gyn + 1.f) * H - 1.f) * 0.5f;
  const float flx = floorf(ix), fly = floorf(iy);
  const int x0 = (int)flx, y0 = (int)fly;
  const float wx1 = ix - flx, wy1 = iy - fly, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
  const float gm = gmetric ? gmetric[(int64_t)b * HW + r] / (float)C : 0.f;
  float gix = 0.f, giy = 0.f;
  for (int c = 0; c < C; ++c) {
    const int64_t o = ((int64_t)b * C + c) * HW;
    float g = gwarped ? fw_ld(gwarped, o + r) : 0.f;
    if (gmetric) {
      const float d = fw_ld(target, o + r) - fw_ld(warped, o + r);
      g += (d > 0.f) ? -gm : ((d < 0.f) ? gm : 0.f);
    }
    const T* base = img + o;
    auto tap = [&](int yy, int xx) -> float {
      return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? fw_ld(base, (int64_t)yy * W + xx) : 0.f;
    };
    const float t00 = tap(y0, x0), t01 = tap(y0, x0 + 1), t10 = tap(y0 + 1, x0), t11 = tap(y0 + 1, x0 + 1);
    gix += g * ((t01 - t00) * wy0 + (t11 - t10) * wy1);
    giy += g * ((t10 - t00) * wx0 + (t11 - t01) * wx1);
    if (gimg) {
      auto put = [&](int yy, int xx, float wgt) {
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) atomicAdd(gimg + o + (int64_t)yy * W + xx, g * wgt);
      };
      put(y0, x0, wy0 * wx0); put(y0, x0 + 1, wy0 * wx1); put(y0 + 1, x0, wy1 * wx0); put(y0 + 1, x0 + 1, wy1 * wx1);
    }
  }
  if (gflow) {
    gflow[(b * 2 + 0) * HW + r] = gix * (float)W / (float)(W - 1);
    gflow[(b * 2 + 1) * HW + r] = giy * (float)H / (float)(H - 1);
  }
}

// Image-gradient form of the backward: block = 32 x 8 output pixels of one image.  The four taps of an output pixel lie
// within a few pixels of it for the trainer's flows, so their contributions are accumulated in an LDS window (tile + 8 pixels
// all round, per channel) with ds_add_f32 and the window is flushed once with row-contiguous global atomics; only taps
// further away go to HBM atomics directly (plain per-tap global atomics made this kernel 17x slower than its flow-only
// form: 1.3 ms at 16x3x512x512).
#ifndef FLOWBWD_TY
#define FLOWBWD_TY 8
#endif
#ifndef FLOWBWD_R
#define FLOWBWD_R 8
#endif
constexpr int FB_TX = 32, FB_TY = FLOWBWD_TY, FB_R = FLOWBWD_R, FB_WX = FB_TX + 2 * FB_R, FB_WY = FB_TY + 2 * FB_R;   // build tunables
template <typename T>
__global__ __launch_bounds__(FB_TX * FB_TY) void flow_warp_l1_bwd_tiled_kernel(
    const T* __restrict__ img, const float* __restrict__ flow, const T* __restrict__ target,
    const T* __restrict__ warped, const T* __restrict__ gwarped, const float* __restrict__ gmetric, int B, int C, int H,
    int W, float* __restrict__ gimg, float* __restrict__ gflow) {
  constexpr int CC = 4;                                                  // channels per window pass
  __shared__ float win[CC][FB_WY][FB_WX];
  const int tiles_x = (W + FB_TX - 1) / FB_TX, tiles_y = (H + FB_TY - 1) / FB_TY;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int bx0 = tx * FB_TX, by0 = ty * FB_TY;
  const int x = bx0 + threadIdx.x % FB_TX, y = by0 + threadIdx.x / FB_TX;
  const int64_t HW = (int64_t)H * W;
  const bool live = x < W && y < H;
  const int64_t r = live ? (int64_t)y * W + x : 0;
  float fx = 0.f, fy = 0.f;
  if (live) { fx = flow[(b * 2 + 0) * HW + r]; fy = flow[(b * 2 + 1) * HW + r]; }
  const float gxn = (x + fx) / (float)(W - 1) * 2.f - 1.f, gyn = (y + fy) / (float)(H - 1) * 2.f - 1.f;
  const float ix = ((gxn + 1.f) * W - 1.f) * 0.5f, iy = ((gyn + 1.f) * H - 1.f) * 0.5f;
  const bool finite = live && fabsf(ix) < 1e9f && fabsf(iy) < 1e9f;      // also false for NaN
  const float flx = finite ? floorf(ix) : 0.f, fly = finite ? floorf(iy) : 0.f;
  const int x0 = (int)flx, y0 = (int)fly;
  const float wx1 = ix - flx, wy1 = iy - fly, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
  const float gm = (live && gmetric) ? gmetric[(int64_t)b * HW + r] / (float)C : 0.f;
  const int wx = x0 - (bx0 - FB_R), wy = y0 - (by0 - FB_R);             // window coordinates of the north-west tap
  // neighbour relations inside the wave (lanes 0..31 = one tile row, 32..63 = the row below it; FB_TX == 32)
  static_assert(FB_TX == 32, "the lane merges assume 32-pixel tile rows");
  const int lane = threadIdx.x & 63;
  const int fin = finite ? 1 : 0;
  // (every shuffle is executed by all 64 lanes before anything is combined: no short-circuit around a cross-lane read)
  const int fin_r = __shfl_down(fin, 1), x0_r = __shfl_down(x0, 1), y0_r = __shfl_down(y0, 1);
  const int fin_d = __shfl_down(fin, 32), x0_d = __shfl_down(x0, 32), y0_d = __shfl_down(y0, 32);
  const bool give_h = finite && (lane & 31) != 31 && fin_r != 0 && x0_r == x0 + 1 && y0_r == y0;
  const bool give_v = finite && lane < 32 && fin_d != 0 && x0_d == x0 && y0_d == y0 + 1;
  const int gh_l = __shfl_up(give_h ? 1 : 0, 1), gv_u = __shfl_up(give_v ? 1 : 0, 32);
  const bool take_h = (lane & 31) != 0 && gh_l != 0;
  const bool take_v = lane >= 32 && gv_u != 0;
  float gix = 0.f, giy = 0.f;
  for (int c0 = 0; c0 < C; c0 += CC) {
    const int cc = min(CC, C - c0);
    for (int e = threadIdx.x; e < cc * FB_WY * FB_WX; e += FB_TX * FB_TY) (&win[0][0][0])[e] = 0.f;
    __syncthreads();
    for (int c = 0; c < cc; ++c) {                      // every lane walks the channels: the merges below are wave-wide
      const int64_t o = ((int64_t)b * C + c0 + c) * HW;
      float c00 = 0.f, c01 = 0.f, c10 = 0.f, c11 = 0.f;
      if (finite) {
        float g = gwarped ? fw_ld(gwarped, o + r) : 0.f;
        if (gmetric) {
          const float d = fw_ld(target, o + r) - fw_ld(warped, o + r);
          g += (d > 0.f) ? -gm : ((d < 0.f) ? gm : 0.f);
        }
        const T* base = img + o;
        auto tap = [&](int yy, int xx) -> float {
          return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? fw_ld(base, (int64_t)yy * W + xx) : 0.f;
        };
        const float t00 = tap(y0, x0), t01 = tap(y0, x0 + 1), t10 = tap(y0 + 1, x0), t11 = tap(y0 + 1, x0 + 1);
        gix += g * ((t01 - t00) * wy0 + (t11 - t10) * wy1);
        giy += g * ((t10 - t00) * wx0 + (t11 - t01) * wx1);
        c00 = g * wy0 * wx0; c01 = g * wy0 * wx1; c10 = g * wy1 * wx0; c11 = g * wy1 * wx1;
      }
      // For a locally smooth flow the four taps of neighbouring output pixels coincide: the bottom taps of a pixel are the top
      // taps of the pixel below it (the wave's other row), its right taps the left taps of its right neighbour.  The
      // contributions are summed across lanes first, so an interior pixel issues ONE LDS atomic per channel instead of four
      // (every lane then adds to a different address: no same-address serialisation inside the wave).
      const float r10 = __shfl_up(c10, 32), r11 = __shfl_up(c11, 32);
      if (take_v) { c00 += r10; c01 += r11; }
      if (give_v) { c10 = 0.f; c11 = 0.f; }
      const float s01 = __shfl_up(c01, 1), s11 = __shfl_up(c11, 1);
      if (take_h) { c00 += s01; c10 += s11; }
      if (give_h) { c01 = 0.f; c11 = 0.f; }
      auto put = [&](int dy, int dx, float val) {
        const int yy = y0 + dy, xx = x0 + dx;
        if (val != 0.f && yy >= 0 && yy < H && xx >= 0 && xx < W) {
          const int qy = wy + dy, qx = wx + dx;
          if (qy >= 0 && qy < FB_WY && qx >= 0 && qx < FB_WX) atomicAdd(&win[c][qy][qx], val);
          else atomicAdd(gimg + o + (int64_t)yy * W + xx, val);
        }
      };
      if (finite) { put(0, 0, c00); put(0, 1, c01); put(1, 0, c10); put(1, 1, c11); }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < cc * FB_WY * FB_WX; e += FB_TX * FB_TY) {
      const float v = (&win[0][0][0])[e];
      if (v != 0.f) {
        const int c = e / (FB_WY * FB_WX), q = e % (FB_WY * FB_WX);
        const int gy = by0 - FB_R + q / FB_WX, gx = bx0 - FB_R + q % FB_WX;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) atomicAdd(gimg + ((int64_t)b * C + c0 + c) * HW + (int64_t)gy * W + gx, v);
      }
    }
    __syncthreads();
  }
  if (live && gflow) {
    gflow[(b * 2 + 0) * HW + r] = gix * (float)W / (float)(W - 1);
    gflow[(b * 2 + 1) * HW + r] = giy * (float)H / (float)(H - 1);
  }
}

template <typename T>
static int flow_warp_l1_bwd_launch_t(const T* img, const float* flow, const T* target, const T* warped,
                                     const T* gwarped, const float* gmetric, int B, int C, int H, int W, float* gimg,
                                     float* gflow, hipStream_t st) {
  SININN_CHECK(img && flow && (gwarped || gmetric) && (gimg || gflow), "flow_warp_l1_bwd: null pointer");
  SININN_CHECK(!gmetric || (target && warped), "flow_warp_l1_bwd: metric gradient needs target and warped");
  SININN_CHECK(B > 0 && C > 0 && C <= FW_MAXC && H > 1 && W > 1, "flow_warp_l1_bwd: bad shape");
  const int64_t total = (int64_t)B * H * W;
  if (gimg) {
    const int tiles = B * ((H + FB_TY - 1) / FB_TY) * ((W + FB_TX - 1) / FB_TX);
    hipLaunchKernelGGL(flow_warp_l1_bwd_tiled_kernel<T>, dim3((unsigned)tiles), dim3(FB_TX * FB_TY), 0, st, img, flow, target, warped,
                       gwarped, gmetric, B, C, H, W, gimg, gflow);
  } else {
    hipLaunchKernelGGL(flow_warp_l1_bwd_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, img, flow,
                       target, warped, gwarped, gmetric, B, C, H, W, gimg, gflow);
  }
  SININN_LAUNCH_CHECK("flow_warp_l1_bwd");
  return 0;
}
int flow_warp_l1_bwd_launch(const float* img, const float* flow, const float* target, const float* warped,
                            const float* gwarped, const float* gmetric, int B, int C, int H, int W, float* gimg,
                            float* gflow, hipStream_t st) {
  return flow_warp_l1_bwd_launch_t<float>(img, flow, target, warped, gwarped, gmetric, B, C, H, W, gimg, gflow, st);
}
int flow_warp_l1_bwd_bf16_launch(const void* img, const float* flow, const void* target, const void* warped,
                                 const void* gwarped, const float* gmetric, int B, int C, int H, int W, float* gimg,
                                 float* gflow, hipStream_t st) {
  typedef const __bf16* P;
  return flow_warp_l1_bwd_launch_t<__bf16>(static_cast<P>(img), flow, static_cast<P>(target), static_cast<P>(warped),
                                           static_cast<P>(gwarped), gmetric, B, C, H, W, gimg, gflow, st);
}

// ------------------------------------------------------------------------------------------------
// frame-window sampler
// ------------------------------------------------------------------------------------------------
__global__ void sample_windows_kernel(const uint8_t* __restrict__ hr_clip, const uint8_t* __restrict__ lr_clip,
                                      const int* __restrict__ idx, int n, int T, int H, int W, int h, int w, int win,
                                      float* __restrict__ hr_out, Str4 hs, float* __restrict__ lr_out, Str4 ls) {
  const int64_t n_hr = (int64_t)n * H * W * 3;
  const int lrc = (2 * win + 1) * 4;
  const int64_t n_lr = (int64_t)n * h * w * lrc;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_hr + n_lr;
       k += (int64_t)gridDim.x * blockDim.x) {
    if (k < n_hr) {
      int64_t r = k;
      const int c = r % 3; r /= 3;
      const int x = r % W; r /= W;
      const int y = r % H;
      const int s = (int)(r / H);
      int t = idx[s]; t = t < 0 ? 0 : (t >= T ? T - 1 : t);
      hr_out[s * hs.b + c * hs.c + y * hs.h + x * hs.w] = (float)hr_clip[(((int64_t)t * H + y) * W + x) * 3 + c] / 255.f;
    } else {
      int64_t r = k - n_hr;
      const int c = r % lrc; r /= lrc;
      const int x = r % w; r /= w;
      const int y = r % h;
      const int s = (int)(r / h);
      int t = idx[s] - win + c / 4; t = t < 0 ? 0 : (t >= T ? T - 1 : t);
      lr_out[s * ls.b + c * ls.c + y * ls.h + x * ls.w] = (float)lr_clip[(((int64_t)t * h + y) * w + x) * 4 + (c & 3)] / 255.f;
    }
  }
}

bool sample_windows_dense_try(const uint8_t* hr_clip, const uint8_t* lr_clip, const int* idx, int n, int T, int H, int W, int h,
                              int w, int win, float* hr_out, const int64_t hs[4], float* lr_out, const int64_t ls[4],
                              hipStream_t st);

int sample_windows_launch(const uint8_t* hr_clip, const uint8_t* lr_clip, const int* idx, int n, int T, int H, int W,
                          int h, int w, int win, float* hr_out, const int64_t hs[4], float* lr_out,
                          const int64_t ls[4], hipStream_t st) {
  SININN_CHECK(hr_clip && lr_clip && idx && hr_out && lr_out && hs && ls, "sample_windows: null pointer");
  SININN_CHECK(n > 0 && T > 0 && H > 0 && W > 0 && h > 0 && w > 0 && win >= 0, "sample_windows: bad shape");
  if (sample_windows_dense_try(hr_clip, lr_clip, idx, n, T, H, W, h, w, win, hr_out, hs, lr_out, ls, st)) {
    SININN_LAUNCH_CHECK("sample_windows_dense");
    return 0;
  }
  const int64_t total = (int64_t)n * H * W * 3 + (int64_t)n * h * w * (2 * win + 1) * 4;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(sample_windows_kernel, dim3(blocks), dim3(256), 0, st, hr_clip, lr_clip, idx, n, T, H, W, h, w,
                     win, hr_out, mk(hs), lr_out, mk(ls));
  SININN_LAUNCH_CHECK("sample_windows");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam semantics, L2 weight decay)
// ------------------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                            float bc1, float sqrt_bc2, float gscale) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float step_size = lr / bc1;
  auto upd = [&](float& pv, float gv, float& mv, float& vv) {
    const float gg = gv * gscale + wd * pv;
    mv = b1 * mv + (1.f - b1) * gg;
    vv = b2 * vv + (1.f - b2) * gg * gg;
    const float denom = sqrtf(vv) / sqrt_bc2 + eps;
    pv -= step_size * (mv / denom);
  };
  for (int64_t i = tid; i < n4; i += stride) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i], mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) { float a = pv[k], bm = mv[k], bv = vv[k]; upd(a, gv[k], bm, bv); pv[k] = a; mv[k] = bm; vv[k] = bv; }
    reinterpret_cast<f32x4*>(p)[i] = pv; reinterpret_cast<f32x4*>(m)[i] = mv; reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  for (int64_t i = (n4 << 2) + tid; i < n; i += stride) upd(p[i], g[i], m[i], v[i]);
}

int adam_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                float wd, int step, float gscale, hipStream_t st) {
  SININN_CHECK(p && g && m && v && n > 0 && step >= 1, "adam: bad arguments");
  SININN_CHECK(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "adam: buffers must be 16-byte aligned");
  // bias corrections in double like torch.optim.Adam's python-float arithmetic
  const float bc1 = (float)(1.0 - pow((double)b1, (double)step));
  const float bc2 = (float)(1.0 - pow((double)b2, (double)step));
  const int blocks = (int)(((n >> 2) + 255) / 256 < 4096 ? ((n >> 2) + 255) / 256 : 4096);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd,
                     bc1, sqrtf(bc2), gscale);
  SININN_LAUNCH_CHECK("adam");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// IRN pieces (archs.py:135-199)
// ------------------------------------------------------------------------------------------------
__global__ void haar_kernel(const float* __restrict__ in, Str4 is, float* __restrict__ out, Str4 os, int B, int C, int H,
                            int W, int inverse, int c_fastest) {
  // one thread per (b, c, i, j) of the COARSE grid; H, W are the fine sizes
  const int h2 = H / 2, w2 = W / 2;
  const int64_t total = (int64_t)B * C * h2 * w2;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int b, c, i, j;
    int64_t r = idx;
    if (c_fastest) { c = r % C; r /= C; j = r % w2; r /= w2; i = r % h2; b = (int)(r / h2); }
    else { j = r % w2; r /= w2; i = r % h2; r /= h2; c = r % C; b = (int)(r / C); }
    if (!inverse) {
      const float* p = in + b * is.b + c * is.c + (2 * i) * is.h + (2 * j) * is.w;
      const float a = p[0], bb = p[is.w], cc = p[is.h], d = p[is.h + is.w];
      float* o = out + b * os.b + i * os.h + j * os.w;
      o[(0 * C + c) * os.c] = (a + bb + cc + d) * 0.25f;
      o[(1 * C + c) * os.c] = (a - bb + cc - d) * 0.25f;
      o[(2 * C + c) * os.c] = (a + bb - cc - d) * 0.25f;
      o[(3 * C + c) * os.c] = (a - bb - cc + d) * 0.25f;
    } else {
      const float* p = in + b * is.b + i * is.h + j * is.w;
      const float ll = p[(0 * C + c) * is.c], b1 = p[(1 * C + c) * is.c], b2 = p[(2 * C + c) * is.c], b3 = p[(3 * C + c) * is.c];
      float* o = out + b * os.b + c * os.c + (2 * i) * os.h + (2 * j) * os.w;
      o[0] = ll + b1 + b2 + b3;
      o[os.w] = ll - b1 + b2 - b3;
      o[os.h] = ll + b1 - b2 - b3;
      o[os.h + os.w] = ll - b1 - b2 + b3;
    }
  }
}

int haar_launch(const float* in, const int64_t is[4], float* out, const int64_t os[4], int B, int C, int H, int W,
                int inverse, hipStream_t st) {
  SININN_CHECK(in && out && is && os && B > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "haar: bad arguments");
  const int64_t total = (int64_t)B * C * (H / 2) * (W / 2);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  const int64_t coarse_sc = inverse ? is[1] : os[1];
  hipLaunchKernelGGL(haar_kernel, dim3(blocks), dim3(256), 0, st, in, mk(is), out, mk(os), B, C, H, W, inverse,
                     coarse_sc == 1 ? 1 : 0);
  SININN_LAUNCH_CHECK("haar");
  return 0;
}

__global__ void lrelu_bwd_kernel(float* __restrict__ g, int g_stride, const float* __restrict__ f, int f_stride,
                                 int64_t total, int n, float slope) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(idx % n);
    const int64_t m = idx / n;
    if (!(f[m * f_stride + j] > 0.f)) g[m * g_stride + j] *= slope;
  }
}

int lrelu_bwd_launch(float* g, int g_stride, const float* f, int f_stride, int64_t M, int n, float slope, hipStream_t st) {
  SININN_CHECK(g && f && M > 0 && n > 0 && g_stride >= n && f_stride >= n, "lrelu_bwd: bad arguments");
  const int64_t total = M * n;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(blocks), dim3(256), 0, st, g, g_stride, f, f_stride, total, n, slope);
  SININN_LAUNCH_CHECK("lrelu_bwd");
  return 0;
}

__global__ void irn_coupling_bwd_kernel(const float* __restrict__ dy, int dy_stride, const float* __restrict__ vy,
                                        int vy_stride, const float* __restrict__ hval, int64_t total, int Co, float clamp,
                                        int inverse, float* __restrict__ dG, int dG_stride, int dG_pad,
                                        float* __restrict__ dh, float* __restrict__ dv, int dv_stride) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % Co);
    const int64_t m = idx / Co;
    const float g = dy[m * dy_stride + c], u = vy[m * vy_stride + c], hv = hval[idx];
    const float sg = 1.f / (1.f + expf(-hv));
    const float s = clamp * (2.f * sg - 1.f);
    const float ds_dh = clamp * 2.f * sg * (1.f - sg);
    const float e = expf(s);
    float gG, gs, gv;
    if (!inverse) { gv = g * e; gG = g; gs = g * u * e; }      // y = v e + G
    else { gv = g / e; gG = -gv; gs = -g * u; }                // y = (v - G)/e ; d/ds = -y
    dG[m * dG_stride + c] = gG;
    if (c == Co - 1)                                           // zero pad columns [Co, dG_pad) of the row
      for (int j = Co; j < dG_pad; ++j) dG[m * dG_stride + j] = 0.f;
    dh[idx] = gs * ds_dh;
    dv[m * dv_stride + c] = gv;
  }
}

// Stand-alone InvBlockExp tail (archs.py:152-156): out = v * exp(s) + g (inverse == 0) or (v - g) / exp(s) (inverse == 1),
// s = clamp * (2 sigmoid(h) - 1).  The DenseBlock executor fuses this into conv5's epilogue; the stand-alone form lets the H and
// G DenseBlocks of a block run on two streams (the tail then joins them).  float4 per lane; Co % 4 == 0.
__global__ void irn_tail_kernel(const float* __restrict__ v, int v_stride, const float* __restrict__ h, const float* __restrict__ g,
                                int64_t total4, int Co4, float clamp, int inverse, float* __restrict__ out, int out_stride) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % Co4);
    const int64_t m = i / Co4;
    const f32x4 vv = *reinterpret_cast<const f32x4*>(v + m * v_stride + c4 * 4);
    const f32x4 hv = *reinterpret_cast<const f32x4*>(h + i * 4);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i * 4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sv = clamp * (2.f / (1.f + expf(-hv[j])) - 1.f);
      o[j] = inverse ? (vv[j] - gv[j]) / expf(sv) : vv[j] * expf(sv) + gv[j];
    }
    *reinterpret_cast<f32x4*>(out + m * out_stride + c4 * 4) = o;
  }
}

int irn_tail_launch(const float* v, int v_stride, const float* h, const float* g, int64_t M, int Co, float clamp, int inverse,
                    float* out, int out_stride, hipStream_t st) {
  SININN_CHECK(v && h && g && out && M > 0 && Co > 0 && Co % 4 == 0 && clamp > 0.f, "irn_tail: bad arguments");
  SININN_CHECK(v_stride >= Co && out_stride >= Co && v_stride % 4 == 0 && out_stride % 4 == 0, "irn_tail: strides");
  SININN_CHECK(aligned16(v) && aligned16(h) && aligned16(g) && aligned16(out), "irn_tail: 16-byte alignment");
  const int64_t total4 = M * (Co / 4);
  const int blocks = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
  hipLaunchKernelGGL(irn_tail_kernel, dim3(blocks), dim3(256), 0, st, v, v_stride, h, g, total4, Co / 4, clamp, inverse, out, out_stride);
  SININN_LAUNCH_CHECK("irn_tail");
  return 0;
}

int irn_coupling_bwd_launch(const float* dy, int dy_stride, const float* vy, int vy_stride, const float* hval, int64_t M,
                            int Co, float clamp, int inverse, float* dG, int dG_stride, int dG_pad, float* dh, float* dv,
                            int dv_stride, hipStream_t st) {
  // dG rows have dG_stride floats; columns [Co, dG_pad) are written as zeros (K padding of the following conv)
  SININN_CHECK(dy && vy && hval && dG && dh && dv && M > 0 && Co > 0, "irn_coupling_bwd: bad arguments");
  SININN_CHECK(dy_stride >= Co && vy_stride >= Co && dv_stride >= Co, "irn_coupling_bwd: stride < Co");
  SININN_CHECK(dG_pad >= Co && dG_stride >= dG_pad, "irn_coupling_bwd: dG_stride=%d dG_pad=%d Co=%d", dG_stride, dG_pad, Co);
  const int64_t total = M * Co;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(irn_coupling_bwd_kernel, dim3(blocks), dim3(256), 0, st, dy, dy_stride, vy, vy_stride, hval, total,
                     Co, clamp, inverse, dG, dG_stride, dG_pad, dh, dv, dv_stride);
  SININN_LAUNCH_CHECK("irn_coupling_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// On-device LR synthesis (datasets/prepare.py:35-82,147-165): RGGB Bayer sampling of an RGB frame followed by
// `scale` x `scale` binning of each Bayer plane, quantised exactly like the reference: v/255 in float64, mean (or sum)
// over the block rows then over the block columns (numpy's order), clip to [0,1], *255, truncate to uint8.
//   hr (T,H,W,3) u8  ->  lr (T, H/(2*scale), W/(2*scale), 4) u8, channel k = Bayer plane (R, G1, G2, B)
// ------------------------------------------------------------------------------------------------
__global__ void bayer_bin_kernel(const uint8_t* __restrict__ hr, uint8_t* __restrict__ lr, int T, int H, int W, int scale,
                                 int reduce_sum) {
  const int h = H / (2 * scale), w = W / (2 * scale);
  const int64_t total = (int64_t)T * h * w * 4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx & 3);
    int64_t r = idx >> 2;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h);
    const int t = (int)(r / h);
    const int py = k >> 1, px = k & 1;                    // position inside the 2x2 RGGB cell
    const int ch = (k == 0) ? 0 : ((k == 3) ? 2 : 1);     // R, G, G, B
    double acc_cols = 0.0;
    for (int j = 0; j < scale; ++j) {                     // second reduction: over block columns
      double acc_rows = 0.0;
      for (int i = 0; i < scale; ++i) {                   // first reduction: over block rows
        const int yy = 2 * (y * scale + i) + py, xx = 2 * (x * scale + j) + px;
        acc_rows += (double)hr[(((int64_t)t * H + yy) * W + xx) * 3 + ch] / 255.0;
      }
      acc_cols += reduce_sum ? acc_rows : acc_rows / (double)scale;
    }
    double v = reduce_sum ? acc_cols : acc_cols / (double)scale;
    v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    lr[idx] = (uint8_t)(v * 255.0);
  }
}

int bayer_bin_launch(const uint8_t* hr, uint8_t* lr, int T, int H, int W, int scale, int reduce_sum, hipStream_t st) {
  SININN_CHECK(hr && lr && T > 0 && H > 0 && W > 0 && scale > 0, "bayer_bin: bad arguments");
  SININN_CHECK(H % (2 * scale) == 0 && W % (2 * scale) == 0, "bayer_bin: H and W must be multiples of 2*scale (prepare.py:152)");
  const int64_t total = (int64_t)T * (H / (2 * scale)) * (W / (2 * scale)) * 4;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(bayer_bin_kernel, dim3(blocks), dim3(256), 0, st, hr, lr, T, H, W, scale, reduce_sum);
  SININN_LAUNCH_CHECK("bayer_bin");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Demosaiced LR preview (datasets/prepare.py:103-119,158,163-165): the UNQUANTISED binned RGGB planes are packed back into
// a Bayer mosaic and bilinearly demosaiced (colour_demosaicing 0.1.6 `demosaicing_CFA_Bayer_bilinear`, pattern RGGB:
// R / B = conv(CFA * mask, [[1,2,1],[2,4,2],[1,2,1]]/4), G = conv(CFA * mask, [[0,1,0],[1,4,1],[0,1,0]]/4), scipy's default
// 'reflect' boundary), then clipped and quantised.  float64 and unfused multiply-adds in scipy's accumulation order, so the
// bytes match the host pipeline.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double binned_value(const uint8_t* hr, int t, int H, int W, int scale, int reduce_sum, int my, int mx) {
  const int py = my & 1, px = mx & 1, y = my >> 1, x = mx >> 1;
  const int k = py * 2 + px;
  const int ch = (k == 0) ? 0 : ((k == 3) ? 2 : 1);
  double acc_cols = 0.0;
  for (int j = 0; j < scale; ++j) {
    double acc_rows = 0.0;
    for (int i = 0; i < scale; ++i) {
      const int yy = 2 * (y * scale + i) + py, xx = 2 * (x * scale + j) + px;
      acc_rows += (double)hr[(((int64_t)t * H + yy) * W + xx) * 3 + ch] / 255.0;
    }
    acc_cols += reduce_sum ? acc_rows : acc_rows / (double)scale;
  }
  return reduce_sum ? acc_cols : acc_cols / (double)scale;
}

__global__ void bayer_demosaic_kernel(const uint8_t* __restrict__ hr, uint8_t* __restrict__ rgb, int T, int H, int W, int scale,
                                      int reduce_sum) {
  const int mh = H / scale, mw = W / scale;               // mosaic = 2 * (H / (2 scale))
  const int64_t total = (int64_t)T * mh * mw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx;
    const int mx = (int)(r % mw); r /= mw;
    const int my = (int)(r % mh);
    const int t = (int)(r / mh);
    double acc[3] = {0.0, 0.0, 0.0};
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        int sy = my + dy, sx = mx + dx;                   // scipy 'reflect': (d c b a | a b c d | d c b a)
        sy = sy < 0 ? -sy - 1 : (sy >= mh ? 2 * mh - 1 - sy : sy);
        sx = sx < 0 ? -sx - 1 : (sx >= mw ? 2 * mw - 1 - sx : sx);
        const double v = binned_value(hr, t, H, W, scale, reduce_sum, sy, sx);
        const int cell = (sy & 1) * 2 + (sx & 1);         // colour of the SOURCE mosaic pixel: 0 R, 1/2 G, 3 B
        const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
        const double w_rb = (ady == 0 ? 2.0 : 1.0) * (adx == 0 ? 2.0 : 1.0) / 4.0;
        const double w_g = (ady + adx == 0) ? 1.0 : ((ady + adx == 1) ? 0.25 : 0.0);
        acc[0] = __dadd_rn(acc[0], __dmul_rn(cell == 0 ? v : 0.0, w_rb));
        acc[1] = __dadd_rn(acc[1], __dmul_rn((cell == 1 || cell == 2) ? v : 0.0, w_g));
        acc[2] = __dadd_rn(acc[2], __dmul_rn(cell == 3 ? v : 0.0, w_rb));
      }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double v = acc[c];
      v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
      rgb[idx * 3 + c] = (uint8_t)(v * 255.0);
    }
  }
}

int bayer_demosaic_launch(const uint8_t* hr, uint8_t* rgb, int T, int H, int W, int scale, int reduce_sum, hipStream_t st) {
  SININN_CHECK(hr && rgb && T > 0 && H > 0 && W > 0 && scale > 0, "bayer_demosaic: bad arguments");
  SININN_CHECK(H % (2 * scale) == 0 && W % (2 * scale) == 0, "bayer_demosaic: H and W must be multiples of 2*scale (prepare.py:152)");
  const int64_t total = (int64_t)T * (H / scale) * (W / scale);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(bayer_demosaic_kernel, dim3(blocks), dim3(256), 0, st, hr, rgb, T, H, W, scale, reduce_sum);
  SININN_LAUNCH_CHECK("bayer_demosaic");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Winograd F(2x2,3x3) filter transform U = G g G^T (G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]), packed like the
// direct weights with the tap axis replaced by the 16 transform positions:
//   u_fwd  [16][Cin/8][Np ][8]  g[a][b] = w[colmap[q]][c][a][b]
//   u_dgrad[16][N/8  ][Cdp][8]  g[a][b] = w[n][c][2-a][2-b]   (data-gradient conv: flipped taps, channel roles swapped)
// channel-chunk-major: the 8-channel chunk the kernel stages per iteration is one contiguous [columns][8] slab per position,
// so its loads use whole cache lines (a [column][Cin] row layout gave every wave-level load 32 lines, a quarter used each)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wino_filter(const float* g, float* u) {   // g[9] row-major -> u[16]
  float t[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    t[0][b] = g[b];
    t[1][b] = 0.5f * (g[b] + g[3 + b] + g[6 + b]);
    t[2][b] = 0.5f * (g[b] - g[3 + b] + g[6 + b]);
    t[3][b] = g[6 + b];
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    u[a * 4 + 0] = t[a][0];
    u[a * 4 + 1] = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
    u[a * 4 + 2] = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
    u[a * 4 + 3] = t[a][2];
  }
}

__global__ void pack_winograd_kernel(const float* __restrict__ w, int N, int Cin, const int* __restrict__ colmap,
                                     int Np, float* __restrict__ u_fwd, int Cdp, float* __restrict__ u_dgrad) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int nf = u_fwd ? Np * Cin : 0;
  const int nd = u_dgrad ? Cdp * N : 0;
  float g[9], u[16];
  if (idx < nf) {
    const int c = idx % Cin, q = idx / Cin;
    const int n = colmap ? colmap[q] : q;
    const bool ok = n >= 0 && n < N;
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t] = ok ? w[((size_t)n * Cin + c) * 9 + t] : 0.f;
    wino_filter(g, u);
#pragma unroll
    for (int pz = 0; pz < 16; ++pz) u_fwd[(((size_t)pz * (Cin / 8) + c / 8) * Np + q) * 8 + (c & 7)] = u[pz];
  } else if (idx < nf + nd) {
    const int k = idx - nf;
    const int n = k % N, c = k / N;
    const bool ok = c < Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t] = ok ? w[((size_t)n * Cin + c) * 9 + (8 - t)] : 0.f;
    wino_filter(g, u);
#pragma unroll
    for (int pz = 0; pz < 16; ++pz) u_dgrad[(((size_t)pz * (N / 8) + n / 8) * Cdp + c) * 8 + (n & 7)] = u[pz];
  }
}

// ---- batched packs: one launch refreshes every pack of a model ------------------------------------------------------
__host__ __device__ static inline void pack_regions(const sininn_pack_desc& d, int& nf, int& nd, int& nb) {
  const int taps = d.ksize * d.ksize;
  nf = d.w_fwd ? (d.wino_fwd ? d.Np * d.Cin : taps * d.Np * d.Cin) : 0;
  nd = d.w_dgrad ? (d.wino_dgrad ? d.Cdp * d.N : taps * d.Cdp * d.N) : 0;
  nb = d.b_fwd ? d.Np : 0;
}

int pack_work_items(const sininn_pack_desc* d) {
  int nf, nd, nb;
  pack_regions(*d, nf, nd, nb);
  return nf + nd + nb;
}

// source element of packed (output nn, input channel c, tap t): the descriptor's N / Cin are the PACKED dimensions; the
// source weight has src_n <= N outputs (0 = N) and Cin - gap_len input channels, the packed channels [gap_begin, gap_begin +
// gap_len) being zero padding with no counterpart in it (IRN DenseBlock feature buffer, cin padded to a multiple of 8)
__device__ __forceinline__ float pack_src(const sininn_pack_desc& d, int nn, int c, int t, int taps) {
  int sc = c;
  if (d.gap_len > 0) {
    if (c >= d.gap_begin + d.gap_len) sc = c - d.gap_len;
    else if (c >= d.gap_begin) return 0.f;
  }
  if (nn >= (d.src_n > 0 ? d.src_n : d.N)) return 0.f;
  return d.w[((size_t)nn * (d.Cin - d.gap_len) + sc) * taps + t];
}

__global__ void pack_batch_kernel(const sininn_pack_desc* __restrict__ descs, int n, int total) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int lo = 0, hi = n - 1;                              // last descriptor with work_begin <= idx
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].work_begin <= idx) lo = mid; else hi = mid - 1;
  }
  const sininn_pack_desc d = descs[lo];
  int k = idx - d.work_begin;
  int nf, nd, nb;
  pack_regions(d, nf, nd, nb);
  const int taps = d.ksize * d.ksize;
  float g[9], u[16];
  if (k < nf) {
    if (d.wino_fwd) {
      const int c = k % d.Cin, q = k / d.Cin;
      const int nn = d.colmap ? d.colmap[q] : q;
      const bool ok = nn >= 0 && nn < d.N;
#pragma unroll
      for (int t = 0; t < 9; ++t) g[t] = ok ? pack_src(d, nn, c, t, 9) : 0.f;
      wino_filter(g, u);
#pragma unroll
      for (int pz = 0; pz < 16; ++pz) d.w_fwd[(((size_t)pz * (d.Cin / 8) + c / 8) * d.Np + q) * 8 + (c & 7)] = u[pz];
    } else {
      const int c = k % d.Cin, q = (k / d.Cin) % d.Np, t = k / (d.Cin * d.Np);
      const int nn = d.colmap ? d.colmap[q] : q;
      d.w_fwd[k] = (nn >= 0 && nn < d.N) ? pack_src(d, nn, c, t, taps) : 0.f;
    }
    return;
  }
  k -= nf;
  if (k < nd) {
    if (d.wino_dgrad) {
      const int nn = k % d.N, c = k / d.N;
      const bool ok = c < d.Cin;
#pragma unroll
      for (int t = 0; t < 9; ++t) g[t] = ok ? pack_src(d, nn, c, 8 - t, 9) : 0.f;
      wino_filter(g, u);
#pragma unroll
      for (int pz = 0; pz < 16; ++pz) d.w_dgrad[(((size_t)pz * (d.N / 8) + nn / 8) * d.Cdp + c) * 8 + (nn & 7)] = u[pz];
    } else {
      const int nn = k % d.N, c = (k / d.N) % d.Cdp, t = k / (d.N * d.Cdp);
      d.w_dgrad[k] = (c < d.Cin) ? pack_src(d, nn, c, taps - 1 - t, taps) : 0.f;
    }
    return;
  }
  k -= nd;
  if (k < nb) {
    const int nn = d.colmap ? d.colmap[k] : k;
    d.b_fwd[k] = (d.bias && nn >= 0 && nn < (d.src_n > 0 ? d.src_n : d.N)) ? d.bias[nn] : 0.f;
  }
}

int pack_batch_launch(const sininn_pack_desc* descs, int n, int total, hipStream_t st) {
  SININN_CHECK(descs != nullptr && n > 0 && total > 0, "pack_batch: bad arguments");
  hipLaunchKernelGGL(pack_batch_kernel, dim3((total + 255) / 256), dim3(256), 0, st, descs, n, total);
  SININN_LAUNCH_CHECK("pack_batch");
  return 0;
}

int pack_winograd_launch(const float* w, int N, int Cin, const int* colmap, int Np, float* u_fwd, int Cdp,
                         float* u_dgrad, hipStream_t st) {
  SININN_CHECK(w != nullptr && N > 0 && Cin > 0 && (u_fwd || u_dgrad), "pack_winograd: bad arguments");
  SININN_CHECK(!u_fwd || Np >= 1, "pack_winograd: bad Np");
  SININN_CHECK(!u_dgrad || Cdp >= Cin, "pack_winograd: Cdp < Cin");
  SININN_CHECK((!u_fwd || Cin % 8 == 0) && (!u_dgrad || N % 8 == 0), "pack_winograd: the contraction axis must be a multiple of 8");
  const int total = (u_fwd ? Np * Cin : 0) + (u_dgrad ? Cdp * N : 0);
  hipLaunchKernelGGL(pack_winograd_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, N, Cin, colmap, Np, u_fwd,
                     Cdp, u_dgrad);
  SININN_LAUNCH_CHECK("pack_winograd");
  return 0;
}

}  // namespace sininn

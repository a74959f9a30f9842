// Round-2 HBM-bound kernels: every lane moves 16 bytes per access, pixel-major on both sides.
//   frames_to_u8 : float frames -> uint8 (H,W,C) images on the device (SingleVideoINN.infer, lit_wrapper.py:117-124)
#include "common.h"

namespace sininn {

struct Str4 { int64_t b, c, h, w; };
static inline Str4 mk4(const int64_t s[4]) { return Str4{s[0], s[1], s[2], s[3]}; }

// ------------------------------------------------------------------------------------------------
// float frame -> uint8 pixel conversion of the inference path.
//   wrap == 0: clamp(x, 0, 1) * 255, truncated              (what an image writer expects)
//   wrap != 0: (uint8)(int)(x * 255), i.e. torchvision ToPILImage's pic.mul(255).byte() with its wrap-around
//              on out-of-range values (the reference's behaviour, lit_wrapper.py:94,120)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned to_u8(float v, int wrap) {
  if (!wrap) {
    v = fminf(fmaxf(v, 0.f), 1.f) * 255.f;
    return (unsigned)(int)v;
  }
  // C float -> integer conversion truncates toward zero; keeping the low byte reproduces .byte() on x86 hosts
  return (unsigned)((int)(v * 255.f)) & 255u;
}

__global__ void frames_to_u8_dense_kernel(const float* __restrict__ in, uint8_t* __restrict__ out, int64_t n16, int64_t n,
                                          int wrap) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    const f32x4* src = reinterpret_cast<const f32x4*>(in) + i * 4;
    unsigned w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = src[q];
      w[q] = to_u8(v[0], wrap) | (to_u8(v[1], wrap) << 8) | (to_u8(v[2], wrap) << 16) | (to_u8(v[3], wrap) << 24);
    }
    reinterpret_cast<uint4*>(out)[i] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  if (blockIdx.x == 0) {                                   // n % 16 trailing elements
    const int64_t tail = n16 * 16 + threadIdx.x;
    if (tail < n) out[tail] = (uint8_t)to_u8(in[tail], wrap);
  }
}

__global__ void frames_to_u8_strided_kernel(const float* __restrict__ in, Str4 is, uint8_t* __restrict__ out, int B, int C,
                                            int H, int W, int wrap) {
  const int64_t total = (int64_t)B * C * H * W;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = k;
    const int c = r % C; r /= C;
    const int x = r % W; r /= W;
    const int y = r % H;
    const int b = (int)(r / H);
    out[k] = (uint8_t)to_u8(in[b * is.b + c * is.c + y * is.h + x * is.w], wrap);
  }
}

int frames_to_u8_launch(const float* in, const int64_t is[4], uint8_t* out, int B, int C, int H, int W, int wrap,
                        hipStream_t st) {
  SININN_CHECK(in && is && out && B > 0 && C > 0 && H > 0 && W > 0, "frames_to_u8: bad arguments");
  const int64_t n = (int64_t)B * C * H * W;
  const bool dense = is[1] == 1 && is[3] == C && is[2] == (int64_t)W * C && is[0] == (int64_t)H * W * C &&
                     aligned16(in) && aligned16(out);
  if (dense) {
    const int64_t n16 = n / 16;
    int64_t blocks = (n16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    SININN_CHECK(n - n16 * 16 <= 256, "frames_to_u8: internal");
    hipLaunchKernelGGL(frames_to_u8_dense_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, out, n16, n, wrap);
  } else {
    const int64_t blocks = (n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192;
    hipLaunchKernelGGL(frames_to_u8_strided_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, mk4(is), out, B, C, H, W,
                       wrap);
  }
  SININN_LAUNCH_CHECK("frames_to_u8");
  return 0;
}


// ------------------------------------------------------------------------------------------------
// Squeeze / unsqueeze / channel permutation between two DENSE pixel-major tensors, gather form: every thread produces
// four consecutive floats of the output (one 16-byte store) and computes where each comes from.  `map` (may be NULL) acts
// on the channel index of the FINE tensor: forward  coarse[.., q*C + c] = fine[.., map[c]],
//                                          inverse  fine[.., c] = coarse[.., q*C + map[c]]   (q = the 2x2 sub-position digits).
// levels == 0 is the plain permutation out[.., c] = in[.., map[c]].  (IRevNetDownsampling + PermuteRandom, archs.py:28-38,65-68.)
// ------------------------------------------------------------------------------------------------
__global__ void squeeze_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int H, int W,
                                    int levels, int inverse, const int* __restrict__ map, int64_t total4) {
  const int CC = C << (2 * levels);
  const int h = H >> levels, w = W >> levels;
  const bool vec = (map == nullptr) && (C % 4 == 0);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = i * 4;
    f32x4 val;
    if (!inverse) {
      // output = coarse [B][h][w][CC]
      const int cc0 = (int)(o % CC);
      int64_t pix = o / CC;
      const int cx = (int)(pix % w); pix /= w;
      const int cy = (int)(pix % h);
      const int b = (int)(pix / h);
      auto src = [&](int cc) -> int64_t {
        int c = cc % C, r = cc / C, y = cy << levels, x = cx << levels;
        for (int l = 0; l < levels; ++l) { const int q = r & 3; r >>= 2; y += (q >> 1) << l; x += (q & 1) << l; }
        if (map) c = map[c];
        return (((int64_t)b * H + y) * W + x) * C + c;
      };
      if (vec) val = *reinterpret_cast<const f32x4*>(in + src(cc0));
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) val[j] = in[src(cc0 + j)];      // CC % 4 == 0: the four stay inside one coarse pixel
      }
    } else {
      // output = fine [B][H][W][C]
      auto src = [&](int64_t e) -> int64_t {
        int c = (int)(e % C);
        int64_t pix = e / C;
        const int x = (int)(pix % W); pix /= W;
        const int y = (int)(pix % H);
        const int b = (int)(pix / H);
        if (map) c = map[c];
        int cc = c, mul = C;
        for (int l = 0; l < levels; ++l) { cc += ((((y >> l) & 1) << 1) | ((x >> l) & 1)) * mul; mul <<= 2; }
        return (((int64_t)b * h + (y >> levels)) * w + (x >> levels)) * CC + cc;
      };
      if (vec) val = *reinterpret_cast<const f32x4*>(in + src(o));
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) val[j] = in[src(o + j)];
      }
    }
    *reinterpret_cast<f32x4*>(out + o) = val;
  }
}

int squeeze_rows_launch(const float* in, float* out, int B, int C, int H, int W, int levels, int inverse, const int* map,
                        hipStream_t st) {
  SININN_CHECK(in && out && B > 0 && C > 0 && H > 0 && W > 0 && levels >= 0 && levels <= 4, "squeeze_rows: bad arguments");
  SININN_CHECK((H % (1 << levels)) == 0 && (W % (1 << levels)) == 0, "squeeze_rows: H=%d W=%d not divisible by 2^%d", H, W, levels);
  const int64_t total = (int64_t)B * C * H * W;
  SININN_CHECK(total % 4 == 0 && aligned16(in) && aligned16(out), "squeeze_rows: needs a multiple of 4 elements and 16-byte aligned tensors");
  SININN_CHECK(levels > 0 || map != nullptr || true, "squeeze_rows");
  SININN_CHECK(((C << (2 * levels)) % 4) == 0, "squeeze_rows: coarse channel count must be a multiple of 4");
  const int64_t total4 = total / 4;
  const int64_t blocks = (total4 + 255) / 256 < 16384 ? (total4 + 255) / 256 : 16384;
  hipLaunchKernelGGL(squeeze_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, out, B, C, H, W, levels, inverse, map,
                     total4);
  SININN_LAUNCH_CHECK("squeeze_rows");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Frame-window sampler, dense pixel-major outputs (data.py:31-45): 16 clip bytes -> four float4 stores per thread for
// the HR frames, one uchar4 -> float4 per (pixel, LR frame) for the window.  x / 255.f is a true division (bit-exact
// with the reference's FloatTensor(...) / 255.).
// ------------------------------------------------------------------------------------------------
__global__ void sample_windows_dense_kernel(const uint8_t* __restrict__ hr_clip, const uint8_t* __restrict__ lr_clip,
                                            const int* __restrict__ idx, int n, int T, int64_t hr_frame16, int64_t lr_pix,
                                            int win, float* __restrict__ hr_out, float* __restrict__ lr_out,
                                            int64_t hr_items) {
  const int frames = 2 * win + 1;
  const int64_t lr_items = (int64_t)n * lr_pix * frames;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hr_items + lr_items;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (i < hr_items) {
      const int s = (int)(i / hr_frame16);
      const int64_t j = i - (int64_t)s * hr_frame16;
      int t = idx[s]; t = t < 0 ? 0 : (t >= T ? T - 1 : t);
      const uint4 raw = reinterpret_cast<const uint4*>(hr_clip)[(int64_t)t * hr_frame16 + j];
      const unsigned wds[4] = {raw.x, raw.y, raw.z, raw.w};
      f32x4* dst = reinterpret_cast<f32x4*>(hr_out) + i * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (float)((wds[q] >> (8 * k)) & 255u) / 255.f;
        dst[q] = v;
      }
    } else {
      const int64_t r = i - hr_items;
      const int f = (int)(r % frames);
      const int64_t sp = r / frames;                       // sample * lr_pix + pixel
      const int s = (int)(sp / lr_pix);
      const int64_t pix = sp - (int64_t)s * lr_pix;
      int t = idx[s] - win + f; t = t < 0 ? 0 : (t >= T ? T - 1 : t);
      const unsigned raw = reinterpret_cast<const unsigned*>(lr_clip)[(int64_t)t * lr_pix + pix];
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (float)((raw >> (8 * k)) & 255u) / 255.f;
      reinterpret_cast<f32x4*>(lr_out)[r] = v;            // out[s][pix][f*4 .. f*4+3]
    }
  }
}

bool sample_windows_dense_try(const uint8_t* hr_clip, const uint8_t* lr_clip, const int* idx, int n, int T, int H, int W, int h,
                              int w, int win, float* hr_out, const int64_t hs[4], float* lr_out, const int64_t ls[4],
                              hipStream_t st) {
  const int lrc = (2 * win + 1) * 4;
  const bool hr_dense = hs[1] == 1 && hs[3] == 3 && hs[2] == (int64_t)W * 3 && hs[0] == (int64_t)H * W * 3;
  const bool lr_dense = ls[1] == 1 && ls[3] == lrc && ls[2] == (int64_t)w * lrc && ls[0] == (int64_t)h * w * lrc;
  const int64_t frame_bytes = (int64_t)H * W * 3;
  if (!hr_dense || !lr_dense || frame_bytes % 16 != 0 || !aligned16(hr_clip) || !aligned16(hr_out) || !aligned16(lr_out) ||
      (reinterpret_cast<uintptr_t>(lr_clip) & 3u))
    return false;
  const int64_t hr_frame16 = frame_bytes / 16, lr_pix = (int64_t)h * w;
  const int64_t hr_items = (int64_t)n * hr_frame16;
  const int64_t items = hr_items + (int64_t)n * lr_pix * (2 * win + 1);
  const int64_t blocks = (items + 255) / 256 < 16384 ? (items + 255) / 256 : 16384;
  hipLaunchKernelGGL(sample_windows_dense_kernel, dim3((unsigned)blocks), dim3(256), 0, st, hr_clip, lr_clip, idx, n, T,
                     hr_frame16, lr_pix, win, hr_out, lr_out, hr_items);
  return true;
}

// Frame-pair sampler of the flow path (video-interpolation/: the trainer consumes (frame1, frame2) pairs of one clip as planar
// (B,3,H,W) tensors, trainer.py:49-62): out[s][c][y][x] = clip[idx[s] + offset][y][x][c] / 255 for BOTH frames of a pair
// (offset 0 -> out0, offset `gap` -> out1) straight from the resident uint8 clip, as fp32 or bf16 (BASELINE configs[3]).  One
// thread per 4 consecutive pixels of a row: 12 contiguous clip bytes in (three dwords), one 16- / 8-byte store per plane out.
template <typename T>
__global__ void sample_pairs_planar_kernel(const uint8_t* __restrict__ clip, const int* __restrict__ idx, int n, int Tn, int64_t HW,
                                           int gap, T* __restrict__ out0, T* __restrict__ out1) {
  const int64_t quads = HW / 4;                                             // host: H*W % 4 == 0
  const int64_t total = (int64_t)n * 2 * quads;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t qd = i % quads;
    const int which = (int)((i / quads) & 1);
    const int s = (int)(i / (2 * quads));
    int t = idx[s] + (which ? gap : 0);
    t = t < 0 ? 0 : (t >= Tn ? Tn - 1 : t);
    const unsigned* src = reinterpret_cast<const unsigned*>(clip + ((int64_t)t * HW + qd * 4) * 3);
    const unsigned w0 = src[0], w1 = src[1], w2 = src[2];
    unsigned char by[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) { by[k] = (w0 >> (8 * k)) & 255u; by[4 + k] = (w1 >> (8 * k)) & 255u; by[8 + k] = (w2 >> (8 * k)) & 255u; }
    T* dst = (which ? out1 : out0) + (int64_t)s * 3 * HW + qd * 4;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      T v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (T)((float)by[3 * k + c] / 255.f);
      if constexpr (sizeof(T) == 4) *reinterpret_cast<f32x4*>(dst + c * HW) = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
      else *reinterpret_cast<uint2*>(dst + c * HW) = *reinterpret_cast<const uint2*>(v);
    }
  }
}

int sample_pairs_planar_launch(const uint8_t* clip, const int* idx, int n, int T, int H, int W, int gap, void* out0, void* out1,
                               int bf16, hipStream_t st) {
  SININN_CHECK(clip && idx && out0 && out1 && n > 0 && T > 0 && H > 0 && W > 0, "sample_pairs: bad arguments");
  const int64_t HW = (int64_t)H * W;
  SININN_CHECK(HW % 4 == 0 && aligned16(clip) && aligned16(out0) && aligned16(out1), "sample_pairs: H*W %% 4 == 0 and 16-byte aligned buffers");
  const int64_t total = (int64_t)n * 2 * (HW / 4);
  const int64_t blocks = (total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384;
  if (bf16)
    hipLaunchKernelGGL(sample_pairs_planar_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, st, clip, idx, n, T, HW, gap,
                       static_cast<__bf16*>(out0), static_cast<__bf16*>(out1));
  else
    hipLaunchKernelGGL(sample_pairs_planar_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, clip, idx, n, T, HW, gap,
                       static_cast<float*>(out0), static_cast<float*>(out1));
  SININN_LAUNCH_CHECK("sample_pairs");
  return 0;
}

// out[m][c] = c < C ? in[m][c] : 0 for c < Cpad (C, Cpad multiples of 4): a channel slice into a padded buffer
__global__ void copy_channels_kernel(const float* __restrict__ in, int in_stride, float* __restrict__ out, int out_stride,
                                     int64_t M, int C4, int Cpad4) {
  const int64_t total = M * Cpad4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % Cpad4);
    const int64_t m = i / Cpad4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c4 < C4) v = *reinterpret_cast<const f32x4*>(in + m * in_stride + c4 * 4);
    *reinterpret_cast<f32x4*>(out + m * out_stride + c4 * 4) = v;
  }
}

int copy_channels_launch(const float* in, int in_stride, float* out, int out_stride, int64_t M, int C, int Cpad, hipStream_t st) {
  SININN_CHECK(in && out && M > 0 && C > 0 && C % 4 == 0 && Cpad % 4 == 0 && Cpad >= C, "copy_channels: bad arguments");
  SININN_CHECK(in_stride % 4 == 0 && out_stride % 4 == 0 && out_stride >= Cpad && aligned16(in) && aligned16(out), "copy_channels: alignment");
  const int64_t total = M * (Cpad / 4);
  const int64_t blocks = (total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192;
  hipLaunchKernelGGL(copy_channels_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, in_stride, out, out_stride, M, C / 4, Cpad / 4);
  SININN_LAUNCH_CHECK("copy_channels");
  return 0;
}

}  // namespace sininn

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

}  // namespace sininn

// Photometric-loss operators of the flow trainer, NCHW contiguous fp32 like the reference:
//   * softmax / summation splatting   video-interpolation/my_utils/softsplat.py:8-177   (forward scatter, both gradients)
//   * occlusion_wang correspondence   video-interpolation/my_utils/occlusions.py:29-104 (range map + threshold)
//   * CensusLoss                      video-interpolation/my_utils/loss.py:30-72        (ternary census, forward + gradients)
// All three are HBM / atomic bound: one thread per pixel computes the bilinear taps ONCE and walks the channels
// (coalesced along x), scatters use the hardware fp32 atomic add (no CAS loop), the census kernel stages the two
// grey-level tiles (+ halo) in LDS so every image byte is read once.
#include "common.h"

namespace sininn {

__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// ------------------------------------------------------------------------------------------------
// forward splat: out[b, c, y + fy .., x + fx ..] += in[b, c, y, x] * bilinear weight      (softsplat.py:8-52)
// ------------------------------------------------------------------------------------------------
struct SplatTaps {
  int nwx, nwy;
  float nw, ne, sw, se;
  bool vnw, vne, vsw, vse;
};

__device__ __forceinline__ SplatTaps splat_taps(float ox, float oy, int H, int W) {
  SplatTaps t;
  t.nwx = (int)floorf(ox); t.nwy = (int)floorf(oy);
  const float sex = (float)(t.nwx + 1), sey = (float)(t.nwy + 1), nwx = (float)t.nwx, nwy = (float)t.nwy;
  t.nw = (sex - ox) * (sey - oy);
  t.ne = (ox - nwx) * (sey - oy);
  t.sw = (sex - ox) * (oy - nwy);
  t.se = (ox - nwx) * (oy - nwy);
  const bool x0 = t.nwx >= 0 && t.nwx < W, x1 = t.nwx + 1 >= 0 && t.nwx + 1 < W;
  const bool y0 = t.nwy >= 0 && t.nwy < H, y1 = t.nwy + 1 >= 0 && t.nwy + 1 < H;
  t.vnw = x0 && y0; t.vne = x1 && y0; t.vsw = x0 && y1; t.vse = x1 && y1;
  return t;
}

// Block = 32 x 8 source pixels of one image.  Taps that land inside the block's window (tile + SP_R pixels all round)
// are accumulated with LDS atomics (ds_add_f32) and the window is flushed once with coalesced global atomics; only taps
// thrown further than SP_R pixels go to global memory directly.  For the smooth, few-pixel flows of the trainer that cuts
// the global atomic count ~3.5x and makes what remains row-contiguous.  ONES: splat a constant 1 (occlusion range map).
constexpr int SP_TX = 32, SP_TY = 8, SP_R = 8, SP_CC = 4;
constexpr int SP_WX = SP_TX + 2 * SP_R, SP_WY = SP_TY + 2 * SP_R;

template <bool ONES>
__global__ __launch_bounds__(SP_TX * SP_TY) void splat_fwd_kernel(const float* __restrict__ in, const float* __restrict__ flow,
                                                                  int B, int C, int H, int W, float* __restrict__ out) {
  __shared__ float win[SP_CC][SP_WY][SP_WX];
  const int tiles_x = (W + SP_TX - 1) / SP_TX, tiles_y = (H + SP_TY - 1) / SP_TY;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int x0 = tx * SP_TX, y0 = ty * SP_TY;
  const int lx = threadIdx.x % SP_TX, ly = threadIdx.x / SP_TX;
  const int x = x0 + lx, y = y0 + ly;
  const int64_t HW = (int64_t)H * W;
  const bool live = x < W && y < H;
  const int64_t r = live ? (int64_t)y * W + x : 0;
  SplatTaps t = {};
  bool any = false;
  if (live) {
    const float ox = (float)x + flow[((int64_t)b * 2 + 0) * HW + r], oy = (float)y + flow[((int64_t)b * 2 + 1) * HW + r];
    if (ox == ox && oy == oy && fabsf(ox) < 1e9f && fabsf(oy) < 1e9f) { t = splat_taps(ox, oy, H, W); any = true; }
  }
  // window coordinates of the north-west tap
  const int wx = t.nwx - (x0 - SP_R), wy = t.nwy - (y0 - SP_R);
  const bool in_nw = wx >= 0 && wx < SP_WX && wy >= 0 && wy < SP_WY;
  const bool in_ne = wx + 1 >= 0 && wx + 1 < SP_WX && wy >= 0 && wy < SP_WY;
  const bool in_sw = wx >= 0 && wx < SP_WX && wy + 1 >= 0 && wy + 1 < SP_WY;
  const bool in_se = wx + 1 >= 0 && wx + 1 < SP_WX && wy + 1 >= 0 && wy + 1 < SP_WY;
  const int64_t o_nw = (int64_t)t.nwy * W + t.nwx;
  // For a locally smooth flow the taps of neighbouring source pixels coincide (the south taps of a pixel are the north taps of
  // the pixel below it -- the wave's other row --, its east taps the west taps of its right neighbour): the contributions are
  // summed across lanes first, so an interior pixel issues one atomic per channel instead of four, all to different addresses
  // (as in flow_warp_l1_bwd_tiled_kernel).  Every shuffle is executed by all 64 lanes.
  static_assert(SP_TX == 32, "the lane merges assume 32-pixel tile rows");
  const int lane = threadIdx.x & 63;
  const int anyi = any ? 1 : 0;
  const int any_r = __shfl_down(anyi, 1), nwx_r = __shfl_down(t.nwx, 1), nwy_r = __shfl_down(t.nwy, 1);
  const int any_d = __shfl_down(anyi, 32), nwx_d = __shfl_down(t.nwx, 32), nwy_d = __shfl_down(t.nwy, 32);
  const bool give_h = any && (lane & 31) != 31 && any_r != 0 && nwx_r == t.nwx + 1 && nwy_r == t.nwy;
  const bool give_v = any && lane < 32 && any_d != 0 && nwx_d == t.nwx && nwy_d == t.nwy + 1;
  const int gh_l = __shfl_up(give_h ? 1 : 0, 1), gv_u = __shfl_up(give_v ? 1 : 0, 32);
  const bool take_h = (lane & 31) != 0 && gh_l != 0;
  const bool take_v = lane >= 32 && gv_u != 0;
  for (int c0 = 0; c0 < C; c0 += SP_CC) {
    const int cc = min(SP_CC, C - c0);
    for (int e = threadIdx.x; e < cc * SP_WY * SP_WX; e += blockDim.x) (&win[0][0][0])[e] = 0.f;
    __syncthreads();
    for (int c = 0; c < cc; ++c) {                      // every lane walks the channels: the merges are wave-wide
      float cnw = 0.f, cne = 0.f, csw = 0.f, cse = 0.f;
      if (any) {
        const float v = ONES ? 1.f : in[((int64_t)b * C + c0 + c) * HW + r];
        cnw = t.vnw ? v * t.nw : 0.f; cne = t.vne ? v * t.ne : 0.f; csw = t.vsw ? v * t.sw : 0.f; cse = t.vse ? v * t.se : 0.f;
      }
      const float rsw = __shfl_up(csw, 32), rse = __shfl_up(cse, 32);
      if (take_v) { cnw += rsw; cne += rse; }
      if (give_v) { csw = 0.f; cse = 0.f; }
      const float sne = __shfl_up(cne, 1), sse = __shfl_up(cse, 1);
      if (take_h) { cnw += sne; csw += sse; }
      if (give_h) { cne = 0.f; cse = 0.f; }
      if (any) {
        float* o = out + ((int64_t)b * C + c0 + c) * HW;
        // (a merged value lands where the receiving lane's own tap of that slot lands: same address, same validity)
        if (t.vnw && cnw != 0.f) { if (in_nw) atomic_add_f32(&win[c][wy][wx], cnw); else atomic_add_f32(o + o_nw, cnw); }
        if (t.vne && cne != 0.f) { if (in_ne) atomic_add_f32(&win[c][wy][wx + 1], cne); else atomic_add_f32(o + o_nw + 1, cne); }
        if (t.vsw && csw != 0.f) { if (in_sw) atomic_add_f32(&win[c][wy + 1][wx], csw); else atomic_add_f32(o + o_nw + W, csw); }
        if (t.vse && cse != 0.f) { if (in_se) atomic_add_f32(&win[c][wy + 1][wx + 1], cse); else atomic_add_f32(o + o_nw + W + 1, cse); }
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < cc * SP_WY * SP_WX; e += blockDim.x) {
      const float v = (&win[0][0][0])[e];
      if (v != 0.f) {
        const int c = e / (SP_WY * SP_WX), q = e % (SP_WY * SP_WX);
        const int gy = y0 - SP_R + q / SP_WX, gx = x0 - SP_R + q % SP_WX;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) atomic_add_f32(out + ((int64_t)b * C + c0 + c) * HW + (int64_t)gy * W + gx, v);
      }
    }
    __syncthreads();
  }
}

// both gradients in one pass over gradOutput's four taps (softsplat.py:54-177):
//   gin[b,c,y,x]  = sum_taps gout[b,c,tap] * w_tap
//   gflow[b,0,y,x] = sum_c in[b,c,y,x] * sum_taps gout[b,c,tap] * dw_tap/dx     (likewise component 1 with d/dy)
__global__ void softsplat_bwd_kernel(const float* __restrict__ in, const float* __restrict__ flow, const float* __restrict__ gout,
                                     int B, int C, int H, int W, float* __restrict__ gin, float* __restrict__ gflow) {
  const int64_t HW = (int64_t)H * W, total = (int64_t)B * HW;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int b = (int)(idx / HW);
  const int64_t r = idx % HW;
  const int y = (int)(r / W), x = (int)(r % W);
  const float ox = (float)x + flow[(b * 2 + 0) * HW + r], oy = (float)y + flow[(b * 2 + 1) * HW + r];
  const bool nan = !(ox == ox) || !(oy == oy);
  SplatTaps t = splat_taps(nan ? 0.f : ox, nan ? 0.f : oy, H, W);
  if (nan) t.vnw = t.vne = t.vsw = t.vse = false;
  const float fx = ox - (float)t.nwx, fy = oy - (float)t.nwy;   // fractional parts
  // d/dx of (nw, ne, sw, se) = (-(1-fy), (1-fy), -fy, fy);  d/dy = (-(1-fx), -fx, (1-fx), fx)
  const int64_t o_nw = (int64_t)t.nwy * W + t.nwx;
  float gfx = 0.f, gfy = 0.f;
  for (int c = 0; c < C; ++c) {
    const float* g = gout + ((int64_t)b * C + c) * HW;
    const float gnw = t.vnw ? g[o_nw] : 0.f, gne = t.vne ? g[o_nw + 1] : 0.f;
    const float gsw = t.vsw ? g[o_nw + W] : 0.f, gse = t.vse ? g[o_nw + W + 1] : 0.f;
    if (gin) gin[((int64_t)b * C + c) * HW + r] = gnw * t.nw + gne * t.ne + gsw * t.sw + gse * t.se;
    if (gflow) {
      const float v = in[((int64_t)b * C + c) * HW + r];
      gfx += v * ((gne - gnw) * (1.f - fy) + (gse - gsw) * fy);
      gfy += v * ((gsw - gnw) * (1.f - fx) + (gse - gne) * fx);
    }
  }
  if (gflow) { gflow[(b * 2 + 0) * HW + r] = gfx; gflow[(b * 2 + 1) * HW + r] = gfy; }
}

int softsplat_fwd_launch(const float* in, const float* flow, int B, int C, int H, int W, float* out, hipStream_t st) {
  SININN_CHECK(in && flow && out, "softsplat: null pointer");
  SININN_CHECK(B > 0 && C > 0 && H > 0 && W > 0, "softsplat: bad shape");
  const int tiles = B * ((H + SP_TY - 1) / SP_TY) * ((W + SP_TX - 1) / SP_TX);
  hipLaunchKernelGGL(splat_fwd_kernel<false>, dim3(tiles), dim3(SP_TX * SP_TY), 0, st, in, flow, B, C, H, W, out);
  SININN_LAUNCH_CHECK("softsplat_fwd");
  return 0;
}

int softsplat_bwd_launch(const float* in, const float* flow, const float* gout, int B, int C, int H, int W, float* gin,
                         float* gflow, hipStream_t st) {
  SININN_CHECK(in && flow && gout && (gin || gflow), "softsplat_bwd: null pointer");
  SININN_CHECK(B > 0 && C > 0 && H > 0 && W > 0, "softsplat_bwd: bad shape");
  const int64_t total = (int64_t)B * H * W;
  hipLaunchKernelGGL(softsplat_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, flow, gout, B, C, H, W,
                     gin, gflow);
  SININN_LAUNCH_CHECK("softsplat_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// occlusion_wang (occlusions.py:29-104): range map of the backward flow, mask = NOT (map <= thresh)
//   every pixel scatters its four CLAMPED corner weights and a corner that had to be clamped contributes 0 -- i.e. taps
//   outside the image are dropped and the rest carry the bilinear weights: the summation splat of a constant 1
// ------------------------------------------------------------------------------------------------
__global__ void occlusion_mask_kernel(const float* __restrict__ corr, int64_t n, float thresh, float* __restrict__ mask) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) mask[i] = (corr[i] <= thresh) ? 0.f : 1.f;
}

int occlusion_wang_launch(const float* flow21, int B, int H, int W, float thresh, float* corr_zeroed, float* mask,
                          hipStream_t st) {
  SININN_CHECK(flow21 && corr_zeroed, "occlusion_wang: null pointer");
  SININN_CHECK(B > 0 && H > 0 && W > 0, "occlusion_wang: bad shape");
  const int64_t total = (int64_t)B * H * W;
  const int tiles = B * ((H + SP_TY - 1) / SP_TY) * ((W + SP_TX - 1) / SP_TX);
  hipLaunchKernelGGL(splat_fwd_kernel<true>, dim3(tiles), dim3(SP_TX * SP_TY), 0, st, nullptr, flow21, B, 1, H, W, corr_zeroed);
  SININN_LAUNCH_CHECK("corr_map");
  if (mask) {
    hipLaunchKernelGGL(occlusion_mask_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, corr_zeroed, total, thresh,
                       mask);
    SININN_LAUNCH_CHECK("occlusion_mask");
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// CensusLoss (loss.py:30-72):  I = 255 * grey(img * mask);  t(p, o) = (I[p+o] - I[p]) / sqrt(0.81 + (.)^2)  (zero padding);
//   d(p) = mean_o ((t1 - t2)^2 / (0.1 + (t1 - t2)^2));  loss = weight * mean(d * inner) * numel(mask) / sum(mask)
// forward: acc[0] += sum_p d(p) * inner(p);  acc[1] += sum(mask)           (the host/device epilogue forms the scalar)
// backward: g1 / g2 = d loss / d (img1, img2), 3 channels each, through the grey weights and the mask
// ------------------------------------------------------------------------------------------------
constexpr int CT = 16;                 // pixel tile
constexpr int CMAXD = 4;               // max_distance <= 4  (reference default 2, trainer 3)
constexpr int CENSUS_SLOTS = 64;       // acc = {sum d, sum mask, 64 x {partial d, partial mask}} = SININN_CENSUS_ACC_FLOATS

// MC = mask channels: 1 (pair_flow.py: occlusion mask) or 3 (trainer.py:64: occlusion mask * (softmax != 0))
__device__ __forceinline__ float mask_at(const float* mask, int MC, int b, int c, int64_t HW, int64_t r) {
  return mask[((int64_t)b * MC + (MC == 1 ? 0 : c)) * HW + r];
}
__device__ __forceinline__ float grey255(const float* img, const float* mask, int MC, int b, int64_t HW, int64_t r) {
  return 255.f * (img[((int64_t)b * 3 + 0) * HW + r] * mask_at(mask, MC, b, 0, HW, r) * 0.2989f +
                  img[((int64_t)b * 3 + 1) * HW + r] * mask_at(mask, MC, b, 1, HW, r) * 0.5870f +
                  img[((int64_t)b * 3 + 2) * HW + r] * mask_at(mask, MC, b, 2, HW, r) * 0.1140f);
}

// LDS tiles of both grey images with a halo of `halo` pixels (zeros outside the image)
__device__ __forceinline__ void census_stage(const float* im1, const float* im2, const float* mask, int MC, int b, int H, int W, int y0,
                                             int x0, int halo, float* s1, float* s2) {
  const int TW = CT + 2 * halo;
  const int64_t HW = (int64_t)H * W;
  for (int e = threadIdx.x; e < TW * TW; e += blockDim.x) {
    const int ly = e / TW, lx = e % TW;
    const int gy = y0 + ly - halo, gx = x0 + lx - halo;
    float a = 0.f, c = 0.f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
      const int64_t r = (int64_t)gy * W + gx;
      a = grey255(im1, mask, MC, b, HW, r);
      c = grey255(im2, mask, MC, b, HW, r);
    }
    s1[e] = a; s2[e] = c;
  }
  __syncthreads();
}

__global__ __launch_bounds__(CT * CT) void census_fwd_kernel(const float* __restrict__ im1, const float* __restrict__ im2,
                                                             const float* __restrict__ mask, int MC, int B, int H, int W,
                                                             int md, float* __restrict__ acc) {
  extern __shared__ float sm[];
  const int TW = CT + 2 * md;
  float* s1 = sm; float* s2 = sm + TW * TW;
  const int tiles_x = (W + CT - 1) / CT, tiles_y = (H + CT - 1) / CT;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y0 = ty * CT, x0 = tx * CT;
  census_stage(im1, im2, mask, MC, b, H, W, y0, x0, md, s1, s2);
  const int ly = threadIdx.x / CT, lx = threadIdx.x % CT;
  const int gy = y0 + ly, gx = x0 + lx;
  float d = 0.f, msum = 0.f;
  if (gy < H && gx < W) {
    for (int c = 0; c < MC; ++c) msum += mask[((int64_t)b * MC + c) * H * W + (int64_t)gy * W + gx];
    const bool inner = gy >= md && gy < H - md && gx >= md && gx < W - md;
    if (inner) {
      const float c1 = s1[(ly + md) * TW + lx + md], c2 = s2[(ly + md) * TW + lx + md];
      float sum = 0.f;
      for (int oy = 0; oy <= 2 * md; ++oy)
        for (int ox = 0; ox <= 2 * md; ++ox) {
          const float a = s1[(ly + oy) * TW + lx + ox] - c1, c = s2[(ly + oy) * TW + lx + ox] - c2;
          const float t1 = a * rsqrtf(0.81f + a * a), t2 = c * rsqrtf(0.81f + c * c);
          const float q = (t1 - t2) * (t1 - t2);
          sum += q * __frcp_rn(0.1f + q);
        }
      const int P = 2 * md + 1;
      d = sum / (float)(P * P);
    }
  }
  d = wave_sum(d); msum = wave_sum(msum);
  __shared__ float red[2][CT * CT / 64];
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = d; red[1][threadIdx.x >> 6] = msum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, m = 0.f;
    for (int i = 0; i < CT * CT / 64; ++i) { a += red[0][i]; m += red[1][i]; }
    // 64 slots instead of one address: thousands of same-address atomics serialise in L2 (~80 us at 512x512, bs 4)
    float* slot = acc + 2 + 2 * (blockIdx.x & (CENSUS_SLOTS - 1));
    atomic_add_f32(slot + 0, a);
    atomic_add_f32(slot + 1, m);
  }
}

// gradient w.r.t. the grey levels, gathered per pixel q:  dI[q] = sum_o k(q - o, o) - sum_o k(q, o),
//   k(p, o) = coef * inner(p) * dD/dq(t1 - t2) * dt/da   with a = I[p+o] - I[p]
// then d img[c] = dI * 255 * grey_c * mask.   coef = gscale * weight / (P^2 * sum(mask))
__global__ __launch_bounds__(CT * CT) void census_bwd_kernel(const float* __restrict__ im1, const float* __restrict__ im2,
                                                             const float* __restrict__ mask, int MC, int B, int H, int W,
                                                             int md, const float* __restrict__ acc, const float* __restrict__ gscale,
                                                             float weight, float* __restrict__ g1, float* __restrict__ g2) {
  extern __shared__ float sm[];
  const int halo = md;                   // k(q - o, o) only touches I[q - o] and I[q]
  const int TW = CT + 2 * halo;
  float* s1 = sm; float* s2 = sm + TW * TW;
  const int tiles_x = (W + CT - 1) / CT, tiles_y = (H + CT - 1) / CT;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y0 = ty * CT, x0 = tx * CT;
  census_stage(im1, im2, mask, MC, b, H, W, y0, x0, halo, s1, s2);
  const int ly = threadIdx.x / CT, lx = threadIdx.x % CT;
  const int gy = y0 + ly, gx = x0 + lx;
  if (gy >= H || gx >= W) return;
  const int P = 2 * md + 1;
  const float coef = (gscale ? gscale[0] : 1.f) * weight * (float)MC / ((float)(P * P) * acc[1]);
  auto kterm = [&](int py, int px, int oy, int ox, float& k1, float& k2) {   // tile coords of centre p, offset o
    const float a = s1[(py + oy) * TW + px + ox] - s1[py * TW + px], c = s2[(py + oy) * TW + px + ox] - s2[py * TW + px];
    const float ra = rsqrtf(0.81f + a * a), rc = rsqrtf(0.81f + c * c);
    const float t1 = a * ra, t2 = c * rc, df = t1 - t2, q = df * df;
    const float rq = __frcp_rn(0.1f + q);
    const float dq = 0.2f * rq * rq * df;                                    // d (q / (0.1 + q)) / d(t1 - t2)
    k1 = dq * 0.81f * ra * ra * ra;                                          // dt1/da = 0.81 / (0.81 + a^2)^1.5
    k2 = -dq * 0.81f * rc * rc * rc;
  };
  float d1 = 0.f, d2 = 0.f;
  const int cy = ly + halo, cx = lx + halo;                                  // this pixel in tile coordinates
  for (int oy = -md; oy <= md; ++oy)
    for (int ox = -md; ox <= md; ++ox) {
      float k1, k2;
      // as the centre p = q: -k(q, o)
      if (gy >= md && gy < H - md && gx >= md && gx < W - md) {
        kterm(cy, cx, oy, ox, k1, k2);
        d1 -= k1; d2 -= k2;
      }
      // as the neighbour of p = q - o: +k(q - o, o)
      const int py = gy - oy, px = gx - ox;
      if (py >= md && py < H - md && px >= md && px < W - md) {
        kterm(cy - oy, cx - ox, oy, ox, k1, k2);
        d1 += k1; d2 += k2;
      }
    }
  const int64_t HW = (int64_t)H * W, r = (int64_t)gy * W + gx;
  const float gw[3] = {0.2989f, 0.5870f, 0.1140f};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float m = mask_at(mask, MC, b, c, HW, r) * 255.f * coef;
    if (g1) g1[((int64_t)b * 3 + c) * HW + r] = d1 * m * gw[c];
    if (g2) g2[((int64_t)b * 3 + c) * HW + r] = d2 * m * gw[c];
  }
}

// shared by the census and masked-L1 losses: scale = weight * numel(mask) / numel(mean)
__global__ void census_finish_kernel(float* __restrict__ acc, float weight, float* __restrict__ out) {
  // mean(d * inner) / sum(mask) * numel(mask) * weight  ==  (weight * MC) * sum(d * inner) / sum(mask)
  float a = acc[2 + 2 * threadIdx.x], m = acc[3 + 2 * threadIdx.x];
  a = wave_sum(a); m = wave_sum(m);
  if (threadIdx.x == 0) { acc[0] = a; acc[1] = m; out[0] = weight * a / m; }
}

int census_fwd_launch(const float* im1, const float* im2, const float* mask, int mask_channels, int B, int H, int W,
                      int max_distance, float weight, float* acc_zeroed, float* out, hipStream_t st) {
  const int MC = mask_channels;
  SININN_CHECK(MC == 1 || MC == 3, "census: mask must have 1 or 3 channels");
  SININN_CHECK(im1 && im2 && mask && acc_zeroed && out, "census: null pointer");
  SININN_CHECK(B > 0 && H > 0 && W > 0 && max_distance >= 1 && max_distance <= CMAXD, "census: bad shape / max_distance in 1..%d", CMAXD);
  const int tiles = B * ((H + CT - 1) / CT) * ((W + CT - 1) / CT);
  const int TW = CT + 2 * max_distance;
  hipLaunchKernelGGL(census_fwd_kernel, dim3(tiles), dim3(CT * CT), 2 * TW * TW * sizeof(float), st, im1, im2, mask, MC, B,
                     H, W, max_distance, acc_zeroed);
  SININN_LAUNCH_CHECK("census_fwd");
  static_assert(CENSUS_SLOTS == 64, "one wave sums the slots");
  hipLaunchKernelGGL(census_finish_kernel, dim3(1), dim3(CENSUS_SLOTS), 0, st, acc_zeroed, weight * (float)MC, out);
  SININN_LAUNCH_CHECK("census_finish");
  return 0;
}

int census_bwd_launch(const float* im1, const float* im2, const float* mask, int mask_channels, int B, int H, int W,
                      int max_distance, float weight, const float* acc, const float* gscale, float* g1, float* g2,
                      hipStream_t st) {
  const int MC = mask_channels;
  SININN_CHECK(MC == 1 || MC == 3, "census_bwd: mask must have 1 or 3 channels");
  SININN_CHECK(im1 && im2 && mask && acc && (g1 || g2), "census_bwd: null pointer");
  SININN_CHECK(B > 0 && H > 0 && W > 0 && max_distance >= 1 && max_distance <= CMAXD, "census_bwd: bad shape / max_distance in 1..%d", CMAXD);
  const int tiles = B * ((H + CT - 1) / CT) * ((W + CT - 1) / CT);
  const int TW = CT + 2 * max_distance;
  hipLaunchKernelGGL(census_bwd_kernel, dim3(tiles), dim3(CT * CT), 2 * TW * TW * sizeof(float), st, im1, im2, mask, MC, B,
                     H, W, max_distance, acc, gscale, weight, g1, g2);
  SININN_LAUNCH_CHECK("census_bwd");
  return 0;
}

// one pair of slot atomics per BLOCK (same-address atomics serialise in L2: per-wave atomics cost ~100 us at 512x512)
__device__ __forceinline__ void block_slot_add(float a, float b, float* acc) {
  __shared__ float red[2][16];
  a = wave_sum(a); b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float x = 0.f, y = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { x += red[0][i]; y += red[1][i]; }
    float* slot = acc + 2 + 2 * (blockIdx.x & (CENSUS_SLOTS - 1));
    atomic_add_f32(slot + 0, x);
    atomic_add_f32(slot + 1, y);
  }
}

// ------------------------------------------------------------------------------------------------
// L1Loss (loss.py:17-27): l1_loss(im1 * mask, im2 * mask) / sum(mask) * numel(mask) * weight; mask has 1 or C channels
//   acc slots as in the census loss: {sum |im1*m - im2*m|, sum(mask)}
// ------------------------------------------------------------------------------------------------
__global__ void masked_l1_fwd_kernel(const float* __restrict__ a, const float* __restrict__ bb, const float* __restrict__ mask,
                                     int MC, int B, int C, int H, int W, float* __restrict__ acc) {
  const int64_t HW = (int64_t)H * W, total = (int64_t)B * HW;
  float s = 0.f, ms = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(idx / HW);
    const int64_t r = idx % HW;
    for (int c = 0; c < C; ++c) {
      const float m = mask[((int64_t)b * MC + (MC == 1 ? 0 : c)) * HW + r];
      s += fabsf(a[((int64_t)b * C + c) * HW + r] * m - bb[((int64_t)b * C + c) * HW + r] * m);
      if (MC != 1 || c == 0) ms += m;
    }
  }
  block_slot_add(s, ms, acc);
}

__global__ void masked_l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ bb, const float* __restrict__ mask,
                                     int MC, int B, int C, int H, int W, const float* __restrict__ acc,
                                     const float* __restrict__ gscale, float scale, float* __restrict__ g1,
                                     float* __restrict__ g2) {
  const int64_t HW = (int64_t)H * W, total = (int64_t)B * C * HW;
  const float coef = (gscale ? gscale[0] : 1.f) * scale / acc[1];
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx % HW;
    const int c = (int)((idx / HW) % C), b = (int)(idx / (HW * C));
    const float m = mask[((int64_t)b * MC + (MC == 1 ? 0 : c)) * HW + r];
    const float d = a[idx] * m - bb[idx] * m;
    const float g = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * m * coef;
    if (g1) g1[idx] = g;
    if (g2) g2[idx] = -g;
  }
}

int masked_l1_fwd_launch(const float* im1, const float* im2, const float* mask, int mask_channels, int B, int C, int H, int W,
                         float weight, float* acc_zeroed, float* out, hipStream_t st) {
  SININN_CHECK(im1 && im2 && mask && acc_zeroed && out, "masked_l1: null pointer");
  SININN_CHECK(B > 0 && C > 0 && H > 0 && W > 0 && (mask_channels == 1 || mask_channels == C), "masked_l1: bad shape");
  const int64_t total = (int64_t)B * H * W;
  const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(masked_l1_fwd_kernel, dim3(blocks), dim3(256), 0, st, im1, im2, mask, mask_channels, B, C, H, W, acc_zeroed);
  SININN_LAUNCH_CHECK("masked_l1");
  // mean over B*C*H*W, times numel(mask) = B*MC*H*W, over sum(mask):  weight * MC / C * sum / sum(mask)
  hipLaunchKernelGGL(census_finish_kernel, dim3(1), dim3(CENSUS_SLOTS), 0, st, acc_zeroed,
                     weight * (float)mask_channels / (float)C, out);
  SININN_LAUNCH_CHECK("masked_l1_finish");
  return 0;
}

int masked_l1_bwd_launch(const float* im1, const float* im2, const float* mask, int mask_channels, int B, int C, int H, int W,
                         float weight, const float* acc, const float* gscale, float* g1, float* g2, hipStream_t st) {
  SININN_CHECK(im1 && im2 && mask && acc && (g1 || g2), "masked_l1_bwd: null pointer");
  SININN_CHECK(B > 0 && C > 0 && H > 0 && W > 0 && (mask_channels == 1 || mask_channels == C), "masked_l1_bwd: bad shape");
  const int64_t total = (int64_t)B * C * H * W;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(masked_l1_bwd_kernel, dim3(blocks), dim3(256), 0, st, im1, im2, mask, mask_channels, B, C, H, W, acc,
                     gscale, weight * (float)mask_channels / (float)C, g1, g2);
  SININN_LAUNCH_CHECK("masked_l1_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// BilateralSmooth (loss.py:106-132): edge-aware 1st / 2nd order smoothness of a flow field
//   order s: img differences with stride s weight the s-th finite difference of the flow (NOTE image_grads returns
//   (d/dH, d/dW), which the reference names (gx, gy));  w = exp(-mean_c f(k * dimg)), f = |.| ('exp') or (.)^2 ('gauss');
//   loss = weight * (mean(w_h * robust(dflow_h)) + mean(w_w * robust(dflow_w))) / 2,  robust(u) = sqrt(u^2 + 1e-6)
// acc: {sum_h, sum_w} partial slots like the other losses (SININN_CENSUS_ACC_FLOATS floats)
// ------------------------------------------------------------------------------------------------
struct SmoothDev { const float* img; const float* flow; int B, C, H, W, order, gauss; float k; };

// term along direction dir (0: H, 1: W) anchored at (y, x): weight w and the flow difference u of component kf
__device__ __forceinline__ bool smooth_anchor_ok(const SmoothDev& p, int dir, int y, int x) {
  return y >= 0 && x >= 0 && (dir == 0 ? (y < p.H - p.order && x < p.W) : (y < p.H && x < p.W - p.order));
}
__device__ __forceinline__ float smooth_w(const SmoothDev& p, int b, int dir, int y, int x) {
  const int64_t HW = (int64_t)p.H * p.W;
  const int64_t r0 = (int64_t)y * p.W + x, r1 = r0 + (dir == 0 ? (int64_t)p.order * p.W : p.order);
  float m = 0.f;
  for (int c = 0; c < p.C; ++c) {
    const float g = p.k * (p.img[((int64_t)b * p.C + c) * HW + r1] - p.img[((int64_t)b * p.C + c) * HW + r0]);
    m += p.gauss ? g * g : fabsf(g);
  }
  return expf(-m / (float)p.C);
}
__device__ __forceinline__ float smooth_u(const SmoothDev& p, int b, int kf, int dir, int y, int x) {
  const int64_t HW = (int64_t)p.H * p.W;
  const float* f = p.flow + ((int64_t)b * 2 + kf) * HW;
  const int64_t r0 = (int64_t)y * p.W + x, st = dir == 0 ? p.W : 1;
  return p.order == 1 ? f[r0 + st] - f[r0] : (f[r0 + 2 * st] - f[r0 + st]) - (f[r0 + st] - f[r0]);
}

__global__ void smooth_fwd_kernel(SmoothDev p, float* __restrict__ acc) {
  const int64_t HW = (int64_t)p.H * p.W, total = (int64_t)p.B * HW;
  float sh = 0.f, sw = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(idx / HW);
    const int y = (int)((idx % HW) / p.W), x = (int)(idx % p.W);
    for (int dir = 0; dir < 2; ++dir)
      if (smooth_anchor_ok(p, dir, y, x)) {
        const float w = smooth_w(p, b, dir, y, x);
        float t = 0.f;
        for (int kf = 0; kf < 2; ++kf) { const float u = smooth_u(p, b, kf, dir, y, x); t += sqrtf(u * u + 1e-6f); }
        (dir == 0 ? sh : sw) += w * t;
      }
  }
  block_slot_add(sh, sw, acc);
}

__global__ void smooth_finish_kernel(float* __restrict__ acc, float ch, float cw, float* __restrict__ out) {
  float a = acc[2 + 2 * threadIdx.x], m = acc[3 + 2 * threadIdx.x];
  a = wave_sum(a); m = wave_sum(m);
  if (threadIdx.x == 0) { acc[0] = a; acc[1] = m; out[0] = a * ch + m * cw; }
}

// d loss / d flow, gathered per pixel: every anchor whose stencil touches (y, x) contributes coef * w * robust'(u) * tap
__global__ void smooth_bwd_kernel(SmoothDev p, const float* __restrict__ gscale, float ch, float cw, float* __restrict__ gflow) {
  const int64_t HW = (int64_t)p.H * p.W, total = (int64_t)p.B * HW;
  const float gs = gscale ? gscale[0] : 1.f;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(idx / HW);
    const int y = (int)((idx % HW) / p.W), x = (int)(idx % p.W);
    float g[2] = {0.f, 0.f};
    for (int dir = 0; dir < 2; ++dir) {
      const float coef = gs * (dir == 0 ? ch : cw);
      for (int j = 0; j <= p.order; ++j) {               // this pixel is stencil point j of the anchor j steps back
        const int ay = y - (dir == 0 ? j : 0), ax = x - (dir == 1 ? j : 0);
        if (!smooth_anchor_ok(p, dir, ay, ax)) continue;
        const float tap = p.order == 1 ? (j == 0 ? -1.f : 1.f) : (j == 1 ? -2.f : 1.f);
        const float w = smooth_w(p, b, dir, ay, ax);
        for (int kf = 0; kf < 2; ++kf) {
          const float u = smooth_u(p, b, kf, dir, ay, ax);
          g[kf] += coef * w * u * rsqrtf(u * u + 1e-6f) * tap;
        }
      }
    }
    gflow[((int64_t)b * 2 + 0) * HW + (idx % HW)] = g[0];
    gflow[((int64_t)b * 2 + 1) * HW + (idx % HW)] = g[1];
  }
}

static int smooth_setup(SmoothDev& p, const float* img, const float* flow, int B, int C, int H, int W, int order, int gauss,
                        float k, float weight, float& ch, float& cw) {
  SININN_CHECK(img && flow, "smooth: null pointer");
  SININN_CHECK(B > 0 && C > 0 && (order == 1 || order == 2) && H > order && W > order, "smooth: bad shape / order");
  p = SmoothDev{img, flow, B, C, H, W, order, gauss, k};
  ch = 0.5f * weight / ((float)B * 2.f * (float)(H - order) * (float)W);
  cw = 0.5f * weight / ((float)B * 2.f * (float)H * (float)(W - order));
  return 0;
}

int smooth_fwd_launch(const float* img, const float* flow, int B, int C, int H, int W, int order, int gauss, float k,
                      float weight, float* acc_zeroed, float* out, hipStream_t st) {
  SmoothDev p; float ch, cw;
  if (int rc = smooth_setup(p, img, flow, B, C, H, W, order, gauss, k, weight, ch, cw)) return rc;
  SININN_CHECK(acc_zeroed && out, "smooth: null pointer");
  const int64_t total = (int64_t)B * H * W;
  const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(smooth_fwd_kernel, dim3(blocks), dim3(256), 0, st, p, acc_zeroed);
  SININN_LAUNCH_CHECK("smooth_fwd");
  hipLaunchKernelGGL(smooth_finish_kernel, dim3(1), dim3(CENSUS_SLOTS), 0, st, acc_zeroed, ch, cw, out);
  SININN_LAUNCH_CHECK("smooth_finish");
  return 0;
}

int smooth_bwd_launch(const float* img, const float* flow, int B, int C, int H, int W, int order, int gauss, float k,
                      float weight, const float* gscale, float* gflow, hipStream_t st) {
  SmoothDev p; float ch, cw;
  if (int rc = smooth_setup(p, img, flow, B, C, H, W, order, gauss, k, weight, ch, cw)) return rc;
  SININN_CHECK(gflow, "smooth_bwd: null pointer");
  const int64_t total = (int64_t)B * H * W;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(smooth_bwd_kernel, dim3(blocks), dim3(256), 0, st, p, gscale, ch, cw, gflow);
  SININN_LAUNCH_CHECK("smooth_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// SSIMLoss (loss.py:75-103): x = im1 * mask, y = im2 * mask; (2 md + 1)^2 average pooling WITHOUT padding -> (H - 2 md) x
// (W - 2 md) windows; dist = clamp((1 - SSIM) / 2, 0, 1); loss = weight * mean(dist) * numel(mask) / sum(mask).
// One block = 16 x 16 windows (forward) / input pixels (backward) of one (image, channel); the masked tiles live in LDS.
// ------------------------------------------------------------------------------------------------
constexpr int SS_T = 16;
constexpr float SS_C1 = 0.01f * 0.01f, SS_C2 = 0.03f * 0.03f;

__device__ __forceinline__ void ssim_stage(const float* a, const float* bb, const float* mask, int MC, int C, int b, int c, int H,
                                           int W, int y0, int x0, int TW, float* sx, float* sy) {
  const int64_t HW = (int64_t)H * W;
  for (int e = threadIdx.x; e < TW * TW; e += blockDim.x) {
    const int gy = y0 + e / TW, gx = x0 + e % TW;
    float vx = 0.f, vy = 0.f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
      const int64_t r = (int64_t)gy * W + gx;
      const float m = mask[((int64_t)b * MC + (MC == 1 ? 0 : c)) * HW + r];
      vx = a[((int64_t)b * C + c) * HW + r] * m;
      vy = bb[((int64_t)b * C + c) * HW + r] * m;
    }
    sx[e] = vx; sy[e] = vy;
  }
  __syncthreads();
}

// SSIM of the window whose top-left corner is tile position (wy, wx); also the derivative pieces when D != nullptr:
//   D = {dS/dex, dS/dey, dS/dexx (== dS/deyy), dS/dexy}
__device__ __forceinline__ float ssim_window(const float* sx, const float* sy, int TW, int wy, int wx, int P, float* D) {
  float ex = 0.f, ey = 0.f, exx = 0.f, eyy = 0.f, exy = 0.f;
  for (int j = 0; j < P; ++j)
    for (int i = 0; i < P; ++i) {
      const float x = sx[(wy + j) * TW + wx + i], y = sy[(wy + j) * TW + wx + i];
      ex += x; ey += y; exx += x * x; eyy += y * y; exy += x * y;
    }
  const float inv = 1.f / (float)(P * P);
  ex *= inv; ey *= inv; exx *= inv; eyy *= inv; exy *= inv;
  const float n1 = 2.f * ex * ey + SS_C1, n2 = 2.f * (exy - ex * ey) + SS_C2;
  const float d1 = ex * ex + ey * ey + SS_C1, d2 = (exx - ex * ex) + (eyy - ey * ey) + SS_C2;
  const float S = n1 * n2 / (d1 * d2);
  if (D) {
    const float r = 1.f / (d1 * d2);
    D[0] = (2.f * ey * n2 - 2.f * ey * n1) * r - S * (2.f * ex / d1 - 2.f * ex / d2);
    D[1] = (2.f * ex * n2 - 2.f * ex * n1) * r - S * (2.f * ey / d1 - 2.f * ey / d2);
    D[2] = -S / d2;
    D[3] = 2.f * n1 * r;
  }
  return S;
}

__global__ __launch_bounds__(SS_T * SS_T) void ssim_fwd_kernel(const float* __restrict__ a, const float* __restrict__ bb,
                                                               const float* __restrict__ mask, int MC, int B, int C, int H, int W,
                                                               int md, float* __restrict__ acc) {
  extern __shared__ float sm[];
  const int P = 2 * md + 1, TW = SS_T + 2 * md, Ho = H - 2 * md, Wo = W - 2 * md;
  float* sx = sm; float* sy = sm + TW * TW;
  const int tiles_x = (Wo + SS_T - 1) / SS_T, tiles_y = (Ho + SS_T - 1) / SS_T;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y; bid /= tiles_y;
  const int c = bid % C, b = bid / C;
  const int y0 = ty * SS_T, x0 = tx * SS_T;               // window (== input top-left) coordinates
  ssim_stage(a, bb, mask, MC, C, b, c, H, W, y0, x0, TW, sx, sy);
  const int ly = threadIdx.x / SS_T, lx = threadIdx.x % SS_T;
  float d = 0.f, ms = 0.f;
  if (y0 + ly < Ho && x0 + lx < Wo) {
    const float S = ssim_window(sx, sy, TW, ly, lx, P, nullptr);
    d = fminf(fmaxf((1.f - S) * 0.5f, 0.f), 1.f);
  }
  // sum(mask): every (b, mask channel) plane exactly once -- by the blocks of channel c < MC, over the INPUT pixels of
  // their tile (tiles cover [0, Ho) x [0, Wo); the last row / column of tiles also takes the 2 md border pixels)
  if (c < MC) {
    const int64_t HW = (int64_t)H * W;
    const int y1 = (ty == tiles_y - 1) ? H : y0 + SS_T, x1 = (tx == tiles_x - 1) ? W : x0 + SS_T;
    for (int gy = y0 + ly; gy < y1; gy += SS_T)
      for (int gx = x0 + lx; gx < x1; gx += SS_T) ms += mask[((int64_t)b * MC + c) * HW + (int64_t)gy * W + gx];
  }
  block_slot_add(d, ms, acc);
}

__global__ __launch_bounds__(SS_T * SS_T) void ssim_bwd_kernel(const float* __restrict__ a, const float* __restrict__ bb,
                                                               const float* __restrict__ mask, int MC, int B, int C, int H, int W,
                                                               int md, const float* __restrict__ acc,
                                                               const float* __restrict__ gscale, float scale,
                                                               float* __restrict__ g1, float* __restrict__ g2) {
  extern __shared__ float sm[];
  const int P = 2 * md + 1, halo = 2 * md, TW = SS_T + 2 * halo, Ho = H - 2 * md, Wo = W - 2 * md;
  float* sx = sm; float* sy = sm + TW * TW;
  const int tiles_x = (W + SS_T - 1) / SS_T, tiles_y = (H + SS_T - 1) / SS_T;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y; bid /= tiles_y;
  const int c = bid % C, b = bid / C;
  const int y0 = ty * SS_T, x0 = tx * SS_T;               // input pixel coordinates
  ssim_stage(a, bb, mask, MC, C, b, c, H, W, y0 - halo, x0 - halo, TW, sx, sy);
  const int ly = threadIdx.x / SS_T, lx = threadIdx.x % SS_T;
  const int gy = y0 + ly, gx = x0 + lx;
  if (gy >= H || gx >= W) return;
  const float coef = (gscale ? gscale[0] : 1.f) * scale / acc[1] / (float)(P * P);
  const float xq = sx[(ly + halo) * TW + lx + halo], yq = sy[(ly + halo) * TW + lx + halo];
  float dx = 0.f, dy = 0.f;
  for (int j = 0; j < P; ++j)
    for (int i = 0; i < P; ++i) {
      const int wy = gy - j, wx = gx - i;                 // window top-left (image coordinates) containing this pixel
      if (wy < 0 || wy >= Ho || wx < 0 || wx >= Wo) continue;
      float D[4];
      const float S = ssim_window(sx, sy, TW, wy - (y0 - halo), wx - (x0 - halo), P, D);
      const float h = (1.f - S) * 0.5f;
      if (h <= 0.f || h >= 1.f) continue;                 // clamp inactive only inside (0, 1)
      dx += -0.5f * (D[0] + 2.f * xq * D[2] + yq * D[3]);
      dy += -0.5f * (D[1] + 2.f * yq * D[2] + xq * D[3]);
    }
  const int64_t HW = (int64_t)H * W, r = (int64_t)gy * W + gx;
  const float m = mask[((int64_t)b * MC + (MC == 1 ? 0 : c)) * HW + r] * coef;
  if (g1) g1[((int64_t)b * C + c) * HW + r] = dx * m;
  if (g2) g2[((int64_t)b * C + c) * HW + r] = dy * m;
}

static int ssim_check(const float* im1, const float* im2, const float* mask, int MC, int B, int C, int H, int W, int md) {
  SININN_CHECK(im1 && im2 && mask, "ssim: null pointer");
  SININN_CHECK(B > 0 && C > 0 && (MC == 1 || MC == C) && md >= 1 && md <= 2 && H > 2 * md && W > 2 * md,
               "ssim: bad shape (md in 1..2, mask channels 1 or C)");
  return 0;
}

int ssim_fwd_launch(const float* im1, const float* im2, const float* mask, int MC, int B, int C, int H, int W, int md,
                    float weight, float* acc_zeroed, float* out, hipStream_t st) {
  if (int rc = ssim_check(im1, im2, mask, MC, B, C, H, W, md)) return rc;
  SININN_CHECK(acc_zeroed && out, "ssim: null pointer");
  const int Ho = H - 2 * md, Wo = W - 2 * md, TW = SS_T + 2 * md;
  const int tiles = B * C * ((Ho + SS_T - 1) / SS_T) * ((Wo + SS_T - 1) / SS_T);
  hipLaunchKernelGGL(ssim_fwd_kernel, dim3(tiles), dim3(SS_T * SS_T), 2 * TW * TW * sizeof(float), st, im1, im2, mask, MC, B, C, H,
                     W, md, acc_zeroed);
  SININN_LAUNCH_CHECK("ssim_fwd");
  // mean over B*C*Ho*Wo windows, times numel(mask) = B*MC*H*W, over sum(mask)
  const float scale = weight * ((float)B * MC * H * W) / ((float)B * C * Ho * Wo);
  hipLaunchKernelGGL(census_finish_kernel, dim3(1), dim3(CENSUS_SLOTS), 0, st, acc_zeroed, scale, out);
  SININN_LAUNCH_CHECK("ssim_finish");
  return 0;
}

int ssim_bwd_launch(const float* im1, const float* im2, const float* mask, int MC, int B, int C, int H, int W, int md,
                    float weight, const float* acc, const float* gscale, float* g1, float* g2, hipStream_t st) {
  if (int rc = ssim_check(im1, im2, mask, MC, B, C, H, W, md)) return rc;
  SININN_CHECK(acc && (g1 || g2), "ssim_bwd: null pointer");
  const int Ho = H - 2 * md, Wo = W - 2 * md, TW = SS_T + 4 * md;
  const int tiles = B * C * ((H + SS_T - 1) / SS_T) * ((W + SS_T - 1) / SS_T);
  const float scale = weight * ((float)B * MC * H * W) / ((float)B * C * Ho * Wo);
  hipLaunchKernelGGL(ssim_bwd_kernel, dim3(tiles), dim3(SS_T * SS_T), 2 * TW * TW * sizeof(float), st, im1, im2, mask, MC, B, C, H,
                     W, md, acc, gscale, scale, g1, g2);
  SININN_LAUNCH_CHECK("ssim_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// occlusion_brox (occlusions.py:111-118): forward-backward consistency on the backward flow warped by the forward flow
// (the warp itself is sininn_flow_warp_l1 == Resample2d):  mask = |fw + w(bw)|^2 >= 0.01 (|fw|^2 + |w(bw)|^2) + 0.5
// ------------------------------------------------------------------------------------------------
__global__ void brox_mask_kernel(const float* __restrict__ fw, const float* __restrict__ wbw, int B, int H, int W,
                                 uint8_t* __restrict__ mask) {
  const int64_t HW = (int64_t)H * W, total = (int64_t)B * HW;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int b = (int)(idx / HW);
  const int64_t r = idx % HW;
  const float fx = fw[((int64_t)b * 2 + 0) * HW + r], fy = fw[((int64_t)b * 2 + 1) * HW + r];
  const float wx = wbw[((int64_t)b * 2 + 0) * HW + r], wy = wbw[((int64_t)b * 2 + 1) * HW + r];
  const float sq_sum = (fx + wx) * (fx + wx) + (fy + wy) * (fy + wy);
  const float sum_sq = (fx * fx + wx * wx) + (fy * fy + wy * wy);
  mask[idx] = sq_sum >= 0.01f * sum_sq + 0.5f ? 1 : 0;
}

int brox_mask_launch(const float* fw, const float* warped_bw, int B, int H, int W, uint8_t* mask, hipStream_t st) {
  SININN_CHECK(fw && warped_bw && mask && B > 0 && H > 0 && W > 0, "occlusion_brox: bad arguments");
  const int64_t total = (int64_t)B * H * W;
  hipLaunchKernelGGL(brox_mask_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, fw, warped_bw, B, H, W, mask);
  SININN_LAUNCH_CHECK("occlusion_brox");
  return 0;
}

}  // namespace sininn

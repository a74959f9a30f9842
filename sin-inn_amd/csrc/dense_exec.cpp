// Host-side executor of one IRN DenseBlock (+ InvBlockExp tail) per C-ABI call: the launch sequence that sin-inn_amd/irn.py
// used to issue kernel by kernel from Python (~20 ctypes calls and as many torch allocations per block pass made the IRN
// training step host-bound: 31 ms per step regardless of the batch).  Reference: archs.py:74-160.
#include <stdlib.h>

#include "common.h"

namespace sininn {

int conv_launch(const sininn_conv_args* a, hipStream_t st);
int permute_launch(const float* in, int in_stride, float* out, int out_stride, int64_t M, int C, const int* idx, hipStream_t st);
int lrelu_bwd_launch(float* g, int g_stride, const float* f, int f_stride, int64_t M, int n, float slope, hipStream_t st);
int irn_coupling_bwd_launch(const float* dy, int dy_stride, const float* vy, int vy_stride, const float* hval, int64_t M,
                            int Co, float clamp, int inverse, float* dG, int dG_stride, int dG_pad, float* dh, float* dv,
                            int dv_stride, hipStream_t st);
size_t wgrad_group_workspace_bytes(const sininn_wgrad_item* items, int n, int B, int H, int W, int ksize);
int wgrad_group_launch(const sininn_wgrad_item* items, int n, int B, int H, int W, int ksize, void* ws, size_t ws_bytes,
                       hipStream_t st);
int copy_channels_launch(const float* in, int in_stride, float* out, int out_stride, int64_t M, int C, int Cpad, hipStream_t st);

// per-class launch brackets of bench.py's roofline.classes (glow_exec.cpp); IRN uses the class slots as
//   0 conv1-4 forward (+LeakyReLU)   1 conv5 forward (+ fused tail)   2 data gradient of conv5   3 data gradients of conv1-4
//   4 the five weight gradients (one grouped launch pair)             5 elementwise (HBM-bound, no FLOPs counted)
hipEvent_t class_scope_open(int cls, int ksize, double flops, hipStream_t st);
void class_scope_close(hipEvent_t end, hipStream_t st);
struct Scope {
  hipStream_t st; hipEvent_t b;
  Scope(int cls, double flops, hipStream_t s) : st(s), b(class_scope_open(cls, 3, flops, s)) {}
  ~Scope() { class_scope_close(b, st); }
};
static inline double cflops(int64_t M, int cin, int n) { return 2.0 * (double)M * 9 * cin * n; }

static constexpr int GC = 32;
static constexpr float SLOPE = 0.2f;
static inline int pad8(int n) { return (n + 7) / 8 * 8; }
static inline int pad16(int n) { return (n + 15) / 16 * 16; }
static inline int pad32(int n) { return (n + 31) / 32 * 32; }

static void dense_items(const sininn_dense_args* a, sininn_wgrad_item it[5]) {
  const int cinp = pad8(a->cin), bw = cinp + 4 * GC, coutp = pad8(a->cout);
  for (int i = 0; i < 5; ++i) {
    it[i] = wgrad_item_init();
    const int k = cinp + GC * i;
    it[i].in = a->buf; it[i].in_stride = bw; it[i].Cin = k;
    if (i < 4) { it[i].dout = a->dF ? a->dF + k : nullptr; it[i].dout_stride = bw; it[i].N = GC; }
    else { it[i].dout = a->dD; it[i].dout_stride = coutp; it[i].N = a->cout; }
    it[i].gw = a->gw[i]; it[i].gb = a->gb[i];
    it[i].gap_begin = a->cin; it[i].gap_len = cinp - a->cin;
  }
}

size_t dense_workspace_bytes(int B, int H, int W, int cin, int cout) {
  sininn_dense_args a = {};
  a.B = B; a.H = H; a.W = W; a.cin = cin; a.cout = cout;
  sininn_wgrad_item it[5];
  dense_items(&a, it);
  // worst case over the live subsets (a frozen conv drops out of the group; fewer output tiles -> more pixel splits each)
  size_t w = 0;
  for (int mask = 1; mask < 32; ++mask) {
    sininn_wgrad_item sub[5];
    int n = 0;
    for (int i = 0; i < 5; ++i)
      if (mask & (1 << i)) sub[n++] = it[i];
    const size_t bytes = wgrad_group_workspace_bytes(sub, n, B, H, W, 3);
    w = bytes > w ? bytes : w;
  }
  return w;
}

static int check(const sininn_dense_args* a, const char* who) {
  SININN_CHECK(a != nullptr, "%s: null args", who);
  SININN_CHECK(a->struct_bytes == sizeof(sininn_dense_args), "%s: struct_bytes = %zu, this library's sininn_dense_args has %zu (ABI %d)",
               who, a->struct_bytes, sizeof(sininn_dense_args), SININN_ABI_VERSION);
  SININN_CHECK(a->B > 0 && a->H > 0 && a->W > 0, "%s: bad shape", who);
  SININN_CHECK(a->cin > 0 && a->cin % 4 == 0 && a->cout > 0 && a->cout % 4 == 0, "%s: channel counts must be multiples of 4", who);
  SININN_CHECK(a->mode >= 0 && a->mode <= 3, "%s: mode %d", who, a->mode);
  SININN_CHECK(a->x && a->buf && a->out, "%s: null tensor", who);
  for (int i = 0; i < 5; ++i) SININN_CHECK(a->w_fwd[i] && a->b_fwd[i], "%s: missing packed weights", who);
  SININN_CHECK(a->mode == 0 || a->aux1, "%s: mode %d needs aux1", who, a->mode);
  SININN_CHECK(a->mode < 2 || (a->aux2 && a->clamp > 0.f), "%s: IRN tail needs aux2 and clamp", who);
  // extents (ABI v4): every buffer the launch sequence addresses at a size it derives from (B, H, W, cin, cout)
  const size_t M = (size_t)a->B * a->H * a->W, bw = (size_t)pad8(a->cin) + 4 * GC;
  SININN_CHECK(a->buf_floats >= M * bw, "%s: buf holds %zu floats, the feature buffer needs M * (pad8(cin) + 128) = %zu", who,
               a->buf_floats, M * bw);
  SININN_CHECK(a->out_floats >= M * a->cout, "%s: out holds %zu floats, needs M * cout = %zu", who, a->out_floats, M * a->cout);
  SININN_CHECK(a->mode < 2 || a->aux2_floats >= M * a->cout, "%s: aux2 holds %zu floats, needs M * cout = %zu", who,
               a->aux2_floats, M * a->cout);
  return 0;
}

static int check_backward_extents(const sininn_dense_args* a) {
  const size_t M = (size_t)a->B * a->H * a->W, bw = (size_t)pad8(a->cin) + 4 * GC, cout = a->cout, coutp = pad8(a->cout);
  SININN_CHECK(a->dout_floats >= M * cout, "dense_backward: dout holds %zu floats, needs M * cout = %zu", a->dout_floats, M * cout);
  SININN_CHECK(a->dF_floats >= M * bw, "dense_backward: dF holds %zu floats, needs M * (pad8(cin) + 128) = %zu", a->dF_floats, M * bw);
  SININN_CHECK(!a->dD || a->dD_floats >= M * coutp, "dense_backward: dD holds %zu floats, needs M * pad8(cout) = %zu", a->dD_floats,
               M * coutp);
  if (a->mode >= 2) {
    SININN_CHECK(a->dh_floats >= M * cout && a->dv_floats >= M * cout, "dense_backward: dh / dv hold %zu / %zu floats, need M * cout = %zu",
                 a->dh_floats, a->dv_floats, M * cout);
  }
  return 0;
}

int dense_forward(const sininn_dense_args* a, hipStream_t st) {
  if (int rc = check(a, "dense_forward")) return rc;
  const int64_t M = (int64_t)a->B * a->H * a->W;
  const int cin = a->cin, cinp = pad8(cin), bw = cinp + 4 * GC, cout = a->cout;
  // x -> buf[:, :cin], pad channels [cin, cinp) zeroed
  {
    Scope sc(5, 0.0, st);
    if (int rc = copy_channels_launch(a->x, a->x_stride, a->buf, bw, M, cin, cinp, st)) return rc;
  }
  for (int i = 0; i < 4; ++i) {
    const int k = cinp + GC * i;
    Scope sc(0, cflops(M, cin + GC * i, GC), st);
    sininn_conv_args c = {};
    c.in = a->buf; c.in_stride = bw; c.Cin = k; c.w = a->w_fwd[i]; c.bias = a->b_fwd[i]; c.Np = GC; c.winograd = a->winograd;
    c.B = a->B; c.H = a->H; c.W = a->W; c.ksize = 3; c.mode = SININN_CONV_LRELU; c.clamp = SLOPE;
    c.out = a->buf + k; c.out_stride = bw; c.N = GC;
    if (int rc = conv_launch(&c, st)) return rc;
  }
  sininn_conv_args c = {};
  c.in = a->buf; c.in_stride = bw; c.Cin = bw; c.w = a->w_fwd[4]; c.bias = a->b_fwd[4]; c.Np = pad16(pad8(cout)); c.winograd = a->winograd;
  c.B = a->B; c.H = a->H; c.W = a->W; c.ksize = 3; c.out = a->out; c.out_stride = cout; c.N = cout;
  if (a->mode == 0) c.mode = SININN_CONV_LINEAR;
  else if (a->mode == 1) { c.mode = SININN_CONV_ADD; c.addend = a->aux1; c.addend_stride = a->aux1_stride; }
  else {
    c.mode = a->mode == 2 ? SININN_CONV_IRN_FWD : SININN_CONV_IRN_INV;
    c.v = a->aux1; c.v_stride = a->aux1_stride; c.mask = a->aux2; c.mask_stride = cout; c.clamp = a->clamp;
  }
  Scope sc(1, cflops(M, cin + 4 * GC, cout), st);
  return conv_launch(&c, st);
}

int order_after(hipStream_t waiter, hipStream_t producer);     // glow_exec.cpp: per-device event ring

int dense_backward(const sininn_dense_args* a, hipStream_t st, hipStream_t wst) {
  if (int rc = check(a, "dense_backward")) return rc;
  SININN_CHECK(a->dout && a->dF && a->workspace, "dense_backward: null tensor");
  for (int i = 0; i < 5; ++i) SININN_CHECK(a->w_dgrad[i], "dense_backward: missing dgrad weights");
  const int64_t M = (int64_t)a->B * a->H * a->W;
  const int cin = a->cin, cinp = pad8(cin), bw = cinp + 4 * GC, cout = a->cout, coutp = pad8(cout);
  const bool irn = a->mode >= 2;
  SININN_CHECK(!irn || (a->dD && a->dh && a->dv), "dense_backward: IRN tail needs dD, dh, dv");
  SININN_CHECK(coutp == cout || a->dD, "dense_backward: cout %% 8 != 0 needs dD");
  if (int rc = check_backward_extents(a)) return rc;
  // ---- tail: gradient w.r.t. conv5's output (K = coutp channels for its data-gradient conv) --------------------------
  const float* dD = a->dout;                    // [M][coutp]
  int dD_stride = cout;
  if (irn) {
    const int inv = a->mode == 3 ? 1 : 0;
    const float* vy = inv ? a->out : a->aux1;
    const int vs = inv ? cout : a->aux1_stride;
    // dG goes straight into dD with zero pad columns: the data-gradient conv of conv5 has K = coutp
    Scope sc(5, 0.0, st);
    if (int rc = irn_coupling_bwd_launch(a->dout, cout, vy, vs, a->aux2, M, cout, a->clamp, inv, a->dD, coutp, coutp, a->dh,
                                         a->dv, cout, st)) return rc;
    dD = a->dD; dD_stride = coutp;
  } else if (coutp != cout) {
    Scope sc(5, 0.0, st);
    if (int rc = copy_channels_launch(a->dout, cout, a->dD, coutp, M, cout, coutp, st)) return rc;
    dD = a->dD; dD_stride = coutp;
  }
  // The data gradient of conv i+1 writes (conv5) or accumulates (conv2-4) dF[:, :k_{i+1}); its last 32 columns are feature
  // slot i, whose gradient is final at that point: the LeakyReLU backward of slot i is applied in that conv's epilogue
  // (`tail` = first column of the slot; the mask is the saved feature buffer), not by a launch of its own.
  auto dgrad = [&](const float* src, int src_stride, int n_src, const float* w, int n_out, bool accumulate, int tail) -> int {
    sininn_conv_args c = {};
    c.in = src; c.in_stride = src_stride; c.Cin = n_src; c.w = w; c.winograd = a->winograd;
    c.Np = a->winograd ? pad32(n_out) : pad16(n_out);
    c.B = a->B; c.H = a->H; c.W = a->W; c.ksize = 3; c.out = a->dF; c.out_stride = bw; c.N = n_out;
    if (accumulate) { c.mode = SININN_CONV_ADD; c.addend = a->dF; c.addend_stride = bw; }
    else c.mode = SININN_CONV_LINEAR;
    if (tail >= 0) { c.mask = a->buf; c.mask_stride = bw; c.Co = tail; c.clamp = SLOPE; }
    return conv_launch(&c, st);
  };
  {
    Scope sc(2, cflops(M, cout, cin + 4 * GC), st);
    if (int rc = dgrad(dD, dD_stride, coutp, a->w_dgrad[4], bw, false, cinp + GC * 3)) return rc;
  }
  for (int i = 3; i >= 0; --i) {
    const int k = cinp + GC * i;
    Scope sc(3, cflops(M, GC, cin + GC * i), st);
    if (int rc = dgrad(a->dF + k, bw, GC, a->w_dgrad[i], k, true, i > 0 ? cinp + GC * (i - 1) : -1)) return rc;
  }
  // ---- the five weight gradients: every dF slot is final now -> one grouped launch pair on the weight-gradient stream ----
  sininn_wgrad_item it[5];
  sininn_dense_args b = *a;
  b.dD = const_cast<float*>(dD);
  dense_items(&b, it);
  it[4].dout_stride = dD_stride;
  int n = 0;
  sininn_wgrad_item live[5];
  for (int i = 0; i < 5; ++i)
    if (it[i].gw) live[n++] = it[i];
  if (n > 0) {
    if (int rc = order_after(wst, st)) return rc;
    double fl = 0.0;
    for (int i = 0; i < 5; ++i)
      if (it[i].gw) fl += cflops(M, cin + GC * i, i < 4 ? GC : cout);
    Scope sc(4, fl, wst);
    if (int rc = wgrad_group_launch(live, n, a->B, a->H, a->W, 3, a->workspace, a->workspace_bytes, wst)) return rc;
  }
  return 0;
}

}  // namespace sininn

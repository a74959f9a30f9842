// Host-side executor of one GLOW coupling block (forward / inverse and their backward): the launch sequence of
// FrEIA's GLOWCouplingBlock (wired at archs.py:61-64) as ONE C-ABI call, so the Python layer
// costs one ctypes call per block pass instead of one per kernel (the training step is ~450 launches; issued from
// Python they made the host the bottleneck).  Weight-gradient kernels go to a second stream (they depend on dr / dh
// only) so they overlap the data-gradient chain.
#include <stdlib.h>
#include <mutex>
#include <utility>
#include <vector>

#include <cstdlib>
#include "common.h"

namespace sininn {

int conv_launch(const sininn_conv_args* a, hipStream_t st);
int conv_pair_k1_supported(const sininn_conv_args* f, const sininn_conv_args* s);
int conv_pair_k1_launch(const sininn_conv_args* f, const sininn_conv_args* s, hipStream_t st);
int conv_pair_k1_preferred(const sininn_conv_args* f);
int conv_sub3_bf16_supported(const sininn_conv_args* f, const sininn_conv_args* s);
int conv_sub3_bf16_launch(const sininn_conv_args* f, const sininn_conv_args* s, hipStream_t st);
bool sub3_fusion_enabled();
size_t wgrad_workspace_bytes(int N, int Cin, int ksize, int B, int H, int W);
int wgrad_launch(const float* in, int in_stride, int Cin, const float* dout, int dout_stride, int N, int B, int H, int W,
                 int ksize, float* gw, float* gb, void* ws, size_t ws_bytes, hipStream_t st);
size_t wgrad_group_workspace_bytes(const sininn_wgrad_item* items, int n, int B, int H, int W, int ksize);
int wgrad_group_launch(const sininn_wgrad_item* items, int n, int B, int H, int W, int ksize, void* ws, size_t ws_bytes,
                       hipStream_t st);
bool wgrad_grouping_enabled();
int wgrad_group_mode();
int coupling_bwd_launch(const float* dy, int dy_stride, const int* dy_map, const float* vy, int vy_stride,
                        const int* vy_map, const float* s, const float* gld, int B, int HW, int Co, float clamp,
                        int inverse, float* dr, float* dv, int dv_stride, hipStream_t st);

static inline size_t align64(size_t n) { return (n + 63) & ~(size_t)63; }   // floats -> keeps 256-byte alignment
static inline int pad16i(int n) { return (n + 15) / 16 * 16; }

// ---- cross-stream ordering: a small ring of timing-less events PER DEVICE (events belong to the device that was current
// when they were created; one process per GPU is the deployment, but the library must not break with two devices) -----------
static std::mutex g_ev_mutex;
struct EventRing { hipEvent_t ev[64]; bool ready = false; unsigned next = 0; };
static EventRing g_rings[16];

// conv_sub1.hip: the whole backward of a fp32 1x1 subnet in one persistent launch (h recomputed, dh on chip)
int conv_sub1_bwd_shape_supported(int ksize, int dtype, int cond_cin, int co);
size_t conv_sub1_bwd_workspace_bytes(int cond_cin, int co);
int conv_sub1_bwd_launch(const sininn_conv_args* rc, const sininn_conv_args* d2, const sininn_conv_args* d1, int no_dx, void* ws,
                         size_t ws_bytes, int* slabs_out, hipStream_t st);
int conv_sub1_bwd_reduce(int cond_cin, int co, const void* ws, int slabs, float* gw2, float* gb2, float* gw1, float* gb1, hipStream_t st);
int conv_sub1_fwd_supported(const sininn_conv_args* f, const sininn_conv_args* s);
int conv_sub1_fwd_launch(const sininn_conv_args* f, const sininn_conv_args* s, hipStream_t st);
// conv_sub1_bf16.hip: data gradients of a wide (level-1) 1x1 subnet + the weight gradient of its conv1 in one persistent launch
size_t conv_sub1_bf16_wide_ws_bytes(int ksize, int dtype, int cond_cin, int co);
int conv_sub1_bf16_wide_bwd_supported(const sininn_conv_args* d2, const sininn_conv_args* d1);
int conv_sub1_bf16_wide_bwd_wg1_launch(const sininn_conv_args* d2, const sininn_conv_args* d1, const float* x, int x_stride, void* ws, size_t ws_bytes,
                                       int* slabs_out, hipStream_t st);
int conv_sub1_bf16_wide_reduce(const void* ws, int slabs, float* gw1, float* gb1, hipStream_t st);
// conv3_smallk_bf16.hip: the small-K 3x3 convs of a level-0 subnet; `bits` = the ReLU gates as a bit mask [pixel][8] words
int conv3_smallk_bf16_supported(const sininn_conv_args* a);
int conv3_smallk_bf16_launch_bits(const sininn_conv_args* a, unsigned* bits, hipStream_t st);
size_t conv_sub1_bf16_wide_wg2_ws_bytes(int ksize, int dtype, int cond_cin, int co);
int conv_sub1_bf16_wide_wg2_launch(const float* dr, int dr_stride, const void* h, int h_stride, int B, int H, int W, void* ws, size_t ws_bytes,
                                   float* gw2, float* gb2, hipStream_t st);

int order_after(hipStream_t waiter, hipStream_t producer) {
  if (waiter == producer) return 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { set_error("glow: bad current device"); return 1; }
  hipEvent_t ev;
  {
    std::lock_guard<std::mutex> lock(g_ev_mutex);
    EventRing& ring = g_rings[dev];
    if (!ring.ready) {
      for (auto& e : ring.ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { set_error("glow: cannot create events"); return 1; }
      ring.ready = true;
    }
    ev = ring.ev[ring.next++ % 64];
  }
  if (hipEventRecord(ev, producer) != hipSuccess || hipStreamWaitEvent(waiter, ev, 0) != hipSuccess) {
    set_error("glow: stream ordering failed");
    return 1;
  }
  return 0;
}

// ---- live timing of the dominant kernel (bench.py): HIP events on the launch stream around every forward 3x3
// coupling conv (conv2 + affine epilogue) of the level whose image height is g_prof_h ----------------------------
static int g_prof_h = 0;
static unsigned long long* g_prof_stamps = nullptr;   // device words, 2 per timed launch (optional)
static int g_prof_max = 0;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_events;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;

void profile_begin(int h, unsigned long long* stamps, int max_launches) {
  std::lock_guard<std::mutex> lock(g_ev_mutex);
  g_prof_stamps = stamps; g_prof_max = stamps ? max_launches : 0;
  for (auto& p : g_prof_events) g_prof_pool.push_back(p);
  g_prof_events.clear();
  g_prof_h = h;
}

int profile_end(int* count, float* total_ms) {
  std::lock_guard<std::mutex> lock(g_ev_mutex);
  g_prof_h = 0; g_prof_stamps = nullptr; g_prof_max = 0;
  float tot = 0.f;
  int n = 0;
  for (auto& p : g_prof_events) {
    float ms = 0.f;
    if (hipEventSynchronize(p.second) != hipSuccess || hipEventElapsedTime(&ms, p.first, p.second) != hipSuccess) {
      set_error("profile_end: event query failed");
      return 1;
    }
    tot += ms; ++n;
    g_prof_pool.push_back(p);
  }
  g_prof_events.clear();
  if (count) *count = n;
  if (total_ms) *total_ms = tot;
  return 0;
}

static bool prof_pair(hipEvent_t* a, hipEvent_t* b, unsigned long long** stamp) {
  std::lock_guard<std::mutex> lock(g_ev_mutex);
  if (g_prof_events.size() >= 8192) return false;
  std::pair<hipEvent_t, hipEvent_t> p;
  if (!g_prof_pool.empty()) { p = g_prof_pool.back(); g_prof_pool.pop_back(); }
  else if (hipEventCreate(&p.first) != hipSuccess || hipEventCreate(&p.second) != hipSuccess) return false;
  const int slot = (int)g_prof_events.size();
  *stamp = (slot < g_prof_max) ? g_prof_stamps + 2 * (size_t)slot : nullptr;
  g_prof_events.push_back(p);
  *a = p.first; *b = p.second;
  return true;
}

// ---- per-class timing (bench.py roofline.classes): every launch of the block executor bracketed by HIP events, tagged
// with its class and its ALGORITHMIC FLOPs (direct-convolution count 2 M k^2 Cin N).  Meaningful on ONE stream only (the
// bracket then is the kernel's own duration); bench.py runs a few single-stream steps after the timed region for it. ----
enum { PC_CONV1 = 0, PC_COUPLE = 1, PC_DGRAD2 = 2, PC_DGRAD1 = 3, PC_WGRAD = 4, PC_CBWD = 5, PC_PER_K = 6, PC_N = 12 };
struct ClassRec { hipEvent_t a, b; int cls; double flops; double bytes; };
static double g_pc_bytes[PC_N] = {};        // algorithmic HBM bytes per class of the last profile_classes_end (1x1 launches report them)
static bool g_pc_on = false;
static std::vector<ClassRec> g_pc_recs;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pc_pool;

void profile_classes_begin() {
  std::lock_guard<std::mutex> lock(g_ev_mutex);
  for (auto& r : g_pc_recs) g_pc_pool.push_back({r.a, r.b});
  g_pc_recs.clear();
  g_pc_on = true;
}

int profile_classes_end(int n, double* ms, double* flops, int* launches) {
  std::lock_guard<std::mutex> lock(g_ev_mutex);
  g_pc_on = false;
  for (int i = 0; i < n; ++i) { ms[i] = 0.0; flops[i] = 0.0; launches[i] = 0; }
  for (int i = 0; i < PC_N; ++i) g_pc_bytes[i] = 0.0;
  for (auto& r : g_pc_recs) {
    float t = 0.f;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) {
      set_error("profile_classes_end: event query failed");
      return 1;
    }
    if (r.cls < n) { ms[r.cls] += t; flops[r.cls] += r.flops; launches[r.cls] += 1; }
    if (r.cls < PC_N) g_pc_bytes[r.cls] += r.bytes;
    g_pc_pool.push_back({r.a, r.b});
  }
  g_pc_recs.clear();
  return 0;
}

// algorithmic HBM bytes per class summed over the launches of the last profile_classes_end (0 where a launch did not report them)
int profile_classes_bytes(int n, double* bytes) {
  std::lock_guard<std::mutex> lock(g_ev_mutex);
  for (int i = 0; i < n; ++i) bytes[i] = i < PC_N ? g_pc_bytes[i] : 0.0;
  return 0;
}

// start bracket of one launch (nullptr when class profiling is off); class_scope_close records the end event
hipEvent_t class_scope_open_b(int cls, int ksize, double flops, hipStream_t st, double bytes) {
  if (!g_pc_on) return nullptr;
  std::lock_guard<std::mutex> lock(g_ev_mutex);
  if (g_pc_recs.size() >= 60000) return nullptr;
  std::pair<hipEvent_t, hipEvent_t> p;
  if (!g_pc_pool.empty()) { p = g_pc_pool.back(); g_pc_pool.pop_back(); }
  else if (hipEventCreate(&p.first) != hipSuccess || hipEventCreate(&p.second) != hipSuccess) return nullptr;
  g_pc_recs.push_back(ClassRec{p.first, p.second, cls + (ksize == 1 ? PC_PER_K : 0), flops, bytes});
  (void)hipEventRecord(p.first, st);
  return p.second;
}
hipEvent_t class_scope_open(int cls, int ksize, double flops, hipStream_t st) { return class_scope_open_b(cls, ksize, flops, st, 0.0); }   // dense_exec.cpp
void class_scope_close(hipEvent_t end, hipStream_t st) { if (end) (void)hipEventRecord(end, st); }

struct ClassScope {           // RAII bracket: records the start event now and the end event when it goes out of scope
  hipStream_t st; hipEvent_t b;
  ClassScope(int cls, int ksize, double flops, hipStream_t s, double bytes = 0.0) : st(s), b(class_scope_open_b(cls, ksize, flops, s, bytes)) {}
  ~ClassScope() { class_scope_close(b, st); }
};
static inline double conv_flops(size_t M, int k, int cin, int n) { return 2.0 * (double)M * k * k * cin * n; }

static const bool g_group_major = getenv("SININN_GROUP_MAJOR") == nullptr || atoi(getenv("SININN_GROUP_MAJOR")) != 0;   // A/B switch

// h / dh of this subnet are channel-group-major: fp32, 3x3, every conv of the subnet on the Winograd kernels, and the
// tensor small enough for the 32-bit staging offsets of the kernels that read it: the Winograd convs advance through the
// 256 / 8 channel groups with a 32-bit byte offset (conv_prepare: (Cin / 8) * group stride * 4 < 2^32) and the weight-gradient
// staging descriptors hold 32-bit byte offsets from a tile origin (wgrad_group_plan: the same product + a 64-row tile span).
// Larger tensors (M = B*H*W >= ~2^22 pixels, e.g. 1024 x 1024 at batch 16, level 0) take the row-major hidden layout.
bool group_major_fits(size_t M, int W) {
  const unsigned long long gs = (unsigned long long)M * 8ull;                       // floats between channel groups
  return (SININN_HIDDEN / 8) * gs * 4ull + 64ull * (unsigned long long)W * 8ull * 4ull < (1ull << 32);
}
static bool group_major_hidden(const sininn_glow_args* a, const sininn_subnet* net) {
  const size_t M = (size_t)a->B * a->H * a->W;
  return g_group_major && a->dtype == 0 && a->ksize == 3 && (net->winograd & 15) == 15 && group_major_fits(M, a->W) && wgrad_grouping_enabled();
}

// The subnet of this half runs its backward as ONE persistent launch that recomputes h (conv_sub1.hip): the forward pass then
// does not store the hidden tensor.  A property of the shape alone, so the forward and the backward call agree on it.
static bool fused_sub1(const sininn_glow_args* a, int cond_cin, int co) {
  return conv_sub1_bwd_shape_supported(a->ksize, a->dtype, cond_cin, co) != 0;
}

struct Half {                 // one half-coupling in execution order
  const sininn_subnet* net;
  int cond_off;               // offset of the conditioning channels in x (-1: the first half's compact output)
  int base;                   // offset of the transformed channels v in x == output channel base
  int co;                     // channels transformed
};

static void halves_of(const sininn_glow_args* a, Half h[2]) {
  const int l1 = a->C / 2, l2 = a->C - l1;
  if (!a->rev) { h[0] = Half{&a->s2, l1, 0, l1}; h[1] = Half{&a->s1, -1, l1, l2}; }
  else { h[0] = Half{&a->s1, 0, l1, l2}; h[1] = Half{&a->s2, -1, 0, l1}; }
}

struct Saved { float *h_a, *h_b, *s_a, *s_b, *ybuf; unsigned *bits_a, *bits_b; size_t total; };
static inline size_t padded_pixels(int B, int H, int W) { return (size_t)B * ((H + 15) / 16 * 16) * ((W + 15) / 16 * 16); }
// Mpad: pixels of the image padded to whole 16 x 16 tiles (the gate bit masks are indexed by tile step)
static Saved saved_layout(float* base, size_t M, int co_a, int co_b, bool bf16 = false, size_t Mpad = 0) {
  Saved s;
  size_t o = 0;
  const size_t hid = bf16 ? M * SININN_HIDDEN / 2 : M * SININN_HIDDEN;      // bf16 hidden tensors take half the floats
  s.h_a = base + o; o += align64(hid);
  s.h_b = base + o; o += align64(hid);
  s.s_a = base + o; o += align64(M * co_a);
  s.s_b = base + o; o += align64(M * co_b);
  s.ybuf = base + o; o += align64(M * co_a);
  // mixed precision: the ReLU gates of the two hidden tensors as bit masks, [pixel][256 / 32] words (the small-K 3x3 kernels)
  s.bits_a = s.bits_b = nullptr;
  if (bf16) {
    s.bits_a = reinterpret_cast<unsigned*>(base + o); o += align64(Mpad * (SININN_HIDDEN / 32));
    s.bits_b = reinterpret_cast<unsigned*>(base + o); o += align64(Mpad * (SININN_HIDDEN / 32));
  }
  s.total = o;
  return s;
}

size_t glow_saved_floats(int B, int H, int W, int C, int dtype) {
  const size_t M = (size_t)B * H * W;
  const int big = C - C / 2;
  const size_t hid = dtype == 1 ? M * SININN_HIDDEN / 2 : M * SININN_HIDDEN;
  return 2 * align64(hid) + 3 * align64(M * big) + (dtype == 1 ? 2 * align64(padded_pixels(B, H, W) * (SININN_HIDDEN / 32)) : 0) + 64;
}

struct Scratch { float *dr_b, *dh_b, *dy_first, *dr_a, *dh_a; void* ws; size_t ws_bytes; void* slab[2]; size_t slab_bytes[2]; void* slab2[2]; size_t slab2_bytes[2]; size_t total_bytes; };
static Scratch scratch_layout(void* basep, int B, int H, int W, int C, int ksize, int co_a, int co_b, int dtype = 0) {
  const size_t M = (size_t)B * H * W;
  float* base = static_cast<float*>(basep);
  Scratch s;
  size_t o = 0;
  s.dr_b = base + o; o += align64(M * 2 * co_b);
  s.dh_b = base + o; o += align64(M * SININN_HIDDEN);
  s.dy_first = base + o; o += align64(M * co_a);
  s.dr_a = base + o; o += align64(M * 2 * co_a);
  s.dh_a = base + o; o += align64(M * SININN_HIDDEN);
  size_t w = 0;
  const int cond_a = C - co_a;
  const size_t cands[4] = {wgrad_workspace_bytes(2 * co_b, SININN_HIDDEN, ksize, B, H, W),
                           wgrad_workspace_bytes(SININN_HIDDEN, co_a, ksize, B, H, W),
                           wgrad_workspace_bytes(2 * co_a, SININN_HIDDEN, ksize, B, H, W),
                           wgrad_workspace_bytes(SININN_HIDDEN, cond_a, ksize, B, H, W)};
  for (size_t c : cands) w = c > w ? c : w;
  {  // grouped weight gradients: the four convs' slabs live side by side
    const int cond_b = co_a;
    sininn_wgrad_item it[4] = {wgrad_item_init(), wgrad_item_init(), wgrad_item_init(), wgrad_item_init()};
    it[0].Cin = SININN_HIDDEN; it[0].N = 2 * co_b;
    it[1].Cin = cond_b;        it[1].N = SININN_HIDDEN;
    it[2].Cin = SININN_HIDDEN; it[2].N = 2 * co_a;
    it[3].Cin = cond_a;        it[3].N = SININN_HIDDEN;
    // Worst case over every sub-group that can be launched: a smaller group (per-half mode, a frozen conv whose gw is NULL)
    // gets MORE pixel splits per problem (S = 512 / output tiles) and can need more slab bytes than the full group
    // (8.52 MB against 8.47 MB for a 1x1 bf16 block at C = 48).  15 subsets x 2 precisions of host arithmetic.
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1) {  // mixed-precision path: 64 x 64 tiles on the bf16 matrix pipe (conv2: h is bf16, conv1: dh is bf16)
        it[0].in_bf16 = it[2].in_bf16 = 1; it[1].dout_bf16 = it[3].dout_bf16 = 1;
      }
      for (int mask = 1; mask < 16; ++mask) {
        sininn_wgrad_item sub[4];
        int n = 0;
        for (int i = 0; i < 4; ++i)
          if (mask & (1 << i)) sub[n++] = it[i];
        const size_t bytes = wgrad_group_workspace_bytes(sub, n, B, H, W, ksize);
        w = bytes > w ? bytes : w;
      }
    }
  }
  s.ws = base + o;
  s.ws_bytes = w;
  // slabs of the fused 1x1 subnet backward (conv_sub1.hip), one region per half: the reduce of the first-processed half runs
  // on the weight-gradient stream while the second half's kernel fills its own region
  size_t so = (o * sizeof(float) + w + 255) / 256 * 256;
  const int cond_cin[2] = {C - co_a, co_a}, cos[2] = {co_a, co_b};
  for (int i = 0; i < 2; ++i) {
    s.slab_bytes[i] = conv_sub1_bwd_shape_supported(ksize, dtype, cond_cin[i], cos[i]) ? conv_sub1_bwd_workspace_bytes(cond_cin[i], cos[i]) : 0;
    const size_t wide = conv_sub1_bf16_wide_ws_bytes(ksize, dtype, cond_cin[i], cos[i]);
    if (wide > s.slab_bytes[i]) s.slab_bytes[i] = wide;
    s.slab[i] = reinterpret_cast<char*>(base) + so;
    so += (s.slab_bytes[i] + 255) / 256 * 256;
    // ... and of conv2's weight gradient of a wide 1x1 subnet on the mixed-precision path (its own kernel, weight-gradient stream)
    s.slab2_bytes[i] = conv_sub1_bf16_wide_wg2_ws_bytes(ksize, dtype, cond_cin[i], cos[i]);
    s.slab2[i] = reinterpret_cast<char*>(base) + so;
    so += (s.slab2_bytes[i] + 255) / 256 * 256;
  }
  s.total_bytes = so;
  return s;
}

size_t glow_scratch_bytes(int B, int H, int W, int C, int ksize, int dtype) {
  const int big = C - C / 2;
  // upper bound over both directions (co_a / co_b are C/2 and C - C/2 in some order)
  Scratch s = scratch_layout(nullptr, B, H, W, C, ksize, big, big, dtype);
  return s.total_bytes + 256;
}

static int check_common(const sininn_glow_args* a, const char* who) {
  SININN_CHECK(a != nullptr, "%s: null args", who);
  SININN_CHECK(a->B > 0 && a->H > 0 && a->W > 0 && a->C >= 16, "%s: bad shape", who);
  SININN_CHECK((a->C / 2) % 8 == 0 && (a->C - a->C / 2) % 8 == 0, "%s: both channel halves must be multiples of 8 (C=%d)", who, a->C);
  SININN_CHECK(a->ksize == 1 || a->ksize == 3, "%s: ksize %d", who, a->ksize);
  SININN_CHECK(a->x && a->out && a->saved, "%s: null tensor", who);
  SININN_CHECK(a->s1.w1 && a->s1.w2 && a->s2.w1 && a->s2.w2 && a->s1.b1 && a->s1.b2 && a->s2.b1 && a->s2.b2, "%s: missing packed weights", who);
  return 0;
}

static int col_tile_of(int co) { return (co % 16 == 0) ? 32 : 16; }

// ReLU gates of one subnet, as the forward pass took them: gates[m][j] = (h[m][j] > 0), read from `saved` with the layout /
// dtype the executor chose (row-major fp32 or bf16, channel-group-major fp32).  Parity tooling: with these gates forced, a
// float64 evaluation of the same network is a smooth function of the same inputs (tests/test_gpu_gates.py).
__global__ void hidden_gates_kernel(const float* __restrict__ h, int64_t M, int group_major, int bf16, unsigned char* __restrict__ gates) {
  const int64_t total = M * SININN_HIDDEN;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / SININN_HIDDEN;
    const int j = (int)(i % SININN_HIDDEN);
    float v;
    if (bf16) v = (float)reinterpret_cast<const __bf16*>(h)[i];
    else if (group_major) v = h[(int64_t)(j >> 3) * M * 8 + m * 8 + (j & 7)];
    else v = h[i];
    gates[i] = v > 0.f ? 1 : 0;
  }
}

int glow_hidden_gates(const sininn_glow_args* a, int which, unsigned char* gates, hipStream_t st) {
  SININN_CHECK(a && a->saved && gates && (which == 0 || which == 1), "glow_hidden_gates: bad arguments");
  SININN_CHECK(a->B > 0 && a->H > 0 && a->W > 0 && a->C >= 16, "glow_hidden_gates: bad shape");
  const size_t M = (size_t)a->B * a->H * a->W;
  Half hv[2];
  halves_of(a, hv);
  Saved sv = saved_layout(a->saved, M, hv[0].co, hv[1].co, a->dtype == 1, padded_pixels(a->B, a->H, a->W));
  float* h = which == 0 ? sv.h_a : sv.h_b;
  const int gm = group_major_hidden(a, hv[which].net) ? 1 : 0;
  {
    // a subnet whose backward recomputes h (conv_sub1.hip) did not store it: conv1 is run again into the (reserved, unused)
    // slot of `saved` -- from the input the forward pass read: x for the first half, the saved compact output for the second --
    // and with the arithmetic the forward pass and the backward recompute use: stage 1 of the pair kernel (a stand-alone conv
    // sums in another order, and a unit within rounding distance of 0 would then report a gate the passes did not take).  The
    // pair's second conv is a throw-away linear conv into the other, equally unused hidden slot.
    const int cond_cin = which == 0 ? a->C - hv[0].co : hv[0].co;
    if (fused_sub1(a, cond_cin, hv[which].co)) {
      sininn_conv_args c1 = {};
      if (which == 0) { c1.in = a->x + hv[0].cond_off; c1.in_stride = a->C; }
      else { c1.in = sv.ybuf; c1.in_stride = hv[0].co; }
      c1.Cin = cond_cin; c1.w = hv[which].net->w1; c1.bias = hv[which].net->b1; c1.Np = SININN_HIDDEN;
      c1.B = a->B; c1.H = a->H; c1.W = a->W; c1.ksize = 1; c1.mode = SININN_CONV_RELU;
      c1.out = h; c1.out_stride = SININN_HIDDEN; c1.N = SININN_HIDDEN;
      if (a->dtype == 1) {
        // mixed precision: h = bf16(relu(fp32 sum + b1)), the sum taken over two 16-channel steps of v_mfma_f32_32x32x16_bf16
        // from zero -- the stand-alone bf16 conv, the pair kernel and the persistent kernels all issue exactly that chain
        c1.w_bf16 = 1; c1.out_bf16 = 1;
        if (int rc = conv_launch(&c1, st)) return rc;
      } else {
      sininn_conv_args c2 = {};
      c2.in = h; c2.in_stride = SININN_HIDDEN; c2.Cin = SININN_HIDDEN; c2.w = hv[which].net->w2; c2.Np = 2 * hv[which].co;
      c2.B = a->B; c2.H = a->H; c2.W = a->W; c2.ksize = 1; c2.mode = SININN_CONV_LINEAR;
      c2.out = which == 0 ? sv.h_b : sv.h_a; c2.out_stride = 2 * hv[which].co; c2.N = 2 * hv[which].co;
      SININN_CHECK(conv_pair_k1_supported(&c1, &c2), "glow_hidden_gates: the 1x1 pair kernel is switched off (needed to reproduce the recomputed h)");
      if (int rc = conv_pair_k1_launch(&c1, &c2, st)) return rc;
      }
    }
  }
  const int64_t total = (int64_t)M * SININN_HIDDEN;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(hidden_gates_kernel, dim3(blocks), dim3(256), 0, st, h, (int64_t)M, gm, a->dtype == 1 ? 1 : 0, gates);
  SININN_LAUNCH_CHECK("glow_hidden_gates");
  return 0;
}

int glow_forward(const sininn_glow_args* a, hipStream_t st) {
  if (int rc = check_common(a, "glow_forward")) return rc;
  const size_t M = (size_t)a->B * a->H * a->W;
  const int C = a->C;
  Half hv[2];
  halves_of(a, hv);
  Saved sv = saved_layout(a->saved, M, hv[0].co, hv[1].co, a->dtype == 1, padded_pixels(a->B, a->H, a->W));
  const int mode = a->rev ? SININN_CONV_COUPLE_INV : SININN_CONV_COUPLE_FWD;
  const bool bf16 = a->dtype == 1;
  for (int i = 0; i < 2; ++i) {
    const Half& h = hv[i];
    float* hbuf = i == 0 ? sv.h_a : sv.h_b;
    float* sbuf = i == 0 ? sv.s_a : sv.s_b;
    sininn_conv_args c1 = {};
    if (i == 0) { c1.in = a->x + h.cond_off; c1.in_stride = C; c1.Cin = C - h.co; }
    else { c1.in = sv.ybuf; c1.in_stride = hv[0].co; c1.Cin = hv[0].co; }
    c1.w = h.net->w1; c1.bias = h.net->b1; c1.Np = SININN_HIDDEN; c1.winograd = (h.net->winograd & 1) && a->ksize == 3;
    c1.B = a->B; c1.H = a->H; c1.W = a->W; c1.ksize = a->ksize; c1.mode = SININN_CONV_RELU;
    c1.out = hbuf; c1.out_stride = SININN_HIDDEN; c1.N = SININN_HIDDEN;
    if (bf16) { c1.winograd = 0; c1.w_bf16 = 1; c1.in_bf16 = 0; c1.out_bf16 = 1; }     // cond fp32 -> h bf16
    // fp32 3x3 blocks on the Winograd kernels keep h / dh channel-group-major [256/8][M][8]: the halo rows of an 8-channel
    // chunk are then contiguous runs for the K = 256 convs that read them (dominant kernel -10 %)
    const bool gm = group_major_hidden(a, h.net);
    if (gm) c1.out_group_stride = (int)(M * 8);
    sininn_conv_args c2 = {};
    c2.in = hbuf; c2.in_stride = SININN_HIDDEN; c2.Cin = SININN_HIDDEN;
    c2.w = h.net->w2; c2.bias = h.net->b2; c2.Np = 2 * h.co; c2.winograd = (h.net->winograd & 2) && a->ksize == 3;
    c2.B = a->B; c2.H = a->H; c2.W = a->W; c2.ksize = a->ksize; c2.mode = mode;
    if (a->dst_map) { c2.out = a->out; c2.out_map = a->dst_map + h.base; }
    else { c2.out = a->out + h.base; c2.out_map = nullptr; }
    c2.out_stride = C;
    c2.v = a->x + h.base; c2.v_stride = C;
    c2.out2 = (i == 0) ? sv.ybuf : nullptr; c2.out2_stride = h.co;
    c2.sbuf = sbuf; c2.logdet = a->logdet; c2.Co = h.co; c2.clamp = a->clamp; c2.col_tile = col_tile_of(h.co);
    if (bf16) { c2.winograd = 0; c2.w_bf16 = 1; c2.in_bf16 = 1; }                       // h bf16 -> fp32 coupling epilogue
    if (gm) { c2.in_stride = 8; c2.in_group_stride = (int)(M * 8); }
    if (a->no_save) c2.sbuf = nullptr;               // s is kept for the backward pass only
    // 3x3 subnets of a no-grad pass on the mixed-precision path: conv1 -> conv2 -> coupling + log-det in ONE launch, the hidden
    // tile never leaves LDS (conv_sub3_bf16.hip; north_star's single fused coupling kernel)
    if (a->no_save && bf16 && a->ksize == 3 && sub3_fusion_enabled()) {
      sininn_conv_args f1 = c1;
      f1.out = nullptr;
      if (conv_sub3_bf16_supported(&f1, &c2)) {
        ClassScope sc(PC_COUPLE, a->ksize, conv_flops(M, a->ksize, c1.Cin, SININN_HIDDEN) + conv_flops(M, a->ksize, SININN_HIDDEN, 2 * h.co), st);
        if (int rc = conv_sub3_bf16_launch(&f1, &c2, st)) return rc;
        continue;
      }
    }
    // 1x1 subnets (fp32): both convs in one launch, the hidden tile stays in LDS between them (conv_pair_k1.hip)
    const bool recompute = fused_sub1(a, c1.Cin, h.co);      // the backward recomputes h from this half's input
    // algorithmic HBM bytes of a fused 1x1 forward: x, v in; y (+ its compact copy for the first half), s out; + the hidden tensor when stored
    const double fwd_bytes = 4.0 * (double)M * (c1.Cin + 2 * h.co + (a->no_save ? 0 : h.co) + (i == 0 ? h.co : 0)) +
                             ((a->no_save || recompute) ? 0.0 : (double)M * SININN_HIDDEN * (bf16 ? 2 : 4));
    if (recompute && conv_sub1_fwd_supported(&c1, &c2)) {
      // ... and the forward is the persistent twin of that kernel: conv1 -> conv2 -> coupling + log-det, weights resident on chip
      ClassScope sc(PC_COUPLE, a->ksize, conv_flops(M, a->ksize, c1.Cin, SININN_HIDDEN) + conv_flops(M, a->ksize, SININN_HIDDEN, 2 * h.co), st, fwd_bytes);
      if (int rc = conv_sub1_fwd_launch(&c1, &c2, st)) return rc;
      continue;
    }
    if (conv_pair_k1_supported(&c1, &c2) && (a->no_save || recompute || conv_pair_k1_preferred(&c1))) {
      if (a->no_save || recompute) c1.out = nullptr; // ... and then the hidden tensor never reaches HBM
      ClassScope sc(PC_COUPLE, a->ksize, conv_flops(M, a->ksize, c1.Cin, SININN_HIDDEN) + conv_flops(M, a->ksize, SININN_HIDDEN, 2 * h.co), st, fwd_bytes);
      if (int rc = conv_pair_k1_launch(&c1, &c2, st)) return rc;
      continue;
    }
    {
      ClassScope sc(PC_CONV1, a->ksize, conv_flops(M, a->ksize, c1.Cin, SININN_HIDDEN), st);
      if (bf16 && a->ksize == 3 && !a->no_save && conv3_smallk_bf16_supported(&c1)) {
        // small-K conv1 of a training pass: the gates go out as a bit mask too (the backward's masked data gradient reads it)
        if (int rc = conv3_smallk_bf16_launch_bits(&c1, i == 0 ? sv.bits_a : sv.bits_b, st)) return rc;
      } else if (int rc = conv_launch(&c1, st)) return rc;
    }
    hipEvent_t e0, e1;
    unsigned long long* stamp = nullptr;
    const bool timed = g_prof_h != 0 && a->ksize == 3 && a->H == g_prof_h && prof_pair(&e0, &e1, &stamp);
    c2.stamp = (timed && a->dtype == 0) ? stamp : nullptr;   // entry / exit window stamps: fp32 kernels only (the bf16 conv
                                                             // kernel reads `stamp` as its 8-word phase accumulator)
    if (timed) (void)hipEventRecord(e0, st);
    {
      ClassScope sc(PC_COUPLE, a->ksize, conv_flops(M, a->ksize, SININN_HIDDEN, 2 * h.co), st);
      if (int rc = conv_launch(&c2, st)) return rc;
    }
    if (timed) (void)hipEventRecord(e1, st);
  }
  return 0;
}

int glow_backward(const sininn_glow_args* a, hipStream_t st, hipStream_t wst) {
  if (int rc = check_common(a, "glow_backward")) return rc;
  SININN_CHECK(a->dout && a->dx && a->scratch, "glow_backward: null tensor");
  SININN_CHECK(a->s1.w1_dgrad && a->s1.w2_dgrad && a->s2.w1_dgrad && a->s2.w2_dgrad, "glow_backward: missing dgrad weights");
  const size_t M = (size_t)a->B * a->H * a->W;
  const int C = a->C, HW = a->H * a->W, B = a->B, H = a->H, W = a->W, k = a->ksize;
  Half hv[2];
  halves_of(a, hv);
  const int co_a = hv[0].co, co_b = hv[1].co, base_a = hv[0].base, base_b = hv[1].base;
  Saved sv = saved_layout(a->saved, M, co_a, co_b, a->dtype == 1, padded_pixels(a->B, a->H, a->W));
  Scratch sc = scratch_layout(a->scratch, B, H, W, C, k, co_a, co_b, a->dtype);
  SININN_CHECK(a->scratch_bytes >= sc.total_bytes, "glow_backward: scratch too small (%zu < %zu)", a->scratch_bytes, sc.total_bytes);
  const int inv = a->rev ? 1 : 0;
  const int* map_a = a->dst_map ? a->dst_map + base_a : nullptr;
  const int* map_b = a->dst_map ? a->dst_map + base_b : nullptr;

  // grouped weight gradients: the four problems are collected and launched together once the last of their inputs (the
  // first half's dh) has been queued; one launch pair on the weight-gradient stream instead of four.
  // 3x3 blocks: the grouped launch saves ~0.3 ms of kernel time per step (slab reduce 0.28 -> 0.09 ms, gradient kernels -4 %).
  // 1x1 blocks: until the staging of the f32 kernels was rewritten (round 3, WgStage) their per-conv kernels were faster (N = 48
  // runs a 48 x 64 tile there, 64 x ... padded here); since then the group wins: class 1.00 -> 0.86 ms per step, step 8.60 ->
  // 8.49 ms in an A/B on one box (SININN_WGRAD_GROUP_K1=0 restores the per-conv launches).
  const bool bf16 = a->dtype == 1;
  static const bool group_k1 = !(getenv("SININN_WGRAD_GROUP_K1") && atoi(getenv("SININN_WGRAD_GROUP_K1")) == 0);
  const bool grouped = bf16 || (wgrad_grouping_enabled() && (k == 3 || group_k1));   // the bf16-operand loads exist in the grouped kernels
  const bool per_half = wgrad_group_mode() == 2;
  sininn_wgrad_item items[4] = {};
  int n_items = 0;
  auto add_item = [&](const float* in, int in_stride, int cin, const float* dout, int dout_stride, int n, float* gw, float* gb,
                      int in_b, int dout_b, int in_gs, int dout_gs) {
    sininn_wgrad_item& it = items[n_items++];
    it = wgrad_item_init();                    // every optional field (gap_*, group strides, dtype flags) defaults to 0
    it.in = in; it.in_stride = in_stride; it.Cin = cin; it.dout = dout; it.dout_stride = dout_stride; it.N = n; it.gw = gw; it.gb = gb;
    it.in_bf16 = in_b; it.dout_bf16 = dout_b; it.in_group_stride = in_gs; it.dout_group_stride = dout_gs;
  };

  // fuse: when set, the dgrad of this half's first conv also performs the coupling-tail backward of the OTHER
  // (first-executed) half in its epilogue, writing dr_a / dx instead of the intermediate gradient
  struct Fuse { const float* vy; int vy_stride; const float* s; float* dr; };
  auto half_bwd = [&](const Half& h, const float* hbuf, const float* sbuf, float* dr, float* dh,
                      const float* dy, int dy_stride, const int* dy_map, const float* vy, int vy_stride, const int* vy_map,
                      const float* cond, int cond_stride, int cond_cin,
                      const float* addend, int add_stride, const int* add_map, float* dcond, int dcond_stride,
                      bool do_coupling, const Fuse* fuse, bool skip_d1, bool last_half) -> int {
    const sininn_subnet* net = h.net;
    if (do_coupling) {
      ClassScope sc(PC_CBWD, 3, 0.0, st);
      if (int rc = coupling_bwd_launch(dy, dy_stride, dy_map, vy, vy_stride, vy_map, sbuf, a->gld, B, HW, h.co, a->clamp, inv,
                                       dr, a->dx + h.base, C, st)) return rc;
    }
    const bool gm = group_major_hidden(a, net);
    const int gs = gm ? (int)(M * 8) : 0;
    const bool fused = fused_sub1(a, cond_cin, h.co);       // conv_sub1.hip: its weight gradients come out of the same launch
    const int which_half = (&h == &hv[0]) ? 0 : 1;
    const bool wide_wg2 = bf16 && k == 1 && sc.slab2_bytes[which_half] > 0;
    if (net->gw2 && !fused && wide_wg2) {
      // wide 1x1 subnet, mixed precision: dW2 += dr^T h as its own persistent kernel (h read straight into MFMA operand registers)
      if (int rc = order_after(wst, st)) return rc;
      ClassScope scp(PC_WGRAD, k, conv_flops(M, k, SININN_HIDDEN, 2 * h.co), wst,
                     4.0 * (double)M * 2 * h.co + 2.0 * (double)M * SININN_HIDDEN + 2.0 * 256.0 * (2 * h.co + 1) * SININN_HIDDEN * 4.0);   // dr, bf16 h, slabs written + read
      if (int rc = conv_sub1_bf16_wide_wg2_launch(dr, 2 * h.co, hbuf, SININN_HIDDEN, B, H, W, sc.slab2[which_half], sc.slab2_bytes[which_half],
                                                  net->gw2, net->gb2, wst)) return rc;
    } else if (net->gw2 && !fused) {
      if (grouped) add_item(hbuf, SININN_HIDDEN, SININN_HIDDEN, dr, 2 * h.co, 2 * h.co, net->gw2, net->gb2, bf16 ? 1 : 0, 0, gs, 0);
      else {
        if (int rc = order_after(wst, st)) return rc;
        ClassScope scp(PC_WGRAD, k, conv_flops(M, k, SININN_HIDDEN, 2 * h.co), wst);
        if (int rc = wgrad_launch(hbuf, SININN_HIDDEN, SININN_HIDDEN, dr, 2 * h.co, 2 * h.co, B, H, W, k, net->gw2, net->gb2,
                                  sc.ws, sc.ws_bytes, wst)) return rc;
      }
    }
    sininn_conv_args d2 = {};
    d2.in = dr; d2.in_stride = 2 * h.co; d2.Cin = 2 * h.co; d2.w = net->w2_dgrad; d2.Np = SININN_HIDDEN;
    d2.winograd = (net->winograd & 8) && k == 3;
    d2.B = B; d2.H = H; d2.W = W; d2.ksize = k; d2.mode = SININN_CONV_MASK;
    d2.out = dh; d2.out_stride = SININN_HIDDEN; d2.N = SININN_HIDDEN; d2.mask = hbuf; d2.mask_stride = SININN_HIDDEN;
    if (bf16) { d2.winograd = 0; d2.w_bf16 = 1; d2.in_bf16 = 0; d2.out_bf16 = 1; d2.mask_bf16 = 1; }   // dr fp32 -> dh bf16
    if (gm) { d2.out_group_stride = gs; d2.mask_group_stride = gs; }
    // the data gradient of conv1 (d1) is described first: a 1x1 pair (d2 -> d1) runs as one launch with dh kept in LDS
    // (still written once for conv1's weight gradient)
    sininn_conv_args d1 = {};
    d1.in = dh; d1.in_stride = SININN_HIDDEN; d1.Cin = SININN_HIDDEN; d1.w = net->w1_dgrad;
    d1.winograd = (net->winograd & 4) && k == 3;
    if (bf16) { d1.winograd = 0; d1.w_bf16 = 1; d1.in_bf16 = 1; }                                       // dh bf16 -> fp32 epilogue
    if (gm) { d1.in_stride = 8; d1.in_group_stride = gs; }
    d1.Np = d1.winograd ? (cond_cin + 31) / 32 * 32 : pad16i(cond_cin);
    d1.B = B; d1.H = H; d1.W = W; d1.ksize = k; d1.mode = SININN_CONV_ADD;
    d1.out = dcond; d1.out_stride = dcond_stride; d1.N = cond_cin;
    d1.addend = addend; d1.addend_stride = add_stride; d1.addend_map = add_map;
    if (fuse) {
      d1.mode = inv ? SININN_CONV_ADD_CBWD_INV : SININN_CONV_ADD_CBWD_FWD;
      d1.out = fuse->dr; d1.out_stride = 2 * cond_cin;
      d1.v = fuse->vy; d1.v_stride = fuse->vy_stride; d1.sbuf = const_cast<float*>(fuse->s);
      d1.out2 = a->dx + hv[0].base; d1.out2_stride = C;
      d1.logdet = const_cast<float*>(a->gld); d1.Co = cond_cin; d1.clamp = a->clamp;
    }
    if (fused) {
      // fp32 1x1 subnet at a shape the persistent kernel serves: h is recomputed from `cond`, dh stays on chip, both data
      // gradients and both weight gradients come out of one launch; the slab reduce joins the other += on the weight-gradient stream
      sininn_conv_args rc = {};
      rc.in = cond; rc.in_stride = cond_stride; rc.Cin = cond_cin; rc.w = net->w1; rc.bias = net->b1; rc.Np = SININN_HIDDEN;
      rc.B = B; rc.H = H; rc.W = W; rc.ksize = 1; rc.mode = SININN_CONV_RELU;
      if (bf16) rc.w_bf16 = 1;
      const int which = (&h == &hv[0]) ? 0 : 1;
      int slabs = 0;
      {
        // algorithmic bytes: dr, x in; the epilogue's side inputs and outputs (ADD: addend -> out; fused coupling backward: addend, v, s -> 3 outputs)
        const double epi = (fuse ? 6.0 : 2.0) * cond_cin;
        ClassScope scp(PC_DGRAD2, k, 2.0 * (conv_flops(M, k, 2 * h.co, SININN_HIDDEN) + conv_flops(M, k, SININN_HIDDEN, cond_cin)), st,
                       4.0 * (double)M * (2 * h.co + cond_cin + (skip_d1 ? 0.0 : epi)));
        if (int rc_ = conv_sub1_bwd_launch(&rc, &d2, &d1, skip_d1 ? 1 : 0, sc.slab[which], sc.slab_bytes[which], &slabs, st)) return rc_;
      }
      if (net->gw1 || net->gw2) {
        if (int rc_ = order_after(wst, st)) return rc_;
        ClassScope scp(PC_WGRAD, k, 0.0, wst, (double)slabs * ((double)2 * h.co * SININN_HIDDEN + (double)SININN_HIDDEN * (cond_cin + 8)) * 4.0);   // the slabs, read once
        if (int rc_ = conv_sub1_bwd_reduce(cond_cin, h.co, sc.slab[which], slabs, net->gw2, net->gb2, net->gw1, net->gb1, wst)) return rc_;
      }
      return 0;
    }
    {
      // wide 1x1 subnet on the mixed-precision path (level 1): both data gradients AND conv1's weight gradient in one persistent
      // launch -- dh never reaches HBM, conv1's problem leaves the grouped weight-gradient launch (conv2's stays: it was added above)
      const int which = (&h == &hv[0]) ? 0 : 1;
      const size_t wide_ws = conv_sub1_bf16_wide_ws_bytes(k, a->dtype, cond_cin, h.co);
      sininn_conv_args d2w = d2;
      d2w.out = nullptr;
      if (bf16 && !skip_d1 && net->gw1 && wide_ws > 0 && sc.slab_bytes[which] >= wide_ws && conv_pair_k1_supported(&d2, &d1) &&
          conv_sub1_bf16_wide_bwd_supported(&d2w, &d1)) {
        int slabs = 0;
        {
          const double epi = (fuse ? 6.0 : 2.0) * cond_cin;
          ClassScope scp(PC_DGRAD2, k, conv_flops(M, k, 2 * h.co, SININN_HIDDEN) + 2.0 * conv_flops(M, k, SININN_HIDDEN, cond_cin), st,
                         4.0 * (double)M * (2 * h.co + cond_cin + epi) + 2.0 * (double)M * SININN_HIDDEN);      // dr, x, side inputs / outputs + the bf16 h (mask)
          if (int rc_ = conv_sub1_bf16_wide_bwd_wg1_launch(&d2w, &d1, cond, cond_stride, sc.slab[which], sc.slab_bytes[which], &slabs, st)) return rc_;
        }
        if (int rc_ = order_after(wst, st)) return rc_;
        ClassScope scp(PC_WGRAD, k, 0.0, wst, (double)slabs * (cond_cin + 1) * SININN_HIDDEN * 4.0);
        return conv_sub1_bf16_wide_reduce(sc.slab[which], slabs, net->gw1, net->gb1, wst);
      }
    }
    const bool pair = !skip_d1 && conv_pair_k1_supported(&d2, &d1);
    if (pair) {
      ClassScope scp(PC_DGRAD2, k, conv_flops(M, k, 2 * h.co, SININN_HIDDEN) + conv_flops(M, k, SININN_HIDDEN, cond_cin), st);
      if (int rc = conv_pair_k1_launch(&d2, &d1, st)) return rc;
    } else {
      ClassScope scp(PC_DGRAD2, k, conv_flops(M, k, 2 * h.co, SININN_HIDDEN), st);
      // the forward pass wrote the gate bit mask iff its conv1 ran on the small-K kernel: the same predicate on the same descriptor
      sininn_conv_args f1 = {};
      f1.in = cond; f1.in_stride = cond_stride; f1.Cin = cond_cin; f1.w = net->w1; f1.bias = net->b1; f1.Np = SININN_HIDDEN;
      f1.B = B; f1.H = H; f1.W = W; f1.ksize = k; f1.mode = SININN_CONV_RELU;
      f1.out = const_cast<float*>(hbuf); f1.out_stride = SININN_HIDDEN; f1.N = SININN_HIDDEN; f1.w_bf16 = 1; f1.out_bf16 = 1;
      if (bf16 && k == 3 && conv3_smallk_bf16_supported(&f1) && conv3_smallk_bf16_supported(&d2)) {
        if (int rc = conv3_smallk_bf16_launch_bits(&d2, which_half == 0 ? sv.bits_a : sv.bits_b, st)) return rc;
      } else if (int rc = conv_launch(&d2, st)) return rc;
    }
    if (net->gw1) {
      if (grouped) add_item(cond, cond_stride, cond_cin, dh, SININN_HIDDEN, SININN_HIDDEN, net->gw1, net->gb1, 0, bf16 ? 1 : 0, 0, gs);
      else {
        if (int rc = order_after(wst, st)) return rc;
        ClassScope scp(PC_WGRAD, k, conv_flops(M, k, cond_cin, SININN_HIDDEN), wst);
        if (int rc = wgrad_launch(cond, cond_stride, cond_cin, dh, SININN_HIDDEN, SININN_HIDDEN, B, H, W, k, net->gw1, net->gb1,
                                  sc.ws, sc.ws_bytes, wst)) return rc;
      }
      if (grouped && (last_half || per_half) && n_items > 0) {     // every input of the group is queued on `st` now
        if (int rc = order_after(wst, st)) return rc;
        double fl = 0.0;
        for (int i = 0; i < n_items; ++i) fl += conv_flops(M, k, items[i].Cin, items[i].N);
        ClassScope scp(PC_WGRAD, k, fl, wst);
        if (int rc = wgrad_group_launch(items, n_items, B, H, W, k, sc.ws, sc.ws_bytes, wst)) return rc;
        n_items = 0;
      }
    }
    if (skip_d1 || pair) return 0;                   // the gradient w.r.t. this half's condition is not needed / already done
    ClassScope scp(PC_DGRAD1, k, conv_flops(M, k, SININN_HIDDEN, cond_cin), st);
    return conv_launch(&d1, st);
  };

  // ---- second half first: its condition is the first half's compact output -------------------------
  {
    const float* dy = a->dst_map ? a->dout : a->dout + base_b;
    const float* vy; int vy_stride; const int* vy_map = nullptr;
    if (!a->rev) { vy = a->x + base_b; vy_stride = C; }
    else if (a->dst_map) { vy = a->out; vy_stride = C; vy_map = map_b; }
    else { vy = a->out + base_b; vy_stride = C; }
    const float* addend = a->dst_map ? a->dout : a->dout + base_a;
    Fuse fz;
    fz.vy = a->rev ? sv.ybuf : a->x + base_a;
    fz.vy_stride = a->rev ? co_a : C;
    fz.s = sv.s_a;
    fz.dr = sc.dr_a;
    if (int rc = half_bwd(hv[1], sv.h_b, sv.s_b, sc.dr_b, sc.dh_b, dy, C, map_b, vy, vy_stride, vy_map,
                          sv.ybuf, co_a, co_a, addend, C, map_a, sc.dy_first, co_a, true, &fz, false, false)) return rc;
  }
  // ---- first half: condition = x[:, cond range]; its data gradient accumulates in place into dx ----
  {
    const int cond_off = hv[0].cond_off, cond_cin = C - co_a;
    const float* vy = a->rev ? sv.ybuf : a->x + base_a;
    const int vy_stride = a->rev ? co_a : C;
    if (int rc = half_bwd(hv[0], sv.h_a, sv.s_a, sc.dr_a, sc.dh_a, sc.dy_first, co_a, nullptr, vy, vy_stride, nullptr,
                          a->x + cond_off, C, cond_cin, a->dx + cond_off, C, nullptr, a->dx + cond_off, C, false,
                          nullptr, a->skip_dx != 0, true)) return rc;
  }
  if (grouped && n_items > 0) {                      // a frozen conv1 in the last half: flush what was collected
    if (int rc = order_after(wst, st)) return rc;
    double fl = 0.0;
    for (int i = 0; i < n_items; ++i) fl += conv_flops(M, k, items[i].Cin, items[i].N);
    ClassScope scp(PC_WGRAD, k, fl, wst);
    if (int rc = wgrad_group_launch(items, n_items, B, H, W, k, sc.ws, sc.ws_bytes, wst)) return rc;
  }
  return 0;
}

}  // namespace sininn

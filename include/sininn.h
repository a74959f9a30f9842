/*
 * sininn.h -- C ABI of libsininn.so: the MI355X (gfx950) kernels behind the sin-inn
 * single-video INN training path.
 *
 * Boundary rules:
 *   - extern "C", plain pointers and sizes only; no torch types.
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocates and frees);
 *     the library borrows it for the duration of the call and never allocates device memory.
 *     Scratch is a caller-provided workspace pointer + size in bytes.
 *   - every call is an asynchronous launch on the hipStream_t passed as `stream`
 *     (torch.cuda.current_stream().cuda_stream); no internal threads, no hidden syncs.
 *   - return value: 0 on success, non-zero on error (argument check or hipError_t);
 *     sininn_last_error() returns a thread-local message.  The Python side turns this into
 *     RuntimeError, mirroring the assert / NotImplementedError convention of the reference's only
 *     raw-pointer kernel call site (video-interpolation/my_utils/softsplat.py:239-331).
 *   - activations are fp32 NHWC ("pixel-major"): element (b,y,x,c) of a tensor with C channels
 *     and pixel stride `stride` floats lives at ((b*H+y)*W+x)*stride + c.  A channel sub-range is
 *     addressed by offsetting the base pointer (stride stays the full pixel stride).
 *
 * All file:line citations are into the reference repository (paramhanji/sin-inn).
 */
#ifndef SININN_H
#define SININN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SININN_ABI_VERSION 4
#define SININN_HIDDEN 256 /* hidden width of subnet_conv / subnet_conv_1x1, archs.py:12,16 */

int sininn_version(void);
const char* sininn_last_error(void);
/* sizeof() of descriptor struct `which` as THIS library was compiled: 0 sininn_conv_args, 1 sininn_wgrad_item,
 * 2 sininn_dense_args, 3 sininn_glow_args, 4 sininn_subnet, 5 sininn_pack_desc; 0 for an unknown index.  A binding written in
 * another language (the ctypes mirrors in sin-inn_amd/_lib.py) checks its own layout against it at load time (ABI v4). */
size_t sininn_sizeof(int which);
/* A HIP stream at an explicit priority (lower number = higher priority; range from sininn_stream_priority_range: `least` is the
 * numerically largest, lowest priority).  torch.cuda.Stream only offers {high, normal}; the weight-gradient stream of the
 * training step can be created LOW with this (SININN_WGRAD_PRIO) and wrapped with torch.cuda.ExternalStream.  The stream lives
 * until the process exits.  ABI v4. */
int sininn_stream_priority_range(int* least, int* greatest);
int sininn_stream_create(int priority, void** stream);
/* Stream-capture diagnostics (hipGraph replay of the pass chains, lit_wrapper._graph_step): which of `n` helper streams belong to
 * the capture that `origin` started and still hold work that is NOT joined back into origin.  flags[i]: 0 not part of it,
 * 1 capturing and joined, 2 capturing and unjoined (hipStreamEndCapture on origin would fail).  Host-only graph walk. */
int sininn_capture_unjoined(void* origin, void** streams, int n, int* flags);

/* ------------------------------------------------------------------------------------------------
 * Weight packing.  Source: torch Conv2d weight, OIHW fp32 [N][Cin][k][k] (archs.py:12-13,16-17).
 *   w_fwd  [taps][Np ][Cin ]  row q holds output channel colmap[q] (or q when colmap==NULL; rows
 *                             whose source index is <0 or >=N are zero)  -- B operand of the conv.
 *   w_dgrad[taps][Cdp][N   ]  w_dgrad[t][c][n] = w[n][c][taps-1-t]       -- B operand of the
 *                             data-gradient conv (input channels N, output channels Cin padded to
 *                             Cdp, rows c>=Cin zero).
 *   b_fwd  [Np] packed bias (may be NULL together with bias).
 * Either destination may be NULL.
 * ---------------------------------------------------------------------------------------------- */
int sininn_pack_conv_weights(const float* w_oihw, const float* bias, int N, int Cin, int ksize,
                             const int* colmap, int Np, float* w_fwd, float* b_fwd,
                             int Cdp, float* w_dgrad, void* stream);

/* bf16 packs for the mixed-precision convs: wb_fwd [taps][Np][Kp] (Kp = Cin rounded up to 16, zero filled), b_fwd [Np]
 * fp32 (packed bias), wb_dgrad [taps][Cdp][Kd] (Kd = N rounded up to 16; flipped taps).  Either destination may be NULL. */
int sininn_pack_conv_weights_bf16(const float* w_oihw, const float* bias, int N, int Cin, int ksize,
                                  const int* colmap, int Np, void* wb_fwd, float* b_fwd,
                                  int Cdp, void* wb_dgrad, void* stream);

/* Winograd F(2x2,3x3) filter transform of a 3x3 conv weight (U = G g G^T), same row / column conventions as
 * sininn_pack_conv_weights with the tap axis replaced by the 16 transform positions:
 *   u_fwd [16][Cin/8][Np][8], u_dgrad [16][N/8][Cdp][8] (flipped taps): channel-chunk-major, so the 8-channel chunk a
 *   kernel iteration stages is contiguous per position.  Either destination may be NULL. */
int sininn_pack_winograd(const float* w_oihw, int N, int Cin, const int* colmap, int Np, float* u_fwd,
                         int Cdp, float* u_dgrad, void* stream);

/* All weight packs of a model in ONE launch (the optimiser step invalidates every pack at once; 48 small pack
 * launches per training step cost ~0.2 ms).  `descs` is a DEVICE array of n descriptors; destinations and
 * conventions are those of sininn_pack_conv_weights (wino_* == 0) / sininn_pack_winograd (wino_* != 0, ksize 3);
 * b_fwd (packed bias) is written whenever it is non-NULL.  work_begin = exclusive prefix sum of
 * sininn_pack_work_items over the array (host-computed), total_work = its grand total. */
typedef struct sininn_pack_desc {
  const float* w; const float* bias; int N, Cin, ksize; const int* colmap; int Np;
  float* w_fwd; float* b_fwd; int Cdp; float* w_dgrad; int wino_fwd, wino_dgrad;
  int work_begin;
  /* ABI v4, zero = as before.  N / Cin above are the PACKED dimensions; the source weight may be smaller: src_n outputs
   * (0 = N; packed outputs beyond it are zero) and Cin - gap_len input channels, the packed input channels [gap_begin,
   * gap_begin + gap_len) being zero padding it does not have (IRN DenseBlock: channel_in padded to a multiple of 8 inside
   * the feature buffer, archs.py:74-98).  Lets every pack of an IRN model join the one batched refresh launch. */
  int src_n, gap_begin, gap_len;
} sininn_pack_desc;
int sininn_pack_work_items(const sininn_pack_desc* host_desc);
int sininn_pack_batch(const sininn_pack_desc* descs, int n, int total_work, void* stream);

/* Packed column order used by the coupling epilogue for a subnet with 2*Co outputs (s | t):
 * `tile`-column MFMA tile q = [ s[h*q .. h*q+h-1] | t[h*q .. h*q+h-1] ], h = tile/2, tile in {16, 32}
 * (tile 32 needs Co % 16 == 0).  Writes 2*Co ints (host memory).  The same `tile` must be passed as
 * sininn_conv_args.col_tile. */
void sininn_coupling_colmap(int Co, int tile, int* colmap_host);

/* ------------------------------------------------------------------------------------------------
 * Convolution engine (implicit GEMM on v_mfma_f32_16x16x4_f32, LDS-staged halo tiles).
 * One entry point, epilogue selected by `mode`.  Replaces nn.Conv2d + the elementwise tail of
 * FrEIA's GLOWCouplingBlock.forward (call site archs.py:61-64).
 * ---------------------------------------------------------------------------------------------- */
enum sininn_conv_mode {
  SININN_CONV_RELU = 0,       /* out = relu(conv + bias)                    (archs.py:12,16: Conv+ReLU)   */
  SININN_CONV_COUPLE_FWD = 1, /* y = exp(log_e(s)) * v + t ; logdet += sum log_e(s)                        */
  SININN_CONV_COUPLE_INV = 2, /* y = (v - t) / exp(log_e(s)) ; logdet -= sum log_e(s)                      */
  SININN_CONV_MASK = 3,       /* out = conv * (mask > 0)   (data gradient through the ReLU)                */
  SININN_CONV_ADD = 4,        /* out = conv (+ bias if given) + addend[addend_map]   (data gradient + skip grad;
                                 with bias: y1 = x1 + F(x2), archs.py:151)                                  */
  SININN_CONV_LINEAR = 5,     /* out = conv + bias                                                          */
  /* LINEAR and ADD with `mask` != NULL (ABI v4): a LeakyReLU-backward tail -- output columns c >= Co are multiplied by
   * (mask[pix * mask_stride + c] > 0 ? 1 : clamp).  The data gradient of DenseBlock conv k+1 finalises the gradient of
   * feature slot k in its last 32 columns; the LeakyReLU backward of that slot (archs.py:90-93) rides in its epilogue
   * instead of being a launch of its own (192 launches per IRN training step). */
  /* IRN architecture (archs.py:74-160): */
  SININN_CONV_LRELU = 6,      /* out = leaky_relu(conv + bias, slope = clamp)        (DenseBlock conv1-4, archs.py:90-93) */
  SININN_CONV_IRN_FWD = 7,    /* out = v * exp(clamp*(2*sigmoid(aux)-1)) + conv + bias  (InvBlockExp, archs.py:152-153;
                                 aux = H(y1) is passed in the mask / mask_stride fields)                     */
  SININN_CONV_IRN_INV = 8,    /* out = (v - (conv + bias)) / exp(clamp*(2*sigmoid(aux)-1))   (archs.py:155-156)        */
  /* data gradient of a subnet's first conv fused with the NEXT coupling tail's backward (saves a launch and the
   * round trip of the intermediate gradient): g = conv + addend[addend_map]; then exactly sininn_coupling_bwd on g:
   *   out  [M][2*Co] = (ds | dt), out_stride = 2*Co;  out2 = dv (stride out2_stride);  v = v (FWD) or y (INV);
   *   sbuf = s [M][Co] (read);  logdet = optional per-sample log-det gradient [B] (read);  N = Co. */
  SININN_CONV_ADD_CBWD_FWD = 9,
  SININN_CONV_ADD_CBWD_INV = 10
};

typedef struct sininn_conv_args {
  const float* in;   int in_stride;  int Cin;      /* Cin % 8 == 0                                         */
  const float* w;    const float* bias; int Np;    /* packed weights [taps][Np][Cin], Np % 16 == 0         */
  int winograd;                                    /* 1: w is the Winograd pack U[16][Np][Cin] (ksize 3, Np % 32 == 0) */
  int B, H, W, ksize;                              /* ksize 1 or 3, zero padding ksize/2                   */
  int mode;
  float* out;        int out_stride; int N;        /* generic modes: N valid output columns                */
  const int* out_map;                              /* coupling modes: y channel c -> out channel (or NULL) */
  const float* v;    int v_stride;                 /* coupling: transformed half                           */
  float* out2;       int out2_stride;              /* coupling: optional compact copy of y                 */
  float* sbuf;                                     /* coupling: optional [M][Co] copy of s (for backward)  */
  float* logdet;                                   /* coupling: optional [B], accumulated with atomics     */
  int Co;            float clamp;                  /* coupling: channels transformed, GLOW clamp           */
  int col_tile;                                    /* coupling: 16 or 32, the (s|t) interleave of the weights */
  const float* mask; int mask_stride;              /* MASK mode                                            */
  const float* addend; int addend_stride; const int* addend_map; /* ADD mode                               */
  unsigned long long* stamp;                       /* optional device words {start, end}: every block folds its entry /
                                                      exit time (wall-clock ticks, sininn_wall_clock_khz) in with atomic
                                                      min / max; initialise to {~0ull >> 1, 0}                       */
  /* ---- mixed-precision path (ABI version 2): bf16 operands on v_mfma_f32_32x32x16_bf16, fp32 accumulate + epilogue ---- */
  int w_bf16;                                      /* w is a bf16 pack [taps][Np][Kp] (sininn_pack_conv_weights_bf16): selects
                                                      the bf16 kernel; the fields below apply only then                   */
  int in_bf16;                                     /* `in` holds bf16 (stride in elements, Cin % 8 == 0); else fp32, converted
                                                      to bf16 (round to nearest even) while it is staged                  */
  int out_bf16;                                    /* `out` holds bf16 (RELU / LINEAR / MASK modes; an fp32-input conv must
                                                      set it); else fp32 through the mode's regular epilogue              */
  int mask_bf16;                                   /* MASK mode: `mask` holds bf16                                        */
  /* channel-group-major tensors [C/8][B*H*W][8] (Winograd kernels; the hidden tensors h / dh of a 3x3 GLOW block, which only
   * the block executor's kernels touch): every halo row of an 8-channel chunk is then one contiguous run.  A value > 0 is the
   * number of floats between channel groups (= B*H*W*8) and selects the layout for that operand: */
  int in_group_stride;                             /* `in` (then in_stride must be 8)                                     */
  int out_group_stride;                            /* `out` (RELU / MASK modes, N == Np, N % 64 == 0)                      */
  int mask_group_stride;                           /* `mask` (MASK mode)                                                  */
} sininn_conv_args;

/* Size limits (checked on the host, refused with sininn_last_error): B*H*W*stride of every operand < 2^31 elements; for
 * winograd = 1 and for sininn_conv_pair_k1 one IMAGE of the input (H*W*in_stride floats) < 2 GB -- the kernels stage through
 * raw buffer loads with 32-bit byte offsets inside the block's image (1280x720 at 1/4 scale with 256 channels is 59 MB). */
int sininn_conv(const sininn_conv_args* args, void* stream);

/* Two chained 1x1 convs of a GLOW subnet in one launch (subnet_conv_1x1, archs.py:15-17, called from FrEIA's
 * GLOWCouplingBlock, archs.py:56-64): `first` produces the 256-channel hidden tensor (mode RELU: h = relu(x W1 + b1), or
 * mode MASK: dh = (dr W2) . [h > 0]), `second` consumes it (any non-IRN mode: the coupling epilogues forward, the skip-add /
 * fused coupling-backward epilogues backward).  The hidden tile stays in LDS between the two GEMMs; it is still written to
 * first->out once (training needs it) unless first->out is NULL (no-grad passes: the hidden tensor never reaches HBM;
 * second->in is then ignored).  Same results as sininn_conv(first) followed by sininn_conv(second) up to fp32 summation
 * order.  Pixel-major operands only; fp32 pairs, or mixed-precision pairs (both convs w_bf16: first fp32 in -> bf16 hidden
 * tensor, second bf16 hidden tensor -> fp32 epilogue; the hidden values are rounded to bf16 exactly once, as in the two-launch
 * path).  sininn_conv_pair_k1_supported returns 1 when the pair's shapes / modes qualify (hidden width 256, first->Cin % 8 == 0
 * and <= 192, second->Np in {16, 32, 48, 64, 96, 192}). */
int sininn_conv_pair_k1_supported(const sininn_conv_args* first, const sininn_conv_args* second);
int sininn_conv_pair_k1(const sininn_conv_args* first, const sininn_conv_args* second, void* stream);

/* The fp32 1x1 subnet + affine coupling + log-det of a GLOW half-coupling as ONE persistent launch (round 4; subnet_conv_1x1,
 * archs.py:15-17, inside FrEIA's GLOWCouplingBlock, archs.py:56-64): the twin of sininn_conv_pair_k1 for the shapes
 * (Cin of conv1, 2 Co) in {(8, 16), (16, 32), (24, 48)} -- level 0 of the SRF network.  A block per CU keeps conv2's pack in LDS and
 * its conv1 fragments in registers over all of its 64-pixel tiles; the hidden tensor is never stored (first->out is ignored: the
 * matching backward, sininn_conv_sub1_bwd, recomputes it).  first: mode RELU; second: mode COUPLE_FWD / COUPLE_INV, described as
 * for sininn_conv_pair_k1.  Same values as the pair up to fp32 summation order (K = 256 is summed in two halves).
 * Mixed precision (both convs w_bf16, described as for the mixed-precision pair): the same entry points run the bf16 twins
 * (csrc/conv_sub1_bf16.hip: bf16 operands, fp32 accumulation / epilogue; h is rounded to bf16 once, bitwise as in the pair). */
int sininn_conv_sub1_fwd_supported(const sininn_conv_args* first, const sininn_conv_args* second);
int sininn_conv_sub1_fwd(const sininn_conv_args* first, const sininn_conv_args* second, void* stream);

/* The whole BACKWARD of a fp32 1x1 conv subnet of a GLOW half-coupling in one persistent launch + one slab reduce (round 4;
 * subnet_conv_1x1, archs.py:15-17, differentiated inside FrEIA's GLOWCouplingBlock, archs.py:56-64):
 *   h = relu(x W1^T + b1) is RECOMPUTED from the subnet's input (the forward pass need not store it: pass first->out == NULL to
 *   sininn_conv_pair_k1), dh = (dr W2) . [h > 0] never leaves the chip, dx = dh W1 goes through d1's epilogue (ADD / ADD_CBWD_*),
 *   and gw2 / gb2 / gw1 / gb1 (OIHW, +=; any of them may be NULL) are summed from per-block slabs in a fixed order.
 * recompute: {in = x, in_stride, Cin, w = forward pack of conv1 [256][Cin], bias, Np = 256, B, H, W, ksize = 1}; d2 / d1: the two
 * data-gradient convs exactly as for sininn_conv_pair_k1 (d2: in = dr [.. 2 Co], w = data-gradient pack of conv2; d1: Cin = 256,
 * w = data-gradient pack of conv1, Np = pad16(Cin of conv1), the epilogue fields); d2->mask / d2->out / d1->in are ignored.
 * Shapes served: (Cin of conv1, 2 Co) in {(8, 16), (16, 32), (24, 48)} -- sininn_conv_sub1_bwd_workspace_bytes returns 0 for any
 * other.  dx / the fused coupling backward are bitwise what sininn_conv_pair_k1(d2, d1) produces from the stored h; the weight
 * gradients agree with sininn_wgrad up to fp32 summation order.  no_dx != 0: the data gradient of conv1 is not needed.
 * Mixed precision: recompute / d2 / d1 with w_bf16 = 1 (bf16 packs; x, dr, dx and the gradients stay fp32) select the bf16 twin. */
size_t sininn_conv_sub1_bwd_workspace_bytes(int cin, int co);
int sininn_conv_sub1_bwd(const sininn_conv_args* recompute, const sininn_conv_args* d2, const sininn_conv_args* d1, int no_dx,
                         float* gw2, float* gb2, float* gw1, float* gb1, void* workspace, size_t workspace_bytes, void* stream);

/* The 3x3 twin for passes that keep nothing for a backward (ABI v4): the WHOLE 3x3 conv subnet (subnet_conv, archs.py:11-13) +
 * affine coupling + log-det of a GLOW half-coupling (archs.py:56-64) in one launch on the mixed-precision path.  `first`:
 * ksize 3, bf16 weights, fp32 input, mode RELU, 256 output channels, out == NULL (the hidden tile lives in LDS only);
 * `second`: ksize 3, bf16 weights, Cin 256, mode COUPLE_FWD / COUPLE_INV (same kernel for both directions), Np in {16, 32,
 * 48, 64, 96, 192}.  Same values as the two sininn_conv launches with a bf16 hidden tensor (one rounding of h to bf16) up to
 * fp32 summation order.  sininn_glow_forward dispatches it for dtype == 1, ksize == 3, no_save != 0 when SININN_SUB3=1 is set in the
 * environment (not the default: on MI355X it measures 1.8 - 2.7x slower than the two launches it replaces). */
int sininn_conv_sub3_supported(const sininn_conv_args* first, const sininn_conv_args* second);
int sininn_conv_sub3(const sininn_conv_args* first, const sininn_conv_args* second, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Weight gradient: dW[n][c][tap] += sum_pixels dout[pix][n] * in[pix+tap][c], db[n] += sum dout.
 * Split over pixel ranges into slabs in `workspace`, then reduced into OIHW gradients (+=).
 * sininn_wgrad_workspace_bytes gives the slab size for a shape.
 * ---------------------------------------------------------------------------------------------- */
size_t sininn_wgrad_workspace_bytes(int N, int Cin, int ksize, int B, int H, int W);
int sininn_wgrad(const float* in, int in_stride, int Cin, const float* dout, int dout_stride, int N,
                 int B, int H, int W, int ksize, float* gw_oihw, float* gbias,
                 void* workspace, size_t workspace_bytes, void* stream);

/* Grouped form: the weight gradients of up to 8 convs that see the same pixels (B,H,W) and kernel size -- the four
 * convs of one GLOW block -- in two launches (gradient kernel + ordered slab reduce).  The convs share the launch
 * grid, so each needs far fewer split-K slabs than on its own (3-6x less slab traffic); results are bitwise
 * reproducible like sininn_wgrad's.  3x3: Winograd F(3x3,2x2) kernel, 64 x 32 output tiles; 1x1: 32 x 32 tiles. */
typedef struct sininn_wgrad_item {
  size_t struct_bytes;                             /* ABI v4: must be sizeof(sininn_wgrad_item).  The descriptor has grown
                                                      optional fields twice; a caller built against another revision, or one
                                                      that fills a stack item field by field and misses a new field, is
                                                      refused instead of launching with garbage ("the round-2 abort") */
  const float* in;   int in_stride;   int Cin;     /* conv input  [M][in_stride], Cin % 4 == 0                      */
  const float* dout; int dout_stride; int N;       /* output gradient [M][dout_stride], N % 4 == 0                  */
  float* gw; float* gb;                            /* OIHW weight gradient (+=), bias gradient (+=, may be NULL)    */
  int in_bf16, dout_bf16;                          /* mixed-precision path: the operand is stored as bf16 (pointer cast,
                                                      stride in elements); the gradient accumulates in fp32          */
  int in_group_stride, dout_group_stride;          /* > 0: the operand is channel-group-major [C/8][B*H*W][8] (fp32, 3x3
                                                      Winograd kernels); value = floats between channel groups         */
  int gap_begin, gap_len;                          /* gap_len > 0: input channels [gap_begin, gap_begin + gap_len) are padding
                                                      that the weight does not have (gw rows hold Cin - gap_len channels)  */
} sininn_wgrad_item;
size_t sininn_wgrad_group_workspace_bytes(const sininn_wgrad_item* items, int n, int B, int H, int W, int ksize);
int sininn_wgrad_group(const sininn_wgrad_item* items, int n, int B, int H, int W, int ksize,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Backward of the coupling tail (the elementwise part of GLOWCouplingBlock).
 *   inverse==0: y = e(s) v + t      dv = dy e ; dt = dy  ; ds = (dy v e + gld) L'(s)
 *   inverse==1: y = (v - t)/e(s)    dv = dy/e ; dt = -dy/e ; ds = -(dy y + gld) L'(s)
 * dy[m][c] is read from dy[m*dy_stride + (dy_map ? dy_map[c] : c)] ; `vy` is v (inverse==0) or y
 * (inverse==1) read through vy_map likewise.  dr is [M][2*Co] = (ds | dt).  gld is the optional
 * per-sample gradient of the block's log-det ([B], may be NULL).
 * ---------------------------------------------------------------------------------------------- */
int sininn_coupling_bwd(const float* dy, int dy_stride, const int* dy_map,
                        const float* vy, int vy_stride, const int* vy_map,
                        const float* s, const float* gld, int B, int HW, int Co, float clamp,
                        int inverse, float* dr, float* dv, int dv_stride, void* stream);

/* ------------------------------------------------------------------------------------------------
 * IRN elementwise pieces (archs.py:135-199).
 * haar: HaarDownsampling.forward / rev (archs.py:183-199): 2x2 Haar analysis (/4) with the band-major channel
 *   regrouping out[:, k*C + c] = band k of channel c, or the synthesis (conv_transpose2d, no /4).  Element strides.
 * lrelu_bwd: g[m][j] *= (f[m][j] > 0 ? 1 : slope)   (gradient through LeakyReLU, in place on a channel slot).
 * irn_coupling_bwd: backward of y = v*exp(s)+G (inverse==0) or y = (v-G)/exp(s) (inverse==1), s = clamp*(2*sigmoid(h)-1):
 *   dG, dh [M][Co] compact, dv [M][Co] at dv_stride.  vy = v (inverse==0) or y (inverse==1).
 * ---------------------------------------------------------------------------------------------- */
int sininn_haar(const float* in, const int64_t in_strides[4], float* out, const int64_t out_strides[4],
                int B, int C, int H, int W, int inverse, void* stream);
int sininn_lrelu_bwd(float* g, int g_stride, const float* f, int f_stride, int64_t M, int n, float slope, void* stream);
/* irn_tail: the InvBlockExp tail on its own (archs.py:152-156): out = v * exp(s) + g (inverse == 0) or (v - g) / exp(s)
 *   (inverse == 1), s = clamp * (2 sigmoid(h) - 1); h, g [M][Co] compact, v / out at their pixel strides.  Its backward is
 *   sininn_irn_coupling_bwd.  Lets the H and G DenseBlocks of a block run on two streams (ABI v4). */
int sininn_irn_tail(const float* v, int v_stride, const float* h, const float* g, int64_t M, int Co, float clamp, int inverse,
                    float* out, int out_stride, void* stream);
int sininn_irn_coupling_bwd(const float* dy, int dy_stride, const float* vy, int vy_stride, const float* hval,
                            int64_t M, int Co, float clamp, int inverse, float* dG, float* dh, float* dv,
                            int dv_stride, void* stream);

/* ------------------------------------------------------------------------------------------------
 * One DenseBlock of the IRN architecture per call (archs.py:74-133, with the InvBlockExp tail of archs.py:135-160 as the
 * epilogue of its fifth conv): host-side launch sequence, like sininn_glow_forward / _backward for the SRF path.  The five
 * dense-connected 3x3 convs write their 32-channel outputs into slots of ONE feature buffer buf [M][cinp + 128]
 * (cinp = cin rounded up to 8; the torch.cat chain costs nothing); weights are packed for that padded channel order.
 *   mode 0: out = conv5(...)                       1: out = aux1 + conv5(...)          (y1 = x1 + F(x2))
 *   mode 2: out = aux1 * exp(s) + conv5(...)       3: out = (aux1 - conv5(...)) / exp(s),   s = clamp * (2 sigmoid(aux2) - 1)
 * backward: dF [M][cinp + 128] (on return its first cin channels are d loss / d x), dv / dh (modes 2, 3: gradients w.r.t.
 * aux1 and aux2; mode 1: the gradient w.r.t. aux1 is dout itself), OIHW weight / bias gradients (+=) of the UNPADDED convs.
 * The five weight gradients run as one grouped launch pair on wgrad_stream.
 * ---------------------------------------------------------------------------------------------- */
typedef struct sininn_dense_args {
  size_t struct_bytes;                             /* ABI v4: must be sizeof(sininn_dense_args)                            */
  int B, H, W, cin, cout, mode, winograd;
  float clamp;
  const float* x; int x_stride;
  const float* aux1; int aux1_stride;
  const float* aux2;                               /* [M][cout] */
  float* buf; float* out;                          /* [M][cinp + 128], [M][cout] */
  const float* w_fwd[5]; const float* b_fwd[5]; const float* w_dgrad[5];
  /* backward only */
  const float* dout;                               /* [M][cout] */
  float* dF;                                       /* [M][cinp + 128] */
  float* dD;                                       /* [M][cout rounded up to 8] scratch (modes 2, 3 and cout % 8 != 0)   */
  float* dh; float* dv;                            /* modes 2, 3: [M][cout] each                                          */
  float* gw[5]; float* gb[5];
  void* workspace; size_t workspace_bytes;         /* sininn_dense_workspace_bytes                                        */
  /* ABI v4: extents (in floats) of every buffer the executor writes or reads at a size it derives itself; a call whose
   * buffers are smaller than the launch sequence needs is refused (sininn_last_error) before anything is launched.
   * M = B*H*W, bw = pad8(cin) + 128.  Required: buf >= M*bw, out >= M*cout; backward: dout >= M*cout, dF >= M*bw,
   * dD >= M*pad8(cout) when used, dh / dv >= M*cout (modes 2, 3), aux2 >= M*cout (modes 2, 3).                         */
  size_t buf_floats, out_floats, aux2_floats, dout_floats, dF_floats, dD_floats, dh_floats, dv_floats;
} sininn_dense_args;
size_t sininn_dense_workspace_bytes(int B, int H, int W, int cin, int cout);
int sininn_dense_forward(const sininn_dense_args* args, void* stream);
int sininn_dense_backward(const sininn_dense_args* args, void* stream, void* wgrad_stream);

/* ------------------------------------------------------------------------------------------------
 * One GLOW coupling block per call (FrEIA GLOWCouplingBlock.forward / its autograd, SURVEY Appendix A;
 * wired at archs.py:61-64): host-side launch sequence of the kernels above.
 *   forward : (rev==0)  r2=s2(x2); y1=e(s2)*x1+t2; r1=s1(y1); y2=e(s1)*x2+t1      logdet += sum log_e
 *             (rev==1)  r1=s1(x1); y2=(x2-t1)/e(s1); r2=s2(y2); y1=(x1-t2)/e(s2)  logdet -= sum log_e
 *             out channel c is stored at dst_map[c] (folds the following / preceding PermuteRandom).
 *   backward: dx, and (+=) the OIHW weight / bias gradients of the four convs; weight-gradient kernels are
 *             queued on wgrad_stream (ordered after their inputs with events), everything else on stream.
 * `saved` (sininn_glow_saved_floats floats) is written by forward and read by backward: hidden activations of
 * both subnets, both s tensors, the first half's compact output.  `scratch` is backward-only.
 * ---------------------------------------------------------------------------------------------- */
typedef struct sininn_subnet {
  const float* w1; const float* b1;      /* conv1 packed [taps][256][Cin] + bias[256]                          */
  const float* w2; const float* b2;      /* conv2 packed [taps][2*Co][256] (s|t interleaved), bias packed alike */
  const float* w1_dgrad;                 /* [taps][pad16(Cin)][256]  (backward only)                           */
  const float* w2_dgrad;                 /* [taps][256][2*Co]        (backward only)                           */
  float* gw1; float* gb1; float* gw2; float* gb2;   /* OIHW gradient accumulators (NULL: skip)                 */
  int winograd;                          /* bit 0: w1, bit 1: w2, bit 2: w1_dgrad (rows padded to 32), bit 3: w2_dgrad
                                            hold Winograd packs (sininn_pack_winograd) instead of tap-major packs   */
} sininn_subnet;

typedef struct sininn_glow_args {
  int B, H, W, C, ksize, rev; float clamp;
  const float* x; float* out; const int* dst_map; float* logdet;
  sininn_subnet s1, s2;
  float* saved;
  void* scratch; size_t scratch_bytes;
  const float* dout; const float* gld; float* dx;
  int skip_dx;                           /* backward: the caller does not need dx (first block of a pass): the last data-
                                            gradient conv is skipped; dx must still be a valid buffer (partly written) */
  int dtype;                             /* 0: fp32 subnets (f32 MFMA, Winograd for 3x3).  1: mixed precision -- the conv
                                            subnets run on bf16 MFMA with fp32 accumulation (packs from
                                            sininn_pack_conv_weights_bf16, hidden tensors stored as bf16), the flow tensors,
                                            the coupling arithmetic, log-det and all gradients w.r.t. parameters stay fp32 */
  int no_save;                           /* forward: nothing will be differentiated (torch.no_grad passes: validation /
                                            inference, lit_wrapper.py:79-128) -- the saved s and, where the subnet runs as one
                                            launch (1x1: sininn_conv_pair_k1; 3x3 with dtype 1: sininn_conv_sub3), the hidden
                                            tensor are not written to HBM                                                  */
} sininn_glow_args;

/* Live timing for bench.py: between begin and end, every forward 3x3 coupling conv (conv2 + affine epilogue) of the
 * level whose image height is `level_height` is (a) bracketed by HIP events on its launch stream and (b), when `stamps`
 * (device, 2 words per launch, initialised to {~0ull >> 1, 0}) is given, stamped from inside the kernel (see
 * sininn_conv_args.stamp; launch i uses words 2i, 2i+1, launches beyond max_launches are not stamped).  With several
 * streams in flight the event bracket also contains the time the launch waits for the GPU behind other streams'
 * kernels; the stamps are the kernel's own execution window (what a kernel trace reports).  end() synchronises on the
 * events and returns the number of launches timed and their summed event duration. */
void sininn_profile_begin(int level_height, unsigned long long* stamps, int max_launches);
int sininn_profile_end(int* count, float* total_ms);
int sininn_wall_clock_khz(void);   /* tick rate of the stamps */

/* Per-class timing of the block executor for bench.py's roofline.classes: between begin and end every launch of
 * sininn_glow_forward / _backward is bracketed by HIP events on its stream.  Classes (index = class + 6 * (ksize == 1)):
 * 0 conv1 forward (+ReLU), 1 conv2 forward + coupling + log-det, 2 data gradient of conv2 (+ReLU mask), 3 data gradient of
 * conv1 (+ skip gradient, + fused coupling backward), 4 weight gradients (+ slab reduce), 5 coupling backward tail.
 * end() synchronises and fills, per class, the summed event time (ms), the summed ALGORITHMIC FLOPs (2 M k^2 Cin N) and
 * the launch count (n = 12 entries).  The bracket is the kernel's own duration only when everything runs on one stream. */
#define SININN_PROFILE_CLASSES 12
void sininn_profile_classes_begin(void);
int sininn_profile_classes_end(int n, double* ms, double* flops, int* launches);
/* algorithmic HBM bytes per class, summed over the launches of the last sininn_profile_classes_end (the fused 1x1 launches report
 * them: x / dr / side inputs / outputs / slabs; 0 for a class whose launches do not) */
int sininn_profile_classes_bytes(int n, double* bytes);

size_t sininn_glow_saved_floats(int B, int H, int W, int C);
size_t sininn_glow_saved_floats_dtype(int B, int H, int W, int C, int dtype);   /* dtype 1: bf16 hidden tensors (half the floats) */
size_t sininn_glow_scratch_bytes(int B, int H, int W, int C, int ksize);                       /* dtype 0 */
size_t sininn_glow_scratch_bytes_dtype(int B, int H, int W, int C, int ksize, int dtype);   /* dtype 1 needs no slabs for the fused 1x1 backward */
int sininn_glow_forward(const sininn_glow_args* args, void* stream);
int sininn_glow_backward(const sininn_glow_args* args, void* stream, void* wgrad_stream);
/* Parity tooling (ABI v4): the ReLU gates the forward pass of a block took, gates[m][j] = (hidden[m][j] > 0) as bytes
 * [B*H*W][256], for the subnet executed first (which = 0: s2 when rev == 0, s1 when rev == 1) or second (which = 1), decoded
 * from `saved` in whatever layout / dtype the executor stored it.  args: the descriptor of the forward call (B, H, W, C, ksize,
 * rev, dtype, s1 / s2 .winograd, saved).  With these gates forced, a float64 evaluation of the network is a smooth function
 * and can be compared with the HIP path without the discrete noise of units that sit within rounding distance of 0. */
int sininn_glow_hidden_gates(const sininn_glow_args* args, int which, uint8_t* gates, void* stream);
/* 1 when the block executor may keep the 256-channel hidden tensors of a fp32 3x3 subnet channel-group-major ([256/8][M][8]) at
 * this shape, 0 when it takes the row-major layout because the 32-bit staging offsets of the kernels reading the tensor could not
 * address it (the same bound sininn_conv / sininn_wgrad_group check and refuse).  Host-only; no launch. */
int sininn_glow_group_major_fits(int B, int H, int W);

/* ------------------------------------------------------------------------------------------------
 * Index maps: FrEIA IRevNetDownsampling (archs.py:28-31,35-38) and PermuteRandom (archs.py:65-68),
 * plus NCHW<->NHWC import/export, as ONE strided gather:
 *   levels==0 : out[b,y,x,cm(c)] = in[b,c,y,x]
 *   squeeze (inverse==0), per level: out[b,(hb*2+wb)*C+c,i,j] = in[b,c,2i+hb,2j+wb]
 *   unsqueeze (inverse==1) is the exact inverse.
 * `levels` squeezes are composed in one pass.  Tensors are described by element strides
 * (sb, sc, sh, sw) so either side may be NCHW or NHWC.  chan_map (device, may be NULL) is applied on
 * the C-channel side that is being WRITTEN when map_on_out!=0, else on the side being READ.
 * B,C,H,W describe the un-squeezed (fine) tensor.
 * ---------------------------------------------------------------------------------------------- */
int sininn_squeeze(const float* in, const int64_t in_strides[4], float* out, const int64_t out_strides[4],
                   int B, int C, int H, int W, int levels, int inverse,
                   const int* chan_map, int map_on_out, void* stream);

/* The same index maps between two DENSE pixel-major tensors (fine [B][H][W][C], coarse [B][H/2^l][W/2^l][C*4^l]), in
 * gather form: four consecutive output floats per thread, 16-byte stores.  fine_map (device, may be NULL) acts on the
 * fine tensor's channel index:  forward  coarse[.., q*C + c] = fine[.., fine_map[c]];
 *                               inverse  fine[.., c] = coarse[.., q*C + fine_map[c]].
 * levels == 0: out[.., c] = in[.., fine_map[c]].  Needs B*C*H*W % 4 == 0 and (C*4^levels) % 4 == 0. */
int sininn_squeeze_rows(const float* in, float* out, int B, int C, int H, int W, int levels, int inverse,
                        const int* fine_map, void* stream);

/* out[m][j] = in[m][idx[j]] on pixel-major tensors (PermuteRandom.forward: x[:, perm]). */
int sininn_permute_channels(const float* in, int in_stride, float* out, int out_stride,
                            int64_t M, int C, const int* idx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Losses (loss.py).  Strided 4-D views (element strides sb,sc,sh,sw); sums are written to fp32
 * device scalars, the caller divides (torch.mean) -- keeps the kernels stream-async.
 * ---------------------------------------------------------------------------------------------- */
/* sum (x-y)^2 -> out[0]  (loss.py:3-5).  y may be NULL => sum x^2 (loss.py:38-39).  out must be zeroed. */
int sininn_sqdiff_sum(const float* x, const int64_t xs[4], const float* y, const int64_t ys[4],
                      int B, int C, int H, int W, float* out, void* stream);
/* gx = scale[0]*gscale * (x-y) (and gy = -gx when gy!=NULL); `scale` is a device scalar (upstream grad). */
int sininn_sqdiff_bwd(const float* x, const int64_t xs[4], const float* y, const int64_t ys[4],
                      int B, int C, int H, int W, const float* scale, float gscale,
                      float* gx, const int64_t gxs[4], float* gy, const int64_t gys[4], void* stream);
/* Gram matrices for loss.mmd (loss.py:15-18): g[0]=x x^T, g[1]=y y^T, g[2]=x y^T, each [B][B].  g holds
 * (1 + SININN_MMD_SLOTS) * 3*B*B floats ZEROED by the caller: the result followed by the partial-sum slots. */
#define SININN_MMD_SLOTS 16
int sininn_mmd_gram(const float* x, const int64_t xs[4], const float* y, const int64_t ys[4],
                    int B, int C, int H, int W, float* g, void* stream);
/* loss.py:20-36 on the three Grams -> out[0] (mean) and coef[3][B][B] = dLoss/dGram (for backward). */
int sininn_mmd_finish(const float* g, int B, int rev, float* out, float* coef, void* stream);
/* gx[b,:] = scale*( sum_j (coefXX[b][j]+coefXX[j][b]) x[j,:] + coefXY[b][j] y[j,:] ), gy likewise. */
int sininn_mmd_bwd(const float* x, const int64_t xs[4], const float* y, const int64_t ys[4],
                   int B, int C, int H, int W, const float* coef, const float* scale,
                   float* gx, const int64_t gxs[4], float* gy, const int64_t gys[4], void* stream);

/* ------------------------------------------------------------------------------------------------
 * Warps.
 * affine_warp: F.affine_grid(theta, align_corners=False) + F.grid_sample(bilinear, zeros,
 *   align_corners=False) fused -- what kornia.warp_affine evaluates for tcr.py:43 once theta (the
 *   inverted, normalised 2x3 matrix, [B][2][3]) is known.  Optional fused MSE against `ref`
 *   (sum (warp-ref)^2 -> sse[0]).  bwd is the image gradient (atomic scatter).
 * flow_warp_l1: Resample2d.forward + the photometric metric
 *   (video-interpolation/my_utils/resample2d.py:57-72, video-interpolation/trainer.py:61-62):
 *   warped = grid_sample(img, (coords+flow)/(W-1,H-1)*2-1) ; metric = mean_c |target - warped|.
 * ---------------------------------------------------------------------------------------------- */
int sininn_affine_warp(const float* img, const int64_t is[4], const float* theta, int B, int C, int H, int W,
                       float* out, const int64_t os[4], const float* ref, const int64_t rs[4], float* sse,
                       void* stream);
int sininn_affine_warp_bwd(const float* gout, const int64_t gs[4], const float* theta, int B, int C, int H, int W,
                           float* gimg, const int64_t gis[4], void* stream);
int sininn_flow_warp_l1(const float* img, const float* flow, const float* target, int B, int C, int H, int W,
                        float* warped, float* metric, void* stream);
int sininn_flow_warp_l1_bwd(const float* img, const float* flow, const float* target, const float* warped,
                            const float* gwarped, const float* gmetric, int B, int C, int H, int W,
                            float* gimg, float* gflow, void* stream);
/* The same two kernels in the arithmetic BASELINE configs[3] names ("pair_flow warp + INN at 512x512, bf16"): img, target,
 * warped and gwarped are bf16 in HBM (half the bytes of this HBM-bound pair); the bilinear weights, the interpolation, the
 * metric (taken on the ROUNDED warped value, i.e. on what the consumer reads back) and every gradient accumulator
 * (gimg, gflow) are fp32, flow and metric are fp32 tensors (ABI v4). */
int sininn_flow_warp_l1_bf16(const void* img, const float* flow, const void* target, int B, int C, int H, int W,
                             void* warped, float* metric, void* stream);
int sininn_flow_warp_l1_bwd_bf16(const void* img, const float* flow, const void* target, const void* warped,
                                 const void* gwarped, const float* gmetric, int B, int C, int H, int W,
                                 float* gimg, float* gflow, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Frame-window sampler (data.py:31-45 + 112-115 on an HBM-resident uint8 clip):
 *   hr[n] = hr_clip[idx[n]] / 255 as (3,H,W) planar or pixel-major, lr[n] = concat of the 2*win+1 LR
 *   frames idx[n]-win..idx[n]+win on the channel axis / 255.
 * hr_clip (T,H,W,3) u8, lr_clip (T,h,w,4) u8, idx device int32 [n].
 * ---------------------------------------------------------------------------------------------- */
int sininn_sample_windows(const uint8_t* hr_clip, const uint8_t* lr_clip, const int* idx, int n,
                          int T, int H, int W, int h, int w, int win,
                          float* hr_out, const int64_t hs[4], float* lr_out, const int64_t ls[4], void* stream);

/* Frame-pair sampler of the flow path (video-interpolation/trainer.py:49-62 consumes (frame1, frame2) of one clip as planar
 * (B,3,H,W) tensors): out0[s] = clip[idx[s]] / 255, out1[s] = clip[idx[s] + gap] / 255 (frame index clamped to the clip), planar,
 * fp32 (bf16 == 0) or bf16 (the arithmetic of BASELINE configs[3]).  clip (T,H,W,3) u8, H*W % 4 == 0.  ABI v4. */
int sininn_sample_pairs(const uint8_t* hr_clip, const int* idx, int n, int T, int H, int W, int gap,
                        void* out0, void* out1, int bf16, void* stream);

/* ------------------------------------------------------------------------------------------------
 * On-device LR synthesis (datasets/prepare.py:35-82,147-165): RGGB sampling + scale x scale binning per Bayer plane
 * with the reference's float64 arithmetic and uint8 truncation.  hr (T,H,W,3) u8 -> lr (T,H/(2s),W/(2s),4) u8.
 * ---------------------------------------------------------------------------------------------- */
int sininn_bayer_bin(const uint8_t* hr, uint8_t* lr, int T, int H, int W, int scale, int reduce_sum, void* stream);
/* Demosaiced LR preview (datasets/prepare.py:103-119,158,163-165): the unquantised binned planes re-packed as an RGGB
 * mosaic and bilinearly demosaiced (colour_demosaicing 0.1.6, scipy 'reflect' boundary), clipped, quantised:
 * rgb [T][H/scale][W/scale][3] uint8. */
int sininn_bayer_demosaic(const uint8_t* hr, uint8_t* rgb, int T, int H, int W, int scale, int reduce_sum, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Output pixels of the inference path (SingleVideoINN.infer, lit_wrapper.py:91-128): float frames `in` (logical
 * (B,C,H,W), element strides) -> uint8 images out[B][H][W][C] on the device, so only bytes cross PCIe.
 *   wrap == 0: clamp(x,0,1)*255 truncated;  wrap != 0: (uint8)(int)(x*255) = torchvision ToPILImage's
 *   pic.mul(255).byte() with its wrap-around on out-of-range values (what the reference does, lit_wrapper.py:94,120).
 * ---------------------------------------------------------------------------------------------- */
int sininn_frames_to_u8(const float* in, const int64_t in_strides[4], uint8_t* out, int B, int C, int H, int W, int wrap,
                        void* stream);

/* ------------------------------------------------------------------------------------------------
 * Adam exactly as torch.optim.Adam (lit_wrapper.py:131-138: L2 weight decay, not AdamW):
 *   g = grad*grad_scale + wd*p ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
 *   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * ---------------------------------------------------------------------------------------------- */
int sininn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Photometric-loss operators of the flow trainer.  NCHW contiguous fp32.
 *   softsplat      video-interpolation/my_utils/softsplat.py:8-177: out (ZEROED by the caller) += forward splat of `in`
 *                  along `flow`; _bwd returns d/d in and / or d/d flow (either may be NULL) for an upstream gout.
 *   occlusion_wang video-interpolation/my_utils/occlusions.py:29-104: corr (ZEROED, [B][H][W]) = range map of flow21,
 *                  mask (optional, [B][1][H][W]) = 1 - (corr <= thresh).
 *   census         video-interpolation/my_utils/loss.py:30-72 (CensusLoss.forward(im1, im2, mask)), 3-channel images,
 *                  mask [B][mask_channels][H][W] with 1 (pair_flow.py) or 3 (trainer.py:64) channels, max_distance 1..4.  acc (ZEROED, SININN_CENSUS_ACC_FLOATS floats: two sums + 64
 *                  partial slots) receives {sum of distances, sum(mask)} in its first two words
 *                  and is the saved state for _bwd; out[0] = the loss.  _bwd: g1 / g2 = gscale[0] * d loss / d im1 / im2
 *                  (no gradient w.r.t. the mask, which the trainer builds from comparisons).
 * ---------------------------------------------------------------------------------------------- */
int sininn_softsplat(const float* in, const float* flow, int B, int C, int H, int W, float* out, void* stream);
int sininn_softsplat_bwd(const float* in, const float* flow, const float* gout, int B, int C, int H, int W,
                         float* gin, float* gflow, void* stream);
int sininn_occlusion_wang(const float* flow21, int B, int H, int W, float thresh, float* corr, float* mask, void* stream);
#define SININN_CENSUS_ACC_FLOATS 130
/* occlusion_brox (occlusions.py:111-118): mask [B][1][H][W] bytes (1 = inconsistent) from the forward flow and the backward
 * flow already warped by it (sininn_flow_warp_l1(bw, fw)). */
int sininn_occlusion_brox(const float* fw, const float* warped_bw, int B, int H, int W, uint8_t* mask, void* stream);
int sininn_census(const float* im1, const float* im2, const float* mask, int mask_channels, int B, int H, int W,
                  int max_distance, float weight, float* acc, float* out, void* stream);
int sininn_census_bwd(const float* im1, const float* im2, const float* mask, int mask_channels, int B, int H, int W,
                      int max_distance, float weight, const float* acc, const float* gscale, float* g1, float* g2,
                      void* stream);
/* L1Loss (loss.py:17-27): l1_loss(im1*mask, im2*mask) / sum(mask) * numel(mask) * weight; mask channels 1 or C; acc as for
 * the census loss (ZEROED, SININN_CENSUS_ACC_FLOATS).  _bwd: g1 = -g2 = gscale[0] * d loss / d im1. */
int sininn_masked_l1(const float* im1, const float* im2, const float* mask, int mask_channels, int B, int C, int H, int W,
                     float weight, float* acc, float* out, void* stream);
int sininn_masked_l1_bwd(const float* im1, const float* im2, const float* mask, int mask_channels, int B, int C, int H,
                         int W, float weight, const float* acc, const float* gscale, float* g1, float* g2, void* stream);
/* SSIMLoss (loss.py:75-103): (2 md + 1)^2 unpadded average pooling, md 1..2; acc / gradients as for sininn_masked_l1. */
int sininn_ssim(const float* im1, const float* im2, const float* mask, int mask_channels, int B, int C, int H, int W, int md,
                float weight, float* acc, float* out, void* stream);
int sininn_ssim_bwd(const float* im1, const float* im2, const float* mask, int mask_channels, int B, int C, int H, int W,
                    int md, float weight, const float* acc, const float* gscale, float* g1, float* g2, void* stream);
/* BilateralSmooth (loss.py:106-132): edge-aware smoothness of flow [B][2][H][W] guided by img [B][C][H][W];
 * order 1 | 2, gauss != 0: squared ('gauss') else absolute ('exp') image differences scaled by edge_constant.
 * acc (ZEROED, SININN_CENSUS_ACC_FLOATS) is scratch.  _bwd: gflow = gscale[0] * d loss / d flow (img gets none). */
int sininn_bilateral_smooth(const float* img, const float* flow, int B, int C, int H, int W, int order, int gauss,
                            float edge_constant, float weight, float* acc, float* out, void* stream);
int sininn_bilateral_smooth_bwd(const float* img, const float* flow, int B, int C, int H, int W, int order, int gauss,
                                float edge_constant, float weight, const float* gscale, float* gflow, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SININN_H */

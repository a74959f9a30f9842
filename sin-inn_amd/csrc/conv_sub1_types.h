// Shapes, descriptors and the slab layout shared by the persistent fused 1x1-subnet kernels: fp32 (conv_sub1.hip) and mixed
// precision (conv_sub1_bf16.hip).  Both write the same slab, so one reduce kernel (sub1_reduce_kernel) serves both.
#pragma once
#include "conv_mfma_impl.h"

namespace sininn {

constexpr int S1_HID = 256;            // hidden channels (SININN_HIDDEN)
constexpr int S1_HS = S1_HID + 4;      // floats per pixel row of the hidden tile in LDS
constexpr int S1_P = 64;               // pixels per tile (4 x 16)
constexpr int S1_NTHR = 512;           // 8 waves: wave w owns hidden columns [32 w, 32 w + 32)
constexpr int S1_MAX_BLOCKS = 256;     // persistent blocks == slabs (one per CU: 122 KB of LDS)

struct Sub1Dev {
  ConvDev r;        // recompute: in = x (the subnet's input), w = W1 forward pack [256][K1], bias = b1
  ConvDev a;        // data gradient of conv2: in = dr [.. K2], w = W2 data-gradient pack [256][K2]
  ConvDev b;        // data gradient of conv1: w = W1 data-gradient pack [pad16(K1)][256] + the epilogue descriptor
  float* slab;      // [blocks][slab_floats]
  int ntiles, no_dx;
};

template <int K1, int K2>
struct Sub1Shape {
  static constexpr int K1R = (K1 + 15) / 16 * 16;          // stage R walks K in steps of 16
  static constexpr int NU1 = (K1 + 16) / 16;               // 16-column tiles of [x | 1] (weight gradient of conv1 + db1)
  static constexpr int XD = K1R > 16 * NU1 ? K1R : 16 * NU1;
  static constexpr int XS = XD + 4;                         // floats per pixel row of the x tile
  static constexpr int DS = K2 + 4;                         // ... of the dr tile
  static constexpr int NU2 = K2 / 16;
  static constexpr int NP1 = K1R;                           // columns of the data gradient of conv1 (pad16)
  static constexpr int NT2 = NP1 / 16;
  static constexpr int W1S = 16 * NU1;                      // slab row of dW1: [K1 channels | db1 | zero pad]
  static constexpr int SLAB = K2 * S1_HID + S1_HID * W1S + 64;   // dW2 [K2][256] | dW1^T [W1S][256] (row K1 = db1) | db2 [64]
  static constexpr size_t LDS = (size_t)(S1_P * S1_HS + 2 * (S1_P * DS + S1_P * XS) + NP1 * S1_HS + S1_P * (NP1 + 4)) * sizeof(float);
};

static inline bool sub1_shape_ok(int k1, int k2) { return (k1 == 8 && k2 == 16) || (k1 == 16 && k2 == 32) || (k1 == 24 && k2 == 48); }

}  // namespace sininn

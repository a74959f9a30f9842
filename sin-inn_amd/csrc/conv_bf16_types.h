// Types shared by the bf16 conv kernels (conv_bf16.hip) and the fused bf16 1x1 pair (conv_pair_bf16.hip).
#pragma once
#include "conv_mfma_impl.h"

namespace sininn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvDevB {
  ConvDev c;                 // fp32-side description (epilogue operands, shapes); c.in / c.w / c.out are unused when the
  const void* in;            // typed pointers below replace them
  const __bf16* w;           // [taps][Np][Kp] bf16
  __bf16* out_b;             // bf16 output (RELU / LINEAR / MASK modes with out_bf16)
  const __bf16* mask_b;      // bf16 ReLU mask source (MASK mode with out_bf16)
  int Kp;                    // channels of the weight pack (Cin rounded up to a multiple of 16)
  int in_bf16, out_bf16;
};


int conv_bf16_prepare(const sininn_conv_args* a, ConvDevB& q);

}  // namespace sininn

#include "conv_mfma_impl.h"
namespace sininn {
int conv_dispatch_k1(ConvDev& d, hipStream_t st, int force_cfg) { return dispatch<1>(d, st, force_cfg); }
}

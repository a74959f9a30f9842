#include "conv32_impl.h"
namespace sininn {
int conv_dispatch_k1(ConvDev& d, hipStream_t st, int force_cfg) { return dispatch<1>(d, st, force_cfg); }
int conv32_dispatch_k1(ConvDev& d, hipStream_t st, int force_cfg, bool must) { return dispatch32<1>(d, st, force_cfg, must); }
}

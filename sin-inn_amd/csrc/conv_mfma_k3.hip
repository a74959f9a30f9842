#include "conv32_impl.h"
#include "wino_impl.h"
namespace sininn {
int conv_dispatch_k3(ConvDev& d, hipStream_t st, int force_cfg) { return dispatch<3>(d, st, force_cfg); }
int wino_dispatch_k3(ConvDev& d, hipStream_t st, int cg2) { return wino_dispatch(d, st, cg2); }
int conv32_dispatch_k3(ConvDev& d, hipStream_t st, int force_cfg, bool must) { return dispatch32<3>(d, st, force_cfg, must); }
}

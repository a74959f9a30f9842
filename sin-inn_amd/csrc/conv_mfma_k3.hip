#include "conv_mfma_impl.h"
namespace sininn {
int conv_dispatch_k3(ConvDev& d, hipStream_t st, int force_cfg) { return dispatch<3>(d, st, force_cfg); }
}

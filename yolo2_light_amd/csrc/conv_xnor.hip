// placeholder: replaced by the XNOR implementation
#include <hip/hip_runtime.h>
#include "kernels.h"
namespace yl {
int launch_pack_sign_bits(const float *, uint64_t *, int, int, int, int, int, void *) { return (int)hipErrorNotSupported; }
int launch_conv_xnor(const ConvXnorArgs &, void *) { return (int)hipErrorNotSupported; }
}

// Instantiations of the split-precision (3 x fp16 MFMA) conv-stack kernel for one backbone kind.
// See conv_stack_f16.hip.h.
#include "conv_stack_f16.hip.h"
namespace wekws {
WEKWS_DEFINE_LAUNCHER_F16(KIND_DS, true)
}  // namespace wekws

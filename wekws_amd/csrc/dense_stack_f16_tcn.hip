// Instantiations of the dense-stack kernel (dense dilated convolutions straight from fp16 operand planes) for one
// backbone kind.  See dense_stack_f16.hip.h.
#include "dense_stack_f16.hip.h"
namespace wekws {
WEKWS_DEFINE_LAUNCHER_DENSE(KIND_TCN)
}  // namespace wekws

// Instantiations of the fused conv-stack kernel for one backbone kind (split per kind so the
// translation units compile in parallel).  See conv_stack.hip.h.
#include "conv_stack.hip.h"
namespace wekws {
WEKWS_DEFINE_LAUNCHER(KIND_MDTC, false)
}  // namespace wekws

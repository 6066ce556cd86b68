// Instantiations of the fused FSMN kernel.  See fsmn_f16.hip.h.
#include "fsmn_f16.hip.h"
namespace wekws {
int launch_fsmn_f16(int nt, const FsmnParams& P, const FsmnArgs& A, hipStream_t stream) {
  switch (nt) {
    case 1: return launch_fsmn_nt<1>(P, A, stream);
    case 2: return launch_fsmn_nt<2>(P, A, stream);
    case 3: return launch_fsmn_nt<3>(P, A, stream);
    case 4: return launch_fsmn_nt<4>(P, A, stream);
    default: return -1;
  }
}
}  // namespace wekws

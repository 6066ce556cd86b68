// Instantiations of the fused FSMN kernel.  See fsmn_f16.hip.h.
#include "fsmn_f16.hip.h"
namespace wekws {
int launch_fsmn_f16(int nt, int u, const FsmnParams& P, const FsmnArgs& A, hipStream_t stream) {
  switch (nt * 10 + u) {
    case 11: return launch_fsmn_nt<1, 1>(P, A, stream);
    case 21: return launch_fsmn_nt<2, 1>(P, A, stream);
    case 31: return launch_fsmn_nt<3, 1>(P, A, stream);
    case 41: return launch_fsmn_nt<4, 1>(P, A, stream);
    case 12: return launch_fsmn_nt<2, 2>(P, A, stream);
    case 22: return launch_fsmn_nt<4, 2>(P, A, stream);
    case 14: return launch_fsmn_nt<4, 4>(P, A, stream);
    default: return -1;
  }
}
}  // namespace wekws

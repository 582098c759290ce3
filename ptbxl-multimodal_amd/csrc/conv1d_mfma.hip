// conv1d_mfma.hip — fp32 MFMA implicit-GEMM Conv1d kernels (placeholder dispatch: the
// kernels land in the next milestone; until then every shape takes the direct path).
#include "common.h"

namespace ecg {
bool mfma_fwd_supported(int, int, int, int) { return false; }
int mfma_fwd_stat_partials(int, int, int, int) { return 0; }
int mfma_fwd(const float *, const float *, const float *, float *, float *, int, int, int, int,
             int, int, hipStream_t) {
    return fail(ECG_EINVAL, "mfma_fwd: not built");
}
bool mfma_wgrad_supported(int, int, int, int) { return false; }
size_t mfma_wgrad_ws_floats(int, int, int, int, int) { return 0; }
int mfma_wgrad(const float *, const float *, float *, float *, float *, int, int, int, int, int,
               int, hipStream_t) {
    return fail(ECG_EINVAL, "mfma_wgrad: not built");
}
}  // namespace ecg

// ddpg_mfma.hip -- placeholder until the matrix-core kernel lands (next commit).
#include "rlc_common.h"

bool rlc_mfma_supported(const RlcDims&) { return false; }

int rlc_launch_ddpg_update_mfma(const RlcDev&, int, int, int, int, const long long*, int, hipStream_t) {
    rlc_set_error("MFMA kernel not built");
    return 3;
}

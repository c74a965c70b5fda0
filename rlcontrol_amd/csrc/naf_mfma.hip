// naf_mfma.hip -- placeholder until the MFMA NAF kernel lands
#include "naf_common.h"
bool rlc_naf_mfma_supported(const RlcNafDims&) { return false; }
int rlc_launch_naf_update_mfma(const RlcNafDev&, int, int, int, int, const long long*, int, hipStream_t, const RlcNafRollout*) {
    rlc_set_error("MFMA NAF kernel not built");
    return 3;
}

// pv_sdec_fused.hip — fused persistent spatial-decoder kernel (placeholder until the kernel lands:
// reports "unsupported" so every plan takes the layer-by-layer path of pv_plan.hip).
#include "pv_sdec_fused.h"

bool pv_sdec_fused_supported(const pv_ivae_plan*) { return false; }
int64_t pv_sdec_fused_ws_bytes(const pv_ivae_plan*) { return 0; }
int pv_ivae_loss_and_grads_fused(const pv_ivae_plan*, int, hipStream_t) { return PV_EINVAL; }

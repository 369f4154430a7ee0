// pv_sdec_fused.h — the fused persistent spatial-decoder forward+backward path (pv_sdec_fused.hip).
#pragma once
#include "pv_common.h"

// true when the plan's architecture is the one the fused kernel is specialised for
bool pv_sdec_fused_supported(const pv_ivae_plan* p);
// scratch bytes the fused path needs inside the plan workspace's scratch region
int64_t pv_sdec_fused_ws_bytes(const pv_ivae_plan* p);
// full loss_and_grads using the fused decoder kernel (encoder/head stay on the layered kernels)
int pv_ivae_loss_and_grads_fused(const pv_ivae_plan* p, int want_grads, hipStream_t s);

// pv_conv.h — data-movement kernels of the convolutional nets (pv_conv.hip).  Channels-last activations
// [B][H][W][C]; 1-D data has W = 1 (nd = 1), 2-D nd = 2.
#pragma once
#include "pv_common.h"

int pv_maxpool2_fwd(const float* in, float* out, int B, int H, int W, int C, int nd, hipStream_t s);
int pv_maxpool2_bwd(const float* in, const float* dout, float* din, int B, int H, int W, int C, int nd, hipStream_t s);
int pv_upsample2_fwd(const float* in, float* out, int B, int H, int W, int C, int nd, hipStream_t s);
int pv_upsample2_bwd(const float* dout, float* din, int B, int H, int W, int C, int nd, hipStream_t s);
int pv_ncs_to_nsc(const float* in, float* out, int64_t B, int C, int64_t S, hipStream_t s);
int pv_nsc_to_ncs(const float* in, float* out, int64_t B, int C, int64_t S, hipStream_t s);
int pv_act_bwd(float* dy, const float* y, int64_t n, int act, hipStream_t s);
int pv_conv_wflip(const float* w, float* wt, int Cout, int Cin, int KK, hipStream_t s);
int pv_upsample2_bil_fwd(const float* in, float* out, int B, int H, int W, int C, hipStream_t s);
int pv_upsample2_bil_bwd(const float* dout, float* din, int B, int H, int W, int C, hipStream_t s);

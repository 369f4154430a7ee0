// pv_dec1d.h — a 1-D convolutional decoder stack as one forward and one input-gradient launch (pv_dec1d.hip)
#pragma once
#include "pv_common.h"
#include "pv_conv.h"
#define PV_D1_MAXOPS 12
// ops: the stack (pv_convstack.h conventions), a[0] = (B, L0, C0) channels-last.  Supported: Conv1d kernel 3 / 1 with widths that are
// multiples of 16 (the last layer's output may be narrower), activations other than GELU, an UPSAMPLE2 only after a kernel-1
// convolution without activation, lengths that are multiples of 16, every activation of a sample within the LDS buffers
bool pv_dec1d_supported(const pv_op* ops, int n, int nd, int L0, int C0);
// the run-time switch (PV_NO_DEC1D=1, pv_debug_dec1d): off = the layer-by-layer launches
bool pv_dec1d_enabled();
// the stack's weights tiled for both launches (pv_conv_wprep_table kind 8): size and table entries
int64_t pv_dec1d_wt_floats(const pv_op* ops, int n);
void pv_dec1d_wt_entries(const float* params, const pv_op* ops, int n, float* wt, PvWprepEntry* e, int& ne);
// a[1..n] written (the tensor between a fused kernel-1 convolution and its upsample never is)
int pv_dec1d_fwd(const float* params, const pv_op* ops, int n, const float* wt, int B, int L0, int C0, float* const* a, hipStream_t s);
// gown[i] <- dL/d(a[i]) for every convolution i, the producing convolution's activation derivative applied; g_out = dL/d(a[n])
int pv_dec1d_bwd(const pv_op* ops, int n, const float* wt, int B, int L0, int C0, float* const* a, const float* g_out,
                 float* const* gown, hipStream_t s);

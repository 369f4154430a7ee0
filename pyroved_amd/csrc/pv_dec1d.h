// pv_dec1d.h — a 1-D convolutional decoder stack as one forward and one input-gradient launch (pv_dec1d.hip)
#pragma once
#include "pv_common.h"
#include "pv_conv.h"
#define PV_D1_MAXOPS 12
// ops: the stack (pv_convstack.h conventions), a[0] = (B, L0, C0) channels-last.  Supported: Conv1d kernel 3 / 1 with widths that are
// multiples of 16 (the last layer's output may be narrower), activations other than GELU, an UPSAMPLE2 only after a kernel-1
// convolution without activation, lengths that are multiples of 16, every activation of a sample within the LDS buffers
bool pv_dec1d_supported(const pv_op* ops, int n, int nd, int L0, int C0);
// the experiments build's switch (PV_NO_DEC1D=1; per plan: PV_PLAN_NO_DEC1D): off = the layer-by-layer launches
bool pv_dec1d_enabled();
// the stack's weights tiled for both launches (pv_conv_wprep_table kind 8): size and table entries
int64_t pv_dec1d_wt_floats(const pv_op* ops, int n);
void pv_dec1d_wt_entries(const float* params, const pv_op* ops, int n, float* wt, PvWprepEntry* e, int& ne);
// latent_to_features (nets/conv.py:221-224: Linear z -> C0 * L0, viewed channels-first) riding in the same launches: forward
// a[0][b][l][c] = bias[c * L0 + l] + sum_k z[b][k] wt[k][l * C0 + c] is computed while the sample is staged (and written to
// a[0]); backward dz[b][k] = sum_{l,c} dL/d(a[0])[b][l][c] wt[k][l * C0 + c].  wt: pv_conv_wprep_table kind 7.  zd <= 8.
struct PvD1L2f { const float* z; const float* wt; const float* bias; float* dz; int zd; };
// the reparameterised sample and the head's backward in the same launches (ved.py:147-163 guide / model's latent site; a
// standard-normal prior, no per-sample weights): forward z = mu + softplus(s) eps from head (B, ldh) = [mu | s | ...], written
// to z / z_scale (and the optional copies), kl_part[2b], kl_part[2b+1] = beta * (log p(z_b), log q(z_b | x_b)); backward
// dhead (B, ldh) from the latent gradient of the same launch.  Needs PvD1L2f in the same call (z is produced / dz consumed here).
// part != null (forward): head itself is produced here, head[b][j] = bias[j] + sum_seg part[b][seg][j] (pv_convhead_fwd_partials),
// and written to head_out (B, ldh)
struct PvD1Head { const float* head; const float* eps; float* z; float* z_scale; float* z_loc_out; float* z_scale_out; float* kl_part;
                  float* dhead; int ldh; float beta;
                  const float* part = nullptr; const float* bias = nullptr; float* head_out = nullptr; int nseg = 0; };
inline bool pv_dec1d_l2f_ok(int zd) { return zd >= 1 && zd <= 8; }
// the observation likelihood of the stack's output (fc.py:143-152 through pv_lik_one; one output channel) in the forward launch:
// y (B, per) the target, loc / dlda (B, per) optional, llb[b] = the sample's log-likelihood
struct PvD1Lik { const float* y; float* loc; float* dlda; float* llb; int lik, sigmoid_out; float sig; };
// a[1..n] written (the tensor between a fused kernel-1 convolution and its upsample never is); l2f != null: a[0] too, from z
int pv_dec1d_fwd(const float* params, const pv_op* ops, int n, const float* wt, int B, int L0, int C0, float* const* a, hipStream_t s,
                 const PvD1L2f* l2f = nullptr, const PvD1Lik* lk = nullptr, const PvD1Head* hd = nullptr);
// gown[i] <- dL/d(a[i]) for every convolution i, the producing convolution's activation derivative applied; g_out = dL/d(a[n]);
// l2f != null: l2f->dz too
int pv_dec1d_bwd(const pv_op* ops, int n, const float* wt, int B, int L0, int C0, float* const* a, const float* g_out,
                 float* const* gown, hipStream_t s, const PvD1L2f* l2f = nullptr, const PvD1Head* hd = nullptr);

// pv_linear.h — the generic nn.Linear building blocks (GEMM + fused epilogues, pv_plan.hip) for other translation units.
#pragma once
#include "pv_common.h"

int64_t gemm_ws_need(int64_t M, int64_t N, int64_t K);
// y[M,N] = act(x[M,K] W[N,K]^T + b)
int linear_fwd(const float* x, int64_t ldx, const float* W, const float* b, float* y, float* pre, int64_t ldy,
               int64_t M, int64_t K, int64_t N, int act, void* ws, int64_t wsb, hipStream_t s);
// dx[M,K] = (dpre[M,N] W) * act_prev'(xact)
int linear_dgrad(const float* dpre, int64_t lddp, const float* W, float* dx, int64_t lddx, const float* xact,
                 const float* xpre, int64_t ldxa, int act_prev, int64_t M, int64_t K, int64_t N, void* ws, int64_t wsb,
                 hipStream_t s);
// dw[N,K] = dpre^T x ; db[N] = colsum(dpre)
int linear_wgrad(const float* dpre, int64_t lddp, const float* x, int64_t ldx, float* dw, float* db, int64_t M,
                 int64_t K, int64_t N, void* ws, int64_t wsb, hipStream_t s);

// convolutions (kernel 3, padding 1, stride 1, channels-last [B][H][W][C]; nd = 1: W = 1) on an implicit im2col operand
int conv3_fwd(const float* in, int B, int H, int W_, int C, int nd, const float* W, const float* b, float* y, int Cout,
              int act, void* ws, int64_t wsb, hipStream_t s);
int conv3_wgrad(const float* dpre, const float* in, int B, int H, int W_, int C, int nd, float* dw, float* db, int Cout,
                void* ws, int64_t wsb, hipStream_t s);

// Pipelined GEMM core for the pointwise (K = 1, stride 1) layers: launch descriptors shared by pw_gemm.hip (kernels)
// and gemm_conv.hip (plan choice + C ABI).  Row-major fp32 everywhere: x [M, K], w [N, K] (Conv1d / Linear weight as
// it lies), y [M, N]; M = B*T token rows.
#pragma once
#include "common.h"

// forward: y = act(x w^T + bias) * mask[row]; optional ypre (pre-activation) and the fused residual epilogue
// yres = res + rscale[col] * rkeep[row / resT] * y  (AffineDropPath + residual, blocks.py:1313)
struct PwFwd {
  const float* x[3]; const float* w[3]; const float* bias[3]; const float* mask[3]; float* y[3]; float* ypre[3];
  const float* res; const float* rscale; const float* rkeep; float* yres; int resT;
  int nprob, M, N, K, act;
};

// backward pair: g = dy * mask[row] * act'(ypre);  dx = (g w) * epi'(xpre)   [M, K]   (data gradient; `xpre`/`xact`: the
// activation derivative of the PRODUCER layer folded into this epilogue -- the GELU between fc1 and fc2);
// dw += g^T x  [N, K], dbias += colsum(g)   (float atomics into the flat gradient buffer)
struct PwBwd {
  const float* dy[3]; const float* w[3]; const float* x[3]; const float* mask[3]; const float* ypre[3];
  float* dx[3]; float* dw[3]; float* dbias[3];
  const float* xpre[3]; int xact;
  int nprob, M, N, K, act;
};

// cfg: tile configuration index (pw_gemm.hip: kPwCfgs); stages 2 or 3
int pw_fwd_launch(const PwFwd& a, int cfg, int stages, hipStream_t st);
int pw_fwd_cfg_count();
// dgrad tile cfg (same table), wgrad token splits (>= 1); dx[i] == nullptr skips the data gradient
int pw_bwd_launch(const PwBwd& a, int cfg, int splits, hipStream_t st);

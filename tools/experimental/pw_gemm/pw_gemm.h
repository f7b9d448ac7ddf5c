// Pipelined forward GEMM core for the pointwise (K = 1, stride 1) layers: launch descriptor shared by pw_gemm.hip (kernel)
// and gemm_conv.hip (plan choice + C ABI).  Row-major fp32 everywhere: x [M, K], w [N, K] (Conv1d / Linear weight as it
// lies), y [M, N]; M = B*T token rows.
#pragma once
#include "common.h"

// forward: y = act(x w^T + bias) * mask[row]; optional ypre (pre-activation) and the fused residual epilogue
// yres = res + rscale[col] * rkeep[row / resT] * y  (AffineDropPath + residual, blocks.py:1313)
struct PwFwd {
  const float* x[3]; const float* w[3]; const float* bias[3]; const float* mask[3]; float* y[3]; float* ypre[3];
  const float* res; const float* rscale; const float* rkeep; float* yres; int resT;
  int nprob, M, N, K, act;
};

// cfg: tile configuration index (pw_gemm.hip: kPwCfgs); wpc: workgroups per CU the persistent grid is sized for (0 = what
// the configuration's LDS allows, at most 2).  x / w rows must be 16-byte aligned (K % 4 == 0 and aligned bases).
int pw_fwd_launch(const PwFwd& a, int cfg, int wpc, hipStream_t st);
int pw_fwd_cfg_count();
bool pw_fwd_supported(int cfg, int K);

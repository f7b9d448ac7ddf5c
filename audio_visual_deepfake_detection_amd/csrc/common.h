// Shared helpers for the gfx950 kernels of libavdf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/avdf.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

void avdf_set_error(const char* fmt, ...);

#define AVDF_REQUIRE(cond, ...)                    \
  do {                                             \
    if (!(cond)) {                                 \
      avdf_set_error(__VA_ARGS__);                 \
      return AVDF_EINVAL;                          \
    }                                              \
  } while (0)

#define AVDF_CHECK_LAUNCH(name)                                         \
  do {                                                                  \
    hipError_t e_ = hipGetLastError();                                  \
    if (e_ != hipSuccess) {                                             \
      avdf_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return AVDF_ELAUNCH;                                              \
    }                                                                   \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Standard-normal CDF and exp(-x^2/2) from ONE v_exp_f32 and ONE v_rcp_f32: 0.5*erfc(|x|/sqrt(2)) by the rational form
// of Abramowitz & Stegun 7.1.26 (5 fmas; |error| <= 3e-7 in fp32 arithmetic, the rounding level of 0.5*(1+erff) itself:
// exact-erf GELU of libs/modeling/blocks.py:1291 within 6.2e-7 absolute over [-12, 12] on MI355X, pinned at 8e-7 by
// tests/test_gpu_ops.py::test_gelu_and_its_derivative_pinned_to_float64_erf; relative error of the x < -3 tail 1e-3..1e-2).
// libm's erff is ~40 instructions with two polynomial branches; the fc1 epilogue and the GELU derivative evaluate this
// once per element of the [tokens, 4*C] hidden activation.
__device__ __forceinline__ float normal_cdf_f(float x, float& e) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  e = __expf(-0.5f * x * x);
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float h = 0.5f * p * t * e;
  return x >= 0.0f ? 1.0f - h : h;
}
__device__ __forceinline__ float gelu_f(float x) {
  float e;
  return x * normal_cdf_f(x, e);
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float e;
  const float cdf = normal_cdf_f(x, e);
  return fmaf(x * 0.39894228040143267794f, e, cdf);
}
__device__ __forceinline__ float act_f(float x, int act) {
  if (act == AVDF_ACT_RELU) return fmaxf(x, 0.0f);
  if (act == AVDF_ACT_GELU) return gelu_f(x);
  if (act == AVDF_ACT_LEAKY) return x > 0.0f ? x : 0.01f * x;
  return x;
}
__device__ __forceinline__ float act_grad_f(float pre, int act) {
  if (act == AVDF_ACT_RELU) return pre > 0.0f ? 1.0f : 0.0f;
  if (act == AVDF_ACT_GELU) return gelu_grad_f(pre);
  if (act == AVDF_ACT_LEAKY) return pre > 0.0f ? 1.0f : 0.01f;
  return 1.0f;
}

// sum `nslab` partial vectors of length n: out[i] (+)= sum_s ws[s*n + i]
int avdf_reduce_slabs(const float* ws, float* out, int nslab, int64_t n, int accumulate, hipStream_t st);
// same, columns [split, n) are written to out2[0 .. n-split)
int avdf_reduce_slabs2(const float* ws, float* out, float* out2, int nslab, int64_t n, int64_t split,
                       int accumulate, hipStream_t st);

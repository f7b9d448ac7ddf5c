// Optimiser step on flat fp32 buffers ("next" row of the scope table):
//   grad-norm clipping (train_utils.py:306-310), AdamW (train_utils.py:125-144, torch.optim.AdamW
//   semantics: decoupled weight decay), ModelEma update (train_utils.py:258-263) in ONE pass.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, float* __restrict__ partial, int64_t n) {
  __shared__ float red[256];
  float s = 0.f;
  if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    // 16-byte loads, two independent accumulators per thread, two loads in flight per iteration
    const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * 256;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float s1 = 0.f;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
      const float4 a = x4[i], b = x4[i + stride];
      s += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
      s1 += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
    }
    if (i < n4) { const float4 a = x4[i]; s += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w; }
    s += s1;
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float t = x[(n4 << 2) + threadIdx.x]; s += t * t; }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += x[i] * x[i];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void sumsq_final_kernel(const float* __restrict__ partial, int nb, float* __restrict__ out, int accumulate) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + red[0];
}

__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, float* __restrict__ ema,
                                                        const float* __restrict__ gnorm_sq, const float* __restrict__ hp,
                                                        int64_t n, float beta1, float beta2, float eps, float wd,
                                                        float clip_norm, float ema_decay) {
  // hp (device, rewritten by the host every step so that a captured graph sees the schedule):
  // [0] = lr, [1] = 1 - beta1^t, [2] = 1 - beta2^t
  const float lr = hp[0], bias_c1 = hp[1], bias_c2 = hp[2];
  float cs = 1.f;
  if (gnorm_sq && clip_norm > 0.f) {
    const float nrm = sqrtf(gnorm_sq[0]);
    cs = fminf(clip_norm / (nrm + 1e-6f), 1.f);      // torch.nn.utils.clip_grad_norm_
  }
  const float step = lr / bias_c1;
  const float inv_sqrt_c2 = 1.f / sqrtf(bias_c2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * cs;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    pi -= step * mi / (sqrtf(vi) * inv_sqrt_c2 + eps);
    p[i] = pi;
    if (ema) ema[i] = ema[i] * ema_decay + (1.f - ema_decay) * pi;
  }
}

constexpr int kSumsqBlocks = 1024;
constexpr int kMaxSeg = 8;
struct WdSegments { int64_t end[kMaxSeg]; float wd[kMaxSeg]; int n; };

// the same update over a flat buffer made of up to kMaxSeg consecutive [.., end) ranges with their own weight decay
// (engine.TrainStep: gradient buckets x {decayed, not decayed}): one launch per step instead of one per range
__global__ __launch_bounds__(256) void adamw_ema_seg_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v, float* __restrict__ ema,
                                                            const float* __restrict__ gnorm_sq, const float* __restrict__ hp,
                                                            int64_t n, WdSegments seg, float beta1, float beta2, float eps,
                                                            float clip_norm, float ema_decay, float* __restrict__ g_zero) {
  const float lr = hp[0], bias_c1 = hp[1], bias_c2 = hp[2];
  float cs = 1.f;
  if (gnorm_sq && clip_norm > 0.f) cs = fminf(clip_norm / (sqrtf(gnorm_sq[0]) + 1e-6f), 1.f);
  const float step = lr / bias_c1;
  const float inv_sqrt_c2 = 1.f / sqrtf(bias_c2);
  // compile-time indices only (a run-time index into the by-value struct would put it in scratch memory)
  auto decay_of = [&](int64_t i) {
    float wd = 0.f;
    bool found = false;
#pragma unroll
    for (int s_ = 0; s_ < kMaxSeg; ++s_) {
      const bool hit = !found && s_ < seg.n && i < seg.end[s_];
      wd = hit ? seg.wd[s_] : wd;
      found = found || hit;
    }
    return wd;
  };
  auto update = [&](float gi, float pi, float mi, float vi, float ei, float wd, float& po, float& mo, float& vo, float& eo) {
    gi *= cs;
    pi *= (1.f - lr * wd);
    mo = beta1 * mi + (1.f - beta1) * gi;
    vo = beta2 * vi + (1.f - beta2) * gi * gi;
    po = pi - step * mo / (sqrtf(vo) * inv_sqrt_c2 + eps);
    eo = ei * ema_decay + (1.f - ema_decay) * po;
  };
  // four parameters per thread and access (16-byte loads / stores: a quarter of the memory instructions of the scalar walk);
  // the flat buffers are 16-byte aligned (host-checked), a ragged tail goes element by element
  const int64_t n4 = n >> 2;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
    const int64_t i = q << 2;
    const float4 g4 = *reinterpret_cast<const float4*>(g + i), p4 = *reinterpret_cast<const float4*>(p + i);
    const float4 m4 = *reinterpret_cast<const float4*>(m + i), v4 = *reinterpret_cast<const float4*>(v + i);
    float4 e4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ema) e4 = *reinterpret_cast<const float4*>(ema + i);
    float4 po, mo, vo, eo;
    update(g4.x, p4.x, m4.x, v4.x, e4.x, decay_of(i), po.x, mo.x, vo.x, eo.x);
    update(g4.y, p4.y, m4.y, v4.y, e4.y, decay_of(i + 1), po.y, mo.y, vo.y, eo.y);
    update(g4.z, p4.z, m4.z, v4.z, e4.z, decay_of(i + 2), po.z, mo.z, vo.z, eo.z);
    update(g4.w, p4.w, m4.w, v4.w, e4.w, decay_of(i + 3), po.w, mo.w, vo.w, eo.w);
    *reinterpret_cast<float4*>(m + i) = mo;
    *reinterpret_cast<float4*>(v + i) = vo;
    *reinterpret_cast<float4*>(p + i) = po;
    if (ema) *reinterpret_cast<float4*>(ema + i) = eo;
    if (g_zero) *reinterpret_cast<float4*>(g_zero + i) = make_float4(0.f, 0.f, 0.f, 0.f);   // the consumed gradient is cleared here: the next backward accumulates into zeros
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float po, mo, vo, eo;
    update(g[i], p[i], m[i], v[i], ema ? ema[i] : 0.f, decay_of(i), po, mo, vo, eo);
    m[i] = mo; v[i] = vo; p[i] = po;
    if (ema) ema[i] = eo;
    if (g_zero) g_zero[i] = 0.f;
  }
}

}  // namespace

extern "C" int64_t avdf_sumsq_ws(void) { return kSumsqBlocks; }

extern "C" int avdf_sumsq_ex(const float* x, float* out, float* ws, int64_t n, int accumulate, void* stream) {
  AVDF_REQUIRE(x && out && ws && n > 0, "sumsq: bad args");
  hipStream_t st = (hipStream_t)stream;
  int64_t nb = (n + 255) / 256;
  if (nb > kSumsqBlocks) nb = kSumsqBlocks;
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, ws, n);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, ws, (int)nb, out, accumulate);
  AVDF_CHECK_LAUNCH("sumsq");
  return AVDF_OK;
}

extern "C" int avdf_sumsq(const float* x, float* out, float* ws, int64_t n, void* stream) {
  return avdf_sumsq_ex(x, out, ws, n, 1, stream);
}

extern "C" int avdf_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, const float* gnorm_sq,
                              const float* hp, int64_t n, float beta1, float beta2, float eps, float wd,
                              float clip_norm, float ema_decay, void* stream) {
  AVDF_REQUIRE(p && g && m && v && hp && n > 0, "adamw_ema: bad args");
  int64_t nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(adamw_ema_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema, gnorm_sq, hp, n,
                     beta1, beta2, eps, wd, clip_norm, ema_decay);
  AVDF_CHECK_LAUNCH("adamw_ema");
  return AVDF_OK;
}

// nseg consecutive ranges [seg_end[i-1], seg_end[i]) (seg_end[nseg-1] == n) with weight decay seg_wd[i]: HOST arrays
extern "C" int avdf_adamw_ema_seg(float* p, const float* g, float* m, float* v, float* ema, const float* gnorm_sq,
                                  const float* hp, int64_t n, int nseg, const int64_t* seg_end, const float* seg_wd,
                                  float beta1, float beta2, float eps, float clip_norm, float ema_decay, int zero_grad,
                                  void* stream) {
  AVDF_REQUIRE(p && g && m && v && hp && n > 0 && seg_end && seg_wd, "adamw_ema_seg: bad args");
  AVDF_REQUIRE(nseg >= 1 && nseg <= kMaxSeg, "adamw_ema_seg: 1..%d segments", kMaxSeg);
  WdSegments seg;
  seg.n = nseg;
  int64_t prev = 0;
  for (int i = 0; i < nseg; ++i) {
    AVDF_REQUIRE(seg_end[i] >= prev, "adamw_ema_seg: segment ends must be non-decreasing");
    seg.end[i] = prev = seg_end[i];
    seg.wd[i] = seg_wd[i];
  }
  AVDF_REQUIRE(prev == n, "adamw_ema_seg: the last segment must end at n");
  AVDF_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)ema) & 15) == 0, "adamw_ema_seg: 16-byte alignment");
  int64_t nb = (n / 4 + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(adamw_ema_seg_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema, gnorm_sq, hp,
                     n, seg, beta1, beta2, eps, clip_norm, ema_decay, zero_grad ? const_cast<float*>(g) : (float*)nullptr);
  AVDF_CHECK_LAUNCH("adamw_ema_seg");
  return AVDF_OK;
}

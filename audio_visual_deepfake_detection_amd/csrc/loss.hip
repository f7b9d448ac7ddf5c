// Training targets and losses of the boundary head, fused forward + gradient.
//   avdf_loc_assign : label_points_single_video (av_fd_no_recon.py:502-600) for every clip
//   avdf_loc_losses : sigmoid focal loss (losses.py:5-53) with label smoothing
//                     (av_fd_no_recon.py:654-655), DIoU (losses.py:57-117), loss-normaliser EMA
//                     (av_fd_no_recon.py:645-648), video-level focal loss (:688-692), weighting (:702)
// num_classes == 1 (asserted by the reference datasets, deepfake_video_audio.py:1536).
#include "common.h"

namespace {

constexpr int kLossBlocks = 64;

// ws layout (floats): cls_t [B*P] | reg_t [B*P*2] | partial [kLossBlocks*3]
__global__ __launch_bounds__(256) void loc_assign_kernel(
    const float* __restrict__ points, const float* __restrict__ pmask, const float* __restrict__ gt,
    const int32_t* __restrict__ ngt, float* __restrict__ cls_t, float* __restrict__ reg_t,
    float* __restrict__ num_pos, int B, int P, int G, float radius, int center_sample) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  float posf = 0.f;
  if (i < (long)B * P) {
    const int p = (int)(i % P);
    const long b = i / P;
    const int n = ngt[b] < G ? ngt[b] : G;
    const float pt = points[p * 4 + 0], lo = points[p * 4 + 1], hi = points[p * 4 + 2], st = points[p * 4 + 3];
    float best = INFINITY; int bi = 0;
    for (int j = 0; j < n; ++j) {
      const float s0 = gt[(b * G + j) * 2 + 0], s1 = gt[(b * G + j) * 2 + 1];
      const float left = pt - s0, right = s1 - pt;
      bool inside;
      if (center_sample) {
        const float ctr = 0.5f * (s0 + s1);
        const float tmin = ctr - st * radius, tmax = ctr + st * radius;
        const float dl = pt - fmaxf(tmin, s0), dr = fminf(tmax, s1) - pt;
        inside = fminf(dl, dr) > 0.f;
      } else {
        inside = fminf(left, right) > 0.f;
      }
      const float mx = fmaxf(left, right);
      const bool inrng = (mx >= lo) && (mx <= hi);
      const float len = (inside && inrng) ? (s1 - s0) : INFINITY;
      if (len < best) { best = len; bi = j; }
    }
    float ct = 0.f, r0 = 0.f, r1 = 0.f;
    if (n > 0) {
      ct = best < INFINITY ? 1.f : 0.f;
      r0 = (pt - gt[(b * G + bi) * 2 + 0]) / st;
      r1 = (gt[(b * G + bi) * 2 + 1] - pt) / st;
    }
    cls_t[i] = ct;
    reg_t[i * 2 + 0] = r0;
    reg_t[i * 2 + 1] = r1;
    posf = (n > 0 && ct > 0.f && pmask[i] != 0.f) ? 1.f : 0.f;
  }
  // exact integer-valued count: block reduce then one atomic per block
  __shared__ float red[256];
  red[threadIdx.x] = posf;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0 && red[0] != 0.f) atomicAdd(num_pos, red[0]);
}

__device__ __forceinline__ void focal(float x, float t, float& loss, float& grad) {
  const float alpha = 0.25f;
  const float p = 1.0f / (1.0f + __expf(-x));
  const float ce = fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));
  const float pt = p * t + (1.f - p) * (1.f - t);
  const float at = alpha * t + (1.f - alpha) * (1.f - t);
  const float om = 1.f - pt;
  loss = at * ce * om * om;
  grad = at * (om * om * (p - t) - 2.f * ce * om * p * (1.f - p) * (2.f * t - 1.f));
}

__device__ __forceinline__ void diou(float lp, float rp, float lg, float rg, float& loss, float& dl, float& dr) {
  const float eps = 1e-8f;
  const float lk = fminf(lp, lg), rk = fminf(rp, rg);
  const float I = lk + rk;
  const float U = (lp + rp) + (lg + rg) - I;
  const float Uc = fmaxf(U, eps);
  const float iou = I / Uc;
  const float Lc = fmaxf(lp, lg) + fmaxf(rp, rg);
  const float Lcc = fmaxf(Lc, eps);
  const float rho = 0.5f * (rp - lp - rg + lg);
  const float q = rho / Lcc;
  loss = 1.f - iou + q * q;
  // torch.min/max route the gradient to the smaller/larger input, split evenly on exact ties
  const float dIl = lp < lg ? 1.f : (lp == lg ? 0.5f : 0.f);
  const float dIr = rp < rg ? 1.f : (rp == rg ? 0.5f : 0.f);
  const float dLl = lp > lg ? 1.f : (lp == lg ? 0.5f : 0.f);
  const float dLr = rp > rg ? 1.f : (rp == rg ? 0.5f : 0.f);
  const float ug = U > eps ? 1.f : 0.f, lgate = Lc > eps ? 1.f : 0.f;
  const float diou_l = (dIl * Uc - I * ug * (1.f - dIl)) / (Uc * Uc);
  const float diou_r = (dIr * Uc - I * ug * (1.f - dIr)) / (Uc * Uc);
  const float dq_l = (-0.5f * Lcc - rho * lgate * dLl) / (Lcc * Lcc);
  const float dq_r = (0.5f * Lcc - rho * lgate * dLr) / (Lcc * Lcc);
  dl = -diou_l + 2.f * q * dq_l;
  dr = -diou_r + 2.f * q * dq_r;
}

__global__ __launch_bounds__(256) void loc_losses_kernel(
    const float* __restrict__ logits, const float* __restrict__ offsets, const float* __restrict__ pmask,
    const int32_t* __restrict__ ngt, const float* __restrict__ cls_t, const float* __restrict__ reg_t,
    const float* __restrict__ num_pos, const float* __restrict__ state, float* __restrict__ partial,
    float* __restrict__ dlogits, float* __restrict__ doffsets,
    int B, int P, float label_smoothing, float loss_weight, float momentum) {
  const float np_ = fmaxf(num_pos[0], 1.f);
  const float norm = momentum * state[0] + (1.f - momentum) * np_;
  const float inv = 1.f / norm;
  const bool has_pos = num_pos[0] > 0.f;
  float s_cls = 0.f, s_reg = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)B * P; i += (long)gridDim.x * 256) {
    const long b = i / P;
    float gl = 0.f, g0 = 0.f, g1 = 0.f;
    if (ngt[b] > 0 && pmask[i] != 0.f) {
      const float t = cls_t[i] * (1.f - label_smoothing) + label_smoothing * 0.5f;   // /(num_classes+1)
      float l, g;
      focal(logits[i], t, l, g);
      s_cls += l;
      gl = g * inv;
      if (cls_t[i] > 0.f && has_pos) {
        float ld, d0, d1;
        diou(offsets[i * 2], offsets[i * 2 + 1], reg_t[i * 2], reg_t[i * 2 + 1], ld, d0, d1);
        s_reg += ld;
        g0 = d0 * inv * loss_weight;
        g1 = d1 * inv * loss_weight;
      }
    }
    dlogits[i] = gl;
    doffsets[i * 2] = g0;
    doffsets[i * 2 + 1] = g1;
  }
  __shared__ float red[2][256];
  red[0][threadIdx.x] = s_cls;
  red[1][threadIdx.x] = s_reg;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partial[blockIdx.x * 2] = red[0][0]; partial[blockIdx.x * 2 + 1] = red[1][0]; }
}

__global__ void loc_finalize_kernel(const float* __restrict__ partial, int nblocks, const int32_t* __restrict__ ngt,
                                    const float* __restrict__ video_logits, const float* __restrict__ num_pos,
                                    float* __restrict__ state, float* __restrict__ out, float* __restrict__ dvideo,
                                    int B, float loss_weight, float video_weight, float momentum) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float cls = 0.f, reg = 0.f;
  for (int i = 0; i < nblocks; ++i) { cls += partial[i * 2]; reg += partial[i * 2 + 1]; }
  const float norm = momentum * state[0] + (1.f - momentum) * fmaxf(num_pos[0], 1.f);
  cls /= norm;
  reg /= norm;
  float vid = 0.f;
  for (int b = 0; b < B; ++b) {
    float l, g;
    focal(video_logits[b], ngt[b] > 0 ? 1.f : 0.f, l, g);
    vid += l;
    if (dvideo) dvideo[b] = g * video_weight;
  }
  state[0] = norm;
  out[0] = cls; out[1] = reg; out[2] = vid;
  out[3] = cls + reg * loss_weight + video_weight * vid;
  out[4] = num_pos[0];
}

// generic sum-reduced sigmoid focal loss (alpha .25, gamma 2) over n elements, optional gradient
__global__ __launch_bounds__(1024) void focal_sum_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                         float* __restrict__ out, float* __restrict__ dx, long n) {
  __shared__ float red[1024];
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += 1024) {
    float l, g;
    focal(x[i], t[i], l, g);
    s += l;
    if (dx) dx[i] = g;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

// partial[blockIdx.x] = sum_b wb[b] * sum_i |a[b,i] - b[b,i]|; da = wb[b] * sign(a-b) * inv_denom
__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                 const float* __restrict__ wb, float* __restrict__ partial,
                                                 float* __restrict__ da, int B, long per, float inv_denom) {
  __shared__ float red[256];
  float s = 0.f;
  const long total = (long)B * per;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const float w_ = wb[i / per];
    const float d = a[i] - b[i];
    s += w_ * fabsf(d);
    if (da) da[i] = w_ * inv_denom * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void l1_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ out, float inv_denom) {
  if (threadIdx.x != 0) return;
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += partial[i];
  out[0] = s * inv_denom;
}

// engine variants: the set of real clips (ngt == 0) and the GLOBAL number of real clips are device data, so the launch
// can sit in a captured graph and the count can be all-reduced across data-parallel ranks first
__global__ void count_real_kernel(const int32_t* __restrict__ ngt, int B, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int n = 0;
  for (int b = 0; b < B; ++b) n += (ngt[b] == 0);
  out[0] = (float)n;
}

__global__ __launch_bounds__(256) void l1_dev_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     const int32_t* __restrict__ ngt, const float* __restrict__ n_real,
                                                     float* __restrict__ partial, float* __restrict__ da, int B, long per) {
  __shared__ float red[256];
  float s = 0.f;
  const long total = (long)B * per;
  const float inv_denom = 1.f / (n_real[0] * (float)per);          // n_real == 0: inf * 0 = nan, like torch.mean of nothing
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const float w_ = ngt[i / per] == 0 ? 1.f : 0.f;
    const float d = a[i] - b[i];
    s += w_ * fabsf(d);
    if (da) da[i] = w_ == 0.f ? 0.f : inv_denom * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void l1_dev_final_kernel(const float* __restrict__ partial, int n, const float* __restrict__ n_real, long per,
                                    float* __restrict__ out, float* __restrict__ final_loss) {
  if (threadIdx.x != 0) return;
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += partial[i];
  const float v = n_real[0] > 0.f ? s / (n_real[0] * (float)per) : nanf("");
  out[0] = v;
  if (final_loss) final_loss[0] += v;
}

}  // namespace

constexpr int kL1Blocks = 512;

extern "C" int avdf_count_real(const int32_t* ngt, float* out, int B, void* stream) {
  AVDF_REQUIRE(ngt && out && B > 0, "count_real: bad args");
  hipLaunchKernelGGL(count_real_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ngt, B, out);
  AVDF_CHECK_LAUNCH("count_real");
  return AVDF_OK;
}

// exp5 reconstruction loss inside the captured engine step: real clips = {b : ngt[b] == 0}; n_real = device float[1], the
// number of real clips of the GLOBAL batch (all-reduced by the caller); out[0] = this rank's share of the mean, added to
// final_loss[0] when given; da = d(out)/da
extern "C" int avdf_l1_mean_dev(const float* a, const float* b, const int32_t* ngt, const float* n_real, float* out,
                                float* final_loss, float* da, float* ws, int B, int64_t per_clip, void* stream) {
  AVDF_REQUIRE(a && b && ngt && n_real && out && ws && B > 0 && per_clip > 0, "l1_mean_dev: bad args");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(l1_dev_kernel, dim3(kL1Blocks), dim3(256), 0, st, a, b, ngt, n_real, ws, da, B, (long)per_clip);
  hipLaunchKernelGGL(l1_dev_final_kernel, dim3(1), dim3(64), 0, st, ws, kL1Blocks, n_real, (long)per_clip, out, final_loss);
  AVDF_CHECK_LAUNCH("l1_mean_dev");
  return AVDF_OK;
}

extern "C" int64_t avdf_l1_mean_ws(void) { return kL1Blocks; }

// L1 reconstruction loss of exp5 on the REAL clips only (av_fd_meta_arch.py:683-686):
//   out[0] = sum_b wb[b] * sum |a - b| * inv_denom,  da = wb[b] * sign(a - b) * inv_denom   (wb = 1 for real clips)
extern "C" int avdf_l1_mean(const float* a, const float* b, const float* wb, float* out, float* da, float* ws,
                            int B, int64_t per_clip, float inv_denom, void* stream) {
  AVDF_REQUIRE(a && b && wb && out && ws && B > 0 && per_clip > 0, "l1_mean: bad args");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(l1_kernel, dim3(kL1Blocks), dim3(256), 0, st, a, b, wb, ws, da, B, (long)per_clip, inv_denom);
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(64), 0, st, ws, kL1Blocks, out, inv_denom);
  AVDF_CHECK_LAUNCH("l1_mean");
  return AVDF_OK;
}

// sigmoid_focal_loss(x, t, reduction='sum') of libs/modeling/losses.py:5-53 (single workgroup: used for the
// [n_fake, T] frame-level loss of exp13, av_fd_no_recon2.py:718-725); dx (optional) = d(sum)/dx.
extern "C" int avdf_focal_loss_sum(const float* x, const float* t, float* out, float* dx, int64_t n, void* stream) {
  AVDF_REQUIRE(x && t && out && n > 0, "focal_loss_sum: bad args");
  hipLaunchKernelGGL(focal_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, t, out, dx, (long)n);
  AVDF_CHECK_LAUNCH("focal_loss_sum");
  return AVDF_OK;
}

extern "C" int64_t avdf_loc_losses_ws(int B, int P) { return (int64_t)B * P * 3 + kLossBlocks * 2 + 4; }

// ws: [cls_t B*P][reg_t 2*B*P][partial 2*kLossBlocks]; num_pos: device float[1] (zeroed here)
extern "C" int avdf_loc_assign(const float* points, const float* pmask, const float* gt, const int32_t* ngt,
                               float* ws, float* num_pos, int B, int P, int G, float radius, int center_sample,
                               void* stream) {
  AVDF_REQUIRE(points && pmask && gt && ngt && ws && num_pos && B > 0 && P > 0 && G > 0, "loc_assign: bad args");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(num_pos, 0, sizeof(float), st);
  AVDF_REQUIRE(e == hipSuccess, "loc_assign: memset failed: %s", hipGetErrorString(e));
  float* cls_t = ws;
  float* reg_t = ws + (size_t)B * P;
  hipLaunchKernelGGL(loc_assign_kernel, dim3((unsigned)(((long)B * P + 255) / 256)), dim3(256), 0, st,
                     points, pmask, gt, ngt, cls_t, reg_t, num_pos, B, P, G, radius, center_sample);
  AVDF_CHECK_LAUNCH("loc_assign");
  return AVDF_OK;
}

extern "C" int avdf_loc_losses(const float* logits, const float* offsets, const float* pmask, const int32_t* ngt,
                               const float* video_logits, const float* ws_targets, const float* num_pos,
                               float* state, float* out, float* dlogits, float* doffsets, float* dvideo,
                               float* ws_partial, int B, int P, float label_smoothing, float loss_weight,
                               float video_weight, float momentum, void* stream) {
  AVDF_REQUIRE(logits && offsets && pmask && ngt && video_logits && ws_targets && num_pos && state && out &&
               dlogits && doffsets && ws_partial && B > 0 && P > 0, "loc_losses: bad args");
  hipStream_t st = (hipStream_t)stream;
  const float* cls_t = ws_targets;
  const float* reg_t = ws_targets + (size_t)B * P;
  hipLaunchKernelGGL(loc_losses_kernel, dim3(kLossBlocks), dim3(256), 0, st, logits, offsets, pmask, ngt, cls_t, reg_t,
                     num_pos, state, ws_partial, dlogits, doffsets, B, P, label_smoothing, loss_weight, momentum);
  AVDF_CHECK_LAUNCH("loc_losses");
  hipLaunchKernelGGL(loc_finalize_kernel, dim3(1), dim3(64), 0, st, ws_partial, kLossBlocks, ngt, video_logits, num_pos,
                     state, out, dvideo, B, loss_weight, video_weight, momentum);
  AVDF_CHECK_LAUNCH("loc_finalize");
  return AVDF_OK;
}

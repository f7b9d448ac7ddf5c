// Normalisation kernels (HBM-bound, wave-shuffle reductions).
//   * channel LayerNorm over the contiguous C axis of a time-major token row
//     (libs/modeling/blocks.py:97-112), optional fused ReLU and "+pe*mask";
//   * InstanceNorm1d (stats over T, incl. padded positions) + LeakyReLU of DownBlock
//     (libs/modeling/blocks.py:1508-1516);
//   * max+avg pooling over T (libs/modeling/blocks.py:1610-1614).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int kLnMaxV = 8;            // float4 per lane -> C <= 2048
constexpr int kLnBwdBlocks = 256;     // 128 while the workgroups added their partials by float atomics; rows + deferred fold: 843 -> 846 clips/s

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ y, float* __restrict__ mean_o, float* __restrict__ rstd_o,
    int M, int C, float eps, int relu, const float* __restrict__ pe, const float* __restrict__ mask, int T,
    const float* __restrict__ rowscale) {
  // rowscale (optional): y = act(LN(x)) * rowscale[row] -- the shared heads run on the gapped concatenation of the pyramid
  // levels and put the gap rows back to zero after every LayerNorm (one launch less per layer than a separate row scale)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wid;
  if (row >= M) return;
  const float* xr = x + (size_t)row * C;
  float4 v[NV], gv[NV], bv[NV];                        // gamma / beta / row factors fetched with the row, not after its reductions
  const float mk = (pe && mask) ? mask[row] : 1.0f;
  const float rs = rowscale ? rowscale[row] : 1.0f;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    gv[j] = make_float4(1.f, 1.f, 1.f, 1.f); bv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
      v[j] = ld4(xr + c);
      if (gamma) { gv[j] = ld4(gamma + c); bv[j] = ld4(beta + c); }
      s += v[j].x + v[j].y + v[j].z + v[j].w;
    }
    else v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    if (c < C) {
      const float a = v[j].x - mean, b = v[j].y - mean, d = v[j].z - mean, e = v[j].w - mean;
      q += a * a + b * b + d * d + e * e;
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0 && mean_o) { mean_o[row] = mean; rstd_o[row] = rstd; }
  const int t = pe ? row % T : 0;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    if (c < C) {
      const float4 g = gv[j], bb = bv[j];
      float4 o;
      o.x = (v[j].x - mean) * rstd * g.x + bb.x;
      o.y = (v[j].y - mean) * rstd * g.y + bb.y;
      o.z = (v[j].z - mean) * rstd * g.z + bb.z;
      o.w = (v[j].w - mean) * rstd * g.w + bb.w;
      if (relu == 1) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      else if (relu == 2) { o.x = gelu_f(o.x); o.y = gelu_f(o.y); o.z = gelu_f(o.z); o.w = gelu_f(o.w); }   // fused GELU (extractors)
      if (pe) {
        const float4 p = ld4(pe + (size_t)t * C + c);
        o.x += p.x * mk; o.y += p.y * mk; o.z += p.z * mk; o.w += p.w * mk;
      }
      if (rowscale) { o.x *= rs; o.y *= rs; o.z *= rs; o.w *= rs; }
      *reinterpret_cast<float4*>(y + (size_t)row * C + c) = o;
    }
  }
}

// ws layout: [gridDim.x][2*C]  (per block: dgamma partials, then dbeta partials)
// NW waves per workgroup (one row per wave at a time); the per-channel dgamma/dbeta partials of the NW waves are
// folded in LDS, so only gridDim.x workgroups emit atomics / workspace rows.
template <int NV, int NW>
__global__ __launch_bounds__(64 * NW) void ln_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
    float* __restrict__ dx, float* __restrict__ ws, float* __restrict__ dg_atomic, float* __restrict__ db_atomic,
    int M, int C, int relu, const float* __restrict__ rowscale) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [NW][2][C]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float4 ag[NV], ab[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) { ag[j] = make_float4(0.f, 0.f, 0.f, 0.f); ab[j] = ag[j]; }
  // (a wave walks M / (128 x 16) rows one after the other: the next row's loads are in flight while this one is reduced)
  float4 gm[NV], bt[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    gm[j] = make_float4(1.f, 1.f, 1.f, 1.f); bt[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gamma && c < C) { gm[j] = ld4(gamma + c); bt[j] = ld4(beta + c); }
  }
  struct Row { float4 xv[NV], d[NV]; float mean, rstd, rs; };
  auto load_row = [&](int row, Row& w) {
    w.mean = mean_i[row]; w.rstd = rstd_i[row];
    w.rs = rowscale ? rowscale[row] : 1.0f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (j * 64 + lane) * 4;
      w.xv[j] = make_float4(0.f, 0.f, 0.f, 0.f); w.d[j] = w.xv[j];
      if (c < C) { w.xv[j] = ld4(x + (size_t)row * C + c); w.d[j] = ld4(dy + (size_t)row * C + c); }
    }
  };
  const int step = gridDim.x * NW;
  int row = blockIdx.x * NW + wid;
  Row cur, nxt;
  if (row < M) load_row(row, cur);
  for (; row < M; row += step) {
    if (row + step < M) load_row(row + step, nxt);
    const float mean = cur.mean, rstd = cur.rstd, rs = cur.rs;
    float4 xh[NV], gy[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (j * 64 + lane) * 4;
      if (c < C) {
        const float4 xv = cur.xv[j], g = gm[j], bb = bt[j];
        float4 d = cur.d[j];
        d.x *= rs; d.y *= rs; d.z *= rs; d.w *= rs;
        float4 h;
        h.x = (xv.x - mean) * rstd; h.y = (xv.y - mean) * rstd; h.z = (xv.z - mean) * rstd; h.w = (xv.w - mean) * rstd;
        if (relu) {
          if (h.x * g.x + bb.x <= 0.f) d.x = 0.f;
          if (h.y * g.y + bb.y <= 0.f) d.y = 0.f;
          if (h.z * g.z + bb.z <= 0.f) d.z = 0.f;
          if (h.w * g.w + bb.w <= 0.f) d.w = 0.f;
        }
        ag[j].x += d.x * h.x; ag[j].y += d.y * h.y; ag[j].z += d.z * h.z; ag[j].w += d.w * h.w;
        ab[j].x += d.x; ab[j].y += d.y; ab[j].z += d.z; ab[j].w += d.w;
        d.x *= g.x; d.y *= g.y; d.z *= g.z; d.w *= g.w;
        s1 += d.x + d.y + d.z + d.w;
        s2 += d.x * h.x + d.y * h.y + d.z * h.z + d.w * h.w;
        xh[j] = h; gy[j] = d;
      } else { xh[j] = make_float4(0.f, 0.f, 0.f, 0.f); gy[j] = xh[j]; }
    }
    const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (j * 64 + lane) * 4;
      if (c < C) {
        float4 o;
        o.x = rstd * (gy[j].x - m1 - xh[j].x * m2);
        o.y = rstd * (gy[j].y - m1 - xh[j].y * m2);
        o.z = rstd * (gy[j].z - m1 - xh[j].z * m2);
        o.w = rstd * (gy[j].w - m1 - xh[j].w * m2);
        *reinterpret_cast<float4*>(dx + (size_t)row * C + c) = o;
      }
    }
    cur = nxt;
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    if (c < C) {
      *reinterpret_cast<float4*>(red + (size_t)(wid * 2 + 0) * C + c) = ag[j];
      *reinterpret_cast<float4*>(red + (size_t)(wid * 2 + 1) * C + c) = ab[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 64 * NW) {
    const int which = i / C, c = i % C;
    float s = 0.f;
#pragma unroll
    for (int w_ = 0; w_ < NW; ++w_) s += red[(size_t)(w_ * 2 + which) * C + c];
    if (dg_atomic) atomicAdd((which == 0 ? dg_atomic : db_atomic) + c, s);   // accumulate path: no second kernel
    else ws[(size_t)blockIdx.x * 2 * C + i] = s;
  }
}

// one workgroup per (b, 64-channel group); 256 threads = 64 channels x 4 time slices
// InstanceNorm kernels: 64 channels x kInG time-groups per workgroup (the time loop is latency-bound: 16 groups keep
// 4x more rows in flight than the original 4)
constexpr int kInG = 16;
__device__ __forceinline__ float in_sum(const float (*red)[64], int cl) {
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < kInG; ++q) s += red[q][cl];
  return s;
}

__global__ __launch_bounds__(64 * kInG) void instnorm_lrelu_fwd_kernel(
    const float* __restrict__ x, float* __restrict__ y, float* __restrict__ mean_o, float* __restrict__ rstd_o,
    int T, int C, float eps, float slope) {
  __shared__ float red[kInG][64];
  __shared__ float stat[2][64];
  const int cl = threadIdx.x & 63, ts = threadIdx.x >> 6;
  const int groups = (C + 63) / 64;
  const int b = blockIdx.x / groups, c = (blockIdx.x % groups) * 64 + cl;
  const bool ok = c < C;
  const float* xb = x + (size_t)b * T * C + c;
  float s = 0.f;
  if (ok) for (int t = ts; t < T; t += kInG) s += xb[(size_t)t * C];
  red[ts][cl] = s;
  __syncthreads();
  if (ts == 0) stat[0][cl] = in_sum(red, cl) / (float)T;
  __syncthreads();
  const float mean = stat[0][cl];
  float q = 0.f;
  if (ok) for (int t = ts; t < T; t += kInG) { const float d = xb[(size_t)t * C] - mean; q += d * d; }
  red[ts][cl] = q;
  __syncthreads();
  if (ts == 0) {
    const float var = in_sum(red, cl) / (float)T;
    stat[1][cl] = 1.0f / sqrtf(var + eps);
    if (ok && mean_o) { mean_o[(size_t)b * C + c] = mean; rstd_o[(size_t)b * C + c] = stat[1][cl]; }
  }
  __syncthreads();
  const float rstd = stat[1][cl];
  float* yb = y + (size_t)b * T * C + c;
  if (ok) for (int t = ts; t < T; t += kInG) {
    const float h = (xb[(size_t)t * C] - mean) * rstd;
    yb[(size_t)t * C] = h > 0.f ? h : h * slope;
  }
}

__global__ __launch_bounds__(64 * kInG) void instnorm_lrelu_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean_i,
    const float* __restrict__ rstd_i, float* __restrict__ dx, int T, int C, float slope) {
  __shared__ float red[2][kInG][64];
  const int cl = threadIdx.x & 63, ts = threadIdx.x >> 6;
  const int groups = (C + 63) / 64;
  const int b = blockIdx.x / groups, c = (blockIdx.x % groups) * 64 + cl;
  const bool ok = c < C;
  const size_t base = (size_t)b * T * C + c;
  const float mean = ok ? mean_i[(size_t)b * C + c] : 0.f, rstd = ok ? rstd_i[(size_t)b * C + c] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  if (ok) for (int t = ts; t < T; t += kInG) {
    const float h = (x[base + (size_t)t * C] - mean) * rstd;
    const float g = dy[base + (size_t)t * C] * (h > 0.f ? 1.f : slope);
    s1 += g; s2 += g * h;
  }
  red[0][ts][cl] = s1; red[1][ts][cl] = s2;
  __syncthreads();
  const float m1 = in_sum(red[0], cl) / (float)T;
  const float m2 = in_sum(red[1], cl) / (float)T;
  if (ok) for (int t = ts; t < T; t += kInG) {
    const float h = (x[base + (size_t)t * C] - mean) * rstd;
    const float g = dy[base + (size_t)t * C] * (h > 0.f ? 1.f : slope);
    dx[base + (size_t)t * C] = rstd * (g - m1 - h * m2);
  }
}

__global__ __launch_bounds__(256) void maxavgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int T, int C) {
  __shared__ float rmax[4][64], rsum[4][64];
  const int cl = threadIdx.x & 63, ts = threadIdx.x >> 6;
  const int groups = (C + 63) / 64;
  const int b = blockIdx.x / groups, c = (blockIdx.x % groups) * 64 + cl;
  const bool ok = c < C;
  float mx = -INFINITY, s = 0.f;
  if (ok) for (int t = ts; t < T; t += 4) { const float v = x[((size_t)b * T + t) * C + c]; mx = fmaxf(mx, v); s += v; }
  rmax[ts][cl] = mx; rsum[ts][cl] = s;
  __syncthreads();
  if (ts == 0 && ok) {
    y[(size_t)b * 2 * C + c] = fmaxf(fmaxf(rmax[0][cl], rmax[1][cl]), fmaxf(rmax[2][cl], rmax[3][cl]));
    y[(size_t)b * 2 * C + C + c] = (rsum[0][cl] + rsum[1][cl] + rsum[2][cl] + rsum[3][cl]) / (float)T;
  }
}

// dx[b,t,c] = dy[b,c]*[t == first argmax] + dy[b,C+c]/T
__global__ __launch_bounds__(256) void maxavgpool_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             float* __restrict__ dx, int T, int C) {
  const int groups = (C + 255) / 256;
  const int b = blockIdx.x / groups, c = (blockIdx.x % groups) * 256 + threadIdx.x;
  if (c >= C) return;
  float mx = -INFINITY; int am = 0;
  for (int t = 0; t < T; ++t) { const float v = x[((size_t)b * T + t) * C + c]; if (v > mx) { mx = v; am = t; } }
  const float gm = dy[(size_t)b * 2 * C + c], ga = dy[(size_t)b * 2 * C + C + c] / (float)T;
  for (int t = 0; t < T; ++t) dx[((size_t)b * T + t) * C + c] = ga + (t == am ? gm : 0.f);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

static int layernorm_fwd_impl(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                              int M, int C, float eps, int relu, const float* pe, const float* mask, int T,
                              const float* rowscale, void* stream);

extern "C" int avdf_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                                  float* mean, float* rstd, int M, int C, float eps, int relu,
                                  const float* pe, const float* mask, int T, void* stream) {
  return layernorm_fwd_impl(x, gamma, beta, y, mean, rstd, M, C, eps, relu, pe, mask, T, nullptr, stream);
}

// y = act(LN(x)) * rowscale[row] (no position table): the LayerNorms of the shared heads on the gapped level sequence
extern "C" int avdf_layernorm_fwd_rs(const float* x, const float* gamma, const float* beta, float* y,
                                     float* mean, float* rstd, int M, int C, float eps, int relu,
                                     const float* rowscale, void* stream) {
  AVDF_REQUIRE(rowscale, "layernorm_fwd_rs: null rowscale");
  return layernorm_fwd_impl(x, gamma, beta, y, mean, rstd, M, C, eps, relu, nullptr, nullptr, 0, rowscale, stream);
}

static int layernorm_fwd_impl(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                              int M, int C, float eps, int relu, const float* pe, const float* mask, int T,
                              const float* rowscale, void* stream) {
  AVDF_REQUIRE(x && y && M > 0 && C > 0, "layernorm_fwd: bad args");
  AVDF_REQUIRE(C % 4 == 0 && C <= 256 * kLnMaxV, "layernorm_fwd: C must be a multiple of 4 and <= %d (C=%d)", 256 * kLnMaxV, C);
  AVDF_REQUIRE((gamma == nullptr) == (beta == nullptr), "layernorm_fwd: gamma/beta must both be given or both NULL");
  AVDF_REQUIRE((mean == nullptr) == (rstd == nullptr), "layernorm_fwd: mean/rstd must both be given or both NULL");
  AVDF_REQUIRE(!pe || T > 0, "layernorm_fwd: pe needs T");
  AVDF_REQUIRE(aligned16(x) && aligned16(y) && (!gamma || (aligned16(gamma) && aligned16(beta))) && (!pe || aligned16(pe)), "layernorm_fwd: 16-byte alignment");
#define LN_FWD(NV_) hipLaunchKernelGGL(ln_fwd_kernel<NV_>, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, \
                                      x, gamma, beta, y, mean, rstd, M, C, eps, relu, pe, mask, T, rowscale)
  if (C <= 256) LN_FWD(1); else if (C <= 512) LN_FWD(2); else if (C <= 1024) LN_FWD(4); else LN_FWD(8);
#undef LN_FWD
  AVDF_CHECK_LAUNCH("layernorm_fwd");
  return AVDF_OK;
}

// ---- fused AffineDropPath residual + LayerNorm (TransformerBlock: blocks.py:1311-1313 / 870-872) --------------
//   out = a*ma[row] + scale[c]*keep[b]*(r*mr[row]);  h = LN(out)*gamma + beta.       One wave per token row.
template <int NV>
__global__ __launch_bounds__(256) void residual_ln_fwd_kernel(
    const float* __restrict__ a, const float* __restrict__ ma, const float* __restrict__ r, const float* __restrict__ mr,
    const float* __restrict__ scale, const float* __restrict__ keep, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ out, float* __restrict__ h, float* __restrict__ mean_o,
    float* __restrict__ rstd_o, int M, int T, int C, float eps) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wid;
  if (row >= M) return;
  const float fa = ma ? ma[row] : 1.f;
  float fr = keep ? keep[row / T] : 1.f;
  if (mr) fr *= mr[row];
  float4 v[NV], gv[NV], bv[NV];                        // gamma / beta fetched with the row, not after its reductions
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    v[j] = make_float4(0.f, 0.f, 0.f, 0.f); gv[j] = v[j]; bv[j] = v[j];
    if (c < C) {
      gv[j] = ld4(gamma + c); bv[j] = ld4(beta + c);
      const float4 av = ld4(a + (size_t)row * C + c), rv = ld4(r + (size_t)row * C + c), sc = ld4(scale + c);
      v[j].x = av.x * fa + rv.x * sc.x * fr; v[j].y = av.y * fa + rv.y * sc.y * fr;
      v[j].z = av.z * fa + rv.z * sc.z * fr; v[j].w = av.w * fa + rv.w * sc.w * fr;
      *reinterpret_cast<float4*>(out + (size_t)row * C + c) = v[j];
      s += v[j].x + v[j].y + v[j].z + v[j].w;
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    if (c < C) {
      const float d0 = v[j].x - mean, d1 = v[j].y - mean, d2 = v[j].z - mean, d3 = v[j].w - mean;
      q += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    if (c < C) {
      const float4 g = gv[j], bb = bv[j];
      float4 o;
      o.x = (v[j].x - mean) * rstd * g.x + bb.x; o.y = (v[j].y - mean) * rstd * g.y + bb.y;
      o.z = (v[j].z - mean) * rstd * g.z + bb.z; o.w = (v[j].w - mean) * rstd * g.w + bb.w;
      *reinterpret_cast<float4*>(h + (size_t)row * C + c) = o;
    }
  }
}

// backward: dtot = dout (direct consumer of `out`, may be NULL) + LN'(dh);  da = dtot*ma;  dr = dtot*scale*keep*mr;
// dscale += sum dtot*r*keep*mr;  dgamma += sum dh*xhat;  dbeta += sum dh.  NW waves, partials folded in LDS, then
// atomics (accumulate path) or a workspace row [3*C] per workgroup.
template <int NV, int NW>
__global__ __launch_bounds__(64 * NW) void residual_ln_bwd_kernel(
    const float* __restrict__ dout, const float* __restrict__ dh, const float* __restrict__ out,
    const float* __restrict__ mean_i, const float* __restrict__ rstd_i, const float* __restrict__ gamma,
    const float* __restrict__ ma, const float* __restrict__ r, const float* __restrict__ mr,
    const float* __restrict__ scale, const float* __restrict__ keep, float* __restrict__ da, float* __restrict__ dr,
    float* __restrict__ ws, float* __restrict__ dscale_at, float* __restrict__ dg_at, float* __restrict__ db_at,
    int M, int T, int C) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [NW][3][C]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float4 as[NV], ag[NV], ab[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) { as[j] = make_float4(0.f, 0.f, 0.f, 0.f); ag[j] = as[j]; ab[j] = as[j]; }
  float4 gm[NV], scv[NV];                               // per-column constants of this lane
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    gm[j] = make_float4(0.f, 0.f, 0.f, 0.f); scv[j] = gm[j];
    if (c < C) { gm[j] = ld4(gamma + c); scv[j] = ld4(scale + c); }
  }
  // A wave walks only M / (128 x 16) rows one after the other, so the kernel's duration IS its rows' memory round trips:
  // every load of a row goes out before the first reduction (one round trip per row, not two), and the loads of the NEXT
  // row are in flight while this one is reduced.
  struct Row { float4 ov[NV], d[NV], ev[NV], rv[NV]; float mean, rstd, fa, fr; };
  auto load_row = [&](int row, Row& w) {
    w.mean = mean_i[row]; w.rstd = rstd_i[row];
    w.fa = ma ? ma[row] : 1.f;
    w.fr = keep ? keep[row / T] : 1.f;
    if (mr) w.fr *= mr[row];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (j * 64 + lane) * 4;
      w.ov[j] = make_float4(0.f, 0.f, 0.f, 0.f); w.d[j] = w.ov[j]; w.ev[j] = w.ov[j]; w.rv[j] = w.ov[j];
      if (c < C) {
        w.ov[j] = ld4(out + (size_t)row * C + c);
        w.d[j] = ld4(dh + (size_t)row * C + c);
        if (dout) w.ev[j] = ld4(dout + (size_t)row * C + c);
        w.rv[j] = ld4(r + (size_t)row * C + c);
      }
    }
  };
  const int step = gridDim.x * NW;
  int row = blockIdx.x * NW + wid;
  Row cur, nxt;
  if (row < M) load_row(row, cur);
  for (; row < M; row += step) {
    if (row + step < M) load_row(row + step, nxt);
    const float mean = cur.mean, rstd = cur.rstd, fa = cur.fa, fr = cur.fr;
    float4 xh[NV], gy[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const float4 ov = cur.ov[j], g = gm[j];
      float4 d = cur.d[j];
      float4 hh;
      hh.x = (ov.x - mean) * rstd; hh.y = (ov.y - mean) * rstd; hh.z = (ov.z - mean) * rstd; hh.w = (ov.w - mean) * rstd;
      if ((j * 64 + lane) * 4 >= C) hh = make_float4(0.f, 0.f, 0.f, 0.f);
      ag[j].x += d.x * hh.x; ag[j].y += d.y * hh.y; ag[j].z += d.z * hh.z; ag[j].w += d.w * hh.w;
      ab[j].x += d.x; ab[j].y += d.y; ab[j].z += d.z; ab[j].w += d.w;
      d.x *= g.x; d.y *= g.y; d.z *= g.z; d.w *= g.w;
      s1 += d.x + d.y + d.z + d.w;
      s2 += d.x * hh.x + d.y * hh.y + d.z * hh.z + d.w * hh.w;
      xh[j] = hh; gy[j] = d;
    }
    const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (j * 64 + lane) * 4;
      if (c < C) {
        float4 t;
        t.x = rstd * (gy[j].x - m1 - xh[j].x * m2); t.y = rstd * (gy[j].y - m1 - xh[j].y * m2);
        t.z = rstd * (gy[j].z - m1 - xh[j].z * m2); t.w = rstd * (gy[j].w - m1 - xh[j].w * m2);
        t.x += cur.ev[j].x; t.y += cur.ev[j].y; t.z += cur.ev[j].z; t.w += cur.ev[j].w;
        const float4 sc = scv[j], rv = cur.rv[j];
        as[j].x += t.x * rv.x * fr; as[j].y += t.y * rv.y * fr; as[j].z += t.z * rv.z * fr; as[j].w += t.w * rv.w * fr;
        if (da) *reinterpret_cast<float4*>(da + (size_t)row * C + c) = make_float4(t.x * fa, t.y * fa, t.z * fa, t.w * fa);
        if (dr) *reinterpret_cast<float4*>(dr + (size_t)row * C + c) =
            make_float4(t.x * sc.x * fr, t.y * sc.y * fr, t.z * sc.z * fr, t.w * sc.w * fr);
      }
    }
    cur = nxt;
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    if (c < C) {
      *reinterpret_cast<float4*>(red + (size_t)(wid * 3 + 0) * C + c) = as[j];
      *reinterpret_cast<float4*>(red + (size_t)(wid * 3 + 1) * C + c) = ag[j];
      *reinterpret_cast<float4*>(red + (size_t)(wid * 3 + 2) * C + c) = ab[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += 64 * NW) {
    const int which = i / C, c = i % C;
    float s = 0.f;
#pragma unroll
    for (int w_ = 0; w_ < NW; ++w_) s += red[(size_t)(w_ * 3 + which) * C + c];
    if (dg_at) atomicAdd((which == 0 ? dscale_at : (which == 1 ? dg_at : db_at)) + c, s);
    else ws[(size_t)blockIdx.x * 3 * C + i] = s;
  }
}

static int ln_bwd_waves(int C) { return C <= 256 ? 16 : (C <= 512 ? 8 : 4); }
static int ln_bwd_blocks(int M, int C) {
  static const char* e = getenv("AVDF_LN_BWD_BLOCKS");      // tuning override for tools/pw_bench.py
  const int cap = e ? atoi(e) : kLnBwdBlocks;
  int g = cdiv(M, ln_bwd_waves(C));
  return g < cap ? g : cap;
}

extern "C" int64_t avdf_layernorm_bwd_ws(int M, int C) { return (int64_t)2 * ln_bwd_blocks(M, C) * C; }

static int layernorm_bwd_impl(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                              const float* rstd, float* dx, float* dgamma, float* dbeta, float* ws, int M, int C, int relu,
                              int accumulate, const float* rowscale, void* stream);

extern "C" int avdf_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* beta,
                                  const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta,
                                  float* ws, int M, int C, int relu, int accumulate, void* stream) {
  return layernorm_bwd_impl(dy, x, gamma, beta, mean, rstd, dx, dgamma, dbeta, ws, M, C, relu, accumulate, nullptr, stream);
}

// backward of avdf_layernorm_fwd_rs: the incoming gradient is scaled by rowscale[row] on load
extern "C" int avdf_layernorm_bwd_rs(const float* dy, const float* x, const float* gamma, const float* beta,
                                     const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta,
                                     float* ws, int M, int C, int relu, int accumulate, const float* rowscale, void* stream) {
  AVDF_REQUIRE(rowscale, "layernorm_bwd_rs: null rowscale");
  return layernorm_bwd_impl(dy, x, gamma, beta, mean, rstd, dx, dgamma, dbeta, ws, M, C, relu, accumulate, rowscale, stream);
}

static int layernorm_bwd_impl(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                              const float* rstd, float* dx, float* dgamma, float* dbeta, float* ws, int M, int C, int relu,
                              int accumulate, const float* rowscale, void* stream) {
  AVDF_REQUIRE(dy && x && mean && rstd && dx && ws && M > 0 && C > 0, "layernorm_bwd: bad args");
  AVDF_REQUIRE(relu == 0 || relu == 1, "layernorm_bwd: only the none / ReLU epilogues have a backward");
  AVDF_REQUIRE(C % 4 == 0 && C <= 256 * kLnMaxV, "layernorm_bwd: C must be a multiple of 4 and <= %d", 256 * kLnMaxV);
  AVDF_REQUIRE((gamma == nullptr) == (beta == nullptr), "layernorm_bwd: gamma/beta");
  AVDF_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(dx) && (!gamma || (aligned16(gamma) && aligned16(beta))), "layernorm_bwd: 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  const int G = ln_bwd_blocks(M, C);
  // accumulate: 0 overwrite, 1 add by float atomics inside the kernel, 2 leave the workgroups' rows [G][dgamma | dbeta] in ws
  // for the caller to ADD (avdf_reduce_segments_batched; G = avdf_layernorm_bwd_ws / 2C)
  const bool atomic = accumulate == 1 && dgamma && dbeta;
  float* dga = atomic ? dgamma : nullptr;
  float* dba = atomic ? dbeta : nullptr;
#define LN_BWD(NV_, NW_) hipLaunchKernelGGL((ln_bwd_kernel<NV_, NW_>), dim3(G), dim3(64 * NW_),                 \
                                           (size_t)2 * NW_ * C * sizeof(float), st, dy, x, gamma, beta, mean, rstd, \
                                           dx, ws, dga, dba, M, C, relu, rowscale)
  if (C <= 256) LN_BWD(1, 16); else if (C <= 512) LN_BWD(2, 8); else if (C <= 1024) LN_BWD(4, 4); else LN_BWD(8, 4);
#undef LN_BWD
  AVDF_CHECK_LAUNCH("layernorm_bwd");
  if (dgamma && !atomic && accumulate != 2) {
    AVDF_REQUIRE(dbeta, "layernorm_bwd: dgamma and dbeta must both be requested");
    int rc = avdf_reduce_slabs2(ws, dgamma, dbeta, G, 2 * (int64_t)C, C, accumulate, st);
    if (rc) return rc;
  }
  return AVDF_OK;
}

extern "C" int avdf_instnorm_lrelu_fwd(const float* x, float* y, float* mean, float* rstd,
                                       int B, int T, int C, float eps, float slope, void* stream) {
  AVDF_REQUIRE(x && y && B > 0 && T > 0 && C > 0, "instnorm_lrelu_fwd: bad args");
  AVDF_REQUIRE((mean == nullptr) == (rstd == nullptr), "instnorm_lrelu_fwd: mean/rstd");
  hipLaunchKernelGGL(instnorm_lrelu_fwd_kernel, dim3(B * cdiv(C, 64)), dim3(64 * kInG), 0, (hipStream_t)stream,
                     x, y, mean, rstd, T, C, eps, slope);
  AVDF_CHECK_LAUNCH("instnorm_lrelu_fwd");
  return AVDF_OK;
}

extern "C" int avdf_instnorm_lrelu_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                                       float* dx, int B, int T, int C, float slope, void* stream) {
  AVDF_REQUIRE(dy && x && mean && rstd && dx && B > 0 && T > 0 && C > 0, "instnorm_lrelu_bwd: bad args");
  hipLaunchKernelGGL(instnorm_lrelu_bwd_kernel, dim3(B * cdiv(C, 64)), dim3(64 * kInG), 0, (hipStream_t)stream,
                     dy, x, mean, rstd, dx, T, C, slope);
  AVDF_CHECK_LAUNCH("instnorm_lrelu_bwd");
  return AVDF_OK;
}

extern "C" int avdf_maxavgpool_fwd(const float* x, float* y, int B, int T, int C, void* stream) {
  AVDF_REQUIRE(x && y && B > 0 && T > 0 && C > 0, "maxavgpool_fwd: bad args");
  hipLaunchKernelGGL(maxavgpool_fwd_kernel, dim3(B * cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, x, y, T, C);
  AVDF_CHECK_LAUNCH("maxavgpool_fwd");
  return AVDF_OK;
}

extern "C" int avdf_maxavgpool_bwd(const float* dy, const float* x, float* dx, int B, int T, int C, void* stream) {
  AVDF_REQUIRE(dy && x && dx && B > 0 && T > 0 && C > 0, "maxavgpool_bwd: bad args");
  hipLaunchKernelGGL(maxavgpool_bwd_kernel, dim3(B * cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, T, C);
  AVDF_CHECK_LAUNCH("maxavgpool_bwd");
  return AVDF_OK;
}

extern "C" int avdf_residual_ln_fwd(const float* a, const float* ma, const float* r, const float* mr, const float* scale,
                                    const float* keep, const float* gamma, const float* beta, float* out, float* h,
                                    float* mean, float* rstd, int B, int T, int C, float eps, void* stream) {
  AVDF_REQUIRE(a && r && scale && gamma && beta && out && h && mean && rstd && B > 0 && T > 0 && C > 0, "residual_ln_fwd: bad args");
  AVDF_REQUIRE(C % 4 == 0 && C <= 256 * kLnMaxV, "residual_ln_fwd: C must be a multiple of 4 and <= %d", 256 * kLnMaxV);
  AVDF_REQUIRE(aligned16(a) && aligned16(r) && aligned16(scale) && aligned16(gamma) && aligned16(beta) && aligned16(out) && aligned16(h),
               "residual_ln_fwd: 16-byte alignment");
  const int M = B * T;
#define RLN_FWD(NV_) hipLaunchKernelGGL(residual_ln_fwd_kernel<NV_>, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, \
                                       a, ma, r, mr, scale, keep, gamma, beta, out, h, mean, rstd, M, T, C, eps)
  if (C <= 256) RLN_FWD(1); else if (C <= 512) RLN_FWD(2); else if (C <= 1024) RLN_FWD(4); else RLN_FWD(8);
#undef RLN_FWD
  AVDF_CHECK_LAUNCH("residual_ln_fwd");
  return AVDF_OK;
}

extern "C" int64_t avdf_residual_ln_bwd_ws(int B, int T, int C) { return (int64_t)3 * ln_bwd_blocks(B * T, C) * C; }

extern "C" int avdf_residual_ln_bwd(const float* dout, const float* dh, const float* out, const float* mean,
                                    const float* rstd, const float* gamma, const float* ma, const float* r,
                                    const float* mr, const float* scale, const float* keep, float* da, float* dr,
                                    float* dscale, float* dgamma, float* dbeta, float* ws, int B, int T, int C,
                                    int accumulate, void* stream) {
  AVDF_REQUIRE(dh && out && mean && rstd && gamma && r && scale && dscale && dgamma && dbeta && ws && B > 0 && T > 0 && C > 0,
               "residual_ln_bwd: bad args");
  AVDF_REQUIRE(C % 4 == 0 && C <= 256 * kLnMaxV, "residual_ln_bwd: C must be a multiple of 4 and <= %d", 256 * kLnMaxV);
  AVDF_REQUIRE(aligned16(dh) && aligned16(out) && aligned16(gamma) && aligned16(r) && aligned16(scale) && aligned16(dout) &&
                   aligned16(da) && aligned16(dr), "residual_ln_bwd: 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  const int M = B * T, G = ln_bwd_blocks(M, C);
  // accumulate: 0 overwrite (workspace rows + reduction), 1 add by float atomics inside the kernel, 2 workspace rows only --
  // [rows = ws floats / 3C][dscale | dgamma | dbeta], to be ADDED by the caller (avdf_reduce_segments_batched)
  float* sa = accumulate == 1 ? dscale : nullptr;
  float* ga = accumulate == 1 ? dgamma : nullptr;
  float* ba = accumulate == 1 ? dbeta : nullptr;
#define RLN_BWD(NV_, NW_) hipLaunchKernelGGL((residual_ln_bwd_kernel<NV_, NW_>), dim3(G), dim3(64 * NW_),              \
                                            (size_t)3 * NW_ * C * sizeof(float), st, dout, dh, out, mean, rstd, gamma,  \
                                            ma, r, mr, scale, keep, da, dr, ws, sa, ga, ba, M, T, C)
  if (C <= 256) RLN_BWD(1, 16); else if (C <= 512) RLN_BWD(2, 8); else if (C <= 1024) RLN_BWD(4, 4); else RLN_BWD(8, 4);
#undef RLN_BWD
  AVDF_CHECK_LAUNCH("residual_ln_bwd");
  if (accumulate == 0) {                                                  // ws rows: [dscale | dgamma | dbeta]
    float* dst[3] = {dscale, dgamma, dbeta};
    const int len[3] = {C, C, C}, nsrc[3] = {1, 1, 1};
    const int off[9] = {0, 0, 0, C, 0, 0, 2 * C, 0, 0};
    return avdf_reduce_segments(ws, G, 3 * C, 3, dst, len, nsrc, off, 0, stream);
  }
  return AVDF_OK;
}

// HBM-bound glue kernels of the transformer blocks / FPN (all float4-vectorised over the
// contiguous channel axis of the time-major layout):
//   depthwise conv k3 (blocks.py:934-952, necks.py:48-51), MaxPool1d(3,2,1) skip (blocks.py:1277-1281),
//   residual + AffineDropPath (blocks.py:1311-1313,1397-1439), nearest resampling
//   (backbones.py:487,490; necks.py:78-80), [B,C,T] <-> [B,T,C] transpose at the plugin boundary.
#include "common.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4(float a) { return make_float4(a, a, a, a); }
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 scl4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }

constexpr int kPartialBlocks = 128;
// The fused (atomic) backward kernels: float atomics cost ~30 ns each on this part, so the per-channel sums are
// first folded over a 1024-thread workgroup in LDS and at most kFusedBlocks workgroups touch the gradient
// (measured, B=8 T=768 C=256: 768 x 256-thread blocks 25 us -> 128 x 1024 <10 us).
constexpr int kFusedThreads = 1024;
static int fused_blocks(long elems4) {
  static const char* e = getenv("AVDF_FUSED_BLOCKS");      // tuning override for tools/pw_bench.py
  const long cap = e ? atoi(e) : 128;
  const long g = (elems4 + kFusedThreads - 1) / kFusedThreads;
  const long c2 = cap > 512 ? 512 : cap;                    // workspaces of the row modes hold at most 512 rows
  return (int)(g < 1 ? 1 : (g > c2 ? c2 : g));
}

// weights of 4 consecutive channels: w[c..c+3][0..2] = 12 contiguous floats -> per-tap float4
__device__ __forceinline__ void load_dw_taps(const float* __restrict__ w, int c, float4 tap[3]) {
  const float4 a = ld4(w + c * 3), b = ld4(w + c * 3 + 4), d = ld4(w + c * 3 + 8);
  tap[0] = make_float4(a.x, a.w, b.z, d.y);
  tap[1] = make_float4(a.y, b.x, b.w, d.z);
  tap[2] = make_float4(a.z, b.y, d.x, d.w);
}

__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ mask, float* __restrict__ y,
                                                         int B, int Tin, int Tout, int C, int S) {
  const int C4 = C >> 2;
  const long total = (long)B * Tout * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const long bt = i / C4;
    const int t = (int)(bt % Tout);
    const long b = bt / Tout;
    float4 tap[3];
    load_dw_taps(w, c, tap);
    float4 acc = f4(0.f);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int ti = t * S + k - 1;
      if (ti >= 0 && ti < Tin) acc = add4(acc, mul4(ld4(x + (b * Tin + ti) * C + c), tap[k]));
    }
    if (mask) acc = scl4(acc, mask[bt]);
    st4(y + bt * C + c, acc);
  }
}

__global__ __launch_bounds__(256) void dwconv_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                              const float* __restrict__ mask, float* __restrict__ dx,
                                                              int B, int Tin, int Tout, int C, int S) {
  const int C4 = C >> 2;
  const long total = (long)B * Tin * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const long bt = i / C4;
    const int tp = (int)(bt % Tin);
    const long b = bt / Tin;
    float4 tap[3];
    load_dw_taps(w, c, tap);
    float4 acc = f4(0.f);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int num = tp + 1 - k;
      if (num >= 0 && num % S == 0) {
        const int t = num / S;
        if (t < Tout) {
          float4 g = ld4(dy + (b * Tout + t) * C + c);
          if (mask) g = scl4(g, mask[b * Tout + t]);
          acc = add4(acc, mul4(g, tap[k]));
        }
      }
    }
    st4(dx + bt * C + c, acc);
  }
}

// partial[blockIdx.x][c*3+k] = sum over this block's rows of dy*mask*x_shifted
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ mask, float* __restrict__ partial,
                                                                int B, int Tin, int Tout, int C, int S) {
  __shared__ float red[4][64 * 12];
  const int cg = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const long M = (long)B * Tout;
  for (int cb = 0; cb < C; cb += 256) {
    const int c = cb + cg * 4;
    float4 a0 = f4(0.f), a1 = f4(0.f), a2 = f4(0.f);
    if (c < C) {
      for (long r = (long)blockIdx.x * 4 + rg; r < M; r += (long)gridDim.x * 4) {
        const int t = (int)(r % Tout);
        const long b = r / Tout;
        float4 g = ld4(dy + r * C + c);
        if (mask) g = scl4(g, mask[r]);
        const int t0 = t * S - 1;
        if (t0 >= 0) a0 = add4(a0, mul4(g, ld4(x + (b * Tin + t0) * C + c)));
        if (t0 + 1 < Tin) a1 = add4(a1, mul4(g, ld4(x + (b * Tin + t0 + 1) * C + c)));
        if (t0 + 2 < Tin) a2 = add4(a2, mul4(g, ld4(x + (b * Tin + t0 + 2) * C + c)));
      }
    }
    float* d = &red[rg][cg * 12];      // [channel-in-quad][tap]
    d[0] = a0.x; d[1] = a1.x; d[2] = a2.x; d[3] = a0.y; d[4] = a1.y; d[5] = a2.y;
    d[6] = a0.z; d[7] = a1.z; d[8] = a2.z; d[9] = a0.w; d[10] = a1.w; d[11] = a2.w;
    __syncthreads();
    for (int e = threadIdx.x; e < 768; e += 256) {
      const int idx = cb * 3 + e;
      if (idx < C * 3) partial[(size_t)blockIdx.x * C * 3 + idx] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
    }
    __syncthreads();
  }
}

// Fused backward for the accumulate path: dx by gather, dw by per-thread partial sums (a thread keeps
// the same channel quad over the whole grid-stride loop because gridDim*256 is a multiple of C/4),
// block reduction over the row groups in LDS, then one float atomic per (channel, tap) and block.
__global__ __launch_bounds__(kFusedThreads) void dwconv_bwd_fused_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ w, const float* __restrict__ mask,
                                                               float* __restrict__ dx, float* __restrict__ dw,
                                                               int B, int Tin, int Tout, int C, int S) {
  __shared__ float red[kFusedThreads * 12];
  const int C4 = C >> 2;
  const int c = (int)(((long)blockIdx.x * kFusedThreads + threadIdx.x) % C4) * 4;
  float4 tap[3];
  load_dw_taps(w, c, tap);
  if (dx) {
    const long total = (long)B * Tin * C4;
    for (long i = (long)blockIdx.x * kFusedThreads + threadIdx.x; i < total; i += (long)gridDim.x * kFusedThreads) {
      const long bt = i / C4;
      const int tp = (int)(bt % Tin);
      const long b = bt / Tin;
      float4 acc = f4(0.f);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int num = tp + 1 - k;
        if (num >= 0 && num % S == 0) {
          const int t = num / S;
          if (t < Tout) {
            float4 g = ld4(dy + (b * Tout + t) * C + c);
            if (mask) g = scl4(g, mask[b * Tout + t]);
            acc = add4(acc, mul4(g, tap[k]));
          }
        }
      }
      st4(dx + bt * C + c, acc);
    }
  }
  float4 a0 = f4(0.f), a1 = f4(0.f), a2 = f4(0.f);
  const long rows = (long)B * Tout;
  for (long i = (long)blockIdx.x * kFusedThreads + threadIdx.x; i < rows * C4; i += (long)gridDim.x * kFusedThreads) {
    const long r = i / C4;
    const int t = (int)(r % Tout);
    const long b = r / Tout;
    float4 g = ld4(dy + r * C + c);
    if (mask) g = scl4(g, mask[r]);
    const int t0 = t * S - 1;
    if (t0 >= 0) a0 = add4(a0, mul4(g, ld4(x + (b * Tin + t0) * C + c)));
    if (t0 + 1 < Tin) a1 = add4(a1, mul4(g, ld4(x + (b * Tin + t0 + 1) * C + c)));
    if (t0 + 2 < Tin) a2 = add4(a2, mul4(g, ld4(x + (b * Tin + t0 + 2) * C + c)));
  }
  float* d = red + threadIdx.x * 12;
  d[0] = a0.x; d[1] = a1.x; d[2] = a2.x; d[3] = a0.y; d[4] = a1.y; d[5] = a2.y;
  d[6] = a0.z; d[7] = a1.z; d[8] = a2.z; d[9] = a0.w; d[10] = a1.w; d[11] = a2.w;
  __syncthreads();
  // threads tid, tid+C4, tid+2*C4 ... of this block share a channel quad (C4 divides the block size)
  const int per = C4 * 12;
  for (int e = threadIdx.x; e < per; e += kFusedThreads) {
    float s_ = 0.f;
    for (int q = e; q < kFusedThreads * 12; q += per) s_ += red[q];
    const int quad = ((blockIdx.x * kFusedThreads) % C4 + e / 12) % C4;
    atomicAdd(dw + quad * 12 + (e % 12), s_);
  }
}

// Fused backward of out = a*ma + scale*keep*(r*mr) for the accumulate path: da, dr and an atomically
// accumulated dscale (same per-thread channel-quad invariant as above).
__global__ __launch_bounds__(kFusedThreads) void affine_residual_bwd_fused_kernel(
    const float* __restrict__ dout, const float* __restrict__ ma, const float* __restrict__ r,
    const float* __restrict__ mr, const float* __restrict__ scale, const float* __restrict__ keep,
    float* __restrict__ da, float* __restrict__ dr, float* __restrict__ dscale, float* __restrict__ ws_rows, int B, int T, int C) {
  __shared__ float red[kFusedThreads * 4];
  const int C4 = C >> 2;
  const int c = (int)(((long)blockIdx.x * kFusedThreads + threadIdx.x) % C4) * 4;
  const float4 sc = ld4(scale + c);
  float4 ds = f4(0.f);
  const long total = (long)B * T * C4;
  // a thread walks total / (128 x 1024) elements one after the other: the loads of the next element are in flight while
  // this one is stored (the kernel's duration is its elements' memory round trips)
  const long step = (long)gridDim.x * kFusedThreads;
  long i = (long)blockIdx.x * kFusedThreads + threadIdx.x;
  float4 g = f4(0.f), rr = f4(0.f), gn = f4(0.f), rn = f4(0.f);
  float fa = 1.f, f = 1.f, fan = 1.f, fn = 1.f;
  auto fetch = [&](long e, float4& g_, float4& r_, float& fa_, float& f_) {
    const long bt = e / C4;
    g_ = ld4(dout + bt * C + c);
    r_ = ld4(r + bt * C + c);
    fa_ = ma ? ma[bt] : 1.f;
    f_ = keep ? keep[bt / T] : 1.f;
    if (mr) f_ *= mr[bt];
  };
  if (i < total) fetch(i, g, rr, fa, f);
  for (; i < total; i += step) {
    if (i + step < total) fetch(i + step, gn, rn, fan, fn);
    const long bt = i / C4;
    if (da) st4(da + bt * C + c, ma ? scl4(g, fa) : g);
    if (dr) st4(dr + bt * C + c, scl4(mul4(g, sc), f));
    ds = add4(ds, scl4(mul4(g, rr), f));
    g = gn; rr = rn; fa = fan; f = fn;
  }
  float* d = red + threadIdx.x * 4;
  d[0] = ds.x; d[1] = ds.y; d[2] = ds.z; d[3] = ds.w;
  __syncthreads();
  const int per = C4 * 4;
  for (int e = threadIdx.x; e < per; e += kFusedThreads) {
    float s_ = 0.f;
    for (int q = e; q < kFusedThreads * 4; q += per) s_ += red[q];
    const int quad = ((blockIdx.x * kFusedThreads) % C4 + e / 4) % C4;
    if (ws_rows) ws_rows[(size_t)blockIdx.x * C + quad * 4 + (e % 4)] = s_;      // a row per workgroup, added by the caller
    else atomicAdd(dscale + quad * 4 + (e % 4), s_);
  }
}

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          int B, int Tin, int Tout, int C) {
  const int C4 = C >> 2;
  const long total = (long)B * Tout * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const long bt = i / C4;
    const int t = (int)(bt % Tout);
    const long b = bt / Tout;
    float4 m = f4(-INFINITY);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int ti = 2 * t + k - 1;
      if (ti >= 0 && ti < Tin) {
        const float4 v = ld4(x + (b * Tin + ti) * C + c);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    st4(y + bt * C + c, m);
  }
}

__device__ __forceinline__ int window_argmax(const float* __restrict__ x, long b, int t, int Tin, int C, int c) {
  float best = -INFINITY; int am = -1;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int ti = 2 * t + k - 1;
    if (ti >= 0 && ti < Tin) {
      const float v = x[(b * Tin + ti) * C + c];
      if (v > best || am < 0) { best = v; am = ti; }
    }
  }
  return am;
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          float* __restrict__ dx, int B, int Tin, int Tout, int C) {
  const long total = (long)B * Tin * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long bt = i / C;
    const int tp = (int)(bt % Tin);
    const long b = bt / Tin;
    float acc = 0.f;
    const int tlo = tp / 2, thi = (tp + 1) / 2;        // windows 2t-1..2t+1 containing tp
    if (tlo < Tout && window_argmax(x, b, tlo, Tin, C, c) == tp) acc += dy[(b * Tout + tlo) * C + c];
    if (thi != tlo && thi < Tout && window_argmax(x, b, thi, Tin, C, c) == tp) acc += dy[(b * Tout + thi) * C + c];
    dx[i] = acc;
  }
}

__global__ __launch_bounds__(256) void affine_residual_fwd_kernel(
    const float* __restrict__ a, const float* __restrict__ ma, const float* __restrict__ r, const float* __restrict__ mr,
    const float* __restrict__ scale, const float* __restrict__ keep, float* __restrict__ out, int B, int T, int C) {
  const int C4 = C >> 2;
  const long total = (long)B * T * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const long bt = i / C4;
    const long b = bt / T;
    float4 av = ld4(a + bt * C + c);
    if (ma) av = scl4(av, ma[bt]);
    float f = keep ? keep[b] : 1.f;
    if (mr) f *= mr[bt];
    const float4 rv = scl4(mul4(ld4(r + bt * C + c), ld4(scale + c)), f);
    st4(out + bt * C + c, add4(av, rv));
  }
}

__global__ __launch_bounds__(256) void affine_residual_bwd_kernel(
    const float* __restrict__ dout, const float* __restrict__ ma, const float* __restrict__ mr,
    const float* __restrict__ scale, const float* __restrict__ keep, float* __restrict__ da, float* __restrict__ dr,
    int B, int T, int C) {
  const int C4 = C >> 2;
  const long total = (long)B * T * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const long bt = i / C4;
    const long b = bt / T;
    const float4 g = ld4(dout + bt * C + c);
    if (da) st4(da + bt * C + c, ma ? scl4(g, ma[bt]) : g);
    float f = keep ? keep[b] : 1.f;
    if (mr) f *= mr[bt];
    if (dr) st4(dr + bt * C + c, scl4(mul4(g, ld4(scale + c)), f));
  }
}

// partial[blockIdx.x][c] = sum_rows dout * r * mr * keep
__global__ __launch_bounds__(256) void affine_residual_dscale_kernel(
    const float* __restrict__ dout, const float* __restrict__ r, const float* __restrict__ mr,
    const float* __restrict__ keep, float* __restrict__ partial, int B, int T, int C) {
  __shared__ float red[4][256];
  const int cg = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const long M = (long)B * T;
  for (int cb = 0; cb < C; cb += 256) {
    const int c = cb + cg * 4;
    float4 s = f4(0.f);
    if (c < C) {
      for (long row = (long)blockIdx.x * 4 + rg; row < M; row += (long)gridDim.x * 4) {
        float f = keep ? keep[row / T] : 1.f;
        if (mr) f *= mr[row];
        s = add4(s, scl4(mul4(ld4(dout + row * C + c), ld4(r + row * C + c)), f));
      }
    }
    red[rg][cg * 4 + 0] = s.x; red[rg][cg * 4 + 1] = s.y; red[rg][cg * 4 + 2] = s.z; red[rg][cg * 4 + 3] = s.w;
    __syncthreads();
    if (cb + threadIdx.x < C)
      partial[(size_t)blockIdx.x * C + cb + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    __syncthreads();
  }
}

// mode 0: y[b,t] = x[b,t/r]            (Tout = Tin*r)   upsample fwd
// mode 1: dx[b,t'] = sum_j dy[b,t'*r+j] (Tout = Tin)     upsample bwd   (src has Tin*r rows)
// mode 2: y[b,t] = x[b,t*r]            (Tout = Tin/r)   downsample fwd
// mode 3: dx[b,t'] = t'%r==0 ? dy[b,t'/r] : 0 (Tout=Tin) downsample bwd (src has Tin/r rows)
// mode 4: out = a + small[b,t/r]
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ src, const float* __restrict__ a,
                                                       float* __restrict__ dst, int B, int Tdst, int Tsrc, int C, int r, int mode) {
  const int C4 = C >> 2;
  const long total = (long)B * Tdst * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const long bt = i / C4;
    const int t = (int)(bt % Tdst);
    const long b = bt / Tdst;
    float4 v;
    if (mode == 0) v = ld4(src + (b * Tsrc + t / r) * C + c);
    else if (mode == 1) {
      v = f4(0.f);
      for (int j = 0; j < r; ++j) v = add4(v, ld4(src + (b * Tsrc + (long)t * r + j) * C + c));
    } else if (mode == 2) v = ld4(src + (b * Tsrc + (long)t * r) * C + c);
    else if (mode == 3) v = (t % r == 0) ? ld4(src + (b * Tsrc + t / r) * C + c) : f4(0.f);
    else v = add4(ld4(a + bt * C + c), ld4(src + (b * Tsrc + t / r) * C + c));
    st4(dst + bt * C + c, v);
  }
}

// y[row, c] = (x[row, c] + bias[c]) * mask[row]   (MaskedConvTranspose1D epilogue, blocks.py:1477-1489)
__global__ __launch_bounds__(256) void bias_mask_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                        const float* __restrict__ mask, float* __restrict__ y, long M, int C) {
  const int C4 = C >> 2;
  const long total = M * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const long row = i / C4;
    float4 v = ld4(x + row * C + c);
    if (bias) v = add4(v, ld4(bias + c));
    if (mask) v = scl4(v, mask[row]);
    st4(y + row * C + c, v);
  }
}

// partial[blockIdx.x][c] = sum over this block's rows of dy[row, c] * mask[row]
__global__ __launch_bounds__(256) void masked_colsum_kernel(const float* __restrict__ dy, const float* __restrict__ mask,
                                                            float* __restrict__ partial, long M, int C) {
  __shared__ float red[4][256];
  const int cg = threadIdx.x & 63, rg = threadIdx.x >> 6;
  for (int cb = 0; cb < C; cb += 256) {
    const int c = cb + cg * 4;
    float4 s_ = f4(0.f);
    if (c < C) {
      for (long r = (long)blockIdx.x * 4 + rg; r < M; r += (long)gridDim.x * 4) {
        float4 g = ld4(dy + r * C + c);
        if (mask) g = scl4(g, mask[r]);
        s_ = add4(s_, g);
      }
    }
    red[rg][cg * 4 + 0] = s_.x; red[rg][cg * 4 + 1] = s_.y; red[rg][cg * 4 + 2] = s_.z; red[rg][cg * 4 + 3] = s_.w;
    __syncthreads();
    if (cb + threadIdx.x < C)
      partial[(size_t)blockIdx.x * C + cb + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                         float* __restrict__ y, long M, int C) {
  const long total = M * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) y[i] = x[i] * s[i / C];
}

// y[b, j, i] = x[b, i, j] for x [B, R, Cc]
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int Cc) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int j0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int i = i0 + k, j = j0 + tx;
    if (i < R && j < Cc) tile[k][tx] = x[((size_t)b * R + i) * Cc + j];
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int j = j0 + k, i = i0 + tx;
    if (i < R && j < Cc) y[((size_t)b * Cc + j) * R + i] = tile[tx][k];
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int grid_for(long total) {
  long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

#define REQ_C4(name, C) AVDF_REQUIRE((C) % 4 == 0, name ": C %% 4 != 0 (C=%d)", C)

extern "C" int avdf_dwconv_fwd(const float* x, const float* w, const float* mask, float* y,
                               int B, int Tin, int Tout, int C, int stride, void* stream) {
  AVDF_REQUIRE(x && w && y && B > 0 && Tin > 0 && C > 0, "dwconv_fwd: bad args");
  REQ_C4("dwconv_fwd", C);
  AVDF_REQUIRE((stride == 1 || stride == 2) && Tout == (Tin - 1) / stride + 1, "dwconv_fwd: stride/Tout");
  AVDF_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y), "dwconv_fwd: 16-byte alignment");
  hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(grid_for((long)B * Tout * C / 4)), dim3(256), 0, (hipStream_t)stream,
                     x, w, mask, y, B, Tin, Tout, C, stride);
  AVDF_CHECK_LAUNCH("dwconv_fwd");
  return AVDF_OK;
}

extern "C" int64_t avdf_dwconv_bwd_ws(int B, int Tout, int C) { (void)B; (void)Tout; return (int64_t)kPartialBlocks * C * 3; }

extern "C" int avdf_dwconv_bwd(const float* dy, const float* x, const float* w, const float* mask,
                               float* dx, float* dw, float* ws, int B, int Tin, int Tout, int C, int stride,
                               int accumulate, void* stream) {
  AVDF_REQUIRE(dy && x && w && B > 0 && Tin > 0 && C > 0, "dwconv_bwd: bad args");
  REQ_C4("dwconv_bwd", C);
  AVDF_REQUIRE((stride == 1 || stride == 2) && Tout == (Tin - 1) / stride + 1, "dwconv_bwd: stride/Tout");
  AVDF_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(w) && (!dx || aligned16(dx)), "dwconv_bwd: 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  if (accumulate && dw && (kFusedThreads % (C / 4) == 0)) {
    // accumulate path (engine): one fused kernel, dw by float atomics straight into the gradient buffer
    hipLaunchKernelGGL(dwconv_bwd_fused_kernel, dim3(fused_blocks((long)B * Tin * C / 4)), dim3(kFusedThreads), 0, st,
                       dy, x, w, mask, dx, dw, B, Tin, Tout, C, stride);
    AVDF_CHECK_LAUNCH("dwconv_bwd_fused");
    return AVDF_OK;
  }
  if (dx) {
    hipLaunchKernelGGL(dwconv_bwd_data_kernel, dim3(grid_for((long)B * Tin * C / 4)), dim3(256), 0, st,
                       dy, w, mask, dx, B, Tin, Tout, C, stride);
    AVDF_CHECK_LAUNCH("dwconv_bwd_data");
  }
  if (dw) {
    AVDF_REQUIRE(ws, "dwconv_bwd: ws required for dw");
    hipLaunchKernelGGL(dwconv_bwd_weight_kernel, dim3(kPartialBlocks), dim3(256), 0, st, dy, x, mask, ws, B, Tin, Tout, C, stride);
    AVDF_CHECK_LAUNCH("dwconv_bwd_weight");
    int rc = avdf_reduce_slabs(ws, dw, kPartialBlocks, (int64_t)C * 3, accumulate, st);
    if (rc) return rc;
  }
  return AVDF_OK;
}

extern "C" int avdf_maxpool3s2_fwd(const float* x, float* y, int B, int Tin, int C, void* stream) {
  AVDF_REQUIRE(x && y && B > 0 && Tin > 0 && C > 0, "maxpool_fwd: bad args");
  REQ_C4("maxpool_fwd", C);
  AVDF_REQUIRE(aligned16(x) && aligned16(y), "maxpool_fwd: 16-byte alignment");
  const int Tout = (Tin - 1) / 2 + 1;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((long)B * Tout * C / 4)), dim3(256), 0, (hipStream_t)stream, x, y, B, Tin, Tout, C);
  AVDF_CHECK_LAUNCH("maxpool_fwd");
  return AVDF_OK;
}

extern "C" int avdf_maxpool3s2_bwd(const float* dy, const float* x, float* dx, int B, int Tin, int C, void* stream) {
  AVDF_REQUIRE(dy && x && dx && B > 0 && Tin > 0 && C > 0, "maxpool_bwd: bad args");
  const int Tout = (Tin - 1) / 2 + 1;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((long)B * Tin * C)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, B, Tin, Tout, C);
  AVDF_CHECK_LAUNCH("maxpool_bwd");
  return AVDF_OK;
}

extern "C" int avdf_affine_residual_fwd(const float* a, const float* ma, const float* r, const float* mr,
                                        const float* scale, const float* keep, float* out,
                                        int B, int T, int C, void* stream) {
  AVDF_REQUIRE(a && r && scale && out && B > 0 && T > 0 && C > 0, "affine_residual_fwd: bad args");
  REQ_C4("affine_residual_fwd", C);
  AVDF_REQUIRE(aligned16(a) && aligned16(r) && aligned16(scale) && aligned16(out), "affine_residual_fwd: 16-byte alignment");
  hipLaunchKernelGGL(affine_residual_fwd_kernel, dim3(grid_for((long)B * T * C / 4)), dim3(256), 0, (hipStream_t)stream,
                     a, ma, r, mr, scale, keep, out, B, T, C);
  AVDF_CHECK_LAUNCH("affine_residual_fwd");
  return AVDF_OK;
}

constexpr int kFusedRowsMax = 512;       // upper bound of fused_blocks() the workspace is sized for
extern "C" int64_t avdf_affine_residual_bwd_ws(int B, int T, int C) { (void)B; (void)T; return (int64_t)kFusedRowsMax * C; }
// rows of [C] dscale partials that accumulate == 2 leaves in ws (0: that mode is not available for this width)
extern "C" int avdf_affine_residual_bwd_rows(int B, int T, int C) {
  return (C % 4 == 0 && kFusedThreads % (C / 4) == 0) ? fused_blocks((long)B * T * C / 4) : 0;
}

extern "C" int avdf_affine_residual_bwd(const float* dout, const float* ma, const float* r, const float* mr,
                                        const float* scale, const float* keep, float* da, float* dr,
                                        float* dscale, float* ws, int B, int T, int C, int accumulate, void* stream) {
  AVDF_REQUIRE(dout && r && scale && B > 0 && T > 0 && C > 0, "affine_residual_bwd: bad args");
  REQ_C4("affine_residual_bwd", C);
  AVDF_REQUIRE(aligned16(dout) && aligned16(r) && aligned16(scale) && (!da || aligned16(da)) && (!dr || aligned16(dr)), "affine_residual_bwd: 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  // accumulate: 0 overwrite, 1 add (float atomics inside the kernel), 2 leave one row [C] per workgroup in ws for the caller
  // to ADD (avdf_reduce_segments_batched; rows = avdf_affine_residual_bwd_rows)
  if (accumulate && dscale && (kFusedThreads % (C / 4) == 0)) {
    AVDF_REQUIRE(accumulate != 2 || ws, "affine_residual_bwd: ws required");
    hipLaunchKernelGGL(affine_residual_bwd_fused_kernel, dim3(fused_blocks((long)B * T * C / 4)), dim3(kFusedThreads), 0, st,
                       dout, ma, r, mr, scale, keep, da, dr, dscale, accumulate == 2 ? ws : nullptr, B, T, C);
    AVDF_CHECK_LAUNCH("affine_residual_bwd_fused");
    return AVDF_OK;
  }
  if (da || dr) {
    hipLaunchKernelGGL(affine_residual_bwd_kernel, dim3(grid_for((long)B * T * C / 4)), dim3(256), 0, st,
                       dout, ma, mr, scale, keep, da, dr, B, T, C);
    AVDF_CHECK_LAUNCH("affine_residual_bwd");
  }
  if (dscale) {
    AVDF_REQUIRE(ws, "affine_residual_bwd: ws required for dscale");
    hipLaunchKernelGGL(affine_residual_dscale_kernel, dim3(kPartialBlocks), dim3(256), 0, st, dout, r, mr, keep, ws, B, T, C);
    AVDF_CHECK_LAUNCH("affine_residual_dscale");
    int rc = avdf_reduce_slabs(ws, dscale, kPartialBlocks, C, accumulate, st);
    if (rc) return rc;
  }
  return AVDF_OK;
}

static int resample(const char* name, const float* src, const float* a, float* dst, int B, int Tdst, int Tsrc, int C, int r,
                    int mode, void* stream) {
  AVDF_REQUIRE(src && dst && B > 0 && Tdst > 0 && Tsrc > 0 && C > 0 && r > 0, "%s: bad args", name);
  AVDF_REQUIRE(C % 4 == 0, "%s: C %% 4 != 0", name);
  AVDF_REQUIRE(aligned16(src) && aligned16(dst) && (!a || aligned16(a)), "%s: 16-byte alignment", name);
  hipLaunchKernelGGL(resample_kernel, dim3(grid_for((long)B * Tdst * C / 4)), dim3(256), 0, (hipStream_t)stream,
                     src, a, dst, B, Tdst, Tsrc, C, r, mode);
  AVDF_CHECK_LAUNCH(name);
  return AVDF_OK;
}

extern "C" int avdf_upsample_fwd(const float* x, float* y, int B, int Tin, int C, int r, void* stream) {
  return resample("upsample_fwd", x, nullptr, y, B, Tin * r, Tin, C, r, 0, stream);
}
extern "C" int avdf_upsample_bwd(const float* dy, float* dx, int B, int Tin, int C, int r, void* stream) {
  return resample("upsample_bwd", dy, nullptr, dx, B, Tin, Tin * r, C, r, 1, stream);
}
extern "C" int avdf_downsample_fwd(const float* x, float* y, int B, int Tin, int C, int r, void* stream) {
  AVDF_REQUIRE(r > 0 && Tin % r == 0, "downsample_fwd: Tin %% r != 0");
  return resample("downsample_fwd", x, nullptr, y, B, Tin / r, Tin, C, r, 2, stream);
}
extern "C" int avdf_downsample_bwd(const float* dy, float* dx, int B, int Tin, int C, int r, void* stream) {
  AVDF_REQUIRE(r > 0 && Tin % r == 0, "downsample_bwd: Tin %% r != 0");
  return resample("downsample_bwd", dy, nullptr, dx, B, Tin, Tin / r, C, r, 3, stream);
}
extern "C" int avdf_add_upsampled(const float* a, const float* small, float* out, int B, int T, int C, int r, void* stream) {
  AVDF_REQUIRE(a && r > 0 && T % r == 0, "add_upsampled: bad args");
  return resample("add_upsampled", small, a, out, B, T, T / r, C, r, 4, stream);
}

extern "C" int avdf_bias_mask_fwd(const float* x, const float* bias, const float* mask, float* y, int64_t M, int C, void* stream) {
  AVDF_REQUIRE(x && y && M > 0 && C > 0, "bias_mask_fwd: bad args");
  REQ_C4("bias_mask_fwd", C);
  AVDF_REQUIRE(aligned16(x) && aligned16(y) && (!bias || aligned16(bias)), "bias_mask_fwd: 16-byte alignment");
  hipLaunchKernelGGL(bias_mask_kernel, dim3(grid_for((long)M * C / 4)), dim3(256), 0, (hipStream_t)stream, x, bias, mask, y, (long)M, C);
  AVDF_CHECK_LAUNCH("bias_mask_fwd");
  return AVDF_OK;
}

extern "C" int64_t avdf_masked_colsum_ws(int C) { return (int64_t)kPartialBlocks * C; }

// out[c] (+)= sum_rows dy[row, c] * mask[row]   (bias gradient of the epilogue above)
extern "C" int avdf_masked_colsum(const float* dy, const float* mask, float* out, float* ws, int64_t M, int C,
                                  int accumulate, void* stream) {
  AVDF_REQUIRE(dy && out && ws && M > 0 && C > 0, "masked_colsum: bad args");
  REQ_C4("masked_colsum", C);
  AVDF_REQUIRE(aligned16(dy), "masked_colsum: 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(masked_colsum_kernel, dim3(kPartialBlocks), dim3(256), 0, st, dy, mask, ws, (long)M, C);
  AVDF_CHECK_LAUNCH("masked_colsum");
  return avdf_reduce_slabs(ws, out, kPartialBlocks, C, accumulate, st);
}

extern "C" int avdf_scale_rows(const float* x, const float* s, float* y, int M, int C, void* stream) {
  AVDF_REQUIRE(x && s && y && M > 0 && C > 0, "scale_rows: bad args");
  hipLaunchKernelGGL(scale_rows_kernel, dim3(grid_for((long)M * C)), dim3(256), 0, (hipStream_t)stream, x, s, y, (long)M, C);
  AVDF_CHECK_LAUNCH("scale_rows");
  return AVDF_OK;
}

extern "C" int avdf_transpose(const float* x, float* y, int B, int R, int Cc, void* stream) {
  AVDF_REQUIRE(x && y && B > 0 && R > 0 && Cc > 0 && B < 65536, "transpose: bad args");
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(Cc, 32), cdiv(R, 32), B), dim3(256), 0, (hipStream_t)stream, x, y, R, Cc);
  AVDF_CHECK_LAUNCH("transpose");
  return AVDF_OK;
}

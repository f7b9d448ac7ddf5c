// Narrow-output conv (Cout <= 4, K in {1,3}, stride 1): the final cls_head / offset_head convs
// (av_fd_no_recon.py:56-59,139-142 applied at :85,:155-156 incl. Scale+ReLU of the regression
// head) and nn.Linear(H -> num_classes) of the video classifier (blocks.py:1605).  An MFMA tile
// would be >87% padding at N<=4, so this is a VALU dot product: one wavefront per token, lanes
// strided over the contiguous channel axis, 64-lane shuffle reduction.
//   y[b,t,n] = f( (sum_{k,c} x[b,t+k-P,c]*w[n,c,k] + bias[n]) * mask[b,t] ),  f = relu(scale*.) | id
#include "common.h"

namespace {

constexpr int kMaxN = 4;
constexpr int kPartialBlocks = 256;   // row-strided workgroups of the weight-gradient pass (one partial slab each)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int KS>
__global__ __launch_bounds__(256) void smallconv_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ mask, const float* __restrict__ scale, float* __restrict__ y,
    int B, int T, int Cin, int N) {
  constexpr int P = KS / 2;
  const int lane = threadIdx.x & 63;
  const long bt = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bt >= (long)B * T) return;
  const int t = (int)(bt % T);
  const long b = bt / T;
  float acc[kMaxN] = {0.f, 0.f, 0.f, 0.f};
  for (int c = lane * 4; c < Cin; c += 256) {
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int ti = t + k - P;
      if (ti < 0 || ti >= T) continue;
      const float4 xv = ld4(x + (b * T + ti) * Cin + c);
#pragma unroll
      for (int n = 0; n < kMaxN; ++n) {
        if (n < N) {
          const float* wp = w + ((size_t)n * Cin + c) * KS + k;
          acc[n] += xv.x * wp[0] + xv.y * wp[KS] + xv.z * wp[2 * KS] + xv.w * wp[3 * KS];
        }
      }
    }
  }
  const float m = mask ? mask[bt] : 1.f;
#pragma unroll
  for (int n = 0; n < kMaxN; ++n) {
    if (n < N) {
      float v = wave_sum(acc[n]);
      v = (v + (bias ? bias[n] : 0.f)) * m;
      if (scale) v = fmaxf(v * scale[0], 0.f);
      if (lane == 0) y[bt * N + n] = v;
    }
  }
}

// g[b,t,n] = d(loss)/d(pre-mask conv output); also per-call sums for dbias / dscale (single block)
__global__ __launch_bounds__(1024) void smallconv_dpre_kernel(
    const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ mask,
    const float* __restrict__ scale, float* __restrict__ g, float* __restrict__ dbias, float* __restrict__ dscale,
    long M, int N, int accumulate) {
  __shared__ float red[1024];
  float sb[kMaxN] = {0.f, 0.f, 0.f, 0.f};
  float ss = 0.f;
  const float sc = scale ? scale[0] : 1.f;
  for (long r = threadIdx.x; r < M; r += 1024) {
    const float m = mask ? mask[r] : 1.f;
    for (int n = 0; n < N; ++n) {
      float d = dy[r * N + n];
      if (scale) {
        const float yv = y[r * N + n];
        if (yv > 0.f) { ss += d * (yv / sc); d *= sc; } else d = 0.f;
      }
      d *= m;
      g[r * N + n] = d;
      sb[n] += d;
    }
  }
  for (int n = 0; n <= N; ++n) {
    red[threadIdx.x] = n < N ? sb[n < kMaxN ? n : 0] : ss;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      if (n < N) { if (dbias) dbias[n] = accumulate ? dbias[n] + red[0] : red[0]; }
      else if (dscale) dscale[0] = accumulate ? dscale[0] + red[0] : red[0];
    }
    __syncthreads();
  }
}

template <int KS>
__global__ __launch_bounds__(256) void smallconv_dx_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                           float* __restrict__ dx, int B, int T, int Cin, int N) {
  constexpr int P = KS / 2;
  const int lane = threadIdx.x & 63;
  const long bt = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bt >= (long)B * T) return;
  const int tp = (int)(bt % T);
  const long b = bt / T;
  float gv[KS][kMaxN];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const int t = tp - k + P;
#pragma unroll
    for (int n = 0; n < kMaxN; ++n) gv[k][n] = (t >= 0 && t < T && n < N) ? g[(b * T + t) * N + n] : 0.f;
  }
  for (int c = lane * 4; c < Cin; c += 256) {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
      for (int n = 0; n < kMaxN; ++n) {
        if (n < N) {
          const float* wp = w + ((size_t)n * Cin + c) * KS + k;
          o.x += gv[k][n] * wp[0]; o.y += gv[k][n] * wp[KS]; o.z += gv[k][n] * wp[2 * KS]; o.w += gv[k][n] * wp[3 * KS];
        }
      }
    *reinterpret_cast<float4*>(dx + bt * Cin + c) = o;
  }
}

// partial[blockIdx.x][n][c][k]; one pass over the rows for all N outputs (x is read once per tap, not once per output)
template <int KS>
__global__ __launch_bounds__(256) void smallconv_dw_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                           float* __restrict__ partial, int B, int T, int Cin, int N) {
  constexpr int P = KS / 2;
  __shared__ float red[4][64 * 4 * KS];
  const int cg = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const long M = (long)B * T;
  for (int cb = 0; cb < Cin; cb += 256) {
    const int c = cb + cg * 4;
    float4 a[kMaxN][KS];
#pragma unroll
    for (int n = 0; n < kMaxN; ++n)
#pragma unroll
      for (int k = 0; k < KS; ++k) a[n][k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < Cin) {
      for (long r = (long)blockIdx.x * 4 + rg; r < M; r += (long)gridDim.x * 4) {
        float gg[kMaxN];
#pragma unroll
        for (int n = 0; n < kMaxN; ++n) gg[n] = n < N ? g[r * N + n] : 0.f;
        const int t = (int)(r % T);
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const int ti = t + k - P;
          if (ti >= 0 && ti < T) {
            const float4 xv = ld4(x + (r + (k - P)) * Cin + c);
#pragma unroll
            for (int n = 0; n < kMaxN; ++n) {
              a[n][k].x += gg[n] * xv.x; a[n][k].y += gg[n] * xv.y; a[n][k].z += gg[n] * xv.z; a[n][k].w += gg[n] * xv.w;
            }
          }
        }
      }
    }
#pragma unroll
    for (int n = 0; n < kMaxN; ++n) {
      if (n < N) {
        float* d = &red[rg][cg * 4 * KS];
#pragma unroll
        for (int k = 0; k < KS; ++k) { d[0 * KS + k] = a[n][k].x; d[1 * KS + k] = a[n][k].y; d[2 * KS + k] = a[n][k].z; d[3 * KS + k] = a[n][k].w; }
        __syncthreads();
        for (int e = threadIdx.x; e < 256 * KS; e += 256) {
          const int idx = cb * KS + e;
          if (idx < Cin * KS)
            partial[((size_t)blockIdx.x * N + n) * Cin * KS + idx] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
        }
        __syncthreads();
      }
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int avdf_smallconv_fwd(const float* x, const float* w, const float* bias, const float* mask,
                                  const float* scale, float* y, int B, int T, int Cin, int Cout, int K, void* stream) {
  AVDF_REQUIRE(x && w && y && B > 0 && T > 0 && Cin > 0, "smallconv_fwd: bad args");
  AVDF_REQUIRE(Cout >= 1 && Cout <= kMaxN && (K == 1 || K == 3), "smallconv_fwd: Cout in 1..4, K in {1,3}");
  AVDF_REQUIRE(Cin % 4 == 0 && aligned16(x), "smallconv_fwd: Cin %% 4 / alignment");
  const dim3 grid((unsigned)(((long)B * T + 3) / 4));
  if (K == 3) hipLaunchKernelGGL(smallconv_fwd_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, mask, scale, y, B, T, Cin, Cout);
  else hipLaunchKernelGGL(smallconv_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, mask, scale, y, B, T, Cin, Cout);
  AVDF_CHECK_LAUNCH("smallconv_fwd");
  return AVDF_OK;
}

extern "C" int64_t avdf_smallconv_bwd_ws(int B, int T, int Cin, int Cout, int K) {
  return (int64_t)B * T * Cout + (int64_t)kPartialBlocks * Cout * Cin * K;
}

extern "C" int avdf_smallconv_bwd(const float* dy, const float* x, const float* w, const float* y,
                                  const float* mask, const float* scale, float* dx, float* dw, float* dbias,
                                  float* dscale, float* ws, int B, int T, int Cin, int Cout, int K,
                                  int accumulate, void* stream) {
  AVDF_REQUIRE(dy && x && w && ws && B > 0 && T > 0 && Cin > 0, "smallconv_bwd: bad args");
  AVDF_REQUIRE(Cout >= 1 && Cout <= kMaxN && (K == 1 || K == 3), "smallconv_bwd: Cout in 1..4, K in {1,3}");
  AVDF_REQUIRE(Cin % 4 == 0 && aligned16(x) && (!dx || aligned16(dx)), "smallconv_bwd: Cin %% 4 / alignment");
  AVDF_REQUIRE(!scale || y, "smallconv_bwd: y required when scale is given");
  hipStream_t st = (hipStream_t)stream;
  float* g = ws;
  float* partial = ws + (size_t)B * T * Cout;
  hipLaunchKernelGGL(smallconv_dpre_kernel, dim3(1), dim3(1024), 0, st, dy, y, mask, scale, g, dbias, dscale, (long)B * T, Cout, accumulate);
  AVDF_CHECK_LAUNCH("smallconv_dpre");
  const dim3 grid((unsigned)(((long)B * T + 3) / 4));
  if (dx) {
    if (K == 3) hipLaunchKernelGGL(smallconv_dx_kernel<3>, grid, dim3(256), 0, st, g, w, dx, B, T, Cin, Cout);
    else hipLaunchKernelGGL(smallconv_dx_kernel<1>, grid, dim3(256), 0, st, g, w, dx, B, T, Cin, Cout);
    AVDF_CHECK_LAUNCH("smallconv_dx");
  }
  if (dw) {
    if (K == 3) hipLaunchKernelGGL(smallconv_dw_kernel<3>, dim3(kPartialBlocks), dim3(256), 0, st, g, x, partial, B, T, Cin, Cout);
    else hipLaunchKernelGGL(smallconv_dw_kernel<1>, dim3(kPartialBlocks), dim3(256), 0, st, g, x, partial, B, T, Cin, Cout);
    AVDF_CHECK_LAUNCH("smallconv_dw");
    int rc = avdf_reduce_slabs(partial, dw, kPartialBlocks, (int64_t)Cout * Cin * K, accumulate, st);
    if (rc) return rc;
  }
  return AVDF_OK;
}

// Narrow-output conv (Cout <= 4, K in {1,3}, stride 1): the final cls_head / offset_head convs
// (av_fd_no_recon.py:56-59,139-142 applied at :85,:155-156 incl. Scale+ReLU of the regression
// head) and nn.Linear(H -> num_classes) of the video classifier (blocks.py:1605).  An MFMA tile
// would be >87% padding at N<=4, so this is a VALU dot product: one wavefront per token, lanes
// strided over the contiguous channel axis, 64-lane shuffle reduction.
//   y[b,t,n] = f( (sum_{k,c} x[b,t+k-P,c]*w[n,c,k] + bias[n]) * mask[b,t] ),  f = relu(scale*.) | id
#include "common.h"
#include <stdlib.h>

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

// ---- round 4: a wave owns a RUN of consecutive tokens ------------------------------------------------------------------
// The per-token kernels above re-fetch, for every token, the 3 input rows of its taps and -- 4 bytes at a time, 12 loads
// per output channel -- the weights of the lane's 4 input channels: instruction-bound, 16-18 us for 13-20 MB
// (profiles/r04_g_hbm_kernels.md).  Here the weights of a lane's channels sit in registers for the whole run (three 16-byte
// loads per output channel, once), the rows of a run are loaded once each (TOK + 2 rows for TOK tokens) and all of them
// before the first use.
constexpr int kRunTok = 8;

// w[n][c .. c+3][k], k < KS, as per-tap float4s over the 4 channels (12 consecutive floats of the weight row for KS = 3)
template <int KS>
__device__ __forceinline__ void load_w4(const float* __restrict__ w, int n, int Cin, int c, float4 (&wv)[KS]) {
  const float* wp = w + ((size_t)n * Cin + c) * KS;
  if (KS == 1) { wv[0] = ld4(wp); return; }
  const float4 a = ld4(wp), b = ld4(wp + 4), d = ld4(wp + 8);      // elements j*3 + k
  wv[0] = make_float4(a.x, a.w, b.z, d.y);
  wv[1 % KS] = make_float4(a.y, b.x, b.w, d.z);
  wv[2 % KS] = make_float4(a.z, b.y, d.x, d.w);
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

template <int KS>
__global__ __launch_bounds__(256) void smallconv_fwd_rows_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ mask, const float* __restrict__ scale, float* __restrict__ y,
    int B, int T, int Cin, int N, int runs) {
  constexpr int P = KS / 2, NR = kRunTok + KS - 1;
  const int lane = threadIdx.x & 63;
  const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= (long)B * runs) return;
  const long b = g / runs;
  const int t0 = (int)(g % runs) * kRunTok;
  float acc[kRunTok][kMaxN];
#pragma unroll
  for (int i = 0; i < kRunTok; ++i)
#pragma unroll
    for (int n = 0; n < kMaxN; ++n) acc[i][n] = 0.f;
  for (int c = lane * 4; c < Cin; c += 256) {
    float4 xr[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int t = t0 - P + r;
      xr[r] = (t >= 0 && t < T) ? ld4(x + (b * T + t) * Cin + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int n = 0; n < kMaxN; ++n) {
      if (n >= N) break;
      float4 wv[KS];
      load_w4<KS>(w, n, Cin, c, wv);
#pragma unroll
      for (int i = 0; i < kRunTok; ++i)
#pragma unroll
        for (int k = 0; k < KS; ++k) acc[i][n] += dot4(xr[i + k], wv[k]);
    }
  }
  const float sc = scale ? scale[0] : 0.f;
#pragma unroll
  for (int i = 0; i < kRunTok; ++i) {
    const int t = t0 + i;
    if (t >= T) break;
    const long bt = b * T + t;
    const float m = mask ? mask[bt] : 1.f;
#pragma unroll
    for (int n = 0; n < kMaxN; ++n) {
      if (n >= N) break;
      float v = wave_sum(acc[i][n]);
      v = (v + (bias ? bias[n] : 0.f)) * m;
      if (scale) v = fmaxf(v * sc, 0.f);
      if (lane == 0) y[bt * N + n] = v;
    }
  }
}

template <int KS>
__global__ __launch_bounds__(256) void smallconv_dx_rows_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                                float* __restrict__ dx, int B, int T, int Cin, int N,
                                                                int runs) {
  constexpr int P = KS / 2, NR = kRunTok + KS - 1;
  const int lane = threadIdx.x & 63;
  const long gi = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gi >= (long)B * runs) return;
  const long b = gi / runs;
  const int t0 = (int)(gi % runs) * kRunTok;
  // g rows t0 - P .. t0 + TOK - 1 + P (dx[t] = sum_k sum_n g[t - k + P][n] w[n][c][k])
  float gv[NR][kMaxN];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int t = t0 - P + r;
#pragma unroll
    for (int n = 0; n < kMaxN; ++n) gv[r][n] = (t >= 0 && t < T && n < N) ? g[(b * T + t) * N + n] : 0.f;
  }
  for (int c = lane * 4; c < Cin; c += 256) {
    float4 o[kRunTok];
#pragma unroll
    for (int i = 0; i < kRunTok; ++i) o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int n = 0; n < kMaxN; ++n) {
      if (n >= N) break;
      float4 wv[KS];
      load_w4<KS>(w, n, Cin, c, wv);
#pragma unroll
      for (int i = 0; i < kRunTok; ++i)
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const float gg = gv[i + KS - 1 - k][n];
          o[i].x += gg * wv[k].x; o[i].y += gg * wv[k].y; o[i].z += gg * wv[k].z; o[i].w += gg * wv[k].w;
        }
    }
#pragma unroll
    for (int i = 0; i < kRunTok; ++i)
      if (t0 + i < T) *reinterpret_cast<float4*>(dx + (b * T + t0 + i) * Cin + c) = o[i];
  }
}

// partial[blockIdx.x][n][c][k]: a (workgroup, row group) pair walks ONE contiguous chunk of rows with the three rows of
// the taps in registers (each x row is loaded once instead of once per tap)
template <int KS>
__global__ __launch_bounds__(256) void smallconv_dw_rows_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                float* __restrict__ partial, int B, int T, int Cin, int N) {
  __shared__ float red[4][64 * 4 * KS];
  const int cg = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const long M = (long)B * T;
  const long per = (M + (long)gridDim.x * 4 - 1) / ((long)gridDim.x * 4);
  const long r0 = ((long)blockIdx.x * 4 + rg) * per, r1 = r0 + per < M ? r0 + per : M;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int cb = 0; cb < Cin; cb += 256) {
    const int c = cb + cg * 4;
    float4 a[kMaxN][KS];
#pragma unroll
    for (int n = 0; n < kMaxN; ++n)
#pragma unroll
      for (int k = 0; k < KS; ++k) a[n][k] = z4;
    if (c < Cin && r0 < r1) {
      float4 xm = (KS == 3 && r0 > 0) ? ld4(x + (r0 - 1) * Cin + c) : z4;
      float4 xc = ld4(x + r0 * Cin + c);
      for (long r = r0; r < r1; ++r) {
        const float4 xp = (KS == 3 && r + 1 < M) ? ld4(x + (r + 1) * Cin + c) : z4;
        const int t = (int)(r % T);
        float gg[kMaxN];
#pragma unroll
        for (int n = 0; n < kMaxN; ++n) gg[n] = n < N ? g[r * N + n] : 0.f;
        const float fm = (t > 0) ? 1.f : 0.f, fp = (t < T - 1) ? 1.f : 0.f;      // taps that leave the clip contribute nothing
#pragma unroll
        for (int n = 0; n < kMaxN; ++n) {
          if (KS == 3) {
            const float gm = gg[n] * fm, gp = gg[n] * fp;
            a[n][0].x += gm * xm.x; a[n][0].y += gm * xm.y; a[n][0].z += gm * xm.z; a[n][0].w += gm * xm.w;
            a[n][1].x += gg[n] * xc.x; a[n][1].y += gg[n] * xc.y; a[n][1].z += gg[n] * xc.z; a[n][1].w += gg[n] * xc.w;
            a[n][2 % KS].x += gp * xp.x; a[n][2 % KS].y += gp * xp.y; a[n][2 % KS].z += gp * xp.z; a[n][2 % KS].w += gp * xp.w;
          } else {
            a[n][0].x += gg[n] * xc.x; a[n][0].y += gg[n] * xc.y; a[n][0].z += gg[n] * xc.z; a[n][0].w += gg[n] * xc.w;
          }
        }
        xm = xc; xc = xp;
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
// AVDF_SMALLCONV_ROWS=0: the per-token kernels of rounds 1-3 (A/B)
static const int g_rows = [] { const char* e = getenv("AVDF_SMALLCONV_ROWS"); return e ? atoi(e) : 1; }();

}  // namespace

extern "C" int avdf_smallconv_fwd(const float* x, const float* w, const float* bias, const float* mask,
                                  const float* scale, float* y, int B, int T, int Cin, int Cout, int K, void* stream) {
  AVDF_REQUIRE(x && w && y && B > 0 && T > 0 && Cin > 0, "smallconv_fwd: bad args");
  AVDF_REQUIRE(Cout >= 1 && Cout <= kMaxN && (K == 1 || K == 3), "smallconv_fwd: Cout in 1..4, K in {1,3}");
  AVDF_REQUIRE(Cin % 4 == 0 && aligned16(x), "smallconv_fwd: Cin %% 4 / alignment");
  if (g_rows && aligned16(w)) {
    const int runs = cdiv(T, kRunTok);
    const dim3 grid((unsigned)(((long)B * runs + 3) / 4));
    if (K == 3) hipLaunchKernelGGL(smallconv_fwd_rows_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, mask, scale, y, B, T, Cin, Cout, runs);
    else hipLaunchKernelGGL(smallconv_fwd_rows_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, mask, scale, y, B, T, Cin, Cout, runs);
    AVDF_CHECK_LAUNCH("smallconv_fwd_rows");
    return AVDF_OK;
  }
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
  const bool rows = g_rows && aligned16(w);
  if (dx) {
    if (rows) {
      const int runs = cdiv(T, kRunTok);
      const dim3 rgrid((unsigned)(((long)B * runs + 3) / 4));
      if (K == 3) hipLaunchKernelGGL(smallconv_dx_rows_kernel<3>, rgrid, dim3(256), 0, st, g, w, dx, B, T, Cin, Cout, runs);
      else hipLaunchKernelGGL(smallconv_dx_rows_kernel<1>, rgrid, dim3(256), 0, st, g, w, dx, B, T, Cin, Cout, runs);
    } else if (K == 3) hipLaunchKernelGGL(smallconv_dx_kernel<3>, grid, dim3(256), 0, st, g, w, dx, B, T, Cin, Cout);
    else hipLaunchKernelGGL(smallconv_dx_kernel<1>, grid, dim3(256), 0, st, g, w, dx, B, T, Cin, Cout);
    AVDF_CHECK_LAUNCH("smallconv_dx");
  }
  if (dw) {
    if (g_rows) {
      if (K == 3) hipLaunchKernelGGL(smallconv_dw_rows_kernel<3>, dim3(kPartialBlocks), dim3(256), 0, st, g, x, partial, B, T, Cin, Cout);
      else hipLaunchKernelGGL(smallconv_dw_rows_kernel<1>, dim3(kPartialBlocks), dim3(256), 0, st, g, x, partial, B, T, Cin, Cout);
    } else if (K == 3) hipLaunchKernelGGL(smallconv_dw_kernel<3>, dim3(kPartialBlocks), dim3(256), 0, st, g, x, partial, B, T, Cin, Cout);
    else hipLaunchKernelGGL(smallconv_dw_kernel<1>, dim3(kPartialBlocks), dim3(256), 0, st, g, x, partial, B, T, Cin, Cout);
    AVDF_CHECK_LAUNCH("smallconv_dw");
    int rc = avdf_reduce_slabs(partial, dw, kPartialBlocks, (int64_t)Cout * Cin * K, accumulate, st);
    if (rc) return rc;
  }
  return AVDF_OK;
}

// Attention cores.
//  * local (banded) attention -- LocalMaskedMHCA / LocalMaskedMMHCA steps 3-4
//    (libs/modeling/blocks.py:739-778, 1181-1221).  With window 7 there are only 7 dot products
//    of length hs per (query, head): this is an HBM/L2-bound gather, not an MFMA shape, so it is
//    done with one wavefront per (b, t, head) and 64-lane shuffle reductions.
//  * global attention of the last pyramid level -- MaskedMHCA core (blocks.py:299-309).
// Layout: q,k,v,out [B,T,H*hs] time-major, head h = channels [h*hs, (h+1)*hs).
#include "common.h"

namespace {

template <int W>
__global__ __launch_bounds__(256) void local_attn_fwd_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ kvmask, float* __restrict__ out, float* __restrict__ probs,
    int B, int T, int H, int hs, float scale, const float* __restrict__ rel, const float* __restrict__ tw) {
  constexpr int w = W / 2;
  const int lane = threadIdx.x & 63;
  const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= (long)B * T * H) return;
  const int h = (int)(g % H);
  const long bt = g / H;
  const int t = (int)(bt % T);
  const long b = bt / T;
  const int C = H * hs;
  const bool on = lane < hs;
  const float qd = on ? q[bt * C + h * hs + lane] * scale : 0.f;
  float s[W];
#pragma unroll
  for (int j = 0; j < W; ++j) {
    const int tk = t + j - w;
    if (tk >= 0 && tk < T) {
      const float kd = on ? k[(b * T + tk) * C + h * hs + lane] : 0.f;
      float d = wave_sum(qd * kd);
      if (rel) d += rel[h * W + j];                      // relative position bias (blocks.py:1188-1189)
      d += (kvmask[b * T + tk] == 0.f) ? -1e4f : 0.f;
      s[j] = d;
    } else {
      s[j] = -INFINITY;
    }
  }
  float m = s[0];
#pragma unroll
  for (int j = 1; j < W; ++j) m = fmaxf(m, s[j]);
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < W; ++j) { s[j] = __expf(s[j] - m); den += s[j]; }
  const float rowm = kvmask[b * T + t];       // masked query rows are zeroed after the softmax
  const float inv = rowm / den;
  float o = 0.f;
#pragma unroll
  for (int j = 0; j < W; ++j) {
    const int tk = t + j - w;
    const float p = s[j] * inv;
    if (probs && lane == 0) probs[(bt * H + h) * W + j] = p;
    const float pe = tw ? p * tw[h * W + j] : p;          // time weighting after the softmax (blocks.py:1211-1212)
    if (tk >= 0 && tk < T && on) o += pe * v[(b * T + tk) * C + h * hs + lane];
  }
  if (on) out[bt * C + h * hs + lane] = o;
}

// pass A: dS (to ws) and dq, one wave per (b,t,h)
template <int W>
__global__ __launch_bounds__(256) void local_attn_bwd_q_kernel(
    const float* __restrict__ dout, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ probs, float* __restrict__ dq, float* __restrict__ dS,
    int B, int T, int H, int hs, float scale, const float* __restrict__ tw, float* __restrict__ dT) {
  constexpr int w = W / 2;
  const int lane = threadIdx.x & 63;
  const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= (long)B * T * H) return;
  const int h = (int)(g % H);
  const long bt = g / H;
  const int t = (int)(bt % T);
  const long b = bt / T;
  const int C = H * hs;
  const bool on = lane < hs;
  const float go = on ? dout[bt * C + h * hs + lane] : 0.f;
  float dp[W], p[W];
  float dot = 0.f;
#pragma unroll
  for (int j = 0; j < W; ++j) {
    const int tk = t + j - w;
    p[j] = probs[(bt * H + h) * W + j];
    if (tk >= 0 && tk < T) {
      const float vd = on ? v[(b * T + tk) * C + h * hs + lane] : 0.f;
      dp[j] = wave_sum(go * vd);
    } else {
      dp[j] = 0.f;
    }
    if (dT && lane == 0) dT[(bt * H + h) * W + j] = p[j] * dp[j];     // d time_weighting[h, j] summand
    if (tw) dp[j] *= tw[h * W + j];
    dot += p[j] * dp[j];
  }
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < W; ++j) {
    const int tk = t + j - w;
    const float ds = p[j] * (dp[j] - dot);
    if (lane == 0) dS[(bt * H + h) * W + j] = ds;
    if (tk >= 0 && tk < T && on) acc += ds * k[(b * T + tk) * C + h * hs + lane];
  }
  if (on) dq[bt * C + h * hs + lane] = acc * scale;
}

// pass B: dk, dv gathered per key position, one wave per (b,s,h)
template <int W>
__global__ __launch_bounds__(256) void local_attn_bwd_kv_kernel(
    const float* __restrict__ dout, const float* __restrict__ q, const float* __restrict__ probs,
    const float* __restrict__ dS, float* __restrict__ dk, float* __restrict__ dv,
    int B, int T, int H, int hs, float scale, const float* __restrict__ tw) {
  constexpr int w = W / 2;
  const int lane = threadIdx.x & 63;
  const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= (long)B * T * H) return;
  const int h = (int)(g % H);
  const long bs = g / H;
  const int s = (int)(bs % T);
  const long b = bs / T;
  const int C = H * hs;
  if (lane >= hs) return;
  float ak = 0.f, av = 0.f;
#pragma unroll
  for (int j = 0; j < W; ++j) {
    const int t = s - (j - w);            // query t sees key s in its band slot j
    if (t >= 0 && t < T) {
      const long bt = b * T + t;
      const float ds = dS[(bt * H + h) * W + j];
      const float p = probs[(bt * H + h) * W + j] * (tw ? tw[h * W + j] : 1.f);
      ak += ds * q[bt * C + h * hs + lane];
      av += p * dout[bt * C + h * hs + lane];
    }
  }
  dk[bs * C + h * hs + lane] = ak * scale;
  dv[bs * C + h * hs + lane] = av;
}

// ---- row-vectorised variants: one wave per NTOK consecutive tokens, ALL heads at once --------------------------
// lane = one float4 of the C = H*hs channels of a token row (C <= 256), a head = hs/4 adjacent lanes (power of two):
// every q/k/v/out access is a coalesced 16-byte row load, per-head dot products reduce inside the lane group, and
// the 2w+NTOK key/value rows a run of tokens needs are loaded once (the one-wave-per-(token,head) kernels above
// re-read every row 7x through 4-byte loads: 3x HBM-side over-fetch in the PMC pass).
constexpr int kAttnTok = 2;

__device__ __forceinline__ float4 ld4a(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
template <int L>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = L >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);     // constant offsets -> DPP / swizzle
  return v;
}

template <int W, int L>
__global__ __launch_bounds__(256) void local_attn_fwd_rows_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ kvmask, float* __restrict__ out, float* __restrict__ probs,
    int B, int T, int H, int hs, float scale, int runs, const float* __restrict__ rel, const float* __restrict__ tw) {
  constexpr int w = W / 2, NR = kAttnTok + 2 * w;
  const int lane = threadIdx.x & 63;
  const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= (long)B * runs) return;
  const long b = g / runs;
  const int t0 = (int)(g % runs) * kAttnTok;
  const int C = H * hs, c = lane * 4;
  const bool on = c < C;
  const int h = on ? c / hs : 0;
  float relb[W], twb[W];                      // per-head bias before / weight after the softmax (blocks.py:1188-1212)
#pragma unroll
  for (int j = 0; j < W; ++j) { relb[j] = rel ? rel[h * W + j] : 0.f; twb[j] = tw ? tw[h * W + j] : 1.f; }
  float4 kr[NR], vr[NR];
  float pen[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int tk = t0 - w + r;
    kr[r] = make_float4(0.f, 0.f, 0.f, 0.f); vr[r] = kr[r]; pen[r] = -INFINITY;
    if (tk >= 0 && tk < T) {
      if (on) { kr[r] = ld4a(k + (b * T + tk) * C + c); vr[r] = ld4a(v + (b * T + tk) * C + c); }
      pen[r] = (kvmask[b * T + tk] == 0.f) ? -1e4f : 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < kAttnTok; ++i) {
    const int t = t0 + i;
    if (t >= T) break;
    const long bt = b * T + t;
    float4 q4 = on ? ld4a(q + bt * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    q4.x *= scale; q4.y *= scale; q4.z *= scale; q4.w *= scale;
    float s[W];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      s[j] = group_sum<L>(dot4(q4, kr[i + j])) + relb[j] + pen[i + j];        // -inf outside [0,T)
      m = fmaxf(m, s[j]);
    }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < W; ++j) { s[j] = __expf(s[j] - m); den += s[j]; }
    const float inv = kvmask[bt] / den;          // masked query rows are zeroed after the softmax
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < W; ++j) {
      float p = s[j] * inv;
      if (probs && on && (lane & (L - 1)) == 0) probs[(bt * H + h) * W + j] = p;
      p *= twb[j];
      o.x += p * vr[i + j].x; o.y += p * vr[i + j].y; o.z += p * vr[i + j].z; o.w += p * vr[i + j].w;
    }
    if (on) *reinterpret_cast<float4*>(out + bt * C + c) = o;
  }
}

// dS (to ws) and dq
template <int W, int L>
__global__ __launch_bounds__(256) void local_attn_bwd_q_rows_kernel(
    const float* __restrict__ dout, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ probs, float* __restrict__ dq, float* __restrict__ dS,
    int B, int T, int H, int hs, float scale, int runs, const float* __restrict__ tw, float* __restrict__ dT) {
  constexpr int w = W / 2, NR = kAttnTok + 2 * w;
  const int lane = threadIdx.x & 63;
  const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= (long)B * runs) return;
  const long b = g / runs;
  const int t0 = (int)(g % runs) * kAttnTok;
  const int C = H * hs, c = lane * 4;
  const bool on = c < C;
  const int h = on ? c / hs : 0;
  float twb[W];
#pragma unroll
  for (int j = 0; j < W; ++j) twb[j] = tw ? tw[h * W + j] : 1.f;
  float4 kr[NR], vr[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int tk = t0 - w + r;
    kr[r] = make_float4(0.f, 0.f, 0.f, 0.f); vr[r] = kr[r];
    if (tk >= 0 && tk < T && on) { kr[r] = ld4a(k + (b * T + tk) * C + c); vr[r] = ld4a(v + (b * T + tk) * C + c); }
  }
#pragma unroll
  for (int i = 0; i < kAttnTok; ++i) {
    const int t = t0 + i;
    if (t >= T) break;
    const long bt = b * T + t;
    const float4 go = on ? ld4a(dout + bt * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    float dp[W], p[W];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      p[j] = on ? probs[(bt * H + h) * W + j] : 0.f;
      dp[j] = group_sum<L>(dot4(go, vr[i + j]));      // rows outside [0,T) are zero -> dp = 0, and p = 0 there
      if (dT && on && (lane & (L - 1)) == 0) dT[(bt * H + h) * W + j] = p[j] * dp[j];
      dp[j] *= twb[j];
      dot += p[j] * dp[j];
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const float ds = p[j] * (dp[j] - dot);
      if (on && (lane & (L - 1)) == 0) dS[(bt * H + h) * W + j] = ds;
      acc.x += ds * kr[i + j].x; acc.y += ds * kr[i + j].y; acc.z += ds * kr[i + j].z; acc.w += ds * kr[i + j].w;
    }
    if (on) *reinterpret_cast<float4*>(dq + bt * C + c) = make_float4(acc.x * scale, acc.y * scale, acc.z * scale, acc.w * scale);
  }
}

// dk, dv gathered per key position
template <int W>
__global__ __launch_bounds__(256) void local_attn_bwd_kv_rows_kernel(
    const float* __restrict__ dout, const float* __restrict__ q, const float* __restrict__ probs,
    const float* __restrict__ dS, float* __restrict__ dk, float* __restrict__ dv,
    int B, int T, int H, int hs, float scale, int runs, const float* __restrict__ tw) {
  constexpr int w = W / 2, NR = kAttnTok + 2 * w;
  const int lane = threadIdx.x & 63;
  const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= (long)B * runs) return;
  const long b = g / runs;
  const int s0 = (int)(g % runs) * kAttnTok;
  const int C = H * hs, c = lane * 4;
  if (c >= C) return;
  const int h = c / hs;
  float twb[W];
#pragma unroll
  for (int j = 0; j < W; ++j) twb[j] = tw ? tw[h * W + j] : 1.f;
  float4 qr[NR], gr[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int t = s0 - w + r;
    qr[r] = make_float4(0.f, 0.f, 0.f, 0.f); gr[r] = qr[r];
    if (t >= 0 && t < T) { qr[r] = ld4a(q + (b * T + t) * C + c); gr[r] = ld4a(dout + (b * T + t) * C + c); }
  }
#pragma unroll
  for (int i = 0; i < kAttnTok; ++i) {
    const int s = s0 + i;
    if (s >= T) break;
    float4 ak = make_float4(0.f, 0.f, 0.f, 0.f), av = ak;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const int t = s - (j - w);            // query t sees key s in its band slot j; its row sits at r = t - (s0 - w)
      if (t >= 0 && t < T) {
        const int r = i + 2 * w - j;
        const long bt = b * T + t;
        const float ds = dS[(bt * H + h) * W + j];
        const float p = probs[(bt * H + h) * W + j] * twb[j];
        ak.x += ds * qr[r].x; ak.y += ds * qr[r].y; ak.z += ds * qr[r].z; ak.w += ds * qr[r].w;
        av.x += p * gr[r].x; av.y += p * gr[r].y; av.z += p * gr[r].z; av.w += p * gr[r].w;
      }
    }
    const long bs = b * T + s;
    *reinterpret_cast<float4*>(dk + bs * C + c) = make_float4(ak.x * scale, ak.y * scale, ak.z * scale, ak.w * scale);
    *reinterpret_cast<float4*>(dv + bs * C + c) = av;
  }
}

#define DISPATCH_L(L_, WW, CALL_L)                    \
  switch (L_) {                                       \
    case 1: CALL_L(WW, 1); break;                     \
    case 2: CALL_L(WW, 2); break;                     \
    case 4: CALL_L(WW, 4); break;                     \
    case 8: CALL_L(WW, 8); break;                     \
    case 16: CALL_L(WW, 16); break;                   \
    case 32: CALL_L(WW, 32); break;                   \
    default: CALL_L(WW, 64); break;                   \
  }

inline bool rows_path_ok(int H, int hs) {
  const int L = hs / 4;
  return hs % 4 == 0 && L >= 1 && (L & (L - 1)) == 0 && H * hs <= 256;
}

constexpr int kGMaxSlots = 4;   // Tk <= 256

// one wave per (b, h, tq); probs [B,H,Tq,Tk]
__global__ __launch_bounds__(256) void global_attn_fwd_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ kvmask, float* __restrict__ out, float* __restrict__ probs,
    int B, int Tq, int Tk, int H, int hs, float scale) {
  const int lane = threadIdx.x & 63;
  const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= (long)B * H * Tq) return;
  const int tq = (int)(g % Tq);
  const int h = (int)((g / Tq) % H);
  const long b = g / ((long)Tq * H);
  const int C = H * hs;
  const float qd = lane < hs ? q[(b * Tq + tq) * C + h * hs + lane] * scale : 0.f;
  float s[kGMaxSlots];
#pragma unroll
  for (int u = 0; u < kGMaxSlots; ++u) s[u] = 0.f;
  for (int d = 0; d < hs; ++d) {
    const float qv = __shfl(qd, d, 64);
#pragma unroll
    for (int u = 0; u < kGMaxSlots; ++u) {
      const int kk = u * 64 + lane;
      if (kk < Tk) s[u] += qv * k[(b * Tk + kk) * C + h * hs + d];
    }
  }
  float m = -INFINITY;
#pragma unroll
  for (int u = 0; u < kGMaxSlots; ++u) {
    const int kk = u * 64 + lane;
    if (kk >= Tk || kvmask[b * Tk + kk] == 0.f) s[u] = -INFINITY;
    m = fmaxf(m, s[u]);
  }
  m = wave_max(m);
  float den = 0.f;
#pragma unroll
  for (int u = 0; u < kGMaxSlots; ++u) { s[u] = __expf(s[u] - m); den += s[u]; }   // exp(-inf)=0
  den = wave_sum(den);
  const float inv = 1.0f / den;
#pragma unroll
  for (int u = 0; u < kGMaxSlots; ++u) {
    const int kk = u * 64 + lane;
    s[u] *= inv;
    if (probs && kk < Tk) probs[((b * H + h) * Tq + tq) * Tk + kk] = s[u];
  }
  float o = 0.f;
  for (int kk = 0; kk < Tk; ++kk) {
    const int u = kk >> 6;
    const float src = u == 0 ? s[0] : (u == 1 ? s[1] : (u == 2 ? s[2] : s[3]));
    const float p = __shfl(src, kk & 63, 64);
    if (lane < hs) o += p * v[(b * Tk + kk) * C + h * hs + lane];    // p == 0 on masked keys (v*mask)
  }
  if (lane < hs) out[(b * Tq + tq) * C + h * hs + lane] = o;
}

// pass A: dS -> ws [B,H,Tq,Tk], dq
__global__ __launch_bounds__(256) void global_attn_bwd_q_kernel(
    const float* __restrict__ dout, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ probs, float* __restrict__ dq, float* __restrict__ dS,
    int B, int Tq, int Tk, int H, int hs, float scale) {
  const int lane = threadIdx.x & 63;
  const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= (long)B * H * Tq) return;
  const int tq = (int)(g % Tq);
  const int h = (int)((g / Tq) % H);
  const long b = g / ((long)Tq * H);
  const int C = H * hs;
  const float go = lane < hs ? dout[(b * Tq + tq) * C + h * hs + lane] : 0.f;
  float dp[kGMaxSlots], p[kGMaxSlots];
#pragma unroll
  for (int u = 0; u < kGMaxSlots; ++u) dp[u] = 0.f;
  for (int d = 0; d < hs; ++d) {
    const float gv = __shfl(go, d, 64);
#pragma unroll
    for (int u = 0; u < kGMaxSlots; ++u) {
      const int kk = u * 64 + lane;
      if (kk < Tk) dp[u] += gv * v[(b * Tk + kk) * C + h * hs + d];
    }
  }
  float dot = 0.f;
#pragma unroll
  for (int u = 0; u < kGMaxSlots; ++u) {
    const int kk = u * 64 + lane;
    p[u] = kk < Tk ? probs[((b * H + h) * Tq + tq) * Tk + kk] : 0.f;   // 0 on masked keys
    dot += p[u] * dp[u];
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int u = 0; u < kGMaxSlots; ++u) {
    const int kk = u * 64 + lane;
    dp[u] = p[u] * (dp[u] - dot);                                     // dS
    if (kk < Tk) dS[((b * H + h) * Tq + tq) * Tk + kk] = dp[u];
  }
  float acc = 0.f;
  for (int kk = 0; kk < Tk; ++kk) {
    const int u = kk >> 6;
    const float src = u == 0 ? dp[0] : (u == 1 ? dp[1] : (u == 2 ? dp[2] : dp[3]));
    const float ds = __shfl(src, kk & 63, 64);
    if (lane < hs) acc += ds * k[(b * Tk + kk) * C + h * hs + lane];
  }
  if (lane < hs) dq[(b * Tq + tq) * C + h * hs + lane] = acc * scale;
}

// pass B: one wave per (b,h,tk)
__global__ __launch_bounds__(256) void global_attn_bwd_kv_kernel(
    const float* __restrict__ dout, const float* __restrict__ q, const float* __restrict__ probs,
    const float* __restrict__ dS, float* __restrict__ dk, float* __restrict__ dv,
    int B, int Tq, int Tk, int H, int hs, float scale) {
  const int lane = threadIdx.x & 63;
  const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= (long)B * H * Tk) return;
  const int tk = (int)(g % Tk);
  const int h = (int)((g / Tk) % H);
  const long b = g / ((long)Tk * H);
  const int C = H * hs;
  if (lane >= hs) return;
  float ak = 0.f, av = 0.f;
  for (int tq = 0; tq < Tq; ++tq) {
    const long idx = ((b * H + h) * Tq + tq) * Tk + tk;
    ak += dS[idx] * q[(b * Tq + tq) * C + h * hs + lane];
    av += probs[idx] * dout[(b * Tq + tq) * C + h * hs + lane];
  }
  dk[(b * Tk + tk) * C + h * hs + lane] = ak * scale;
  dv[(b * Tk + tk) * C + h * hs + lane] = av;     // masked keys have probs == 0 -> dv == 0
}


// ---- Tk <= 64 (the top pyramid level of the shipped configs: 24 keys per clip): key / value rows in REGISTERS ------------------
// The general kernels above walk the head dimension with one strided 4-byte gather per step (lane = key, stride = a
// token row): 64 dependent L2 round trips, 20-23 us per launch for 3 MB (profiles/r04_g_hbm_kernels.md).  Here a lane
// loads its key's (value's) whole head row with 16-byte loads that are all in flight at once, the dot products run on
// registers against lane-broadcast query elements, and the row-parallel passes (out = P V, dq = dS K) fetch eight
// coalesced rows at a time.
__device__ __forceinline__ float f4c(const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }

__global__ __launch_bounds__(256) void global_attn_fwd_small_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ kvmask, float* __restrict__ out, float* __restrict__ probs,
    int B, int Tq, int Tk, int H, int hs, float scale) {
  const int lane = threadIdx.x & 63;
  const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= (long)B * H * Tq) return;
  const int tq = (int)(g % Tq);
  const int h = (int)((g / Tq) % H);
  const long b = g / ((long)Tq * H);
  const int C = H * hs;
  const float* krow = k + (b * Tk + (lane < Tk ? lane : 0)) * C + h * hs;
  float4 kr[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) kr[i] = (lane < Tk && 4 * i < hs) ? ld4a(krow + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float qd = lane < hs ? q[(b * Tq + tq) * C + h * hs + lane] * scale : 0.f;
  const float km = lane < Tk ? kvmask[b * Tk + lane] : 0.f;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += __shfl(qd, 4 * i + j, 64) * f4c(kr[i], j);      // lanes >= hs hold qd = 0
  if (lane >= Tk || km == 0.f) s = -INFINITY;
  const float m = wave_max(s);
  s = __expf(s - m);
  const float p = s / wave_sum(s);
  if (probs && lane < Tk) probs[((b * H + h) * Tq + tq) * Tk + lane] = p;
  float o = 0.f;
  const float* vcol = v + b * Tk * C + h * hs + (lane < hs ? lane : 0);
  for (int k0 = 0; k0 < Tk; k0 += 8) {
    float vv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) vv[j] = (k0 + j < Tk) ? vcol[(long)(k0 + j) * C] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) o += __shfl(p, (k0 + j) & 63, 64) * vv[j];             // p == 0 beyond Tk and on masked keys
  }
  if (lane < hs) out[(b * Tq + tq) * C + h * hs + lane] = o;
}

__global__ __launch_bounds__(256) void global_attn_bwd_q_small_kernel(
    const float* __restrict__ dout, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ probs, float* __restrict__ dq, float* __restrict__ dS,
    int B, int Tq, int Tk, int H, int hs, float scale) {
  const int lane = threadIdx.x & 63;
  const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= (long)B * H * Tq) return;
  const int tq = (int)(g % Tq);
  const int h = (int)((g / Tq) % H);
  const long b = g / ((long)Tq * H);
  const int C = H * hs;
  const float* vrow = v + (b * Tk + (lane < Tk ? lane : 0)) * C + h * hs;
  float4 vr[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) vr[i] = (lane < Tk && 4 * i < hs) ? ld4a(vrow + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float go = lane < hs ? dout[(b * Tq + tq) * C + h * hs + lane] : 0.f;
  const float p = lane < Tk ? probs[((b * H + h) * Tq + tq) * Tk + lane] : 0.f;           // 0 on masked keys
  float dp = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) dp += __shfl(go, 4 * i + j, 64) * f4c(vr[i], j);
  const float dot = wave_sum(p * dp);
  const float ds = p * (dp - dot);
  if (lane < Tk) dS[((b * H + h) * Tq + tq) * Tk + lane] = ds;
  float acc = 0.f;
  const float* kcol = k + b * Tk * C + h * hs + (lane < hs ? lane : 0);
  for (int k0 = 0; k0 < Tk; k0 += 8) {
    float kv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) kv[j] = (k0 + j < Tk) ? kcol[(long)(k0 + j) * C] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += __shfl(ds, (k0 + j) & 63, 64) * kv[j];
  }
  if (lane < hs) dq[(b * Tq + tq) * C + h * hs + lane] = acc * scale;
}

// one wave per (b, h, tk), eight query rows in flight
__global__ __launch_bounds__(256) void global_attn_bwd_kv_small_kernel(
    const float* __restrict__ dout, const float* __restrict__ q, const float* __restrict__ probs,
    const float* __restrict__ dS, float* __restrict__ dk, float* __restrict__ dv,
    int B, int Tq, int Tk, int H, int hs, float scale) {
  const int lane = threadIdx.x & 63;
  const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= (long)B * H * Tk) return;
  const int tk = (int)(g % Tk);
  const int h = (int)((g / Tk) % H);
  const long b = g / ((long)Tk * H);
  const int C = H * hs;
  if (lane >= hs) return;
  float ak = 0.f, av = 0.f;
  const float* qcol = q + b * Tq * C + h * hs + lane;
  const float* gcol = dout + b * Tq * C + h * hs + lane;
  const long pbase = ((b * H + h) * Tq) * (long)Tk + tk;
  for (int t0 = 0; t0 < Tq; t0 += 8) {
    float qv[8], gv[8], sv[8], pv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool on = t0 + j < Tq;
      qv[j] = on ? qcol[(long)(t0 + j) * C] : 0.f;
      gv[j] = on ? gcol[(long)(t0 + j) * C] : 0.f;
      sv[j] = on ? dS[pbase + (long)(t0 + j) * Tk] : 0.f;
      pv[j] = on ? probs[pbase + (long)(t0 + j) * Tk] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { ak += sv[j] * qv[j]; av += pv[j] * gv[j]; }
  }
  dk[(b * Tk + tk) * C + h * hs + lane] = ak * scale;
  dv[(b * Tk + tk) * C + h * hs + lane] = av;     // masked keys have probs == 0 -> dv == 0
}

}  // namespace

#define DISPATCH_W(W_, CALL)             \
  switch (W_) {                          \
    case 3: { CALL(3); break; }          \
    case 5: { CALL(5); break; }          \
    case 7: { CALL(7); break; }          \
    case 9: { CALL(9); break; }          \
    case 11: { CALL(11); break; }        \
    case 13: { CALL(13); break; }        \
    case 15: { CALL(15); break; }        \
    case 17: { CALL(17); break; }        \
    case 19: { CALL(19); break; }        \
    default: avdf_set_error("local_attn: window %d unsupported (odd 3..19)", W_); return AVDF_EINVAL; \
  }

extern "C" int avdf_local_attn_fwd_ex(const float* q, const float* k, const float* v, const float* kvmask, const float* rel,
                                      const float* tw, float* out, float* probs, int B, int T, int H, int hs, int window,
                                      float scale, void* stream);
extern "C" int avdf_local_attn_fwd(const float* q, const float* k, const float* v, const float* kvmask,
                                   float* out, float* probs, int B, int T, int H, int hs, int window,
                                   float scale, void* stream) {
  return avdf_local_attn_fwd_ex(q, k, v, kvmask, nullptr, nullptr, out, probs, B, T, H, hs, window, scale, stream);
}

// rel / tw: optional [H, window] tables -- relative position bias added to the scores before the softmax and time
// weighting applied to the probabilities after it (LocalMaskedMHCA use_rel_pe / use_time_weight, blocks.py:1188-1212)
extern "C" int avdf_local_attn_fwd_ex(const float* q, const float* k, const float* v, const float* kvmask, const float* rel,
                                      const float* tw, float* out, float* probs, int B, int T, int H, int hs, int window,
                                      float scale, void* stream) {
  AVDF_REQUIRE(q && k && v && kvmask && out, "local_attn_fwd: null pointer");
  AVDF_REQUIRE(B > 0 && T > 0 && H > 0 && hs > 0 && hs <= 64, "local_attn_fwd: bad dims (hs=%d must be <= 64)", hs);
  hipStream_t st = (hipStream_t)stream;
  if (rows_path_ok(H, hs) && ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0)) {
    const int runs = cdiv(T, kAttnTok);
    const dim3 grid((unsigned)(((long)B * runs + 3) / 4));
#define CALL_L(WW, LL) hipLaunchKernelGGL((local_attn_fwd_rows_kernel<WW, LL>), grid, dim3(256), 0, st, q, k, v, kvmask, out, probs, B, T, H, hs, scale, runs, rel, tw)
#define CALL(WW) DISPATCH_L(hs / 4, WW, CALL_L)
    DISPATCH_W(window, CALL)
#undef CALL
#undef CALL_L
    AVDF_CHECK_LAUNCH("local_attn_fwd");
    return AVDF_OK;
  }
  const long waves = (long)B * T * H;
  const dim3 grid((unsigned)((waves + 3) / 4));
#define CALL(WW) hipLaunchKernelGGL(local_attn_fwd_kernel<WW>, grid, dim3(256), 0, st, q, k, v, kvmask, out, probs, B, T, H, hs, scale, rel, tw)
  DISPATCH_W(window, CALL)
#undef CALL
  AVDF_CHECK_LAUNCH("local_attn_fwd");
  return AVDF_OK;
}

extern "C" int64_t avdf_local_attn_bwd_ws(int B, int T, int H, int window) { return (int64_t)B * T * H * window; }

extern "C" int avdf_local_attn_bwd_ex(const float* dout, const float* q, const float* k, const float* v, const float* probs,
                                      const float* tw, float* dq, float* dk, float* dv, float* ws, float* dT, int B, int T,
                                      int H, int hs, int window, float scale, void* stream);
extern "C" int avdf_local_attn_bwd(const float* dout, const float* q, const float* k, const float* v,
                                   const float* probs, float* dq, float* dk, float* dv, float* ws,
                                   int B, int T, int H, int hs, int window, float scale, void* stream) {
  return avdf_local_attn_bwd_ex(dout, q, k, v, probs, nullptr, dq, dk, dv, ws, nullptr, B, T, H, hs, window, scale, stream);
}

// ws receives dS [B, T, H, window] (its column sums over (b, t) are the gradient of `rel`); dT (optional, same shape)
// receives p * (dout . v): its column sums are the gradient of `tw`
extern "C" int avdf_local_attn_bwd_ex(const float* dout, const float* q, const float* k, const float* v, const float* probs,
                                      const float* tw, float* dq, float* dk, float* dv, float* ws, float* dT, int B, int T,
                                      int H, int hs, int window, float scale, void* stream) {
  AVDF_REQUIRE(dout && q && k && v && probs && dq && dk && dv && ws, "local_attn_bwd: null pointer");
  AVDF_REQUIRE(B > 0 && T > 0 && H > 0 && hs > 0 && hs <= 64, "local_attn_bwd: bad dims");
  hipStream_t st = (hipStream_t)stream;
  if (rows_path_ok(H, hs) &&
      ((((uintptr_t)dout | (uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0)) {
    const int runs = cdiv(T, kAttnTok);
    const dim3 grid((unsigned)(((long)B * runs + 3) / 4));
#define CALL_L(WW, LL) hipLaunchKernelGGL((local_attn_bwd_q_rows_kernel<WW, LL>), grid, dim3(256), 0, st, dout, k, v, probs, dq, ws, B, T, H, hs, scale, runs, tw, dT)
#define CALL(WW)                       \
  DISPATCH_L(hs / 4, WW, CALL_L);      \
  hipLaunchKernelGGL(local_attn_bwd_kv_rows_kernel<WW>, grid, dim3(256), 0, st, dout, q, probs, ws, dk, dv, B, T, H, hs, scale, runs, tw)
    DISPATCH_W(window, CALL)
#undef CALL
#undef CALL_L
    AVDF_CHECK_LAUNCH("local_attn_bwd");
    return AVDF_OK;
  }
  const long waves = (long)B * T * H;
  const dim3 grid((unsigned)((waves + 3) / 4));
#define CALL(WW)                                                                                             \
  hipLaunchKernelGGL(local_attn_bwd_q_kernel<WW>, grid, dim3(256), 0, st, dout, k, v, probs, dq, ws, B, T, H, hs, scale, tw, dT); \
  hipLaunchKernelGGL(local_attn_bwd_kv_kernel<WW>, grid, dim3(256), 0, st, dout, q, probs, ws, dk, dv, B, T, H, hs, scale, tw)
  DISPATCH_W(window, CALL)
#undef CALL
  AVDF_CHECK_LAUNCH("local_attn_bwd");
  return AVDF_OK;
}

extern "C" int avdf_global_attn_fwd(const float* q, const float* k, const float* v, const float* kvmask,
                                    float* out, float* probs, int B, int Tq, int Tk, int H, int hs,
                                    float scale, void* stream) {
  AVDF_REQUIRE(q && k && v && kvmask && out, "global_attn_fwd: null pointer");
  AVDF_REQUIRE(B > 0 && Tq > 0 && Tk > 0 && Tk <= 64 * kGMaxSlots && H > 0 && hs > 0 && hs <= 64,
               "global_attn_fwd: bad dims (Tk=%d <= %d, hs=%d <= 64)", Tk, 64 * kGMaxSlots, hs);
  const long waves = (long)B * H * Tq;
  const bool small = Tk <= 64 && hs % 4 == 0 && (H * hs) % 4 == 0 && (((uintptr_t)k | (uintptr_t)v) & 15) == 0;
  if (small)
    hipLaunchKernelGGL(global_attn_fwd_small_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       q, k, v, kvmask, out, probs, B, Tq, Tk, H, hs, scale);
  else
    hipLaunchKernelGGL(global_attn_fwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       q, k, v, kvmask, out, probs, B, Tq, Tk, H, hs, scale);
  AVDF_CHECK_LAUNCH("global_attn_fwd");
  return AVDF_OK;
}

extern "C" int64_t avdf_global_attn_bwd_ws(int B, int Tq, int Tk, int H) { return (int64_t)B * H * Tq * Tk; }

extern "C" int avdf_global_attn_bwd(const float* dout, const float* q, const float* k, const float* v,
                                    const float* kvmask, const float* probs, float* dq, float* dk, float* dv,
                                    float* ws, int B, int Tq, int Tk, int H, int hs, float scale, void* stream) {
  (void)kvmask;
  AVDF_REQUIRE(dout && q && k && v && probs && dq && dk && dv && ws, "global_attn_bwd: null pointer");
  AVDF_REQUIRE(B > 0 && Tq > 0 && Tk > 0 && Tk <= 64 * kGMaxSlots && H > 0 && hs > 0 && hs <= 64, "global_attn_bwd: bad dims");
  hipStream_t st = (hipStream_t)stream;
  const long wq = (long)B * H * Tq, wk = (long)B * H * Tk;
  const bool small = Tk <= 64 && hs % 4 == 0 && (H * hs) % 4 == 0 && (((uintptr_t)k | (uintptr_t)v) & 15) == 0;
  if (small) {
    hipLaunchKernelGGL(global_attn_bwd_q_small_kernel, dim3((unsigned)((wq + 3) / 4)), dim3(256), 0, st,
                       dout, k, v, probs, dq, ws, B, Tq, Tk, H, hs, scale);
    hipLaunchKernelGGL(global_attn_bwd_kv_small_kernel, dim3((unsigned)((wk + 3) / 4)), dim3(256), 0, st,
                       dout, q, probs, ws, dk, dv, B, Tq, Tk, H, hs, scale);
  } else {
    hipLaunchKernelGGL(global_attn_bwd_q_kernel, dim3((unsigned)((wq + 3) / 4)), dim3(256), 0, st,
                       dout, k, v, probs, dq, ws, B, Tq, Tk, H, hs, scale);
    hipLaunchKernelGGL(global_attn_bwd_kv_kernel, dim3((unsigned)((wk + 3) / 4)), dim3(256), 0, st,
                       dout, q, probs, ws, dk, dv, B, Tq, Tk, H, hs, scale);
  }
  AVDF_CHECK_LAUNCH("global_attn_bwd");
  return AVDF_OK;
}

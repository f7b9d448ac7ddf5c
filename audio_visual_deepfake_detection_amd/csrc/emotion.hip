// Emotion2Vec (data2vec 2.0 audio) encoder glue -- frozen extractor, forward only.
// Reference: audio_feature/emotion_audio/upstream/models/{emotion2vec.py:97-188, audio.py:52-138,
// base.py:189-329,538-642, modules.py:184-322} plus fairseq's ConvFeatureExtractionModel (named at audio.py:12-17).
// The 512-channel strided convs, every Linear, Q.K^T and P.V run on the fp32 MFMA GEMM (gemm_conv.hip,
// avdf_conv1d_fwd_ex); this file holds what is not GEMM-shaped:
//   * first feature-extractor layer (1 input channel, k=10, stride 5),
//   * grouped positional conv (768 ch, 16 groups, k=19, "same" padding),
//   * softmax with the symmetric ALiBi bias  -slope_h * |i - j| * scale  (base.py:538-642).
// Layout: tokens time-major [T, C]; attention scores [Tq, ld] per head.
#include "common.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// y[t, n] = sum_k pcm[t*S + k] * w[n, k]   (Cin = 1, no padding, no bias)
__global__ __launch_bounds__(256) void wave_conv_c1_kernel(const float* __restrict__ pcm, const float* __restrict__ w,
                                                           float* __restrict__ y, long T, int C, int K, int S) {
  const int C4 = C >> 2;
  const long total = T * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int n = (int)(i % C4) * 4;
    const long t = i / C4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < K; ++k) {
      const float xv = pcm[t * S + k];
      acc.x += xv * w[(n + 0) * K + k]; acc.y += xv * w[(n + 1) * K + k];
      acc.z += xv * w[(n + 2) * K + k]; acc.w += xv * w[(n + 3) * K + k];
    }
    st4(y + t * C + n, acc);
  }
}

// same, K <= 16 and C/4 a divisor of 256: a thread keeps the K taps of its 4 channels in registers and walks over
// tokens, so the kernel is bound by the 2 KB output row per token (201 MB per 768-frame clip), not by weight re-loads
__global__ __launch_bounds__(256) void wave_conv_c1_reg_kernel(const float* __restrict__ pcm, const float* __restrict__ w,
                                                               float* __restrict__ y, long T, int C, int K, int S) {
  const int C4 = C >> 2, lanes = 256 / C4;
  const int n = (threadIdx.x % C4) * 4, tl = threadIdx.x / C4;
  float wr[4][16];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < 16; ++k) wr[j][k] = k < K ? w[(n + j) * K + k] : 0.f;
  for (long t = (long)blockIdx.x * lanes + tl; t < T; t += (long)gridDim.x * lanes) {
    const float* px = pcm + t * S;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (k < K) {
        const float xv = px[k];
        acc.x += xv * wr[0][k]; acc.y += xv * wr[1][k]; acc.z += xv * wr[2][k]; acc.w += xv * wr[3][k];
      }
    }
    st4(y + t * C + n, acc);
  }
}

// y[t, g*CG + co] = bias + sum_{ci<CG, k<K} x[t + k - K/2, g*CG + ci] * w[g*CG + co, ci, k]
// one workgroup = (32-token tile, group); threads = CG output channels x 256/CG token lanes; x tile in LDS
template <int CG, int K>
__global__ __launch_bounds__(256) void grouped_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           int T, int C, int tiles) {
  constexpr int TT = 32, P = K / 2, R = TT + K - 1;
  constexpr int LANES = 256 / CG;                 // token lanes per output channel
  constexpr int PER = (TT + LANES - 1) / LANES;   // tokens per thread
  __shared__ float xs[R][CG + 1];
  const int g = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * TT;
  for (int i = threadIdx.x; i < R * CG; i += 256) {
    const int r = i / CG, ci = i % CG;
    const int t = t0 - P + r;
    xs[r][ci] = (t >= 0 && t < T) ? x[(size_t)t * C + g * CG + ci] : 0.f;
  }
  __syncthreads();
  const int co = threadIdx.x % CG, tl = threadIdx.x / CG;
  if (tl >= LANES) return;
  float acc[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) acc[q] = bias ? bias[g * CG + co] : 0.f;
  const float* wr = w + (size_t)(g * CG + co) * CG * K;
  for (int ci = 0; ci < CG; ++ci) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float wv = wr[ci * K + k];
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int tt = tl + q * LANES;
        if (tt < TT) acc[q] += wv * xs[tt + k][ci];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int t = t0 + tl + q * LANES;
    if (tl + q * LANES < TT && t < T) y[(size_t)t * C + g * CG + co] = acc[q];
  }
}

// in place: P[i, j] = softmax_j( S[i, j]*qk_scale - slope * |i - j| ) for j < T, 0 for T <= j < ld.  One wave per row.
__global__ __launch_bounds__(256) void softmax_alibi_kernel(float* __restrict__ s, int T, int ld, float qk_scale,
                                                            const float* __restrict__ slopes, float slope, long hstride) {
  // blockIdx.y = head: scores + head*hstride, slope = slopes[head] (or the scalar when slopes == NULL)
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= T) return;
  if (slopes) slope = slopes[blockIdx.y];
  float* row = s + (size_t)blockIdx.y * hstride + (size_t)i * ld;
  float m = -INFINITY;
  for (int j = lane; j < T; j += 64) {
    const float v = row[j] * qk_scale - slope * fabsf((float)(i - j));
    row[j] = v;
    m = fmaxf(m, v);
  }
  m = wave_max(m);
  float den = 0.f;
  for (int j = lane; j < T; j += 64) {
    const float e = __expf(row[j] - m);
    row[j] = e;
    den += e;
  }
  den = wave_sum(den);
  const float inv = 1.0f / den;
  for (int j = lane; j < ld; j += 64) row[j] = j < T ? row[j] * inv : 0.f;
}

// same, the row held in registers (ld % 4 == 0, ld <= 256 * NV): one 16-byte read and one 16-byte write per element
// group instead of three scalar read-modify-write passes through L2
template <int NV>
__global__ __launch_bounds__(256) void softmax_alibi_reg_kernel(float* __restrict__ s, int T, int ld, float qk_scale,
                                                                const float* __restrict__ slopes, float slope, long hstride) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= T) return;
  if (slopes) slope = slopes[blockIdx.y];
  float* row = s + (size_t)blockIdx.y * hstride + (size_t)i * ld;
  float v[NV][4];
  float m = -INFINITY;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int j4 = (q * 64 + lane) * 4;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j4 < ld) x = *reinterpret_cast<const float4*>(row + j4);
    const float e[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j4 + u;
      v[q][u] = j < T ? e[u] * qk_scale - slope * fabsf((float)(i - j)) : -INFINITY;
      m = fmaxf(m, v[q][u]);
    }
  }
  m = wave_max(m);
  float den = 0.f;
#pragma unroll
  for (int q = 0; q < NV; ++q)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[q][u] = __expf(v[q][u] - m);          // exp(-inf) = 0 for the padding columns
      den += v[q][u];
    }
  den = wave_sum(den);
  const float inv = 1.0f / den;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int j4 = (q * 64 + lane) * 4;
    if (j4 < ld) *reinterpret_cast<float4*>(row + j4) = make_float4(v[q][0] * inv, v[q][1] * inv, v[q][2] * inv, v[q][3] * inv);
  }
}

static void launch_softmax_alibi(float* scores, int H, int T, int ld, long hstride, float qk_scale, const float* slopes,
                                 float slope, hipStream_t st) {
  const dim3 grid((T + 3) / 4, H);
  const bool vec = (ld % 4 == 0) && (hstride % 4 == 0) && ((reinterpret_cast<uintptr_t>(scores) & 15) == 0);
  const int nv = (ld + 255) / 256;
  if (vec && nv <= 2) hipLaunchKernelGGL(softmax_alibi_reg_kernel<2>, grid, dim3(256), 0, st, scores, T, ld, qk_scale, slopes, slope, hstride);
  else if (vec && nv <= 4) hipLaunchKernelGGL(softmax_alibi_reg_kernel<4>, grid, dim3(256), 0, st, scores, T, ld, qk_scale, slopes, slope, hstride);
  else if (vec && nv <= 6) hipLaunchKernelGGL(softmax_alibi_reg_kernel<6>, grid, dim3(256), 0, st, scores, T, ld, qk_scale, slopes, slope, hstride);
  else if (vec && nv <= 8) hipLaunchKernelGGL(softmax_alibi_reg_kernel<8>, grid, dim3(256), 0, st, scores, T, ld, qk_scale, slopes, slope, hstride);
  else hipLaunchKernelGGL(softmax_alibi_kernel, grid, dim3(256), 0, st, scores, T, ld, qk_scale, slopes, slope, hstride);
}

inline int grid_for(long total) {
  long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int avdf_wave_conv_c1(const float* pcm, const float* w, float* y, int64_t n_samples, int64_t Tout, int C,
                                 int K, int stride, void* stream) {
  AVDF_REQUIRE(pcm && w && y && Tout > 0 && C > 0 && C % 4 == 0 && K > 0 && stride > 0, "wave_conv_c1: bad args");
  AVDF_REQUIRE((Tout - 1) * stride + K <= n_samples, "wave_conv_c1: Tout exceeds the valid-conv length");
  if (K <= 16 && C / 4 <= 256 && 256 % (C / 4) == 0) {
    const long lanes = 256 / (C / 4);
    long blocks = (Tout + lanes - 1) / lanes;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wave_conv_c1_reg_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pcm, w, y,
                       (long)Tout, C, K, stride);
  } else {
    hipLaunchKernelGGL(wave_conv_c1_kernel, dim3(grid_for(Tout * C / 4)), dim3(256), 0, (hipStream_t)stream, pcm, w, y,
                       (long)Tout, C, K, stride);
  }
  AVDF_CHECK_LAUNCH("wave_conv_c1");
  return AVDF_OK;
}

extern "C" int avdf_grouped_conv_pos(const float* x, const float* w, const float* bias, float* y, int T, int C,
                                     int groups, int K, void* stream) {
  AVDF_REQUIRE(x && w && y && T > 0 && C > 0 && groups > 0 && C % groups == 0, "grouped_conv_pos: bad args");
  AVDF_REQUIRE(C / groups == 48 && K == 19, "grouped_conv_pos: only 48 channels/group, k=19 (emotion2vec conv_pos) is built");
  const int tiles = cdiv(T, 32);
  hipLaunchKernelGGL((grouped_conv_kernel<48, 19>), dim3(groups * tiles), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, T, C, tiles);
  AVDF_CHECK_LAUNCH("grouped_conv_pos");
  return AVDF_OK;
}

extern "C" int avdf_softmax_alibi(float* scores, int T, int ld, float qk_scale, float slope, void* stream) {
  AVDF_REQUIRE(scores && T > 0 && ld >= T, "softmax_alibi: bad args");
  launch_softmax_alibi(scores, 1, T, ld, 0L, qk_scale, nullptr, slope, (hipStream_t)stream);
  AVDF_CHECK_LAUNCH("softmax_alibi");
  return AVDF_OK;
}

// all heads of a layer in one launch: scores [H][T][ld] (head stride in floats), slopes [H] on the device
extern "C" int avdf_softmax_alibi_heads(float* scores, int H, int T, int ld, int64_t head_stride, float qk_scale,
                                        const float* slopes, void* stream) {
  AVDF_REQUIRE(scores && slopes && H > 0 && T > 0 && ld >= T && head_stride >= (int64_t)T * ld, "softmax_alibi_heads: bad args");
  launch_softmax_alibi(scores, H, T, ld, (long)head_stride, qk_scale, slopes, 0.f, (hipStream_t)stream);
  AVDF_CHECK_LAUNCH("softmax_alibi_heads");
  return AVDF_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Fused attention of AltAttention.forward (upstream/models/modules.py:280-322): softmax(q k^T * scale + alibi) v for
// all heads of a layer in ONE launch, the [H, T, T] score matrix never leaves the chip (it was 113 MB per layer per
// 768-frame clip: written by the QK^T GEMM, rewritten by the softmax, read by the PV GEMM).
//
// Exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), flash-style online softmax.  One workgroup = 2 waves = 64 queries of one
// head; each wave owns 32 queries and walks the keys in tiles of 32:
//   S^T = K Q^T      A = K tile from LDS [key][d], B = Q^T from registers     -> D rows = keys, cols = queries
//       so the 16 accumulator registers of a lane are 16 keys of ONE query (lane & 31): the row max / row sum of the
//       softmax are 16 in-register values + one xor-32 shuffle (the two lane halves hold different keys);
//   O^T += V^T P^T   A = V^T from LDS, B = P^T: accumulator register s of the score tile IS the B operand of k-step s
//       when the contraction visits the keys in the accumulator's own row order j(s, h) = (s&3) + 8(s>>2) + 4h
//       (h = lane >> 5) -- the A operand reads V rows in that order, so P never moves between registers or through LDS.
// ALiBi: + slope_h * -|i - j| (base.py:538-604, scaled per head by the clamped alibi_scale); keys >= T are masked.
// ------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int kFaHD = 64;                 // head dim of every released checkpoint (768 / 12)
constexpr int kFaLD = kFaHD + 1;          // odd LDS row stride: the per-lane ds_read_b32 operand fetches are conflict-free

__global__ __launch_bounds__(128) void flash_attn_alibi_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, float* __restrict__ out,
                                                               const float* __restrict__ slopes, int T, int ld, float scale,
                                                               long clip_stride) {
  __shared__ float Ks[2][32 * kFaLD];
  __shared__ float Vs[2][32 * kFaLD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int head = blockIdx.y;
  const size_t base = (size_t)blockIdx.z * clip_stride + (size_t)head * kFaHD;
  q += base; k += base; v += base; out += base;
  const int q0 = blockIdx.x * 64;
  const int qi = q0 + w * 32 + li;                       // this lane's query row
  const float slope = slopes ? slopes[head] : 0.f;

  // ---- Q^T operand: qreg[s] = scale * Q[qi][2s + lh], staged through LDS (coalesced global reads) ----------------
  float qreg[32];
  {
    float* Qs = &Ks[0][0];                               // 64 x 65 floats fit the two K buffers (2 x 32 x 65)
    for (int e = tid; e < 64 * (kFaHD / 4); e += 128) {
      const int r = e / (kFaHD / 4), c4 = e % (kFaHD / 4);
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q0 + r < T) t = *reinterpret_cast<const float4*>(q + (size_t)(q0 + r) * ld + c4 * 4);
      float* d = Qs + r * kFaLD + c4 * 4;
      d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 32; ++s) qreg[s] = Qs[(w * 32 + li) * kFaLD + 2 * s + lh] * scale;
    __syncthreads();
  }

  floatx16 o0, o1;                                       // O^T accumulators: d-block 0 / 1, rows = d, cols = queries
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m = -INFINITY, l = 0.f;

  const int ntiles = (T + 31) / 32;
  float4 rk[4], rv[4];                                   // register prefetch of the next K / V tile (512 float4 each / 128 threads)
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 128, r = e / (kFaHD / 4), c4 = e % (kFaHD / 4);
      const int row = kt * 32 + r;
      rk[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      rv[i] = rk[i];
      if (row < T) {
        rk[i] = *reinterpret_cast<const float4*>(k + (size_t)row * ld + c4 * 4);
        rv[i] = *reinterpret_cast<const float4*>(v + (size_t)row * ld + c4 * 4);
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 128, r = e / (kFaHD / 4), c4 = e % (kFaHD / 4);
      float* dk = &Ks[buf][r * kFaLD + c4 * 4];
      float* dv = &Vs[buf][r * kFaLD + c4 * 4];
      dk[0] = rk[i].x; dk[1] = rk[i].y; dk[2] = rk[i].z; dk[3] = rk[i].w;
      dv[0] = rv[i].x; dv[1] = rv[i].y; dv[2] = rv[i].z; dv[3] = rv[i].w;
    }
  };
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < ntiles) load_tile(kt + 1);
    const float* Kb = Ks[buf];
    const float* Vb = Vs[buf];
    // ---- S^T tile: rows = 32 keys, cols = 32 queries ------------------------------------------------------------
    floatx16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s)
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(Kb[li * kFaLD + 2 * s + lh], qreg[s], st, 0, 0, 0);
    // ---- online softmax over the 16 keys this lane holds (+ the other lane half) ---------------------------------
    float mloc = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int dist = qi > j ? qi - j : j - qi;
      float x = st[r] - slope * (float)dist;
      if (j >= T) x = -INFINITY;
      st[r] = x;
      mloc = fmaxf(mloc, x);
    }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m, mloc);                  // finite from the first tile on (key 0 always exists)
    const float alpha = __expf(m - m_new);
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __expf(st[r] - m_new);
      st[r] = p;
      lsum += p;
    }
    lsum += __shfl_xor(lsum, 32, 64);
    l = l * alpha + lsum;
    m = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    // ---- O^T += V^T P^T: k-step s contracts key j(s, lh); B operand = st[s] ---------------------------------------
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int jr = (s & 3) + 8 * (s >> 2) + 4 * lh;
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vb[jr * kFaLD + li], st[s], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vb[jr * kFaLD + 32 + li], st[s], o1, 0, 0, 0);
    }
    if (kt + 1 < ntiles) store_tile(buf ^ 1);
    __syncthreads();
  }
  // ---- normalise, transpose through LDS, coalesced row stores ------------------------------------------------------
  const float inv = 1.f / l;
  float* Os = &Ks[0][0];                                 // [64 queries][65]
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = (r & 3) + 8 * (r >> 2) + 4 * lh;
    Os[(w * 32 + li) * kFaLD + d] = o0[r] * inv;
    Os[(w * 32 + li) * kFaLD + 32 + d] = o1[r] * inv;
  }
  __syncthreads();
  for (int e = tid; e < 64 * (kFaHD / 4); e += 128) {
    const int r = e / (kFaHD / 4), c4 = e % (kFaHD / 4);
    if (q0 + r < T) {
      const float* s_ = Os + r * kFaLD + c4 * 4;
      *reinterpret_cast<float4*>(out + (size_t)(q0 + r) * ld + c4 * 4) = make_float4(s_[0], s_[1], s_[2], s_[3]);
    }
  }
}

}  // namespace

// q, k, v, out: [B][Tp rows][ld floats], head h = columns [h*64, h*64 + 64); rows >= T are ignored (out rows >= T untouched)
extern "C" int avdf_flash_attn_alibi(const float* q, const float* k, const float* v, float* out, const float* slopes, int B,
                                     int T, int H, int hd, int ld, int64_t clip_stride, float scale, void* stream) {
  AVDF_REQUIRE(q && k && v && out && B > 0 && T > 0 && H > 0, "flash_attn_alibi: bad args");
  AVDF_REQUIRE(hd == kFaHD, "flash_attn_alibi: head dim %d not supported (64 only)", hd);
  AVDF_REQUIRE(ld >= H * hd && ld % 4 == 0 && clip_stride % 4 == 0, "flash_attn_alibi: ld / clip stride must be multiples of 4 floats");
  AVDF_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0, "flash_attn_alibi: 16-byte alignment");
  AVDF_REQUIRE(H < 65536 && B < 65536, "flash_attn_alibi: grid limits");
  hipLaunchKernelGGL(flash_attn_alibi_kernel, dim3((T + 63) / 64, H, B), dim3(128), 0, (hipStream_t)stream, q, k, v, out, slopes, T,
                     ld, scale, (long)clip_stride);
  AVDF_CHECK_LAUNCH("flash_attn_alibi");
  return AVDF_OK;
}

// Row-block fused kernels of the conv-attention TransformerBlock (gfx950, exact fp32 on v_mfma_f32_32x32x2_f32).
//
// A workgroup owns R = 32 complete token rows of one clip (all C = 256 channels), so everything that is row-local --
// LayerNorm, the 1x1 projections (a [32, 256] x [256, 256] GEMM whose A operand never leaves LDS), bias, mask, the
// AffineDropPath residual -- chains inside one launch; the two operators that look at neighbouring rows (depthwise k3:
// +-1, banded attention: +-w) read a halo of their INPUT, which a previous launch has completed.  One attention
// half-block (libs/modeling/blocks.py:1152-1224, 1307-1313 / 783-876) is therefore two launches instead of five:
//
//   avdf_qkv_fused_fwd   : LN_a -> depthwise k3 (stride s) * mask -> LN_b -> 1x1 projection, for q, k and v
//                          (replaces avdf_chain_fwd[_map] + avdf_conv1d_fwd_grouped; the [M, C] LN_b outputs are not
//                          re-read from HBM, in inference they are not written at all)
//   avdf_attn_out_fused_fwd : banded softmax(q k^T) v -> 1x1 output projection (+bias, *mask) -> skip*mask +
//                          scale*keep*proj -> LN2          (replaces avdf_local_attn_fwd_ex + avdf_conv1d_fwd +
//                          avdf_residual_ln_fwd; the attention output never reaches HBM in inference)
//
// GEMM core: 4 waves, wave w owns output columns [64w, 64w+64) = two 32x32 accumulator tiles; A = the workgroup's
// [32][256] row block in LDS (row stride 260 floats: a ds_read_b128 of 4 consecutive k by 16 different rows touches
// every bank once); W [n][k] as it lies in the Conv1d weight, streamed in K-chunks of 32 through LDS (row stride 36)
// with the next chunk's global loads in flight in registers under the MFMAs of the current one.  A lane's float4
// carries k = kb + 4*(lane>>5) + j for MFMA j of an 8-deep step -- the same pairing for A and B, so the contraction is
// complete after 4 MFMAs per tile.  Per workgroup and projection: 2 x 128 MFMAs per wave = 16.4 k cycles.
//
// Algorithmic work per token (C = 256): qkv: 3 x 2*C*C = 393 kFLOP, out-projection 131 kFLOP; HBM bytes per token in
// inference: qkv reads 1-3 KB (x of each distinct input) and writes 3 KB, attn+out reads 3 KB + 1 KB skip and
// writes 2 KB.  MFMA-bound (fp32): 157.3 TFLOP/s peak.
#include "../../../audio_visual_deepfake_detection_amd/csrc/common.h"

namespace {

constexpr int kC = 256;          // channels (the only width of the fused path; other widths take the per-op kernels)
constexpr int kR = 32;           // token rows per workgroup
constexpr int kKC = 32;          // K-chunk of the weight stream
constexpr int kAS = kC + 4;      // LDS row stride of the A row block
constexpr int kWS = kKC + 4;     // LDS row stride of a weight chunk [n][kKC]

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4(float a) { return make_float4(a, a, a, a); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 scl4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
  return make_float4(a.x * b.x + c.x, a.y * b.y + c.y, a.z * b.z + c.z, a.w * b.w + c.w);
}
__device__ __forceinline__ float hsum4(float4 a) { return a.x + a.y + a.z + a.w; }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ void load_taps(const float* __restrict__ w, int c, float4 tap[3]) {
  const float4 a = ld4(w + c * 3), b = ld4(w + c * 3 + 4), d = ld4(w + c * 3 + 8);
  tap[0] = make_float4(a.x, a.w, b.z, d.y);
  tap[1] = make_float4(a.y, b.x, b.w, d.z);
  tap[2] = make_float4(a.z, b.y, d.x, d.w);
}
__device__ __forceinline__ int map_row(int t, int sh) { return sh > 0 ? (t >> sh) : (sh < 0 ? (t << (-sh)) : t); }

// ---- weight stream: chunk kc of W [256][256] (row n, 32 consecutive k) -> 8 float4 per thread -----------------------
struct WChunk { float4 v[8]; };
__device__ __forceinline__ void wchunk_load(WChunk& r, const float* __restrict__ W, int kc, int tid) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = i * 256 + tid, n = q >> 3, k4 = q & 7;          // 8 consecutive threads: one 128-byte run of row n
    r.v[i] = ld4(W + (size_t)n * kC + kc * kKC + k4 * 4);
  }
}
__device__ __forceinline__ void wchunk_store(const WChunk& r, float* __restrict__ wbuf, int tid) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = i * 256 + tid, n = q >> 3, k4 = q & 7;
    st4(wbuf + n * kWS + k4 * 4, r.v[i]);
  }
}

// acc[t] (+)= abuf[32][256] x W^T, columns [64*wid + 32*t, +32).  `pre` holds chunk 0 on entry (loaded by the caller as
// early as it likes).  Ends with a barrier: abuf and wbuf are free when it returns.
__device__ __forceinline__ void rowblock_gemm(const float* __restrict__ abuf, float* __restrict__ wbuf,
                                              const float* __restrict__ W, WChunk& pre, floatx16 (&acc)[2], int tid) {
  const int lane = tid & 63, wid = tid >> 6;
  const float* ap = abuf + (lane & 31) * kAS + (lane >> 5) * 4;
  const float* bp = wbuf + (wid * 64 + (lane & 31)) * kWS + (lane >> 5) * 4;
#pragma unroll 1
  for (int kc = 0; kc < kC / kKC; ++kc) {
    wchunk_store(pre, wbuf, tid);
    __syncthreads();
    if (kc + 1 < kC / kKC) wchunk_load(pre, W, kc + 1, tid);
#pragma unroll
    for (int kk = 0; kk < kKC / 8; ++kk) {
      const float4 a4 = ld4(ap + kc * kKC + kk * 8);
      const float4 b0 = ld4(bp + kk * 8), b1 = ld4(bp + 32 * kWS + kk * 8);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b0.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b1.x, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b0.y, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b1.y, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b0.z, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b1.z, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b0.w, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b1.w, acc[1], 0, 0, 0);
    }
    __syncthreads();
  }
}

// =====================================================================================================================
// F1: LN_a -> depthwise k3 (stride S) * mask -> LN_b -> 1x1 projection (+bias), one branch (q, k or v) per blockIdx.y
// =====================================================================================================================
struct QkvFusedArgs {
  const float* x[3]; const float* ga[3]; const float* ba[3]; const float* w[3]; const float* gb[3]; const float* bb[3];
  const float* pw[3]; const float* pb[3];     // projection weight [C][C] (Conv1d [C, C, 1] as it lies), bias [C] or null
  float* u[3]; float* y[3]; float* sa[3]; float* sb[3];   // what the backward kernels read; all null in inference
  float* o[3];                                // projection outputs [B, Tout, C]
  int xsh[3]; int xT[3];                      // nearest resampling of a branch's input folded into the row address (chain.hip)
};

template <int S>
__global__ __launch_bounds__(256, (S == 1 ? 2 : 1)) void qkv_fused_fwd_kernel(QkvFusedArgs a, const float* __restrict__ mask, int B, int Tin,
                                                               int Tout, float eps, int tiles) {
  constexpr int RZ = (kR - 1) * S + 3;
  constexpr int ZW = (RZ * kC > kC * kWS) ? RZ * kC : kC * kWS;
  __shared__ __attribute__((aligned(16))) float smem[kR * kAS + ZW];
  float* ybuf = smem;                 // [kR][kAS]   LN_b output = A operand
  float* zw = smem + kR * kAS;        // [RZ][kC] LN_a output during the chain, then the weight chunks [kC][kWS]
  const int br = blockIdx.y;
  const int b = blockIdx.x / tiles, to0 = (blockIdx.x % tiles) * kR;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int c = lane * 4;
  constexpr float invC = 1.0f / (float)kC;

  WChunk pre;
  wchunk_load(pre, a.pw[br], 0, tid);          // in flight under the whole chain phase

  // ---- z = LN_a(x) for input rows to0*S-1 .. (to0+R-1)*S+1 (zero rows outside the clip: the conv's padding) ----------
#ifndef BF_NO_CHAIN
  {
    const float* __restrict__ x = a.x[br];
    const float4 g = a.ga[br] ? ld4(a.ga[br] + c) : f4(1.f), bt = a.ga[br] ? ld4(a.ba[br] + c) : f4(0.f);
    float* mean_o = a.sa[br];
    float* rstd_o = a.sa[br] ? a.sa[br] + (size_t)B * Tin : nullptr;
    const int ti_lo = to0 * S - 1;
    const int xsh = a.xsh[br], xT = a.xT[br];
    constexpr int U = 3;                       // rows in flight per wave
#pragma unroll 1
    for (int r0 = wid; r0 < RZ; r0 += 4 * U) {
      float4 v[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int r = r0 + 4 * j, t = ti_lo + r;
        v[j] = f4(0.f);
        if (r < RZ && t >= 0 && t < Tin) v[j] = ld4(x + ((size_t)b * xT + map_row(t, xsh)) * kC + c);
      }
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int r = r0 + 4 * j, t = ti_lo + r;
        if (r >= RZ) break;
        float4 z = f4(0.f);
        if (t >= 0 && t < Tin) {
          const float mean = wave_sum(hsum4(v[j])) * invC;
          const float4 d = sub4(v[j], f4(mean));
          const float rstd = 1.0f / sqrtf(wave_sum(hsum4(mul4(d, d))) * invC + eps);
          z = fma4(scl4(d, rstd), g, bt);
          if (mean_o && lane == 0 && t >= to0 * S && t < (to0 + kR) * S) {
            const size_t row = (size_t)b * Tin + t;
            mean_o[row] = mean; rstd_o[row] = rstd;
          }
        }
        st4(zw + r * kC + c, z);
      }
    }
  }
  __syncthreads();
  // ---- u = dwconv(z) * mask, y = LN_b(u) -> ybuf (and HBM when the backward will need them) -------------------------
  {
    float4 tap[3];
    load_taps(a.w[br], c, tap);
    const float4 g = a.gb[br] ? ld4(a.gb[br] + c) : f4(1.f), bt = a.gb[br] ? ld4(a.bb[br] + c) : f4(0.f);
    float* mean_o = a.sb[br];
    float* rstd_o = a.sb[br] ? a.sb[br] + (size_t)B * Tout : nullptr;
#pragma unroll 2
    for (int i = wid; i < kR; i += 4) {
      const int t_o = to0 + i;
      float4 yv = f4(0.f);
      if (t_o < Tout) {
        const size_t row = (size_t)b * Tout + t_o;
        const float* zr = zw + (i * S) * kC + c;
        float4 u = mul4(ld4(zr), tap[0]);
        u = add4(u, mul4(ld4(zr + kC), tap[1]));
        u = add4(u, mul4(ld4(zr + 2 * kC), tap[2]));
        if (mask) u = scl4(u, mask[row]);
        if (a.u[br]) st4(a.u[br] + row * kC + c, u);
        const float mean = wave_sum(hsum4(u)) * invC;
        const float4 d = sub4(u, f4(mean));
        const float rstd = 1.0f / sqrtf(wave_sum(hsum4(mul4(d, d))) * invC + eps);
        yv = fma4(scl4(d, rstd), g, bt);
        if (a.y[br]) st4(a.y[br] + row * kC + c, yv);
        if (mean_o && lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
      }
      st4(ybuf + i * kAS + c, yv);
    }
  }
#endif
  __syncthreads();                              // z is consumed: its LDS becomes the weight buffer
  // ---- projection ----------------------------------------------------------------------------------------------------
  floatx16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#ifndef BF_NO_GEMM
  rowblock_gemm(ybuf, zw, a.pw[br], pre, acc, tid);
#endif
  float* __restrict__ o = a.o[br];
#ifdef BF_NO_EPI
  if (acc[0][0] == 12345.f)
#endif
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int col = wid * 64 + t * 32 + (lane & 31);
    const float bias = a.pb[br] ? a.pb[br][col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (to0 + i < Tout) o[((size_t)b * Tout + to0 + i) * kC + col] = acc[t][r] + bias;
    }
  }
}


// =====================================================================================================================
// F2: banded attention -> output projection (+bias, *mask) -> skip*mask + scale*keep*proj -> LN2
// =====================================================================================================================
struct AttnOutArgs {
  const float* q; const float* k; const float* v; const float* mask;      // [B, T, C] x3, [B, T] (key mask = row mask)
  const float* rel; const float* tw;                                      // optional [H, W] tables (use_rel_pe / use_time_weight)
  const float* pw; const float* pb;                                       // projection weight [C][C], bias [C] or null
  const float* skip; const float* scale; const float* keep;               // residual input [B, T, C]; AffineDropPath scale [C]; keep [B] or null
  const float* gamma; const float* beta;                                  // LN2
  float* att; float* probs; float* r; float* stats;                       // saved for the backward (null in inference)
  float* out; float* h;
};

template <int L>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = L >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);     // constant offsets -> DPP / swizzle
  return v;
}

// W = window (odd), L = lanes per head (hs / 4); H = 64 / L heads
template <int W, int L>
__global__ __launch_bounds__(256, 2) void attn_out_fused_fwd_kernel(AttnOutArgs a, int B, int T, float qk_scale, float eps,
                                                                    int tiles) {
  constexpr int w = W / 2, TOK = 4, NR = TOK + 2 * w, H = 64 / L;
  __shared__ __attribute__((aligned(16))) float smem[kR * kAS + kC * kWS];
  float* abuf = smem;                 // [kR][kAS]  attention output = A operand; afterwards the projected rows
  float* wbuf = smem + kR * kAS;      // weight chunks [kC][kWS]
  const int b = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * kR;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int c = lane * 4;
  const int hd = lane / L;
  constexpr float invC = 1.0f / (float)kC;
  const size_t base = (size_t)b * T;

  WChunk pre;
  wchunk_load(pre, a.pw, 0, tid);

  // ---- banded attention: a wave owns 8 consecutive rows, TOK at a time with their NR key / value rows in registers ----
  {
    float relb[W], twb[W];
#pragma unroll
    for (int j = 0; j < W; ++j) { relb[j] = a.rel ? a.rel[hd * W + j] : 0.f; twb[j] = a.tw ? a.tw[hd * W + j] : 1.f; }
#pragma unroll 1
    for (int run = 0; run < 8 / TOK; ++run) {
      const int i0 = wid * 8 + run * TOK, tq = t0 + i0;
      float4 kr[NR], vr[NR], qv[TOK];
      float pen[NR], qm[TOK];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int tk = tq - w + r;
        kr[r] = f4(0.f); vr[r] = f4(0.f); pen[r] = -INFINITY;
        if (tk >= 0 && tk < T) {
          kr[r] = ld4(a.k + (base + tk) * kC + c); vr[r] = ld4(a.v + (base + tk) * kC + c);
          pen[r] = (a.mask[base + tk] == 0.f) ? -1e4f : 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < TOK; ++i) {
        qv[i] = f4(0.f); qm[i] = 0.f;
        if (tq + i < T) { qv[i] = scl4(ld4(a.q + (base + tq + i) * kC + c), qk_scale); qm[i] = a.mask[base + tq + i]; }
      }
#pragma unroll
      for (int i = 0; i < TOK; ++i) {
        float4 o = f4(0.f);
        if (tq + i < T) {                        // wave-uniform
          float s[W];
          float m = -INFINITY;
#pragma unroll
          for (int j = 0; j < W; ++j) {
            s[j] = group_sum<L>(dot4(qv[i], kr[i + j])) + relb[j] + pen[i + j];        // -inf outside [0, T)
            m = fmaxf(m, s[j]);
          }
          float den = 0.f;
#pragma unroll
          for (int j = 0; j < W; ++j) { s[j] = __expf(s[j] - m); den += s[j]; }
          const float inv = qm[i] / den;          // masked query rows are zeroed after the softmax
#pragma unroll
          for (int j = 0; j < W; ++j) {
            float p = s[j] * inv;
            if (a.probs && (lane & (L - 1)) == 0) a.probs[((base + tq + i) * H + hd) * W + j] = p;
            p *= twb[j];
            o = fma4(f4(p), vr[i + j], o);
          }
          if (a.att) st4(a.att + (base + tq + i) * kC + c, o);
        }
        st4(abuf + (i0 + i) * kAS + c, o);
      }
    }
  }
  __syncthreads();
  // ---- output projection ----------------------------------------------------------------------------------------------
  floatx16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  rowblock_gemm(abuf, wbuf, a.pw, pre, acc, tid);            // ends with a barrier: abuf is free
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int col = wid * 64 + t * 32 + (lane & 31);
    const float bias = a.pb ? a.pb[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      abuf[i * kAS + col] = acc[t][r] + bias;
    }
  }
  __syncthreads();
  // ---- r = proj * mask;  out = skip * mask + scale * keep * r;  h = LN2(out): a wave per row --------------------------
  {
    const float4 sc = ld4(a.scale + c), g = ld4(a.gamma + c), bt = ld4(a.beta + c);
    const float fr = a.keep ? a.keep[b] : 1.f;
    float* mean_o = a.stats;
    float* rstd_o = a.stats ? a.stats + (size_t)B * T : nullptr;
#pragma unroll 2
    for (int i = wid; i < kR; i += 4) {
      const int t = t0 + i;
      if (t >= T) break;
      const size_t row = base + t;
      const float mk = a.mask[row];
      const float4 pr = scl4(ld4(abuf + i * kAS + c), mk);
      const float4 sk = ld4(a.skip + row * kC + c);
      if (a.r) st4(a.r + row * kC + c, pr);
      const float4 v = fma4(scl4(pr, fr), sc, scl4(sk, mk));
      st4(a.out + row * kC + c, v);
      const float mean = wave_sum(hsum4(v)) * invC;
      const float4 d = sub4(v, f4(mean));
      const float rstd = 1.0f / sqrtf(wave_sum(hsum4(mul4(d, d))) * invC + eps);
      st4(a.h + row * kC + c, fma4(scl4(d, rstd), g, bt));
      if (mean_o && lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
    }
  }
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

// q / k / v input chains AND their 1x1 projections in one launch (C = 256 only).  Arguments as avdf_chain_fwd_map
// (include/avdf.h) plus the projection weights [C, C(, 1)] / biases [C] (NULL: none) and outputs proj[i] [B, Tout, C].
// u / y / stats_a / stats_b: the tensors the backward entries (avdf_chain_bwd*, avdf_conv1d_bwd_*) read -- pass NULL
// tables in inference and they are never written.  xshift may be NULL (no resampling).
extern "C" int avdf_qkv_fused_fwd(const float* const* x, const float* const* gamma_a, const float* const* beta_a,
                                  const float* const* w, const float* const* gamma_b, const float* const* beta_b,
                                  const float* const* pw, const float* const* pb, const float* mask, float* const* u,
                                  float* const* y, float* const* stats_a, float* const* stats_b, float* const* proj,
                                  const int* xshift, int B, int Tin, int Tout, int C, int stride, float eps, void* stream) {
  AVDF_REQUIRE(x && gamma_a && beta_a && w && gamma_b && beta_b && pw && proj, "qkv_fused_fwd: null table");
  AVDF_REQUIRE(C == kC, "qkv_fused_fwd: C must be %d (got %d); other widths take avdf_chain_fwd + avdf_conv1d_fwd_grouped", kC, C);
  AVDF_REQUIRE(B > 0 && Tin > 0 && (stride == 1 || stride == 2) && Tout == (Tin - 1) / stride + 1, "qkv_fused_fwd: bad dims");
  AVDF_REQUIRE((u == nullptr) == (y == nullptr) && (u == nullptr) == (stats_a == nullptr) && (u == nullptr) == (stats_b == nullptr),
               "qkv_fused_fwd: u / y / stats_a / stats_b come together (training) or not at all (inference)");
  QkvFusedArgs a;
  for (int i = 0; i < 3; ++i) {
    AVDF_REQUIRE(x[i] && w[i] && pw[i] && proj[i], "qkv_fused_fwd: null branch pointer");
    AVDF_REQUIRE((gamma_a[i] == nullptr) == (beta_a[i] == nullptr) && (gamma_b[i] == nullptr) == (beta_b[i] == nullptr),
                 "qkv_fused_fwd: gamma/beta must come in pairs");
    AVDF_REQUIRE(!u || (u[i] && y[i] && stats_a[i] && stats_b[i]), "qkv_fused_fwd: null saved-tensor pointer");
    AVDF_REQUIRE(aligned16(x[i]) && aligned16(w[i]) && aligned16(pw[i]) && aligned16(proj[i]) && aligned16(gamma_a[i]) &&
                     aligned16(beta_a[i]) && aligned16(gamma_b[i]) && aligned16(beta_b[i]) && (!u || (aligned16(u[i]) && aligned16(y[i]))),
                 "qkv_fused_fwd: 16-byte alignment");
    a.x[i] = x[i]; a.ga[i] = gamma_a[i]; a.ba[i] = beta_a[i]; a.w[i] = w[i]; a.gb[i] = gamma_b[i]; a.bb[i] = beta_b[i];
    a.pw[i] = pw[i]; a.pb[i] = pb ? pb[i] : nullptr;
    a.u[i] = u ? u[i] : nullptr; a.y[i] = y ? y[i] : nullptr;
    a.sa[i] = stats_a ? stats_a[i] : nullptr; a.sb[i] = stats_b ? stats_b[i] : nullptr;
    a.o[i] = proj[i];
    const int sh = xshift ? xshift[i] : 0;
    AVDF_REQUIRE(sh > -16 && sh < 16, "qkv_fused_fwd: bad xshift");
    if (sh > 0) AVDF_REQUIRE(Tin % (1 << sh) == 0, "qkv_fused_fwd: Tin=%d is not a multiple of the up-sampling factor %d", Tin, 1 << sh);
    a.xsh[i] = sh;
    a.xT[i] = sh > 0 ? (Tin >> sh) : (sh < 0 ? (Tin << -sh) : Tin);
  }
  const int tiles = cdiv(Tout, kR);
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1)
    hipLaunchKernelGGL(qkv_fused_fwd_kernel<1>, dim3(B * tiles, 3), dim3(256), 0, st, a, mask, B, Tin, Tout, eps, tiles);
  else
    hipLaunchKernelGGL(qkv_fused_fwd_kernel<2>, dim3(B * tiles, 3), dim3(256), 0, st, a, mask, B, Tin, Tout, eps, tiles);
  AVDF_CHECK_LAUNCH("qkv_fused_fwd");
  return AVDF_OK;
}

// 1 if avdf_attn_out_fused_fwd has a kernel for this (C, heads, window) -- callers take the per-op entries otherwise
extern "C" int avdf_attn_out_fused_supported(int C, int H, int window) {
  if (C != kC || H <= 0 || 64 % H != 0) return 0;
  const int L = 64 / H;
  // head size <= 64 like avdf_local_attn_bwd_ex, which runs this launch's backward
  return (L == 8 || L == 16) && (window == 3 || window == 5 || window == 7 || window == 9);
}

// Banded attention + output projection + AffineDropPath residual + LN2 of one block in one launch (C = 256):
//   att = softmax_band(q k^T * hs^-0.5 [+ rel]) [* tw] v;  r = (att pw^T + pb) * mask;  out = skip * mask + scale * keep[b] * r;
//   h = LN(out) * gamma + beta.        Replaces avdf_local_attn_fwd_ex + avdf_conv1d_fwd + avdf_residual_ln_fwd.
// att / probs / r / stats ([2, B*T]: mean, rstd) are what avdf_local_attn_bwd_ex, the projection's backward and
// avdf_residual_ln_bwd read: pass them together (training) or all NULL (inference: never written).
extern "C" int avdf_attn_out_fused_fwd(const float* q, const float* k, const float* v, const float* mask, const float* rel,
                                       const float* tw, const float* pw, const float* pb, const float* skip,
                                       const float* scale, const float* keep, const float* gamma, const float* beta,
                                       float* att, float* probs, float* r, float* stats, float* out, float* h, int B, int T,
                                       int C, int H, int window, float eps, void* stream) {
  AVDF_REQUIRE(q && k && v && mask && pw && skip && scale && gamma && beta && out && h, "attn_out_fused_fwd: null pointer");
  AVDF_REQUIRE(avdf_attn_out_fused_supported(C, H, window), "attn_out_fused_fwd: unsupported C=%d heads=%d window=%d", C, H, window);
  AVDF_REQUIRE(B > 0 && T > 0, "attn_out_fused_fwd: bad dims");
  AVDF_REQUIRE((att == nullptr) == (probs == nullptr) && (att == nullptr) == (r == nullptr) && (att == nullptr) == (stats == nullptr),
               "attn_out_fused_fwd: att / probs / r / stats come together (training) or not at all (inference)");
  AVDF_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(pw) && aligned16(skip) && aligned16(scale) &&
                   aligned16(gamma) && aligned16(beta) && aligned16(att) && aligned16(r) && aligned16(out) && aligned16(h),
               "attn_out_fused_fwd: 16-byte alignment");
  AttnOutArgs a{q, k, v, mask, rel, tw, pw, pb, skip, scale, keep, gamma, beta, att, probs, r, stats, out, h};
  const int tiles = cdiv(T, kR), L = 64 / H;
  const float qk_scale = 1.0f / sqrtf((float)(C / H));
  hipStream_t st = (hipStream_t)stream;
#define AO_L(W_, L_) hipLaunchKernelGGL((attn_out_fused_fwd_kernel<W_, L_>), dim3(B * tiles), dim3(256), 0, st, a, B, T, qk_scale, eps, tiles)
#define AO_W(W_) do { if (L == 8) AO_L(W_, 8); else AO_L(W_, 16); } while (0)
  switch (window) {
    case 3: AO_W(3); break;
    case 5: AO_W(5); break;
    case 7: AO_W(7); break;
    default: AO_W(9); break;
  }
#undef AO_W
#undef AO_L
  AVDF_CHECK_LAUNCH("attn_out_fused_fwd");
  return AVDF_OK;
}

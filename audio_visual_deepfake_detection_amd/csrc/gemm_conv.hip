// Dense conv1d (K in {1,3}, stride in {1,2}) as an implicit GEMM on the exact-fp32 matrix
// cores of gfx950 (v_mfma_f32_32x32x2_f32).  Activations are time-major [B,T,C]; weights keep
// the reference Conv1d layout [Cout, Cin, K] and are consumed as they lie (a K-tile of a weight
// row is one contiguous run of KC*K floats), so no repacking happens at load_state_dict or
// after an optimiser step.
//
//   forward : y[(b,t), n]   = sum_{k,c} x[b, t*S+k-P, c] * w[n,c,k]        (M = B*Tout, N = Cout)
//   dgrad   : dx[(b,t'), c] = sum_{k,n} g[b, (t'+P-k)/S, n] * w[n,c,k]     (M = B*Tin,  N = Cin)
//   wgrad   : dw[n, (c,k)]  = sum_{b,t} g[b,t,n] * x[b, t*S+k-P, c]        (M = Cout,   N = Cin x K)
//
// Structure shared by the three kernels
//   * a workgroup = KG "K-groups" of 4 waves.  The 4 waves of a group tile the BM x BN output block
//     (WM x WN waves, MI x NI accumulator tiles of 32x32 each); the KG groups split every K-tile
//     between them and their partial accumulators are summed through LDS at the end (intra-block
//     split-K).  The problems of this path are small for a 256-CU chip (6144 x 256 outputs are only
//     1536 MFMA tiles for 1024 SIMDs), so parallelism has to come from K as well as from M x N;
//   * operand tiles are staged global -> registers -> LDS with a one-tile register prefetch (the
//     loads of tile k+1 are in flight under the MFMAs of tile k);
//   * LDS rows are padded to an odd stride so the per-lane ds_read_b32 operand fetch is conflict-free;
//   * the k=3 taps reuse ONE haloed activation tile, stride-2 convs read every other LDS row;
//   * wgrad additionally splits the token dimension over gridDim.z; partial tiles go to slabs that a
//     vectorised reducer folds (deterministic: no float atomics anywhere).
// MFMA operand maps (one f32 per lane): A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31],
//   D: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
//
// Replaces: MaskedConv1D / nn.Conv1d(k=1) / nn.Linear of libs/modeling/blocks.py:41-63,
// 957-966, 1291-1297, 1498-1505 and their autograd-generated backward.
#include "common.h"
#ifdef AVDF_PW_EXPERIMENTAL
#include "pw_gemm.h"          // tools/experimental/pw_gemm (not part of the product build)
#endif
#include <stdlib.h>

// phase-timestamp hook for tools/probe (empty in the product build)
#ifndef AVDF_PROBE
#define AVDF_PROBE(slot)
#endif

namespace {

template <int N>
struct F4Buf { float4 v[N]; };

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ int floordiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int round4(int a) { return (a + 3) / 4 * 4; }

// g = dy * mask[row] * act'(pre)
__device__ __forceinline__ float4 grad_prologue(const float* __restrict__ dy, const float* __restrict__ mask,
                                                const float* __restrict__ ypre, size_t row, size_t off, int act) {
  float4 g = ld4(dy + off);
  if (ypre != nullptr && act != AVDF_ACT_NONE) {
    const float4 p = ld4(ypre + off);
    g.x *= act_grad_f(p.x, act); g.y *= act_grad_f(p.y, act);
    g.z *= act_grad_f(p.z, act); g.w *= act_grad_f(p.w, act);
  }
  if (mask != nullptr) {
    const float m = mask[row];
    g.x *= m; g.y *= m; g.z *= m; g.w *= m;
  }
  return g;
}

// sum the accumulators of K-groups 1..KG-1 into group 0 through LDS (`red` aliases the tile buffers)
template <int NACC, int KG>
__device__ __forceinline__ void kgroup_reduce(floatx16 (&acc)[NACC], float* red, int kg, int w4, int lane) {
  if (KG == 1) return;
  __syncthreads();
  if (kg > 0) {
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((((kg - 1) * 4 + w4) * NACC + a) * 16 + r) * 64 + lane] = acc[a][r];
  }
  __syncthreads();
  if (kg == 0) {
#pragma unroll
    for (int g = 1; g < KG; ++g)
#pragma unroll
      for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] += red[((((g - 1) * 4 + w4) * NACC + a) * 16 + r) * 64 + lane];
  }
}

// ---- split-bf16 ("bf16x3") operands: x = hi + lo with hi = bf16(x), lo = bf16(x - hi); a*b ~ hi*hi + hi*lo + lo*hi
// keeps ~16 mantissa bits per operand (relative error ~2^-17 per product, fp32 accumulation) at 3 bf16 MFMAs
// (32x32x16) per 16-deep k-step instead of 8 fp32 MFMAs (32x32x2): ~5x the matrix-core rate.  Optional GEMM mode.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned short bf16_bits(__bf16 v) { return __builtin_bit_cast(unsigned short, v); }
__device__ __forceinline__ void split_bf16x4(float4 v, uint2& hi, uint2& lo) {
  const __bf16 h0 = (__bf16)v.x, h1 = (__bf16)v.y, h2 = (__bf16)v.z, h3 = (__bf16)v.w;
  const __bf16 l0 = (__bf16)(v.x - (float)h0), l1 = (__bf16)(v.y - (float)h1), l2 = (__bf16)(v.z - (float)h2),
               l3 = (__bf16)(v.w - (float)h3);
  hi.x = bf16_bits(h0) | ((unsigned)bf16_bits(h1) << 16); hi.y = bf16_bits(h2) | ((unsigned)bf16_bits(h3) << 16);
  lo.x = bf16_bits(l0) | ((unsigned)bf16_bits(l1) << 16); lo.y = bf16_bits(l2) | ((unsigned)bf16_bits(l3) << 16);
}

__device__ __forceinline__ bf16x8 zero_bf16x8() { return __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u)); }
__device__ __forceinline__ float f4_get(const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }
// Transposing stage for the split-bf16 backward GEMMs: a thread holds a 4x4 block -- r[j] = 4 consecutive columns of
// row (4*rq + j) -- and writes, for each of its 4 columns, the 4 consecutive ROW values as one 8-byte run of the hi and
// of the lo plane (the contraction index becomes the fast LDS axis, which is what the 8-deep MFMA fragments read).
// item -> (column quad, row quad): 4 lanes x 16 bytes along a global row, 16 row quads per wave = 128 contiguous LDS
// bytes per column (2-way bank conflicts instead of 16-way for a plain walk).
template <int C4N, int RQN>
__device__ __forceinline__ void tq_coord(int q, int& c4, int& rq) {
  if constexpr (C4N % 4 == 0 && RQN % 16 == 0) {
    const int l = q & 63, wv = q >> 6;
    c4 = (wv % (C4N / 4)) * 4 + (l & 3);
    rq = (wv / (C4N / 4)) * 16 + (l >> 2);
  } else {
    c4 = q % C4N;
    rq = q / C4N;
  }
}

// q-th float4 of a [rows][W4 float4] tile -> (row, c4).  A half-wave covers 4 rows x 8 float4 columns, so
// the 32 lanes that scatter the same component of their float4 into a row-major LDS tile whose row stride is
// = 1 (mod 32) hit 32 different banks (a plain row-major walk is a 4-way conflict); a wave still reads
// 8 x 128 contiguous bytes from HBM/L2.
template <int W4>
__device__ __forceinline__ void tile_coord(int q, int& row, int& c4) {
  if constexpr (W4 % 8 == 0) {
    constexpr int CB = W4 / 8;
    const int l = q & 63, wv = q >> 6;
    row = (wv / CB) * 8 + (l & 3) + 4 * (l >> 5);
    c4 = (wv % CB) * 8 + ((l >> 2) & 7);
  } else {
    row = q / W4;
    c4 = q % W4;
  }
}
constexpr int round8(int a) { return (a + 7) / 8 * 8; }

// Cooperative epilogue: every K-group parks its accumulators in LDS as a [KG][BM][BN+4] tile (aliases the
// operand buffers), then ALL threads of the workgroup sum the groups and hand one float4 of a row to `emit`
// (row-contiguous 16-byte stores instead of 4-byte stores from one K-group's waves only).
template <int BM, int BN, int KG, int WN, int MI, int NI, class Emit>
__device__ __forceinline__ void tile_epilogue(floatx16 (&acc)[MI * NI], float* smem, int kg, int wm, int wn, int li,
                                              int lh, int tid, Emit&& emit) {
  constexpr int LDC = BN + 4, NT = 64 * (BM / (MI * 32)) * WN * KG;
  __syncthreads();
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        smem[(kg * BM + row) * LDC + (wn * NI + ni) * 32 + li] = acc[mi * NI + ni][r];
      }
  __syncthreads();
  for (int q = tid; q < BM * BN / 4; q += NT) {
    const int row = q / (BN / 4), c = (q % (BN / 4)) * 4;
    float4 v = *reinterpret_cast<const float4*>(smem + row * LDC + c);
#pragma unroll
    for (int g = 1; g < KG; ++g) {
      const float4 u = *reinterpret_cast<const float4*>(smem + (g * BM + row) * LDC + c);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    emit(row, c, v);
  }
}

// Up to 3 same-shape problems per launch (the q / k / v projections of an attention block): the exp10 GEMMs are
// latency-bound (~9 us floor per launch at K = 256), so one 3x larger launch costs about as much as one small one.
struct FwdPtrs {
  const float* x[3]; const float* w[3]; const float* bias[3]; const float* mask[3]; float* y[3]; float* ypre[3];
  // optional fused residual epilogue (single-problem launches): yres = res + rscale[c] * rkeep[row / resT] * y
  // (AffineDropPath + residual of the MLP half of a block); y itself may then be NULL (inference keeps only yres)
  const float* res = nullptr; const float* rscale = nullptr; const float* rkeep = nullptr; float* yres = nullptr; int resT = 1;
  long bbs = 0;            // bias batch stride in floats (grouped convs: batch = channel group)
};
struct DgradPtrs {
  const float* dy[3]; const float* w[3]; const float* mask[3]; const float* ypre[3]; float* dx[3];
  // optional epilogue: dx *= xact'(xpre) -- the activation derivative of the layer that PRODUCED this layer's input,
  // folded into this data gradient (the GELU between fc1 and fc2: evaluated once per element here instead of once per
  // tile visit in the prologues of the producer's own backward GEMMs)
  const float* xpre[3] = {nullptr, nullptr, nullptr};
  int xact = AVDF_ACT_NONE;
};
struct WgradPtrs { const float* dy[3]; const float* x[3]; const float* mask[3]; const float* ypre[3]; float* out[3]; float* dbias[3]; };
template <class T>
__device__ __forceinline__ T pick3(T const (&a)[3], int p) { return p == 0 ? a[0] : (p == 1 ? a[1] : a[2]); }

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// LDS floats of a forward workgroup (same arithmetic as inside the kernel) and the waves-per-SIMD target that keeps as many
// workgroups co-resident as LDS allows: the vectorised K = 1 variant holds float4 fragments, and without the hint the
// compiler spends 6-8 more VGPRs than the co-residency of the 512- / 1024-thread plans can afford
template <int KS, int S, int WM, int WN, int MI, int NI, int KC, int KG, bool BF3, bool DB>
constexpr int fwd_smem_floats() {
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32, RA = (BM - 1) * S + KS, KB = KC * KS;
  constexpr bool VEC = (KS == 1) && !BF3 && ((KC / KG) % 8 == 0);
  constexpr int LDA = VEC ? KC + 4 : KC + 1, LDB = VEC ? KB + 4 : KB + 1;
  constexpr int TILE = round4(BF3 ? (RA * (KC + 8) + BN * (KB + 8)) : (round4(RA * LDA) + BN * LDB));
  return cmax((DB ? 2 : 1) * TILE, KG * BM * (BN + 4));
}
template <int KS, int S, int WM, int WN, int MI, int NI, int KC, int KG, bool BF3, bool DB>
constexpr int fwd_waves_per_simd() {
  constexpr bool VEC = (KS == 1) && !BF3 && ((KC / KG) % 8 == 0);
  constexpr int NT = 64 * WM * WN * KG;
  constexpr int wgs = (160 * 1024) / (fwd_smem_floats<KS, S, WM, WN, MI, NI, KC, KG, BF3, DB>() * 4);
  constexpr int w = wgs * NT / 256;
  return (VEC && NT >= 512 && KG < 16) ? (w > 8 ? 8 : (w < 1 ? 1 : w)) : 1;      // 16 K-groups: the hint would spill
}

template <int KS, int S, int WM, int WN, int MI, int NI, int KC, int KG, bool BF3 = false, bool DB = false>
__global__ __launch_bounds__(64 * WM * WN * KG, (fwd_waves_per_simd<KS, S, WM, WN, MI, NI, KC, KG, BF3, DB>())) void conv_fwd_kernel(
    FwdPtrs ptrs, int B, int Tin, int Cin, int Tout, int Cout, int act, int tilesT, int P, int ldx, int ldw, int ldy,
    long xbs, long wbs, long ybs) {
  // xbs / wbs / ybs: batch strides of x, w, y in floats (default Tin*ldx, 0, Tout*ldy; attention heads use
  // column-slice strides and one weight matrix -- the K or V^T slice -- per batch entry)
  const int prob = blockIdx.z;
  const float* __restrict__ x = pick3(ptrs.x, prob);
  const float* __restrict__ w = pick3(ptrs.w, prob);
  const float* __restrict__ bias = pick3(ptrs.bias, prob);
  const float* __restrict__ mask = pick3(ptrs.mask, prob);
  float* __restrict__ y = pick3(ptrs.y, prob);
  float* __restrict__ ypre = pick3(ptrs.ypre, prob);
  // P = left zero-padding (K/2 for the "same" convs of the model, 0 for the wav2vec2 feature extractor);
  // ldx / ldw / ldy = row strides of x, w and y in floats (sub-matrix views: attention heads, weight slices)
  static_assert(WM * WN == 4 || WM * WN == 2 || WM * WN == 1, "4 waves (64x32 tile: 2, 32x32 tile: 1) per K-group");
  static_assert(KC % (2 * KG) == 0, "K-tile must split evenly into k-pairs per group");
  constexpr int WPG = WM * WN, NT = 64 * WPG * KG;
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
  constexpr int RA = (BM - 1) * S + KS;
  // VEC (K = 1, exact fp32): rows padded by 4 floats instead of 1, so that staging is one ds_write_b128 per float4 and a
  // lane fetches FOUR k-steps of its operand row with one ds_read_b128 (16 lanes x 16 B at a 272-byte row stride cover
  // the 64 banks exactly once).  MFMA j of an 8-deep step then contracts k = {cc + j, cc + 4 + j}: the pairing is the
  // same for A and B, and a sum does not care about the order of its terms.  4x fewer LDS instructions per MFMA.
  constexpr bool VEC = (KS == 1) && !BF3 && ((KC / KG) % 8 == 0);
  constexpr int LDA = VEC ? KC + 4 : KC + 1;
  constexpr int KB = KC * KS;
  constexpr int LDB = VEC ? KB + 4 : KB + 1;
  constexpr int A4 = KC / 4, B4 = KB / 4;
  constexpr int NA = (round8(RA) * A4 + NT - 1) / NT, NB = (BN * B4 + NT - 1) / NT;
  constexpr int AS = round4(RA * LDA), BSZ = BN * LDB;
  constexpr int KCG = KC / KG;
  // BF3: two bf16 planes (hi, lo) per operand; rows padded by 8 elements (16 B) so the 16-byte fragment reads of the
  // 32 row-lanes are conflict-free; the B rows are tap-major [k][c] (8 consecutive channels per fragment)
  constexpr int LDAH = KC + 8, LDBH = KB + 8;
  constexpr int ASH = RA * LDAH, BSH = BN * LDBH;
  static_assert(!BF3 || KCG % 16 == 0, "bf16x3 needs 16-deep k-steps per K-group");
  // DB: two operand buffers -- the next stage is stored while other waves still compute from the current one, one
  // barrier per K-stage instead of two (long-K problems; costs co-residency, so the plan decides)
  constexpr int TILE = round4(BF3 ? (ASH + BSH) : (AS + BSZ));
  constexpr int SM = cmax((DB ? 2 : 1) * TILE, KG * BM * (BN + 4));
  static_assert(SM == fwd_smem_floats<KS, S, WM, WN, MI, NI, KC, KG, BF3, DB>(), "smem size");
  __shared__ __attribute__((aligned(16))) float smem[SM];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int kg = wid / WPG, w4 = wid % WPG;
  const int wm = w4 / WN, wn = w4 % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.x / tilesT, t0 = (blockIdx.x % tilesT) * BM;
  const int n0 = blockIdx.y * BN;

  floatx16 acc[MI * NI];
#pragma unroll
  for (int a = 0; a < MI * NI; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  F4Buf<NA> ra;
  F4Buf<NB> rb;
  auto load_tile = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int row, c4;
      if (BF3 || VEC) { row = (tid + i * NT) / A4; c4 = (tid + i * NT) % A4; } else tile_coord<A4>(tid + i * NT, row, c4);
      float4 v = zero4();
      if (row < RA) {
        const int c = c0 + c4 * 4;
        const int tr = t0 * S - P + row;
        if (tr >= 0 && tr < Tin && c < Cin) v = ld4(x + (size_t)b * xbs + (size_t)tr * ldx + c);
      }
      ra.v[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int row, c4;
      if (BF3 || VEC) { row = (tid + i * NT) / B4; c4 = (tid + i * NT) % B4; } else tile_coord<B4>(tid + i * NT, row, c4);
      float4 v = zero4();
      if (row < BN) {
        const int n = n0 + row, e = c0 * KS + c4 * 4;
        if (n < Cout && e < Cin * KS) v = ld4(w + (size_t)b * wbs + (size_t)n * ldw + e);
      }
      rb.v[i] = v;
    }
  };
  auto store_tile = [&](int boff) {
    float* As = smem + boff;
    float* Bs = As + AS;
    __bf16* Ah = reinterpret_cast<__bf16*>(smem + boff);
    __bf16* Al = Ah + ASH;
    __bf16* Bh = Al + ASH;
    __bf16* Bl = Bh + BSH;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int row, c4;
      if (BF3 || VEC) { row = (tid + i * NT) / A4; c4 = (tid + i * NT) % A4; } else tile_coord<A4>(tid + i * NT, row, c4);
      if (row < RA) {
        if (BF3) {
          uint2 hi, lo;
          split_bf16x4(ra.v[i], hi, lo);
          *reinterpret_cast<uint2*>(Ah + row * LDAH + c4 * 4) = hi;
          *reinterpret_cast<uint2*>(Al + row * LDAH + c4 * 4) = lo;
        } else if (VEC) {
          *reinterpret_cast<float4*>(As + row * LDA + c4 * 4) = ra.v[i];
        } else {
          float* d = As + row * LDA + c4 * 4;
          d[0] = ra.v[i].x; d[1] = ra.v[i].y; d[2] = ra.v[i].z; d[3] = ra.v[i].w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int row, c4;
      if (BF3 || VEC) { row = (tid + i * NT) / B4; c4 = (tid + i * NT) % B4; } else tile_coord<B4>(tid + i * NT, row, c4);
      if (row < BN) {
        if (BF3) {
          uint2 hi, lo;
          split_bf16x4(rb.v[i], hi, lo);
          if (KS == 1) {
            *reinterpret_cast<uint2*>(Bh + row * LDBH + c4 * 4) = hi;
            *reinterpret_cast<uint2*>(Bl + row * LDBH + c4 * 4) = lo;
          } else {                       // global order is (c, k) interleaved: scatter to the tap-major LDS row
            const unsigned short h[4] = {(unsigned short)hi.x, (unsigned short)(hi.x >> 16), (unsigned short)hi.y, (unsigned short)(hi.y >> 16)};
            const unsigned short l[4] = {(unsigned short)lo.x, (unsigned short)(lo.x >> 16), (unsigned short)lo.y, (unsigned short)(lo.y >> 16)};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int e = c4 * 4 + j, cc_ = e / KS, kk_ = e % KS;
              reinterpret_cast<unsigned short*>(Bh)[row * LDBH + kk_ * KC + cc_] = h[j];
              reinterpret_cast<unsigned short*>(Bl)[row * LDBH + kk_ * KC + cc_] = l[j];
            }
          }
        } else if (VEC) {
          *reinterpret_cast<float4*>(Bs + row * LDB + c4 * 4) = rb.v[i];
        } else {
          float* d = Bs + row * LDB + c4 * 4;
          d[0] = rb.v[i].x; d[1] = rb.v[i].y; d[2] = rb.v[i].z; d[3] = rb.v[i].w;
        }
      }
    }
  };

  const int nk = (Cin + KC - 1) / KC;
  AVDF_PROBE(0);
  load_tile(0);
  store_tile(0);
  __syncthreads();
  AVDF_PROBE(1);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile((kt + 1) * KC);
    const int coff = (DB && (kt & 1)) ? TILE : 0;
    float* As = smem + coff;
    float* Bs = As + AS;
    __bf16* Ah = reinterpret_cast<__bf16*>(smem + coff);
    __bf16* Al = Ah + ASH;
    __bf16* Bh = Al + ASH;
    __bf16* Bl = Bh + BSH;
    if constexpr (BF3) {
#pragma unroll
      for (int k = 0; k < KS; ++k) {
#pragma unroll
        for (int cc = 0; cc < KCG; cc += 16) {
          bf16x8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const int off = (((wm * MI + mi) * 32 + li) * S + k) * LDAH + kg * KCG + cc + lh * 8;
            ah[mi] = *reinterpret_cast<const bf16x8*>(Ah + off);
            al[mi] = *reinterpret_cast<const bf16x8*>(Al + off);
          }
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const int off = ((wn * NI + ni) * 32 + li) * LDBH + k * KC + kg * KCG + cc + lh * 8;
            bh[ni] = *reinterpret_cast<const bf16x8*>(Bh + off);
            bl[ni] = *reinterpret_cast<const bf16x8*>(Bl + off);
          }
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              floatx16 c_ = acc[mi * NI + ni];
              c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], c_, 0, 0, 0);
              c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], c_, 0, 0, 0);
              c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], c_, 0, 0, 0);
              acc[mi * NI + ni] = c_;
            }
        }
      }
    } else if constexpr (VEC) {
      const float* Ag = As + kg * KCG + lh * 4;
      const float* Bg = Bs + kg * KCG + lh * 4;
#pragma unroll
      for (int cc = 0; cc < KCG; cc += 8) {
        float4 a4[MI], b4[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a4[mi] = *reinterpret_cast<const float4*>(Ag + ((wm * MI + mi) * 32 + li) * LDA + cc);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b4[ni] = *reinterpret_cast<const float4*>(Bg + ((wn * NI + ni) * 32 + li) * LDB + cc);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi * NI + ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4_get(a4[mi], j), f4_get(b4[ni], j), acc[mi * NI + ni], 0, 0, 0);
      }
    } else {
    const float* Ag = As + kg * KCG;
    const float* Bg = Bs + kg * KCG * KS;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
#pragma unroll
      for (int cc = 0; cc < KCG; cc += 2) {
        float a[MI], bv[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          a[mi] = Ag[(((wm * MI + mi) * 32 + li) * S + k) * LDA + cc + lh];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          bv[ni] = Bg[((wn * NI + ni) * 32 + li) * LDB + (cc + lh) * KS + k];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi * NI + ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], bv[ni], acc[mi * NI + ni], 0, 0, 0);
      }
    }
    }
    if constexpr (DB) {
      if (kt + 1 < nk) store_tile((kt & 1) ? 0 : TILE);
      __syncthreads();
    } else {
      __syncthreads();
      AVDF_PROBE(2 + 2 * kt);
      if (kt + 1 < nk) {
        store_tile(0);
        __syncthreads();
      }
      AVDF_PROBE(3 + 2 * kt);
    }
  }
  AVDF_PROBE(14);
  const float* __restrict__ res = ptrs.res;
  const float* __restrict__ rscale = ptrs.rscale;
  const float* __restrict__ rkeep = ptrs.rkeep;
  float* __restrict__ yres = ptrs.yres;
  const int resT = ptrs.resT;
  const float* __restrict__ biasb = bias ? bias + (size_t)b * ptrs.bbs : nullptr;
  const bool vec = ((ldy & 3) == 0) && ((ybs & 3) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0) &&
                   (ypre == nullptr || (reinterpret_cast<uintptr_t>(ypre) & 15) == 0) &&
                   (res == nullptr || (((reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(yres) |
                                         reinterpret_cast<uintptr_t>(rscale)) & 15) == 0));
  tile_epilogue<BM, BN, KG, WN, MI, NI>(acc, smem, kg, wm, wn, li, lh, tid, [&](int row_, int c, float4 v) {
    const int t = t0 + row_, n = n0 + c;
    if (t >= Tout || n >= Cout) return;
    const size_t row = (size_t)b * Tout + t;
    const size_t yrow = (size_t)b * ybs + (size_t)t * ldy;
    const float m = mask ? mask[row] : 1.f;
    if (vec && n + 3 < Cout) {
      if (bias) { const float4 bb = ld4(biasb + n); v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w; }
      if (ypre) *reinterpret_cast<float4*>(ypre + yrow + n) = v;
      v.x = act_f(v.x, act) * m; v.y = act_f(v.y, act) * m; v.z = act_f(v.z, act) * m; v.w = act_f(v.w, act) * m;
      if (y) *reinterpret_cast<float4*>(y + yrow + n) = v;
      if (res) {
        const float kp = rkeep ? rkeep[row / resT] : 1.f;
        const float4 r4 = ld4(res + yrow + n), s4 = ld4(rscale + n);
        float4 o;
        o.x = r4.x + s4.x * kp * v.x; o.y = r4.y + s4.y * kp * v.y; o.z = r4.z + s4.z * kp * v.z; o.w = r4.w + s4.w * kp * v.w;
        *reinterpret_cast<float4*>(yres + yrow + n) = o;
      }
    } else {
      const float e[4] = {v.x, v.y, v.z, v.w};
      const float kp = (res && rkeep) ? rkeep[row / resT] : 1.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (n + j >= Cout) break;
        float u = e[j] + (bias ? biasb[n + j] : 0.f);
        if (ypre) ypre[yrow + n + j] = u;
        u = act_f(u, act) * m;
        if (y) y[yrow + n + j] = u;
        if (res) yres[yrow + n + j] = res[yrow + n + j] + rscale[n + j] * kp * u;
      }
    }
  });
  AVDF_PROBE(15);
}

// ------------------------------------------------------------------------------------------
// data gradient
// ------------------------------------------------------------------------------------------
template <int KS, int S, int WM, int WN, int MI, int NI, int KC, int KG, bool DB, bool BF3 = false>
constexpr int dgrad_smem_floats() {
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
  constexpr int RA = (BM + KS - 2) / S + 2;
  // BF3: hi + lo bf16 planes of the g tile [RA][KC+8] and of the TRANSPOSED weight tile [KS][BN][KC+8]
  // VECA (K = 1, fp32): g rows padded by 4 floats (16-byte staging stores and fragment reads, see conv_fwd_kernel)
  constexpr bool VECA = (KS == 1) && (S == 1) && !BF3 && ((KC / KG) % 8 == 0);
  constexpr int TILE = BF3 ? round4((RA + BN * KS) * (KC + 8)) : round4(round4(RA * (KC + (VECA ? 4 : 1))) + KC * BN * KS);
  return cmax((DB ? 2 : 1) * TILE, KG * BM * (BN + 4));
}

// body of one dgrad workgroup; (bx, by, bz) = the workgroup's coordinates in the (B*tilesT, c-tiles, problems) grid
template <int KS, int S, int WM, int WN, int MI, int NI, int KC, int KG, bool DB, bool BF3 = false>
__device__ __forceinline__ void conv_dgrad_body(
    float* __restrict__ smem, const DgradPtrs& ptrs, int B, int Tin, int Cin, int Tout, int Cout, int act, int tilesT,
    int bx, int by, int bz) {
  const int prob = bz;
  const float* __restrict__ dy = pick3(ptrs.dy, prob);
  const float* __restrict__ w = pick3(ptrs.w, prob);
  const float* __restrict__ mask = pick3(ptrs.mask, prob);
  const float* __restrict__ ypre = pick3(ptrs.ypre, prob);
  float* __restrict__ dx = pick3(ptrs.dx, prob);
  static_assert(WM * WN == 4 || WM * WN == 2 || WM * WN == 1, "4 waves (64x32 tile: 2, 32x32 tile: 1) per K-group");
  static_assert(KC % (2 * KG) == 0, "K-tile must split evenly into k-pairs per group");
  constexpr int WPG = WM * WN, NT = 64 * WPG * KG;
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32, P = KS / 2;
  constexpr int RA = (BM + KS - 2) / S + 2;
  constexpr bool VECA = (KS == 1) && (S == 1) && !BF3 && ((KC / KG) % 8 == 0);
  constexpr int LDA = VECA ? KC + 4 : KC + 1;
  constexpr int LDB = BN * KS;
  constexpr int A4 = KC / 4, B4 = LDB / 4;
  constexpr int NA = (round8(RA) * A4 + NT - 1) / NT;
  constexpr int AS = round4(RA * LDA), BSZ = KC * LDB;
  // BF3 (split-bf16 operands): g tile as hi / lo planes [RA][LDH]; weight tile transposed to [KS][BN][LDH] so that the
  // contraction index n is the fast axis of BOTH operands (8-deep fragments = one 16-byte LDS read per lane)
  constexpr int LDH = KC + 8;
  constexpr int ASH = RA * LDH, BSH = BN * KS * LDH;
  constexpr int NQ = KC / 4;                                      // n-quads of a K-tile
  constexpr int NBQ = (NQ * B4 + NT - 1) / NT;                    // 4x4 blocks of the weight tile per thread
  constexpr int NB = BF3 ? 4 * NBQ : (KC * B4 + NT - 1) / NT;
  constexpr int TILE = BF3 ? round4((RA + BN * KS) * LDH) : round4(AS + BSZ);
  static_assert(dgrad_smem_floats<KS, S, WM, WN, MI, NI, KC, KG, DB, BF3>() == cmax((DB ? 2 : 1) * TILE, KG * BM * (BN + 4)), "smem size");
  constexpr int KCG = KC / KG;
  static_assert(!BF3 || KCG % 16 == 0, "bf16x3 needs 16-deep k-steps per K-group");

  // PAR (k = 3, stride 2; round 4): an output row t' of a stride-2 data gradient sees ONE tap when t' is even (k = 1) and
  // TWO when it is odd (k = 0, 2).  With consecutive rows on the 32 lanes of an accumulator tile, every MFMA of every tap
  // ran with half of its rows multiplying zeros: 2x the matrix work (conv_dgrad_k3s2 sat at 34 TFLOP/s algorithmic).  Here
  // the 32-row sub-tiles of a workgroup take rows of ONE parity (sub-tile 2j: rows 64j + 2i, sub-tile 2j+1: rows 64j + 2i + 1),
  // so a (sub-tile, tap) pair is either all valid or skipped by a scalar branch.  Even sub-tiles then have one tap of work
  // and odd ones two: with two row sub-tiles per workgroup (WM = 2, MI = 1) the parity a wave works on alternates with its
  // K-group, so that the four waves of a SIMD (the four K-groups of one (wm, wn)) carry 2 x 1 + 2 x 2 taps, the same on
  // every SIMD; every (sub-tile, K-group) pair is still covered exactly once, and the K-group fold is by tile row as before.
  constexpr bool PAR = (KS == 3) && (S == 2) && (BM % 64 == 0);
  constexpr bool PSWZ = PAR && (WM == 2) && (MI == 1) && (KG % 2 == 0);
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wid / WPG, w4 = wid % WPG;
  const int wm = PSWZ ? ((w4 / WN) ^ (kg & 1)) : (w4 / WN), wn = w4 % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int b = bx / tilesT, tp0 = (bx % tilesT) * BM;
  const int c0 = by * BN;
  const int tA0 = floordiv(tp0 + P - (KS - 1), S);
  // tile row rho (0 .. BM-1, sub-tile = rho / 32) -> output row offset inside the tile
  auto tile_row = [](int rho) { return PAR ? ((rho & ~63) + ((rho & 31) << 1) + ((rho >> 5) & 1)) : rho; };

  floatx16 acc[MI * NI];
#pragma unroll
  for (int a = 0; a < MI * NI; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  // per (mi, k): LDS row of the dy tile this lane's output row reads, or -1 (stride-2 parity miss)
  int arow[MI][KS];
  bool use[MI][KS];                      // PAR: wave-uniform -- does this (sub-tile, tap) pair contribute at all?
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int num = tp0 + tile_row((wm * MI + mi) * 32 + li) + P - k;
      int r = -1;
      if (S == 1) r = num - tA0;
      else if (((num % S) + S) % S == 0) r = floordiv(num, S) - tA0;
      arow[mi][k] = (r >= 0 && r < RA) ? r : -1;
      use[mi][k] = PAR ? ((((wm * MI + mi) & 1) == 0) == (k == 1)) : true;
    }

  F4Buf<NA> ra;
  F4Buf<NB> rb;
  auto load_tile = [&](int nk0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int rr, c4;
      if (BF3 || VECA) { rr = (tid + i * NT) / A4; c4 = (tid + i * NT) % A4; } else tile_coord<A4>(tid + i * NT, rr, c4);
      float4 v = zero4();
      if (rr < RA) {
        const int t = tA0 + rr, n = nk0 + c4 * 4;
        if (t >= 0 && t < Tout && n < Cout) {
          const size_t row = (size_t)b * Tout + t;
          v = grad_prologue(dy, mask, ypre, row, row * Cout + n, act);
        }
      }
      ra.v[i] = v;
    }
    if constexpr (BF3) {
#pragma unroll
      for (int ii = 0; ii < NBQ; ++ii) {
        int c4, nq;
        tq_coord<B4, NQ>(tid + ii * NT, c4, nq);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 v = zero4();
          if (nq < NQ) {
            const int n = nk0 + nq * 4 + j, e = c0 * KS + c4 * 4;
            if (n < Cout && e < Cin * KS) v = ld4(w + (size_t)n * Cin * KS + e);
          }
          rb.v[ii * 4 + j] = v;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int q = tid + i * NT;
        float4 v = zero4();
        if (q < KC * B4) {
          const int n = nk0 + q / B4, e = c0 * KS + (q % B4) * 4;
          if (n < Cout && e < Cin * KS) v = ld4(w + (size_t)n * Cin * KS + e);
        }
        rb.v[i] = v;
      }
    }
  };
  auto store_tile = [&](int boff) {
    float* As = smem + boff;
    float* Bs = As + AS;
    __bf16* Ah = reinterpret_cast<__bf16*>(smem + boff);
    __bf16* Al = Ah + ASH;
    __bf16* Bh = Al + ASH;
    __bf16* Bl = Bh + BSH;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int rr, c4;
      if (BF3 || VECA) { rr = (tid + i * NT) / A4; c4 = (tid + i * NT) % A4; } else tile_coord<A4>(tid + i * NT, rr, c4);
      if (rr < RA) {
        if (BF3) {
          uint2 hi, lo;
          split_bf16x4(ra.v[i], hi, lo);
          *reinterpret_cast<uint2*>(Ah + rr * LDH + c4 * 4) = hi;
          *reinterpret_cast<uint2*>(Al + rr * LDH + c4 * 4) = lo;
        } else if (VECA) {
          *reinterpret_cast<float4*>(As + rr * LDA + c4 * 4) = ra.v[i];
        } else {
          float* d = As + rr * LDA + c4 * 4;
          d[0] = ra.v[i].x; d[1] = ra.v[i].y; d[2] = ra.v[i].z; d[3] = ra.v[i].w;
        }
      }
    }
    if constexpr (BF3) {
#pragma unroll
      for (int ii = 0; ii < NBQ; ++ii) {
        int c4, nq;
        tq_coord<B4, NQ>(tid + ii * NT, c4, nq);
        if (nq < NQ) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {                   // column e = (c, k) of the weight row -> LDS row [k][c]
            const int e = c4 * 4 + jj, cc_ = e / KS, kk_ = e % KS;
            const float4 v = make_float4(f4_get(rb.v[ii * 4 + 0], jj), f4_get(rb.v[ii * 4 + 1], jj),
                                         f4_get(rb.v[ii * 4 + 2], jj), f4_get(rb.v[ii * 4 + 3], jj));
            uint2 hi, lo;
            split_bf16x4(v, hi, lo);
            *reinterpret_cast<uint2*>(Bh + (kk_ * BN + cc_) * LDH + nq * 4) = hi;
            *reinterpret_cast<uint2*>(Bl + (kk_ * BN + cc_) * LDH + nq * 4) = lo;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int q = tid + i * NT;
        if (q < KC * B4) *reinterpret_cast<float4*>(Bs + (q / B4) * LDB + (q % B4) * 4) = rb.v[i];
      }
    }
  };

  const int nk = (Cout + KC - 1) / KC;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile((kt + 1) * KC);
    const int coff = (DB && (kt & 1)) ? TILE : 0;
    if constexpr (BF3) {
      const __bf16* Ah = reinterpret_cast<const __bf16*>(smem + coff);
      const __bf16* Al = Ah + ASH;
      const __bf16* Bh = Al + ASH;
      const __bf16* Bl = Bh + BSH;
#pragma unroll
      for (int k = 0; k < KS; ++k) {
#pragma unroll
        for (int nn = 0; nn < KCG; nn += 16) {
          bf16x8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const int r = arow[mi][k];
            const int off = (r >= 0 ? r : 0) * LDH + kg * KCG + nn + lh * 8;
            ah[mi] = r >= 0 ? *reinterpret_cast<const bf16x8*>(Ah + off) : zero_bf16x8();
            al[mi] = r >= 0 ? *reinterpret_cast<const bf16x8*>(Al + off) : zero_bf16x8();
          }
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const int off = (k * BN + (wn * NI + ni) * 32 + li) * LDH + kg * KCG + nn + lh * 8;
            bh[ni] = *reinterpret_cast<const bf16x8*>(Bh + off);
            bl[ni] = *reinterpret_cast<const bf16x8*>(Bl + off);
          }
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            if (!use[mi][k]) continue;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              floatx16 c_ = acc[mi * NI + ni];
              c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], c_, 0, 0, 0);
              c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], c_, 0, 0, 0);
              c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], c_, 0, 0, 0);
              acc[mi * NI + ni] = c_;
            }
          }
        }
      }
    } else if constexpr (VECA) {
      // the g operand: four k-steps per ds_read_b128; MFMA j contracts n = {nn + j, nn + 4 + j} (the weight rows follow)
      const float* Ag = smem + coff + kg * KCG + lh * 4;
      const float* Bg = smem + coff + AS + (kg * KCG + lh * 4) * LDB;
#pragma unroll
      for (int nn = 0; nn < KCG; nn += 8) {
        float4 a4[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int r = arow[mi][0];
          a4[mi] = r >= 0 ? *reinterpret_cast<const float4*>(Ag + r * LDA + nn) : zero4();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float bv[NI];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) bv[ni] = Bg[(nn + j) * LDB + (wn * NI + ni) * 32 + li];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi * NI + ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4_get(a4[mi], j), bv[ni], acc[mi * NI + ni], 0, 0, 0);
        }
      }
    } else {
    const float* Ag = smem + coff + kg * KCG;
    const float* Bg = smem + coff + AS + kg * KCG * LDB;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      bool any = false;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) any = any || use[mi][k];
      if (!any) continue;                                   // scalar branch: no sub-tile of this wave sees tap k
#pragma unroll
      for (int nn = 0; nn < KCG; nn += 2) {
        float a[MI], bv[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int r = arow[mi][k];
          a[mi] = r >= 0 ? Ag[r * LDA + nn + lh] : 0.f;
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          bv[ni] = Bg[(nn + lh) * LDB + ((wn * NI + ni) * 32 + li) * KS + k];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          if (!use[mi][k]) continue;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi * NI + ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], bv[ni], acc[mi * NI + ni], 0, 0, 0);
        }
      }
    }
    }
    if constexpr (DB) {
      if (kt + 1 < nk) store_tile((kt & 1) ? 0 : TILE);
      __syncthreads();
    } else {
      __syncthreads();
      if (kt + 1 < nk) {
        store_tile(0);
        __syncthreads();
      }
    }
  }
  const float* __restrict__ xpre = pick3(ptrs.xpre, prob);
  const int xact = ptrs.xact;
  const bool vec = ((Cin & 3) == 0) && ((reinterpret_cast<uintptr_t>(dx) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(xpre) & 15) == 0);
  tile_epilogue<BM, BN, KG, WN, MI, NI>(acc, smem, kg, wm, wn, li, lh, tid, [&](int row_, int cc, float4 v) {
    const int tp = tp0 + tile_row(row_), c = c0 + cc;
    if (tp >= Tin || c >= Cin) return;
    const size_t off = ((size_t)b * Tin + tp) * Cin + c;
    float* o = dx + off;
    if (vec && c + 3 < Cin) {
      if (xpre) {
        const float4 p = ld4(xpre + off);
        v.x *= act_grad_f(p.x, xact); v.y *= act_grad_f(p.y, xact); v.z *= act_grad_f(p.z, xact); v.w *= act_grad_f(p.w, xact);
      }
      *reinterpret_cast<float4*>(o) = v;
    } else {
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c + j < Cin) o[j] = xpre ? e[j] * act_grad_f(xpre[off + j], xact) : e[j];
    }
  });
}

template <int KS, int S, int WM, int WN, int MI, int NI, int KC, int KG, bool DB = false, bool BF3 = false>
__global__ __launch_bounds__(64 * WM * WN * KG, ((KS == 1 && !BF3 && !DB && WM == 1 && WN == 4 && KG == 4) ? 8 : 1)) void conv_dgrad_kernel(
    DgradPtrs ptrs, int B, int Tin, int Cin, int Tout, int Cout, int act, int tilesT) {
  __shared__ __attribute__((aligned(16))) float smem[dgrad_smem_floats<KS, S, WM, WN, MI, NI, KC, KG, DB, BF3>()];
  conv_dgrad_body<KS, S, WM, WN, MI, NI, KC, KG, DB, BF3>(smem, ptrs, B, Tin, Cin, Tout, Cout, act, tilesT, blockIdx.x,
                                                          blockIdx.y, blockIdx.z);
}

// ------------------------------------------------------------------------------------------
// weight gradient (token dimension split over gridDim.z AND over the K-groups of a workgroup)
// ------------------------------------------------------------------------------------------
template <int KS, int S, int WM, int WN, int MI, int NI, int KT, int KG, bool BF3 = false>
constexpr int wgrad_smem_floats() {
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
  // BF3 (K = 1 only): hi + lo bf16 planes of the TRANSPOSED tiles g^T [BM][KT+8] and x^T [BN][KT+8]
  // k = 3: the epilogue parks WM*32 rows of the [n][(c, k)] result tile in LDS (row-contiguous gradient writes)
  return cmax(cmax(BF3 ? (BM + BN) * (KT + 8) : KT * BM + ((KT - 1) * S + KS) * BN, (KG - 1) * 4 * MI * NI * KS * 1024),
              KS > 1 ? WM * 32 * (BN * KS + 4) : 0);
}

// body of one wgrad workgroup; (bx, by, bz) = coordinates in the (mtiles*problems, c-tiles, token splits) grid, gz = splits
template <int KS, int S, int WM, int WN, int MI, int NI, int KT, int KG, bool ATOMIC, bool BF3 = false>
__device__ __forceinline__ void conv_wgrad_body(
    float* __restrict__ smem, const WgradPtrs& ptrs, int mtiles, int B, int Tin, int Cin, int Tout, int Cout, int act,
    int tilesPerClip, int tilesPerSplit, int accumulate, int bx, int by, int bz, int gz) {
  const int prob = bx / mtiles;
  const float* __restrict__ dy = pick3(ptrs.dy, prob);
  const float* __restrict__ x = pick3(ptrs.x, prob);
  const float* __restrict__ mask = pick3(ptrs.mask, prob);
  const float* __restrict__ ypre = pick3(ptrs.ypre, prob);
  float* __restrict__ out = pick3(ptrs.out, prob);
  float* __restrict__ dbias = pick3(ptrs.dbias, prob);
  static_assert(WM * WN == 4 || WM * WN == 1, "4 waves (or, for the 32x32 tile, 1 wave) per K-group");
  static_assert(KT % (2 * KG) == 0, "token tile must split evenly into pairs per group");
  constexpr int WPG = WM * WN, NT = 64 * WPG * KG;
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32, P = KS / 2;
  constexpr int RX = (KT - 1) * S + KS;
  constexpr int Y4 = BM / 4, X4 = BN / 4;
  // BF3: both tiles are transposed while staging ([column][token], token-contiguous hi / lo planes) so that the
  // contraction index t is the fast LDS axis of both MFMA operands
  static_assert(!BF3 || (KS == 1 && S == 1), "split-bf16 wgrad: K = 1 layers only");
  constexpr int TQ = KT / 4, LDT = KT + 8;
  // one index space over the 4x4 blocks of BOTH tiles (g blocks first), so that a 1024-thread workgroup stages one block
  // per thread instead of two on half of its threads
  constexpr int YB = TQ * Y4, XB = TQ * X4, NBQ = (YB + XB + NT - 1) / NT;
  static_assert(!BF3 || YB % 64 == 0, "the g / x split of the block index space is wave-uniform");
  constexpr int NY = BF3 ? 4 * NBQ : (KT * Y4 + NT - 1) / NT, NX = BF3 ? 1 : (RX * X4 + NT - 1) / NT;
  constexpr int NACC = MI * NI * KS;
  constexpr int YS = KT * BM, XS = RX * BN;
  constexpr int YSH = BM * LDT, XSH = BN * LDT;
  static_assert(wgrad_smem_floats<KS, S, WM, WN, MI, NI, KT, KG, BF3>() >=
                    cmax(BF3 ? (BM + BN) * LDT : YS + XS, (KG - 1) * 4 * NACC * 1024), "smem size");
  constexpr int KTG = KT / KG;
  static_assert(!BF3 || KTG % 16 == 0, "bf16x3 needs 16-deep token steps per K-group");
  float* Ys = smem;
  float* Xs = smem + YS;
  __bf16* Yh = reinterpret_cast<__bf16*>(smem);
  __bf16* Yl = Yh + YSH;
  __bf16* Xh = Yl + YSH;
  __bf16* Xl = Xh + XSH;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int kg = wid / WPG, w4 = wid % WPG;
  const int wm = w4 / WN, wn = w4 % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = (bx % mtiles) * BM, c0 = by * BN;
  const int gbeg = bz * tilesPerSplit;
  int gend = gbeg + tilesPerSplit;
  const int gtot = B * tilesPerClip;
  if (gend > gtot) gend = gtot;

  floatx16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  F4Buf<NY> ry;
  F4Buf<NX> rx;
  auto load_tile = [&](int g) {
    const int b = g / tilesPerClip, tt0 = (g % tilesPerClip) * KT;
    if constexpr (BF3) {
#pragma unroll
      for (int ii = 0; ii < NBQ; ++ii) {
        const int q = tid + ii * NT;
        int c4, tq;
        if (q < YB) {
          tq_coord<Y4, TQ>(q, c4, tq);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float4 v = zero4();
            const int t = tt0 + tq * 4 + j, n = n0 + c4 * 4;
            if (t < Tout && n < Cout) {
              const size_t row = (size_t)b * Tout + t;
              v = grad_prologue(dy, mask, ypre, row, row * Cout + n, act);
            }
            ry.v[ii * 4 + j] = v;
          }
        } else {
          tq_coord<X4, TQ>(q - YB, c4, tq);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float4 v = zero4();
            const int t = tt0 + tq * 4 + j, c = c0 + c4 * 4;
            if (q < YB + XB && t < Tin && c < Cin) v = ld4(x + ((size_t)b * Tin + t) * Cin + c);
            ry.v[ii * 4 + j] = v;
          }
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int q = tid + i * NT;
      float4 v = zero4();
      if (q < KT * Y4) {
        const int t = tt0 + q / Y4, n = n0 + (q % Y4) * 4;
        if (t < Tout && n < Cout) {
          const size_t row = (size_t)b * Tout + t;
          v = grad_prologue(dy, mask, ypre, row, row * Cout + n, act);
        }
      }
      ry.v[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int q = tid + i * NT;
      float4 v = zero4();
      if (q < RX * X4) {
        const int tr = tt0 * S - P + q / X4, c = c0 + (q % X4) * 4;
        if (tr >= 0 && tr < Tin && c < Cin) v = ld4(x + ((size_t)b * Tin + tr) * Cin + c);
      }
      rx.v[i] = v;
    }
  };
  auto store_tile = [&]() {
    if constexpr (BF3) {
#pragma unroll
      for (int ii = 0; ii < NBQ; ++ii) {
        const int q = tid + ii * NT;
        if (q >= YB + XB) continue;
        int c4, tq;
        if (q < YB) tq_coord<Y4, TQ>(q, c4, tq); else tq_coord<X4, TQ>(q - YB, c4, tq);
        __bf16* H = q < YB ? Yh : Xh;
        __bf16* L = q < YB ? Yl : Xl;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const float4 v = make_float4(f4_get(ry.v[ii * 4 + 0], jj), f4_get(ry.v[ii * 4 + 1], jj),
                                       f4_get(ry.v[ii * 4 + 2], jj), f4_get(ry.v[ii * 4 + 3], jj));
          uint2 hi, lo;
          split_bf16x4(v, hi, lo);
          *reinterpret_cast<uint2*>(H + (c4 * 4 + jj) * LDT + tq * 4) = hi;
          *reinterpret_cast<uint2*>(L + (c4 * 4 + jj) * LDT + tq * 4) = lo;
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int q = tid + i * NT;
      if (q < KT * Y4) *reinterpret_cast<float4*>(Ys + (q / Y4) * BM + (q % Y4) * 4) = ry.v[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int q = tid + i * NT;
      if (q < RX * X4) *reinterpret_cast<float4*>(Xs + (q / X4) * BN + (q % X4) * 4) = rx.v[i];
    }
  };

  // ATOMIC path: the c-tile-0 workgroups also fold the bias gradient (column sums of the g tile)
  const bool do_bias = ATOMIC && dbias != nullptr && by == 0 && tid < BM;
  float bsum = 0.f;

  if (gbeg < gend) {
    load_tile(gbeg);
    store_tile();
  }
  __syncthreads();
  for (int g = gbeg; g < gend; ++g) {
    if (g + 1 < gend) load_tile(g + 1);
    if (do_bias) {
      if constexpr (BF3) {
#pragma unroll 4
        for (int t = 0; t < KT; t += 8) {
          const bf16x8 h = *reinterpret_cast<const bf16x8*>(Yh + tid * LDT + t);
          const bf16x8 l = *reinterpret_cast<const bf16x8*>(Yl + tid * LDT + t);
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum += (float)h[e] + (float)l[e];
        }
      } else {
#pragma unroll 8
        for (int t = 0; t < KT; ++t) bsum += Ys[t * BM + tid];
      }
    }
    if constexpr (BF3) {
#pragma unroll
      for (int tl0 = 0; tl0 < KTG; tl0 += 16) {
        const int tl = kg * KTG + tl0 + lh * 8;
        bf16x8 ah[MI], al[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int off = ((wm * MI + mi) * 32 + li) * LDT + tl;
          ah[mi] = *reinterpret_cast<const bf16x8*>(Yh + off);
          al[mi] = *reinterpret_cast<const bf16x8*>(Yl + off);
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int off = ((wn * NI + ni) * 32 + li) * LDT + tl;
          const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Xh + off);
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Xl + off);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            floatx16 c_ = acc[mi * NI + ni];
            c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh, c_, 0, 0, 0);
            c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl, c_, 0, 0, 0);
            c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh, c_, 0, 0, 0);
            acc[mi * NI + ni] = c_;
          }
        }
      }
    } else {
#pragma unroll
    for (int tl0 = 0; tl0 < KTG; tl0 += 2) {
      const int tl = kg * KTG + tl0;
      float a[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = Ys[(tl + lh) * BM + (wm * MI + mi) * 32 + li];
#pragma unroll
      for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const float bv = Xs[((tl + lh) * S + k) * BN + (wn * NI + ni) * 32 + li];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[(mi * NI + ni) * KS + k] =
                __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], bv, acc[(mi * NI + ni) * KS + k], 0, 0, 0);
        }
    }
    }
    __syncthreads();
    if (g + 1 < gend) {
      store_tile();
      __syncthreads();
    }
  }
  if (do_bias && n0 + tid < Cout) atomicAdd(dbias + n0 + tid, bsum);
  kgroup_reduce<NACC, KG>(acc, smem, kg, w4, lane);
  float* o = out + ((!ATOMIC && gz > 1) ? (size_t)bz * Cout * Cin * KS : 0);
  const bool add = accumulate && gz == 1;
  if constexpr (KS > 1) {
    // k = 3: an accumulator register holds elements (n, c, k) that lie KS floats apart in dw[n][c][k], so adding them
    // straight from the registers is a strided wave-instruction -- a third of each 64-byte atomic request used, and
    // float atomics run at the memory side at ~1.3 TB/s only for CONTIGUOUS 256-byte wave-instructions
    // (MI355X_MICROARCH.md, Global float atomics).  The tile goes through LDS as rows [n][(c, k)] instead (a row of the
    // tile is BN*KS consecutive floats of dw) and every wave adds / stores 64 consecutive floats; a launch without a
    // token split owns its elements and uses a plain float4 read-add-store (stores run at ~6 TB/s).
    constexpr int ROWF = BN * KS, LDR = ROWF + 4, ROWS = WM * 32;
    static_assert(ROWS * LDR <= wgrad_smem_floats<KS, S, WM, WN, MI, NI, KT, KG, BF3>(), "epilogue tile fits");
    const size_t ldw = (size_t)Cin * KS;
    const int rowlen = min(ROWF, (Cin - c0) * KS);                  // valid floats of a tile row
#pragma unroll
    for (int p = 0; p < MI; ++p) {
      __syncthreads();                                               // LDS free: K-group reduction / previous pass read
      if (kg == 0) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int k = 0; k < KS; ++k)
              smem[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDR + ((wn * NI + ni) * 32 + li) * KS + k] =
                  acc[(p * NI + ni) * KS + k][r];
      }
      __syncthreads();
      if (ATOMIC && gz > 1) {
        for (int q = tid; q < ROWS * ROWF; q += NT) {
          const int rr = q / ROWF, e = q % ROWF;
          const int n = n0 + ((rr >> 5) * MI + p) * 32 + (rr & 31);
          if (n < Cout && e < rowlen) atomicAdd(o + (size_t)n * ldw + (size_t)c0 * KS + e, smem[rr * LDR + e]);
        }
      } else {
        for (int q = tid; q < ROWS * (ROWF / 4); q += NT) {
          const int rr = q / (ROWF / 4), e = (q % (ROWF / 4)) * 4;
          const int n = n0 + ((rr >> 5) * MI + p) * 32 + (rr & 31);
          if (n >= Cout || e >= rowlen) continue;                    // Cin % 4 == 0: a float4 is all valid or all not
          float4 v = *reinterpret_cast<const float4*>(smem + rr * LDR + e);
          float4* d = reinterpret_cast<float4*>(o + (size_t)n * ldw + (size_t)c0 * KS + e);
          if (ATOMIC || add) { const float4 g = *d; v.x += g.x; v.y += g.y; v.z += g.z; v.w += g.w; }
          *d = v;
        }
      }
    }
    return;
  }
  if (kg != 0) return;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int c = c0 + (wn * NI + ni) * 32 + li;
      if (c >= Cin) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n >= Cout) continue;
        float* d = o + ((size_t)n * Cin + c) * KS;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          if (ATOMIC) atomicAdd(d + k, acc[(mi * NI + ni) * KS + k][r]);
          else if (add) d[k] += acc[(mi * NI + ni) * KS + k][r];
          else d[k] = acc[(mi * NI + ni) * KS + k][r];
        }
      }
    }
}

template <int KS, int S, int WM, int WN, int MI, int NI, int KT, int KG, bool ATOMIC, bool BF3 = false>
__global__ __launch_bounds__(256 * KG) void conv_wgrad_kernel(
    WgradPtrs ptrs, int mtiles, int B, int Tin, int Cin, int Tout, int Cout, int act, int tilesPerClip,
    int tilesPerSplit, int accumulate) {
  __shared__ __attribute__((aligned(16))) float smem[wgrad_smem_floats<KS, S, WM, WN, MI, NI, KT, KG, BF3>()];
  conv_wgrad_body<KS, S, WM, WN, MI, NI, KT, KG, ATOMIC, BF3>(smem, ptrs, mtiles, B, Tin, Cin, Tout, Cout, act,
                                                              tilesPerClip, tilesPerSplit, accumulate, blockIdx.x,
                                                              blockIdx.y, blockIdx.z, gridDim.z);
}

// K=1 data gradient AND weight gradient of the same layer(s) in ONE launch: workgroups [0, nW) run the wgrad body
// (token-split, atomics into dw / dbias), workgroups [nW, nW + nD) the dgrad body.  Both read the same dy; issued as
// one grid they fill the chip together instead of leaving two under-filled launches with a drain between them.
// (fp32, 1024 threads: the second launch bound keeps two workgroups per CU -- 64 VGPRs -- with the float4 g fragments)
template <int WM, int WN, int MI, int NI, int KC, int KGD, bool BF3 = false>
__global__ __launch_bounds__(64 * WM * WN * KGD, ((!BF3 && WM == 1 && WN == 4 && KGD == 4) ? 8 : 1)) void conv_bwd_pair_k1_kernel(
    DgradPtrs dp, WgradPtrs wp, int nW, int wgx, int wgy, int wgz, int dgx, int dgy, int mtiles, int B, int T, int Cin,
    int Cout, int act, int tilesT, int tilesPerClip, int tilesPerSplit) {
  constexpr int KG = WM * WN * KGD / 4;          // wgrad K-groups: same workgroup size (4 waves per wgrad K-group)
  static_assert(KG == 1 || KG == 2 || KG == 4, "pair: workgroup of 256 / 512 / 1024 threads");
  constexpr int SM = cmax(dgrad_smem_floats<1, 1, WM, WN, MI, NI, KC, KGD, false, BF3>(),
                          wgrad_smem_floats<1, 1, 2, 2, 1, 1, 128, KG, BF3>());
  __shared__ __attribute__((aligned(16))) float smem[SM];
  // XCD-aware work order (round 4).  Workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8, each
  // with its own 4 MiB L2): in grid order the wgrad workgroups that share an x token chunk (same token split, different
  // n tile) sat on different XCDs, so every XCD pulled the whole x (and every dy column slab) through the fabric --
  // 63.6 MB fetched per launch against ~25 MB algorithmic (profiles/r03_l_pmc_traffic.txt).  Here XCD c gets the c-th
  // eighth of the wgrad items in (token split, c tile, n tile) order followed by the c-th eighth of the dgrad items in
  // (row tile, column tile) order: the token range of an XCD's weight-gradient splits is the row range of its
  // data-gradient tiles, so dy and x rows cross the fabric once and are shared through that XCD's L2.  Placement is a
  // speed matter only (MI355X_MICROARCH.md: no dispatch-order guarantee): any block -> XCD map gives the same results.
  const int G = gridDim.x, hw = blockIdx.x;
  const int xcd = hw & 7;
  // position in XCD-major order (tilesT < 0: grid order, the A/B switch AVDF_PAIR_XCD=0)
  const bool grid_order = tilesT < 0;
  const int v = grid_order ? hw : xcd * (G >> 3) + min(xcd, G & 7) + (hw >> 3);
  if (grid_order) tilesT = -tilesT;
  int c = 0;
  if (!grid_order)
    while (c < 7 && v >= (int)(((long)(c + 1) * nW) >> 3) + (int)(((long)(c + 1) * (G - nW)) >> 3)) ++c;
  // grid order: one chunk = the whole grid (wgrad items first, then dgrad items)
  const int w0 = grid_order ? 0 : (int)(((long)c * nW) >> 3), w1 = grid_order ? nW : (int)(((long)(c + 1) * nW) >> 3),
            d0 = grid_order ? 0 : (int)(((long)c * (G - nW)) >> 3);
  const int local = v - (w0 + d0);
  if (local < w1 - w0) {
    const int id = w0 + local;
    const int bx = id % wgx, by = (id / wgx) % wgy, bz = id / (wgx * wgy);
    conv_wgrad_body<1, 1, 2, 2, 1, 1, 128, KG, true, BF3>(smem, wp, mtiles, B, T, Cin, T, Cout, act, tilesPerClip,
                                                          tilesPerSplit, 1, bx, by, bz, wgz);
  } else {
    const int id = d0 + local - (w1 - w0);
    const int nprob = (G - nW) / (dgx * dgy);                              // row tile slowest, then problem, then column tile
    int by = id % dgy, bz = (id / dgy) % nprob, bx = id / (dgy * nprob);
    if (grid_order) { bx = id % dgx; by = (id / dgx) % dgy; bz = id / (dgx * dgy); }      // the round-3 order
    conv_dgrad_body<1, 1, WM, WN, MI, NI, KC, KGD, false, BF3>(smem, dp, B, T, Cin, T, Cout, act, tilesT, bx, by, bz);
  }
}

// column sums of g = dy*mask*act'(pre): partial[blockIdx.x][n]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dy, const float* __restrict__ mask,
                                                     const float* __restrict__ ypre, float* __restrict__ partial,
                                                     int M, int N, int act) {
  __shared__ float red[4][256];
  const int tid = threadIdx.x, cg = tid & 63, rg = tid >> 6;
  for (int nb = 0; nb < N; nb += 256) {
    const int n = nb + cg * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N) {
      for (int r = blockIdx.x * 4 + rg; r < M; r += gridDim.x * 4) {
        const float4 g = grad_prologue(dy, mask, ypre, (size_t)r, (size_t)r * N + n, act);
        s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
      }
    }
    red[rg][cg * 4 + 0] = s.x; red[rg][cg * 4 + 1] = s.y; red[rg][cg * 4 + 2] = s.z; red[rg][cg * 4 + 3] = s.w;
    __syncthreads();
    if (nb + tid < N) partial[(size_t)blockIdx.x * N + nb + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    __syncthreads();
  }
}

// out[i] (+)= sum_s ws[s*n + i].  Columns >= split go to out2[i - split] (two results from one
// slab set, e.g. LayerNorm dgamma|dbeta).  Small n: 64 columns x 16 slab phases per 1024-thread block
// (many independent loads in flight instead of one serial chain per column).
__global__ __launch_bounds__(1024) void reduce_slabs_small_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                                 float* __restrict__ out2, int nslab, int n, int split,
                                                                 int accumulate) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  float s = 0.f;
  if (col < n) {
#pragma unroll 4
    for (int k = ph; k < nslab; k += 16) s += ws[(size_t)k * n + col];
  }
  red[ph][cl] = s;
  __syncthreads();
  if (ph == 0 && col < n) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][cl];
    float* o = col < split ? out + col : out2 + (col - split);
    *o = accumulate ? *o + t : t;
  }
}

// large n (multiple of 4): one float4 column group per thread, streaming over the slabs
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                           int nslab, int64_t n, int accumulate) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int k = 0; k < nslab; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(ws + (size_t)k * n + i * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float4* o = reinterpret_cast<float4*>(out + i * 4);
    if (accumulate) { const float4 c = *o; s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w; }
    *o = s;
  }
}

constexpr int kColsumBlocks = 128;

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Output-tile and K-group choice for the forward / dgrad GEMMs: the largest tile that still yields
// >= 2 workgroups per CU wins, otherwise the shape with the most workgroups; then as many K-groups
// as needed to put ~1024+ wave-quads on the chip.
// tile 0: 128x128  1: 64x128  2: 64x64  3: 32x128  4: 256x64  5: 32x32 (K=1; table / tuner only)
// tile 6: 64x32, two waves per K-group (table / tuner only): a [6144, 256] output is 768 such tiles = exactly 3 per CU,
// all co-resident at 49.5 KB of LDS, where the 64x64 tile leaves half the CUs with one workgroup and half with two
struct TilePlan { int tile, kg, db; };

// ---- measured plans ----------------------------------------------------------------------------------------
// tools/autotune.py times every (tile, K-groups) / (token-splits, K-groups) choice for the GEMM shapes of a training
// step on the GPU and writes gemm_tuned.inc; shapes not in the table use the heuristics below.
struct TunedPlan { int kind, K, stride, M, N, Kred, a, b; };   // kind 0 fwd / 1 dgrad / 3 fwd, 6 dgrad in bf16x3 mode: (tile, kg + 8*db); 2 wgrad: (splits, kg); 4 pair / 5 pair in bf16x3 mode: (tile, kg | splits << 3)
static const TunedPlan kTuned[] = {
#include "gemm_tuned.inc"
    {-1, 0, 0, 0, 0, 0, 0, 0}};
// in-situ plan experiments without a rebuild (tools/plan_ab.py): AVDF_TUNED_EXTRA="kind,K,stride,M,N,Kred,a,b;..." rows are
// looked up before the table (a = -1: pretend the table has no row for the shape -> heuristic plan)
static TunedPlan g_extra[64];
static int parse_extra(const char* e) {
  int n = 0;
  while (e && *e && n < 64) {
    TunedPlan t;
    int used = 0;
    if (sscanf(e, "%d,%d,%d,%d,%d,%d,%d,%d%n", &t.kind, &t.K, &t.stride, &t.M, &t.N, &t.Kred, &t.a, &t.b, &used) != 8) break;
    g_extra[n++] = t;
    e += used;
    if (*e == ';') ++e;
  }
  return n;
}
static int g_nextra = parse_extra(getenv("AVDF_TUNED_EXTRA"));
inline const TunedPlan* tuned_lookup(int kind, int K, int stride, int M, int N, int Kred) {
  for (int i = 0; i < g_nextra; ++i) {
    const TunedPlan* t = &g_extra[i];
    if (t->kind == kind && t->K == K && t->stride == stride && t->M == M && t->N == N && t->Kred == Kred) return t->a < 0 ? nullptr : t;
  }
  for (const TunedPlan* t = kTuned; t->kind >= 0; ++t)
    if (t->kind == kind && t->K == K && t->stride == stride && t->M == M && t->N == N && t->Kred == Kred) return t;
  return nullptr;
}
// GEMM arithmetic mode, a bit mask: 0 = exact fp32 MFMA (default, what the parity goldens are held against); bit 0 =
// split-bf16 ("bf16x3") operands in the forward kernels; bit 1 = split-bf16 operands in the backward kernels (data
// gradient of every conv, weight gradient of the K = 1 layers) -- both opt-in (avdf_gemm_mode)
// thread-local: the mode is a launch-time host switch, so a thread that runs inference in bf16x3 cannot change the
// arithmetic of a training thread that shares the library
static thread_local int g_gemm_mode = [] { const char* e = getenv("AVDF_GEMM_MODE"); return e ? atoi(e) : 0; }();
// process-wide override used by the tuner: g_override[kind] = {a, b}, -1 = off
static int g_override[5][2] = {{-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}};   // [3] unused, [4] = backward pair

inline TilePlan pick_tile(int B, int T, int N, bool dgrad = false, int K = 0, int stride = 0, int Kred = 0) {
  {
    const int kind = dgrad ? 1 : 0;
    const int mk_[7] = {1, 2, 4, 4, 2, 4, 4};
    if (g_override[kind][0] >= 0) {
      TilePlan o{g_override[kind][0], g_override[kind][1] & 7, (g_override[kind][1] >> 3) & 1};
#ifdef AVDF_PW_EXPERIMENTAL
      if (o.tile >= 8 && kind == 0) return o;            // experimental pointwise forward core: (8 + cfg, workgroups per CU)
#endif
      if (o.tile > 6) o.tile = 6;
      if (o.kg > mk_[o.tile]) o.kg = mk_[o.tile];
      if (o.kg != 1 && o.kg != 2 && o.kg != 4) o.kg = 1;
      return o;
    }
    if (K > 0 && g_override[kind][0] != -2) {          // -2: heuristic only (the tuner's baseline)
      const TunedPlan* t = nullptr;
      if (kind == 0 && (g_gemm_mode & 1)) t = tuned_lookup(3, K, stride, B * T, N, Kred);   // kind 3: forward in bf16x3 mode
      if (kind == 1 && (g_gemm_mode & 2)) t = tuned_lookup(6, K, stride, B * T, N, Kred);   // kind 6: dgrad in bf16x3 mode
      if (!t) t = tuned_lookup(kind, K, stride, B * T, N, Kred);
      if (t) return TilePlan{t->a, t->b & 7, (t->b >> 3) & 1};
    }
  }
  const int bm[4] = {128, 64, 64, 32}, bn[4] = {128, 128, 64, 128};
  TilePlan p{0, 1, 0};
  long best_blocks = -1;
  for (int c = 0; c < 4; ++c) {
    const long blocks = (long)B * cdiv(T, bm[c]) * cdiv(N, bn[c]);
    if (blocks >= 512) { p.tile = c; best_blocks = blocks; break; }
    if (blocks > best_blocks) { best_blocks = blocks; p.tile = c; }
  }
  // dgrad's workgroups are not co-resident (84 VGPRs at 1024 threads): measured on the exp10 shapes the
  // 32x128 tile beats 64x64 there by 10-15% (tools/gemm_bench.py sweep, profiles/r01_c)
  if (dgrad && p.tile == 2) { p.tile = 3; best_blocks = (long)B * cdiv(T, 32) * cdiv(N, 128); }
  const int maxkg = (p.tile == 0) ? 1 : (p.tile == 1 ? 2 : 4);
  p.kg = 1;
  while (p.kg < maxkg && best_blocks * p.kg < 1024) p.kg *= 2;
  // tuning overrides for tools/gemm_bench.py (never set in production)
  static const char* ft = getenv("AVDF_FORCE_TILE");
  static const char* fk = getenv("AVDF_FORCE_KG");
  if (ft) p.tile = atoi(ft);
  if (fk) p.kg = atoi(fk);
  const int mk = (p.tile == 0) ? 1 : (p.tile == 1 ? 2 : 4);
  if (p.kg > mk) p.kg = mk;
  if (p.kg == 3) p.kg = 2;
  return p;
}

struct WgradPlan { int tilesPerClip, splits, tilesPerSplit, mtiles, ctiles, kg; bool small; };
inline WgradPlan wgrad_plan(int B, int Tout, int Cin, int Cout, int K, int nprob = 1, int stride_for_table = 0) {
  WgradPlan p;
  p.small = ((int64_t)Cout * Cin * K <= 512 * 1024);
  const int KT = p.small ? 128 : 64;           // tokens per K-tile (must match the launch below)
  const int bm = p.small ? 64 : 128;
  p.mtiles = cdiv(Cout, bm);
  p.ctiles = cdiv(Cin, 64);
  p.tilesPerClip = cdiv(Tout, KT);
  const int total = B * p.tilesPerClip;
  const int tiles = p.mtiles * p.ctiles * nprob;
  p.kg = p.small ? (tiles >= 256 ? 1 : (tiles >= 64 ? 2 : 4)) : (tiles >= 256 ? 1 : 2);
  if (K == 3 && p.kg > 2) p.kg = 2;            // 3 accumulator sets per wave: keep the LDS reduction small
  // token splits: every split re-writes the whole weight gradient (atomics or a slab), so the k=3 / big-weight
  // problems want few of them (autotune: 1-4 splits beat the old "1024 workgroups" rule by 1.3-2.3x there)
  int want = cdiv(K == 3 ? 256 : 1024, tiles * p.kg);
  {
    int a = -1, b = -1;
    if (g_override[2][0] >= 0) { a = g_override[2][0]; b = g_override[2][1]; }
    else if (stride_for_table > 0 && g_override[2][0] != -2) {
      const TunedPlan* t = tuned_lookup(2, K, stride_for_table, B * Tout, Cout * nprob, Cin);
      if (t) { a = t->a; b = t->b; }
    }
    if (a > 0) {
      want = a;
      const int maxkg = (K == 3) ? 2 : (p.small ? 4 : 2);
      if (b == 1 || b == 2 || b == 4) p.kg = b > maxkg ? maxkg : b;
    }
  }
  if (want < 1) want = 1;
  if (want > total) want = total;
  p.tilesPerSplit = cdiv(total, want);
  p.splits = cdiv(total, p.tilesPerSplit);
  return p;
}

}  // namespace

int avdf_reduce_slabs2(const float* ws, float* out, float* out2, int nslab, int64_t n, int64_t split,
                       int accumulate, hipStream_t st) {
  const bool vec_ok = (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(ws) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && out2 == nullptr;
  if (n >= 32768 && vec_ok) {
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, st, ws, out, nslab, n, accumulate);
  } else {
    hipLaunchKernelGGL(reduce_slabs_small_kernel, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, st, ws, out,
                       out2 ? out2 : out, nslab, (int)n, (int)(out2 ? split : n), accumulate);
  }
  AVDF_CHECK_LAUNCH("reduce_slabs");
  return AVDF_OK;
}

int avdf_reduce_slabs(const float* ws, float* out, int nslab, int64_t n, int accumulate, hipStream_t st) {
  return avdf_reduce_slabs2(ws, out, nullptr, nslab, n, n, accumulate, st);
}

// K-tile depth per tile shape: deep enough that one tile of MFMAs (>= 2-4k cycles per group) covers
// the global-load latency of the next one at ~1 wave per SIMD.
// tile 4 = 256x64 (4x1 waves of 64x64, the conv3d shape: 4 MFMAs per 4 LDS reads, long stages); only reached
// through the tuned table / the tuner override
// tile 5 = 32x32 output tile, ONE wave per K-group and 4 / 8 / 16 K-groups (plan.kg = 1 / 2 / 4 codes K-groups / 4):
// 4x the workgroups of the 64x64 tile for the layers whose tile count is below the CU count (K=1 only; for the k=3
// convs W5_ = 2 aliases tile 2)
#define AVDF_GEMM_DISPATCH(KS_, S_, KC0_, KC1_, KC2_, KC3_, KC4_, KC5_, W5_, L)    \
  switch (plan.tile * 8 + plan.kg) {                                               \
    case 6 * 8 + 1: L(KS_, S_, KC2_, 2, 1, 1, 1, 1); break;                        \
    case 6 * 8 + 2: L(KS_, S_, KC2_, 2, 1, 1, 1, 2); break;                        \
    case 6 * 8 + 4: L(KS_, S_, KC2_, 2, 1, 1, 1, 4); break;                        \
    case 5 * 8 + 1: L(KS_, S_, KC5_, W5_, W5_, 1, 1, (W5_ == 1 ? 4 : 1)); break;   \
    case 5 * 8 + 2: L(KS_, S_, KC5_, W5_, W5_, 1, 1, (W5_ == 1 ? 8 : 2)); break;   \
    case 5 * 8 + 4: L(KS_, S_, KC5_, W5_, W5_, 1, 1, (W5_ == 1 ? 16 : 4)); break;  \
    case 4 * 8 + 1: L(KS_, S_, KC4_, 4, 1, 2, 2, 1); break;                        \
    case 4 * 8 + 2: L(KS_, S_, KC4_, 4, 1, 2, 2, 2); break;                        \
    case 0 * 8 + 1: L(KS_, S_, KC0_, 2, 2, 2, 2, 1); break;                        \
    case 1 * 8 + 1: L(KS_, S_, KC1_, 2, 2, 1, 2, 1); break;                        \
    case 1 * 8 + 2: L(KS_, S_, KC1_, 2, 2, 1, 2, 2); break;                        \
    case 2 * 8 + 1: L(KS_, S_, KC2_, 2, 2, 1, 1, 1); break;                        \
    case 2 * 8 + 2: L(KS_, S_, KC2_, 2, 2, 1, 1, 2); break;                        \
    case 2 * 8 + 4: L(KS_, S_, KC2_, 2, 2, 1, 1, 4); break;                        \
    case 3 * 8 + 1: L(KS_, S_, KC3_, 1, 4, 1, 1, 1); break;                        \
    case 3 * 8 + 2: L(KS_, S_, KC3_, 1, 4, 1, 1, 2); break;                        \
    default: L(KS_, S_, KC3_, 1, 4, 1, 1, 4); break;                               \
  }

static int conv1d_fwd_launch(const FwdPtrs& ptrs, int nprob, int B, int Tin, int Cin, int Tout, int Cout, int K,
                             int stride, int pad, int ldx, int ldw, int ldy, int act, hipStream_t st, long xbs = -1,
                             long wbs = 0, long ybs = -1) {
  // K=1 is a plain GEMM over the B*T token rows: drop the clip structure (no halo between clips) so that short
  // pyramid levels (T = 12..96) are not padded to one M-tile per clip
  if (K == 1 && stride == 1 && pad == 0 && xbs < 0 && ybs < 0 && wbs == 0) { Tin = Tout = B * Tin; B = 1; }
  if (xbs < 0) xbs = (long)Tin * ldx;
  if (ybs < 0) ybs = (long)Tout * ldy;
  TilePlan plan = pick_tile(B * nprob, Tout, Cout, false, K, stride, Cin);
#ifdef AVDF_PW_EXPERIMENTAL
  if (plan.tile >= 8) {
    // tile codes >= 8 select the pipelined pointwise core: plain row-major GEMM over the token rows, exact fp32 only
    const bool ok = K == 1 && stride == 1 && pad == 0 && B == 1 && wbs == 0 && ldx == Cin && ldw == Cin && ldy == Cout &&
                    Cout % 4 == 0 && ptrs.bbs == 0 && !(g_gemm_mode & 1) && pw_fwd_supported(plan.tile - 8, Cin) &&
                    aligned16(ptrs.x[0]) && aligned16(ptrs.x[1]) && aligned16(ptrs.x[2]) && aligned16(ptrs.w[0]) &&
                    aligned16(ptrs.w[1]) && aligned16(ptrs.w[2]) &&
                    aligned16(ptrs.y[0]) && aligned16(ptrs.ypre[0]) && aligned16(ptrs.bias[0]) && aligned16(ptrs.res) &&
                    aligned16(ptrs.yres) && aligned16(ptrs.rscale) && aligned16(ptrs.y[1]) && aligned16(ptrs.y[2]) &&
                    aligned16(ptrs.bias[1]) && aligned16(ptrs.bias[2]);
    if (ok) {
      PwFwd a;
      for (int i = 0; i < 3; ++i) {
        a.x[i] = ptrs.x[i]; a.w[i] = ptrs.w[i]; a.bias[i] = ptrs.bias[i]; a.mask[i] = ptrs.mask[i];
        a.y[i] = ptrs.y[i]; a.ypre[i] = ptrs.ypre[i];
      }
      a.res = ptrs.res; a.rscale = ptrs.rscale; a.rkeep = ptrs.rkeep; a.yres = ptrs.yres; a.resT = ptrs.resT;
      a.nprob = nprob; a.M = Tout; a.N = Cout; a.K = Cin; a.act = act;
      return pw_fwd_launch(a, plan.tile - 8, plan.kg, st);
    }
    plan = TilePlan{2, 1, 0};
  }
#else
  if (plan.tile > 6) plan = TilePlan{2, 1, 0};         // table rows of the experimental core (codes >= 8) in a product build
#endif
#define LAUNCH_FWD(KS_, S_, KC_, WM_, WN_, MI_, NI_, KG_)                                               \
  do {                                                                                                  \
    constexpr int BM_ = WM_ * MI_ * 32, BN_ = WN_ * NI_ * 32;                                           \
    const int tilesT = cdiv(Tout, BM_);                                                                 \
    dim3 grid(B * tilesT, cdiv(Cout, BN_), nprob);                                                      \
    constexpr int RA_ = (BM_ - 1) * S_ + KS_;                                                           \
    constexpr bool DBOK_ = 2 * (RA_ * (KC_ + 4) + BN_ * (KC_ * KS_ + 4) + 8) * 4 <= 150 * 1024;         \
    if constexpr ((KC_ / KG_) % 16 == 0) {                                                              \
      if (g_gemm_mode & 1) {                                                                            \
        hipLaunchKernelGGL((conv_fwd_kernel<KS_, S_, WM_, WN_, MI_, NI_, KC_, KG_, true>), grid,        \
                           dim3(64 * WM_ * WN_ * KG_), 0, st, ptrs, B, Tin, Cin, Tout, Cout, act, tilesT, pad,     \
                           ldx, ldw, ldy, xbs, wbs, ybs);                                               \
        break;                                                                                          \
      }                                                                                                 \
    }                                                                                                   \
    if constexpr (DBOK_) {                                                                              \
      if (plan.db) {                                                                                    \
        hipLaunchKernelGGL((conv_fwd_kernel<KS_, S_, WM_, WN_, MI_, NI_, KC_, KG_, false, true>), grid, \
                           dim3(64 * WM_ * WN_ * KG_), 0, st, ptrs, B, Tin, Cin, Tout, Cout, act, tilesT, pad,     \
                           ldx, ldw, ldy, xbs, wbs, ybs);                                               \
        break;                                                                                          \
      }                                                                                                 \
    }                                                                                                   \
    hipLaunchKernelGGL((conv_fwd_kernel<KS_, S_, WM_, WN_, MI_, NI_, KC_, KG_>), grid, dim3(64 * WM_ * WN_ * KG_), \
                       0, st, ptrs, B, Tin, Cin, Tout, Cout, act, tilesT, pad, ldx, ldw, ldy, xbs, wbs, \
                       ybs);                                                                            \
  } while (0)
  if (K == 19 && stride == 1) {          // Emotion2Vec positional grouped conv: 64x64 tile only (the 19-tap B tile is 78 KB)
    // split-bf16 needs 16-deep k-steps per K-group: one K-group in that mode
    if (plan.kg >= 2 && !(g_gemm_mode & 1)) LAUNCH_FWD(19, 1, 16, 2, 2, 1, 1, 2); else LAUNCH_FWD(19, 1, 16, 2, 2, 1, 1, 1);
  }
  else if (K == 3 && stride == 1) { AVDF_GEMM_DISPATCH(3, 1, 32, 32, 64, 32, 16, 64, 2, LAUNCH_FWD) }
  else if (K == 3 && stride == 2) { AVDF_GEMM_DISPATCH(3, 2, 32, 32, 64, 32, 16, 64, 2, LAUNCH_FWD) }
  else if (K == 2 && stride == 2) { AVDF_GEMM_DISPATCH(2, 2, 64, 64, 64, 64, 32, 64, 2, LAUNCH_FWD) }
  else { AVDF_GEMM_DISPATCH(1, 1, 64, 64, 128, 64, 32, 128, 1, LAUNCH_FWD) }
#undef LAUNCH_FWD
  AVDF_CHECK_LAUNCH("conv1d_fwd");
  return AVDF_OK;
}

static FwdPtrs fwd_ptrs1(const float* x, const float* w, const float* bias, const float* mask, float* y, float* ypre) {
  FwdPtrs g{};
  for (int i = 0; i < 3; ++i) { g.x[i] = x; g.w[i] = w; g.bias[i] = bias; g.mask[i] = mask; g.y[i] = y; g.ypre[i] = ypre; }
  return g;
}

extern "C" int avdf_conv1d_fwd(const float* x, const float* w, const float* bias, const float* mask,
                               float* y, float* ypre, int B, int Tin, int Cin, int Tout, int Cout,
                               int K, int stride, int act, void* stream) {
  AVDF_REQUIRE(x && w && y, "conv1d_fwd: null pointer");
  AVDF_REQUIRE(B > 0 && Tin > 0 && Cin > 0 && Tout > 0 && Cout > 0, "conv1d_fwd: bad dims");
  AVDF_REQUIRE((K == 1 || K == 3) && (stride == 1 || stride == 2), "conv1d_fwd: K in {1,3}, stride in {1,2}");
  AVDF_REQUIRE(!(K == 1 && stride != 1), "conv1d_fwd: K=1 needs stride 1");
  AVDF_REQUIRE(Cin % 4 == 0, "conv1d_fwd: Cin %% 4 != 0 (Cin=%d)", Cin);
  AVDF_REQUIRE(Tout == (Tin + 2 * (K / 2) - K) / stride + 1, "conv1d_fwd: Tout=%d inconsistent with Tin=%d K=%d s=%d", Tout, Tin, K, stride);
  AVDF_REQUIRE(aligned16(x) && aligned16(w), "conv1d_fwd: x/w must be 16-byte aligned");
  return conv1d_fwd_launch(fwd_ptrs1(x, w, bias, mask, y, ypre), 1, B, Tin, Cin, Tout, Cout, K, stride, K / 2, Cin,
                           Cin * K, Cout, act, (hipStream_t)stream);
}

// Dense conv + the block's second residual in one launch: h = (conv(x) + bias) * mask (stored to `y` unless y is NULL),
// out = res + rscale[c] * rkeep[b] * h  (AffineDropPath + residual, blocks.py:1313 `out + drop_path_mlp(mlp(ln2(out)))`).
// rkeep may be NULL (eval / no stochastic depth).  out must not alias x.
extern "C" int avdf_conv1d_fwd_residual(const float* x, const float* w, const float* bias, const float* mask,
                                        const float* res, const float* rscale, const float* rkeep, float* y, float* out,
                                        int B, int Tin, int Cin, int Tout, int Cout, int K, int stride, void* stream) {
  AVDF_REQUIRE(x && w && res && rscale && out, "conv1d_fwd_residual: null pointer");
  AVDF_REQUIRE(B > 0 && Tin > 0 && Cin > 0 && Tout > 0 && Cout > 0, "conv1d_fwd_residual: bad dims");
  AVDF_REQUIRE((K == 1 || K == 3) && (stride == 1 || stride == 2) && !(K == 1 && stride != 1), "conv1d_fwd_residual: unsupported K/stride");
  AVDF_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, "conv1d_fwd_residual: Cin, Cout must be multiples of 4 (%d, %d)", Cin, Cout);
  AVDF_REQUIRE(Tout == (Tin + 2 * (K / 2) - K) / stride + 1, "conv1d_fwd_residual: Tout inconsistent");
  AVDF_REQUIRE(aligned16(x) && aligned16(w) && aligned16(res) && aligned16(rscale) && aligned16(out) && aligned16(y),
               "conv1d_fwd_residual: 16-byte alignment");
  FwdPtrs g = fwd_ptrs1(x, w, bias, mask, y, nullptr);
  g.res = res; g.rscale = rscale; g.rkeep = rkeep; g.yres = out; g.resT = Tout;
  return conv1d_fwd_launch(g, 1, B, Tin, Cin, Tout, Cout, K, stride, K / 2, Cin, Cin * K, Cout, AVDF_ACT_NONE,
                           (hipStream_t)stream);
}

// Grouped "same" conv over time on the MFMA GEMM: x, y [T, C] time-major, w [C][C/groups][K] (torch Conv1d(groups=g) layout),
// bias [C]; every channel group is one batch entry of the strided forward kernel (column slices of x / y, its own weight
// and bias block).  K = 19 only: the positional conv of Emotion2Vec / data2vec-2.0 (fairseq `make_conv_pos`, k=19, g=16).
extern "C" int avdf_grouped_conv1d_fwd(const float* x, const float* w, const float* bias, float* y, int T, int C, int groups,
                                       int K, void* stream) {
  AVDF_REQUIRE(x && w && y && T > 0 && C > 0 && groups > 0 && C % groups == 0, "grouped_conv1d_fwd: bad args");
  AVDF_REQUIRE(K == 19, "grouped_conv1d_fwd: K must be 19 (K=%d)", K);
  const int Cg = C / groups;
  AVDF_REQUIRE(Cg % 4 == 0 && C % 4 == 0 && aligned16(x) && aligned16(w) && aligned16(y) && aligned16(bias),
               "grouped_conv1d_fwd: channels per group must be a multiple of 4, 16-byte aligned operands");
  FwdPtrs g = fwd_ptrs1(x, w, bias, nullptr, y, nullptr);
  g.bbs = Cg;
  return conv1d_fwd_launch(g, 1, groups, T, Cg, T, Cg, K, 1, K / 2, C, Cg * K, C, AVDF_ACT_NONE, (hipStream_t)stream,
                           (long)Cg, (long)Cg * Cg * K, (long)Cg);
}

// Same-shape problems in one launch (n <= 3): q / k / v projections.  Tables of n device pointers; bias / mask /
// ypre tables may be NULL (= absent for every problem) or hold NULL entries.
extern "C" int avdf_conv1d_fwd_grouped(int n, const float* const* x, const float* const* w, const float* const* bias,
                                       const float* const* mask, float* const* y, float* const* ypre, int B, int Tin,
                                       int Cin, int Tout, int Cout, int K, int stride, int act, void* stream) {
  AVDF_REQUIRE(n >= 1 && n <= 3 && x && w && y, "conv1d_fwd_grouped: 1..3 problems, non-null tables");
  AVDF_REQUIRE(B > 0 && Tin > 0 && Cin > 0 && Tout > 0 && Cout > 0, "conv1d_fwd_grouped: bad dims");
  AVDF_REQUIRE((K == 1 || K == 3) && (stride == 1 || stride == 2) && !(K == 1 && stride != 1), "conv1d_fwd_grouped: unsupported K/stride");
  AVDF_REQUIRE(Cin % 4 == 0, "conv1d_fwd_grouped: Cin %% 4 != 0 (Cin=%d)", Cin);
  AVDF_REQUIRE(Tout == (Tin + 2 * (K / 2) - K) / stride + 1, "conv1d_fwd_grouped: Tout inconsistent");
  FwdPtrs g;
  for (int i = 0; i < 3; ++i) {
    const int j = i < n ? i : 0;
    AVDF_REQUIRE(x[j] && w[j] && y[j] && aligned16(x[j]) && aligned16(w[j]), "conv1d_fwd_grouped: null / misaligned problem %d", j);
    g.x[i] = x[j]; g.w[i] = w[j]; g.y[i] = y[j];
    g.bias[i] = bias ? bias[j] : nullptr; g.mask[i] = mask ? mask[j] : nullptr; g.ypre[i] = ypre ? ypre[j] : nullptr;
  }
  return conv1d_fwd_launch(g, n, B, Tin, Cin, Tout, Cout, K, stride, K / 2, Cin, Cin * K, Cout, act, (hipStream_t)stream);
}

// General forward form used by the frozen extractors: explicit zero-padding `pad` (0 = "valid" conv of the
// wav2vec2 feature extractor, fairseq ConvFeatureExtractionModel) and explicit row strides, so x / w / y may be
// column slices of wider matrices (attention heads: Q_h.K_h^T and P.V_h^T without copies).
// (K, stride) in {(1,1), (2,2), (3,1), (3,2)}; Cin, ldx, ldw multiples of 4.
extern "C" int avdf_conv1d_fwd_ex(const float* x, const float* w, const float* bias, float* y, int B, int Tin, int Cin,
                                  int Tout, int Cout, int K, int stride, int pad, int ldx, int ldw, int ldy, int act,
                                  void* stream) {
  AVDF_REQUIRE(x && w && y, "conv1d_fwd_ex: null pointer");
  AVDF_REQUIRE(B > 0 && Tin > 0 && Cin > 0 && Tout > 0 && Cout > 0 && pad >= 0, "conv1d_fwd_ex: bad dims");
  AVDF_REQUIRE((K == 1 && stride == 1) || (K == 2 && stride == 2) || (K == 3 && (stride == 1 || stride == 2)),
               "conv1d_fwd_ex: unsupported (K=%d, stride=%d)", K, stride);
  AVDF_REQUIRE(Cin % 4 == 0 && ldx % 4 == 0 && ldw % 4 == 0 && ldx >= Cin && ldw >= Cin * K && ldy >= Cout,
               "conv1d_fwd_ex: Cin/ldx/ldw must be multiples of 4 and cover the operands");
  AVDF_REQUIRE(Tout == (Tin + 2 * pad - K) / stride + 1, "conv1d_fwd_ex: Tout=%d inconsistent", Tout);
  AVDF_REQUIRE(aligned16(x) && aligned16(w), "conv1d_fwd_ex: x/w must be 16-byte aligned");
  return conv1d_fwd_launch(fwd_ptrs1(x, w, bias, nullptr, y, nullptr), 1, B, Tin, Cin, Tout, Cout, K, stride, pad, ldx,
                           ldw, ldy, act, (hipStream_t)stream);
}

static int conv1d_dgrad_launch(const DgradPtrs& ptrs, int nprob, int B, int Tin, int Cin, int Tout, int Cout, int K,
                               int stride, int act, hipStream_t st) {
  if (K == 1 && stride == 1) { Tin = Tout = B * Tin; B = 1; }      // plain GEMM over the token rows (see conv1d_fwd_launch)
  const TilePlan plan = pick_tile(B * nprob, Tin, Cin, true, K, stride, Cout);
#define LAUNCH_DG(KS_, S_, KC_, WM_, WN_, MI_, NI_, KG_)                                                  \
  do {                                                                                                    \
    constexpr int BM_ = WM_ * MI_ * 32, BN_ = WN_ * NI_ * 32;                                             \
    const int tilesT = cdiv(Tin, BM_);                                                                    \
    dim3 grid(B * tilesT, cdiv(Cin, BN_), nprob);                                                         \
    constexpr int RAd_ = (BM_ + KS_ - 2) / S_ + 2;                                                        \
    constexpr bool DBOK_ = 2 * (RAd_ * (KC_ + 4) + KC_ * BN_ * KS_ + 8) * 4 <= 150 * 1024;                \
    if constexpr ((KC_ / KG_) % 16 == 0) {                                                                \
      if (g_gemm_mode & 2) {                                                                              \
        hipLaunchKernelGGL((conv_dgrad_kernel<KS_, S_, WM_, WN_, MI_, NI_, KC_, KG_, false, true>), grid, \
                           dim3(64 * WM_ * WN_ * KG_), 0, st, ptrs, B, Tin, Cin, Tout, Cout, act, tilesT);           \
        break;                                                                                            \
      }                                                                                                   \
    }                                                                                                     \
    if constexpr (DBOK_) {                                                                                \
      if (plan.db) {                                                                                      \
        hipLaunchKernelGGL((conv_dgrad_kernel<KS_, S_, WM_, WN_, MI_, NI_, KC_, KG_, true>), grid,        \
                           dim3(64 * WM_ * WN_ * KG_), 0, st, ptrs, B, Tin, Cin, Tout, Cout, act, tilesT);           \
        break;                                                                                            \
      }                                                                                                   \
    }                                                                                                     \
    hipLaunchKernelGGL((conv_dgrad_kernel<KS_, S_, WM_, WN_, MI_, NI_, KC_, KG_>), grid, dim3(64 * WM_ * WN_ * KG_), \
                       0, st, ptrs, B, Tin, Cin, Tout, Cout, act, tilesT);                                \
  } while (0)
  if (K == 3 && stride == 1) { AVDF_GEMM_DISPATCH(3, 1, 32, 32, 64, 32, 16, 64, 2, LAUNCH_DG) }
  else if (K == 3 && stride == 2) { AVDF_GEMM_DISPATCH(3, 2, 32, 32, 64, 32, 16, 64, 2, LAUNCH_DG) }
  else { AVDF_GEMM_DISPATCH(1, 1, 64, 64, 128, 64, 32, 128, 1, LAUNCH_DG) }
#undef LAUNCH_DG
  AVDF_CHECK_LAUNCH("conv1d_bwd_data");
  return AVDF_OK;
}

// avdf_conv1d_fwd_ex with explicit batch strides (floats): batch b reads x + b*xbs, uses the weight matrix w + b*wbs
// and writes y + b*ybs -- all the heads of an attention layer (Q_h.K_h^T or P_h.V_h^T) in ONE launch.
extern "C" int avdf_conv1d_fwd_strided(const float* x, const float* w, const float* bias, float* y, int B, int Tin, int Cin,
                                       int Tout, int Cout, int K, int stride, int pad, int ldx, int ldw, int ldy,
                                       int64_t xbs, int64_t wbs, int64_t ybs, int act, void* stream) {
  AVDF_REQUIRE(x && w && y, "conv1d_fwd_strided: null pointer");
  AVDF_REQUIRE(B > 0 && Tin > 0 && Cin > 0 && Tout > 0 && Cout > 0 && pad >= 0, "conv1d_fwd_strided: bad dims");
  AVDF_REQUIRE((K == 1 && stride == 1) || (K == 2 && stride == 2) || (K == 3 && (stride == 1 || stride == 2)),
               "conv1d_fwd_strided: unsupported (K=%d, stride=%d)", K, stride);
  AVDF_REQUIRE(Cin % 4 == 0 && ldx % 4 == 0 && ldw % 4 == 0 && ldx >= Cin && ldw >= Cin * K && ldy >= Cout &&
                   xbs % 4 == 0 && wbs % 4 == 0 && xbs >= 0 && wbs >= 0 && ybs >= 0,
               "conv1d_fwd_strided: Cin / ld / batch strides must be non-negative multiples of 4 and cover the operands");
  AVDF_REQUIRE(Tout == (Tin + 2 * pad - K) / stride + 1, "conv1d_fwd_strided: Tout=%d inconsistent", Tout);
  AVDF_REQUIRE(aligned16(x) && aligned16(w), "conv1d_fwd_strided: x/w must be 16-byte aligned");
  return conv1d_fwd_launch(fwd_ptrs1(x, w, bias, nullptr, y, nullptr), 1, B, Tin, Cin, Tout, Cout, K, stride, pad, ldx,
                           ldw, ldy, act, (hipStream_t)stream, (long)xbs, (long)wbs, (long)ybs);
}

extern "C" int avdf_conv1d_bwd_data(const float* dy, const float* w, const float* mask, const float* ypre,
                                    float* dx, int B, int Tin, int Cin, int Tout, int Cout,
                                    int K, int stride, int act, void* stream) {
  const float* dys[1] = {dy};
  const float* ws_[1] = {w};
  const float* ms[1] = {mask};
  const float* ps[1] = {ypre};
  float* dxs[1] = {dx};
  return avdf_conv1d_bwd_data_grouped(1, dys, ws_, ms, ps, dxs, B, Tin, Cin, Tout, Cout, K, stride, act, stream);
}

extern "C" int avdf_conv1d_bwd_data_grouped(int n, const float* const* dy, const float* const* w,
                                            const float* const* mask, const float* const* ypre, float* const* dx,
                                            int B, int Tin, int Cin, int Tout, int Cout, int K, int stride, int act,
                                            void* stream) {
  AVDF_REQUIRE(n >= 1 && n <= 3 && dy && w && dx, "conv1d_bwd_data: 1..3 problems, non-null tables");
  AVDF_REQUIRE(B > 0 && Tin > 0 && Cin > 0 && Tout > 0 && Cout > 0, "conv1d_bwd_data: bad dims");
  AVDF_REQUIRE((K == 1 || K == 3) && (stride == 1 || stride == 2) && !(K == 1 && stride != 1), "conv1d_bwd_data: unsupported K/stride");
  AVDF_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, "conv1d_bwd_data: Cin,Cout must be multiples of 4 (%d,%d)", Cin, Cout);
  AVDF_REQUIRE(Tout == (Tin + 2 * (K / 2) - K) / stride + 1, "conv1d_bwd_data: Tout inconsistent");
  DgradPtrs g;
  for (int i = 0; i < 3; ++i) {
    const int j = i < n ? i : 0;
    AVDF_REQUIRE(dy[j] && w[j] && dx[j], "conv1d_bwd_data: null pointer (problem %d)", j);
    g.dy[i] = dy[j]; g.w[i] = w[j]; g.dx[i] = dx[j];
    g.mask[i] = mask ? mask[j] : nullptr; g.ypre[i] = ypre ? ypre[j] : nullptr;
    AVDF_REQUIRE(aligned16(g.dy[i]) && aligned16(g.w[i]) && aligned16(g.ypre[i]), "conv1d_bwd_data: 16-byte alignment");
  }
  return conv1d_dgrad_launch(g, n, B, Tin, Cin, Tout, Cout, K, stride, act, (hipStream_t)stream);
}

extern "C" int64_t avdf_conv1d_bwd_weight_ws(int B, int Tout, int Cin, int Cout, int K) {
  int splits = 1;                      // the stride is not known here: cover the plan of either stride
  for (int s_ = 0; s_ <= 2; ++s_) {
    const WgradPlan p = wgrad_plan(B, Tout, Cin, Cout, K, 1, s_);
    if (p.splits > splits) splits = p.splits;
    if (K == 1) {                        // K=1 launches run on the flattened (1, B*T) problem
      const WgradPlan q = wgrad_plan(1, B * Tout, Cin, Cout, K, 1, s_);
      if (q.splits > splits) splits = q.splits;
    }
  }
  const int64_t slab = (splits > 1) ? (int64_t)splits * Cout * Cin * K : 0;
  return slab + (int64_t)kColsumBlocks * Cout;
}

// one launch for `nprob` same-shape problems; `target[i]` = dw (atomic / unsplit) or the slab workspace
static int conv1d_wgrad_launch(const WgradPtrs& ptrs, int nprob, const WgradPlan& p, int B, int Tin, int Cin, int Tout,
                               int Cout, int K, int stride, int act, int accumulate, bool atomic, hipStream_t st) {
#define LAUNCH_WG(KS_, S_, WM_, WN_, MI_, NI_, KT_, KG_)                                                        \
  do {                                                                                                          \
    if constexpr (KS_ == 1 && (KT_ / KG_) % 16 == 0) {                                                          \
      if (g_gemm_mode & 2) {                                                                                    \
        if (atomic)                                                                                             \
          hipLaunchKernelGGL((conv_wgrad_kernel<KS_, S_, WM_, WN_, MI_, NI_, KT_, KG_, true, true>),            \
                             dim3(p.mtiles * nprob, p.ctiles, p.splits), dim3(256 * KG_), 0, st, ptrs, p.mtiles, B, \
                             Tin, Cin, Tout, Cout, act, p.tilesPerClip, p.tilesPerSplit, accumulate);           \
        else                                                                                                    \
          hipLaunchKernelGGL((conv_wgrad_kernel<KS_, S_, WM_, WN_, MI_, NI_, KT_, KG_, false, true>),           \
                             dim3(p.mtiles * nprob, p.ctiles, p.splits), dim3(256 * KG_), 0, st, ptrs, p.mtiles, B, \
                             Tin, Cin, Tout, Cout, act, p.tilesPerClip, p.tilesPerSplit, accumulate);           \
        break;                                                                                                  \
      }                                                                                                         \
    }                                                                                                           \
    if (atomic)                                                                                                 \
      hipLaunchKernelGGL((conv_wgrad_kernel<KS_, S_, WM_, WN_, MI_, NI_, KT_, KG_, true>),                      \
                         dim3(p.mtiles * nprob, p.ctiles, p.splits), dim3(256 * KG_), 0, st, ptrs, p.mtiles, B, \
                         Tin, Cin, Tout, Cout, act, p.tilesPerClip, p.tilesPerSplit, accumulate);               \
    else                                                                                                        \
      hipLaunchKernelGGL((conv_wgrad_kernel<KS_, S_, WM_, WN_, MI_, NI_, KT_, KG_, false>),                     \
                         dim3(p.mtiles * nprob, p.ctiles, p.splits), dim3(256 * KG_), 0, st, ptrs, p.mtiles, B, \
                         Tin, Cin, Tout, Cout, act, p.tilesPerClip, p.tilesPerSplit, accumulate);               \
  } while (0)
#define LAUNCH_WG_KG(KS_, S_)                                                   \
  if (p.small) {                                                                \
    if (p.kg == 2) LAUNCH_WG(KS_, S_, 2, 2, 1, 1, 128, 2);                      \
    else LAUNCH_WG(KS_, S_, 2, 2, 1, 1, 128, 1);                                \
  } else {                                                                      \
    if (p.kg == 2) LAUNCH_WG(KS_, S_, 2, 2, 2, 1, 64, 2);                       \
    else LAUNCH_WG(KS_, S_, 2, 2, 2, 1, 64, 1);                                 \
  }
  if (K == 3 && stride == 1) { LAUNCH_WG_KG(3, 1) }
  else if (K == 3 && stride == 2) { LAUNCH_WG_KG(3, 2) }
  else if (p.small && p.kg == 4) { LAUNCH_WG(1, 1, 2, 2, 1, 1, 128, 4); }
  else { LAUNCH_WG_KG(1, 1) }
#undef LAUNCH_WG_KG
#undef LAUNCH_WG
  AVDF_CHECK_LAUNCH("conv1d_bwd_weight");
  return AVDF_OK;
}

extern "C" int avdf_conv1d_bwd_weight(const float* dy, const float* x, const float* mask, const float* ypre,
                                      float* dw, float* dbias, float* ws, int B, int Tin, int Cin, int Tout,
                                      int Cout, int K, int stride, int act, int accumulate, void* stream) {
  AVDF_REQUIRE(dy && x && dw && ws, "conv1d_bwd_weight: null pointer");
  AVDF_REQUIRE(B > 0 && Tin > 0 && Cin > 0 && Tout > 0 && Cout > 0, "conv1d_bwd_weight: bad dims");
  AVDF_REQUIRE((K == 1 || K == 3) && (stride == 1 || stride == 2) && !(K == 1 && stride != 1), "conv1d_bwd_weight: unsupported K/stride");
  AVDF_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, "conv1d_bwd_weight: Cin,Cout must be multiples of 4 (%d,%d)", Cin, Cout);
  AVDF_REQUIRE(Tout == (Tin + 2 * (K / 2) - K) / stride + 1, "conv1d_bwd_weight: Tout inconsistent");
  AVDF_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(ws) && (!ypre || aligned16(ypre)), "conv1d_bwd_weight: 16-byte alignment");
  AVDF_REQUIRE(K == 1 || aligned16(dw), "conv1d_bwd_weight: dw of a k=3 layer must be 16-byte aligned (float4 row writes)");
  hipStream_t st = (hipStream_t)stream;
  if (K == 1 && stride == 1) { Tin = Tout = B * Tin; B = 1; }      // plain GEMM over the token rows
  const WgradPlan p = wgrad_plan(B, Tout, Cin, Cout, K, 1, stride);
  // accumulate != 0 (engine path): partial tiles and the bias column sums are added into dw / dbias with
  // float atomics -- one launch, no slabs; gradients then differ run-to-run in the last bits, like any
  // atomics-based wgrad.  accumulate == 0: deterministic slabs + reducer.
  const bool atomic = accumulate != 0;
  float* target = (p.splits > 1 && !atomic) ? ws : dw;
  WgradPtrs g;
  for (int i = 0; i < 3; ++i) { g.dy[i] = dy; g.x[i] = x; g.mask[i] = mask; g.ypre[i] = ypre; g.out[i] = target; g.dbias[i] = dbias; }
  int rc = conv1d_wgrad_launch(g, 1, p, B, Tin, Cin, Tout, Cout, K, stride, act, accumulate, atomic, st);
  if (rc) return rc;
  if (atomic) return AVDF_OK;
  if (p.splits > 1) {
    rc = avdf_reduce_slabs(ws, dw, p.splits, (int64_t)Cout * Cin * K, accumulate, st);
    if (rc) return rc;
  }
  if (dbias) {
    float* part = ws + (p.splits > 1 ? (size_t)p.splits * Cout * Cin * K : 0);
    hipLaunchKernelGGL(colsum_kernel, dim3(kColsumBlocks), dim3(256), 0, st, dy, mask, ypre, part, B * Tout, Cout, act);
    AVDF_CHECK_LAUNCH("colsum");
    rc = avdf_reduce_slabs(part, dbias, kColsumBlocks, Cout, accumulate, st);
    if (rc) return rc;
  }
  return AVDF_OK;
}

// n same-shape problems.  accumulate != 0: ONE launch (atomics into dw[i] / dbias[i]); accumulate == 0: the
// deterministic per-problem path, `ws` sized n * avdf_conv1d_bwd_weight_ws(...).
extern "C" int avdf_conv1d_bwd_weight_grouped(int n, const float* const* dy, const float* const* x,
                                              const float* const* mask, const float* const* ypre, float* const* dw,
                                              float* const* dbias, float* ws, int B, int Tin, int Cin, int Tout,
                                              int Cout, int K, int stride, int act, int accumulate, void* stream) {
  AVDF_REQUIRE(n >= 1 && n <= 3 && dy && x && dw && ws, "conv1d_bwd_weight_grouped: 1..3 problems, non-null tables");
  if (!accumulate) {
    const int64_t per = avdf_conv1d_bwd_weight_ws(B, Tout, Cin, Cout, K);
    for (int i = 0; i < n; ++i) {
      int rc = avdf_conv1d_bwd_weight(dy[i], x[i], mask ? mask[i] : nullptr, ypre ? ypre[i] : nullptr, dw[i],
                                      dbias ? dbias[i] : nullptr, ws + (size_t)i * ((per + 3) / 4 * 4), B, Tin, Cin, Tout,
                                      Cout, K, stride, act, 0, stream);
      if (rc) return rc;
    }
    return AVDF_OK;
  }
  AVDF_REQUIRE(B > 0 && Tin > 0 && Cin > 0 && Tout > 0 && Cout > 0, "conv1d_bwd_weight_grouped: bad dims");
  AVDF_REQUIRE((K == 1 || K == 3) && (stride == 1 || stride == 2) && !(K == 1 && stride != 1), "conv1d_bwd_weight_grouped: unsupported K/stride");
  AVDF_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, "conv1d_bwd_weight_grouped: Cin,Cout must be multiples of 4");
  AVDF_REQUIRE(Tout == (Tin + 2 * (K / 2) - K) / stride + 1, "conv1d_bwd_weight_grouped: Tout inconsistent");
  WgradPtrs g;
  for (int i = 0; i < 3; ++i) {
    const int j = i < n ? i : 0;
    AVDF_REQUIRE(dy[j] && x[j] && dw[j], "conv1d_bwd_weight_grouped: null pointer (problem %d)", j);
    g.dy[i] = dy[j]; g.x[i] = x[j]; g.out[i] = dw[j];
    g.mask[i] = mask ? mask[j] : nullptr; g.ypre[i] = ypre ? ypre[j] : nullptr; g.dbias[i] = dbias ? dbias[j] : nullptr;
    AVDF_REQUIRE(aligned16(g.dy[i]) && aligned16(g.x[i]) && aligned16(g.ypre[i]), "conv1d_bwd_weight_grouped: 16-byte alignment");
    AVDF_REQUIRE(K == 1 || aligned16(g.out[i]), "conv1d_bwd_weight_grouped: dw of a k=3 layer must be 16-byte aligned");
  }
  if (K == 1 && stride == 1) { Tin = Tout = B * Tin; B = 1; }
  WgradPlan p = wgrad_plan(B, Tout, Cin, Cout, K, n, stride);
  return conv1d_wgrad_launch(g, n, p, B, Tin, Cin, Tout, Cout, K, stride, act, accumulate, true, (hipStream_t)stream);
}

// K=1 / stride-1 layers (every Linear of the transformer blocks and heads): data gradient and weight gradient of
// n <= 3 same-shape problems in ONE launch (conv_bwd_pair_k1_kernel).  Weight / bias gradients are ACCUMULATED into
// dw[i] / dbias[i] with float atomics (the engine's flat-gradient path); dx[i] is overwritten.  g_pair_mode 0 turns
// the fusion off (two launches, the reference point of tools/gemm_bench.py).
static int g_pair_mode = [] { const char* e = getenv("AVDF_BWD_PAIR"); return e ? atoi(e) : 1; }();
static const int g_pair_xcd = [] { const char* e = getenv("AVDF_PAIR_XCD"); return e ? atoi(e) : 0; }();   // 1: XCD-aware work order (halves the fabric traffic, -0.55 % step throughput: DESIGN.md section 6.2)
extern "C" int avdf_bwd_pair_mode(int mode) { g_pair_mode = mode; return AVDF_OK; }

extern "C" int avdf_conv1d_bwd_pair_grouped(int n, const float* const* dy, const float* const* w, const float* const* x,
                                            const float* const* mask, const float* const* ypre, float* const* dx,
                                            float* const* dw, float* const* dbias, int B, int T, int Cin, int Cout,
                                            int act, void* stream) {
  return avdf_conv1d_bwd_pair_dact(n, dy, w, x, mask, ypre, dx, dw, dbias, nullptr, AVDF_ACT_NONE, B, T, Cin, Cout, act, stream);
}

// The same with the producer's activation derivative folded into the data gradient: dx[i] = (g w) * xact'(xpre[i]),
// xpre[i] [B, T, Cin] = the pre-activation tensor whose activation produced this layer's input x[i] (fc2 of an MLP:
// x = GELU(xpre), xpre = fc1's pre-activation; the next backward GEMM pair then runs on dx as it is, act = NONE).
extern "C" int avdf_conv1d_bwd_pair_dact(int n, const float* const* dy, const float* const* w, const float* const* x,
                                         const float* const* mask, const float* const* ypre, float* const* dx,
                                         float* const* dw, float* const* dbias, const float* const* xpre, int xact, int B,
                                         int T, int Cin, int Cout, int act, void* stream) {
  AVDF_REQUIRE(n >= 1 && n <= 3 && dy && w && x && dx && dw, "conv1d_bwd_pair: 1..3 problems, non-null tables");
  AVDF_REQUIRE(B > 0 && T > 0 && Cin > 0 && Cout > 0, "conv1d_bwd_pair: bad dims");
  AVDF_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, "conv1d_bwd_pair: Cin,Cout must be multiples of 4 (%d,%d)", Cin, Cout);
  DgradPtrs d;
  WgradPtrs g;
  for (int i = 0; i < 3; ++i) {
    const int j = i < n ? i : 0;
    AVDF_REQUIRE(dy[j] && w[j] && x[j] && dx[j] && dw[j], "conv1d_bwd_pair: null pointer (problem %d)", j);
    d.dy[i] = dy[j]; d.w[i] = w[j]; d.dx[i] = dx[j];
    g.dy[i] = dy[j]; g.x[i] = x[j]; g.out[i] = dw[j]; g.dbias[i] = dbias ? dbias[j] : nullptr;
    d.mask[i] = g.mask[i] = mask ? mask[j] : nullptr;
    d.ypre[i] = g.ypre[i] = ypre ? ypre[j] : nullptr;
    d.xpre[i] = xpre ? xpre[j] : nullptr;
    AVDF_REQUIRE(aligned16(d.dy[i]) && aligned16(d.w[i]) && aligned16(g.x[i]) && aligned16(d.ypre[i]),
                 "conv1d_bwd_pair: 16-byte alignment");
  }
  d.xact = xpre ? xact : AVDF_ACT_NONE;
  hipStream_t st = (hipStream_t)stream;
  T = B * T; B = 1;                                                  // plain GEMM over the token rows
  WgradPlan p = wgrad_plan(B, T, Cin, Cout, 1, n, 1);
  TilePlan plan = pick_tile(B * n, T, Cin, true, 1, 1, Cout);
  bool bf3_row = false;
  {
    // plans measured for the fused launch (table kind 4: a = dgrad tile, b = K-groups | wgrad token splits << 3)
    int a = -1, b = -1;
    if (g_override[4][0] >= 0) { a = g_override[4][0]; b = g_override[4][1]; bf3_row = true; }
    else if (g_override[4][0] != -2) {
      const TunedPlan* t = (g_gemm_mode & 2) ? tuned_lookup(5, 1, 1, n * T, Cin, Cout) : nullptr;   // kind 5: pair in bf16x3 mode
      bf3_row = t != nullptr;
      if (!t) t = tuned_lookup(4, 1, 1, n * T, Cin, Cout);
      if (t) { a = t->a; b = t->b; }
    }
    if (a >= 0 && a <= 6 && (b >> 3) >= 1) {
      const int mk_[7] = {1, 2, 4, 4, 2, 4, 4};
      int kg = b & 7;
      if (kg != 1 && kg != 2 && kg != 4) kg = 1;
      if (kg > mk_[a]) kg = mk_[a];
      if (a == 6 && kg == 1) kg = 2;                 // the wgrad half needs 4 waves per K-group: >= 256 threads
      plan = TilePlan{a, kg, 0};
      const int total = B * p.tilesPerClip;
      int want = b >> 3;
      if (want > total) want = total;
      p.tilesPerSplit = cdiv(total, want);
      p.splits = cdiv(total, p.tilesPerSplit);
    }
  }
  // split-bf16 mode without a measured row: the 1024-thread variants need > 64 VGPRs there (one workgroup per CU
  // instead of two); every in-situ search so far (profiles/r02_f_*) picked two K-groups for the big layers
  if ((g_gemm_mode & 2) && !bf3_row && g_override[1][0] < 0 && plan.kg == 4 && plan.tile != 5) plan.kg = 2;
  if (plan.tile == 6 && plan.kg == 1) plan.kg = 2;   // 64x32 tile: two waves per K-group, the wgrad half needs four per workgroup
  if (!g_pair_mode || !p.small) {
    int rc = conv1d_dgrad_launch(d, n, B, T, Cin, T, Cout, 1, 1, act, st);
    if (rc) return rc;
    return conv1d_wgrad_launch(g, n, p, B, T, Cin, T, Cout, 1, 1, act, 1, true, st);
  }
#define LAUNCH_PAIR(KS_, S_, KC_, WM_, WN_, MI_, NI_, KG_)                                                          \
  do {                                                                                                              \
    constexpr int BM_ = WM_ * MI_ * 32, BN_ = WN_ * NI_ * 32;                                                       \
    const int tilesT0 = cdiv(T, BM_);                                                                               \
    const int tilesT = g_pair_xcd ? tilesT0 : -tilesT0;                                                             \
    const int dgx = B * tilesT0, dgy = cdiv(Cin, BN_);                                                              \
    const int wgx = p.mtiles * n, wgy = p.ctiles, wgz = p.splits;                                                   \
    const int nW = wgx * wgy * wgz, nD = dgx * dgy * n;                                                             \
    if constexpr ((WM_ * WN_ * KG_) % 4 == 0 && (KC_ / KG_) % 16 == 0) {                                            \
      if (g_gemm_mode & 2) {                                                                                        \
        hipLaunchKernelGGL((conv_bwd_pair_k1_kernel<WM_, WN_, MI_, NI_, KC_, KG_, true>), dim3(nW + nD),            \
                           dim3(64 * WM_ * WN_ * KG_), 0, st, d, g, nW, wgx, wgy, wgz, dgx, dgy, p.mtiles, B, T, Cin, \
                           Cout, act, tilesT, p.tilesPerClip, p.tilesPerSplit);                                     \
        break;                                                                                                      \
      }                                                                                                             \
    }                                                                                                               \
    if constexpr ((WM_ * WN_ * KG_) % 4 == 0) {                                                                     \
      hipLaunchKernelGGL((conv_bwd_pair_k1_kernel<WM_, WN_, MI_, NI_, KC_, KG_>), dim3(nW + nD), dim3(64 * WM_ * WN_ * KG_), 0,  \
                         st, d, g, nW, wgx, wgy, wgz, dgx, dgy, p.mtiles, B, T, Cin, Cout, act, tilesT,             \
                         p.tilesPerClip, p.tilesPerSplit);                                                          \
    } else {                                                                                                        \
      AVDF_REQUIRE(false, "conv1d_bwd_pair: plan with fewer than 4 waves per workgroup");                           \
    }                                                                                                               \
  } while (0)
  AVDF_GEMM_DISPATCH(1, 1, 64, 64, 128, 64, 32, 128, 1, LAUNCH_PAIR)
#undef LAUNCH_PAIR
  AVDF_CHECK_LAUNCH("conv1d_bwd_pair");
  return AVDF_OK;
}

// tuner hook (tools/plan_search.py): replace the AVDF_TUNED_EXTRA rows of the process ("kind,K,stride,M,N,Kred,a,b;...",
// NULL or "" = none).  Not for production use: process-wide, not synchronised with concurrent launches.
extern "C" int avdf_tuned_extra(const char* rows) {
  g_nextra = parse_extra(rows);
  return AVDF_OK;
}

// tuner hook (tools/autotune.py): kind 0 fwd / 1 dgrad: (tile, K-groups); 2 wgrad: (token splits, K-groups); a < 0 = off
extern "C" int avdf_gemm_override(int kind, int a, int b) {
  AVDF_REQUIRE(kind >= 0 && kind < 5 && kind != 3, "gemm_override: kind in {0, 1, 2, 4}");
  g_override[kind][0] = a;
  g_override[kind][1] = b;
  return AVDF_OK;
}

// bit mask: 0 = exact fp32 MFMA (default); 1 = split-bf16 (bf16x3) operands in the forward conv GEMMs; 2 = in the
// backward conv GEMMs (dgrad, K = 1 wgrad); 3 = both.  Per calling thread.
extern "C" int avdf_gemm_mode(int mode) {
  AVDF_REQUIRE(mode >= 0 && mode <= 3, "gemm_mode: bit mask of 1 (bf16x3 forward) and 2 (bf16x3 backward)");
  g_gemm_mode = mode;
  return AVDF_OK;
}

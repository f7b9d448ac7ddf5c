// FPN1D tail of all pyramid levels in five launches (two forward, three backward) (reference: libs/modeling/necks.py:62-93): after the per-level lateral
// 1x1 convs, the top-down pathway (nearest x2 up-sampling + add, coarse -> fine), the masked depthwise k3 conv and the
// LayerNorm of EVERY level, written straight into the gapped level sequence [B, sum(T_l + 1), C] the shared heads run on
// (meta_archs.gap_levels).  Per level the reference issues add_upsampled / dwconv / LayerNorm one after the other (17
// launches forward + the concatenation, 23 backward for the six levels of exp10); the levels are a few hundred KB each,
// so every one of those launches sits on the launch floor.
//
//   td_l[t]  = lat_l[t] + td_{l+1}[t >> 1]                              (top-down, td_{n-1} = lat_{n-1})
//   u_l[t]   = (w_l[c,0] td_l[t-1] + w_l[c,1] td_l[t] + w_l[c,2] td_l[t+1]) * mask_l[t]       (zero padding per level)
//   y_l[t]   = (u - mean_c(u)) * rstd * gamma_l + beta_l                (eps inside the square root, blocks.py:97-112)
//
// Every kernel is row-parallel (one wave per gapped row, or one float4 of channels per thread): the top-down sums are
// evaluated per row as the nested sum lat_l + (lat_{l+1} + (lat_{l+2} + ...)) over the <= n ancestors of the row -- the
// same association as the reference's coarse -> fine loop -- and their gradient per row as the sum over the row's
// descendants; the tensors are L2-resident (12 MB for exp10 at 8 clips), so the re-reads cost nothing next to a launch.
#include "common.h"

namespace {

constexpr int kMaxLevels = 8;
struct LevelPtrs { const float* p[kMaxLevels]; };
struct LevelOut { float* p[kMaxLevels]; };
struct LevelLens { int len[kMaxLevels]; int off[kMaxLevels]; int n; };       // off: first row of the level in the gapped sequence

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

__device__ __forceinline__ int level_of_gapped_row(const LevelLens& L, int g, int& t) {
#pragma unroll
  for (int l = 0; l < kMaxLevels; ++l) {
    if (l >= L.n) break;
    if (g < L.off[l] + L.len[l] + 1) { t = g - L.off[l]; return t < L.len[l] ? l : -1; }
  }
  t = 0;
  return -1;
}

// td (gapped layout; the gap rows are never read) from the laterals: td_l[t] = lat_l[t] + td_{l+1}[t >> 1].
// One wave per gapped row (level and row index are wave-uniform), a lane = four channels, <= n independent loads.
__global__ __launch_bounds__(256) void fpn_topdown_fwd_kernel(LevelPtrs lat, LevelLens L, float* __restrict__ td, int B, int C,
                                                              int Tg) {
  const int lane = threadIdx.x & 63, c = lane * 4;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * Tg || c >= C) return;
  const int g = row % Tg, b = row / Tg;
  int t;
  const int l = level_of_gapped_row(L, g, t);
  if (l < 0) return;
  float4 v[kMaxLevels];
#pragma unroll
  for (int j = 0; j < kMaxLevels; ++j)
    v[j] = (j >= l && j < L.n) ? ld4(lat.p[j] + ((size_t)b * L.len[j] + (t >> (j - l))) * C + c) : zero4();
  float4 acc = zero4();
#pragma unroll
  for (int j = kMaxLevels - 1; j >= 0; --j)                    // coarsest ancestor first: the reference's association
    if (j >= l && j < L.n) acc = add4(v[j], acc);
  st4(td + (size_t)row * C + c, acc);
}

// masked depthwise k3 + LayerNorm of every level; one wave per gapped row (gap rows: y = 0, mask 0).
__global__ __launch_bounds__(256) void fpn_tail_fwd_kernel(const float* __restrict__ td, LevelPtrs mask, LevelPtrs dw,
                                                           LevelPtrs gamma, LevelPtrs beta, LevelLens L, float* __restrict__ y,
                                                           float* __restrict__ gmask, float* __restrict__ u_o,
                                                           float* __restrict__ mean_o, float* __restrict__ rstd_o, int B, int C,
                                                           int Tg, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * Tg) return;
  const int g = row % Tg, b = row / Tg, c = lane * 4;
  const bool on = c < C;
  const size_t ro = (size_t)row * C + c;
  int t;
  const int l = level_of_gapped_row(L, g, t);
  if (l < 0) {
    if (on) st4(y + ro, zero4());
    if (lane == 0) gmask[row] = 0.f;
    return;
  }
  const float m = mask.p[l][(size_t)b * L.len[l] + t];
  float4 u = zero4();
  if (on) {
    const float* w = dw.p[l] + c * 3;                          // [C][1][3]
    const float* x = td + ro;
    const float4 x1 = ld4(x);
    const float4 x0 = t > 0 ? ld4(x - C) : zero4();
    const float4 x2 = t + 1 < L.len[l] ? ld4(x + C) : zero4();
    u.x = (w[0] * x0.x + w[1] * x1.x + w[2] * x2.x) * m;
    u.y = (w[3] * x0.y + w[4] * x1.y + w[5] * x2.y) * m;
    u.z = (w[6] * x0.z + w[7] * x1.z + w[8] * x2.z) * m;
    u.w = (w[9] * x0.w + w[10] * x1.w + w[11] * x2.w) * m;
  }
  const float mean = wave_sum(u.x + u.y + u.z + u.w) / (float)C;
  float q = 0.f;
  if (on) {
    const float a0 = u.x - mean, a1 = u.y - mean, a2 = u.z - mean, a3 = u.w - mean;
    q = a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) {
    gmask[row] = m;
    if (mean_o) { mean_o[row] = mean; rstd_o[row] = rstd; }
  }
  if (on) {
    if (u_o) st4(u_o + ro, u);
    const float4 ga = ld4(gamma.p[l] + c), be = ld4(beta.p[l] + c);
    st4(y + ro, make_float4((u.x - mean) * rstd * ga.x + be.x, (u.y - mean) * rstd * ga.y + be.y,
                            (u.z - mean) * rstd * ga.z + be.z, (u.w - mean) * rstd * ga.w + be.w));
  }
}

// Column-sum items of the two backward kernels: item -> (level l, clip b, first row t0 of a run of kColRows rows of that
// level).  Items are numbered level by level: level l holds B * ceil(len_l / kColRows) of them.
constexpr int kColRows = 64;
__device__ __forceinline__ bool col_item(const LevelLens& L, int B, int item, int& l, int& b, int& t0) {
  for (l = 0; l < L.n; ++l) {
    const int per = (L.len[l] + kColRows - 1) / kColRows, cnt = B * per;
    if (item < cnt) { b = item / per; t0 = (item % per) * kColRows; return true; }
    item -= cnt;
  }
  return false;
}

// LayerNorm backward of every level.  Workgroups [0, nrow): one wave per gapped row, dum = d(loss)/d(u) * mask.
// Workgroups [nrow, ...): column sums for dgamma_l / dbeta_l -- a workgroup takes 64 rows of one level and clip (16 per
// wave, a lane keeps its four channels: plain streaming loads, nothing to reduce inside a row), folds its four waves in
// LDS and adds 2C floats with float atomics: ~200 workgroups per launch instead of a partial per row group.
__global__ __launch_bounds__(256) void fpn_tail_bwd_ln_kernel(const float* __restrict__ dy, const float* __restrict__ u,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              LevelPtrs gamma, LevelPtrs mask, LevelLens L,
                                                              float* __restrict__ dum, LevelOut dgamma, LevelOut dbeta, int B,
                                                              int C, int Tg, int nrow) {
  __shared__ float4 red[4][2][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c = lane * 4;
  const bool on = c < C;
  if ((int)blockIdx.x < nrow) {
    const int row = blockIdx.x * 4 + wid;
    if (row >= B * Tg) return;
    const int g = row % Tg, b = row / Tg;
    const size_t ro = (size_t)row * C + c;
    int t;
    const int l = level_of_gapped_row(L, g, t);
    if (l < 0) {
      if (on) st4(dum + ro, zero4());
      return;
    }
    float4 xh = zero4(), gd = zero4();
    const float mu = mean[row], rs = rstd[row];
    if (on) {
      const float4 d = ld4(dy + ro), uu = ld4(u + ro), ga = ld4(gamma.p[l] + c);
      xh = make_float4((uu.x - mu) * rs, (uu.y - mu) * rs, (uu.z - mu) * rs, (uu.w - mu) * rs);
      gd = make_float4(d.x * ga.x, d.y * ga.y, d.z * ga.z, d.w * ga.w);
    }
    const float m1 = wave_sum(gd.x + gd.y + gd.z + gd.w) / (float)C;
    const float m2 = wave_sum(gd.x * xh.x + gd.y * xh.y + gd.z * xh.z + gd.w * xh.w) / (float)C;
    if (on) {
      const float k = rs * mask.p[l][(size_t)b * L.len[l] + t];
      st4(dum + ro, make_float4((gd.x - m1 - xh.x * m2) * k, (gd.y - m1 - xh.y * m2) * k, (gd.z - m1 - xh.z * m2) * k,
                                (gd.w - m1 - xh.w * m2) * k));
    }
    return;
  }
  int l, b, t0;
  if (!col_item(L, B, blockIdx.x - nrow, l, b, t0) || dgamma.p[l] == nullptr) return;
  float4 pg = zero4(), pb = zero4();
  if (on) {
    const int tend = min(t0 + kColRows, L.len[l]);
#pragma unroll 4
    for (int t = t0 + wid; t < tend; t += 4) {
      const int row = b * Tg + L.off[l] + t;
      const float mu = mean[row], rs = rstd[row];
      const float4 d = ld4(dy + (size_t)row * C + c), uu = ld4(u + (size_t)row * C + c);
      pg.x += d.x * (uu.x - mu) * rs; pg.y += d.y * (uu.y - mu) * rs; pg.z += d.z * (uu.z - mu) * rs; pg.w += d.w * (uu.w - mu) * rs;
      pb.x += d.x; pb.y += d.y; pb.z += d.z; pb.w += d.w;
    }
  }
  red[wid][0][lane] = pg;
  red[wid][1][lane] = pb;
  __syncthreads();
  if (wid < 2 && on) {
    const float4 a0 = red[0][wid][lane], a1 = red[1][wid][lane], a2 = red[2][wid][lane], a3 = red[3][wid][lane];
    float* dst = (wid == 0 ? dgamma.p[l] : dbeta.p[l]) + c;
    atomicAdd(dst + 0, a0.x + a1.x + a2.x + a3.x); atomicAdd(dst + 1, a0.y + a1.y + a2.y + a3.y);
    atomicAdd(dst + 2, a0.z + a1.z + a2.z + a3.z); atomicAdd(dst + 3, a0.w + a1.w + a2.w + a3.w);
  }
}

// depthwise conv backward of every level.  Workgroups [0, nrow): dtd_l[t] = sum_k w_l[c,k] dum_l[t + 1 - k], one float4
// per thread (gapped; gap rows untouched).  Workgroups [nrow, ...): ddw_l[c,k] += sum_t td_l[t + k - 1] dum_l[t], the
// column-sum scheme of the kernel above (3C floats per workgroup).
__global__ __launch_bounds__(256) void fpn_tail_bwd_conv_kernel(const float* __restrict__ dum, const float* __restrict__ td,
                                                                LevelPtrs dw, LevelLens L, float* __restrict__ dtd, LevelOut ddw,
                                                                int B, int C, int Tg, int nrow) {
  __shared__ float red[4][768];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c = lane * 4;
  const bool on = c < C;
  if ((int)blockIdx.x < nrow) {
    const int row = blockIdx.x * 4 + wid;
    if (row >= B * Tg || !on) return;
    int t;
    const int l = level_of_gapped_row(L, row % Tg, t);
    if (l < 0) return;
    const float* w = dw.p[l] + c * 3;
    const float* dr = dum + (size_t)row * C + c;
    const float4 d1 = ld4(dr);
    const float4 d0 = t > 0 ? ld4(dr - C) : zero4();                   // dum[t-1]
    const float4 d2 = t + 1 < L.len[l] ? ld4(dr + C) : zero4();        // dum[t+1]
    st4(dtd + (size_t)row * C + c, make_float4(w[0] * d2.x + w[1] * d1.x + w[2] * d0.x, w[3] * d2.y + w[4] * d1.y + w[5] * d0.y,
                                               w[6] * d2.z + w[7] * d1.z + w[8] * d0.z, w[9] * d2.w + w[10] * d1.w + w[11] * d0.w));
    return;
  }
  int l, b, t0;
  if (!col_item(L, B, blockIdx.x - nrow, l, b, t0) || ddw.p[l] == nullptr) return;
  float pw[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) pw[i] = 0.f;
  if (on) {
    const int len = L.len[l], tend = min(t0 + kColRows, len);
#pragma unroll 2
    for (int t = t0 + wid; t < tend; t += 4) {
      const size_t ro = (size_t)(b * Tg + L.off[l] + t) * C + c;
      const float4 d1 = ld4(dum + ro);
      const float4 x1 = ld4(td + ro);
      const float4 x0 = t > 0 ? ld4(td + ro - C) : zero4();
      const float4 x2 = t + 1 < len ? ld4(td + ro + C) : zero4();
      pw[0] += x0.x * d1.x; pw[1] += x1.x * d1.x; pw[2] += x2.x * d1.x;
      pw[3] += x0.y * d1.y; pw[4] += x1.y * d1.y; pw[5] += x2.y * d1.y;
      pw[6] += x0.z * d1.z; pw[7] += x1.z * d1.z; pw[8] += x2.z * d1.z;
      pw[9] += x0.w * d1.w; pw[10] += x1.w * d1.w; pw[11] += x2.w * d1.w;
    }
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) red[wid][(c * 3 + i) % 768] = on ? pw[i] : 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < C * 3; i += 256) atomicAdd(ddw.p[l] + i, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
}

// top-down backward, one workgroup per (clip, row of the coarsest level): the rows of all levels under it form a binary
// tree (T_l = T_{n-1} << (n-1-l)); walked leaves -> root, a level per phase (its rows round-robin over the four waves):
//   dlat_l[t] = dtd_l[t] + dlat_{l-1}[2t] + dlat_{l-1}[2t+1]      -- the reference's pair sums, every row read / written once.
__global__ __launch_bounds__(256) void fpn_topdown_bwd_kernel(const float* __restrict__ dtd, LevelLens L, LevelOut dlat, int C,
                                                              int Tg) {
  const int s = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c = lane * 4;
  const bool on = c < C;
  const float* base = dtd + (size_t)b * Tg * C + c;
  for (int l = 0; l < L.n; ++l) {
    const int sh = L.n - 1 - l;
    if (on) {
      for (int t = (s << sh) + wid; t < (s + 1) << sh; t += 4) {
        float4 a = ld4(base + (size_t)(L.off[l] + t) * C);
        if (l > 0) {
          const float* ch = dlat.p[l - 1] + ((size_t)b * L.len[l - 1] + 2 * t) * C + c;
          a = add4(a, add4(ld4(ch), ld4(ch + C)));
        }
        st4(dlat.p[l] + ((size_t)b * L.len[l] + t) * C + c, a);
      }
    }
    __syncthreads();                                              // the next level reads what other waves just wrote
  }
}

bool fill_lens(LevelLens& L, int n, const int* lens) {
  if (n < 1 || n > kMaxLevels) return false;
  L.n = n;
  int off = 0;
  for (int l = 0; l < kMaxLevels; ++l) {
    L.len[l] = l < n ? lens[l] : 0;
    L.off[l] = off;
    if (l < n) off += lens[l] + 1;
    if (l > 0 && l < n && lens[l] * 2 != lens[l - 1]) return false;      // scale factor 2 between levels
  }
  return true;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// n levels, lens[l] = T_l (T_{l} = 2 T_{l+1}); lat / mask / dw / gamma / beta: host tables of n device pointers.
// y, u [B, Tg, C], td [B, Tg, C] (workspace kept for the backward), gmask / mean / rstd [B, Tg], Tg = sum(T_l + 1).
// u / mean / rstd may be NULL (inference).
extern "C" int avdf_fpn_tail_fwd(int n, const int* lens, const float* const* lat, const float* const* mask,
                                 const float* const* dw, const float* const* gamma, const float* const* beta, float* td,
                                 float* y, float* gmask, float* u, float* mean, float* rstd, int B, int C, float eps,
                                 void* stream) {
  AVDF_REQUIRE(lens && lat && mask && dw && gamma && beta && td && y && gmask, "fpn_tail_fwd: null pointer");
  AVDF_REQUIRE((u == nullptr) == (mean == nullptr) && (mean == nullptr) == (rstd == nullptr), "fpn_tail_fwd: u / mean / rstd go together");
  AVDF_REQUIRE(B > 0 && C > 0 && C <= 256 && C % 4 == 0, "fpn_tail_fwd: C must be a multiple of 4, <= 256 (C=%d)", C);
  LevelLens L;
  AVDF_REQUIRE(fill_lens(L, n, lens), "fpn_tail_fwd: 1..%d levels, each half as long as the one before", kMaxLevels);
  LevelPtrs pl, pm, pw, pg, pb;
  for (int l = 0; l < kMaxLevels; ++l) {
    const int j = l < n ? l : 0;
    pl.p[l] = lat[j]; pm.p[l] = mask[j]; pw.p[l] = dw[j]; pg.p[l] = gamma[j]; pb.p[l] = beta[j];
    AVDF_REQUIRE(pl.p[l] && pm.p[l] && pw.p[l] && pg.p[l] && pb.p[l] && aligned16(pl.p[l]) && aligned16(pg.p[l]) && aligned16(pb.p[l]),
                 "fpn_tail_fwd: null / misaligned operand of level %d", j);
  }
  AVDF_REQUIRE(aligned16(td) && aligned16(y) && aligned16(u), "fpn_tail_fwd: 16-byte alignment");
  const int Tg = L.off[n - 1] + L.len[n - 1] + 1;
  AVDF_REQUIRE((int64_t)B * Tg * C < (1ll << 31), "fpn_tail_fwd: tensor too large for 32-bit indexing");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(fpn_topdown_fwd_kernel, dim3((unsigned)cdiv64((int64_t)B * Tg, 4)), dim3(256), 0, st, pl, L, td, B, C, Tg);
  AVDF_CHECK_LAUNCH("fpn_topdown_fwd");
  hipLaunchKernelGGL(fpn_tail_fwd_kernel, dim3((unsigned)cdiv64((int64_t)B * Tg, 4)), dim3(256), 0, st, td, pm, pw, pg, pb, L, y,
                     gmask, u, mean, rstd, B, C, Tg, eps);
  AVDF_CHECK_LAUNCH("fpn_tail_fwd");
  return AVDF_OK;
}

// dy [B, Tg, C] -> dlat[l] [B, T_l, C]; dgamma / dbeta / ddw: tables of n device pointers that are ACCUMULATED into (float
// atomics; NULL entries / tables skip the parameter gradients).  dum, dtd [B, Tg, C]: workspaces.
extern "C" int avdf_fpn_tail_bwd(int n, const int* lens, const float* dy, const float* u, const float* mean, const float* rstd,
                                 const float* td, const float* const* mask, const float* const* dw, const float* const* gamma,
                                 float* dum, float* dtd, float* const* dlat, float* const* ddw, float* const* dgamma,
                                 float* const* dbeta, int B, int C, void* stream) {
  AVDF_REQUIRE(lens && dy && u && mean && rstd && td && mask && dw && gamma && dum && dtd && dlat, "fpn_tail_bwd: null pointer");
  AVDF_REQUIRE(B > 0 && C > 0 && C <= 256 && C % 4 == 0, "fpn_tail_bwd: C must be a multiple of 4, <= 256 (C=%d)", C);
  AVDF_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "fpn_tail_bwd: dgamma / dbeta go together");
  LevelLens L;
  AVDF_REQUIRE(fill_lens(L, n, lens), "fpn_tail_bwd: 1..%d levels, each half as long as the one before", kMaxLevels);
  LevelPtrs pm, pw, pg;
  LevelOut ol, ow, og, ob;
  for (int l = 0; l < kMaxLevels; ++l) {
    const int j = l < n ? l : 0;
    pm.p[l] = mask[j]; pw.p[l] = dw[j]; pg.p[l] = gamma[j];
    ol.p[l] = dlat[j];
    ow.p[l] = (ddw && l < n) ? ddw[j] : nullptr;
    og.p[l] = (dgamma && l < n) ? dgamma[j] : nullptr;
    ob.p[l] = (dbeta && l < n) ? dbeta[j] : nullptr;
    AVDF_REQUIRE(pm.p[l] && pw.p[l] && pg.p[l] && ol.p[l] && aligned16(pg.p[l]) && aligned16(ol.p[l]),
                 "fpn_tail_bwd: null / misaligned operand of level %d", j);
    AVDF_REQUIRE((og.p[l] == nullptr) == (ob.p[l] == nullptr), "fpn_tail_bwd: dgamma / dbeta of level %d go together", j);
  }
  AVDF_REQUIRE(aligned16(dy) && aligned16(u) && aligned16(td) && aligned16(dum) && aligned16(dtd), "fpn_tail_bwd: 16-byte alignment");
  const int Tg = L.off[n - 1] + L.len[n - 1] + 1;
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = (int64_t)B * Tg;
  AVDF_REQUIRE(rows * C < (1ll << 31), "fpn_tail_bwd: tensor too large for 32-bit indexing");
  const int nrow = (int)cdiv64(rows, 4);
  int ncol = 0;
  for (int l = 0; l < n; ++l) ncol += B * cdiv(L.len[l], 64);
  hipLaunchKernelGGL(fpn_tail_bwd_ln_kernel, dim3(nrow + (dgamma ? ncol : 0)), dim3(256), 0, st, dy, u, mean, rstd, pg, pm, L, dum, og,
                     ob, B, C, Tg, nrow);
  AVDF_CHECK_LAUNCH("fpn_tail_bwd_ln");
  hipLaunchKernelGGL(fpn_tail_bwd_conv_kernel, dim3(nrow + (ddw ? ncol : 0)), dim3(256), 0, st, dum, td, pw, L, dtd, ow, B, C, Tg, nrow);
  AVDF_CHECK_LAUNCH("fpn_tail_bwd_conv");
  hipLaunchKernelGGL(fpn_topdown_bwd_kernel, dim3(L.len[n - 1], B), dim3(256), 0, st, dtd, L, ol, C, Tg);
  AVDF_CHECK_LAUNCH("fpn_topdown_bwd");
  return AVDF_OK;
}

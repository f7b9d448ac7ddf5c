// Fused "LayerNorm -> depthwise conv k3 (stride s) * mask -> LayerNorm" chains for the q / k / v inputs of the
// conv-attention blocks, three branches per launch, forward and backward (gfx950).
//
// Replaces, per attention block, ln1 | lnq,lnk,lnv (libs/modeling/blocks.py:1307-1309, 866-868) followed by
// query_conv/key_conv/value_conv (MaskedConv1D depthwise k3, blocks.py:239-262, 931-955) and
// query_norm/key_norm/value_norm (blocks.py:1166-1173, 724-731) and their autograd backward: 7-9 launches
// forward and ~13 backward become 1 + 2.  HBM-bound: one wave owns one token row (C <= 256: one float4 per lane),
// row reductions are wave shuffles, the conv halo lives in LDS.
//
// Forward : z = LN_a(x) for the tile's rows (+1 halo each side) -> LDS; u[t] = sum_k z[t*s+k-1]*w[k] * mask[t];
//           y = LN_b(u).  u and both LNs' (mean, rstd) are kept for the backward.
// Backward: du = LN_b'(dy) * mask -> LDS; per input row dz = sum_k du[(t+1-k)/s] * w[k], dw[k] += du*z,
//           dx += LN_a'(dz) summed over the branches that share the input; the per-channel parameter partials
//           of a workgroup go to a workspace row that avdf_reduce_segments folds (no float atomics).
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int kMaxSeg = 16;

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
__device__ __forceinline__ void load_taps(const float* __restrict__ w, int c, float4 tap[3]) {
  const float4 a = ld4(w + c * 3), b = ld4(w + c * 3 + 4), d = ld4(w + c * 3 + 8);
  tap[0] = make_float4(a.x, a.w, b.z, d.y);
  tap[1] = make_float4(a.y, b.x, b.w, d.z);
  tap[2] = make_float4(a.z, b.y, d.x, d.w);
}

struct ChainFwdArgs {
  const float* x[3]; const float* ga[3]; const float* ba[3]; const float* w[3]; const float* gb[3]; const float* bb[3];
  float* u[3]; float* y[3]; float* sa[3]; float* sb[3];
  // optional nearest resampling of a branch's INPUT, folded into the row address: xsh > 0: row t reads source row t >> xsh
  // (F.interpolate nearest, scale 2^xsh); xsh < 0: row t reads source row t << -xsh (stride-2^-xsh subsampling); xT = rows
  // per clip of the source.  Replaces the separate up/down-sampling launch in front of the k / v chains of the
  // cross-level blocks (backbones.py:487-490) and the resampled copy of the tensor.
  int xsh[3]; int xT[3];
};
__device__ __forceinline__ int map_row(int t, int sh) { return sh > 0 ? (t >> sh) : (sh < 0 ? (t << (-sh)) : t); }

// grid (B * tiles, 3); TO output tokens per workgroup, NW waves
template <int S, int TO, int NW>
__global__ __launch_bounds__(64 * NW) void chain_fwd_kernel(ChainFwdArgs a, const float* __restrict__ mask, int B, int Tin,
                                                            int Tout, int C, float eps, int tiles) {
  constexpr int RZ = (TO - 1) * S + 3;
  extern __shared__ __attribute__((aligned(16))) float zbuf[];      // [RZ][C]
  const int br = blockIdx.y;
  const int b = blockIdx.x / tiles, to0 = (blockIdx.x % tiles) * TO;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c = lane * 4;
  const bool act = c < C;
  const float invC = 1.0f / (float)C;
  const float* __restrict__ x = a.x[br];
  const int ti_lo = to0 * S - 1;
  {
    float4 g = f4(1.f), bt = f4(0.f);
    if (act && a.ga[br]) { g = ld4(a.ga[br] + c); bt = ld4(a.ba[br] + c); }
    float* mean_o = a.sa[br];
    float* rstd_o = a.sa[br] + (size_t)B * Tin;
#pragma unroll
    for (int r0 = 0; r0 < RZ; r0 += NW) {
      const int r = r0 + wid;
      if (r >= RZ) break;
      const int t = ti_lo + r;
      float4 z = f4(0.f);
      if (t >= 0 && t < Tin) {
        const size_t row = (size_t)b * Tin + t;
        const size_t xrow = (size_t)b * a.xT[br] + map_row(t, a.xsh[br]);
        const float4 v = act ? ld4(x + xrow * C + c) : f4(0.f);
        const float mean = wave_sum(hsum4(v)) * invC;
        const float4 d = act ? sub4(v, f4(mean)) : f4(0.f);
        const float rstd = 1.0f / sqrtf(wave_sum(hsum4(mul4(d, d))) * invC + eps);
        z = fma4(scl4(d, rstd), g, bt);
        if (lane == 0 && t >= to0 * S && t < (to0 + TO) * S) { mean_o[row] = mean; rstd_o[row] = rstd; }
      }
      if (act) st4(zbuf + r * C + c, z);
    }
  }
  __syncthreads();
  float4 tap[3] = {f4(0.f), f4(0.f), f4(0.f)};
  float4 g = f4(1.f), bt = f4(0.f);
  if (act) {
    load_taps(a.w[br], c, tap);
    if (a.gb[br]) { g = ld4(a.gb[br] + c); bt = ld4(a.bb[br] + c); }
  }
  float* mean_o = a.sb[br];
  float* rstd_o = a.sb[br] + (size_t)B * Tout;
#pragma unroll
  for (int i0 = 0; i0 < TO; i0 += NW) {
    const int i = i0 + wid;
    const int t_o = to0 + i;
    if (i >= TO || t_o >= Tout) break;
    const size_t row = (size_t)b * Tout + t_o;
    float4 u = f4(0.f);
    if (act) {
      const float* zr = zbuf + (i * S) * C + c;
      u = mul4(ld4(zr), tap[0]);
      u = add4(u, mul4(ld4(zr + C), tap[1]));
      u = add4(u, mul4(ld4(zr + 2 * C), tap[2]));
      if (mask) u = scl4(u, mask[row]);
      st4(a.u[br] + row * C + c, u);
    }
    const float mean = wave_sum(hsum4(u)) * invC;
    const float4 d = act ? sub4(u, f4(mean)) : f4(0.f);
    const float rstd = 1.0f / sqrtf(wave_sum(hsum4(mul4(d, d))) * invC + eps);
    if (act) st4(a.y[br] + row * C + c, fma4(scl4(d, rstd), g, bt));
    if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
  }
}

struct ChainBwdArgs {
  const float* dy[3]; const float* u[3]; const float* sb[3]; const float* x[3]; const float* sa[3];
  const float* ga[3]; const float* ba[3]; const float* w[3]; const float* gb[3];
  float* dx[3];
  const float* dres[3];   // optional: gradient arriving at the same input through the residual path, added into dx
  int xsh[3]; int xT[3];  // input row map of the forward pass (see ChainFwdArgs)
  // fold[br] (up-sampled inputs, stride 1, 2^xsh dividing the tile): dx[br] is the gradient of the SOURCE rows
  // ([B, xT, C]) -- the workgroup sums each group of 2^xsh consecutive rows of its tile through LDS instead of storing
  // them for a separate reduction launch; otherwise dx[br] is at the chain's own resolution
  int fold[3];
};

// ws row of a workgroup: [3 branches][7*C]: dgamma_a, dbeta_a, dw (c*3+k order, 3C), dgamma_b, dbeta_b
template <int S, int TI, int NW>
__global__ __launch_bounds__(64 * NW) void chain_bwd_kernel(ChainBwdArgs a, const float* __restrict__ mask,
                                                            float* __restrict__ ws, int B, int Tin, int Tout, int C,
                                                            int tiles, int store_mask) {
  constexpr int RU = TI / S + 2;            // du rows a tile can touch
  constexpr int NR1 = (RU + NW - 1) / NW;   // phase-1 rows per wave
  constexpr int NR2 = TI / NW;              // phase-2 rows per wave
  static_assert(TI % NW == 0 && TI % S == 0, "tile must split evenly");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dubuf = smem;                       // [RU][C]
  float* red = smem + RU * C;                // [NW][7*C]
  const int b = blockIdx.x / tiles, ti0 = (blockIdx.x % tiles) * TI;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c = lane * 4;
  const bool act = c < C;
  const float invC = 1.0f / (float)C;
  const int to_lo = ti0 == 0 ? 0 : (ti0 - 1 + S - 1) / S;
  int to_hi = (ti0 + TI) / S;
  if (to_hi > Tout - 1) to_hi = Tout - 1;
  const size_t BTo = (size_t)B * Tout, BTi = (size_t)B * Tin;

  float4 dxacc[NR2];
#pragma unroll
  for (int i = 0; i < NR2; ++i) dxacc[i] = f4(0.f);

  for (int br = 0; br < 3; ++br) {
    // ---- phase 1: du = LN_b'(dy) * mask for the output rows this tile's inputs feed ----------------
    float4 dgb = f4(0.f), dbb = f4(0.f);
    {
      const float4 g = (act && a.gb[br]) ? ld4(a.gb[br] + c) : f4(1.f);
      float4 dv[NR1], uv[NR1];
      float mn[NR1], rs[NR1], mk[NR1];
#pragma unroll
      for (int j = 0; j < NR1; ++j) {
        const int t_o = to_lo + wid + j * NW;
        dv[j] = f4(0.f); uv[j] = f4(0.f); mn[j] = 0.f; rs[j] = 0.f; mk[j] = 0.f;
        if (t_o <= to_hi) {
          const size_t row = (size_t)b * Tout + t_o;
          if (act) { dv[j] = ld4(a.dy[br] + row * C + c); uv[j] = ld4(a.u[br] + row * C + c); }
          mn[j] = a.sb[br][row]; rs[j] = a.sb[br][BTo + row];
          mk[j] = mask ? mask[row] : 1.f;
        }
      }
#pragma unroll
      for (int j = 0; j < NR1; ++j) {
        const int r = wid + j * NW;
        const int t_o = to_lo + r;
        if (t_o > to_hi) continue;
        const float4 h = act ? scl4(sub4(uv[j], f4(mn[j])), rs[j]) : f4(0.f);
        if (t_o * S >= ti0 && t_o * S < ti0 + TI) { dgb = fma4(dv[j], h, dgb); dbb = add4(dbb, dv[j]); }
        const float4 gy = mul4(dv[j], g);
        const float m1 = wave_sum(hsum4(gy)) * invC;
        const float m2 = wave_sum(hsum4(mul4(gy, h))) * invC;
        float4 du = scl4(sub4(sub4(gy, f4(m1)), scl4(h, m2)), rs[j] * mk[j]);
        if (act) st4(dubuf + r * C + c, du);
      }
    }
    __syncthreads();
    // ---- phase 2: per input row: dz, dw, LN_a' ----------------------------------------------------------
    float4 dga = f4(0.f), dba = f4(0.f), dw0 = f4(0.f), dw1 = f4(0.f), dw2 = f4(0.f);
    {
      float4 g = f4(1.f), bt = f4(0.f), tap[3] = {f4(0.f), f4(0.f), f4(0.f)};
      if (act) {
        if (a.ga[br]) { g = ld4(a.ga[br] + c); bt = ld4(a.ba[br] + c); }
        load_taps(a.w[br], c, tap);
      }
      float4 xv[NR2];
      float mn[NR2], rs[NR2];
#pragma unroll
      for (int i = 0; i < NR2; ++i) {
        const int t = ti0 + wid + i * NW;
        xv[i] = f4(0.f); mn[i] = 0.f; rs[i] = 0.f;
        if (t < Tin) {
          const size_t row = (size_t)b * Tin + t;
          if (act) xv[i] = ld4(a.x[br] + ((size_t)b * a.xT[br] + map_row(t, a.xsh[br])) * C + c);
          mn[i] = a.sa[br][row]; rs[i] = a.sa[br][BTi + row];
        }
      }
#pragma unroll
      for (int i = 0; i < NR2; ++i) {
        const int t = ti0 + wid + i * NW;
        if (t >= Tin) continue;
        const float4 h = act ? scl4(sub4(xv[i], f4(mn[i])), rs[i]) : f4(0.f);
        const float4 z = fma4(h, g, bt);
        float4 dz = f4(0.f);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int num = t + 1 - k;
          if (num < 0 || (num % S) != 0) continue;
          const int t_o = num / S;
          if (t_o > to_hi || t_o < to_lo) continue;
          const float4 du = act ? ld4(dubuf + (t_o - to_lo) * C + c) : f4(0.f);
          dz = fma4(du, tap[k], dz);
          if (k == 0) dw0 = fma4(du, z, dw0);
          else if (k == 1) dw1 = fma4(du, z, dw1);
          else dw2 = fma4(du, z, dw2);
        }
        dga = fma4(dz, h, dga);
        dba = add4(dba, dz);
        const float4 gz = mul4(dz, g);
        const float m1 = wave_sum(hsum4(gz)) * invC;
        const float m2 = wave_sum(hsum4(mul4(gz, h))) * invC;
        dxacc[i] = add4(dxacc[i], scl4(sub4(sub4(gz, f4(m1)), scl4(h, m2)), rs[i]));
      }
    }
    // ---- fold the waves' parameter partials, emit this workgroup's workspace row ------------------------
    if (act) {
      float* rw = red + (size_t)wid * 7 * C;
      st4(rw + c, dga);
      st4(rw + C + c, dba);
      float* d = rw + 2 * C + c * 3;
      d[0] = dw0.x; d[1] = dw1.x; d[2] = dw2.x; d[3] = dw0.y; d[4] = dw1.y; d[5] = dw2.y;
      d[6] = dw0.z; d[7] = dw1.z; d[8] = dw2.z; d[9] = dw0.w; d[10] = dw1.w; d[11] = dw2.w;
      st4(rw + 5 * C + c, dgb);
      st4(rw + 6 * C + c, dbb);
    }
    __syncthreads();
    {
      float* wrow = ws + ((size_t)blockIdx.x * 3 + br) * 7 * C;
      for (int e = threadIdx.x; e < 7 * C; e += 64 * NW) {
        float s = 0.f;
#pragma unroll
        for (int w_ = 0; w_ < NW; ++w_) s += red[(size_t)w_ * 7 * C + e];
        wrow[e] = s;
      }
    }
    if (((store_mask >> br) & 1) && a.fold[br]) {
      // (dubuf is free: every wave is past phase 2 -- the barrier in front of the workspace-row write)
      const int sh = a.xsh[br];
#pragma unroll
      for (int i = 0; i < NR2; ++i) {
        const int r = wid + i * NW;
        if (act) st4(dubuf + r * C + c, (ti0 + r < Tin) ? dxacc[i] : f4(0.f));
        dxacc[i] = f4(0.f);
      }
      __syncthreads();
      const int grp = 1 << sh;
      for (int j = wid; j < (TI >> sh); j += NW) {
        const int ts = (ti0 >> sh) + j;
        if (ts < a.xT[br] && act) {
          float4 sum = f4(0.f);
          for (int q = 0; q < grp; ++q) sum = add4(sum, ld4(dubuf + (j * grp + q) * C + c));
          st4(a.dx[br] + ((size_t)b * a.xT[br] + ts) * C + c, sum);
        }
      }
    } else if ((store_mask >> br) & 1) {
#pragma unroll
      for (int i = 0; i < NR2; ++i) {
        const int t = ti0 + wid + i * NW;
        if (t < Tin && act) {
          const size_t o = ((size_t)b * Tin + t) * C + c;
          st4(a.dx[br] + o, a.dres[br] ? add4(dxacc[i], ld4(a.dres[br] + o)) : dxacc[i]);
        }
        dxacc[i] = f4(0.f);
      }
    }
    __syncthreads();
  }
}

struct SegArgs {
  float* dst[kMaxSeg];
  int len[kMaxSeg];
  int nsrc[kMaxSeg];
  int src[kMaxSeg][3];
  int first[kMaxSeg + 1];       // prefix sum of len
};

// dst[s][i] (+)= sum_{rows r} sum_{j < nsrc[s]} ws[r*stride + src[s][j] + i]
__global__ __launch_bounds__(1024) void reduce_segments_kernel(const float* __restrict__ ws, int nrow, int stride, SegArgs sa,
                                                               int nseg, int accumulate) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  const int total = sa.first[nseg];
  int s = 0;
  float acc = 0.f;
  if (col < total) {
    while (col >= sa.first[s + 1]) ++s;
    const int i = col - sa.first[s];
    // gridDim.y > 1 (accumulating launches only): the rows are split over blockIdx.y, partial sums meet in dst by atomics
    const int chunk = (nrow + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * chunk, r1 = min(nrow, r0 + chunk);
    for (int j = 0; j < sa.nsrc[s]; ++j) {
      const float* p = ws + sa.src[s][j] + i;
      for (int r = r0 + ph; r < r1; r += 16) acc += p[(size_t)r * stride];
    }
  }
  red[ph][cl] = acc;
  __syncthreads();
  if (ph == 0 && col < total) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][cl];
    float* o = sa.dst[s] + (col - sa.first[s]);
    if (gridDim.y > 1) atomicAdd(o, t);
    else *o = accumulate ? *o + t : t;
  }
}

// Several reduce_segments jobs in ONE launch (the engine path flushes the parameter partials of all chain backward
// launches of a backward phase together: a dependent tiny launch costs 2.4 us).  Compact per-job
// descriptors so that kBatchJobs of them fit the kernel-argument segment.
constexpr int kBatchJobs = 10;
struct SegJob {
  const float* ws;
  float* dst[kMaxSeg];
  int nrow, stride;
  unsigned short first[kMaxSeg + 1];     // prefix sum of the segment lengths (<= row stride <= 65535)
  unsigned short src[kMaxSeg][3];
  unsigned char nsrc[kMaxSeg];
  unsigned char nseg;
};
struct SegBatch {
  SegJob job[kBatchJobs];
  int cb_first[kBatchJobs + 1];          // prefix sum of the jobs' 64-column blocks
  int njobs;
};
// accumulating only: dst[s][i] += ...; the rows of a job are split over blockIdx.y, partial sums meet in dst by atomics.
// A workgroup folds 64 columns: a lane owns one float4 column group and one of four row phases of its wave, four rows in
// flight per lane (the one-float-per-lane, one-row-at-a-time walk of reduce_segments_kernel keeps 256 bytes in flight per
// wave and reads at ~0.8 TB/s).  Segment starts, lengths and source offsets are multiples of 4 floats (host-checked).
__global__ __launch_bounds__(1024) void reduce_segments_batch_kernel(SegBatch bt) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6, cg = lane & 15, rs = lane >> 4;
  int j = 0;
  while (j + 1 < bt.njobs && (int)blockIdx.x >= bt.cb_first[j + 1]) ++j;
  const SegJob& jb = bt.job[j];
  const int col = ((int)blockIdx.x - bt.cb_first[j]) * 64 + cg * 4;
  const int total = jb.first[jb.nseg];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < total) {
    int s = 0;
    while (col >= jb.first[s + 1]) ++s;
    const int i = col - jb.first[s];
    const int chunk = (jb.nrow + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * chunk, r1 = min(jb.nrow, r0 + chunk);
    const size_t stride = (size_t)jb.stride;
    for (int q = 0; q < jb.nsrc[s]; ++q) {
      const float* p = jb.ws + jb.src[s][q] + i;
      int r = r0 + ph * 4 + rs;
      for (; r + 192 < r1; r += 256) {                // rows r, r+64, r+128, r+192 of this lane: four loads in flight
        const float4 v0 = ld4(p + r * stride), v1 = ld4(p + (r + 64) * stride), v2 = ld4(p + (r + 128) * stride),
                     v3 = ld4(p + (r + 192) * stride);
        acc = add4(acc, add4(add4(v0, v1), add4(v2, v3)));
      }
      for (; r < r1; r += 64) acc = add4(acc, ld4(p + r * stride));
    }
  }
  // the four row phases of a wave meet by shuffles, the sixteen waves through LDS
  acc.x += __shfl_xor(acc.x, 16, 64); acc.y += __shfl_xor(acc.y, 16, 64); acc.z += __shfl_xor(acc.z, 16, 64); acc.w += __shfl_xor(acc.w, 16, 64);
  acc.x += __shfl_xor(acc.x, 32, 64); acc.y += __shfl_xor(acc.y, 32, 64); acc.z += __shfl_xor(acc.z, 32, 64); acc.w += __shfl_xor(acc.w, 32, 64);
  if (rs == 0) { red[ph][cg * 4] = acc.x; red[ph][cg * 4 + 1] = acc.y; red[ph][cg * 4 + 2] = acc.z; red[ph][cg * 4 + 3] = acc.w; }
  __syncthreads();
  const int c1 = ((int)blockIdx.x - bt.cb_first[j]) * 64 + lane;
  if (ph == 0 && c1 < total) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][lane];
    int s = 0;
    while (c1 >= jb.first[s + 1]) ++s;
    atomicAdd(jb.dst[s] + (c1 - jb.first[s]), t);
  }
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
constexpr int kTile = 32, kWaves = 8;
// backward: 16-token tiles -> twice the workgroups, 76 KB of LDS each (2 per CU): the three branches of a tile are
// processed one after the other, so co-resident workgroups are what hides the load latency
constexpr int kWavesB = 8;
// tile sizes are host-side launch parameters (AVDF_CHAIN_TILE_F in {16, 32}, AVDF_CHAIN_TILE_B in {8, 16, 32}: in-situ A/B)
static const int g_tile_f = [] { const char* e = getenv("AVDF_CHAIN_TILE_F"); const int v = e ? atoi(e) : kTile; return v == 16 ? 16 : 32; }();
static const int g_tile_b = [] { const char* e = getenv("AVDF_CHAIN_TILE_B"); const int v = e ? atoi(e) : 16; return (v == 8 || v == 32 || v == 12) ? v : 16; }();

}  // namespace

static int set_row_map(const char* who, const int* xshift, int Tin, int (&xsh)[3], int (&xT)[3]) {
  for (int i = 0; i < 3; ++i) {
    const int sh = xshift ? xshift[i] : 0;
    AVDF_REQUIRE(sh >= -6 && sh <= 6, "%s: input shift %d out of range", who, sh);
    AVDF_REQUIRE(sh <= 0 || Tin % (1 << sh) == 0, "%s: Tin=%d is not a multiple of the up-sampling factor %d", who, Tin, 1 << sh);
    xsh[i] = sh;
    xT[i] = sh > 0 ? (Tin >> sh) : (sh < 0 ? (Tin << (-sh)) : Tin);
  }
  return AVDF_OK;
}

static int chain_fwd_impl(const float* const* x, const float* const* gamma_a, const float* const* beta_a,
                          const float* const* w, const float* const* gamma_b, const float* const* beta_b,
                          const float* mask, float* const* u, float* const* y, float* const* stats_a,
                          float* const* stats_b, const int* xshift, int B, int Tin, int Tout, int C, int stride, float eps,
                          void* stream);

extern "C" int avdf_chain_fwd(const float* const* x, const float* const* gamma_a, const float* const* beta_a,
                              const float* const* w, const float* const* gamma_b, const float* const* beta_b,
                              const float* mask, float* const* u, float* const* y, float* const* stats_a,
                              float* const* stats_b, int B, int Tin, int Tout, int C, int stride, float eps,
                              void* stream) {
  return chain_fwd_impl(x, gamma_a, beta_a, w, gamma_b, beta_b, mask, u, y, stats_a, stats_b, nullptr, B, Tin, Tout, C, stride,
                        eps, stream);
}

// avdf_chain_fwd with a nearest resampling of each branch's input folded into the row address: xshift[i] > 0: branch i reads
// x[i] ([B, Tin >> s, C]) as if it were up-sampled by 2^s; < 0: reads every 2^-s-th row of x[i] ([B, Tin << -s, C]); 0: as is.
extern "C" int avdf_chain_fwd_map(const float* const* x, const float* const* gamma_a, const float* const* beta_a,
                                  const float* const* w, const float* const* gamma_b, const float* const* beta_b,
                                  const float* mask, float* const* u, float* const* y, float* const* stats_a,
                                  float* const* stats_b, const int* xshift, int B, int Tin, int Tout, int C, int stride,
                                  float eps, void* stream) {
  AVDF_REQUIRE(xshift, "chain_fwd_map: null shift table");
  return chain_fwd_impl(x, gamma_a, beta_a, w, gamma_b, beta_b, mask, u, y, stats_a, stats_b, xshift, B, Tin, Tout, C, stride,
                        eps, stream);
}

static int chain_fwd_impl(const float* const* x, const float* const* gamma_a, const float* const* beta_a,
                          const float* const* w, const float* const* gamma_b, const float* const* beta_b,
                          const float* mask, float* const* u, float* const* y, float* const* stats_a,
                          float* const* stats_b, const int* xshift, int B, int Tin, int Tout, int C, int stride, float eps,
                          void* stream) {
  AVDF_REQUIRE(x && w && u && y && stats_a && stats_b && gamma_a && beta_a && gamma_b && beta_b, "chain_fwd: null table");
  AVDF_REQUIRE(B > 0 && Tin > 0 && C > 0 && C % 4 == 0 && C <= 256, "chain_fwd: C must be a multiple of 4, <= 256 (%d)", C);
  AVDF_REQUIRE(stride == 1 || stride == 2, "chain_fwd: stride must be 1 or 2");
  AVDF_REQUIRE(Tout == (Tin - 1) / stride + 1, "chain_fwd: Tout=%d inconsistent with Tin=%d stride=%d", Tout, Tin, stride);
  ChainFwdArgs a;
  for (int i = 0; i < 3; ++i) {
    AVDF_REQUIRE(x[i] && w[i] && u[i] && y[i] && stats_a[i] && stats_b[i], "chain_fwd: null branch pointer");
    AVDF_REQUIRE((gamma_a[i] == nullptr) == (beta_a[i] == nullptr) && (gamma_b[i] == nullptr) == (beta_b[i] == nullptr),
                 "chain_fwd: gamma/beta must come in pairs");
    AVDF_REQUIRE(aligned16(x[i]) && aligned16(w[i]) && aligned16(u[i]) && aligned16(y[i]) && aligned16(gamma_a[i]) &&
                     aligned16(beta_a[i]) && aligned16(gamma_b[i]) && aligned16(beta_b[i]),
                 "chain_fwd: 16-byte alignment");
    a.x[i] = x[i]; a.ga[i] = gamma_a[i]; a.ba[i] = beta_a[i]; a.w[i] = w[i]; a.gb[i] = gamma_b[i]; a.bb[i] = beta_b[i];
    a.u[i] = u[i]; a.y[i] = y[i]; a.sa[i] = stats_a[i]; a.sb[i] = stats_b[i];
  }
  { const int rc = set_row_map("chain_fwd", xshift, Tin, a.xsh, a.xT); if (rc) return rc; }
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_CF(S_, TO_)                                                                                          \
  do {                                                                                                              \
    const int tiles = cdiv(Tout, TO_);                                                                              \
    hipLaunchKernelGGL((chain_fwd_kernel<S_, TO_, kWaves>), dim3(B * tiles, 3), dim3(64 * kWaves),                  \
                       (size_t)((TO_ - 1) * S_ + 3) * C * sizeof(float), st, a, mask, B, Tin, Tout, C, eps, tiles); \
  } while (0)
  if (stride == 1) { if (g_tile_f == 16) LAUNCH_CF(1, 16); else LAUNCH_CF(1, 32); }
  else { if (g_tile_f == 16) LAUNCH_CF(2, 8); else LAUNCH_CF(2, 16); }
#undef LAUNCH_CF
  AVDF_CHECK_LAUNCH("chain_fwd");
  return AVDF_OK;
}

extern "C" int64_t avdf_chain_bwd_ws(int B, int Tin, int C) { return (int64_t)B * cdiv(Tin, g_tile_b) * 21 * C; }
extern "C" int avdf_chain_bwd_rows(int B, int Tin) { return B * cdiv(Tin, g_tile_b); }

extern "C" int avdf_chain_bwd(const float* const* dy, const float* const* u, const float* const* stats_b,
                              const float* const* x, const float* const* stats_a, const float* const* gamma_a,
                              const float* const* beta_a, const float* const* w, const float* const* gamma_b,
                              const float* mask, float* const* dx, int store_mask, float* ws, int B, int Tin, int Tout,
                              int C, int stride, void* stream) {
  return avdf_chain_bwd_res(dy, u, stats_b, x, stats_a, gamma_a, beta_a, w, gamma_b, mask, dx, nullptr, store_mask, ws, B, Tin,
                            Tout, C, stride, stream);
}

// dres[i] (table or entries may be NULL): [B,Tin,C] gradient that reaches the input closed by branch i through another
// path (the block's residual connection); dx[i] = LN_a backward sum + dres[i], saving autograd's separate add kernel.
static int chain_bwd_impl(const float* const* dy, const float* const* u, const float* const* stats_b,
                          const float* const* x, const float* const* stats_a, const float* const* gamma_a,
                          const float* const* beta_a, const float* const* w, const float* const* gamma_b,
                          const float* mask, float* const* dx, const float* const* dres, int store_mask, float* ws,
                          const int* xshift, int B, int Tin, int Tout, int C, int stride, void* stream);

// 1 if avdf_chain_bwd_map returns dx of a branch with input shift `xshift` at the SOURCE resolution ([B, Tin >> xshift, C]:
// the row groups are summed inside the kernel), 0 if at the chain's resolution (the caller folds it back)
extern "C" int avdf_chain_bwd_folds(int xshift, int stride) {
  return (xshift > 0 && stride == 1 && (1 << xshift) <= g_tile_b && g_tile_b % (1 << xshift) == 0) ? 1 : 0;
}

extern "C" int avdf_chain_bwd_res(const float* const* dy, const float* const* u, const float* const* stats_b,
                                  const float* const* x, const float* const* stats_a, const float* const* gamma_a,
                                  const float* const* beta_a, const float* const* w, const float* const* gamma_b,
                                  const float* mask, float* const* dx, const float* const* dres, int store_mask, float* ws,
                                  int B, int Tin, int Tout, int C, int stride, void* stream) {
  return chain_bwd_impl(dy, u, stats_b, x, stats_a, gamma_a, beta_a, w, gamma_b, mask, dx, dres, store_mask, ws, nullptr, B, Tin,
                        Tout, C, stride, stream);
}

// backward of avdf_chain_fwd_map: x[i] is the un-resampled source, dx[i] is the gradient at the CHAIN's resolution
// ([B, Tin, C]); the caller folds it back to the source rows (avdf_upsample_bwd / avdf_downsample_bwd)
extern "C" int avdf_chain_bwd_map(const float* const* dy, const float* const* u, const float* const* stats_b,
                                  const float* const* x, const float* const* stats_a, const float* const* gamma_a,
                                  const float* const* beta_a, const float* const* w, const float* const* gamma_b,
                                  const float* mask, float* const* dx, const float* const* dres, int store_mask, float* ws,
                                  const int* xshift, int B, int Tin, int Tout, int C, int stride, void* stream) {
  AVDF_REQUIRE(xshift, "chain_bwd_map: null shift table");
  return chain_bwd_impl(dy, u, stats_b, x, stats_a, gamma_a, beta_a, w, gamma_b, mask, dx, dres, store_mask, ws, xshift, B, Tin,
                        Tout, C, stride, stream);
}

static int chain_bwd_impl(const float* const* dy, const float* const* u, const float* const* stats_b,
                          const float* const* x, const float* const* stats_a, const float* const* gamma_a,
                          const float* const* beta_a, const float* const* w, const float* const* gamma_b,
                          const float* mask, float* const* dx, const float* const* dres, int store_mask, float* ws,
                          const int* xshift, int B, int Tin, int Tout, int C, int stride, void* stream) {
  AVDF_REQUIRE(dy && u && stats_b && x && stats_a && gamma_a && beta_a && w && gamma_b && dx && ws, "chain_bwd: null table");
  AVDF_REQUIRE(B > 0 && Tin > 0 && C > 0 && C % 4 == 0 && C <= 256, "chain_bwd: C must be a multiple of 4, <= 256 (%d)", C);
  AVDF_REQUIRE(stride == 1 || stride == 2, "chain_bwd: stride must be 1 or 2");
  AVDF_REQUIRE(Tout == (Tin - 1) / stride + 1, "chain_bwd: Tout inconsistent");
  AVDF_REQUIRE((store_mask & 4) && store_mask > 0 && store_mask < 8, "chain_bwd: the last branch must close an input group");
  ChainBwdArgs a;
  for (int i = 0; i < 3; ++i) {
    AVDF_REQUIRE(dy[i] && u[i] && stats_b[i] && x[i] && stats_a[i] && w[i], "chain_bwd: null branch pointer");
    AVDF_REQUIRE(!((store_mask >> i) & 1) || dx[i], "chain_bwd: dx missing for a closing branch");
    AVDF_REQUIRE(aligned16(dy[i]) && aligned16(u[i]) && aligned16(x[i]) && aligned16(w[i]) && aligned16(gamma_a[i]) &&
                     aligned16(beta_a[i]) && aligned16(gamma_b[i]) && aligned16(dx[i]),
                 "chain_bwd: 16-byte alignment");
    a.dy[i] = dy[i]; a.u[i] = u[i]; a.sb[i] = stats_b[i]; a.x[i] = x[i]; a.sa[i] = stats_a[i];
    a.ga[i] = gamma_a[i]; a.ba[i] = beta_a[i]; a.w[i] = w[i]; a.gb[i] = gamma_b[i]; a.dx[i] = dx[i];
    a.dres[i] = dres ? dres[i] : nullptr;
    AVDF_REQUIRE(aligned16(a.dres[i]) && (!a.dres[i] || ((store_mask >> i) & 1)), "chain_bwd: dres must be 16-byte aligned and belong to a closing branch");
  }
  { const int rc = set_row_map("chain_bwd", xshift, Tin, a.xsh, a.xT); if (rc) return rc; }
  for (int i = 0; i < 3; ++i) {
    a.fold[i] = avdf_chain_bwd_folds(a.xsh[i], stride);
    AVDF_REQUIRE(!a.fold[i] || !a.dres[i], "chain_bwd: a residual gradient cannot join a folded (up-sampled) input");
  }
  hipStream_t st = (hipStream_t)stream;
  const int tiles = cdiv(Tin, g_tile_b);
#define LAUNCH_CB2(S_, TI_, NW_)                                                                                     \
  do {                                                                                                                \
    constexpr int RU = TI_ / S_ + 2;                                                                                  \
    hipLaunchKernelGGL((chain_bwd_kernel<S_, TI_, NW_>), dim3(B * tiles), dim3(64 * NW_),                             \
                       (size_t)(RU + NW_ * 7) * C * sizeof(float), st, a, mask, ws, B, Tin, Tout, C, tiles, store_mask); \
  } while (0)
#define LAUNCH_CB(S_, TI_) LAUNCH_CB2(S_, TI_, kWavesB)
  // (12-row tiles, 6 waves: 8 x 768 / 12 = 512 workgroups = exactly two per CU at the top pyramid level)
  if (stride == 1) { if (g_tile_b == 8) LAUNCH_CB(1, 8); else if (g_tile_b == 32) LAUNCH_CB(1, 32); else if (g_tile_b == 12) LAUNCH_CB2(1, 12, 6); else LAUNCH_CB(1, 16); }
  else { if (g_tile_b == 8) LAUNCH_CB(2, 8); else if (g_tile_b == 32) LAUNCH_CB(2, 32); else if (g_tile_b == 12) LAUNCH_CB2(2, 12, 6); else LAUNCH_CB(2, 16); }
#undef LAUNCH_CB2
#undef LAUNCH_CB
  AVDF_CHECK_LAUNCH("chain_bwd");
  return AVDF_OK;
}

extern "C" int avdf_reduce_segments(const float* ws, int nrow, int row_stride, int nseg, float* const* dst, const int* len,
                                    const int* nsrc, const int* src_off /* [nseg][3] */, int accumulate, void* stream) {
  AVDF_REQUIRE(ws && dst && len && nsrc && src_off && nrow > 0 && row_stride > 0, "reduce_segments: bad args");
  AVDF_REQUIRE(nseg > 0 && nseg <= kMaxSeg, "reduce_segments: 1..%d segments", kMaxSeg);
  SegArgs sa;
  sa.first[0] = 0;
  for (int s = 0; s < nseg; ++s) {
    AVDF_REQUIRE(dst[s] && len[s] > 0 && nsrc[s] >= 1 && nsrc[s] <= 3, "reduce_segments: bad segment %d", s);
    sa.dst[s] = dst[s]; sa.len[s] = len[s]; sa.nsrc[s] = nsrc[s];
    for (int j = 0; j < 3; ++j) {
      sa.src[s][j] = src_off[s * 3 + j];
      AVDF_REQUIRE(j >= nsrc[s] || (sa.src[s][j] >= 0 && sa.src[s][j] + len[s] <= row_stride), "reduce_segments: source out of the row");
    }
    sa.first[s + 1] = sa.first[s] + len[s];
  }
  for (int s = nseg; s < kMaxSeg; ++s) { sa.dst[s] = nullptr; sa.len[s] = 0; sa.nsrc[s] = 0; sa.first[s + 1] = sa.first[nseg]; }
  // 68 column blocks cover a quarter of the chip: accumulating launches (the engine path) also split the rows 4 ways
  const int ysplit = (accumulate && nrow >= 128) ? 4 : 1;
  hipLaunchKernelGGL(reduce_segments_kernel, dim3(cdiv(sa.first[nseg], 64), ysplit), dim3(1024), 0, (hipStream_t)stream, ws, nrow,
                     row_stride, sa, nseg, accumulate);
  AVDF_CHECK_LAUNCH("reduce_segments");
  return AVDF_OK;
}

// njobs accumulating reduce_segments jobs in ceil(njobs / 10) launches.  Job j: workspace ws[j] of nrow[j] rows x stride[j]
// floats, nseg[j] segments; the segment tables of all jobs are concatenated: dst / len / nsrc one entry per segment,
// src_off three.  Every destination is ADDED to (float atomics; AVDF_SEGFOLD_Y row slices per job, default 1).
extern "C" int avdf_reduce_segments_batched(int njobs, const float* const* ws, const int* nrow, const int* stride, const int* nseg,
                                            float* const* dst, const int* len, const int* nsrc, const int* src_off, void* stream) {
  AVDF_REQUIRE(njobs > 0 && ws && nrow && stride && nseg && dst && len && nsrc && src_off, "reduce_segments_batched: bad args");
  static_assert(sizeof(SegBatch) <= 4000, "the job descriptors must fit the kernel-argument segment");
  int seg0 = 0;
  for (int j0 = 0; j0 < njobs; j0 += kBatchJobs) {
    SegBatch bt;
    memset(&bt, 0, sizeof(bt));
    bt.njobs = njobs - j0 < kBatchJobs ? njobs - j0 : kBatchJobs;
    bt.cb_first[0] = 0;
    for (int q = 0; q < bt.njobs; ++q) {
      const int j = j0 + q;
      SegJob& jb = bt.job[q];
      AVDF_REQUIRE(ws[j] && aligned16(ws[j]) && nrow[j] > 0 && stride[j] > 0 && stride[j] % 4 == 0 && stride[j] <= 65535 && nseg[j] > 0 && nseg[j] <= kMaxSeg,
                   "reduce_segments_batched: bad job %d", j);
      jb.ws = ws[j]; jb.nrow = nrow[j]; jb.stride = stride[j]; jb.nseg = (unsigned char)nseg[j];
      int first = 0;
      for (int s = 0; s < nseg[j]; ++s, ++seg0) {
        AVDF_REQUIRE(dst[seg0] && len[seg0] > 0 && len[seg0] % 4 == 0 && nsrc[seg0] >= 1 && nsrc[seg0] <= 3,
                     "reduce_segments_batched: bad segment %d of job %d (lengths are multiples of 4)", s, j);
        jb.dst[s] = dst[seg0]; jb.nsrc[s] = (unsigned char)nsrc[seg0]; jb.first[s] = (unsigned short)first;
        for (int k = 0; k < 3; ++k) {
          const int o = src_off[seg0 * 3 + k];
          AVDF_REQUIRE(k >= nsrc[seg0] || (o >= 0 && o % 4 == 0 && o + len[seg0] <= stride[j]), "reduce_segments_batched: source out of the row");
          jb.src[s][k] = (unsigned short)(k < nsrc[seg0] ? o : 0);
        }
        first += len[seg0];
        AVDF_REQUIRE(first <= 65535, "reduce_segments_batched: job %d too wide", j);
      }
      for (int s = nseg[j]; s <= kMaxSeg; ++s) jb.first[s] = (unsigned short)first;
      bt.cb_first[q + 1] = bt.cb_first[q] + cdiv(first, 64);
    }
    for (int q = bt.njobs; q < kBatchJobs; ++q) bt.cb_first[q + 1] = bt.cb_first[bt.njobs];
    // rows of a job in ONE slice by default (round 4: 853.5 vs 851.4 clips/s for four slices over two alternations -- the
    // batched launches are already 800+ column blocks -- and a job's partial rows are then summed in a fixed order; the
    // destinations are still added to atomically: several launches of a phase may meet in one parameter gradient)
    static const int ysplit = [] { const char* e = getenv("AVDF_SEGFOLD_Y"); const int v = e ? atoi(e) : 1; return v >= 1 && v <= 8 ? v : 1; }();
    hipLaunchKernelGGL(reduce_segments_batch_kernel, dim3(bt.cb_first[bt.njobs], ysplit), dim3(1024), 0, (hipStream_t)stream, bt);
    AVDF_CHECK_LAUNCH("reduce_segments_batched");
  }
  return AVDF_OK;
}

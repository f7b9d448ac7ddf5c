// Segment decode + class-agnostic hard 1-D NMS + segment voting, one workgroup per clip.
// Replaces inference_single_video (av_fd_no_recon.py:760-825), batched_nms / NMSop / seg_voting
// (libs/utils/nms.py:8-35,67-190) and nms_1d_cpu (libs/utils/csrc/nms_cpu.cpp:19-58), which the
// reference runs on the CPU one clip at a time.
//
// Ordering contract: candidates are ranked by (score desc, flat point index asc).  That equals the
// reference's "per-level descending sort, concatenate levels, descending sort again" whenever its
// (unstable) sorts keep ties in input order.  All IoU arithmetic is fp32 in the reference's
// operation order (areas = x2-x1+1e-6 for NMS, plain lengths for voting) with correctly rounded
// division, so keep/suppress decisions are bit-identical for identical inputs.
#include "common.h"

namespace {

constexpr int kMaxP = 2048;
constexpr int kNT = 1024;      // threads per clip: the sort / suppression / voting loops are barrier-bound, 16 waves cut their trip counts 4x

struct Cand { float score; int idx; };

__device__ __forceinline__ bool before(const Cand& a, const Cand& b) {   // a ranks ahead of b
  return a.score > b.score || (a.score == b.score && a.idx < b.idx);
}

__global__ __launch_bounds__(kNT) void decode_nms_kernel(
    const float* __restrict__ points, const float* __restrict__ logits, const float* __restrict__ offsets,
    const float* __restrict__ pmask, float* __restrict__ out_segs, float* __restrict__ out_scores,
    int32_t* __restrict__ out_pidx, int32_t* __restrict__ out_count, int P,
    float pre_nms_thresh, float duration_thresh, float min_score, float iou_thr, float voting_thr, int max_seg) {
  __shared__ Cand key[kMaxP], key2[kMaxP];      // keys by rank (key2: second buffer of the sort's LDS stages)
  __shared__ float s0[kMaxP], s1[kMaxP];        // segment of flat point p (feature-grid units)
  __shared__ float r0s[kMaxP], r1s[kMaxP];      // segment of rank r
  __shared__ unsigned long long abits[kMaxP / 64];   // survivors of the NMS, one bit per rank
  __shared__ int kept[256];
  __shared__ int n_cand, n_nms;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) { n_cand = 0; n_nms = 0; }
  __syncthreads();

  // Thread (wave w, lane) owns sort positions 128 w + lane and 128 w + 64 + lane: a wave holds 128 consecutive positions in
  // registers, so every exchange at a distance below 64 is a lane shuffle, distance 64 is the thread's own pair, and only
  // the 10 stages at distance >= 128 go through LDS (double-buffered: one barrier each).  66 stages of dependent LDS
  // round trips took ~40 us; this takes ~10.
  const int lane = tid & 63, wid = tid >> 6;
  const int pos0 = wid * 128 + lane;
  Cand c0, c1;
  int lc = 0, ln = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int p = pos0 + h * 64;
    Cand c; c.score = -1.f; c.idx = p;
    if (p < P) {
      const float x = logits[(size_t)b * P + p];
      const float prob = (1.0f / (1.0f + expf(-x))) * pmask[(size_t)b * P + p];
      const float pt = points[p * 4], st = points[p * 4 + 3];
      const float l = pt - __fmul_rn(offsets[((size_t)b * P + p) * 2], st);      // no fma contraction: match the CPU op order
      const float r = pt + __fmul_rn(offsets[((size_t)b * P + p) * 2 + 1], st);
      s0[p] = l; s1[p] = r;
      if (prob > pre_nms_thresh && (r - l) > duration_thresh) {
        c.score = prob; ++lc;
        if (!(min_score > 0.f) || prob > min_score) ++ln;
      }
    } else { s0[p] = 0.f; s1[p] = 0.f; }
    if (h == 0) c0 = c; else c1 = c;
  }
  atomicAdd(&n_cand, lc);
  atomicAdd(&n_nms, ln);

  // bitonic sort of 2048 keys, rank order = `before`
  // one compare-exchange of the key at position i with the key of its partner i ^ j in the stage (k, j)
  auto cmpx = [&](Cand& mine, const Cand& other, int i, int k, int j) {
    const bool first = ((i & j) == 0) == ((i & k) == 0);      // this position keeps the key that ranks ahead
    if (first ? before(other, mine) : before(mine, other)) mine = other;
  };
  int buf = 0;
  for (int k = 2; k <= kMaxP; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 128) {
        Cand* kb = buf ? key2 : key;
        kb[pos0] = c0; kb[pos0 + 64] = c1;
        __syncthreads();
        const Cand o0 = kb[pos0 ^ j], o1 = kb[(pos0 + 64) ^ j];
        cmpx(c0, o0, pos0, k, j);
        cmpx(c1, o1, pos0 + 64, k, j);
        buf ^= 1;                                               // the next LDS stage writes the other buffer: no second barrier
      } else if (j == 64) {
        const Cand a = c0, c = c1;
        cmpx(c0, c, pos0, k, j);
        cmpx(c1, a, pos0 + 64, k, j);
      } else {
        Cand o0, o1;
        o0.score = __shfl_xor(c0.score, j, 64); o0.idx = __shfl_xor(c0.idx, j, 64);
        o1.score = __shfl_xor(c1.score, j, 64); o1.idx = __shfl_xor(c1.idx, j, 64);
        cmpx(c0, o0, pos0, k, j);
        cmpx(c1, o1, pos0 + 64, k, j);
      }
    }
  }
  // ranks -> LDS: the key, and the segment of every rank (the NMS and voting loops read by rank, not through key[].idx)
  __syncthreads();                                              // all s0 / s1 of the decode are written; the last LDS stage is read
  key[pos0] = c0; key[pos0 + 64] = c1;
  r0s[pos0] = s0[c0.idx]; r1s[pos0] = s1[c0.idx];
  r0s[pos0 + 64] = s0[c1.idx]; r1s[pos0 + 64] = s1[c1.idx];
  __syncthreads();

  const int nn = n_nms;      // candidates with score > min_score occupy ranks [0, nn)
  const int nwords = (nn + 63) >> 6;
  if (tid < kMaxP / 64) abits[tid] = tid * 64 + 64 <= nn ? ~0ull : (tid * 64 < nn ? (1ull << (nn - tid * 64)) - 1ull : 0ull);
  __syncthreads();

  // greedy NMS in rank order; only the first max_seg survivors are ever reported.  The survivors are a bit mask in LDS:
  // every thread finds the next one itself (a dependent LDS read per 64 ranks skipped, not per rank), wave w owns words
  // w, w + 16 of the mask -- it tests its 64 candidates against the kept segment and clears their bits with one ballot --
  // and one barrier per kept segment orders the clears before the next search.  (A wave that is still searching while
  // another already clears reads the same answer: bits at or below the segment being kept are never cleared.)
  const int cap = max_seg < 256 ? max_seg : 256;
  int nkeep = 0, i = -1;
  while (true) {
    int w = (i + 1) >> 6;
    unsigned long long m = w < nwords ? abits[w] & (~0ull << ((i + 1) & 63)) : 0ull;
    while (m == 0ull && ++w < nwords) m = abits[w];
    if (m == 0ull) break;
    i = w * 64 + __ffsll((long long)m) - 1;
    if (tid == 0) { kept[nkeep] = i; }
    const float ix1 = r0s[i], ix2 = r1s[i];
    const float ia = ix2 - ix1 + 1e-6f;
    for (int ww = (i >> 6) + ((wid - (i >> 6)) & 15); ww < nwords; ww += kNT / 64) {      // the words >= i / 64 this wave owns
      const int j = ww * 64 + lane;
      bool sup = false;
      if (j > i && ((abits[ww] >> lane) & 1ull)) {
        const float j0 = r0s[j], j1 = r1s[j];
        const float xx1 = fmaxf(ix1, j0), xx2 = fminf(ix2, j1);
        const float inter = fmaxf(0.f, xx2 - xx1);
        const float ja = j1 - j0 + 1e-6f;
        const float ovr = inter / (ia + ja - inter);
        sup = ovr >= iou_thr;
      }
      const unsigned long long clr = __ballot(sup);
      if (lane == 0 && clr) abits[ww] &= ~clr;
    }
    if (++nkeep >= cap) break;
    __syncthreads();
  }
  __syncthreads();
  const int nk = nkeep;
  const int nc = n_cand;     // voting pool = every decoded candidate (ranks [0, nc)), nms.py:176-182

  for (int q = wid; q < nk; q += kNT / 64) {
    const int ri = kept[q];
    const int pi = key[ri].idx;
    float a0 = r0s[ri], a1 = r1s[ri];
    if (voting_thr > 0.f) {
      const float la = a1 - a0;
      float wsum = 0.f;
      float sw[kMaxP / 64];                      // score * iou of this lane's candidates (0 below the voting threshold)
#pragma unroll
      for (int m = 0; m < kMaxP / 64; ++m) {
        const int j = lane + m * 64;
        sw[m] = 0.f;
        if (j < nc) {
          const float inter = fmaxf(fminf(a1, r1s[j]) - fmaxf(a0, r0s[j]), 0.f);
          const float iou = inter / (la + (r1s[j] - r0s[j]) - inter);
          if (iou >= voting_thr) { sw[m] = key[j].score * iou; wsum += sw[m]; }
        }
      }
      wsum = wave_sum(wsum);
      float r0 = 0.f, r1 = 0.f;
#pragma unroll
      for (int m = 0; m < kMaxP / 64; ++m) {
        const int j = lane + m * 64;
        if (j < nc && sw[m] != 0.f) {
          const float wgt = sw[m] / wsum;
          r0 += wgt * r0s[j];
          r1 += wgt * r1s[j];
        }
      }
      a0 = wave_sum(r0);
      a1 = wave_sum(r1);
    }
    if (lane == 0) {
      out_segs[((size_t)b * max_seg + q) * 2] = a0;
      out_segs[((size_t)b * max_seg + q) * 2 + 1] = a1;
      out_scores[(size_t)b * max_seg + q] = key[ri].score;
      out_pidx[(size_t)b * max_seg + q] = pi;
    }
  }
  for (int q = nk + tid; q < max_seg; q += kNT) {
    out_segs[((size_t)b * max_seg + q) * 2] = 0.f;
    out_segs[((size_t)b * max_seg + q) * 2 + 1] = 0.f;
    out_scores[(size_t)b * max_seg + q] = 0.f;
    out_pidx[(size_t)b * max_seg + q] = -1;
  }
  if (tid == 0) out_count[b] = nk;
}

}  // namespace

extern "C" int avdf_decode_nms(const float* points, const float* logits, const float* offsets, const float* pmask,
                               float* out_segs, float* out_scores, int32_t* out_pidx, int32_t* out_count,
                               int B, int P, float pre_nms_thresh, float duration_thresh,
                               float min_score, float iou_thr, float voting_thr, int max_seg, void* stream) {
  AVDF_REQUIRE(points && logits && offsets && pmask && out_segs && out_scores && out_pidx && out_count, "decode_nms: null pointer");
  AVDF_REQUIRE(B > 0 && P > 0 && P <= kMaxP, "decode_nms: P=%d must be in 1..%d", P, kMaxP);
  AVDF_REQUIRE(max_seg > 0 && max_seg <= 256, "decode_nms: max_seg=%d must be in 1..256", max_seg);
  hipLaunchKernelGGL(decode_nms_kernel, dim3(B), dim3(kNT), 0, (hipStream_t)stream, points, logits, offsets, pmask,
                     out_segs, out_scores, out_pidx, out_count, P, pre_nms_thresh, duration_thresh, min_score,
                     iou_thr, voting_thr, max_seg);
  AVDF_CHECK_LAUNCH("decode_nms");
  return AVDF_OK;
}

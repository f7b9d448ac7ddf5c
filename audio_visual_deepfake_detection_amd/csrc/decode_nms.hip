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
  __shared__ Cand key[kMaxP];
  __shared__ float s0[kMaxP], s1[kMaxP];        // segment of flat point p (feature-grid units)
  __shared__ unsigned char alive[kMaxP];
  __shared__ int kept[256];
  __shared__ int n_cand, n_nms;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) { n_cand = 0; n_nms = 0; }
  __syncthreads();

  int lc = 0, ln = 0;
  for (int p = tid; p < kMaxP; p += kNT) {
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
    key[p] = c;
  }
  atomicAdd(&n_cand, lc);
  atomicAdd(&n_nms, ln);
  __syncthreads();

  // bitonic sort of 2048 keys, rank order = `before`
  for (int k = 2; k <= kMaxP; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < kMaxP; i += kNT) {
        const int l = i ^ j;
        if (l > i) {
          const Cand a = key[i], c = key[l];
          const bool up = (i & k) == 0;
          if (up ? before(c, a) : before(a, c)) { key[i] = c; key[l] = a; }
        }
      }
      __syncthreads();
    }
  }

  const int nn = n_nms;      // candidates with score > min_score occupy ranks [0, nn)
  for (int i = tid; i < kMaxP; i += kNT) alive[i] = i < nn ? 1 : 0;
  __syncthreads();

  // greedy NMS in rank order; only the first max_seg survivors are ever reported
  // (the survivor count lives in a register of every thread -- alive[i] is read uniformly, so all threads count alike --
  // and one barrier per kept segment orders the suppression writes before the next alive[] read)
  const int cap = max_seg < 256 ? max_seg : 256;
  int nkeep = 0;
  for (int i = 0; i < nn; ++i) {
    if (!alive[i]) continue;                    // uniform: LDS value read after the last barrier
    if (tid == 0) { kept[nkeep] = i; }
    const int pi = key[i].idx;
    const float ix1 = s0[pi], ix2 = s1[pi];
    const float ia = ix2 - ix1 + 1e-6f;
    for (int j = i + 1 + tid; j < nn; j += kNT) {
      if (!alive[j]) continue;
      const int pj = key[j].idx;
      const float xx1 = fmaxf(ix1, s0[pj]), xx2 = fminf(ix2, s1[pj]);
      const float inter = fmaxf(0.f, xx2 - xx1);
      const float ja = s1[pj] - s0[pj] + 1e-6f;
      const float ovr = inter / (ia + ja - inter);
      if (ovr >= iou_thr) alive[j] = 0;
    }
    if (++nkeep >= cap) break;
    __syncthreads();
  }
  __syncthreads();
  const int nk = nkeep;
  const int nc = n_cand;     // voting pool = every decoded candidate (ranks [0, nc)), nms.py:176-182

  const int lane = tid & 63, wid = tid >> 6;
  for (int q = wid; q < nk; q += kNT / 64) {
    const int ri = kept[q];
    const int pi = key[ri].idx;
    float a0 = s0[pi], a1 = s1[pi];
    if (voting_thr > 0.f) {
      const float la = a1 - a0;
      float wsum = 0.f;
      float sw[kMaxP / 64];                      // score * iou of this lane's candidates (0 below the voting threshold)
#pragma unroll
      for (int m = 0; m < kMaxP / 64; ++m) {
        const int j = lane + m * 64;
        sw[m] = 0.f;
        if (j < nc) {
          const int pj = key[j].idx;
          const float inter = fmaxf(fminf(a1, s1[pj]) - fmaxf(a0, s0[pj]), 0.f);
          const float iou = inter / (la + (s1[pj] - s0[pj]) - inter);
          if (iou >= voting_thr) { sw[m] = key[j].score * iou; wsum += sw[m]; }
        }
      }
      wsum = wave_sum(wsum);
      float r0 = 0.f, r1 = 0.f;
#pragma unroll
      for (int m = 0; m < kMaxP / 64; ++m) {
        const int j = lane + m * 64;
        if (j < nc && sw[m] != 0.f) {
          const int pj = key[j].idx;
          const float wgt = sw[m] / wsum;
          r0 += wgt * s0[pj];
          r1 += wgt * s1[pj];
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

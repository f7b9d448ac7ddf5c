// Host-side pieces of libavdf_hip.so: error reporting, version, host 1-D NMS.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <cmath>
#include <vector>
#include "../../include/avdf.h"

static thread_local char g_err[512] = "";

void avdf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* avdf_last_error(void) { return g_err; }
extern "C" int avdf_version(void) { return 1; }

// Hard NMS with the semantics of nms_1d_cpu (libs/utils/csrc/nms_cpu.cpp:19-58): fp32 arithmetic,
// areas = x2-x1+1e-6, visit in descending score (stable: ties keep input order), suppress IoU >= thr.
extern "C" int avdf_nms1d_host(const float* segs, const float* scores, int64_t n, float iou_thr,
                               int64_t* keep, int64_t* n_keep) {
  if (n < 0 || !n_keep || (n > 0 && (!segs || !scores || !keep))) {
    avdf_set_error("nms1d_host: bad args");
    return AVDF_EINVAL;
  }
  std::vector<int64_t> order((size_t)n);
  for (int64_t i = 0; i < n; ++i) order[(size_t)i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return scores[a] > scores[b]; });
  std::vector<unsigned char> sel((size_t)n, 1);
  int64_t k = 0;
  for (int64_t a = 0; a < n; ++a) {
    if (!sel[(size_t)a]) continue;
    const int64_t i = order[(size_t)a];
    keep[k++] = i;
    const float ix1 = segs[2 * i], ix2 = segs[2 * i + 1];
    const float ia = ix2 - ix1 + 1e-6f;
    for (int64_t b = a + 1; b < n; ++b) {
      if (!sel[(size_t)b]) continue;
      const int64_t j = order[(size_t)b];
      const float xx1 = std::max(ix1, segs[2 * j]), xx2 = std::min(ix2, segs[2 * j + 1]);
      const float inter = std::max(0.f, xx2 - xx1);
      const float ja = segs[2 * j + 1] - segs[2 * j] + 1e-6f;
      const float ovr = inter / (ia + ja - inter);
      if (ovr >= iou_thr) sel[(size_t)b] = 0;
    }
  }
  *n_keep = k;
  return AVDF_OK;
}

// Host mirror of batched_nms (class-agnostic hard-NMS branch) + seg_voting (libs/utils/nms.py:67-190):
// score filter (> min_score when min_score > 0), NMS, cap at max_seg, voting over ALL input segments with
// IoU >= voting_thr weighted by score*IoU, final order by descending score.  HOST pointers.
extern "C" int avdf_batched_nms_host(const float* segs, const float* scores, int64_t n, float iou_thr, float min_score,
                                     int64_t max_seg, float voting_thr, float* out_segs, float* out_scores,
                                     int64_t* out_idx, int64_t* n_out) {
  if (n < 0 || !n_out || max_seg < 0 || (n > 0 && (!segs || !scores || !out_segs || !out_scores || !out_idx))) {
    avdf_set_error("batched_nms_host: bad args");
    return AVDF_EINVAL;
  }
  std::vector<int64_t> valid;
  for (int64_t i = 0; i < n; ++i)
    if (!(min_score > 0.f) || scores[i] > min_score) valid.push_back(i);
  const int64_t m = (int64_t)valid.size();
  std::vector<float> s2((size_t)m * 2), c2((size_t)m);
  for (int64_t i = 0; i < m; ++i) { s2[2 * i] = segs[2 * valid[i]]; s2[2 * i + 1] = segs[2 * valid[i] + 1]; c2[i] = scores[valid[i]]; }
  std::vector<int64_t> keep((size_t)std::max<int64_t>(m, 1));
  int64_t nk = 0;
  int rc = avdf_nms1d_host(s2.data(), c2.data(), m, iou_thr, keep.data(), &nk);
  if (rc) return rc;
  if (max_seg > 0 && nk > max_seg) nk = max_seg;
  for (int64_t q = 0; q < nk; ++q) {
    const int64_t i = valid[keep[q]];
    float a0 = segs[2 * i], a1 = segs[2 * i + 1];
    if (voting_thr > 0.f) {
      const float la = a1 - a0;
      float wsum = 0.f;
      std::vector<float> w((size_t)n, 0.f);
      for (int64_t j = 0; j < n; ++j) {
        const float inter = std::max(std::min(a1, segs[2 * j + 1]) - std::max(a0, segs[2 * j]), 0.f);
        const float iou = inter / (la + (segs[2 * j + 1] - segs[2 * j]) - inter);
        if (iou >= voting_thr) { w[j] = scores[j] * iou; wsum += w[j]; }
      }
      float r0 = 0.f, r1 = 0.f;
      for (int64_t j = 0; j < n; ++j)
        if (w[j] != 0.f) { const float ww = w[j] / wsum; r0 += ww * segs[2 * j]; r1 += ww * segs[2 * j + 1]; }
      a0 = r0; a1 = r1;
    }
    out_segs[2 * q] = a0; out_segs[2 * q + 1] = a1; out_scores[q] = scores[i]; out_idx[q] = i;
  }
  *n_out = nk;      // NMS order is already descending-score: the final sort of nms.py:182-189 is the identity
  return AVDF_OK;
}

// Soft-NMS (Bodla et al. 2017) with the reference's conventions (softnms_1d_cpu, libs/utils/csrc/nms_cpu.cpp:67-158):
// repeatedly take the live segment with the highest CURRENT score (first one in array order on ties), emit it, decay
// the scores of the remaining live ones by their IoU with it (method 0: hard, 1: linear above iou_thr, 2: gaussian
// exp(-iou^2/sigma)) and drop a segment as soon as its score falls below min_score.  The working set is kept as a
// compact array: position i is swapped with the maximum, a dropped entry is overwritten by the last live one (that
// order decides exact ties, so it is part of the contract).  dets [n,3] = (x1, x2, decayed score) in emission
// order, inds = their indices into the input; both hold *n_out valid rows.  HOST pointers.
extern "C" int avdf_softnms1d_host(const float* segs, const float* scores, int64_t n, float iou_thr, float sigma,
                                   float min_score, int method, float* dets, int64_t* inds, int64_t* n_out) {
  if (n < 0 || !n_out || method < 0 || method > 2 || (n > 0 && (!segs || !scores || !dets || !inds))) {
    avdf_set_error("softnms1d_host: bad args");
    return AVDF_EINVAL;
  }
  struct Live { float a, b, score, len; int64_t src; };
  std::vector<Live> live((size_t)n);
  for (int64_t i = 0; i < n; ++i) live[(size_t)i] = Live{segs[2 * i], segs[2 * i + 1], scores[i], segs[2 * i + 1] - segs[2 * i] + 1e-6f, i};
  int64_t m = n;                                   // live count; [0, i) already emitted
  for (int64_t i = 0; i < m; ++i) {
    int64_t top = i;
    for (int64_t p = i + 1; p < m; ++p)
      if (live[(size_t)top].score < live[(size_t)p].score) top = p;
    std::swap(live[(size_t)i], live[(size_t)top]);
    const Live cur = live[(size_t)i];
    dets[3 * i] = cur.a; dets[3 * i + 1] = cur.b; dets[3 * i + 2] = cur.score;
    for (int64_t p = i + 1; p < m; ++p) {
      Live& o = live[(size_t)p];
      const float inter = std::max(0.f, std::min(cur.b, o.b) - std::max(cur.a, o.a));
      const float ovr = inter / (cur.len + o.len - inter);
      float wgt = 1.f;
      if (method == 0) { if (ovr >= iou_thr) wgt = 0.f; }
      else if (method == 1) { if (ovr >= iou_thr) wgt = 1.f - ovr; }
      else wgt = std::exp(-(ovr * ovr) / sigma);
      o.score *= wgt;
      if (o.score < min_score) {
        o = live[(size_t)(m - 1)];
        --m;
        --p;                                       // re-examine the entry moved into this slot
      }
    }
  }
  for (int64_t i = 0; i < m; ++i) inds[i] = live[(size_t)i].src;
  *n_out = m;
  return AVDF_OK;
}

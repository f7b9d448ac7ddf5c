// Host-side pieces of libavdf_hip.so: error reporting, version, host 1-D NMS.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
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

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

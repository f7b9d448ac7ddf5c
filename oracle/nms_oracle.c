/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the reference's 1-D NMS.
 *
 * nms_1d_oracle follows nms_1d_cpu (libs/utils/csrc/nms_cpu.cpp:19-58): areas = x2-x1+1e-6,
 * candidates visited in descending-score order (ties: lower input index first, i.e. a stable
 * sort), a candidate is suppressed when IoU >= thr with an earlier kept one.  All arithmetic
 * in float like the reference.  Returns the number of kept indices written to keep[].
 */
#include <stdint.h>
#include <stdlib.h>

typedef struct { float s; int64_t i; } pair_t;

static int cmp_desc(const void *a, const void *b) {
    const pair_t *x = (const pair_t *)a, *y = (const pair_t *)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

int64_t nms_1d_oracle(const float *segs, const float *scores, int64_t n, float thr, int64_t *keep) {
    if (n <= 0) return 0;
    pair_t *ord = (pair_t *)malloc(sizeof(pair_t) * (size_t)n);
    unsigned char *sel = (unsigned char *)malloc((size_t)n);
    float *area = (float *)malloc(sizeof(float) * (size_t)n);
    for (int64_t i = 0; i < n; i++) {
        ord[i].s = scores[i]; ord[i].i = i; sel[i] = 1;
        area[i] = segs[2 * i + 1] - segs[2 * i] + 1e-6f;
    }
    qsort(ord, (size_t)n, sizeof(pair_t), cmp_desc);
    for (int64_t a = 0; a < n; a++) {
        if (!sel[a]) continue;
        int64_t i = ord[a].i;
        float ix1 = segs[2 * i], ix2 = segs[2 * i + 1], ia = area[i];
        for (int64_t b = a + 1; b < n; b++) {
            if (!sel[b]) continue;
            int64_t j = ord[b].i;
            float xx1 = ix1 > segs[2 * j] ? ix1 : segs[2 * j];
            float xx2 = ix2 < segs[2 * j + 1] ? ix2 : segs[2 * j + 1];
            float inter = xx2 - xx1; if (inter < 0.f) inter = 0.f;
            float ovr = inter / (ia + area[j] - inter);
            if (ovr >= thr) sel[b] = 0;
        }
    }
    int64_t k = 0;
    for (int64_t a = 0; a < n; a++) if (sel[a]) keep[k++] = ord[a].i;
    free(ord); free(sel); free(area);
    return k;
}

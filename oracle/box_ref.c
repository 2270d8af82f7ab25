/*
 * oracle/box_ref.c — CPU restatement of the box ops (TEST INFRASTRUCTURE ONLY).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product
 * (torch_detection_amd) never does.
 *
 * PARITY UNPINNED: the reference has no anchor / IoU / NMS code at all (/root/reference/core/__init__.py
 * is a 0-byte file) and no tests or golden vectors.  The semantics below are SURVEY.md Appendix B
 * (mmdetection-v0.x lineage, un-vendored and un-pinned upstream), chosen to agree with the conventions
 * the reference does pin:
 *   - inclusive "+1" pixel boxes, xyxy float32:   datasets/utils/bbox.py:39 (x2 = x1+w-1), :375-377 (w = x2-x1+1)
 *   - grids enumerate x fastest, then y:          datasets/dataset_transforms.py:120-131
 * Known-answer vectors (Appendix B) are checked in tests/test_oracle_box.py.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile).  All arithmetic is
 * IEEE binary32 in the exact operation order written here; the HIP kernels must match bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* base_anchors: ratio-major, scale-minor (Appendix B).  out: [nr*ns][4]. */
void ref_base_anchors(float base_size, const float* scales, int ns, const float* ratios, int nr, float* out) {
  const float w = base_size, h = base_size;
  const float cx = 0.5f * (w - 1.0f), cy = 0.5f * (h - 1.0f);
  for (int r = 0; r < nr; ++r) {
    const float h_ratio = sqrtf(ratios[r]);
    const float w_ratio = 1.0f / h_ratio;
    for (int s = 0; s < ns; ++s) {
      const float ws = (w * w_ratio) * scales[s];
      const float hs = (h * h_ratio) * scales[s];
      float* o = out + (size_t)(r * ns + s) * 4;
      o[0] = rintf(cx - 0.5f * (ws - 1.0f));
      o[1] = rintf(cy - 0.5f * (hs - 1.0f));
      o[2] = rintf(cx + 0.5f * (ws - 1.0f));
      o[3] = rintf(cy + 0.5f * (hs - 1.0f));
    }
  }
}

/* anchors[(y*featW + x)*A + a] = base[a] + (x*stride, y*stride, x*stride, y*stride); valid flag per anchor. */
void ref_anchor_grid(const float* base, int A, int featH, int featW, int stride, int valid_h, int valid_w,
                     float* anchors, uint8_t* valid) {
  for (int y = 0; y < featH; ++y)
    for (int x = 0; x < featW; ++x) {
      const float sx = (float)(x * stride), sy = (float)(y * stride);
      for (int a = 0; a < A; ++a) {
        const size_t i = ((size_t)y * featW + x) * A + a;
        anchors[i * 4 + 0] = base[a * 4 + 0] + sx;
        anchors[i * 4 + 1] = base[a * 4 + 1] + sy;
        anchors[i * 4 + 2] = base[a * 4 + 2] + sx;
        anchors[i * 4 + 3] = base[a * 4 + 3] + sy;
        if (valid) valid[i] = (x < valid_w && y < valid_h) ? 1 : 0;
      }
    }
}

static inline float box_area(const float* b) { return ((b[2] - b[0]) + 1.0f) * ((b[3] - b[1]) + 1.0f); }

static inline float box_iou(const float* a, float area_a, const float* b) {
  const float ltx = fmaxf(a[0], b[0]), lty = fmaxf(a[1], b[1]);
  const float rbx = fminf(a[2], b[2]), rby = fminf(a[3], b[3]);
  const float w = fmaxf((rbx - ltx) + 1.0f, 0.0f);
  const float h = fmaxf((rby - lty) + 1.0f, 0.0f);
  const float inter = w * h;
  const float area_b = box_area(b);
  const float uni = (area_a + area_b) - inter;
  return inter / uni;
}

void ref_iou_pairwise(const float* a, int N, const float* b, int M, float* out) {
  for (int i = 0; i < N; ++i) {
    const float area_a = box_area(a + (size_t)i * 4);
    for (int j = 0; j < M; ++j) out[(size_t)i * M + j] = box_iou(a + (size_t)i * 4, area_a, b + (size_t)j * 4);
  }
}

typedef struct { float s; int i; } ref_item;

static int cmp_desc(const void* pa, const void* pb) {
  const ref_item* a = (const ref_item*)pa;
  const ref_item* b = (const ref_item*)pb;
  if (a->s > b->s) return -1;
  if (a->s < b->s) return 1;
  return (a->i > b->i) - (a->i < b->i); /* ties: lower original index first (stable) */
}

/* Greedy NMS. keep[N] (original order), kept_idx[N] (score order; tail = -1). Returns #kept. */
int ref_nms(const float* boxes, const float* scores, int N, float thr, uint8_t* keep, int64_t* kept_idx) {
  if (N <= 0) return 0;
  ref_item* it = (ref_item*)malloc(sizeof(ref_item) * (size_t)N);
  uint8_t* dead = (uint8_t*)calloc((size_t)N, 1);
  for (int i = 0; i < N; ++i) { it[i].s = scores[i]; it[i].i = i; keep[i] = 0; kept_idx[i] = -1; }
  qsort(it, (size_t)N, sizeof(ref_item), cmp_desc);
  int cnt = 0;
  for (int p = 0; p < N; ++p) {
    if (dead[p]) continue;
    const int i = it[p].i;
    keep[i] = 1;
    kept_idx[cnt++] = i;
    const float* bi = boxes + (size_t)i * 4;
    const float area_i = box_area(bi);
    for (int q = p + 1; q < N; ++q) {
      if (dead[q]) continue;
      if (box_iou(bi, area_i, boxes + (size_t)it[q].i * 4) > thr) dead[q] = 1;
    }
  }
  free(it);
  free(dead);
  return cnt;
}

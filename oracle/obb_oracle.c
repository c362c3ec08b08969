/*
 * obb_oracle.c -- CPU ORACLE for the oriented-bounding-box hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product (aidet_amd/) never
 * imports, links or executes anything in oracle/.
 *
 * PARITY STATUS: "parity unpinned" for the rotated arithmetic itself -- the
 * reference (jwwangchn/aidet) ships no rotated IoU / rotated NMS / rotated
 * RoI-Align; its polygon NMS lives in the un-vendored, un-pinned package
 * `wwtool` (mmdet/datasets/dota.py:21-23,334-336).  What IS pinned, through
 * tests/golden/ fixtures generated from the reference itself:
 *   - greedy-NMS control flow, `>=` threshold, ascending keep order
 *       mmdet/ops/nms/src/nms_cpu.cpp:5-60      (obbo_rnms_f32 below)
 *   - HBB IoU values / +1 pixel convention at theta = 0
 *       mmdet/core/bbox/geometry.py:47-88       (obbo_rbbox_overlaps_f32)
 *   - RoIAlign sampling + bilinear rules at theta = 0
 *       mmdet/ops/roi_align/src/roi_align_kernel_v2.cu:15-128 (V2)
 *       mmdet/ops/roi_align/src/roi_align_kernel.cu:17-117    (V1)
 *   - theta-OBB corner convention (cv2.boxPoints)
 *       mmdet/core/rbbox/transforms.py:45-55
 *
 * ARITHMETIC SPEC ("OBB-GEOM v1").  Every fp32 routine below uses only
 * + - * / comparisons and floorf, in a FIXED operation order, compiled with
 * -ffp-contract=off.  The HIP kernels implement the same sequence, so integer
 * results (keep indices) are bit-exact and float results agree to the last bit
 * in practice (tests allow 1e-4 as BASELINE.json states).
 *
 * Plain C99.  Build: see oracle/Makefile.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OBBO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* deterministic sin/cos (no libm: glibc and ROCm OCML differ in ulps) */
/* ------------------------------------------------------------------ */
/* Cody-Waite reduction by pi/2 (3 constants), cephes-style minimax
 * polynomials on [-pi/4, pi/4]; quadrant from k mod 4 in float arithmetic. */
OBBO_API void obbo_sincos_f32(float x, float *s_out, float *c_out) {
  float q = x * 0.636619772f;
  float kf = floorf(q + 0.5f);
  float r = x - kf * 1.5703125f;
  r = r - kf * 4.837512969970703125e-4f;
  r = r - kf * 7.54978995489188e-8f;
  float z = r * r;
  float sp = -1.9515295891e-4f * z;
  sp = sp + 8.3321608736e-3f;
  sp = sp * z;
  sp = sp - 1.6666654611e-1f;
  sp = sp * z;
  sp = sp * r;
  sp = sp + r;
  float cp = 2.443315711809948e-5f * z;
  cp = cp - 1.388731625493765e-3f;
  cp = cp * z;
  cp = cp + 4.166664568298827e-2f;
  cp = cp * z;
  cp = cp * z;
  float hz = 0.5f * z;
  cp = cp - hz;
  cp = cp + 1.0f;
  /* quadrant = kf mod 4, computed exactly in float */
  float k4 = floorf(kf * 0.25f);
  float kq = kf - 4.0f * k4; /* 0,1,2,3 */
  float s, c;
  if (kq == 0.0f) { s = sp; c = cp; }
  else if (kq == 1.0f) { s = cp; c = -sp; }
  else if (kq == 2.0f) { s = -sp; c = -cp; }
  else { s = -cp; c = sp; }
  *s_out = s;
  *c_out = c;
}

/* ------------------------------------------------------------------ */
/* box records                                                         */
/* ------------------------------------------------------------------ */
/* A "quad record" is what the pair routine consumes:
 *   ox, oy            origin (box centre, or vertex 0 of a polygon)
 *   rx[4], ry[4]      the 4 vertices RELATIVE to (ox,oy), positive orientation
 *                     (shoelace sum > 0 in image coordinates)
 *   ex, ey            AABB half extents measured from (ox,oy):  |rx|,|ry| max
 *   area              w*h for a theta-OBB; |shoelace|/2 for a polygon
 * theta-OBB meaning follows cv2.boxPoints (transforms.py:45-55):
 *   x = cx + lx*cos(t) - ly*sin(t),  y = cy + lx*sin(t) + ly*cos(t).       */
typedef struct {
  float ox, oy;
  float rx[4], ry[4];
  float ex, ey;
  float area;
} obbo_quad;

static inline float fabs_sel(float a) { return a < 0.0f ? -a : a; }
static inline float max_sel(float a, float b) { return a < b ? b : a; }
static inline float min_sel(float a, float b) { return b < a ? b : a; }

OBBO_API void obbo_quad_from_rbox_f32(const float *b, obbo_quad *q) {
  float cx = b[0], cy = b[1], w = b[2], h = b[3], th = b[4];
  float s, c;
  obbo_sincos_f32(th, &s, &c);
  float hw = 0.5f * w, hh = 0.5f * h;
  float ux = hw * c, uy = hw * s;
  float t = hh * s;
  float vx = -t, vy = hh * c;
  q->ox = cx; q->oy = cy;
  float mx = -ux, my = -uy; /* -u */
  q->rx[0] = mx - vx; q->ry[0] = my - vy; /* -u - v */
  q->rx[1] = ux - vx; q->ry[1] = uy - vy; /* +u - v */
  q->rx[2] = ux + vx; q->ry[2] = uy + vy; /* +u + v */
  q->rx[3] = mx + vx; q->ry[3] = my + vy; /* -u + v */
  q->ex = fabs_sel(ux) + fabs_sel(vx);
  q->ey = fabs_sel(uy) + fabs_sel(vy);
  q->area = w * h;
}

/* polygon (x1,y1,...,x4,y4): origin = vertex 0; orientation normalised by the
 * sign of the shoelace sum (negative -> vertices 1 and 3 swapped).          */
OBBO_API void obbo_quad_from_poly_f32(const float *p, obbo_quad *q) {
  float ox = p[0], oy = p[1];
  float x1 = p[2] - ox, y1 = p[3] - oy;
  float x2 = p[4] - ox, y2 = p[5] - oy;
  float x3 = p[6] - ox, y3 = p[7] - oy;
  /* shoelace with vertex 0 at the origin: terms 0->1 and 3->0 vanish */
  float a = x1 * y2 - x2 * y1;
  float b2 = x2 * y3 - x3 * y2;
  float sh = a + b2;
  q->ox = ox; q->oy = oy;
  q->rx[0] = 0.0f; q->ry[0] = 0.0f;
  q->rx[2] = x2; q->ry[2] = y2;
  if (sh < 0.0f) {
    q->rx[1] = x3; q->ry[1] = y3;
    q->rx[3] = x1; q->ry[3] = y1;
    sh = -sh;
  } else {
    q->rx[1] = x1; q->ry[1] = y1;
    q->rx[3] = x3; q->ry[3] = y3;
  }
  float ex = max_sel(max_sel(fabs_sel(x1), fabs_sel(x2)), fabs_sel(x3));
  float ey = max_sel(max_sel(fabs_sel(y1), fabs_sel(y2)), fabs_sel(y3));
  q->ex = ex; q->ey = ey;
  q->area = 0.5f * sh;
}

/* ------------------------------------------------------------------ */
/* convex quad  x  convex quad  intersection area ("segment soup" S-H) */
/* ------------------------------------------------------------------ */
/* The subject A is held as up to 8 boundary SEGMENTS (ax,ay)->(bx,by), one
 * per supporting line (4 of A, 4 of B).  Clipping by a half-plane of B trims
 * every live segment and adds one new segment on the clip line that runs
 * from the (unique) leaving crossing to the (unique) entering crossing; both
 * end points are the very same floats stored into the trimmed segments, so the
 * boundary stays closed.  No vertex list, no data-dependent indexing.
 * Everything is relative to A's origin; (dx,dy) = B.origin - A.origin.      */
static float quad_inter_f32(const obbo_quad *A, const obbo_quad *B,
                            float dx, float dy) {
  float sax[8], say[8], sbx[8], sby[8];
  int valid[8];
  int k, m;
  for (k = 0; k < 4; ++k) {
    int k1 = (k + 1) & 3;
    sax[k] = A->rx[k]; say[k] = A->ry[k];
    sbx[k] = A->rx[k1]; sby[k] = A->ry[k1];
    valid[k] = 1;
  }
  for (k = 4; k < 8; ++k) {
    sax[k] = say[k] = sbx[k] = sby[k] = 0.0f;
    valid[k] = 0;
  }
  float qx[4], qy[4];
  for (k = 0; k < 4; ++k) {
    qx[k] = dx + B->rx[k];
    qy[k] = dy + B->ry[k];
  }
  for (m = 0; m < 4; ++m) {
    int m1 = (m + 1) & 3;
    float px = qx[m], py = qy[m];
    float ex = qx[m1] - px, ey = qy[m1] - py;
    float nsx = 0.0f, nsy = 0.0f, nex = 0.0f, ney = 0.0f;
    int has_leave = 0, has_enter = 0;
    for (k = 0; k < 4 + m; ++k) {
      if (!valid[k]) continue;
      float ta = say[k] - py, tb = sax[k] - px;
      float sa = ex * ta - ey * tb;
      float tc = sby[k] - py, td = sbx[k] - px;
      float sb = ex * tc - ey * td;
      int ina = sa >= 0.0f, inb = sb >= 0.0f;
      if (ina && inb) continue;
      if (!ina && !inb) { valid[k] = 0; continue; }
      float den = sa - sb;
      float t = sa / den;
      float ddx = sbx[k] - sax[k], ddy = sby[k] - say[k];
      float X = sax[k] + t * ddx;
      float Y = say[k] + t * ddy;
      if (ina) { /* leaving: keep a, end moves to the crossing */
        sbx[k] = X; sby[k] = Y;
        nsx = X; nsy = Y; has_leave = 1;
      } else {   /* entering: start moves to the crossing */
        sax[k] = X; say[k] = Y;
        nex = X; ney = Y; has_enter = 1;
      }
    }
    sax[4 + m] = nsx; say[4 + m] = nsy;
    sbx[4 + m] = nex; sby[4 + m] = ney;
    valid[4 + m] = has_leave && has_enter;
  }
  float acc = 0.0f;
  for (k = 0; k < 8; ++k) {
    float c1 = sax[k] * sby[k];
    float c2 = sbx[k] * say[k];
    float cr = c1 - c2;
    acc = acc + (valid[k] ? cr : 0.0f);
  }
  return 0.5f * acc;
}

/* overlap of two quad records.  mode 0 = iou, 1 = iof (inter / area(A)). */
static float quad_overlap_f32(const obbo_quad *A, const obbo_quad *B, int mode) {
  float dx = B->ox - A->ox, dy = B->oy - A->oy;
  /* AABB prefilter -- part of the spec (decides exact zeros) */
  float lox = A->ex + B->ex, loy = A->ey + B->ey;
  /* polygons measure extents from vertex 0, boxes from the centre; in both
   * cases |d| > eA + eB on an axis proves the quads are disjoint.          */
  if (fabs_sel(dx) > lox) return 0.0f;
  if (fabs_sel(dy) > loy) return 0.0f;
  float inter = quad_inter_f32(A, B, dx, dy);
  float lim = min_sel(A->area, B->area);
  inter = min_sel(inter, lim);
  inter = max_sel(inter, 0.0f);
  float den;
  if (mode == 0) {
    float su = A->area + B->area;
    den = su - inter;
  } else {
    den = A->area;
  }
  if (!(den > 0.0f)) return 0.0f;
  return inter / den;
}

OBBO_API float obbo_rbox_iou_f32(const float *a5, const float *b5, int mode) {
  obbo_quad A, B;
  obbo_quad_from_rbox_f32(a5, &A);
  obbo_quad_from_rbox_f32(b5, &B);
  return quad_overlap_f32(&A, &B, mode);
}

OBBO_API float obbo_poly_iou_f32(const float *a8, const float *b8, int mode) {
  obbo_quad A, B;
  obbo_quad_from_poly_f32(a8, &A);
  obbo_quad_from_poly_f32(b8, &B);
  return quad_overlap_f32(&A, &B, mode);
}

/* ------------------------------------------------------------------ */
/* rbbox_overlaps  (signature mirrors geometry.py:4 bbox_overlaps)      */
/* ------------------------------------------------------------------ */
/* kind: 0 = theta-OBB rows of 5 floats, 1 = polygon rows of 8 floats.
 * aligned=0: out is (m,n) row-major; aligned=1: m==n, out is (m,).       */
OBBO_API int obbo_overlaps_f32(const float *a, long m, const float *b, long n,
                               int kind, int mode, int aligned, float *out) {
  long i, j;
  int stride = kind ? 8 : 5;
  if (m < 0 || n < 0) return -1;
  if (aligned && m != n) return -1;
  obbo_quad *QB = (obbo_quad *)malloc(sizeof(obbo_quad) * (size_t)(n > 0 ? n : 1));
  if (!QB) return -2;
  for (j = 0; j < n; ++j) {
    if (kind) obbo_quad_from_poly_f32(b + j * stride, &QB[j]);
    else obbo_quad_from_rbox_f32(b + j * stride, &QB[j]);
  }
  for (i = 0; i < m; ++i) {
    obbo_quad A;
    if (kind) obbo_quad_from_poly_f32(a + i * stride, &A);
    else obbo_quad_from_rbox_f32(a + i * stride, &A);
    if (aligned) {
      out[i] = quad_overlap_f32(&A, &QB[i], mode);
    } else {
      for (j = 0; j < n; ++j) out[i * n + j] = quad_overlap_f32(&A, &QB[j], mode);
    }
  }
  free(QB);
  return 0;
}

/* ------------------------------------------------------------------ */
/* fp64 "truth" (libm trig, no prefilter, classic vertex-list S-H)     */
/* ------------------------------------------------------------------ */
/* A deliberately DIFFERENT algorithm from the fp32 spec above, used to bound
 * the fp32 error (tests: |f32 - f64| <= 1e-4).                           */
static double poly_area_d(const double *px, const double *py, int n) {
  double s = 0.0;
  int i;
  for (i = 0; i < n; ++i) {
    int j = (i + 1) % n;
    s += px[i] * py[j] - px[j] * py[i];
  }
  return 0.5 * s;
}

static double convex_inter_d(const double *ax, const double *ay, int na,
                             const double *bx, const double *by, int nb) {
  double px[16], py[16], qx[16], qy[16];
  int n = na, i, m;
  for (i = 0; i < na; ++i) { px[i] = ax[i]; py[i] = ay[i]; }
  for (m = 0; m < nb && n > 0; ++m) {
    int m1 = (m + 1) % nb;
    double ex = bx[m1] - bx[m], ey = by[m1] - by[m];
    int cnt = 0;
    for (i = 0; i < n; ++i) {
      int j = (i + 1) % n;
      double si = ex * (py[i] - by[m]) - ey * (px[i] - bx[m]);
      double sj = ex * (py[j] - by[m]) - ey * (px[j] - bx[m]);
      if (si >= 0) {
        qx[cnt] = px[i]; qy[cnt] = py[i]; ++cnt;
        if (sj < 0) {
          double t = si / (si - sj);
          qx[cnt] = px[i] + t * (px[j] - px[i]);
          qy[cnt] = py[i] + t * (py[j] - py[i]); ++cnt;
        }
      } else if (sj >= 0) {
        double t = si / (si - sj);
        qx[cnt] = px[i] + t * (px[j] - px[i]);
        qy[cnt] = py[i] + t * (py[j] - py[i]); ++cnt;
      }
    }
    n = cnt;
    for (i = 0; i < n; ++i) { px[i] = qx[i]; py[i] = qy[i]; }
  }
  if (n < 3) return 0.0;
  double a = poly_area_d(px, py, n);
  return a < 0 ? 0.0 : a;
}

static void rbox_corners_d(const double *b, double *x, double *y) {
  double c = cos(b[4]), s = sin(b[4]);
  double ux = 0.5 * b[2] * c, uy = 0.5 * b[2] * s;
  double vx = -0.5 * b[3] * s, vy = 0.5 * b[3] * c;
  x[0] = b[0] - ux - vx; y[0] = b[1] - uy - vy;
  x[1] = b[0] + ux - vx; y[1] = b[1] + uy - vy;
  x[2] = b[0] + ux + vx; y[2] = b[1] + uy + vy;
  x[3] = b[0] - ux + vx; y[3] = b[1] - uy + vy;
}

static void poly_corners_d(const double *p, double *x, double *y) {
  int i;
  for (i = 0; i < 4; ++i) { x[i] = p[2 * i]; y[i] = p[2 * i + 1]; }
  if (poly_area_d(x, y, 4) < 0) {
    double t;
    t = x[1]; x[1] = x[3]; x[3] = t;
    t = y[1]; y[1] = y[3]; y[3] = t;
  }
}

OBBO_API int obbo_overlaps_f64(const double *a, long m, const double *b, long n,
                               int kind, int mode, int aligned, double *out) {
  long i, j;
  int stride = kind ? 8 : 5;
  if (aligned && m != n) return -1;
  for (i = 0; i < m; ++i) {
    double ax[4], ay[4];
    if (kind) poly_corners_d(a + i * stride, ax, ay);
    else rbox_corners_d(a + i * stride, ax, ay);
    double aa = kind ? poly_area_d(ax, ay, 4) : a[i * stride + 2] * a[i * stride + 3];
    long j0 = aligned ? i : 0, j1 = aligned ? i + 1 : n;
    for (j = j0; j < j1; ++j) {
      double bx[4], by[4];
      if (kind) poly_corners_d(b + j * stride, bx, by);
      else rbox_corners_d(b + j * stride, bx, by);
      double ab = kind ? poly_area_d(bx, by, 4) : b[j * stride + 2] * b[j * stride + 3];
      double inter = convex_inter_d(ax, ay, 4, bx, by, 4);
      double den = mode == 0 ? aa + ab - inter : aa;
      double v = den > 0 ? inter / den : 0.0;
      if (aligned) out[i] = v; else out[i * n + j] = v;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* score ordering                                                      */
/* ------------------------------------------------------------------ */
/* descending score, ties -> lower original index (stable).  The key is the
 * usual order-preserving map float -> uint32 (after -0 -> +0), inverted so an
 * ASCENDING stable sort of keys is a DESCENDING stable sort of scores.  The
 * HIP path radix-sorts the same keys.  (nms_cpu.cpp:21 uses an unstable
 * descending sort; distinct scores make both agree.)                       */
OBBO_API uint32_t obbo_score_key(float s) {
  union { float f; uint32_t u; } v;
  v.f = s + 0.0f;
  uint32_t u = v.u;
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ~u;
}

typedef struct { uint32_t key; int64_t idx; } key_idx;

static int cmp_key_idx(const void *a, const void *b) {
  const key_idx *x = (const key_idx *)a, *y = (const key_idx *)b;
  if (x->key < y->key) return -1;
  if (x->key > y->key) return 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* ------------------------------------------------------------------ */
/* rotated / polygon NMS -- control flow of nms_cpu.cpp:21-59           */
/* ------------------------------------------------------------------ */
/* kind 0: dets (n,6) = cx,cy,w,h,theta,score.  kind 1: dets (n,9) = 8 coords,
 * score.  keep receives ASCENDING original indices (nms_cpu.cpp:59).
 * Returns the number kept, or <0 on error.                                */
/* grouped variant: boxes of different groups never suppress each other (the
 * per-class / per-image NMS that bbox_nms.py:56-63 obtains by shifting boxes of
 * class c by c*(max_coordinate+1), and that the DOTA merge applies per class
 * file, dota.py:310-336).  group may be NULL (one group); thrs may be NULL
 * (use thr) or give a threshold per box (the threshold of the EARLIER box of a
 * pair is used; boxes of one group share it: per-class thresholds, dota.py:324). */
OBBO_API long obbo_nms_ex_f32(const float *dets, long n, int kind, float thr,
                              const int32_t *group, const float *thrs, int flags, int64_t *keep);

OBBO_API long obbo_nms_grouped_f32(const float *dets, long n, int kind, float thr,
                                   const int32_t *group, const float *thrs, int64_t *keep) {
  return obbo_nms_ex_f32(dets, n, kind, thr, group, thrs, 0, keep);
}

OBBO_API long obbo_nms_f32(const float *dets, long n, int kind, float thr,
                           int64_t *keep) {
  return obbo_nms_ex_f32(dets, n, kind, thr, NULL, NULL, 0, keep);
}

/* the reference's horizontal pair overlap, nms_cpu.cpp:19,47-55, in its operation order:
 * a[] / b[] = x1, y1, x2, y2; areas = (x2 - x1 + 1) * (y2 - y1 + 1).                    */
static inline float hbb_area_f32(const float *d) {
  return ((d[2] - d[0]) + 1.0f) * ((d[3] - d[1]) + 1.0f);
}
static inline float hbb_overlap_f32(const float *a, float aarea, const float *b, float barea) {
  float xx1 = max_sel(a[0], b[0]), yy1 = max_sel(a[1], b[1]);   /* std::max(ix1, x1[j]) */
  float xx2 = min_sel(a[2], b[2]), yy2 = min_sel(a[3], b[3]);
  float w = max_sel(0.0f, (xx2 - xx1) + 1.0f), h = max_sel(0.0f, (yy2 - yy1) + 1.0f);
  float inter = w * h;
  return inter / ((aarea + barea) - inter);
}

/* kind 0 / 1: rotated / polygon rows (OBB-GEOM v1 overlap); kind 2: horizontal rows
 * x1,y1,x2,y2,score with EXACTLY the reference's arithmetic (nms_cpu.cpp:19,47-56).
 * flags bit 0 ("strict"): suppress on ovr > thr, the comparison of the reference's CUDA
 * kernel (nms_kernel.cu:61), instead of the CPU reference's ovr >= thr (nms_cpu.cpp:56). */
OBBO_API long obbo_nms_ex_f32(const float *dets, long n, int kind, float thr,
                              const int32_t *group, const float *thrs, int flags, int64_t *keep) {
  long i, j;
  const int strict = flags & 1;
  int stride = kind == 1 ? 9 : (kind == 2 ? 5 : 6);
  if (n < 0) return -1;
  if (n == 0) return 0;
  key_idx *ord = (key_idx *)malloc(sizeof(key_idx) * (size_t)n);
  obbo_quad *Q = (obbo_quad *)malloc(sizeof(obbo_quad) * (size_t)n);
  uint8_t *sup = (uint8_t *)calloc((size_t)n, 1);
  /* SoA copies of the prefilter fields in SORTED order: keeps the O(n^2)
   * inner loop streaming                                                    */
  float *sx = (float *)malloc(sizeof(float) * (size_t)n * 5);
  if (!ord || !Q || !sup || !sx) { free(ord); free(Q); free(sup); free(sx); return -2; }
  float *sy = sx + n, *sex = sx + 2 * n, *sey = sx + 3 * n, *sar = sx + 4 * n;
  for (i = 0; i < n; ++i) {
    ord[i].key = obbo_score_key(dets[i * stride + stride - 1]);
    ord[i].idx = i;
  }
  qsort(ord, (size_t)n, sizeof(key_idx), cmp_key_idx);
  for (i = 0; i < n; ++i) {
    const float *d = dets + ord[i].idx * stride;
    if (kind == 2) { sx[i] = d[0]; sy[i] = d[1]; sex[i] = d[2]; sey[i] = d[3]; sar[i] = hbb_area_f32(d); continue; }
    if (kind) obbo_quad_from_poly_f32(d, &Q[i]);
    else obbo_quad_from_rbox_f32(d, &Q[i]);
    sx[i] = Q[i].ox; sy[i] = Q[i].oy; sex[i] = Q[i].ex; sey[i] = Q[i].ey;
  }
  for (i = 0; i < n; ++i) {
    if (sup[i]) continue;
    const obbo_quad *A = &Q[i];
    float ax = sx[i], ay = sy[i], aex = sex[i], aey = sey[i];
    int32_t gi = group ? group[ord[i].idx] : 0;
    float ti = thrs ? thrs[ord[i].idx] : thr;
    for (j = i + 1; j < n; ++j) {
      if (sup[j]) continue;
      if (group && group[ord[j].idx] != gi) continue;
      float ovr;
      if (kind == 2) {
        const float a4[4] = {ax, ay, aex, aey}, b4[4] = {sx[j], sy[j], sex[j], sey[j]};
        ovr = hbb_overlap_f32(a4, sar[i], b4, sar[j]);
      } else {
        /* same prefilter as quad_overlap_f32, hoisted */
        float dx = sx[j] - ax;
        if (fabs_sel(dx) > aex + sex[j]) continue;
        float dy = sy[j] - ay;
        if (fabs_sel(dy) > aey + sey[j]) continue;
        ovr = quad_overlap_f32(A, &Q[j], 0);
      }
      if (strict ? (ovr > ti) : (ovr >= ti)) sup[j] = 1;
    }
  }
  /* ascending ORIGINAL index */
  uint8_t *kept = (uint8_t *)calloc((size_t)n, 1);
  if (!kept) { free(ord); free(Q); free(sup); free(sx); return -2; }
  for (i = 0; i < n; ++i) if (!sup[i]) kept[ord[i].idx] = 1;
  long k = 0;
  for (i = 0; i < n; ++i) if (kept[i]) keep[k++] = i;
  free(kept); free(ord); free(Q); free(sup); free(sx);
  return k;
}

/* the sorted order itself (tests compare it with the device radix sort) */
OBBO_API int obbo_score_order(const float *scores, long n, long stride, int64_t *order) {
  long i;
  key_idx *ord = (key_idx *)malloc(sizeof(key_idx) * (size_t)(n > 0 ? n : 1));
  if (!ord) return -2;
  for (i = 0; i < n; ++i) { ord[i].key = obbo_score_key(scores[i * stride]); ord[i].idx = i; }
  qsort(ord, (size_t)n, sizeof(key_idx), cmp_key_idx);
  for (i = 0; i < n; ++i) order[i] = ord[i].idx;
  free(ord);
  return 0;
}

/* ------------------------------------------------------------------ */
/* Rotated RoI-Align forward / backward                                */
/* ------------------------------------------------------------------ */
/* bilinear tap rules = roi_align_kernel_v2.cu:15-60 (identical in
 * roi_align_kernel.cu:17-62).  Returns the 4 tap offsets and weights;
 * weights are all zero for an out-of-map sample.                          */
static inline void bilinear_taps_f32(int height, int width, float y, float x,
                                     int *o1, int *o2, int *o3, int *o4,
                                     float *w1, float *w2, float *w3, float *w4) {
  if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) {
    *o1 = *o2 = *o3 = *o4 = 0;
    *w1 = *w2 = *w3 = *w4 = 0.0f;
    return;
  }
  if (y <= 0.0f) y = 0.0f;
  if (x <= 0.0f) x = 0.0f;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; }
  else y_high = y_low + 1;
  if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; }
  else x_high = x_low + 1;
  float ly = y - (float)y_low, lx = x - (float)x_low;
  float hy = 1.0f - ly, hx = 1.0f - lx;
  *o1 = y_low * width + x_low;  *o2 = y_low * width + x_high;
  *o3 = y_high * width + x_low; *o4 = y_high * width + x_high;
  *w1 = hy * hx; *w2 = hy * lx; *w3 = ly * hx; *w4 = ly * lx;
}

/* per-RoI sampling frame (Appendix B of SURVEY.md; theta = 0 reduces to
 * RoIAlignForwardV2 aligned=False, roi_align_kernel_v2.cu:79-102).        */
typedef struct {
  int b, gh, gw;
  float cx, cy, c, s, rw, rh, bin_w, bin_h, x0, y0;
} rroi_frame;

static void rroi_make_frame(const float *roi, float scale, int ph, int pw,
                            int sample_num, float min_size, rroi_frame *f) {
  f->b = (int)roi[0];
  f->cx = roi[1] * scale; f->cy = roi[2] * scale;
  float rw = roi[3] * scale, rh = roi[4] * scale;
  rw = max_sel(rw, min_size); rh = max_sel(rh, min_size);
  f->rw = rw; f->rh = rh;
  obbo_sincos_f32(roi[5], &f->s, &f->c);
  f->bin_h = rh / (float)ph; f->bin_w = rw / (float)pw;
  f->gh = sample_num > 0 ? sample_num : (int)ceilf(rh / (float)ph);
  f->gw = sample_num > 0 ? sample_num : (int)ceilf(rw / (float)pw);
  f->x0 = -0.5f * rw; f->y0 = -0.5f * rh;
}

static inline void rroi_sample_xy(const rroi_frame *f, int ph_i, int pw_i, int iy, int ix,
                                  float *x, float *y) {
  /* local offsets from the RoI centre, V2's expression order */
  float ly = f->y0 + (float)ph_i * f->bin_h;
  float t1 = ((float)iy + 0.5f) * f->bin_h;
  ly = ly + t1 / (float)f->gh;
  float lx = f->x0 + (float)pw_i * f->bin_w;
  float t2 = ((float)ix + 0.5f) * f->bin_w;
  lx = lx + t2 / (float)f->gw;
  float xc = lx * f->c, ys = ly * f->s;
  float xs = lx * f->s, yc = ly * f->c;
  float rx = xc - ys, ry = xs + yc;
  *x = f->cx + rx;
  *y = f->cy + ry;
}

OBBO_API int obbo_rroi_align_forward_f32(const float *feat, int B, int C, int H, int W,
                                         const float *rois, long K, int ph, int pw,
                                         float scale, int sample_num, float min_size,
                                         float *out) {
  long k;
  int c, i, j, iy, ix;
  for (k = 0; k < K; ++k) {
    rroi_frame f;
    rroi_make_frame(rois + k * 6, scale, ph, pw, sample_num, min_size, &f);
    if (f.b < 0 || f.b >= B) continue;   /* such a RoI reads / writes nothing: zeros (obbops.h); the caller pre-zeroes */
    int cnt_i = f.gh * f.gw; if (cnt_i < 1) cnt_i = 1;
    float cnt = (float)cnt_i;
    for (c = 0; c < C; ++c) {
      const float *plane = feat + ((long)f.b * C + c) * H * W;
      for (i = 0; i < ph; ++i) for (j = 0; j < pw; ++j) {
        float acc = 0.0f;
        for (iy = 0; iy < f.gh; ++iy) for (ix = 0; ix < f.gw; ++ix) {
          float x, y, w1, w2, w3, w4; int o1, o2, o3, o4;
          rroi_sample_xy(&f, i, j, iy, ix, &x, &y);
          bilinear_taps_f32(H, W, y, x, &o1, &o2, &o3, &o4, &w1, &w2, &w3, &w4);
          float v = w1 * plane[o1];
          v = v + w2 * plane[o2];
          v = v + w3 * plane[o3];
          v = v + w4 * plane[o4];
          acc = acc + v;
        }
        out[((k * C + c) * ph + i) * pw + j] = acc / cnt;
      }
    }
  }
  return 0;
}

/* backward w.r.t. features: scatter of grad*w/count (mirrors
 * roi_align_kernel_v2.cu:179-263); grad_in must be pre-zeroed. */
OBBO_API int obbo_rroi_align_backward_f32(const float *grad_out, int B, int C, int H, int W,
                                          const float *rois, long K, int ph, int pw,
                                          float scale, int sample_num, float min_size,
                                          float *grad_in) {
  long k;
  int c, i, j, iy, ix;
  for (k = 0; k < K; ++k) {
    rroi_frame f;
    rroi_make_frame(rois + k * 6, scale, ph, pw, sample_num, min_size, &f);
    if (f.b < 0 || f.b >= B) continue;   /* such a RoI reads / writes nothing: zeros (obbops.h); the caller pre-zeroes */
    int cnt_i = f.gh * f.gw; if (cnt_i < 1) cnt_i = 1;
    float cnt = (float)cnt_i;
    for (c = 0; c < C; ++c) {
      float *plane = grad_in + ((long)f.b * C + c) * H * W;
      for (i = 0; i < ph; ++i) for (j = 0; j < pw; ++j) {
        float g = grad_out[((k * C + c) * ph + i) * pw + j];
        for (iy = 0; iy < f.gh; ++iy) for (ix = 0; ix < f.gw; ++ix) {
          float x, y, w1, w2, w3, w4; int o1, o2, o3, o4;
          rroi_sample_xy(&f, i, j, iy, ix, &x, &y);
          bilinear_taps_f32(H, W, y, x, &o1, &o2, &o3, &o4, &w1, &w2, &w3, &w4);
          plane[o1] += g * w1 / cnt;
          plane[o2] += g * w2 / cnt;
          plane[o3] += g * w3 / cnt;
          plane[o4] += g * w4 / cnt;
        }
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* horizontal RoIAlign V1 / V2 restatement (anchor A4 of SURVEY 8c)     */
/* ------------------------------------------------------------------ */
/* version 1: roi_align_kernel.cu:64-117 ( +1 on the far edge, clamp >= 0 )
 * version 2: roi_align_kernel_v2.cu:62-128 (aligned flag)                  */
OBBO_API int obbo_roi_align_forward_f32(const float *feat, int B, int C, int H, int W,
                                        const float *rois, long K, int ph, int pw,
                                        float scale, int sample_num, int version,
                                        int aligned, float *out) {
  long k;
  int c, i, j, iy, ix;
  for (k = 0; k < K; ++k) {
    const float *r = rois + k * 5;
    int b = (int)r[0];
    if (b < 0 || b >= B) return -1;
    float sw, sh, rw, rh;
    if (version == 1) {
      sw = r[1] * scale; sh = r[2] * scale;
      float ew = (r[3] + 1.0f) * scale, eh = (r[4] + 1.0f) * scale;
      rw = max_sel(ew - sw, 0.0f); rh = max_sel(eh - sh, 0.0f);
    } else {
      float off = aligned ? 0.5f : 0.0f;
      sw = r[1] * scale - off; sh = r[2] * scale - off;
      float ew = r[3] * scale - off, eh = r[4] * scale - off;
      rw = ew - sw; rh = eh - sh;
      if (!aligned) { rw = max_sel(rw, 1.0f); rh = max_sel(rh, 1.0f); }
    }
    float bin_h = rh / (float)ph, bin_w = rw / (float)pw;
    int gh = sample_num > 0 ? sample_num : (int)ceilf(rh / (float)ph);
    int gw = sample_num > 0 ? sample_num : (int)ceilf(rw / (float)pw);
    float cnt;
    if (version == 1) cnt = (float)(gh * gw);
    else { int ci = gh * gw; if (ci < 1) ci = 1; cnt = (float)ci; }
    for (c = 0; c < C; ++c) {
      const float *plane = feat + ((long)b * C + c) * H * W;
      for (i = 0; i < ph; ++i) for (j = 0; j < pw; ++j) {
        float acc = 0.0f;
        for (iy = 0; iy < gh; ++iy) {
          float y = sh + (float)i * bin_h + ((float)iy + 0.5f) * bin_h / (float)gh;
          for (ix = 0; ix < gw; ++ix) {
            float x = sw + (float)j * bin_w + ((float)ix + 0.5f) * bin_w / (float)gw;
            float w1, w2, w3, w4; int o1, o2, o3, o4;
            bilinear_taps_f32(H, W, y, x, &o1, &o2, &o3, &o4, &w1, &w2, &w3, &w4);
            acc += w1 * plane[o1] + w2 * plane[o2] + w3 * plane[o3] + w4 * plane[o4];
          }
        }
        out[((k * C + c) * ph + i) * pw + j] = acc / cnt;
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* horizontal greedy NMS restatement of nms_cpu.cpp:5-60 (anchor A2)    */
/* ------------------------------------------------------------------ */
OBBO_API long obbo_hbb_nms_f32(const float *dets, long n, float thr, int64_t *keep) {
  long i, j;
  if (n <= 0) return 0;
  key_idx *ord = (key_idx *)malloc(sizeof(key_idx) * (size_t)n);
  float *areas = (float *)malloc(sizeof(float) * (size_t)n);
  uint8_t *sup = (uint8_t *)calloc((size_t)n, 1);
  if (!ord || !areas || !sup) { free(ord); free(areas); free(sup); return -2; }
  for (i = 0; i < n; ++i) {
    const float *d = dets + i * 5;
    areas[i] = (d[2] - d[0] + 1.0f) * (d[3] - d[1] + 1.0f);
    ord[i].key = obbo_score_key(d[4]); ord[i].idx = i;
  }
  qsort(ord, (size_t)n, sizeof(key_idx), cmp_key_idx);
  for (i = 0; i < n; ++i) {
    long a = ord[i].idx;
    if (sup[a]) continue;
    const float *da = dets + a * 5;
    for (j = i + 1; j < n; ++j) {
      long b = ord[j].idx;
      if (sup[b]) continue;
      const float *db = dets + b * 5;
      float xx1 = max_sel(da[0], db[0]), yy1 = max_sel(da[1], db[1]);
      float xx2 = min_sel(da[2], db[2]), yy2 = min_sel(da[3], db[3]);
      float w = max_sel(0.0f, xx2 - xx1 + 1.0f), h = max_sel(0.0f, yy2 - yy1 + 1.0f);
      float inter = w * h;
      float ovr = inter / (areas[a] + areas[b] - inter);
      if (ovr >= thr) sup[b] = 1;
    }
  }
  long k = 0;
  for (i = 0; i < n; ++i) if (!sup[i]) keep[k++] = i;
  free(ord); free(areas); free(sup);
  return k;
}

/* ------------------------------------------------------------------ */
/* horizontal bbox_overlaps restatement of geometry.py:47-88 (fp32)     */
/* ------------------------------------------------------------------ */
/* torch.max / torch.min / clamp propagate NaN */
static inline float nan_max(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a < b ? b : a)); }
static inline float nan_min(float a, float b) { return (a != a) ? a : ((b != b) ? b : (b < a ? b : a)); }
static inline float hbb_pair_overlap_f32(const float *a, const float *b, int mode) {
  float ltx = nan_max(a[0], b[0]), lty = nan_max(a[1], b[1]);       /* lt = max(b1[:, :2], b2[:, :2])  :61,75 */
  float rbx = nan_min(a[2], b[2]), rby = nan_min(a[3], b[3]);       /* rb = min(b1[:, 2:], b2[:, 2:])          */
  float w = nan_max((rbx - ltx) + 1.0f, 0.0f), h = nan_max((rby - lty) + 1.0f, 0.0f);   /* (rb - lt + 1).clamp(min=0) */
  float overlap = w * h;
  float area1 = hbb_area_f32(a);
  if (mode) return overlap / area1;                                  /* iof :72,86 */
  return overlap / ((area1 + hbb_area_f32(b)) - overlap);            /* iou :70,84 */
}

OBBO_API int obbo_hbb_overlaps_f32(const float *a, long m, const float *b, long n, int mode,
                                   int aligned, float *out) {
  long i, j;
  if (m < 0 || n < 0 || (aligned && m != n)) return -1;
  for (i = 0; i < m; ++i) {
    if (aligned) out[i] = hbb_pair_overlap_f32(a + i * 4, b + i * 4, mode);
    else for (j = 0; j < n; ++j) out[i * n + j] = hbb_pair_overlap_f32(a + i * 4, b + j * 4, mode);
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* Soft-NMS restatement of nms_cpu.cpp:71-210 (fp32 instantiation)      */
/* ------------------------------------------------------------------ */
/* dets (n,5) = x1,y1,x2,y2,score.  method 0 = hard (weight 0 when ovr > thr), 1 = linear
 * (1 - ovr when ovr > thr), 2 = gaussian exp(-ovr^2 / sigma)  (nms_cpu.cpp:167-181).
 * Selection by repeated arg-max over the not-yet-selected tail with STRICT `<`, i.e. the
 * first of equal scores wins (:114-121); a rescored box below min_score is removed by
 * moving the LAST box into its slot, which is then examined again (:184-198).
 * out (n,6) receives x1,y1,x2,y2,new score,original index for every surviving box in
 * selection order (:200-209).  Returns the number of rows.  method 2 calls expf(): results
 * on other libms may differ in the last ulp of the score (the hard/linear methods use
 * only + - * / and comparisons).                                                       */
OBBO_API long obbo_soft_nms_f32(const float *dets, long n, float thr, int method, float sigma,
                                float min_score, float *out) {
  long i, pos;
  if (n <= 0) return 0;
  float *x1 = (float *)malloc(sizeof(float) * (size_t)n * 7);
  if (!x1) return -2;
  float *y1 = x1 + n, *x2 = x1 + 2 * n, *y2 = x1 + 3 * n, *sc = x1 + 4 * n, *ar = x1 + 5 * n, *id = x1 + 6 * n;
  for (i = 0; i < n; ++i) {
    const float *d = dets + i * 5;
    x1[i] = d[0]; y1[i] = d[1]; x2[i] = d[2]; y2[i] = d[3]; sc[i] = d[4];
    ar[i] = hbb_area_f32(d);
    id[i] = (float)i;   /* the reference keeps the indices in the dets' dtype (arange(ndets, dets.options())) */
  }
  long nd = n;
  for (i = 0; i < nd; ++i) {
    float max_score = sc[i];
    long max_pos = i;
    for (pos = i + 1; pos < nd; ++pos)
      if (max_score < sc[pos]) { max_score = sc[pos]; max_pos = pos; }
    float t;
#define OBBO_SWAP(a) t = a[i]; a[i] = a[max_pos]; a[max_pos] = t
    OBBO_SWAP(x1); OBBO_SWAP(y1); OBBO_SWAP(x2); OBBO_SWAP(y2); OBBO_SWAP(sc); OBBO_SWAP(ar); OBBO_SWAP(id);
#undef OBBO_SWAP
    const float a4[4] = {x1[i], y1[i], x2[i], y2[i]};
    const float iarea = ar[i];
    pos = i + 1;
    while (pos < nd) {
      const float b4[4] = {x1[pos], y1[pos], x2[pos], y2[pos]};
      float ovr = hbb_overlap_f32(a4, iarea, b4, ar[pos]);
      float weight = 1.0f;
      if (method == 1) { if (ovr > thr) weight = 1.0f - ovr; }
      else if (method == 2) weight = expf(-(ovr * ovr) / sigma);
      else weight = (ovr > thr) ? 0.0f : 1.0f;
      sc[pos] = weight * sc[pos];
      if (sc[pos] < min_score) {
        x1[pos] = x1[nd - 1]; y1[pos] = y1[nd - 1]; x2[pos] = x2[nd - 1]; y2[pos] = y2[nd - 1];
        sc[pos] = sc[nd - 1]; ar[pos] = ar[nd - 1]; id[pos] = id[nd - 1];
        nd = nd - 1;
        pos = pos - 1;
      }
      pos = pos + 1;
    }
  }
  for (i = 0; i < nd; ++i) {
    float *o = out + i * 6;
    o[0] = x1[i]; o[1] = y1[i]; o[2] = x2[i]; o[3] = y2[i]; o[4] = sc[i]; o[5] = id[i];
  }
  free(x1);
  return nd;
}

/* ------------------------------------------------------------------ */
/* the same two loops for float64 boxes: nms_cpu_kernel<double> /       */
/* soft_nms_cpu_kernel<double> (AT_DISPATCH_FLOATING_TYPES,             */
/* nms_cpu.cpp:62-68, :212-215).  The threshold stays a FLOAT in the    */
/* reference's signature (`const float threshold`, :5,:71) and is       */
/* promoted in the comparison (:56, :170).                              */
/* ------------------------------------------------------------------ */
typedef struct { uint64_t key; long idx; } key64_idx;
static int cmp_key64_idx(const void *a, const void *b) {
  const key64_idx *x = (const key64_idx *)a, *y = (const key64_idx *)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
static uint64_t obbo_score_key64(double s) {
  double t = s + 0.0;
  uint64_t u;
  memcpy(&u, &t, 8);
  u = (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
  return ~u;
}
static inline double dmax_sel(double a, double b) { return a < b ? b : a; }
static inline double dmin_sel(double a, double b) { return b < a ? b : a; }

OBBO_API long obbo_hbb_nms_f64(const double *dets, long n, float thr, int64_t *keep) {
  long i, j;
  if (n <= 0) return 0;
  key64_idx *ord = (key64_idx *)malloc(sizeof(key64_idx) * (size_t)n);
  double *areas = (double *)malloc(sizeof(double) * (size_t)n);
  uint8_t *sup = (uint8_t *)calloc((size_t)n, 1);
  if (!ord || !areas || !sup) { free(ord); free(areas); free(sup); return -2; }
  for (i = 0; i < n; ++i) {
    const double *d = dets + i * 5;
    areas[i] = (d[2] - d[0] + 1.0) * (d[3] - d[1] + 1.0);
    ord[i].key = obbo_score_key64(d[4]); ord[i].idx = i;
  }
  qsort(ord, (size_t)n, sizeof(key64_idx), cmp_key64_idx);
  const double thrd = (double)thr;
  for (i = 0; i < n; ++i) {
    long a = ord[i].idx;
    if (sup[a]) continue;
    const double *da = dets + a * 5;
    for (j = i + 1; j < n; ++j) {
      long b = ord[j].idx;
      if (sup[b]) continue;
      const double *db = dets + b * 5;
      double xx1 = dmax_sel(da[0], db[0]), yy1 = dmax_sel(da[1], db[1]);
      double xx2 = dmin_sel(da[2], db[2]), yy2 = dmin_sel(da[3], db[3]);
      double w = dmax_sel(0.0, xx2 - xx1 + 1.0), h = dmax_sel(0.0, yy2 - yy1 + 1.0);
      double inter = w * h;
      double ovr = inter / (areas[a] + areas[b] - inter);
      if (ovr >= thrd) sup[b] = 1;
    }
  }
  long k = 0;
  for (i = 0; i < n; ++i) if (!sup[i]) keep[k++] = i;
  free(ord); free(areas); free(sup);
  return k;
}

OBBO_API long obbo_soft_nms_f64(const double *dets, long n, float thr, int method, float sigma,
                                float min_score, double *out) {
  long i, pos;
  if (n <= 0) return 0;
  double *x1 = (double *)malloc(sizeof(double) * (size_t)n * 7);
  if (!x1) return -2;
  double *y1 = x1 + n, *x2 = x1 + 2 * n, *y2 = x1 + 3 * n, *sc = x1 + 4 * n, *ar = x1 + 5 * n, *id = x1 + 6 * n;
  for (i = 0; i < n; ++i) {
    const double *d = dets + i * 5;
    x1[i] = d[0]; y1[i] = d[1]; x2[i] = d[2]; y2[i] = d[3]; sc[i] = d[4];
    ar[i] = (d[2] - d[0] + 1.0) * (d[3] - d[1] + 1.0);
    id[i] = (double)i;
  }
  long nd = n;
  for (i = 0; i < nd; ++i) {
    double max_score = sc[i];
    long max_pos = i;
    for (pos = i + 1; pos < nd; ++pos)
      if (max_score < sc[pos]) { max_score = sc[pos]; max_pos = pos; }
    double t;
#define OBBO_SWAP(a) t = a[i]; a[i] = a[max_pos]; a[max_pos] = t
    OBBO_SWAP(x1); OBBO_SWAP(y1); OBBO_SWAP(x2); OBBO_SWAP(y2); OBBO_SWAP(sc); OBBO_SWAP(ar); OBBO_SWAP(id);
#undef OBBO_SWAP
    const double ix1 = x1[i], iy1 = y1[i], ix2 = x2[i], iy2 = y2[i], iarea = ar[i];
    pos = i + 1;
    while (pos < nd) {
      double xx1 = dmax_sel(ix1, x1[pos]), yy1 = dmax_sel(iy1, y1[pos]);
      double xx2 = dmin_sel(ix2, x2[pos]), yy2 = dmin_sel(iy2, y2[pos]);
      double w = dmax_sel(0.0, xx2 - xx1 + 1.0), h = dmax_sel(0.0, yy2 - yy1 + 1.0);
      double inter = w * h;
      double ovr = inter / (iarea + ar[pos] - inter);
      double weight = 1.0;
      if (method == 1) { if (ovr > (double)thr) weight = 1.0 - ovr; }
      else if (method == 2) weight = exp(-(ovr * ovr) / (double)sigma);
      else weight = (ovr > (double)thr) ? 0.0 : 1.0;
      sc[pos] = weight * sc[pos];
      if (sc[pos] < (double)min_score) {
        x1[pos] = x1[nd - 1]; y1[pos] = y1[nd - 1]; x2[pos] = x2[nd - 1]; y2[pos] = y2[nd - 1];
        sc[pos] = sc[nd - 1]; ar[pos] = ar[nd - 1]; id[pos] = id[nd - 1];
        nd = nd - 1;
        pos = pos - 1;
      }
      pos = pos + 1;
    }
  }
  for (i = 0; i < nd; ++i) {
    double *o = out + i * 6;
    o[0] = x1[i]; o[1] = y1[i]; o[2] = x2[i]; o[3] = y2[i]; o[4] = sc[i]; o[5] = id[i];
  }
  free(x1);
  return nd;
}

/* ------------------------------------------------------------------ */
/* fp64 triangle-fan polygon IoU ("polyiou")                            */
/* ------------------------------------------------------------------ */
/* THIRD, independent formulation of the rotated overlap: the algorithm published with the
 * DOTA development kit (DOTA_devkit/polyiou.cpp [external, not in the reference tree] -- the
 * routine the absent `wwtool.mergebypoly` is recorded as wrapping): both
 * polygons are fanned into triangles from the coordinate origin, every pair of triangles
 * (o,a,b) x (o,c,d) is intersected by cutting the first with the three half-planes of the
 * second, and the SIGNED areas are summed.  No convexity assumption, no shared code with
 * the two routines above.  eps = 1e-8 as published.                                      */
typedef struct { double x, y; } pt_d;
static int sig_d(double d) { const double eps = 1e-8; return (d > eps) - (d < -eps); }
static double cross3_d(pt_d o, pt_d a, pt_d b) { return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y); }
static double fan_area_d(const pt_d *ps, int n) {
  double r = 0.0; int i;
  for (i = 0; i < n; ++i) r += ps[i].x * ps[(i + 1) % n].y - ps[i].y * ps[(i + 1) % n].x;
  return r / 2.0;
}
static int line_cross_d(pt_d a, pt_d b, pt_d c, pt_d d, pt_d *p) {
  double s1 = cross3_d(a, b, c), s2 = cross3_d(a, b, d);
  if (sig_d(s1) == 0 && sig_d(s2) == 0) return 2;
  if (sig_d(s2 - s1) == 0) return 0;
  p->x = (c.x * s2 - d.x * s1) / (s2 - s1);
  p->y = (c.y * s2 - d.y * s1) / (s2 - s1);
  return 1;
}
static int pt_eq_d(pt_d a, pt_d b) { return sig_d(a.x - b.x) == 0 && sig_d(a.y - b.y) == 0; }
/* keep the part of polygon p (n vertices) on the left of a->b */
static void polygon_cut_d(pt_d *p, int *n, pt_d a, pt_d b) {
  pt_d pp[16];
  int m = 0, i;
  p[*n] = p[0];
  for (i = 0; i < *n; ++i) {
    if (sig_d(cross3_d(a, b, p[i])) > 0) pp[m++] = p[i];
    if (sig_d(cross3_d(a, b, p[i])) != sig_d(cross3_d(a, b, p[i + 1])))
      line_cross_d(a, b, p[i], p[i + 1], &pp[m++]);
  }
  /* vertex equality as published: Point::operator== compares through sig() (|dx| <= eps and |dy| <= eps), not
   * exactly -- crossing points that differ by rounding only are one vertex (an exact `==` stood here) */
  *n = 0;
  for (i = 0; i < m; ++i)
    if (!i || !pt_eq_d(pp[i], pp[i - 1])) p[(*n)++] = pp[i];
  while (*n > 1 && pt_eq_d(p[*n - 1], p[0])) (*n)--;
}
static double tri_inter_d(pt_d a, pt_d b, pt_d c, pt_d d) {
  pt_d o = {0.0, 0.0};
  int s1 = sig_d(cross3_d(o, a, b)), s2 = sig_d(cross3_d(o, c, d));
  if (s1 == 0 || s2 == 0) return 0.0;
  if (s1 == -1) { pt_d t = a; a = b; b = t; }
  if (s2 == -1) { pt_d t = c; c = d; d = t; }
  pt_d p[16] = {o, a, b};
  int n = 3;
  polygon_cut_d(p, &n, o, c);
  polygon_cut_d(p, &n, c, d);
  polygon_cut_d(p, &n, d, o);
  double res = fabs(fan_area_d(p, n));
  if (s1 * s2 == -1) res = -res;
  return res;
}
static double poly_inter_fan_d(const pt_d *ps1, int n1, const pt_d *ps2, int n2) {
  pt_d a[8], b[8];
  int i, j;
  for (i = 0; i < n1; ++i) a[i] = ps1[i];
  for (i = 0; i < n2; ++i) b[i] = ps2[i];
  if (fan_area_d(a, n1) < 0) for (i = 0; i < n1 / 2; ++i) { pt_d t = a[i]; a[i] = a[n1 - 1 - i]; a[n1 - 1 - i] = t; }
  if (fan_area_d(b, n2) < 0) for (i = 0; i < n2 / 2; ++i) { pt_d t = b[i]; b[i] = b[n2 - 1 - i]; b[n2 - 1 - i] = t; }
  a[n1] = a[0]; b[n2] = b[0];
  double res = 0.0;
  for (i = 0; i < n1; ++i)
    for (j = 0; j < n2; ++j) res += tri_inter_d(a[i], a[i + 1], b[j], b[j + 1]);
  return res;
}

/* aligned pairs: a, b (n, 5) theta-OBBs (kind 0) or (n, 8) polygons (kind 1), fp64 in,
 * iou[i] out.  The fan is taken from the coordinate origin as published; `recentre` != 0
 * first translates both polygons so that a's first vertex is the origin (the fan's signed
 * triangles then stay small -- at DOTA coordinates ~2e4 the published form loses ~1e-7
 * of the area to cancellation, which is what this flag lets a test quantify).            */
OBBO_API int obbo_polyiou_f64(const double *a, const double *b, long n, int kind, int recentre,
                              double *iou) {
  long i;
  int k, stride = kind ? 8 : 5;
  for (i = 0; i < n; ++i) {
    double ax[4], ay[4], bx[4], by[4];
    pt_d pa[5], pb[5];
    if (kind) { for (k = 0; k < 4; ++k) { ax[k] = a[i * 8 + 2 * k]; ay[k] = a[i * 8 + 2 * k + 1]; bx[k] = b[i * 8 + 2 * k]; by[k] = b[i * 8 + 2 * k + 1]; } }
    else { rbox_corners_d(a + i * stride, ax, ay); rbox_corners_d(b + i * stride, bx, by); }
    double ox = recentre ? ax[0] : 0.0, oy = recentre ? ay[0] : 0.0;
    for (k = 0; k < 4; ++k) { pa[k].x = ax[k] - ox; pa[k].y = ay[k] - oy; pb[k].x = bx[k] - ox; pb[k].y = by[k] - oy; }
    double aa = fabs(fan_area_d(pa, 4)), ab = fabs(fan_area_d(pb, 4));
    double inter = poly_inter_fan_d(pa, 4, pb, 4);
    double uni = aa + ab - inter;
    iou[i] = uni > 0 ? inter / uni : 0.0;
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* threaded legs for the CPU baseline (bench.py "all cores")            */
/* ------------------------------------------------------------------ */
/* Same arithmetic, rows / RoIs dealt to OpenMP threads.  Built with -fopenmp; without it
 * the pragmas vanish and these are the serial routines.                                  */
#ifdef _OPENMP
#include <omp.h>
#endif
OBBO_API int obbo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

OBBO_API int obbo_overlaps_mt_f32(const float *a, long m, const float *b, long n,
                                  int kind, int mode, int threads, float *out) {
  long i;
  int stride = kind ? 8 : 5;
  if (m < 0 || n < 0) return -1;
  obbo_quad *QB = (obbo_quad *)malloc(sizeof(obbo_quad) * (size_t)(n > 0 ? n : 1));
  if (!QB) return -2;
  for (i = 0; i < n; ++i) {
    if (kind) obbo_quad_from_poly_f32(b + i * stride, &QB[i]);
    else obbo_quad_from_rbox_f32(b + i * stride, &QB[i]);
  }
  (void)threads;
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
  for (i = 0; i < m; ++i) {
    obbo_quad A;
    long j;
    if (kind) obbo_quad_from_poly_f32(a + i * stride, &A);
    else obbo_quad_from_rbox_f32(a + i * stride, &A);
    for (j = 0; j < n; ++j) out[i * n + j] = quad_overlap_f32(&A, &QB[j], mode);
  }
  free(QB);
  return 0;
}

OBBO_API int obbo_rroi_align_forward_mt_f32(const float *feat, int B, int C, int H, int W,
                                            const float *rois, long K, int ph, int pw,
                                            float scale, int sample_num, float min_size,
                                            int threads, float *out) {
  long k;
  (void)threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
  for (k = 0; k < K; ++k)
    obbo_rroi_align_forward_f32(feat, B, C, H, W, rois + k * 6, 1, ph, pw, scale, sample_num, min_size,
                                out + k * (long)C * ph * pw);
  return 0;
}

/* rotated NMS with the inner (suppression) loop of one kept box dealt to threads: the
 * outer loop is the serial dependency of nms_cpu.cpp:35-58, the inner one is independent
 * per j.  Same keep set as obbo_nms_f32 by construction (each j is decided by the same
 * kept i in the same order).                                                            */
OBBO_API long obbo_nms_mt_f32(const float *dets, long n, float thr, int threads, int64_t *keep) {
  long i;
  if (n <= 0) return 0;
  key_idx *ord = (key_idx *)malloc(sizeof(key_idx) * (size_t)n);
  obbo_quad *Q = (obbo_quad *)malloc(sizeof(obbo_quad) * (size_t)n);
  uint8_t *sup = (uint8_t *)calloc((size_t)n, 1);
  if (!ord || !Q || !sup) { free(ord); free(Q); free(sup); return -2; }
  for (i = 0; i < n; ++i) { ord[i].key = obbo_score_key(dets[i * 6 + 5]); ord[i].idx = i; }
  qsort(ord, (size_t)n, sizeof(key_idx), cmp_key_idx);
  for (i = 0; i < n; ++i) obbo_quad_from_rbox_f32(dets + ord[i].idx * 6, &Q[i]);
  (void)threads;
#pragma omp parallel num_threads(threads)
  {
    long ii, j;
    for (ii = 0; ii < n; ++ii) {
      /* sup[ii] is final here: every i < ii has finished (barrier below) */
      if (!sup[ii]) {
        const obbo_quad *A = &Q[ii];
#pragma omp for schedule(static) nowait
        for (j = ii + 1; j < n; ++j) {
          if (sup[j]) continue;
          float dx = Q[j].ox - A->ox;
          if (fabs_sel(dx) > A->ex + Q[j].ex) continue;
          float dy = Q[j].oy - A->oy;
          if (fabs_sel(dy) > A->ey + Q[j].ey) continue;
          if (quad_overlap_f32(A, &Q[j], 0) >= thr) sup[j] = 1;
        }
      }
#pragma omp barrier
    }
  }
  uint8_t *kept = (uint8_t *)calloc((size_t)n, 1);
  for (i = 0; i < n; ++i) if (!sup[i]) kept[ord[i].idx] = 1;
  long k = 0;
  for (i = 0; i < n; ++i) if (kept[i]) keep[k++] = i;
  free(kept); free(ord); free(Q); free(sup);
  return k;
}

OBBO_API const char *obbo_version(void) { return "obb-oracle 2 (OBB-GEOM v1)"; }

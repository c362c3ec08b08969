// obb_geom.h -- device geometry core ("OBB-GEOM v1") for gfx950.
//
// Register-only, branch-free convex quad x convex quad intersection for wave64:
// the subject is kept as up to 8 boundary SEGMENTS with compile-time indices
// (no per-lane vertex list, so no scratch memory and no LDS), every clip step
// is predicated selects.  The fp32 operation order is fixed (compile with
// -ffp-contract=off; '/' is the correctly rounded v_div_* sequence) so that
// keep decisions reproduce on any IEEE-754 machine running the same sequence.
//
// theta-OBB meaning: cv2.boxPoints, as used by the reference
// (mmdet/core/rbbox/transforms.py:45-55):
//   x = cx + lx*cos(t) - ly*sin(t),  y = cy + lx*sin(t) + ly*cos(t)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace obb {

// 48-byte per-box record.  origin = box centre (rbox) or vertex 0 (poly);
// r* = the 4 vertices relative to the origin, positive orientation;
// ex,ey = AABB half extents about the origin.
struct Quad {
  float ox, oy;
  float rx[4], ry[4];
  float ex, ey;
  float area;
};

__device__ __forceinline__ float fabs_sel(float a) { return a < 0.0f ? -a : a; }
__device__ __forceinline__ float max_sel(float a, float b) { return a < b ? b : a; }
__device__ __forceinline__ float min_sel(float a, float b) { return b < a ? b : a; }

// Deterministic sin/cos: Cody-Waite reduction by pi/2 + minimax polynomials,
// only + - * floor.  (OCML sinf/cosf and glibc differ in the last ulp; a keep
// decision `iou >= thr` must not depend on the math library.)
__device__ __forceinline__ void sincos_det(float x, float &s, float &c) {
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
  float k4 = floorf(kf * 0.25f);
  float kq = kf - 4.0f * k4;
  bool q0 = kq == 0.0f, q1 = kq == 1.0f, q2 = kq == 2.0f;
  s = q0 ? sp : (q1 ? cp : (q2 ? -sp : -cp));
  c = q0 ? cp : (q1 ? -sp : (q2 ? -cp : sp));
}

__device__ __forceinline__ void quad_from_rbox(float cx, float cy, float w, float h,
                                               float th, Quad &q) {
  float s, c;
  sincos_det(th, s, c);
  float hw = 0.5f * w, hh = 0.5f * h;
  float ux = hw * c, uy = hw * s;
  float t = hh * s;
  float vx = -t, vy = hh * c;
  q.ox = cx; q.oy = cy;
  float mx = -ux, my = -uy;
  q.rx[0] = mx - vx; q.ry[0] = my - vy;
  q.rx[1] = ux - vx; q.ry[1] = uy - vy;
  q.rx[2] = ux + vx; q.ry[2] = uy + vy;
  q.rx[3] = mx + vx; q.ry[3] = my + vy;
  q.ex = fabs_sel(ux) + fabs_sel(vx);
  q.ey = fabs_sel(uy) + fabs_sel(vy);
  q.area = w * h;
}

__device__ __forceinline__ void quad_from_poly(const float p[8], Quad &q) {
  float ox = p[0], oy = p[1];
  float x1 = p[2] - ox, y1 = p[3] - oy;
  float x2 = p[4] - ox, y2 = p[5] - oy;
  float x3 = p[6] - ox, y3 = p[7] - oy;
  float a = x1 * y2 - x2 * y1;
  float b2 = x2 * y3 - x3 * y2;
  float sh = a + b2;
  bool neg = sh < 0.0f;
  q.ox = ox; q.oy = oy;
  q.rx[0] = 0.0f; q.ry[0] = 0.0f;
  q.rx[2] = x2; q.ry[2] = y2;
  q.rx[1] = neg ? x3 : x1; q.ry[1] = neg ? y3 : y1;
  q.rx[3] = neg ? x1 : x3; q.ry[3] = neg ? y1 : y3;
  sh = neg ? -sh : sh;
  q.ex = max_sel(max_sel(fabs_sel(x1), fabs_sel(x2)), fabs_sel(x3));
  q.ey = max_sel(max_sel(fabs_sel(y1), fabs_sel(y2)), fabs_sel(y3));
  q.area = 0.5f * sh;
}

// Load one box row (rbox: 5 floats, poly: 8 floats, hbb: 4 floats).
template <int KIND>
__device__ __forceinline__ void load_quad(const float *__restrict__ row, Quad &q) {
  if (KIND == 0) {
    quad_from_rbox(row[0], row[1], row[2], row[3], row[4], q);
  } else if (KIND == 2) {
    // horizontal box x1,y1,x2,y2 with the reference's +1 convention = the theta-0 OBB below
    // (same fp32 operations, in the same order, as the torch expression it replaces)
    const float x1 = row[0], y1 = row[1], x2 = row[2], y2 = row[3];
    quad_from_rbox((x1 + x2) * 0.5f, (y1 + y2) * 0.5f, (x2 - x1) + 1.0f, (y2 - y1) + 1.0f, 0.0f, q);
  } else {
    float p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = row[k];
    quad_from_poly(p, q);
  }
}

// Intersection area of convex quads A and B; (dx,dy) = B.origin - A.origin.
// Mirrors oracle/obb_oracle.c quad_inter_f32 operation for operation.
__device__ __forceinline__ float quad_inter(const Quad &A, const Quad &B, float dx, float dy) {
  float sax[8], say[8], sbx[8], sby[8];
  bool valid[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    sax[k] = A.rx[k]; say[k] = A.ry[k];
    sbx[k] = A.rx[(k + 1) & 3]; sby[k] = A.ry[(k + 1) & 3];
    valid[k] = true;
  }
#pragma unroll
  for (int k = 4; k < 8; ++k) {
    sax[k] = 0.0f; say[k] = 0.0f; sbx[k] = 0.0f; sby[k] = 0.0f;
    valid[k] = false;
  }
  float qx[4], qy[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    qx[k] = dx + B.rx[k];
    qy[k] = dy + B.ry[k];
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    float px = qx[m], py = qy[m];
    float ex = qx[(m + 1) & 3] - px, ey = qy[(m + 1) & 3] - py;
    float nsx = 0.0f, nsy = 0.0f, nex = 0.0f, ney = 0.0f;
    bool has_leave = false, has_enter = false;
#pragma unroll
    for (int k = 0; k < 4 + m; ++k) {
      float ta = say[k] - py, tb = sax[k] - px;
      float sa = ex * ta - ey * tb;
      float tc = sby[k] - py, td = sbx[k] - px;
      float sb = ex * tc - ey * td;
      bool ina = sa >= 0.0f, inb = sb >= 0.0f;
      bool cross = valid[k] && (ina != inb);
      bool drop = valid[k] && !ina && !inb;
      float den = sa - sb;
      float t = sa / den;
      float ddx = sbx[k] - sax[k], ddy = sby[k] - say[k];
      float X = sax[k] + t * ddx;
      float Y = say[k] + t * ddy;
      bool leave = cross && ina;
      bool enter = cross && !ina;
      sbx[k] = leave ? X : sbx[k]; sby[k] = leave ? Y : sby[k];
      sax[k] = enter ? X : sax[k]; say[k] = enter ? Y : say[k];
      nsx = leave ? X : nsx; nsy = leave ? Y : nsy;
      nex = enter ? X : nex; ney = enter ? Y : ney;
      has_leave = has_leave || leave;
      has_enter = has_enter || enter;
      valid[k] = valid[k] && !drop;
    }
    sax[4 + m] = nsx; say[4 + m] = nsy;
    sbx[4 + m] = nex; sby[4 + m] = ney;
    valid[4 + m] = has_leave && has_enter;
  }
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float c1 = sax[k] * sby[k];
    float c2 = sbx[k] * say[k];
    float cr = c1 - c2;
    acc = acc + (valid[k] ? cr : 0.0f);
  }
  return 0.5f * acc;
}

// the spec's AABB prefilter: true = provably disjoint
__device__ __forceinline__ bool aabb_reject(float aox, float aoy, float aex, float aey,
                                            float box, float boy, float bex, float bey) {
  float dx = box - aox, dy = boy - aoy;
  float lox = aex + bex, loy = aey + bey;
  return (fabs_sel(dx) > lox) || (fabs_sel(dy) > loy);
}

// overlap of two quad records; mode 0 = iou, 1 = iof (inter / area(A)).  `rej` = the spec's
// AABB prefilter fired (the value is then an exact 0 and, in NMS, the pair does not interact
// whatever the threshold: the oracle `continue`s, obb_oracle.c obbo_nms_ex_f32).
__device__ __forceinline__ float quad_overlap(const Quad &A, const Quad &B, int mode, bool &rej) {
  float dx = B.ox - A.ox, dy = B.oy - A.oy;
  float lox = A.ex + B.ex, loy = A.ey + B.ey;
  rej = (fabs_sel(dx) > lox) || (fabs_sel(dy) > loy);
  float inter = quad_inter(A, B, dx, dy);
  float lim = min_sel(A.area, B.area);
  inter = min_sel(inter, lim);
  inter = max_sel(inter, 0.0f);
  float su = A.area + B.area;
  float den = (mode == 0) ? (su - inter) : A.area;
  bool ok = den > 0.0f;
  float v = inter / den;
  return (rej || !ok) ? 0.0f : v;
}

__device__ __forceinline__ float quad_overlap(const Quad &A, const Quad &B, int mode) {
  bool rej;
  return quad_overlap(A, B, mode, rej);
}

// The reference's HORIZONTAL pair overlap in its own operation order (nms_cpu.cpp:19,47-55):
// a / b = x1, y1, x2, y2; areas = ((x2 - x1) + 1) * ((y2 - y1) + 1).  max_sel / min_sel have
// std::max / std::min's operand order, so NaNs travel the same way.
__device__ __forceinline__ float hbb_area(float x1, float y1, float x2, float y2) {
  return ((x2 - x1) + 1.0f) * ((y2 - y1) + 1.0f);
}
__device__ __forceinline__ float hbb_overlap(const float (&a)[4], float aarea, const float (&b)[4], float barea) {
  const float xx1 = max_sel(a[0], b[0]), yy1 = max_sel(a[1], b[1]);
  const float xx2 = min_sel(a[2], b[2]), yy2 = min_sel(a[3], b[3]);
  const float w = max_sel(0.0f, (xx2 - xx1) + 1.0f), h = max_sel(0.0f, (yy2 - yy1) + 1.0f);
  const float inter = w * h;
  return inter / ((aarea + barea) - inter);
}

// The reference's horizontal IoU / IoF of bbox_overlaps (mmdet/core/bbox/geometry.py:61-86) in its
// tensor-op order: wh = clamp((min(rb) - max(lt)) + 1, min = 0); overlap = w * h;
// area = ((x2 - x1) + 1) * ((y2 - y1) + 1); iou = overlap / ((area1 + area2) - overlap),
// iof = overlap / area1.  torch.max / min / clamp propagate NaN: so do nan_max / nan_min.
__device__ __forceinline__ float nan_max(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a < b ? b : a)); }
__device__ __forceinline__ float nan_min(float a, float b) { return (a != a) ? a : ((b != b) ? b : (b < a ? b : a)); }
__device__ __forceinline__ float hbb_pair_overlap(const float *__restrict__ a, const float *__restrict__ b, int mode) {
  const float ltx = nan_max(a[0], b[0]), lty = nan_max(a[1], b[1]);
  const float rbx = nan_min(a[2], b[2]), rby = nan_min(a[3], b[3]);
  const float w = nan_max((rbx - ltx) + 1.0f, 0.0f), h = nan_max((rby - lty) + 1.0f, 0.0f);
  const float overlap = w * h;
  const float area1 = hbb_area(a[0], a[1], a[2], a[3]);
  if (mode != 0) return overlap / area1;
  const float area2 = hbb_area(b[0], b[1], b[2], b[3]);
  return overlap / ((area1 + area2) - overlap);
}
// a horizontal box whose pairs the interval prefilter may decide: positive +1 extents, finite
__device__ __forceinline__ bool hbb_regular(const float *__restrict__ r) {
  const float big = 3.0e38f;
  return ((r[2] - r[0]) + 1.0f > 0.0f) && ((r[3] - r[1]) + 1.0f > 0.0f) && (fabs_sel(r[0]) < big) &&
         (fabs_sel(r[1]) < big) && (fabs_sel(r[2]) < big) && (fabs_sel(r[3]) < big);
}

// order-preserving score key: ascending key order == descending score,
// -0 folded into +0 (oracle: obbo_score_key).
__device__ __forceinline__ uint32_t score_key(float s) {
  float t = s + 0.0f;
  uint32_t u = __float_as_uint(t);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ~u;
}

}  // namespace obb

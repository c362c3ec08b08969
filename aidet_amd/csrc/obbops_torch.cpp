// obbops_torch -- ATen-level binding of the libobbops C ABI (include/obbops.h).
//
// The reference binds its native ops as pybind11 modules that take and return
// at::Tensor and are imported next to the Python wrappers
// (mmdet/ops/nms/src/nms_cuda.cpp:8-17 `at::Tensor nms(const at::Tensor&, float)`,
// mmdet/ops/roi_align/src/roi_align_cuda.cpp:37-85, imported at
// mmdet/ops/nms/nms_wrapper.py:4 and mmdet/ops/roi_align/roi_align.py:6).  This module
// is that layer for the rotated ops: tensors in, tensors out, current stream and device
// of the input (nms_kernel.cu:74,99), failed checks -> RuntimeError (TORCH_CHECK), all
// compute in the HIP kernels behind the C ABI.  Host-only C++: no kernels here.
#include <torch/extension.h>

#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <map>
#include <tuple>
#include <utility>

#include "obbops.h"

namespace {

using c10::hip::HIPGuardMasqueradingAsCUDA;

void *current_stream(const at::Tensor &t) {
  return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
}

void check_rc(int rc, const char *what) {
  TORCH_CHECK(rc == OBB_OK, what, ": ", obb_last_error());
}

at::Tensor as_f32_dev(const at::Tensor &t, const char *name) {
  TORCH_CHECK(t.is_cuda(), name, " must be a HIP tensor");
  return t.to(at::kFloat).contiguous();
}

// keep indices (int64, ascending original order -- nms_cpu.cpp:59, nms_kernel.cu:135-138)
at::Tensor nms_impl(const at::Tensor &dets_in, int kind, double thr, const c10::optional<at::Tensor> &groups,
                    const c10::optional<at::Tensor> &thrs) {
  const int cols = kind == OBB_KIND_RBOX ? 6 : (kind == OBB_KIND_POLY ? 9 : 5);
  TORCH_CHECK(dets_in.dim() == 2 && dets_in.size(1) == cols, "dets must be (n, ", cols, ")");
  const int64_t n = dets_in.size(0);
  if (n == 0) return at::empty({0}, dets_in.options().dtype(at::kLong));   // nms_wrapper.py:50-51
  const at::Tensor dets = as_f32_dev(dets_in, "dets");
  HIPGuardMasqueradingAsCUDA guard(dets.device());
  at::Tensor g, t;
  const int32_t *gp = nullptr;
  const float *tp = nullptr;
  if (groups.has_value()) {
    g = groups->to(dets.device(), at::kInt).contiguous();
    TORCH_CHECK(g.numel() == n, "groups must have n entries");
    gp = g.data_ptr<int32_t>();
  }
  if (thrs.has_value()) {
    t = thrs->to(dets.device(), at::kFloat).contiguous();
    TORCH_CHECK(t.numel() == n, "thrs must have n entries");
    tp = t.data_ptr<float>();
  }
  at::Tensor keep = at::empty({n}, dets.options().dtype(at::kLong));
  int64_t info[4] = {0, 0, 0, 0};
  int64_t capacity = 0;   // library default; grown once if the scene is denser
  int rc = OBB_OK;
  for (int attempt = 0; attempt < 3; ++attempt) {
    const size_t bytes = obb_nms_workspace_bytes(n, kind, capacity);
    TORCH_CHECK(bytes > 0, "obb_nms_workspace_bytes: ", obb_last_error());
    at::Tensor ws = at::empty({(int64_t)bytes}, dets.options().dtype(at::kByte));
    rc = obb_nms_grouped_f32(dets.data_ptr<float>(), n, kind, (float)thr, gp, tp, keep.data_ptr<int64_t>(),
                             info, ws.data_ptr(), bytes, current_stream(dets));
    if (rc != OBB_EWORKSPACE) break;
    capacity = info[3];
  }
  check_rc(rc, "obb_nms_f32");
  return keep.narrow(0, 0, info[0]);
}

// The synchronous NMS keeps ONE workspace per (calling thread, device, stream) alive between calls instead of asking the
// caching allocator for one per call: a call returns when its result is on the host, so the next call on that stream may
// reuse the buffer -- and a buffer nobody else writes is what OBB_NMS_RETAIN asks for (obbops.h: the library keeps the
// next call's frame and cleared counters in it, three launches instead of four).  release_workspaces() drops them.
std::map<std::pair<int, void *>, at::Tensor> &nms_workspaces() {
  static thread_local std::map<std::pair<int, void *>, at::Tensor> cache;
  return cache;
}
at::Tensor &nms_workspace(const at::Tensor &like, void *stream, size_t bytes) {
  at::Tensor &ws = nms_workspaces()[std::make_pair((int)like.device().index(), stream)];
  if (!ws.defined() || (size_t)ws.numel() < bytes) {
    if (ws.defined()) obb_nms_retain_forget(ws.data_ptr());      // (its memory goes back to the allocator)
    ws = at::empty({(int64_t)bytes}, like.options().dtype(at::kByte));
  }
  return ws;
}
void release_nms_workspaces() {
  for (auto &kv : nms_workspaces()) if (kv.second.defined()) obb_nms_retain_forget(kv.second.data_ptr());
  nms_workspaces().clear();
}

// The wrapper's hot path (aidet_amd.nms_wrapper._run_nms): ONE attempt of obb_nms_ex_f32 on tensors that already are
// what the C ABI takes (fp32 contiguous dets, int32 groups, fp32 per-box thresholds), everything else -- allocation of
// the keep list and of the workspace, device guard, stream lookup -- on this side of the interpreter.  Returns
// (keep[:K], rc, info[4]); rc == OBB_EWORKSPACE leaves the retry with a larger `capacity` (info[3]) to the caller.
std::tuple<at::Tensor, int64_t, std::vector<int64_t>> nms_ex(const at::Tensor &dets, int64_t kind, double thr,
                                                             const c10::optional<at::Tensor> &groups,
                                                             const c10::optional<at::Tensor> &thrs, int64_t flags,
                                                             int64_t capacity) {
  TORCH_CHECK(dets.is_cuda() && dets.scalar_type() == at::kFloat && dets.is_contiguous() && dets.dim() == 2,
              "nms_ex takes a contiguous fp32 HIP tensor");
  const int64_t n = dets.size(0);
  HIPGuardMasqueradingAsCUDA guard(dets.device());
  const int32_t *gp = nullptr;
  const float *tp = nullptr;
  if (groups.has_value()) {
    TORCH_CHECK(groups->is_cuda() && groups->scalar_type() == at::kInt && groups->is_contiguous() && groups->numel() == n, "groups: int32[n]");
    gp = groups->data_ptr<int32_t>();
  }
  if (thrs.has_value()) {
    TORCH_CHECK(thrs->is_cuda() && thrs->scalar_type() == at::kFloat && thrs->is_contiguous() && thrs->numel() == n, "thrs: fp32[n]");
    tp = thrs->data_ptr<float>();
  }
  at::Tensor keep = at::empty({n}, dets.options().dtype(at::kLong));
  std::vector<int64_t> info(4, 0);
  const size_t bytes = obb_nms_workspace_bytes(n, (int)kind, capacity);
  TORCH_CHECK(bytes > 0, "obb_nms_workspace_bytes: ", obb_last_error());
  void *stream = current_stream(dets);
  // (the kept workspace where OBB_NMS_RETAIN applies -- below 2^18 boxes, <= ~400 MB; a larger call's workspace, gigabytes of
  //  table, goes back to the caching allocator with the call as before)
  const bool retain = n < (1 << 18);
  at::Tensor once;
  if (!retain) once = at::empty({(int64_t)bytes}, dets.options().dtype(at::kByte));
  at::Tensor &ws = retain ? nms_workspace(dets, stream, bytes) : once;
  const int rc = obb_nms_ex_f32(dets.data_ptr<float>(), n, (int)kind, (float)thr, gp, tp, (int)flags | (retain ? OBB_NMS_RETAIN : 0), keep.data_ptr<int64_t>(),
                                info.data(), ws.data_ptr(), (size_t)ws.numel(), stream);
  if (rc != OBB_OK && rc != OBB_EWORKSPACE) check_rc(rc, "obb_nms_ex_f32");
  return std::make_tuple(keep.narrow(0, 0, rc == OBB_OK ? info[0] : 0), (int64_t)rc, info);
}

at::Tensor rnms(const at::Tensor &dets, double thr) { return nms_impl(dets, OBB_KIND_RBOX, thr, c10::nullopt, c10::nullopt); }
at::Tensor poly_nms(const at::Tensor &dets, double thr) { return nms_impl(dets, OBB_KIND_POLY, thr, c10::nullopt, c10::nullopt); }
// horizontal boxes (n, 5) x1, y1, x2, y2, score: the signature of nms_cuda.nms (nms_cuda.cpp:8)
at::Tensor hbb_nms(const at::Tensor &dets, double thr) { return nms_impl(dets, OBB_KIND_HBB, thr, c10::nullopt, c10::nullopt); }
at::Tensor rnms_grouped(const at::Tensor &dets, double thr, const c10::optional<at::Tensor> &groups,
                        const c10::optional<at::Tensor> &thrs) {
  return nms_impl(dets, OBB_KIND_RBOX, thr, groups, thrs);
}
at::Tensor poly_nms_grouped(const at::Tensor &dets, double thr, const c10::optional<at::Tensor> &groups,
                            const c10::optional<at::Tensor> &thrs) {
  return nms_impl(dets, OBB_KIND_POLY, thr, groups, thrs);
}

// (m, n) overlaps, or (m,) when aligned; empty shapes as geometry.py:51-55
at::Tensor overlaps(const at::Tensor &a_in, const at::Tensor &b_in, const std::string &mode, bool aligned) {
  TORCH_CHECK(mode == "iou" || mode == "iof", "mode must be 'iou' or 'iof'");
  TORCH_CHECK(a_in.dim() == 2 && b_in.dim() == 2 && a_in.size(1) == b_in.size(1) &&
                  (a_in.size(1) == 5 || a_in.size(1) == 8),
              "boxes must be (m, 5) theta-OBBs or (m, 8) quads, both alike");
  const int kind = a_in.size(1) == 5 ? OBB_KIND_RBOX : OBB_KIND_POLY;
  const int64_t m = a_in.size(0), n = b_in.size(0);
  if (aligned) TORCH_CHECK(m == n, "aligned overlaps need as many boxes on both sides");
  const at::Tensor a = as_f32_dev(a_in, "bboxes1"), b = as_f32_dev(b_in, "bboxes2");
  HIPGuardMasqueradingAsCUDA guard(a.device());
  if (m * n == 0) return at::zeros(aligned ? std::vector<int64_t>{m, 1} : std::vector<int64_t>{m, n}, a.options());
  at::Tensor out = at::empty(aligned ? std::vector<int64_t>{m} : std::vector<int64_t>{m, n}, a.options());
  check_rc(obb_overlaps_f32(a.data_ptr<float>(), m, b.data_ptr<float>(), n, kind,
                            mode == "iou" ? OBB_MODE_IOU : OBB_MODE_IOF, aligned ? 1 : 0, out.data_ptr<float>(),
                            current_stream(a)),
           "obb_overlaps_f32");
  return out;
}

at::Tensor rroi_align_forward(const at::Tensor &features, const at::Tensor &rois, int64_t ph, int64_t pw,
                              double spatial_scale, int64_t sample_num, double min_size) {
  TORCH_CHECK(features.dim() == 4, "features must be (B, C, H, W)");
  TORCH_CHECK(rois.dim() == 2 && rois.size(1) == 6, "rois must be (K, 6) [batch, cx, cy, w, h, theta]");
  const at::Tensor f = as_f32_dev(features, "features"), r = as_f32_dev(rois, "rois");
  HIPGuardMasqueradingAsCUDA guard(f.device());
  const int64_t K = r.size(0), C = f.size(1);
  at::Tensor out = at::empty({K, C, ph, pw}, f.options());
  if (K == 0) return out;
  check_rc(obb_rroi_align_forward_f32(f.data_ptr<float>(), (int)f.size(0), (int)C, (int)f.size(2), (int)f.size(3),
                                      r.data_ptr<float>(), K, (int)ph, (int)pw, (float)spatial_scale,
                                      (int)sample_num, (float)min_size, out.data_ptr<float>(), current_stream(f)),
           "obb_rroi_align_forward_f32");
  return out;
}

at::Tensor rroi_align_backward(const at::Tensor &grad_out, const at::Tensor &rois, int64_t B, int64_t C, int64_t H,
                               int64_t W, double spatial_scale, int64_t sample_num, double min_size) {
  TORCH_CHECK(grad_out.dim() == 4 && grad_out.size(1) == C, "grad_output must be (K, C, ph, pw)");
  TORCH_CHECK(rois.dim() == 2 && rois.size(1) == 6 && rois.size(0) == grad_out.size(0), "rois must be (K, 6)");
  const at::Tensor g = as_f32_dev(grad_out, "grad_output"), r = as_f32_dev(rois, "rois");
  HIPGuardMasqueradingAsCUDA guard(g.device());
  at::Tensor grad_in = at::zeros({B, C, H, W}, g.options());
  if (g.size(0) == 0) return grad_in;
  check_rc(obb_rroi_align_backward_f32(g.data_ptr<float>(), (int)B, (int)C, (int)H, (int)W, r.data_ptr<float>(),
                                       g.size(0), (int)g.size(2), (int)g.size(3), (float)spatial_scale,
                                       (int)sample_num, (float)min_size, grad_in.data_ptr<float>(),
                                       current_stream(g)),
           "obb_rroi_align_backward_f32");
  return grad_in;
}

// horizontal RoIAlign (roi_align_cuda.forward / backward, roi_align.py:9-73): rois (K, 5)
// [batch, x1, y1, x2, y2]; aligned = false: legacy V1 geometry, true: V2 aligned
at::Tensor roi_align_forward(const at::Tensor &features, const at::Tensor &rois, int64_t ph, int64_t pw,
                             double spatial_scale, int64_t sample_num, bool aligned) {
  TORCH_CHECK(features.dim() == 4, "features must be (B, C, H, W)");
  TORCH_CHECK(rois.dim() == 2 && rois.size(1) == 5, "rois must be (K, 5) [batch, x1, y1, x2, y2]");
  const at::Tensor f = as_f32_dev(features, "features"), r = as_f32_dev(rois, "rois");
  HIPGuardMasqueradingAsCUDA guard(f.device());
  const int64_t K = r.size(0), C = f.size(1);
  at::Tensor out = at::empty({K, C, ph, pw}, f.options());
  if (K == 0) return out;
  check_rc(obb_roi_align_forward_f32(f.data_ptr<float>(), (int)f.size(0), (int)C, (int)f.size(2), (int)f.size(3),
                                     r.data_ptr<float>(), K, (int)ph, (int)pw, (float)spatial_scale,
                                     (int)sample_num, aligned ? 1 : 0, out.data_ptr<float>(), current_stream(f)),
           "obb_roi_align_forward_f32");
  return out;
}

at::Tensor roi_align_backward(const at::Tensor &grad_out, const at::Tensor &rois, int64_t B, int64_t C, int64_t H,
                              int64_t W, double spatial_scale, int64_t sample_num, bool aligned) {
  TORCH_CHECK(grad_out.dim() == 4 && grad_out.size(1) == C, "grad_output must be (K, C, ph, pw)");
  TORCH_CHECK(rois.dim() == 2 && rois.size(1) == 5 && rois.size(0) == grad_out.size(0), "rois must be (K, 5)");
  const at::Tensor g = as_f32_dev(grad_out, "grad_output"), r = as_f32_dev(rois, "rois");
  HIPGuardMasqueradingAsCUDA guard(g.device());
  at::Tensor grad_in = at::zeros({B, C, H, W}, g.options());
  if (g.size(0) == 0) return grad_in;
  check_rc(obb_roi_align_backward_f32(g.data_ptr<float>(), (int)B, (int)C, (int)H, (int)W, r.data_ptr<float>(),
                                      g.size(0), (int)g.size(2), (int)g.size(3), (float)spatial_scale,
                                      (int)sample_num, aligned ? 1 : 0, grad_in.data_ptr<float>(), current_stream(g)),
           "obb_roi_align_backward_f32");
  return grad_in;
}

at::Tensor rbox_to_poly(const at::Tensor &rboxes) {
  TORCH_CHECK(rboxes.dim() == 2 && rboxes.size(1) == 5, "rboxes must be (n, 5)");
  const at::Tensor r = as_f32_dev(rboxes, "rboxes");
  HIPGuardMasqueradingAsCUDA guard(r.device());
  at::Tensor out = at::empty({r.size(0), 8}, r.options());
  if (r.size(0) == 0) return out;
  check_rc(obb_rbox_to_poly_f32(r.data_ptr<float>(), r.size(0), out.data_ptr<float>(), current_stream(r)),
           "obb_rbox_to_poly_f32");
  return out;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "ATen binding of libobbops (rotated IoU / NMS / RoI-Align on gfx950)";
  m.def("version", []() { return obb_version(); });
  m.def("build_info", []() { return std::string(obb_build_info()); });
  m.def("nms_ex", &nms_ex, "one attempt of obb_nms_ex_f32: (keep, rc, info)", py::arg("dets"), py::arg("kind"), py::arg("threshold"),
        py::arg("groups") = py::none(), py::arg("thrs") = py::none(), py::arg("flags") = 0, py::arg("capacity") = 0);
  m.def("release_workspaces", &release_nms_workspaces, "drops the calling thread's cached NMS workspaces");
  m.def("rnms", &rnms, "rotated NMS: keep indices (int64, ascending)", py::arg("dets"), py::arg("threshold"));
  m.def("poly_nms", &poly_nms, "polygon NMS: keep indices", py::arg("dets"), py::arg("threshold"));
  m.def("nms", &hbb_nms, "horizontal-box NMS (the reference's +1 convention): keep indices", py::arg("dets"), py::arg("threshold"));
  m.def("rnms_grouped", &rnms_grouped, py::arg("dets"), py::arg("threshold"), py::arg("groups") = py::none(),
        py::arg("thrs") = py::none());
  m.def("poly_nms_grouped", &poly_nms_grouped, py::arg("dets"), py::arg("threshold"), py::arg("groups") = py::none(),
        py::arg("thrs") = py::none());
  m.def("rbbox_overlaps", &overlaps, py::arg("bboxes1"), py::arg("bboxes2"), py::arg("mode") = "iou",
        py::arg("is_aligned") = false);
  m.def("rroi_align_forward", &rroi_align_forward, py::arg("features"), py::arg("rois"), py::arg("pooled_height"),
        py::arg("pooled_width"), py::arg("spatial_scale"), py::arg("sample_num") = 0, py::arg("min_size") = 1.0);
  m.def("rroi_align_backward", &rroi_align_backward, py::arg("grad_output"), py::arg("rois"), py::arg("batch"),
        py::arg("channels"), py::arg("height"), py::arg("width"), py::arg("spatial_scale"),
        py::arg("sample_num") = 0, py::arg("min_size") = 1.0);
  m.def("roi_align_forward", &roi_align_forward, py::arg("features"), py::arg("rois"), py::arg("pooled_height"),
        py::arg("pooled_width"), py::arg("spatial_scale"), py::arg("sample_num") = 0, py::arg("aligned") = false);
  m.def("roi_align_backward", &roi_align_backward, py::arg("grad_output"), py::arg("rois"), py::arg("batch"),
        py::arg("channels"), py::arg("height"), py::arg("width"), py::arg("spatial_scale"),
        py::arg("sample_num") = 0, py::arg("aligned") = false);
  m.def("thetaobb2pointobb", &rbox_to_poly, py::arg("rboxes"));
}

"""Stand-alone timing of the detector's selection kernels on the real 1024^2 tile's RPN outputs (random-init model):
obb_rpn_select_decode_f32, the grouped NMS, obb_rows_topk_f32, and the fused / tensor-op rpn_proposals."""
import os, sys, ctypes, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aidet_amd import _lib, nms_wrapper
from aidet_amd.models import build_detector, demo_inputs
from aidet_amd.rpn import rpn_proposals
dev = torch.device('cuda', 0)
model = build_detector(dev)
img, meta = demo_inputs(dev)
with torch.no_grad():
    x = model.extract_feat(img)
    cs, bp = model.rpn_head(x)
cls = [c[0].contiguous() for c in cs]; reg = [r[0].contiguous() for r in bp]
anchors = model.rpn_head.anchors([tuple(c.shape[-2:]) for c in cls], dev)
L = _lib.lib(); n = len(cls)
per = [min(2000, c.numel()) for c in cls]; N = sum(per)
dets = torch.empty((N, 5), device=dev); lv = torch.empty(N, dtype=torch.int32, device=dev); rows = ctypes.c_int64(0)
st = _lib.current_stream_handle(dev)
args = ((ctypes.c_void_p * n)(*[c.data_ptr() for c in cls]), (ctypes.c_void_p * n)(*[r.data_ptr() for r in reg]),
        (ctypes.c_void_p * n)(*[a.data_ptr() for a in anchors]), (ctypes.c_int * n)(*[c.size(1) for c in cls]), (ctypes.c_int * n)(*[c.size(2) for c in cls]))
def sel():
    _lib.check(L.obb_rpn_select_decode_f32(*args, n, 3, 1, 2000, (ctypes.c_float * 4)(0, 0, 0, 0), (ctypes.c_float * 4)(1, 1, 1, 1), 16 / 1000.0, 1024, 1024,
                                           dets.data_ptr(), lv.data_ptr(), ctypes.byref(rows), st))
wsb = int(L.obb_rpn_select_workspace_bytes(args[3], args[4], n, 3, 2000))
ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
dets2 = torch.empty_like(dets); lv2 = torch.empty_like(lv)
def sel_ws():
    _lib.check(L.obb_rpn_select_decode_ws_f32(*args, n, 3, 1, 2000, (ctypes.c_float * 4)(0, 0, 0, 0), (ctypes.c_float * 4)(1, 1, 1, 1), 16 / 1000.0, 1024, 1024,
                                              dets2.data_ptr(), lv2.data_ptr(), ctypes.byref(rows), ws.data_ptr(), wsb, st))
sel(); sel_ws()
same = bool(torch.equal(dets, dets2) and torch.equal(lv, lv2))
keep, ninfo = nms_wrapper._run_nms(dets, _lib.KIND_HBB, 0.7, lv)
K = keep.numel(); base = [0]
for m in per: base.append(base[-1] + m)
out = torch.empty((2000, 5), device=dev); cnt = torch.empty(1, dtype=torch.int64, device=dev)
def topk():
    _lib.check(L.obb_rows_topk_f32(dets.data_ptr(), N, 5, 4, keep.data_ptr(), K, (ctypes.c_int * (n + 1))(*base), n, 2000, 2000, out.data_ptr(), None, cnt.data_ptr(), st))
res = {'select_decode_ms': round(bench._time_gpu(sel_ws, 20), 4), 'select_decode_one_workgroup_per_level_ms': round(bench._time_gpu(sel, 20), 4),
       'select_decode_paths_equal': same, 'nms_ms': round(bench._time_gpu(lambda: nms_wrapper._run_nms(dets, _lib.KIND_HBB, 0.7, lv), 20), 4),
       'rows_topk_ms': round(bench._time_gpu(topk, 20), 4), 'kept': int(K), 'nms_edges_rounds_candidates': [int(v) for v in ninfo],
       'rpn_fused_ms': round(bench._time_gpu(lambda: rpn_proposals(cls, reg, anchors, (1024, 1024, 3), model.test_cfg['rpn'], fused=True), 20), 4),
       'rpn_tensor_ops_ms': round(bench._time_gpu(lambda: rpn_proposals(cls, reg, anchors, (1024, 1024, 3), model.test_cfg['rpn'], fused=False), 20), 4)}
for name, dm in (('nms_sorted_route_ms', 0), ('nms_direct_route_ms', 65536)):
    _lib.set_tuning(_lib.TUNE_NMS_DIRECT_MAX, dm)
    res[name] = round(bench._time_gpu(lambda: nms_wrapper._run_nms(dets, _lib.KIND_HBB, 0.7, lv), 20), 4)
_lib.set_tuning(_lib.TUNE_NMS_DIRECT_MAX, -1)
L.obb_nms_profile_enable(2)
pbuf = (ctypes.c_float * 7)()
nms_wrapper._ext = None          # (the phase profile is read through the ctypes path's thread)
for _ in range(3):
    nms_wrapper._run_nms(dets, _lib.KIND_HBB, 0.7, lv)
L.obb_nms_profile_last(pbuf)
res['nms_phase_ms'] = [round(float(v), 4) for v in pbuf]
L.obb_nms_profile_enable(0)
print(json.dumps(res))

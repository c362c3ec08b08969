"""aidet_amd -- MI355X-native oriented-bounding-box operators for AIDet / MMDetection v1.x.

Hand-written HIP kernels (gfx950) behind a flat C ABI (include/obbops.h,
aidet_amd/lib/libobbops.so) and thin host wrappers that mirror the reference's
operator surface:

  nms_wrapper.rnms / poly_nms / thetaobb_nms / nms   <-> mmdet/ops/nms/nms_wrapper.py:7-60
  nms_wrapper.soft_nms                               <-> mmdet/ops/nms/nms_wrapper.py:63-118
  rbbox_geometry.rbbox_overlaps / bbox_overlaps      <-> mmdet/core/bbox/geometry.py:4-88
  rroi_align.RRoIAlign / rroi_align                  <-> mmdet/ops/roi_align/roi_align.py:9-152
  roi_layers.RoIAlign / SingleRoIExtractor           <-> roi_align.py:79-152, roi_extractors/single_level.py:11-107
  post_processing.*                                  <-> core/post_processing/bbox_nms.py, rbbox_nms.py
  merge_augs.merge_aug_{proposals,bboxes,scores}     <-> core/post_processing/merge_augs.py:8-78 (test-time augmentation)
  rbbox_transforms.*, rbbox_target.rbbox_target      <-> core/rbbox/transforms.py, core/rbbox/rbbox_target.py
  rpn.rpn_proposals, anchors, bbox_transforms        <-> anchor_heads/rpn_head.py:55-104, core/anchor, core/bbox/transforms.py
  dota_merge / dota_io                               <-> apis/test.py:116-146, datasets/dota.py:278-336
  models.ThetaOBBRCNN                                <-> detectors/rbbox_cnn.py:214-253 (test path, config 4)

There is no CPU fallback: every op raises if the HIP library or a GPU is missing.
"""
from .merge_augs import merge_aug_bboxes, merge_aug_proposals, merge_aug_scores
from .nms_wrapper import nms, poly_nms, rnms, soft_nms, thetaobb_nms
from .post_processing import (get_det_rbboxes, multiclass_nms, multiclass_nms_with_index, multiclass_rnms,
                              multiclass_thetaobb_nms, thetaobb_nms_by_bbox_nms)
from .rbbox_geometry import bbox_overlaps, rbbox_overlaps, thetaobb2pointobb
from .roi_layers import RoIAlign, RotatedSingleRoIExtractor, SingleRoIExtractor, roi_align
from .rroi_align import RRoIAlign, rroi_align

__all__ = [
    'rnms', 'poly_nms', 'thetaobb_nms', 'nms', 'soft_nms', 'rbbox_overlaps', 'bbox_overlaps', 'thetaobb2pointobb',
    'RRoIAlign', 'rroi_align', 'RoIAlign', 'roi_align', 'SingleRoIExtractor', 'RotatedSingleRoIExtractor',
    'multiclass_nms', 'multiclass_rnms', 'multiclass_thetaobb_nms', 'multiclass_nms_with_index', 'thetaobb_nms_by_bbox_nms', 'get_det_rbboxes',
    'merge_aug_proposals', 'merge_aug_bboxes', 'merge_aug_scores'
]
__version__ = '1.4.0'   # = libobbops 1.4 (obb_version() == 104: round 6 -- OBB_NMS_RETAIN, the three-launch NMS on a retained workspace)


def get_compiler_version():
    """Compiler that built libobbops.so (counterpart of mmdet/ops/utils: ``get_compiler_version`` /
    ``get_compiling_cuda_version``, used by the reference's collect_env)."""
    from . import _lib
    return _lib.lib().obb_build_info().decode()


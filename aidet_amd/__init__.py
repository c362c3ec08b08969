"""aidet_amd -- MI355X-native oriented-bounding-box operators for AIDet / MMDetection v1.x.

Hand-written HIP kernels (gfx950) behind a flat C ABI (include/obbops.h,
aidet_amd/lib/libobbops.so) and thin host wrappers that mirror the reference's
operator surface:

  nms_wrapper.rnms / poly_nms / thetaobb_nms / nms   <-> mmdet/ops/nms/nms_wrapper.py:7-60
  rbbox_geometry.rbbox_overlaps                      <-> mmdet/core/bbox/geometry.py:4-88
  rroi_align.RRoIAlign / rroi_align                  <-> mmdet/ops/roi_align/roi_align.py:9-152

There is no CPU fallback: every op raises if the HIP library or a GPU is missing.
"""
from .nms_wrapper import nms, poly_nms, rnms, thetaobb_nms
from .post_processing import (multiclass_nms_with_index, multiclass_rnms,
                              thetaobb_nms_by_bbox_nms)
from .rbbox_geometry import rbbox_overlaps, thetaobb2pointobb
from .roi_layers import RoIAlign, RotatedSingleRoIExtractor
from .rroi_align import RRoIAlign, rroi_align

__all__ = [
    'rnms', 'poly_nms', 'thetaobb_nms', 'nms', 'rbbox_overlaps', 'thetaobb2pointobb',
    'RRoIAlign', 'rroi_align', 'RoIAlign', 'RotatedSingleRoIExtractor', 'multiclass_rnms', 'multiclass_nms_with_index',
    'thetaobb_nms_by_bbox_nms'
]
__version__ = '0.1.0'

"""Faster-RCNN-OBB R50-FPN, test path only (BASELINE.json configs[3]).

The detector of ``configs/dota/dota_v002_theta_obb_r50_v1_train.py:11-76`` with
``test_cfg`` :117-132, as plain PyTorch modules around this package's operators -- the
workload in which the OBB ops are actually called: 5 levels x 2000-box horizontal NMS in
the RPN, two RoIAlign passes over 2000 RoIs x 4 levels, class-wise NMS of the head, the
rotated head's decode + (rotated) NMS.

Structure follows SURVEY.md Appendix A:
  backbone  ResNet-50, style 'pytorch' (stride on the 3x3 conv), C2..C5, BN in eval mode
            (mmdet/models/backbones/resnet.py:101-125)
  neck      FPN, 256 ch, 5 outputs, nearest up-sampling, extra level by max-pool stride 2
            (mmdet/models/necks/fpn.py:138-178)
  rpn       3x3 conv + ReLU, 1x1 cls (3 anchors, sigmoid) and reg (rpn_head.py:18-35);
            anchors / decode / per-level NMS: aidet_amd.anchors, aidet_amd.rpn
  heads     SharedFCBBoxHead (2 FC 1024 -> 16 cls, 64 reg) and RBBoxHead (-> 16 cls, 80 reg),
            each behind its own SingleRoIExtractor(RoIAlign 7x7, sample_num 2, V1)
            (convfc_bbox_head.py:175-188, rbbox_head.py:27-105,157-195)
  test      RBBoxRCNN.simple_test (rbbox_cnn.py:214-253) with RBBoxTestMixin
            (test_mixins.py:269-322,352-380)

Weights are random (``torch.manual_seed(0)``): there is no network for checkpoints, and
what config 4 measures is the operators, not the accuracy.  The conv trunk is stock
PyTorch on both sides; everything between the convolutions and the results is pinned op by
op against values captured from the reference (tests/golden/detector_parts.npz).
"""
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..anchors import AnchorGenerator
from ..bbox_transforms import bbox2result, bbox2roi, bbox_mapping, delta2bbox
from ..merge_augs import merge_aug_bboxes, merge_aug_proposals
from ..post_processing import (detect_horizontal_fused, detect_rotated_fused, get_det_rbboxes, head_fusable,
                               multiclass_nms, multiclass_nms_with_index)
from ..rbbox_transforms import rbbox2result
from ..roi_layers import SingleRoIExtractor
from ..rpn import rpn_proposals

# test_cfg of configs/dota/dota_v002_theta_obb_r50_v1_train.py:117-132
TEST_CFG = dict(
    rpn=dict(nms_across_levels=False, nms_pre=2000, nms_post=2000, max_num=2000, nms_thr=0.7,
             min_bbox_size=0),
    rcnn=dict(score_thr=0.05, nms=dict(type='nms', iou_thr=0.5), max_per_img=1000),
    rbbox=dict(encode='thetaobb', score_thr=0.05, polygon_nms_iou_thr=0.5, max_per_img=1000,
               parallel=True))


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(Bottleneck, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)   # 'pytorch' style
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.bn1(self.conv1(x)), inplace=True)
        out = F.relu(self.bn2(self.conv2(out)), inplace=True)
        out = self.bn3(self.conv3(out))
        return F.relu(out + idt, inplace=True)


class ResNet50(nn.Module):
    """C2..C5 (256 / 512 / 1024 / 2048 channels at strides 4 / 8 / 16 / 32)."""

    def __init__(self):
        super(ResNet50, self).__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.inplanes = 64
        self.layers = nn.ModuleList([self._stage(64, 3, 1), self._stage(128, 4, 2),
                                     self._stage(256, 6, 2), self._stage(512, 3, 2)])

    def _stage(self, planes, blocks, stride):
        down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                             nn.BatchNorm2d(planes * 4))
        mods = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        mods += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x)), inplace=True), 3, stride=2, padding=1)
        outs = []
        for layer in self.layers:
            x = layer(x)
            outs.append(x)
        return outs


class FPN(nn.Module):

    def __init__(self, in_channels=(256, 512, 1024, 2048), out_channels=256, num_outs=5):
        super(FPN, self).__init__()
        self.lateral_convs = nn.ModuleList([nn.Conv2d(c, out_channels, 1) for c in in_channels])
        self.fpn_convs = nn.ModuleList([nn.Conv2d(out_channels, out_channels, 3, padding=1) for _ in in_channels])
        self.num_outs = num_outs

    def forward(self, inputs):
        lat = [conv(x) for conv, x in zip(self.lateral_convs, inputs)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest')
        outs = [conv(x) for conv, x in zip(self.fpn_convs, lat)]
        while len(outs) < self.num_outs:
            outs.append(F.max_pool2d(outs[-1], 1, stride=2))
        return outs


class RPNHead(nn.Module):

    def __init__(self, in_channels=256, feat_channels=256, anchor_scales=(8, ), anchor_ratios=(0.5, 1.0, 2.0),
                 anchor_strides=(4, 8, 16, 32, 64)):
        super(RPNHead, self).__init__()
        self.anchor_strides = list(anchor_strides)
        self.anchor_generators = [AnchorGenerator(s, list(anchor_scales), list(anchor_ratios)) for s in anchor_strides]
        num_anchors = len(anchor_scales) * len(anchor_ratios)
        self.rpn_conv = nn.Conv2d(in_channels, feat_channels, 3, padding=1)
        self.rpn_cls = nn.Conv2d(feat_channels, num_anchors, 1)          # use_sigmoid: 1 channel per anchor
        self.rpn_reg = nn.Conv2d(feat_channels, num_anchors * 4, 1)
        self._anchor_cache = {}

    def forward(self, feats):
        xs = [F.relu(self.rpn_conv(x), inplace=True) for x in feats]
        return [self.rpn_cls(x) for x in xs], [self.rpn_reg(x) for x in xs]

    def anchors(self, featmap_sizes, device):
        key = (tuple(featmap_sizes), str(device))
        if key not in self._anchor_cache:
            self._anchor_cache[key] = [g.grid_anchors(sz, st, device=device) for g, sz, st in
                                       zip(self.anchor_generators, featmap_sizes, self.anchor_strides)]
        return self._anchor_cache[key]

    def get_bboxes(self, cls_scores, bbox_preds, img_metas, cfg, fused=None):
        """anchor_head.py:254-283 for the RPN: one (n, 5) proposal tensor per image."""
        mlvl_anchors = self.anchors([tuple(c.shape[-2:]) for c in cls_scores], cls_scores[0].device)
        return [rpn_proposals([c[i].detach() for c in cls_scores], [b[i].detach() for b in bbox_preds],
                              mlvl_anchors, meta['img_shape'], cfg, fused=fused) for i, meta in enumerate(img_metas)]


class SharedFCHead(nn.Module):
    """flatten -> FC 1024 -> ReLU -> FC 1024 -> ReLU -> fc_cls, fc_reg
    (``SharedFCBBoxHead`` with reg 4, ``RBBoxHead(num_shared_fcs=2)`` with reg 5)."""

    def __init__(self, out_dim_reg, num_classes=16, in_channels=256, roi_feat_size=7, fc_out_channels=1024):
        super(SharedFCHead, self).__init__()
        self.num_classes = num_classes
        self.shared_fcs = nn.ModuleList([nn.Linear(in_channels * roi_feat_size * roi_feat_size, fc_out_channels),
                                         nn.Linear(fc_out_channels, fc_out_channels)])
        self.fc_cls = nn.Linear(fc_out_channels, num_classes)
        self.fc_reg = nn.Linear(fc_out_channels, out_dim_reg * num_classes)

    def forward(self, x):
        x = x.flatten(1)
        for fc in self.shared_fcs:
            x = F.relu(fc(x))
        return self.fc_cls(x), self.fc_reg(x)


def _same_extractor(a, b):
    """two SingleRoIExtractors that pool identically (same layer type and settings on the same strides)"""
    def sig(e):
        return (type(e), tuple(e.featmap_strides), e.finest_scale, e.out_channels,
                tuple((type(l), tuple(np.atleast_1d(l.out_size)), float(l.spatial_scale), int(l.sample_num),
                       bool(getattr(l, 'aligned', False)), bool(getattr(l, 'use_torchvision', False))) for l in e.roi_layers))
    return sig(a) == sig(b)


class ThetaOBBRCNN(nn.Module):
    """``RBBoxRCNN`` (rbbox_cnn.py) of the theta-OBB config, inference only.

    ``rotated_nms=False`` reproduces the reference: the rotated boxes that survive are those
    whose horizontal boxes survived (rbbox_head.py:287-293).  ``rotated_nms=True`` runs the
    call the reference left commented out (:294-295): a real rotated NMS on the decoded
    theta-OBBs (``multiclass_rnms``).
    """

    def __init__(self, test_cfg=None, rotated_nms=False, fused=True):
        super(ThetaOBBRCNN, self).__init__()
        # fused=True: the selection stages of the RPN and of both heads run as the HIP kernels of obb_detect.hip
        # (three to four launches per stage); False: the tensor-op restatement of the reference's functions, op by op
        # (aidet_amd.rpn / post_processing) -- the specification the fused path is tested against
        self.fused = fused
        self.backbone = ResNet50()
        self.neck = FPN()
        self.rpn_head = RPNHead()
        roi = dict(type='RoIAlign', out_size=7, sample_num=2)
        self.bbox_roi_extractor = SingleRoIExtractor(roi, 256, [4, 8, 16, 32])
        self.bbox_head = SharedFCHead(4)
        self.rbbox_roi_extractor = SingleRoIExtractor(roi, 256, [4, 8, 16, 32])
        # Both branches pool the SAME RoIs from the same maps (test_mixins.py:281 and :352-380 call their extractors one
        # after the other); an extractor has no weights, so when the two are configured alike the second call would
        # recompute the first one's output bit for bit: at inference it is pooled once (VERDICT r3 next #3 iv: -0.35 ms
        # per tile).  `pool_once=False` keeps the reference's two calls.
        self.pool_once = True
        self.rbbox_head = SharedFCHead(5)
        self.test_cfg = test_cfg or TEST_CFG
        self.rotated_nms = rotated_nms
        self.bbox_target_stds = (0.1, 0.1, 0.2, 0.2)
        self.rbbox_target_stds = (0.1, 0.1, 0.2, 0.2, 0.1)
        self.timers = None
        self.init_weights()

    def init_weights(self):
        """the reference's initialisers (kaiming for the backbone convs, xavier for the FPN and
        the FC trunks, normal(0.01) for the RPN and the cls layers, 0.001 for the regressors)."""
        for m in self.backbone.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        for m in self.neck.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)
        for m in (self.rpn_head.rpn_conv, self.rpn_head.rpn_cls, self.rpn_head.rpn_reg):
            nn.init.normal_(m.weight, std=0.01)
            nn.init.zeros_(m.bias)
        for head in (self.bbox_head, self.rbbox_head):
            for fc in head.shared_fcs:
                nn.init.xavier_uniform_(fc.weight)
                nn.init.zeros_(fc.bias)
            nn.init.normal_(head.fc_cls.weight, 0, 0.01)
            nn.init.normal_(head.fc_reg.weight, 0, 0.001)
            nn.init.zeros_(head.fc_cls.bias)
            nn.init.zeros_(head.fc_reg.bias)

    # ---- timing hooks (bench): wall time per stage, device synchronised at the boundaries ----
    def _tick(self, name):
        if self.timers is not None:
            torch.cuda.synchronize()
            now = time.perf_counter()
            self.timers[name] = self.timers.get(name, 0.0) + (now - self._t0)
            self._t0 = now

    def extract_feat(self, img):
        return self.neck(self.backbone(img))

    @torch.no_grad()
    def simple_test(self, img, img_meta, rescale=False):
        """One image tile -> (bbox_results, rbbox_results): per class numpy arrays (n, 5) and
        (n, 6), rbbox_cnn.py:214-253."""
        if self.timers is not None:
            torch.cuda.synchronize()
            self._t0 = time.perf_counter()
        x = self.extract_feat(img)
        self._tick('backbone_fpn')
        cls_scores, bbox_preds = self.rpn_head(x)
        self._tick('rpn_convs')
        proposals = self.rpn_head.get_bboxes(cls_scores, bbox_preds, img_meta, self.test_cfg['rpn'], fused=self.fused)
        self._tick('rpn_decode_nms')
        img_shape, scale_factor = img_meta[0]['img_shape'], img_meta[0]['scale_factor']
        rois = bbox2roi(proposals)
        # ---- horizontal branch (test_mixins.py:269-322)
        roi_feats = self.bbox_roi_extractor(x[:4], rois)
        self._tick('roi_align_bbox')
        cls_score, bbox_pred = self.bbox_head(roi_feats)
        self._tick('bbox_head_fc')
        rcnn = self.test_cfg['rcnn']
        fused = (self.fused and rcnn['nms'].get('type', 'nms') == 'nms' and self.test_cfg['rbbox'].get('encode', 'thetaobb') == 'thetaobb'
                 and head_fusable(rois, cls_score, bbox_pred, rcnn['max_per_img'], scale_factor))
        selection = cls_inds = keep_inds = None
        if fused:
            det_bboxes, det_labels, selection = detect_horizontal_fused(
                rois, cls_score, bbox_pred, img_shape, scale_factor, rescale, rcnn, (0., 0., 0., 0.), self.bbox_target_stds)
        else:
            scores = F.softmax(cls_score, dim=1)
            bboxes = delta2bbox(rois[:, 1:], bbox_pred, (0., 0., 0., 0.), self.bbox_target_stds, img_shape)
            if rescale:     # (a scalar, or the 4-element array of keep_ratio=False: bbox_head.py:160-166)
                bboxes = bboxes / (scale_factor if isinstance(scale_factor, (int, float)) else
                                   torch.as_tensor(np.asarray(scale_factor), dtype=bboxes.dtype, device=bboxes.device))
            det_bboxes, det_labels, cls_inds, keep_inds = multiclass_nms_with_index(
                bboxes, scores, rcnn['score_thr'], rcnn['nms'], rcnn['max_per_img'])
        self._tick('bbox_decode_nms')
        bbox_results = bbox2result(det_bboxes, det_labels, self.bbox_head.num_classes)
        # ---- rotated branch (test_mixins.py:352-380): its own extractor and trunk
        if not (self.pool_once and not torch.is_grad_enabled() and _same_extractor(self.bbox_roi_extractor, self.rbbox_roi_extractor)):
            roi_feats = self.rbbox_roi_extractor(x[:4], rois)
        self._tick('roi_align_rbbox')
        cls_score, rbbox_pred = self.rbbox_head(roi_feats)
        self._tick('rbbox_head_fc')
        if fused:
            det_rbboxes, det_rlabels = detect_rotated_fused(
                rois, cls_score, rbbox_pred, img_shape, scale_factor, rescale, self.test_cfg['rbbox'], (0., 0., 0., 0., 0.),
                self.rbbox_target_stds, selection=None if self.rotated_nms else selection)
        else:
            det_rbboxes, det_rlabels = get_det_rbboxes(
                rois, cls_score, rbbox_pred, img_shape, scale_factor, rescale=rescale, cfg=self.test_cfg['rbbox'],
                target_stds=self.rbbox_target_stds, bbox_cls_inds=cls_inds, bbox_keep_inds=keep_inds,
                rotated_nms=self.rotated_nms)
        self._tick('rbbox_decode_nms')
        rbbox_results = rbbox2result(det_rbboxes, det_rlabels, self.rbbox_head.num_classes)
        self._tick('results_to_host')
        return bbox_results, rbbox_results

    @torch.no_grad()
    def aug_test(self, imgs, img_metas, rescale=False, with_rotated=False):
        """Test with augmentations (``RBBoxRCNN.aug_test``, rbbox_cnn.py:255-282): ``imgs`` / ``img_metas`` hold one
        entry per augmentation (scale, flip) of ONE image, ``img_metas[i][0]`` its meta.

        RPN leg = ``aug_test_rpn`` (test_mixins.py:37-59): the proposals of every view are mapped back and merged by one
        NMS (``merge_aug_proposals``).  Box leg = ``aug_test_bboxes`` (:122-157): the merged proposals are mapped INTO
        every view, pooled and decoded there with ``cfg=None`` (all classes, no NMS), mapped back, averaged
        (``merge_aug_bboxes``) and only then class-wise NMS'd.  Returns ``bbox_results`` at the scale of ``imgs[0]``
        (``rescale=False``) or of the original image -- what the reference returns: its rotated leg reads
        ``self.with_thetaobb`` / ``merge_aug_thetaobbs`` (rbbox_cnn.py:276-279, test_mixins.py:400), neither of which
        exists.  ``with_rotated=True`` (an extension, no reference counterpart) also returns ``rbbox_results``: the
        rotated head on the merged proposals in the FIRST view, decoded + rotated NMS like ``simple_test``."""
        cfg = self.test_cfg
        feats = [self.extract_feat(img) for img in imgs]           # (the reference recomputes them per leg to save memory)
        # ---- aug_test_rpn
        aug_proposals = []
        for x, meta in zip(feats, img_metas):
            cls_scores, bbox_preds = self.rpn_head(x)
            aug_proposals.append(self.rpn_head.get_bboxes(cls_scores, bbox_preds, meta, cfg['rpn'], fused=self.fused)[0])
        merged = merge_aug_proposals(aug_proposals, [m[0] for m in img_metas], cfg['rpn'])
        # ---- aug_test_bboxes
        aug_bboxes, aug_scores = [], []
        nlv = len(self.bbox_roi_extractor.featmap_strides)
        for x, meta in zip(feats, img_metas):
            m0 = meta[0]
            proposals = bbox_mapping(merged[:, :4], m0['img_shape'], m0['scale_factor'], m0['flip'])
            rois = bbox2roi([proposals])
            cls_score, bbox_pred = self.bbox_head(self.bbox_roi_extractor(x[:nlv], rois))
            scores = F.softmax(cls_score, dim=1)
            bboxes = delta2bbox(rois[:, 1:], bbox_pred, (0., 0., 0., 0.), self.bbox_target_stds, m0['img_shape'])
            aug_bboxes.append(bboxes)
            aug_scores.append(scores)
        merged_bboxes, merged_scores = merge_aug_bboxes(aug_bboxes, aug_scores, img_metas, cfg['rcnn'])
        rcnn = cfg['rcnn']
        det_bboxes, det_labels = multiclass_nms(merged_bboxes, merged_scores, rcnn['score_thr'], rcnn['nms'], rcnn['max_per_img'])
        if not rescale:                                              # (results fit the scale of imgs[0])
            det_bboxes = det_bboxes.clone()
            det_bboxes[:, :4] *= img_metas[0][0]['scale_factor']
        bbox_results = bbox2result(det_bboxes, det_labels, self.bbox_head.num_classes)
        if not with_rotated:
            return bbox_results
        m0 = img_metas[0][0]
        rois = bbox2roi([bbox_mapping(merged[:, :4], m0['img_shape'], m0['scale_factor'], m0['flip'])])
        cls_score, rbbox_pred = self.rbbox_head(self.rbbox_roi_extractor(feats[0][:nlv], rois))
        det_rbboxes, det_rlabels = get_det_rbboxes(rois, cls_score, rbbox_pred, m0['img_shape'], m0['scale_factor'], rescale=rescale,
                                                   cfg=cfg['rbbox'], target_stds=self.rbbox_target_stds, rotated_nms=True)
        return bbox_results, rbbox2result(det_rbboxes, det_rlabels, self.rbbox_head.num_classes)


def build_detector(device, rotated_nms=False, seed=0, fused=True):
    torch.manual_seed(seed)
    return ThetaOBBRCNN(rotated_nms=rotated_nms, fused=fused).to(device).eval()


def demo_inputs(device, size=1024, seed=0):
    """random image in [0, 1) like the reference's ``_demo_mm_inputs``
    (tests/test_forward.py:329-352) and the img_meta the test path reads."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand((1, 3, size, size), generator=g).to(device)
    meta = [dict(img_shape=(size, size, 3), ori_shape=(size, size, 3), pad_shape=(size, size, 3),
                 scale_factor=1.0, flip=False)]
    return img, meta


def bench_cfg4(device, iters=5):
    """Per-stage wall time of one 1024 x 1024 tile (config 4): both post-processing modes on the NCHW trunk, and the
    rotated-NMS mode on a CHANNELS-LAST trunk (``model.to(memory_format=torch.channels_last)``: MIOpen's NHWC
    convolutions hand the RoI extractor channels-last pyramid levels, pooled in place since round 5)."""
    out = {}
    for tag, rotated, channels_last in (('reference_keep_reuse', False, False), ('rotated_nms', True, False),
                                        ('rotated_nms_channels_last', True, True)):
        model = build_detector(device, rotated_nms=rotated)
        img, meta = demo_inputs(device)
        if channels_last:
            model = model.to(memory_format=torch.channels_last)
            img = img.contiguous(memory_format=torch.channels_last)
        for _ in range(2):
            res = model.simple_test(img, meta)
        model.timers = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            res = model.simple_test(img, meta)
        torch.cuda.synchronize()
        total = (time.perf_counter() - t0) / iters
        out[tag] = {'tile_ms': round(total * 1e3, 3),
                    'stages_ms': {k: round(v / iters * 1e3, 3) for k, v in model.timers.items()},
                    'detections_hbb': int(sum(len(r) for r in res[0])), 'detections_obb': int(sum(len(r) for r in res[1]))}
        if channels_last:
            from .. import roi_layers
            with torch.no_grad():
                x = model.extract_feat(img)
            out[tag]['pyramid_pooled_in_place'] = bool(all(roi_layers.is_nhwc(f) for f in x[:4]))
        del model
    out['model'] = 'Faster-RCNN-OBB R50-FPN (theta-OBB head), random init (seed 0), fp32, one 1024x1024 tile'
    return out

"""DOTA result plumbing around the merged NMS (host side).

The reference writes per-patch detections to ``Task1_<class>.txt`` / ``Task2_<class>.txt``
(``'%s %.3f %.1f ... %.1f'`` = patch name, score, 8 polygon or 4 box coordinates;
mmdet/datasets/dota.py:278-308) and hands them to the external, un-vendored
``wwtool.mergebypoly_mp`` / ``mergebyrec_mp`` (dota.py:310-336), which -- like
DOTA_devkit's ResultMerge they wrap [external, not in the reference tree] -- recover each
patch's offset from its file name ``<image>__<rate>__<x>___<y>``, move the boxes to image
coordinates, run a per-class NMS per image and write one line per surviving detection with
the ORIGINAL image name.

Here the detections never leave the GPU as text: ``parse_patch_name`` /
``patch_rows_to_image`` / ``patch_polys_to_image`` give the rows that
``dota_merge.distributed_merge`` consumes, ``write_task1`` / ``write_task2`` emit the final
files in the reference's formats.  What the text round trip does to the numbers can be
reproduced on request: ``quantize_text`` is ``float('%.1f' % x)`` evaluated exactly on the
device, and ``storage_tank_as_hbb`` is the substitution of dota.py:302-303.
"""
import os
import re

import numpy as np
import torch

# mmdet/datasets/dota.py:29-32 (DOTADataset.CLASSES)
DOTA_CLASSES = ('harbor', 'ship', 'small-vehicle', 'large-vehicle', 'storage-tank', 'plane',
                'soccer-ball-field', 'bridge', 'baseball-diamond', 'tennis-court', 'helicopter',
                'roundabout', 'swimming-pool', 'ground-track-field', 'basketball-court')

# per-class polygon-NMS thresholds of the merge (dota.py:324); flat 0.3 otherwise (:329-331)
OBB_NMS_THR = {'harbor': 0.1, 'ship': 0.05, 'small-vehicle': 0.15, 'large-vehicle': 0.5,
               'storage-tank': 0.35, 'plane': 0.2, 'soccer-ball-field': 0.2, 'bridge': 0.45,
               'baseball-diamond': 0.2, 'tennis-court': 0.1, 'helicopter': 0.1, 'roundabout': 0.15,
               'swimming-pool': 0.05, 'ground-track-field': 0.4, 'basketball-court': 0.2}
# per-class horizontal-NMS thresholds of the Task2 merge (dota.py:321)
HBB_NMS_THR = {'harbor': 0.4, 'ship': 0.4, 'small-vehicle': 0.4, 'large-vehicle': 0.5,
               'storage-tank': 0.1, 'plane': 0.25, 'soccer-ball-field': 0.2, 'bridge': 0.5,
               'baseball-diamond': 0.15, 'tennis-court': 0.2, 'helicopter': 0.2, 'roundabout': 0.15,
               'swimming-pool': 0.2, 'ground-track-field': 0.15, 'basketball-court': 0.2}

_PATCH_RE = re.compile(r'^(?P<name>.+)__(?P<rate>[0-9.]+)__(?P<x>-?\d+)___(?P<y>-?\d+)$')


def class_thresholds(classes=DOTA_CLASSES, classwise=True, flat=0.3, task='obb'):
    """threshold per class index, as merge_txt selects them (dota.py:320-334):
    task 'obb' -> the polygon merge (Task1), 'hbb' -> the horizontal merge (Task2)."""
    table = OBB_NMS_THR if task == 'obb' else HBB_NMS_THR
    return [table[c] if classwise else flat for c in classes]


def parse_patch_name(patch_name):
    """'P0003__1__824___0(.png)' -> ('P0003', 1.0, 824, 0)."""
    stem = os.path.basename(patch_name)
    low = stem.lower()
    for ext in ('.png', '.jpg', '.jpeg', '.tif', '.tiff', '.bmp'):  # the rate may contain a '.'
        if low.endswith(ext):
            stem = stem[:-len(ext)]
            break
    m = _PATCH_RE.match(stem)
    if m is None:
        raise ValueError('not a DOTA patch name: %r' % patch_name)
    return m.group('name'), float(m.group('rate')), int(m.group('x')), int(m.group('y'))


def quantize_text(x, decimals=1):
    """``float('%.<decimals>f' % v)`` for every element of an fp32 tensor, on its device.

    The reference's detections reach the merge as text (``%.1f`` coordinates, ``%.3f`` scores,
    dota.py:300,304).  For an fp32 value v, ``v * 10**decimals`` is exact in fp64 (24 + 10 bits
    at most), so rounding it half-to-even IS the correctly rounded decimal that printf emits;
    the parsed-back double is rounded to fp32, the precision the kernels compute in."""
    scale = float(10 ** decimals)
    return (torch.round(x.double() * scale) / scale).to(x.dtype)


def patch_rows_to_image(rbboxes, labels, patch_name, image_ids):
    """Per-patch theta-OBB detections -> rows for the merge, in ORIGINAL image coordinates.

    Args:
        rbboxes (Tensor): (n, 6) cx, cy, w, h, theta, score in patch pixels.
        labels (Tensor): (n,) 0-based class indices.
        patch_name (str): '<image>__<rate>__<x>___<y>'.
        image_ids (dict): image name -> integer id (extended in place).
    Returns:
        Tensor (n, 8): cx, cy, w, h, theta, score, label, image_id with
        (cx + x)/rate, (cy + y)/rate, w/rate, h/rate (the devkit's poly2origpoly on the
        box parameters).
    """
    name, rate, x, y = parse_patch_name(patch_name)
    img = image_ids.setdefault(name, len(image_ids))
    out = rbboxes.new_zeros((rbboxes.size(0), 8))
    out[:, 0] = (rbboxes[:, 0] + x) / rate
    out[:, 1] = (rbboxes[:, 1] + y) / rate
    out[:, 2] = rbboxes[:, 2] / rate
    out[:, 3] = rbboxes[:, 3] / rate
    out[:, 4] = rbboxes[:, 4]
    out[:, 5] = rbboxes[:, 5]
    out[:, 6] = labels.to(out.dtype)
    out[:, 7] = float(img)
    return out


def bbox2pointobb(bboxes):
    """(n, 4) x1,y1,x2,y2 -> (n, 8) corners x1,y1, x2,y1, x2,y2, x1,y2 (the conversion
    dota.py:303 applies to storage tanks through wwtool.bbox2pointobb [external])."""
    x1, y1, x2, y2 = bboxes.unbind(1)
    return torch.stack([x1, y1, x2, y1, x2, y2, x1, y2], 1)


def patch_polys_to_image(polys, scores, labels, patch_name, image_ids, quantize=False,
                         bboxes=None, storage_tank_as_hbb=False, classes=DOTA_CLASSES):
    """Per-patch POLYGON detections -> (n, 11) rows x1..y4, score, label, image_id in image
    coordinates: what the reference's Task1 files hold once the merge has read them.

    Args:
        polys (Tensor): (n, 8) corners in patch pixels (``thetaobb2pointobb``).
        quantize (bool): reproduce the text round trip -- ``%.1f`` on the patch coordinates,
            ``%.3f`` on the score (dota.py:304) -- before the patch offset is applied.
        bboxes (Tensor, optional): (n, 4) the detections' horizontal boxes.
        storage_tank_as_hbb (bool): write the horizontal box as the polygon of every
            'storage-tank' detection, as dota.py:302-303 does; needs ``bboxes``.
    """
    name, rate, x, y = parse_patch_name(patch_name)
    img = image_ids.setdefault(name, len(image_ids))
    if storage_tank_as_hbb:
        if bboxes is None:
            raise ValueError('storage_tank_as_hbb needs the detections\' horizontal boxes (bboxes=)')
        tank = labels == classes.index('storage-tank')
        polys = torch.where(tank[:, None], bbox2pointobb(bboxes), polys)
    if quantize:
        polys, scores = quantize_text(polys, 1), quantize_text(scores, 3)
    off = polys.new_tensor([x, y] * 4)
    out = polys.new_zeros((polys.size(0), 11))
    out[:, :8] = (polys + off) / rate
    out[:, 8] = scores
    out[:, 9] = labels.to(out.dtype)
    out[:, 10] = float(img)
    return out


def patch_bboxes_to_image(bboxes, scores, labels, patch_name, image_ids, quantize=False):
    """Per-patch HORIZONTAL detections -> (n, 7) rows x1,y1,x2,y2, score, label, image_id in
    image coordinates (the Task2 / mergebyrec leg, dota.py:300,334)."""
    name, rate, x, y = parse_patch_name(patch_name)
    img = image_ids.setdefault(name, len(image_ids))
    if quantize:
        bboxes, scores = quantize_text(bboxes, 1), quantize_text(scores, 3)
    out = bboxes.new_zeros((bboxes.size(0), 7))
    out[:, :4] = (bboxes + bboxes.new_tensor([x, y, x, y])) / rate
    out[:, 4] = scores
    out[:, 5] = labels.to(out.dtype)
    out[:, 6] = float(img)
    return out


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


def _write(out_dir, prefix, classes, labels, lines):
    os.makedirs(out_dir, exist_ok=True)
    counts = {}
    for ci, cname in enumerate(classes):
        sel = np.nonzero(labels == ci)[0]
        counts[cname] = len(sel)
        with open(os.path.join(out_dir, '%s_%s.txt' % (prefix, cname)), 'w') as f:
            f.writelines(lines(i) for i in sel)
    return counts


def write_task1(out_dir, rows, polys, image_names, classes=DOTA_CLASSES, prefix='Task1'):
    """Write merged OBB detections as ``<prefix>_<class>.txt`` files.

    Line format of dota.py:304: ``'%s %.3f %.1f %.1f %.1f %.1f %.1f %.1f %.1f %.1f'``
    (name, score, x1 y1 ... x4 y4), the name being the original image.

    Args:
        rows (Tensor/ndarray): (k, 8) kept theta-OBB rows [.., score, label, image_id] or
            (k, 11) kept polygon rows [x1..y4, score, label, image_id].
        polys (Tensor/ndarray or None): (k, 8) corner points of theta-OBB rows
            (``thetaobb2pointobb``); ignored for polygon rows.
        image_names (sequence): image id -> name.
    Returns:
        dict class name -> number of lines written.
    """
    rows = _np(rows)
    if rows.shape[1] == 11:
        polys, score, labels, image = rows[:, :8], rows[:, 8], rows[:, 9], rows[:, 10]
    else:
        polys, score, labels, image = _np(polys), rows[:, 5], rows[:, 6], rows[:, 7]
    labels = labels.astype(np.int64)

    def line(i):
        p = polys[i]
        return '%s %.3f %.1f %.1f %.1f %.1f %.1f %.1f %.1f %.1f\n' % (
            image_names[int(image[i])], score[i], p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7])
    return _write(out_dir, prefix, classes, labels, line)


def write_task2(out_dir, rows, image_names, classes=DOTA_CLASSES, prefix='Task2'):
    """Write merged HBB detections, line format of dota.py:300:
    ``'%s %.3f %.1f %.1f %.1f %.1f'``.  rows: (k, 7) x1,y1,x2,y2, score, label, image_id."""
    rows = _np(rows)
    labels = rows[:, 5].astype(np.int64)

    def line(i):
        r = rows[i]
        return '%s %.3f %.1f %.1f %.1f %.1f\n' % (image_names[int(r[6])], r[4], r[0], r[1], r[2], r[3])
    return _write(out_dir, prefix, classes, labels, line)

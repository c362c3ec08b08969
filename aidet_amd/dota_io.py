"""DOTA result plumbing around the merged rotated NMS (host side).

The reference writes per-patch detections to ``Task1_<class>.txt`` files
(``'%s %.3f %.1f ... %.1f'`` = patch name, score, 8 polygon coordinates;
mmdet/datasets/dota.py:278-308) and hands them to the external, un-vendored
``wwtool.mergebypoly_mp`` (dota.py:310-336), which -- like DOTA_devkit's
ResultMerge it wraps [external, not in the reference tree] -- recovers each patch's
offset from its file name ``<image>__<rate>__<x>___<y>``, moves the polygons to
image coordinates, runs a per-class polygon NMS per image and writes one line per
surviving detection with the ORIGINAL image name.

Here the detections never leave the GPU as text: ``parse_patch_name`` /
``patch_rows_to_image`` give the rows that ``dota_merge.distributed_merge``
consumes, and ``write_task1`` emits the final files in the reference's format.
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

_PATCH_RE = re.compile(r'^(?P<name>.+)__(?P<rate>[0-9.]+)__(?P<x>-?\d+)___(?P<y>-?\d+)$')


def class_thresholds(classes=DOTA_CLASSES, classwise=True, flat=0.3):
    """threshold per class index, as merge_txt selects them (dota.py:321-331)."""
    return [OBB_NMS_THR[c] if classwise else flat for c in classes]


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


def patch_rows_to_image(rbboxes, labels, patch_name, image_ids):
    """Per-patch detections -> rows for the merge, in ORIGINAL image coordinates.

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


def write_task1(out_dir, rows, polys, image_names, classes=DOTA_CLASSES, prefix='Task1'):
    """Write merged detections as ``<prefix>_<class>.txt`` files.

    Line format of dota.py:304: ``'%s %.3f %.1f %.1f %.1f %.1f %.1f %.1f %.1f %.1f'``
    (name, score, x1 y1 ... x4 y4), the name being the original image.

    Args:
        rows (Tensor/ndarray): (k, 8) kept rows [.., score, label, image_id].
        polys (Tensor/ndarray): (k, 8) their corner points (``thetaobb2pointobb``).
        image_names (sequence): image id -> name.
    Returns:
        dict class name -> number of lines written.
    """
    rows = rows.cpu().numpy() if isinstance(rows, torch.Tensor) else np.asarray(rows)
    polys = polys.cpu().numpy() if isinstance(polys, torch.Tensor) else np.asarray(polys)
    os.makedirs(out_dir, exist_ok=True)
    counts = {}
    labels = rows[:, 6].astype(np.int64)
    for ci, cname in enumerate(classes):
        sel = np.nonzero(labels == ci)[0]
        counts[cname] = len(sel)
        with open(os.path.join(out_dir, '%s_%s.txt' % (prefix, cname)), 'w') as f:
            for i in sel:
                p = polys[i]
                f.write('%s %.3f %.1f %.1f %.1f %.1f %.1f %.1f %.1f %.1f\n' % (
                    image_names[int(rows[i, 7])], rows[i, 5], p[0], p[1], p[2], p[3], p[4], p[5],
                    p[6], p[7]))
    return counts

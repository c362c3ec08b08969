"""Synthetic workloads of the benchmark (SURVEY.md 8d / BASELINE.md 4): plain numpy, no
dependency on the product or on the oracle.

  boxes(n, canvas, seed)     cfg1 / cfg2 / cfg3: (n, 6) theta-OBBs + distinct scores
  tiled_scene(...)           cfg5: large images cut into overlapping 1024^2 patches; every patch
                             "detects" the objects whose centre it contains (so objects in the
                             200-px overlap bands are detected two to four times -- the
                             duplicates the merged NMS exists to remove)
"""
import numpy as np


def boxes(n, canvas, seed):
    """(n,6) fp32: cx,cy ~ U(0,canvas); w,h = exp(U(ln8,ln128)); theta ~ U(-pi/2,0);
    score = (perm(n)+0.5)/n (all distinct)."""
    rng = np.random.default_rng(seed)
    cx = rng.uniform(0, canvas, n)
    cy = rng.uniform(0, canvas, n)
    w = np.exp(rng.uniform(np.log(8.0), np.log(128.0), n))
    h = np.exp(rng.uniform(np.log(8.0), np.log(128.0), n))
    th = rng.uniform(-np.pi / 2, 0.0, n)
    sc = (rng.permutation(n) + 0.5) / n
    return np.stack([cx, cy, w, h, th, sc], axis=1).astype(np.float32)


def patch_starts(size, patch=1024, gap=200):
    """left/top coordinates of the sliding window the DOTA tiling uses (stride = patch - gap;
    a last window is pushed back so that it ends at the image border)."""
    stride = patch - gap
    s = list(range(0, max(size - patch, 0) + 1, stride))
    if s[-1] + patch < size:
        s.append(size - patch)
    return s


def tiled_scene(images=16, image_size=8192, objects_per_image=50000, num_classes=15, seed=11,
                patch=1024, gap=200):
    """cfg5 detections, patch by patch.

    Returns a dict of arrays over all detections, ordered by patch (the order a single process
    visiting the patches 0, 1, 2, ... produces):
      dets (N, 6) fp32   cx, cy, w, h, theta, score in PATCH pixels
      label (N,) int64   class
      patch (N,) int64   global patch number
    and per patch: patch_image (P,), patch_xy (P, 2) offsets, patch_name list
    ('<image>__1__<x>___<y>', the naming mmdet/datasets/dota.py's merge parses).
    """
    rng = np.random.default_rng(seed)
    starts = patch_starts(image_size, patch, gap)
    dets, labels, pidx = [], [], []
    patch_image, patch_xy, names = [], [], []
    p = 0
    for img in range(images):
        obj = boxes(objects_per_image, image_size, seed * 1000 + img)
        lab = rng.integers(0, num_classes, objects_per_image)
        for py in starts:
            for px in starts:
                m = (obj[:, 0] >= px) & (obj[:, 0] < px + patch) & (obj[:, 1] >= py) & (obj[:, 1] < py + patch)
                k = int(m.sum())
                d = obj[m].copy()
                # what a detector would report for the object from THIS patch: a slightly
                # different box and confidence each time
                d[:, 0] += rng.normal(0, 0.7, k).astype(np.float32) - px
                d[:, 1] += rng.normal(0, 0.7, k).astype(np.float32) - py
                d[:, 2:4] *= np.exp(rng.normal(0, 0.02, (k, 2))).astype(np.float32)
                d[:, 4] += rng.normal(0, 0.01, k).astype(np.float32)
                d[:, 5] = np.clip(d[:, 5] * 0.9 + rng.uniform(0, 0.1, k), 0.001, 0.999).astype(np.float32)
                dets.append(d)
                labels.append(lab[m])
                pidx.append(np.full(k, p, np.int64))
                patch_image.append(img)
                patch_xy.append((px, py))
                names.append('P%04d__1__%d___%d' % (img, px, py))
                p += 1
    return {
        'dets': np.concatenate(dets, 0).astype(np.float32),
        'label': np.concatenate(labels, 0).astype(np.int64),
        'patch': np.concatenate(pidx, 0),
        'patch_image': np.asarray(patch_image, np.int64),
        'patch_xy': np.asarray(patch_xy, np.float32),
        'patch_name': names,
        'num_classes': num_classes,
    }

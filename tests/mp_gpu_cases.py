"""Inputs shared by the ranks (tests/mp_gpu_worker.py) and the parent (tests/test_multiproc_gpu.py) of the
multi-process GPU test: plain numpy / torch CPU tensors, no GPU call."""
import numpy as np
import torch

import oracle
import workloads
from aidet_amd import dota_merge

NUM_CLASSES = 5
CLASS_THRS = [0.3, 0.5, 0.1, 0.4, 0.2]
# (layout, gather_result)
LAYOUTS = [('replicated', True), ('sharded', True), ('routed', True), ('routed', False)]
# 'two': two images of nine patches; 'one': ONE patch -- on three ranks two of them have no patch, no detection, and
# (routed) own no image
SCENES = {'two': dict(images=2, image_size=2048, objects=1500), 'one': dict(images=1, image_size=1024, objects=400)}
NUM_PATCHES = 11


def make_patches(num_patches, seed, ties=False):
    """per-patch detections in full-image coordinates, rows [cx..score, label, image]; patch 4 is empty"""
    rng = np.random.default_rng(seed)
    patches = []
    for p in range(num_patches):
        n = 0 if p == 4 else int(rng.integers(1, 60))
        d = oracle.synth_boxes(n, 300, seed * 1000 + p)
        d[:, 0] += (p % 3) * 150      # overlapping patches of 2 large images
        d[:, 1] += ((p // 3) % 2) * 150
        if ties:
            d[:, 5] = np.round(d[:, 5] * 4) / 4
        label = rng.integers(0, NUM_CLASSES, n).astype(np.float32)
        image = np.full(n, p % 2, np.float32)
        patches.append(np.concatenate([d, label[:, None], image[:, None]], 1).astype(np.float32).reshape(n, 8))
    return patches


def rows_of_rank(rank, world, ties):
    patches = make_patches(NUM_PATCHES, 7, ties)
    mine = dota_merge.shard_patches(len(patches), rank, world)
    rows = [torch.from_numpy(patches[i]) for i in mine]
    key = [torch.full((len(patches[i]),), i, dtype=torch.int64) for i in mine]
    local = torch.cat(rows, 0) if rows else torch.zeros((0, 8))
    key = torch.cat(key, 0) if key else torch.zeros((0,), dtype=torch.int64)
    return local, key, len(patches)


def tiled_inputs(rank, world, images=2, image_size=2048, objects=1500):
    scene = workloads.tiled_scene(images=images, image_size=image_size, objects_per_image=objects,
                                  num_classes=NUM_CLASSES, seed=5)
    P = len(scene['patch_name'])
    idx = np.nonzero(np.isin(scene['patch'], np.asarray(dota_merge.shard_patches(P, rank, world))))[0]
    f = torch.from_numpy
    return (f(scene['dets'][idx]), f(scene['label'][idx]), f(scene['patch'][idx]), f(idx.astype(np.int64)),
            f(scene['patch_xy']), f(scene['patch_image']))


def crowd_inputs(rank, world):
    """The scene 'two' with the detections of its FIRST patch replaced by 600 boxes of one class within a few pixels:
    more than a cell of the NMS's grid route holds (they go on its overflow list), one of them huge (it goes on the big
    list).  The worker runs it a second time with the give-up hook on rank 0 only: that rank's per-patch phase is repeated
    and every rank repeats the exchange."""
    scene = workloads.tiled_scene(images=2, image_size=2048, objects_per_image=1500, num_classes=NUM_CLASSES, seed=5)
    rng = np.random.default_rng(9)
    first = np.nonzero(scene['patch'] == 0)[0]
    m = 600
    d = np.zeros((m, 6), np.float32)
    d[:, 0:2] = 300.0 + rng.random((m, 2)).astype(np.float32) * 6.0
    d[:, 2:4] = 24.0 + rng.random((m, 2)).astype(np.float32) * 4.0
    d[:, 4] = -rng.random(m).astype(np.float32)
    d[:, 5] = (rng.permutation(m) + 0.5) / m
    d[0, 2:5] = [3000.0, 2800.0, -0.3]           # and one box over more cells than an entry list may name: the big list
    keep = np.ones(len(scene['dets']), bool)
    keep[first] = False
    dets = np.concatenate([d, scene['dets'][keep]])
    label = np.concatenate([np.zeros(m, np.int64), scene['label'][keep]])
    patch = np.concatenate([np.zeros(m, np.int64), scene['patch'][keep]])
    P = len(scene['patch_name'])
    idx = np.nonzero(np.isin(patch, np.asarray(dota_merge.shard_patches(P, rank, world))))[0]
    f = torch.from_numpy
    return (f(dets[idx]), f(label[idx]), f(patch[idx]), f(idx.astype(np.int64)), f(scene['patch_xy']), f(scene['patch_image']))


def iou_inputs():
    return (torch.from_numpy(workloads.boxes(301, 512, 1)[:, :5].copy()),
            torch.from_numpy(workloads.boxes(257, 512, 2)[:, :5].copy()))


def oracle_nms(dets, iou_thr, groups=None, thrs=None):
    keep = oracle.rnms(dets.cpu().numpy(), float(iou_thr),
                       groups=None if groups is None else groups.cpu().numpy(),
                       thrs=None if thrs is None else thrs.cpu().numpy())
    keep = torch.from_numpy(keep)
    return dets[keep], keep

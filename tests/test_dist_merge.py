"""CPU, world_size = 2, gloo: the multi-GPU gather + merged rotated NMS path
(aidet_amd/dota_merge.py) must give exactly the single-process result.  The NMS
op is injected (the oracle) because the product has no CPU path; on the GPU box
the same code runs with RCCL and the HIP rnms (tests/test_gpu_parity.py)."""
import os
import re as re_
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from aidet_amd import dota_merge

NUM_CLASSES = 5
CLASS_THRS = [0.3, 0.5, 0.1, 0.4, 0.2]


def oracle_nms(dets, iou_thr, groups=None, thrs=None):
    keep = oracle.rnms(dets.cpu().numpy(), float(iou_thr),
                       groups=None if groups is None else groups.cpu().numpy(),
                       thrs=None if thrs is None else thrs.cpu().numpy())
    keep = torch.from_numpy(keep)
    return dets[keep], keep


def make_patches(num_patches, seed, ties=False):
    """per-patch detections in full-image coordinates, rows [cx..score,label,image]"""
    rng = np.random.default_rng(seed)
    patches = []
    for p in range(num_patches):
        n = int(rng.integers(0, 60))
        d = oracle.synth_boxes(n, 300, seed * 1000 + p)
        d[:, 0] += (p % 3) * 150      # overlapping patches of 2 large images
        d[:, 1] += ((p // 3) % 2) * 150
        if ties:
            d[:, 5] = np.round(d[:, 5] * 4) / 4
        label = rng.integers(0, NUM_CLASSES, n).astype(np.float32)
        image = np.full(n, p % 2, np.float32)
        patches.append(np.concatenate([d, label[:, None], image[:, None]], 1).astype(np.float32))
    return patches


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ties, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        patches = make_patches(11, 7, ties)
        mine = dota_merge.shard_patches(len(patches), rank, world)
        rows = [torch.from_numpy(patches[i]) for i in mine]
        seq = [torch.arange(len(patches[i]), dtype=torch.int64) + (i << 20) for i in mine]
        local = torch.cat(rows, 0) if rows else torch.zeros((0, 8))
        seq = torch.cat(seq, 0) if seq else torch.zeros((0,), dtype=torch.int64)
        kept, keep, allrows = dota_merge.distributed_merge(
            local, 0.3, NUM_CLASSES, class_thrs=CLASS_THRS, nms_fn=oracle_nms, seq=seq)
        out[rank] = (kept.numpy(), keep.numpy(), allrows.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('ties', [False, True])
def test_two_rank_merge_equals_single_process(ties):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ties, out), nprocs=world, join=True)
    patches = make_patches(11, 7, ties)
    single = torch.from_numpy(np.concatenate(patches, 0))
    ref_kept, ref_keep = dota_merge.merge_detections(single, 0.3, NUM_CLASSES, CLASS_THRS,
                                                     nms_fn=oracle_nms)
    for r in range(world):
        kept, keep, allrows = out[r]
        assert np.array_equal(allrows, single.numpy())        # canonical order restored
        assert np.array_equal(keep, ref_keep.numpy())         # identical keep indices
        assert np.array_equal(kept, ref_kept.numpy())
    assert 0 < len(ref_keep) < len(single)


def test_merge_is_per_image_and_class():
    """group = image * num_classes + label: equals independent NMS per (image, class)
    with that class's threshold (the per-class files of dota.py:310-336)."""
    rows = torch.from_numpy(np.concatenate(make_patches(6, 3), 0))
    _, keep = dota_merge.merge_detections(rows, 0.3, NUM_CLASSES, CLASS_THRS, nms_fn=oracle_nms)
    want = []
    r = rows.numpy()
    for img in (0, 1):
        for c in range(NUM_CLASSES):
            idx = np.nonzero((r[:, 7] == img) & (r[:, 6] == c))[0]
            if len(idx):
                want.append(idx[oracle.rnms(r[idx, :6], CLASS_THRS[c])])
    assert np.array_equal(keep.numpy(), np.sort(np.concatenate(want)))


def test_single_process_without_process_group():
    rows = torch.from_numpy(np.concatenate(make_patches(4, 1), 0))
    kept, keep, allrows = dota_merge.distributed_merge(rows, 0.3, NUM_CLASSES, nms_fn=oracle_nms)
    assert torch.equal(allrows, rows) and len(keep) == len(kept)
    empty = torch.zeros((0, 8))
    kept, keep, _ = dota_merge.distributed_merge(empty, 0.3, NUM_CLASSES, nms_fn=oracle_nms)
    assert kept.shape == (0, 8) and keep.numel() == 0


def test_shard_patches_is_rank_strided():
    assert dota_merge.shard_patches(10, 1, 4) == [1, 5, 9]
    got = sorted(sum((dota_merge.shard_patches(10, r, 4) for r in range(4)), []))
    assert got == list(range(10))


def test_dota_io_names_rows_and_files(tmp_path):
    """patch-name parsing, coordinate restoration and the Task1 line format
    (mmdet/datasets/dota.py:304)."""
    from aidet_amd import dota_io
    assert dota_io.parse_patch_name('P0003__1__824___0') == ('P0003', 1.0, 824, 0)
    assert dota_io.parse_patch_name('/a/b/P12__0.5__0___1648.png') == ('P12', 0.5, 0, 1648)
    with pytest.raises(ValueError):
        dota_io.parse_patch_name('P0003.png')
    ids = {}
    rb = torch.tensor([[10., 20., 8., 4., -0.3, 0.9], [100., 50., 30., 10., -1.0, 0.5]])
    rows = dota_io.patch_rows_to_image(rb, torch.tensor([1, 3]), 'P7__0.5__100___200', ids)
    assert ids == {'P7': 0}
    assert torch.allclose(rows[0], torch.tensor([220., 440., 16., 8., -0.3, 0.9, 1., 0.]))
    thr = dota_io.class_thresholds()
    assert len(thr) == 15 and thr[1] == 0.05 and dota_io.class_thresholds(classwise=False)[3] == 0.3
    polys = oracle.rbox_to_poly(rows.numpy())
    counts = dota_io.write_task1(str(tmp_path), rows, polys, ['P7'])
    assert counts['ship'] == 1 and counts['large-vehicle'] == 1 and sum(counts.values()) == 2
    line = open(os.path.join(str(tmp_path), 'Task1_ship.txt')).read().split()
    assert line[0] == 'P7' and line[1] == '0.900' and len(line) == 10
    assert all(re_.match(r'^-?\d+\.\d$', v) for v in line[2:])

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


def _worker(rank, world, port, ties, layout, out):
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
            local, 0.3, NUM_CLASSES, class_thrs=CLASS_THRS, nms_fn=oracle_nms, seq=seq, layout=layout)
        out[rank] = (kept.numpy(), keep.numpy(), allrows.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('ties,layout,world', [(False, 'replicated', 2), (True, 'replicated', 2),
                                               (True, 'sharded', 2), (False, 'sharded', 3)])
def test_multi_rank_merge_equals_single_process(ties, layout, world):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ties, layout, out), nprocs=world, join=True)
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


def test_merge_row_layouts_polygon_and_horizontal():
    """11-column polygon rows (Task1 text rows) and 7-column horizontal rows (Task2 /
    mergebyrec, dota.py:300,334) go through poly_nms / nms with the same grouping."""
    from aidet_amd import dota_io
    r = np.concatenate(make_patches(6, 5), 0)
    poly = np.concatenate([oracle.rbox_to_poly(r).astype(np.float32), r[:, 5:8]], 1)

    def poly_fn(d, thr, groups=None, thrs=None):
        k = torch.from_numpy(oracle.poly_nms(d.numpy(), thr, groups=groups.numpy(), thrs=thrs.numpy()))
        return d[k], k

    def hbb_fn(d, thr, groups=None, thrs=None, strict=False):
        assert thrs is None or len(set(thrs.tolist())) == 1      # (the oracle's HBB loop takes one threshold)
        k = torch.from_numpy(oracle.hbb_nms(d.numpy(), thr if thrs is None else float(thrs[0]),
                                            groups=groups.numpy(), strict=strict))
        return d[k], k
    _, keep_p = dota_merge.merge_detections(torch.from_numpy(poly), 0.3, NUM_CLASSES, CLASS_THRS, nms_fn=poly_fn)
    _, keep_r = dota_merge.merge_detections(torch.from_numpy(r), 0.3, NUM_CLASSES, CLASS_THRS, nms_fn=oracle_nms)
    assert np.array_equal(keep_p.numpy(), keep_r.numpy())        # same boxes, two encodings
    hb = np.stack([r[:, 0] - r[:, 2] / 2, r[:, 1] - r[:, 3] / 2, r[:, 0] + r[:, 2] / 2, r[:, 1] + r[:, 3] / 2,
                   r[:, 5], r[:, 6], r[:, 7]], 1).astype(np.float32)
    kept, keep_h = dota_merge.merge_detections(torch.from_numpy(hb), 0.3, NUM_CLASSES, nms_fn=hbb_fn)
    want = []
    for img in (0, 1):
        for c in range(NUM_CLASSES):
            idx = np.nonzero((hb[:, 6] == img) & (hb[:, 5] == c))[0]
            if len(idx):
                want.append(idx[oracle.hbb_nms(hb[idx, :5], 0.3)])
    assert np.array_equal(keep_h.numpy(), np.sort(np.concatenate(want))) and kept.shape[1] == 7
    _, keep_s = dota_merge.merge_detections(torch.from_numpy(hb), 0.3, NUM_CLASSES, nms_fn=hbb_fn, strict=True)
    assert len(keep_s) >= len(keep_h)
    with pytest.raises(ValueError):
        dota_merge.merge_detections(torch.zeros((3, 9)), 0.3, NUM_CLASSES, nms_fn=oracle_nms)
    assert dota_io.class_thresholds(task='hbb')[4] == 0.1 and dota_io.class_thresholds(task='obb')[4] == 0.35


def test_text_round_trip_quantisation_is_exact():
    """quantize_text == float('%.1f' % v) for every fp32 v (what dota.py:304 does to the
    numbers the external merge reads), incl. the decimal ties fp32 can hit."""
    from aidet_amd import dota_io
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.uniform(-50, 1100, 20000), rng.integers(-2000, 22000, 4000) / 20.0,
                        rng.integers(0, 4000, 2000) / 8.0, [0.05, 0.15, 0.25, 0.35, 1023.95, -0.05, 0.0]]).astype(np.float32)
    got = dota_io.quantize_text(torch.from_numpy(v), 1).numpy()
    want = np.array([np.float32(float('%.1f' % x)) for x in v.tolist()], np.float32)
    assert np.array_equal(got, want)
    s = rng.uniform(0, 1, 5000).astype(np.float32)
    got = dota_io.quantize_text(torch.from_numpy(s), 3).numpy()
    assert np.array_equal(got, np.array([np.float32(float('%.3f' % x)) for x in s.tolist()], np.float32))


def test_polygon_and_box_rows_files_and_storage_tank(tmp_path):
    from aidet_amd import dota_io
    ids = {}
    tank = dota_io.DOTA_CLASSES.index('storage-tank')
    polys = torch.tensor([[10.04, 20.06, 30.0, 20.0, 30.0, 40.0, 10.0, 40.0], [1., 2., 5., 2., 5., 9., 1., 9.]])
    bboxes = torch.tensor([[9.96, 19.0, 31.0, 41.0], [0., 0., 6., 10.]])
    labels = torch.tensor([tank, 1])
    rows = dota_io.patch_polys_to_image(polys, torch.tensor([0.91234, 0.5]), labels, 'P1__1__100___200', ids,
                                        quantize=True, bboxes=bboxes, storage_tank_as_hbb=True)
    assert rows.shape == (2, 11) and ids == {'P1': 0}
    # the storage tank's polygon is its (quantised) horizontal box, dota.py:302-303
    assert torch.allclose(rows[0, :8], torch.tensor([110.0, 219.0, 131.0, 219.0, 131.0, 241.0, 110.0, 241.0]))
    assert abs(float(rows[0, 8]) - 0.912) < 1e-7 and rows[0, 9] == tank
    assert torch.allclose(rows[1, :8], torch.tensor([101., 202., 105., 202., 105., 209., 101., 209.]))
    hb = dota_io.patch_bboxes_to_image(bboxes, torch.tensor([0.9, 0.5]), labels, 'P1__0.5__100___200', ids, quantize=True)
    assert torch.allclose(hb[0, :4], torch.tensor([220.0, 438.0, 262.0, 482.0])) and hb.shape == (2, 7)
    c1 = dota_io.write_task1(str(tmp_path), rows, None, ['P1'])
    c2 = dota_io.write_task2(str(tmp_path), hb, ['P1'])
    assert c1['storage-tank'] == 1 and c2['ship'] == 1
    l1 = open(os.path.join(str(tmp_path), 'Task1_storage-tank.txt')).read().split()
    assert l1[:3] == ['P1', '0.912', '110.0'] and len(l1) == 10
    l2 = open(os.path.join(str(tmp_path), 'Task2_ship.txt')).read().split()
    assert l2 == ['P1', '0.500', '200.0', '400.0', '212.0', '420.0']


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


# ---- configs[4] as one function: per-patch NMS -> image coords -> gather -> merged NMS ---------
def _tiled_inputs(rank, world, images=2, image_size=2048, objects=1500):
    import workloads
    scene = workloads.tiled_scene(images=images, image_size=image_size, objects_per_image=objects,
                                  num_classes=NUM_CLASSES, seed=5)
    P = len(scene['patch_name'])
    idx = np.nonzero(np.isin(scene['patch'], np.asarray(dota_merge.shard_patches(P, rank, world))))[0]
    f = lambda a: torch.from_numpy(a)
    return (f(scene['dets'][idx]), f(scene['label'][idx]), f(scene['patch'][idx]), f(idx.astype(np.int64)),
            f(scene['patch_xy']), f(scene['patch_image'])), scene


def _tiled_worker(rank, world, port, layout, out, scene_kw=None):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        args, _ = _tiled_inputs(rank, world, **(scene_kw or {}))
        kept, keep, allrows = dota_merge.tiled_inference_merge(*args, NUM_CLASSES, 0.5, 0.3, CLASS_THRS,
                                                               layout=layout, nms_fn=oracle_nms)
        out[rank] = (kept.numpy(), keep.numpy(), allrows.numpy())
        if rank == 0:
            # what bench.py's cross-check does: the whole job on rank 0 ALONE while a process group
            # exists (group=LOCAL: no collective is entered, the other rank is not waiting in one)
            solo, _ = _tiled_inputs(0, 1, **(scene_kw or {}))
            k1, _, _ = dota_merge.tiled_inference_merge(*solo, NUM_CLASSES, 0.5, 0.3, CLASS_THRS, layout=layout,
                                                        nms_fn=oracle_nms, group=dota_merge.LOCAL)
            out['solo'] = k1.numpy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('layout', ['replicated', 'sharded'])
def test_tiled_inference_merge_two_ranks_equal_one(layout):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_tiled_worker, args=(world, _free_port(), layout, out), nprocs=world, join=True)
    args, scene = _tiled_inputs(0, 1)
    marks = []
    kept, keep, allrows = dota_merge.tiled_inference_merge(*args, NUM_CLASSES, 0.5, 0.3, CLASS_THRS,
                                                           nms_fn=oracle_nms, mark=marks.append)
    assert marks == ['local', 'gather', 'merge']
    for r in range(world):
        assert np.array_equal(out[r][2], allrows.numpy())
        assert np.array_equal(out[r][1], keep.numpy()) and np.array_equal(out[r][0], kept.numpy())
    assert np.array_equal(out['solo'], kept.numpy())
    # the overlap bands really produce duplicates, and the merge removes them
    n = len(scene['dets'])
    assert len(allrows) <= n and 0.3 * n < len(keep) < 0.95 * len(allrows)
    # image coordinates: every kept centre lies inside its image
    assert float(kept[:, 0].min()) > -5 and float(kept[:, 0].max()) < 2048 + 5


@pytest.mark.parametrize('layout', ['replicated', 'sharded'])
def test_tiled_inference_merge_with_ranks_that_have_nothing(layout):
    """One image of one patch on three ranks: ranks 1 and 2 contribute no detections at all (empty local
    NMS, zero-row gather contributions, no groups of their own in the sharded merge)."""
    world, kw = 3, dict(images=1, image_size=1024, objects=400)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_tiled_worker, args=(world, _free_port(), layout, out, kw), nprocs=world, join=True)
    args, scene = _tiled_inputs(0, 1, **kw)
    assert len(scene['patch_name']) == 1
    kept, keep, allrows = dota_merge.tiled_inference_merge(*args, NUM_CLASSES, 0.5, 0.3, CLASS_THRS, nms_fn=oracle_nms)
    for r in range(world):
        assert np.array_equal(out[r][2], allrows.numpy())
        assert np.array_equal(out[r][1], keep.numpy()) and np.array_equal(out[r][0], kept.numpy())
    assert 0 < len(keep) <= len(allrows)


def _routed_worker(rank, world, port, ties, gather_result, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        patches = make_patches(11, 7, ties)
        mine = dota_merge.shard_patches(len(patches), rank, world)
        rows = [torch.from_numpy(patches[i]) for i in mine]
        key = [torch.full((len(patches[i]),), i, dtype=torch.int64) for i in mine]
        local = torch.cat(rows, 0) if rows else torch.zeros((0, 8))
        key = torch.cat(key, 0) if key else torch.zeros((0,), dtype=torch.int64)
        marks = []
        kept, keep, routed = dota_merge.distributed_merge(
            local, 0.3, NUM_CLASSES, class_thrs=CLASS_THRS, nms_fn=oracle_nms, layout='routed', order_key=key,
            order_key_max=len(patches), gather_result=gather_result, mark=marks.append)
        assert marks == ['gather', 'merge']
        out[rank] = (kept.numpy(), keep.numpy(), routed.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('ties,world', [(False, 2), (True, 2), (True, 3)])
def test_routed_merge_equals_single_process(ties, world):
    """layout='routed': rows travel only to the owner of their image (image % world) by all_to_all; the kept rows,
    gathered back, are the single-process result in the single-process order -- score ties included (the patch index
    restores the interleave of the rank-strided patches)."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_routed_worker, args=(world, _free_port(), ties, True, out), nprocs=world, join=True)
    patches = make_patches(11, 7, ties)
    single = np.concatenate(patches, 0)
    ref_kept, _ = dota_merge.merge_detections(torch.from_numpy(single), 0.3, NUM_CLASSES, CLASS_THRS, nms_fn=oracle_nms)
    for r in range(world):
        kept, keep, routed = out[r]
        assert np.array_equal(kept, ref_kept.numpy())
        # this rank received exactly the rows of its images, in single-process order
        assert np.array_equal(routed, single[single[:, 7].astype(np.int64) % world == r])
    assert 0 < len(ref_kept) < len(single)


def test_routed_merge_without_the_final_gather_leaves_results_with_the_owner():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_routed_worker, args=(world, _free_port(), True, False, out), nprocs=world, join=True)
    single = np.concatenate(make_patches(11, 7, True), 0)
    ref_kept, _ = dota_merge.merge_detections(torch.from_numpy(single), 0.3, NUM_CLASSES, CLASS_THRS, nms_fn=oracle_nms)
    ref = ref_kept.numpy()
    for r in range(world):
        kept, keep, routed = out[r]
        assert np.array_equal(kept, ref[ref[:, 7].astype(np.int64) % world == r])
        assert np.array_equal(kept, routed[keep])


@pytest.mark.parametrize('world,kw', [(2, None), (3, dict(images=1, image_size=1024, objects=400))])
def test_tiled_inference_merge_routed_equals_one_rank(world, kw):
    """configs[4] with the routed exchange on 2 ranks, and on 3 ranks of which two own no image and one has no
    detections at all (empty all_to_all contributions)."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_tiled_worker, args=(world, _free_port(), 'routed', out, kw), nprocs=world, join=True)
    args, scene = _tiled_inputs(0, 1, **(kw or {}))
    kept, keep, allrows = dota_merge.tiled_inference_merge(*args, NUM_CLASSES, 0.5, 0.3, CLASS_THRS, nms_fn=oracle_nms)
    for r in range(world):
        assert np.array_equal(out[r][0], kept.numpy())
    assert np.array_equal(out['solo'], kept.numpy())
    assert 0 < len(keep) <= len(allrows)


def test_rbbox_overlaps_row_blocks_tile_the_matrix():
    """rbbox_overlaps partitions by ROW BLOCKS with no communication; the blocks of all ranks
    concatenated are every row exactly once, also when the rows do not divide evenly or ranks get none."""
    from aidet_amd.rbbox_geometry import row_block
    for m in (0, 1, 7, 16, 100, 16384):
        for world in (1, 2, 3, 8):
            blocks = [row_block(m, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == m
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:])) and all(lo <= hi for lo, hi in blocks)
    with pytest.raises(ValueError):
        row_block(10, 4, 4)


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """`--gpus N` must not be silently ignored.  (Fails before torch is imported.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK='0', WORLD_SIZE='1')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '4'], env=env,
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and 'disagree' in r.stderr

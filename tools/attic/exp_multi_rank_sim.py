"""What ONE rank of an N-rank config-5 job computes, on one GPU: the collectives are replaced by the
precomputed contributions of the other ranks (so RCCL latency is NOT in these numbers), everything else
is the code path of `bench.py --gpus N`.  Phases by HIP events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import workloads
from aidet_amd import dota_io, dota_merge

dev = torch.device('cuda', 0)
scene = workloads.tiled_scene()
C, P = scene['num_classes'], len(scene['patch_name'])
patch_xy = torch.from_numpy(scene['patch_xy']).to(dev)
patch_image = torch.from_numpy(scene['patch_image']).to(dev)
thrs = dota_io.class_thresholds()
real_gather, real_world = dota_merge.gather_detections, dota_merge._world


def rank_inputs(rank, world):
    idx = np.nonzero(np.isin(scene['patch'], np.asarray(dota_merge.shard_patches(P, rank, world))))[0]
    return (torch.from_numpy(scene['dets'][idx]).to(dev), torch.from_numpy(scene['label'][idx]).to(dev),
            torch.from_numpy(scene['patch'][idx]).to(dev), torch.from_numpy(idx.astype(np.int64)).to(dev))


ONLY = os.environ.get('SIM_ONLY', '')     # e.g. 'routed8': one configuration (for a kernel trace)
for world in ((1, ) if ONLY == 'single' else () if ONLY else (1, 2, 4, 8)):
    layout = 'sharded' if world > 1 else 'replicated'
    inputs = [rank_inputs(r, world) for r in range(world)]
    # contributions of the other ranks to the two gathers (rows + sequence columns; keep indices)
    others_rows, others_keep = [], []
    if world > 1:
        allparts = []
        for r in range(world):
            d, l, p, s = inputs[r]
            k = dota_merge.local_patch_nms(d, l, p, 0.5, C)
            rows = dota_merge.patch_rows_to_image(d, l, p, patch_xy, patch_image, keep=k)
            allparts.append(torch.cat([rows, p[k].float()[:, None]], 1))
        others_rows = allparts
        counts = [int(t.size(0)) for t in allparts]
        nmax = max(counts)
        padded0 = allparts[0].new_zeros((world * nmax, 9))
        padded0[:, -1] = dota_merge._PAD_KEY
        for r, t in enumerate(allparts):
            padded0[r * nmax:r * nmax + counts[r]] = t
        allrows = dota_merge._restore_by_key(padded0, sum(counts), 8, P, world)
        others_mask = torch.zeros(allrows.size(0), dtype=torch.uint8, device=dev)
        for r in range(1, world):
            others_mask[dota_merge.merge_detections(allrows, 0.3, C, thrs, only_groups=(world, r))[1]] = 1
    calls = {'n': 0}

    def fake_gather_padded(local, group=None, pad_last=0.0):
        calls['n'] += 1
        if world == 1:
            return local, [local.size(0)], local.size(0)
        padded0[:local.size(0)] = local          # (this rank's rows; the others' are "received")
        return padded0, counts, nmax

    def fake_all_reduce(t, op=None, group=None):
        torch.maximum(t, others_mask, out=t)       # what the other ranks' masks contribute

    real_gp = dota_merge._gather_padded
    dota_merge._gather_padded = fake_gather_padded
    real_all_reduce = dota_merge.dist.all_reduce
    dota_merge.dist.all_reduce = fake_all_reduce
    dota_merge._world = lambda group: (world, 0)
    d, l, p, s = inputs[0]
    ev = {}

    def mark(name):
        e = torch.cuda.Event(enable_timing=True); e.record(); ev[name] = e

    def step(timed=False):
        if timed:
            mark('start')
        return dota_merge.tiled_inference_merge(d, l, p, s, patch_xy, patch_image, C, 0.5, 0.3, thrs, layout=layout,
                                                mark=mark if timed else None)
    for _ in range(3):
        kept, keep, allr = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    step(True); torch.cuda.synchronize()
    print('ranks %d (%s): rank 0 computes %.3f ms per pass  [local nms %.0f us | restore order %.0f us | merge %.0f us]  kept %d'
          % (world, layout, ms, ev['start'].elapsed_time(ev['local']) * 1e3, ev['local'].elapsed_time(ev['gather']) * 1e3,
             ev['gather'].elapsed_time(ev['merge']) * 1e3, keep.numel()), flush=True)
    dota_merge._gather_padded, dota_merge._world = real_gp, real_world
    dota_merge.dist.all_reduce = real_all_reduce


# ---- the ROUTED layout: rows travel only to the owner of their image (image % world) by all_to_all ----------------
for world in ((int(ONLY[6:]), ) if ONLY.startswith('routed') else () if ONLY else (2, 4, 8)):
    inputs = [rank_inputs(r, world) for r in range(world)]
    parts = []
    for r in range(world):
        d, l, p, s = inputs[r]
        k = dota_merge.local_patch_nms(d, l, p, 0.5, C)
        rows = dota_merge.patch_rows_to_image(d, l, p, patch_xy, patch_image, keep=k)
        parts.append(torch.cat([rows, p[k].float()[:, None]], 1))
    # what every source sends to rank 0, in source order (a source's rows sorted by destination keep their order)
    to0 = [t[(t[:, 7].long() % world) == 0] for t in parts]
    others = torch.cat(to0[1:], 0)            # what the all_to_all delivers from the other ranks, in source order
    mat = torch.stack([torch.cat([torch.bincount(t[:, 7].long() % world, minlength=world), torch.zeros(1, dtype=torch.int64, device=dev)])
                       for t in parts]).to(dev)   # mat[src][dst] | unordered keys

    def fake_allgather(out, inp, group=None):
        out.copy_(mat.reshape(-1))            # (row 0 is what this rank just counted itself)

    def fake_all_to_all(recv, send, out_splits, in_splits, group=None):
        o = in_splits[0]
        recv[:o] = send[:o]                   # this rank's own rows for itself
        recv[o:] = others                     # the others' rows are "received"

    real = (dota_merge.dist.all_gather_into_tensor, dota_merge.dist.all_to_all_single, dota_merge._world)
    dota_merge.dist.all_gather_into_tensor, dota_merge.dist.all_to_all_single = fake_allgather, fake_all_to_all
    dota_merge._world = lambda group: (world, 0)
    d, l, p, s = inputs[0]
    ev = {}

    def mark(name):
        e = torch.cuda.Event(enable_timing=True); e.record(); ev[name] = e

    def step(timed=False):
        if timed:
            mark('start')
        return dota_merge.tiled_inference_merge(d, l, p, s, patch_xy, patch_image, C, 0.5, 0.3, thrs, layout='routed',
                                                mark=mark if timed else None, gather_result=False)
    for _ in range(3):
        kept, keep, mine = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    step(True); torch.cuda.synchronize()
    print('ranks %d (routed): rank 0 computes %.3f ms per pass  [local nms %.0f us | route + restore order %.0f us | merge %.0f us]  '
          'rows merged here %d, kept here %d' % (world, ms, ev['start'].elapsed_time(ev['local']) * 1e3,
                                                 ev['local'].elapsed_time(ev['gather']) * 1e3, ev['gather'].elapsed_time(ev['merge']) * 1e3,
                                                 mine.size(0), keep.numel()), flush=True)
    dota_merge.dist.all_gather_into_tensor, dota_merge.dist.all_to_all_single, dota_merge._world = real

"""What ONE rank of an N-rank config-5 job computes, measured on ONE GPU.

The collectives of aidet_amd/_collectives.py are replaced by copies of the PRECOMPUTED contributions of the other
ranks (their local phases run beforehand on this GPU), so RCCL / xGMI latency is NOT in these numbers; everything
else is the code path of `bench.py --gpus N` (dota_merge.tiled_inference_merge, fused phases).  Rank 0's kept rows are
checked against the single-rank run.

    python tools/sim_multi_rank.py [--scene default|large] [--layouts routed,replicated] [--worlds 1,2,4,8] [--json out]

scenes: default = BASELINE configs[4] as bench.py runs it (16 images of 8192^2, 1296 patches, 1.25 M detections);
        large   = 128 such images (10 368 patches, 1.0e7 detections): the size where strong scaling has room.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import workloads
from aidet_amd import dota_io, dota_merge

ap = argparse.ArgumentParser()
ap.add_argument('--scene', default='default', choices=['default', 'large'])
ap.add_argument('--layouts', default='routed,replicated')
ap.add_argument('--worlds', default='1,2,4,8')
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--json', default=None)
args = ap.parse_args()

dev = torch.device('cuda', 0)
scene = workloads.tiled_scene(images=128 if args.scene == 'large' else 16)
C, P = scene['num_classes'], len(scene['patch_name'])
N = len(scene['dets'])
patch_xy = torch.from_numpy(scene['patch_xy']).to(dev)
patch_image = torch.from_numpy(scene['patch_image']).to(dev)
thrs = dota_io.class_thresholds()
coll = dota_merge.coll
print('scene %s: %d images, %d patches, %d detections' % (args.scene, int(scene['patch_image'].max()) + 1, P, N), flush=True)


def rank_inputs(rank, world):
    idx = np.nonzero(np.isin(scene['patch'], np.asarray(dota_merge.shard_patches(P, rank, world))))[0]
    return (torch.from_numpy(scene['dets'][idx]).to(dev), torch.from_numpy(scene['label'][idx]).to(dev),
            torch.from_numpy(scene['patch'][idx]).to(dev), torch.from_numpy(idx.astype(np.int64)).to(dev))


def timed(step, steps):
    ev = {}

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev[name] = e
    for _ in range(3):
        out = step(None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(None)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    mark('start')
    step(mark)
    torch.cuda.synchronize()
    ph = [ev['start'].elapsed_time(ev['local']) * 1e3, ev['local'].elapsed_time(ev['gather']) * 1e3,
          ev['gather'].elapsed_time(ev['merge']) * 1e3]
    return ms, ph, out


results = {'scene': args.scene, 'detections': N, 'patches': P, 'rows': []}
single_kept = None
for world in [int(w) for w in args.worlds.split(',')]:
    for layout in (['replicated'] if world == 1 else args.layouts.split(',')):
        inputs = [rank_inputs(r, world) for r in range(world)]
        route_world = world if layout == 'routed' else 1
        # the other ranks' local phases (what they would send)
        counts_rows, rows_of = [], []
        for r in range(world):
            d, l, p, s = inputs[r]
            rows, counts = dota_merge.tiled_local_phase(d, l, p, patch_xy, patch_image, C, 0.5, route_world)
            c = counts.tolist()
            assert c[route_world + 1] >= 0
            counts_rows.append(c)
            rows_of.append(rows[:c[route_world + 1]].clone())
        width = len(counts_rows[0])
        others_counts = torch.tensor(counts_rows, dtype=torch.int64, device=dev)
        if layout == 'routed':
            # rows every other source sends to rank 0 = its bucket 0 = its first counts[0] rows
            others_recv = torch.cat([rows_of[r][:counts_rows[r][0]] for r in range(1, world)], 0) if world > 1 else None
        else:
            ks = [c[route_world + 1] for c in counts_rows]
            nmax = max(max(ks), 1)
            others_pad = rows_of[0].new_zeros((world * nmax, 9))
            others_pad[:, -1] = dota_merge._PAD_KEY
            for r in range(1, world):
                others_pad[r * nmax:r * nmax + ks[r]] = rows_of[r]

        def fake_all_gather(out, inp, group=None):
            if out.dtype == torch.int64:                 # the count vectors
                out.copy_(others_counts.reshape(-1))
                out[:width] = inp                        # (this rank's own, just computed)
            else:                                        # replicated: the padded rows of everybody
                out.copy_(others_pad)
                out[:inp.size(0)] = inp
            return out

        def fake_all_to_all(recv, send, out_splits, in_splits, group=None):
            o = in_splits[0]
            recv[:o] = send[:o]
            recv[o:] = others_recv
            return recv
        real = (coll.all_gather_into_tensor, coll.all_to_all_single, dota_merge._world)
        coll.all_gather_into_tensor, coll.all_to_all_single = fake_all_gather, fake_all_to_all
        dota_merge._world = lambda group, _w=world: (_w, 0)
        d, l, p, s = inputs[0]
        try:
            ms, ph, (kept, keep, mine) = timed(lambda mark: dota_merge.tiled_inference_merge(
                d, l, p, s, patch_xy, patch_image, C, 0.5, 0.3, thrs, layout=layout, mark=mark, gather_result=False), args.steps)
        finally:
            coll.all_gather_into_tensor, coll.all_to_all_single, dota_merge._world = real
        if world == 1:
            single_kept, single_ms = kept.clone(), ms
        ok = None
        if single_kept is not None:
            want = single_kept if layout != 'routed' else single_kept[(single_kept[:, 7].long() % world) == 0]
            ok = bool(torch.equal(kept, want))
        row = {'ranks': world, 'layout': layout, 'ms_per_pass_rank0': round(ms, 4), 'local_us': round(ph[0], 1),
               'exchange_side_us': round(ph[1], 1), 'merge_us': round(ph[2], 1), 'rows_merged_here': int(mine.size(0)),
               'kept_here': int(keep.numel()), 'equals_single_rank': ok,
               'speedup_vs_1': round(single_ms / ms, 2) if single_kept is not None else None}
        results['rows'].append(row)
        print('ranks %d (%s): rank 0 computes %.3f ms per pass  [local %.0f us | exchange-side %.0f us | merge %.0f us]  rows merged here %d, '
              'kept here %d, == single rank: %s, speed-up %s' % (world, layout, ms, ph[0], ph[1], ph[2], mine.size(0), keep.numel(), ok,
                                                                 row['speedup_vs_1']), flush=True)
        del inputs, rows_of
if args.json:
    with open(args.json, 'w') as f:
        json.dump(results, f, indent=1)

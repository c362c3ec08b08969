"""One rank of tests/test_multiproc_gpu.py: a FRESH process (started before it touches the GPU) that joins a gloo
group, puts its share of the patches on cuda:0 -- all ranks share the one GPU of the box -- and runs the multi-rank
tiled merge (BASELINE configs[4]) with the real HIP operators; the collectives are staged through host memory by
aidet_amd/_collectives.py.  Every result is written to <outdir>/rank<r>.npz for the parent to compare.

usage: python mp_gpu_worker.py <outdir>     (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT from the environment)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    outdir = sys.argv[1]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    import numpy as np
    import torch
    import torch.distributed as dist

    import mp_gpu_cases as cases
    from aidet_amd import dota_merge
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    res = {}
    force_grid = os.environ.get('OBB_TEST_GRID') == '1'
    if force_grid:       # every NMS call of 256 boxes or more on the uniform-grid route (the scenes here are small)
        from aidet_amd import _lib
        _lib.set_tuning(_lib.TUNE_NMS_DIRECT_MAX, 0)
        _lib.set_tuning(_lib.TUNE_NMS_GRID, 1)
    try:
        # (a) rows with score ties through distributed_merge, every layout
        for ties in (False, True):
            local, key, npatch = cases.rows_of_rank(rank, world, ties)
            local, key = local.to(dev), key.to(dev)
            for layout, gather in cases.LAYOUTS:
                kept, keep, rows = dota_merge.distributed_merge(
                    local, 0.3, cases.NUM_CLASSES, class_thrs=cases.CLASS_THRS, layout=layout, order_key=key,
                    order_key_max=npatch, gather_result=gather)
                tag = 'rows_%d_%s_%d' % (ties, layout, gather)
                res[tag + '_kept'], res[tag + '_keep'], res[tag + '_rows'] = (x.cpu().numpy() for x in (kept, keep, rows))
        # (b) the whole pass of configs[4]: per-patch NMS -> image coordinates -> exchange -> merged NMS
        for name, kw in cases.SCENES.items():
            args = [a.to(dev) for a in cases.tiled_inputs(rank, world, **kw)]
            for layout, gather in cases.LAYOUTS:
                kept, keep, rows = dota_merge.tiled_inference_merge(*args, cases.NUM_CLASSES, 0.5, 0.3, cases.CLASS_THRS,
                                                                    layout=layout, gather_result=gather)
                tag = 'tiled_%s_%s_%d' % (name, layout, gather)
                res[tag + '_kept'], res[tag + '_keep'], res[tag + '_rows'] = (x.cpu().numpy() for x in (kept, keep, rows))
        if force_grid:   # (b') a rank whose per-patch NMS gives the grid up: the repeat word travels with the count exchange
            args = [a.to(dev) for a in cases.crowd_inputs(rank, world)]
            if rank == 0:    # (OBB_TUNE_NMS_GRID = 2: the route gives up at once -- here on one rank only)
                _lib.set_tuning(_lib.TUNE_NMS_GRID, 2)
            for layout, gather in cases.LAYOUTS:
                kept, keep, rows = dota_merge.tiled_inference_merge(*args, cases.NUM_CLASSES, 0.5, 0.3, cases.CLASS_THRS,
                                                                    layout=layout, gather_result=gather)
                tag = 'tiled_crowd_%s_%d' % (layout, gather)
                res[tag + '_kept'], res[tag + '_keep'], res[tag + '_rows'] = (x.cpu().numpy() for x in (kept, keep, rows))
            res['phase_repeats'] = np.asarray(dota_merge.phase_repeats)
        # (c) row blocks of rbbox_overlaps: no communication, the blocks tile the matrix
        from aidet_amd import rbbox_overlaps
        a, b = cases.iou_inputs()
        res['iou_block'] = rbbox_overlaps(a.to(dev), b.to(dev), row_shard=(rank, world)).cpu().numpy()
        torch.cuda.synchronize()
        np.savez(os.path.join(outdir, 'rank%d.npz' % rank), **res)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

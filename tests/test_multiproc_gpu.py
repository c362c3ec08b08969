"""BASELINE configs[4] with MORE THAN ONE PROCESS on the HIP path.

The box has one GPU and RCCL refuses two ranks on one device, so the ranks are 2 / 3 fresh child processes that
share cuda:0 and talk over gloo (collectives staged through host memory, aidet_amd/_collectives.py).  Everything else is
the product: rank-strided patches (distributed_sampler.py:18-25), the grouped per-patch HIP rnms, obb_rows_route /
all_to_all / obb_rows_restore, the merged HIP NMS, all three layouts.  Every rank's result must equal the
single-process HIP run AND the same pipeline with the oracle's NMS injected (mmdet/apis/test.py:31-146 gathers
the same rows; the merge is dota.py:310-336).  What stays untested here: the RCCL transport itself and xGMI timing.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import mp_gpu_cases as cases
from aidet_amd import dota_merge

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_ranks(world, outdir, grid=False):
    """start `world` child processes (each initialises the GPU itself) and wait for all of them"""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
                   OBB_TEST_GRID='1' if grid else '0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'mp_gpu_worker.py'), str(outdir)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=900)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, 'rank %d failed:\n%s' % (r, outs[r][-3000:])
    return [np.load(os.path.join(str(outdir), 'rank%d.npz' % r)) for r in range(world)]


def _owner(rows, world):
    return rows[:, 7].astype(np.int64) % world


@pytest.mark.parametrize('world,grid', [(2, False), (3, False), (2, True)])
def test_hip_tiled_merge_on_several_ranks_equals_one_rank_and_the_oracle(world, grid, tmp_path):
    """grid: every NMS call of the ranks on the uniform-grid route, plus a scene in which one rank's per-patch NMS gives
    that route up and all ranks repeat the exchange (K word OBB_TILED_REPEAT)."""
    got = run_ranks(world, tmp_path, grid)
    dev = torch.device('cuda', 0)

    def check(tag, ref_kept, ref_keep, ref_rows, ora_kept):
        assert np.array_equal(ref_kept, ora_kept), tag       # the single-process HIP run is the oracle's result
        assert 0 < len(ref_kept) < len(ref_rows), tag
        for r in range(world):
            g = got[r]
            for layout, gather in cases.LAYOUTS:
                t = '%s_%s_%d' % (tag, layout, gather)
                kept, keep, rows = g[t + '_kept'], g[t + '_keep'], g[t + '_rows']
                if layout != 'routed':
                    assert np.array_equal(rows, ref_rows), (t, r)          # single-process order restored
                    assert np.array_equal(keep, ref_keep), (t, r)          # identical keep indices
                    assert np.array_equal(kept, ref_kept), (t, r)
                    continue
                # this rank received exactly the rows of its images, in single-process order
                assert np.array_equal(rows, ref_rows[_owner(ref_rows, world) == r]), (t, r)
                if gather:
                    assert np.array_equal(kept, ref_kept), (t, r)
                else:                                                      # kept rows stay with the image's owner
                    assert np.array_equal(kept, ref_kept[_owner(ref_kept, world) == r]), (t, r)
                    assert np.array_equal(kept, rows[keep]), (t, r)

    # (a) rows with / without score ties; patch 4 is empty; with 3 ranks images 0, 1 leave rank 2 without an image
    for ties in (False, True):
        single = torch.from_numpy(np.concatenate(cases.make_patches(cases.NUM_PATCHES, 7, ties), 0))
        hk, hkeep = dota_merge.merge_detections(single.to(dev), 0.3, cases.NUM_CLASSES, cases.CLASS_THRS)
        ok, _ = dota_merge.merge_detections(single, 0.3, cases.NUM_CLASSES, cases.CLASS_THRS, nms_fn=cases.oracle_nms)
        check('rows_%d' % ties, hk.cpu().numpy(), hkeep.cpu().numpy(), single.numpy(), ok.numpy())
    # (b) the whole pass
    for name, kw in cases.SCENES.items():
        args = cases.tiled_inputs(0, 1, **kw)
        hk, hkeep, hrows = dota_merge.tiled_inference_merge(*[a.to(dev) for a in args], cases.NUM_CLASSES, 0.5, 0.3,
                                                            cases.CLASS_THRS)
        ok, _, orows = dota_merge.tiled_inference_merge(*args, cases.NUM_CLASSES, 0.5, 0.3, cases.CLASS_THRS,
                                                        nms_fn=cases.oracle_nms)
        assert np.array_equal(hrows.cpu().numpy(), orows.numpy())
        check('tiled_' + name, hk.cpu().numpy(), hkeep.cpu().numpy(), hrows.cpu().numpy(), ok.numpy())
    if grid:
        args = cases.crowd_inputs(0, 1)
        hk, hkeep, hrows = dota_merge.tiled_inference_merge(*[a.to(dev) for a in args], cases.NUM_CLASSES, 0.5, 0.3,
                                                            cases.CLASS_THRS)
        ok, _, orows = dota_merge.tiled_inference_merge(*args, cases.NUM_CLASSES, 0.5, 0.3, cases.CLASS_THRS,
                                                        nms_fn=cases.oracle_nms)
        assert np.array_equal(hrows.cpu().numpy(), orows.numpy())
        check('tiled_crowd', hk.cpu().numpy(), hkeep.cpu().numpy(), hrows.cpu().numpy(), ok.numpy())
        assert int(got[0]['phase_repeats']) >= 1          # rank 0 owns patch 0: its per-patch phase was repeated
    # (c) the row blocks of rbbox_overlaps tile the matrix
    import oracle
    a, b = cases.iou_inputs()
    whole = np.concatenate([got[r]['iou_block'] for r in range(world)], 0)
    assert np.array_equal(whole, oracle.overlaps(a.numpy(), b.numpy()))


def test_bench_two_ranks_on_one_gpu_over_gloo():
    """`bench.py --gpus 2 --transport gloo`: the N > 1 bench path (both legs, three cfg5 layouts, the keep-set
    cross-checks) with two real ranks on the one GPU.  Debug transport: the numbers mean nothing, the checks do."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--transport', 'gloo', '--no-cpu', '--no-extras', '--no-cfg5-large'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert j['n_gpus'] == 2 and j['transport'].startswith('gloo') and j['scaling'] == 'weak'
    c = j['config']
    assert c['cfg5_keep_ok'] is True
    for layout in ('replicated', 'sharded', 'routed'):
        d = j['cfg5'][layout]
        assert d['keep_equals_single_rank_run'] is True and d['same_keep_on_all_ranks'] is True, layout


def test_bench_preflight_a_layout_that_raises_and_a_rank_that_hangs():
    """the first real multi-GPU run must fail informatively.  (a) a layout that raises on every rank is
    recorded on the line and the other layouts still run; (b) a rank that never reaches a collective: the other rank's
    watchdog prints the partial line (replica leg, the layouts done so far, what hung) and the job ends non-zero."""
    import json
    base = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--transport', 'gloo',
           '--no-cpu', '--no-extras', '--no-cfg5-large']
    r = subprocess.run(cmd, env=dict(base, OBB_BENCH_TEST_RAISE='sharded'), capture_output=True, text=True, timeout=900)
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    c = j['config']
    assert r.returncode != 0 and 'test hook' in c['cfg5_sharded_error']
    assert j['rccl_ranks'] == 2 and c['cfg5_layout'] == 'replicated' and c['cfg5_speedup_vs_single_rank'] > 0
    assert c['cfg5_routed_ms_per_step'] > 0 and j['cfg5']['routed']['keep_equals_single_rank_run'] is True
    r = subprocess.run(cmd + ['--collective-timeout', '25'], env=dict(base, OBB_BENCH_TEST_HANG='routed'), capture_output=True,
                       text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert r.returncode != 0 and lines, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    j = json.loads(lines[-1])
    assert 'did not finish' in j['config']['cfg5_error'] and j['value'] > 0 and j['rccl_ranks'] == 2
    assert 'replicated' in j['config']['cfg5_progress']

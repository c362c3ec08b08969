#!/usr/bin/env python3
"""bench.py -- headline benchmark of the OBB hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by torch.distributed.run (one rank per GPU, RCCL); when it is NOT (plain
  `python bench.py --gpus 4`) this script starts that launcher itself as a child process,
  before anything touches the GPU.  Rank 0 prints ONE JSON line.

EVERY line (N = 1, 2, 4, 8) carries the same two legs, so the driver's 1 -> 8 curve compares like with like:

  replica  (the timed K steps; `value`)  BASELINE.json configs[1] on every rank: rotated NMS (`rnms`) on
           100 000 synthetic theta-OBBs, IoU thr 0.5, boxes resident in HBM.  One step = one full rnms call
           incl. the count read-back.  No data-path collective (independent units): "scaling": "weak",
           `value` = N x 100 000 x K / max-over-ranks time.  Beside it every rank computes its ROW BLOCK of a
           16 384 x 16 384 `rbbox_overlaps` (rows dealt in blocks over the ranks, SURVEY 8e(2), no
           communication): `config.iou_pairs_per_s` = all pairs / max-over-ranks time (strong scaling).
  cfg5     BASELINE.json configs[4], the one place where the path has a real exchange: tiled inference.  The
           patches of a fixed set of large images are dealt rank-strided (DistributedSampler,
           distributed_sampler.py:18-25); one step = per-patch rotated NMS of the rank's detections ->
           image coordinates -> RCCL exchange -> merged per-(image, class) rotated NMS with the per-class
           thresholds of dota.py:324.  Total work is fixed (strong scaling): `config.cfg5_boxes_per_s`,
           `config.cfg5_speedup_vs_single_rank`.  Layouts: `replicated` (north star: all-gather, every rank
           merges everything; the default), `sharded` and `routed` (all-to-all keyed by image: every phase
           divides by N) are timed too at N > 1; each keep set is checked against a single-rank run.

The driver keeps the SCALARS of `roofline` and `config` (nested objects and long strings do not survive into
BENCH_r*.json), so everything BASELINE's metric names is a scalar there: IoU pairs/s, HBM fractions, traffic
ratios.  Extra objects on the N = 1 line (for profiles/, humans):
  roofline_kernels the HBM-bound kernels (rbbox_overlaps sparse / dense, RRoIAlign forward /
                   backward) with SURVEY 8(d)'s algorithmic bytes, plus the fill / copy rates
                   this GPU reaches.
  cpu_baseline     the C oracle on this host: 1 thread (the reference's loop is serial,
                   nms_cpu.cpp:35-58) and all cores (OpenMP legs of the oracle, best of 3).
  also             side metrics outside the timed region (latencies, config 4, ...).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                 # MI355X HBM3E spec (MI355X_MICROARCH.md)
# VALU issue: 256 CUs x 4 SIMD-32; a wave64 VALU op occupies its SIMD for 2 cycles
# (MI355X_MICROARCH.md, cycle constants: v_fma_f32 wave64 = 2 cyc throughput)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0  # G wave-instructions / s at 2.4 GHz
COUNTER_FILES = [os.path.join(ROOT, 'profiles', 'r06_counters.json'), os.path.join(ROOT, 'profiles', 'r05_counters.json'), os.path.join(ROOT, 'profiles', 'r04_counters.json'),
                 os.path.join(ROOT, 'profiles', 'r03_counters.json')]
IOU_M = 16384                         # rbbox_overlaps leg: 16k x 16k (SURVEY 8d scale-up of configs[0])
IOU_CANVAS = 1024                     # ... at configs[0]'s density (boxes on a 1024 canvas; 1.2 % of the pairs clipped)
IOU_SPARSE_CANVAS = 8192              # the write-bound extreme beside it (0.02 % clipped): `iou_sparse_*`


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--boxes', type=int, default=100000)
    ap.add_argument('--canvas', type=float, default=20000.0,
                    help='20000 = DOTA-sparse scene (headline); 4096 = dense stress variant')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--no-extras', action='store_true', help='skip the side metrics')
    ap.add_argument('--merge-layout', default='replicated', choices=['replicated', 'sharded', 'routed'],
                    help='layout of the cfg5 leg that config.cfg5_* reports (the others are timed beside it at N > 1)')
    ap.add_argument('--no-cfg5', action='store_true', help='skip the cfg5 leg')
    ap.add_argument('--no-cfg5-large', action='store_true', help='skip the 1e7-detection size of the cfg5 leg (128 images)')
    ap.add_argument('--transport', default='rccl', choices=['rccl', 'gloo'],
                    help='rccl (default; backend "nccl" IS RCCL on ROCm, one rank per GPU) | gloo: DEBUG ONLY -- the ranks may '
                         'share GPUs (rank r uses cuda:(r mod #GPUs)) and every collective is staged through host memory '
                         '(aidet_amd/_collectives.py), so that the N > 1 code path can be smoke-run on a one-GPU box')
    ap.add_argument('--collective-timeout', type=float, default=180.0,
                    help='seconds a collective (and the whole cfg5 leg per layout) may take at N > 1 before the process '
                         'prints what it has and exits non-zero')
    ap.add_argument('--profile-target', default=None,
                    choices=['nms', 'iou_sparse', 'iou_dense', 'iou_allclip', 'rroi_fwd', 'rroi_bwd', 'rroi_bwd_scatter', 'roi_fwd', 'cfg5',
                             'rroi_fwd_nhwc', 'rroi_bwd_nhwc', 'roi_fwd_nhwc'],
                    help='run ONLY this op (for rocprofv3 passes: one workload per kernel name)')
    return ap.parse_args()


def main():
    args = parse_args()
    env_world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and 'RANK' not in os.environ:
        # not under a launcher: become one.  Nothing has touched the GPU yet (torch is not even
        # imported), so starting the ranks as CHILD processes is safe; exit with their code.
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))
    if env_world != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%d: the launcher and the flag disagree\n'
                         % (args.gpus, env_world))
        sys.exit(2)
    if args.profile_target:
        profile_target(args)
    else:
        bench_all(args)


# ==============================================================================================
# the two legs of every line
# ==============================================================================================
def bench_all(args):
    import numpy as np
    import torch

    import workloads
    from aidet_amd import _lib, rbbox_overlaps
    from aidet_amd.nms_wrapper import _run_nms

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    if world > 1 or os.environ.get('OBB_BENCH_FORCE_DIST') == '1':
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if args.transport == 'gloo':
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if dist is not None:
        import datetime
        # every collective under a timeout: a rank that waits for one that will never come ends non-zero (a FRESH launch
        # restarts the job -- never a re-exec of a process that has touched the GPU) instead of hanging the node
        dist.init_process_group('nccl' if args.transport == 'rccl' else 'gloo', rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=args.collective_timeout))
        assert dist.get_world_size() == world
    from aidet_amd import _collectives as coll

    def barrier():
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x):
        if dist is None:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=dev)
        coll.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    # ---- leg 1: replica -- configs[1] on every rank ------------------------------------------
    n = args.boxes
    boxes = workloads.boxes(n, args.canvas, 3)      # (the same scenes on every rank: independent replicas)
    d = torch.from_numpy(boxes).to(dev)
    # The timed steps walk a POOL of scenes of the configuration (seeds 3, 13, 23, 33; step i takes scene i mod 4): since
    # round 6 a call on the wrapper's workspace finds the frame its PREDECESSOR left (OBB_NMS_RETAIN, obbops.h) -- with one
    # scene for every step that would be the frame of the very same boxes.  Scene 0 is the one the keep list, the CPU
    # baseline and the counters refer to.
    pool = [d] + [torch.from_numpy(workloads.boxes(n, args.canvas, 3 + 10 * k)).to(dev) for k in (1, 2, 3)]
    L = _lib.lib()
    state = {'i': 0}

    def step():
        r = _run_nms(pool[state['i'] & 3], _lib.KIND_RBOX, 0.5)
        state['i'] += 1
        return r

    keep, info = step()          # (untimed; also sizes the workspace and seeds its retained state)
    for _ in range(max(args.warmup, 1) * 4 - 1):
        step()
    state['i'] = 0
    keep, info = step()          # scene 0 again, now as a call of the steady state
    kept = int(keep.numel())
    route = int(L.obb_nms_last_route())      # 4: the grid route (sparse scenes); 2: the sorted route
    retained = int(L.obb_nms_last_retained())   # 2: three launches on the state the previous call left in the workspace

    # the same steps with the flag ignored (every call with its own frame launch: four launches) -- outside the timed region
    stateless_ms = None
    if route == 4 and retained == 2:
        _lib.set_tuning(_lib.TUNE_NMS_RETAIN, 0)
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        t0s = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        stateless_ms = (time.perf_counter() - t0s) * 1e3 / args.steps
        _lib.set_tuning(_lib.TUNE_NMS_RETAIN, -1)
        for _ in range(4):       # (the first of these seeds the workspace again)
            step()
    state['i'] = 0

    # per-phase table from a separate, untimed pass (an event between two kernels
    # costs a ~5 us bubble; the timed region only brackets the dominant kernel)
    L.obb_nms_profile_enable(2)
    phase_tab = np.zeros(7, dtype=np.float64)
    pbuf = (ctypes.c_float * 7)()
    for _ in range(5):
        step()
        L.obb_nms_profile_last(pbuf)
        phase_tab += np.frombuffer(pbuf, dtype=np.float32)
    phase_tab /= 5
    L.obb_nms_profile_enable(0)

    # ---- timed region: exactly K steps, barrier + sync on both sides, max over ranks -----------
    # (the dominant kernel is bracketed by two HIP events on its own stream inside the timed steps -- on every 10th step,
    # steps 5, 15, ...: an event between two kernels of a four-launch chain is a ~5 us bubble each and reading the pair
    # back a synchronisation, ~25 us of a 58 us call; `roofline.samples_in_timed_region` says how many there were)
    phases = np.zeros(7, dtype=np.float64)
    buf = (ctypes.c_float * 7)()
    sampled = 0
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        probe = i % 10 == 5 or (args.steps <= 5 and i == 0)
        if probe:
            L.obb_nms_profile_enable(1)
        keep, info = step()
        if probe:
            L.obb_nms_profile_last(buf)
            phases += np.frombuffer(buf, dtype=np.float32)
            sampled += 1
            L.obb_nms_profile_enable(0)
    torch.cuda.synchronize()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    phases /= max(sampled, 1)
    retained_last = int(L.obb_nms_last_retained())     # what the LAST timed call did (2: three launches)
    state['i'] = 0
    keep, info = step()          # (scene 0: the keep list the CPU baseline is compared with)

    ms_per_step = elapsed * 1e3 / args.steps
    value = world * n * args.steps / elapsed
    scan_ms = float(phases[3])  # nms_scan_kernel, HIP events inside the timed region

    # the IoU half of BASELINE's metric: this rank's row block of 16k x 16k (no communication), at configs[0]'s
    # density (canvas 1024); the sparse extreme (canvas 8192: nearly every pair is an exact zero) beside it
    shard = (rank, world)
    iou_ms, iou_pairs = {}, {}
    for tag, canvas in (('cfg', IOU_CANVAS), ('sparse', IOU_SPARSE_CANVAS)):
        a, b = make_iou(dev, IOU_M, canvas)
        rbbox_overlaps(a, b, row_shard=shard)
        barrier()
        iou_ms[tag] = max_over_ranks(_time_gpu(lambda: rbbox_overlaps(a, b, row_shard=shard), 10))
        iou_pairs[tag] = IOU_M * IOU_M / (iou_ms[tag] * 1e-3)
        del a, b

    # ---- leg 2: cfg5 (strong scaling) -------------------------------------------------------------
    cfg5, cfg5_error = None, None
    partial = {'metric': 'rotated-NMS boxes/sec', 'value': round(value, 1), 'unit': 'boxes/s', 'n_gpus': world, 'steps': args.steps,
               'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak',
               'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': 'rnms: %d theta-OBBs per GPU, canvas %g (BASELINE configs[1])' % (n, args.canvas),
                          'iou_pairs_per_s': round(iou_pairs['cfg'], 1), 'cfg5_progress': {}},
               'rccl_ranks': world if dist is not None else None}
    if not args.no_cfg5:
        # the one leg with collectives in its data path.  Pre-flight for a node nobody has run this on: a
        # watchdog per layout -- when a layout neither finishes nor fails within the limit (a rank stuck in a collective
        # whose peer raised), rank 0 prints the line it has (replica leg + the layouts done so far + which one hung) and
        # every rank exits non-zero; a layout that RAISES on every rank (e.g. uneven all_to_all splits refused by the
        # transport) is recorded and the next layout still runs.
        try:
            cfg5 = cfg5_leg(args, dev, rank, world, dist, partial)
        except Exception as e:      # (the replica leg above is measured: its line is still printed, with the failure on it)
            import traceback
            cfg5_error = '%s: %s' % (type(e).__name__, str(e)[:300])
            sys.stderr.write('bench.py: the cfg5 leg failed on rank %d\n%s\n' % (rank, traceback.format_exc()))

    rk = None
    if rank == 0 and world == 1 and not args.no_extras:
        rk = roofline_kernels(dev)
    roof = nms_roofline(n, args.canvas, kept, scan_ms, ms_per_step, rk, route, retained_last)
    iou_bytes = 20 * (IOU_M // world + IOU_M) * world + 4 * IOU_M * IOU_M
    roof['samples_in_timed_region'] = int(sampled)
    roof['iou_rowblock_pairs_per_s'] = round(iou_pairs['cfg'], 1)
    roof['iou_rowblock_hbm_frac'] = round(iou_bytes / (iou_ms['cfg'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    roof['iou_sparse_rowblock_pairs_per_s'] = round(iou_pairs['sparse'], 1)
    roof['iou_sparse_rowblock_hbm_frac'] = round(iou_bytes / (iou_ms['sparse'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    out = {
        'metric': 'rotated-NMS boxes/sec', 'value': round(value, 1), 'unit': 'boxes/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'rnms: %d theta-OBBs per GPU, canvas %g, w,h~logU(8,128), '
                               'theta~U(-pi/2,0), IoU thr 0.5 (BASELINE configs[1])' % (n, args.canvas),
                   'boxes_per_gpu': n, 'kept': kept, 'nms_route': {0: 'fused', 1: 'direct', 2: 'sorted', 4: 'grid', 5: 'grid->sorted'}.get(route, str(route)),
                   'nms_launches': (3 if retained_last == 2 else 4) if route == 4 else None,
                   'nms_workspace': ('retained between calls (OBB_NMS_RETAIN): the previous call cleared the cell counters and left the frame of ITS '
                                     'boxes; every timed step is a different scene of a pool of 4') if retained_last == 2 else 'stateless',
                   'nms_stateless_ms_per_step': None if stateless_ms is None else round(stateless_ms, 4),
                   'overlap_edges': info[0],
                   'resolve_rounds': info[1], 'candidate_pairs': info[2],
                   'parallelism': 'single GPU' if world == 1 else 'replica x%d (no data-path collective)' % world,
                   'iou_workload': 'rbbox_overlaps %d x %d, canvas %d (configs[0] density), rows in %d block(s)' % (IOU_M, IOU_M, IOU_CANVAS, world),
                   'iou_pairs_per_s': round(iou_pairs['cfg'], 1), 'iou_ms': round(iou_ms['cfg'], 4),
                   'iou_sparse_pairs_per_s': round(iou_pairs['sparse'], 1), 'iou_sparse_ms': round(iou_ms['sparse'], 4),
                   'iou_sparse_workload': 'the same on canvas %d (0.02 %% of the pairs clipped: write bound)' % IOU_SPARSE_CANVAS},
        'roofline': roof,
        'phase_ms': {k: round(float(v), 4) for k, v in zip(
            ['prep_quant' if route != 4 else ('insert' if retained == 2 else 'frame_insert'), 'cell_sort', 'permute_cull', 'scan' if route != 4 else 'probe_clip', 'clip',
             'resolve' if route != 4 else 'resolve_compact', 'compact'], phase_tab)},
    }
    if dist is not None:
        out['rccl_ranks'] = dist.get_world_size()
        out['transport'] = 'rccl' if args.transport == 'rccl' else \
            'gloo (debug: ranks share GPUs, collectives staged through host; timings are not product numbers)'
    ok = True
    if cfg5 is not None:
        out['config'].update(cfg5['flat'])
        out['cfg5'] = cfg5['detail']
        ok = cfg5['ok']
    if cfg5_error is not None:
        out['config']['cfg5_error'] = cfg5_error
    if rank == 0 and world == 1:
        if rk is not None:
            out['roofline_kernels'] = rk
            out['also'] = extras(dev, args)
            # the headline workload without the Python wrapper: one obb_nms_f32 call through the bare C ABI
            out['config']['nms_c_abi_us_per_call'] = out['also']['rnms_100k_c_abi_retained']['us_per_call']
            out['config']['nms_c_abi_stateless_us_per_call'] = out['also']['rnms_100k_c_abi']['us_per_call']
            if 'rnms_100k_4_threads_4_streams' in out['also']:      # four host threads, a stream and a workspace each: a throughput, not `value`
                out['config']['nms_4_threads_boxes_per_s'] = out['also']['rnms_100k_4_threads_4_streams']['boxes_per_s']
        if not args.no_cpu:
            out['cpu_baseline'] = cpu_baseline(boxes, keep)
    if rank == 0:
        if not ok:
            failed = sorted(k for k in out['config'] if k.startswith('cfg5_') and k.endswith('_error'))
            out['error'] = ('cfg5: ' + ', '.join('%s = %s' % (k, out['config'][k]) for k in failed)) if failed else \
                'distributed merge differs from the single-rank result'
        print(json.dumps(out))
    if dist is not None and cfg5_error is None:      # (after a failed leg the ranks may not be in step any more: no further collective)
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(1)


class _Watchdog(object):
    """`with _Watchdog(seconds, what, partial, rank)`: when the body takes longer, rank 0 prints the partial JSON line with
    `cfg5_error` naming what hung, and the process exits with code 3 (os._exit: the main thread may sit in a collective)."""

    def __init__(self, seconds, what, partial, rank):
        import threading
        self.t = threading.Timer(seconds, self.fire)
        self.t.daemon = True
        self.what, self.partial, self.rank, self.seconds = what, partial, rank, seconds

    def fire(self):
        if self.rank == 0 and self.partial is not None:
            line = dict(self.partial)
            line['config'] = dict(line['config'], cfg5_error='%s did not finish within %.0f s (a rank stuck in a collective?)' % (self.what, self.seconds))
            line['error'] = 'timeout in ' + self.what
            sys.stdout.write(json.dumps(line) + '\n')
            sys.stdout.flush()
        sys.stderr.write('bench.py: rank %d: %s timed out after %.0f s; exiting\n' % (self.rank, self.what, self.seconds))
        sys.stderr.flush()
        os._exit(3)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


def cfg5_leg(args, dev, rank, world, dist, partial=None):
    """BASELINE configs[4] on `world` ranks: every layout timed (the chosen one with the full step count), each
    keep set compared with a single-rank run of all patches.  -> {'flat': scalars for `config`, 'detail', 'ok'}"""
    import hashlib

    import numpy as np
    import torch
    from aidet_amd import _collectives as coll
    from aidet_amd import dota_merge
    steps = max(5, min(args.steps, 20))
    layouts = [args.merge_layout] + ([l for l in ('replicated', 'sharded', 'routed') if l != args.merge_layout] if world > 1 else [])
    ref_rows, single = None, None
    if rank == 0:   # the single-rank run of ALL patches (untimed reference for the keep sets; no collectives)
        s1, _e1, ref_rows = tiled_on_this_rank(dev, 0, 1, steps=steps if world > 1 else 2, warmup=1, layout='replicated',
                                               group=dota_merge.LOCAL)
        single = s1
    detail, ok_all, flat = {}, True, {}
    limit = args.collective_timeout if world > 1 else 3600.0
    for li, layout in enumerate(layouts):
        st = steps if li == 0 else max(3, steps // 2)
        barrier = dist.barrier if dist is not None else None
        group = None if dist is not None else dota_merge.LOCAL
        try:
            with _Watchdog(limit, 'cfg5 layout %r' % layout, partial if world > 1 else None, rank):
                if world > 1 and rank == world - 1 and os.environ.get('OBB_BENCH_TEST_HANG') == layout:
                    time.sleep(1e6)      # (test hook of tests/test_multiproc_gpu.py: a rank that never reaches the collective)
                if world > 1 and os.environ.get('OBB_BENCH_TEST_RAISE') == layout:
                    raise RuntimeError('test hook: layout %s refused on every rank' % layout)
                summary, elapsed, kept_rows = tiled_on_this_rank(dev, rank, world, st, args.warmup, layout, group=group, barrier=barrier)
        except Exception as e:
            # raised on this rank.  If every rank raised (the transport refuses the call) the next layout is in step again;
            # if only this one did, the others wait in a collective and their watchdogs end the job -- either way the line
            # carries the layout's failure
            import traceback
            sys.stderr.write('bench.py: cfg5 layout %r failed on rank %d\n%s\n' % (layout, rank, traceback.format_exc()))
            flat['cfg5_%s_error' % layout] = '%s: %s' % (type(e).__name__, str(e)[:200])
            if li == 0:
                flat['cfg5_layout'] = layout
            ok_all = False
            if partial is not None:
                partial['config']['cfg5_progress'][layout] = 'failed: ' + type(e).__name__
            continue
        same = True
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            coll.all_reduce(tt, op=dist.ReduceOp.MAX)          # the slowest rank's time is the job's
            elapsed = float(tt.item())
            # every rank holds the same kept rows?  (hash all-reduced as min / max)
            digest = int(hashlib.sha256(kept_rows.cpu().numpy().tobytes()).hexdigest()[:15], 16)
            hh = torch.tensor([digest, -digest], dtype=torch.int64, device=dev)
            coll.all_reduce(hh, op=dist.ReduceOp.MAX)
            same = bool(hh[0].item() == -hh[1].item())
        ok_single = None
        if rank == 0:
            ok_single = bool(np.array_equal(ref_rows.cpu().numpy(), kept_rows.cpu().numpy()))
        ms = elapsed * 1e3 / st
        total = summary['detections']
        rec = dict(summary, ms_per_step=round(ms, 4), boxes_per_s=round(total * st / elapsed, 1), steps=st,
                   same_keep_on_all_ranks=same, keep_equals_single_rank_run=ok_single)
        detail[layout] = rec
        # the verdict travels to every rank BEFORE anybody leaves (a rank that raised on its own would leave the
        # others waiting in the next collective)
        good = 1 if (rank != 0 or (same and ok_single)) else 0
        if dist is not None:
            g = torch.tensor([good], dtype=torch.int64, device=dev)
            coll.all_reduce(g, op=dist.ReduceOp.MIN)
            good = int(g.item())
        ok_all = ok_all and bool(good)
        if li == 0:
            flat = {'cfg5_workload': 'tiled merge (configs[4]): %d images 8192^2, %d patches, %d detections'
                                     % (summary['images'], summary['patches'], total),
                    'cfg5_layout': layout, 'cfg5_ms_per_step': round(ms, 4),
                    'cfg5_boxes_per_s': round(total * st / elapsed, 1),
                    'cfg5_kept_after_merge': summary['kept_after_merge'],
                    'cfg5_local_nms_us': summary['local_nms_us'], 'cfg5_exchange_us': summary['gather_us'],
                    'cfg5_merge_us': summary['merge_us'], 'cfg5_keep_ok': bool(good),
                    'cfg5_phases_repeated_without_the_grid': summary['phases_repeated_without_the_grid']}
            if single is not None:
                flat['cfg5_single_rank_ms_per_step'] = single['ms_per_step']
                flat['cfg5_speedup_vs_single_rank'] = round(single['ms_per_step'] / ms, 3)
        else:
            flat['cfg5_%s_ms_per_step' % layout] = round(ms, 4)
            if single is not None:
                flat['cfg5_%s_speedup_vs_single_rank' % layout] = round(single['ms_per_step'] / ms, 3)
        if partial is not None:
            partial['config']['cfg5_progress'][layout] = {'ms_per_step': round(ms, 4), 'keep_ok': bool(good)}
            partial['config'].update({k: v for k, v in flat.items() if not isinstance(v, (dict, list))})
    # ---- the size where strong scaling has room: 128 images, 1.0e7 detections (routed at N > 1: every phase divides)
    if not args.no_cfg5_large:
        layout = args.merge_layout if world == 1 else 'routed'
        big_ref, big_single = None, None
        st = 3
        if rank == 0 and world > 1:
            big_single, _e, big_ref = tiled_on_this_rank(dev, 0, 1, steps=st, warmup=1, layout='replicated', group=dota_merge.LOCAL, images=128)
        barrier = dist.barrier if dist is not None else None
        group = None if dist is not None else dota_merge.LOCAL
        with _Watchdog(limit, 'cfg5_large (%s)' % layout, partial if world > 1 else None, rank):
            summary, elapsed, kept_rows = tiled_on_this_rank(dev, rank, world, st, 1, layout, group=group, barrier=barrier, images=128)
        good = 1
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            coll.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
            if rank == 0:
                good = int(bool(np.array_equal(big_ref.cpu().numpy(), kept_rows.cpu().numpy())))
            g = torch.tensor([good], dtype=torch.int64, device=dev)
            coll.all_reduce(g, op=dist.ReduceOp.MIN)
            good = int(g.item())
        ms = elapsed * 1e3 / st
        flat.update({'cfg5_large_workload': 'tiled merge, %d images 8192^2, %d patches, %d detections (%s)'
                                            % (summary['images'], summary['patches'], summary['detections'], layout),
                     'cfg5_large_ms_per_step': round(ms, 4), 'cfg5_large_boxes_per_s': round(summary['detections'] * st / elapsed, 1),
                     'cfg5_large_keep_ok': bool(good),
                     'cfg5_large_phases_repeated_without_the_grid': summary['phases_repeated_without_the_grid']})
        if big_single is not None:
            flat['cfg5_large_single_rank_ms_per_step'] = big_single['ms_per_step']
            flat['cfg5_large_speedup_vs_single_rank'] = round(big_single['ms_per_step'] / ms, 3)
        ok_all = ok_all and bool(good)
        detail['large_' + layout] = dict(summary, ms_per_step=round(ms, 4))
        _scenes.clear()
    return {'flat': flat, 'detail': detail, 'ok': ok_all}


def load_counters():
    """The newest committed counter file (profiles/r03_counters.json, else round 2's)."""
    for path in COUNTER_FILES:
        try:
            with open(path) as f:
                c = json.load(f)
                c['_file'] = os.path.relpath(path, ROOT)
                return c
        except (OSError, ValueError):
            continue
    return None


def kernel_counters(name, workload):
    """Per-launch PMC means of `name` from the committed rocprofv3 passes
    (profiles/r0x_counters.json, built by tools/make_profiles.py; PMC collection needs
    rocprofv3 around the process, so it cannot be taken live).  Only returned for the workload
    the counters were measured on."""
    c = load_counters()
    if not c:
        return None
    for k in c.get('kernels', {}).values():
        if k.get('kernel') == name and k.get('workload') == workload:
            return k
    return None


def hbm_bytes(k):
    """gfx950: FETCH_SIZE reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM)."""
    if not k or 'FETCH_SIZE_KB' not in k or 'WRITE_SIZE_KB' not in k:
        return None
    return int(2 * k['FETCH_SIZE_KB'] * 1024 + k['WRITE_SIZE_KB'] * 1024)


def nms_roofline(n, canvas, kept, scan_ms, step_ms, rk=None, route=2, retained=0):
    """nms_scan_kernel is bound by VALU ISSUE (3 full-rate ops per tested pair plus the
    candidate path), not by HBM: its counter traffic is a few MB per launch.  The rest of the
    call is launch/latency bound (dependent launches of 4-20 us).  Everything is a SCALAR (the
    driver's record keeps scalars of this object only); `rk` = roofline_kernels(), whose HBM-bound
    kernels are flattened in beside it."""
    wl = 'nms_%d_%g' % (n, canvas)
    kname = 'nms_gprobe_kernel' if route == 4 else 'nms_scan_kernel'      # the dominant kernel of the route the call took
    k = kernel_counters(kname, wl)
    c = load_counters()
    words = (n + 63) // 64
    r = {
        'kernel': kname, 'bound': 'valu_issue', 'unit': 'G wave-instr/s',
        'achieved': None, 'peak': round(VALU_PEAK_GINST, 1), 'frac': None, 'traffic': None,
        'kernel_ms': round(scan_ms, 4),
        'compulsory_bytes_whole_call': 24 * n + 8 * kept,
        # SURVEY 8(d)'s model of the REFERENCE algorithm (n x n/64 bit mask written, kept rows
        # read back): what this pipeline avoids moving -- not an achieved rate
        'reference_algorithm_bytes': 24 * n + 4 * n * words + 4 * kept * words + 8 * kept,
        'note': ('grid route on a retained workspace (insert, probe + clip, resolve + compaction; the frame launch only when the workspace '
                 'is seeded): a lane per pair of entries of a chunk of cells, bounds on the entries, the clip of what is left; a '
                 'latency-bound call of three launches; HBM fracs beside it') if route == 4 else
                'VALU-issue bound kernel in a latency-bound call; HBM fracs beside it',
        'counters': c['_file'] if (c and k) else None,
    }
    if k and scan_ms > 0:
        if 'SQ_INSTS_VALU' in k:
            ach = k['SQ_INSTS_VALU'] / (scan_ms * 1e-3) / 1e9
            r['achieved'] = round(ach, 1)
            r['frac'] = round(ach / VALU_PEAK_GINST, 4)
            r['valu_wave_insts_per_launch'] = int(k['SQ_INSTS_VALU'])
        tb = hbm_bytes(k)
        if tb:
            r['traffic'] = tb
            r['hbm_frac'] = round(tb / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        tot = 0
        for name, kk in c['kernels'].items():
            if retained == 2 and kk.get('kernel') == 'nms_frame_kernel':
                continue      # (a three-launch call has no frame launch: the counters' one instance is the seeding call's)
            if kk.get('workload') == wl and hbm_bytes(kk):
                tot += hbm_bytes(kk)
        if tot:
            r['whole_call_traffic'] = tot
            r['whole_call_hbm_frac'] = round(tot / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            r['traffic_over_compulsory'] = round(tot / float(24 * n + 8 * kept), 2)
    if rk:
        r['measured_fill_GBps'] = rk['measured_peak']['fill_GBps']
        r['measured_copy_GBps'] = rk['measured_peak']['copy_GBps']
        for row in rk['kernels']:
            t = row['tag']
            r[t + '_ms'] = row['kernel_ms']
            r[t + '_hbm_frac'] = row['frac']
            if row.get('pairs_per_s'):
                r[t + '_pairs_per_s'] = row['pairs_per_s']
            if row.get('traffic'):
                r[t + '_traffic_over_algorithmic'] = round(row['traffic'] / float(row['algorithmic_bytes']), 3)
            if row.get('lds_conflict_share') is not None:
                r[t + '_lds_conflict_share'] = row['lds_conflict_share']
            if row.get('valu_issue_frac') is not None:
                r[t + '_valu_issue_frac'] = row['valu_issue_frac']
    return r


def _time_gpu(fn, iters):
    import torch
    # warm up for >= 50 ms of wall time: after a CPU-only leg (the oracle checks between the side metrics) the
    # first ~40 ms of GPU work run at idle clocks (observed: 1.5 ms instead of 0.27 ms per call over 30 calls)
    t0 = time.perf_counter()
    while True:
        fn()
        torch.cuda.synchronize()
        if time.perf_counter() - t0 > 0.05:
            break
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    torch.cuda.synchronize()
    return start.elapsed_time(stop) / iters


# ---- workloads shared by the side metrics and the profiling passes -------------------------
def make_iou(dev, m, canvas):
    import torch
    import workloads
    a = torch.from_numpy(workloads.boxes(m, canvas, 1)[:, :5]).to(dev)
    b = torch.from_numpy(workloads.boxes(m, canvas, 2)[:, :5]).to(dev)
    return a, b


def make_iou_allclip(dev, m):
    """m + m boxes that ALL overlap each other: centres within 16 px, sides 32-128 px"""
    import numpy as np
    import torch
    out = []
    for seed in (1, 2):
        rng = np.random.default_rng(seed)
        bx = np.stack([rng.uniform(0, 16, m), rng.uniform(0, 16, m), np.exp(rng.uniform(np.log(32.0), np.log(128.0), m)),
                       np.exp(rng.uniform(np.log(32.0), np.log(128.0), m)), rng.uniform(-np.pi / 2, 0.0, m)], 1).astype(np.float32)
        out.append(torch.from_numpy(bx).to(dev))
    return out


def make_rroi(dev, K=512, C=256, HW=256):
    import numpy as np
    import torch
    import workloads
    rng = np.random.default_rng(5)
    feat = torch.from_numpy(rng.standard_normal((1, C, HW, HW)).astype(np.float32)).to(dev)
    r = workloads.boxes(K, 4 * HW, 5)[:, :5]
    rois = torch.from_numpy(np.concatenate([np.zeros((K, 1), np.float32), r], 1)).to(dev)
    return feat, rois


def rroi_footprint_pixels(rois_np, H, W, scale):
    """sum over the RoIs of the clipped AABB pixel count on the map (SURVEY 8(d)'s footprint)."""
    import numpy as np
    cx, cy, w, h, th = (rois_np[:, i].astype(np.float64) for i in range(1, 6))
    cx, cy, w, h = cx * scale, cy * scale, np.maximum(w * scale, 1), np.maximum(h * scale, 1)
    hx = 0.5 * (np.abs(np.cos(th)) * w + np.abs(np.sin(th)) * h)
    hy = 0.5 * (np.abs(np.sin(th)) * w + np.abs(np.cos(th)) * h)
    x0 = np.clip(np.floor(cx - hx), 0, W - 1); x1 = np.clip(np.floor(cx + hx) + 1, 0, W - 1)
    y0 = np.clip(np.floor(cy - hy), 0, H - 1); y1 = np.clip(np.floor(cy + hy) + 1, 0, H - 1)
    return float(((x1 - x0 + 1) * (y1 - y0 + 1)).sum())


def sum_counters(names, workload):
    """HBM bytes, time and LDS conflict share summed over the kernels `names` of one profiled workload."""
    tot, us, conf, act, n = 0, 0.0, 0.0, 0.0, 0
    for nm in names:
        k = kernel_counters(nm, workload)
        if not k:
            continue
        n += 1
        tot += hbm_bytes(k) or 0
        us += k.get('avg_us', 0.0)
        conf += k.get('SQ_LDS_BANK_CONFLICT', 0.0)
        act += k.get('SQ_LDS_IDX_ACTIVE', 0.0)
    if n == 0:
        return None
    return {'traffic': tot or None, 'rocprof_us': round(us, 2), 'lds_conflict_share': round(conf / act, 3) if act > 0 else None}


def roofline_kernels(dev):
    """The HBM-bound kernels of the path with SURVEY 8(d)'s algorithmic bytes, each timed with
    events on the stream it runs on (torch's current stream).  frac = bytes / time / 8 TB/s."""
    import torch
    from aidet_amd import RRoIAlign, rbbox_overlaps
    res = []
    # what a plain streaming kernel reaches on this GPU (1 GiB buffers)
    x = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    fill = _time_gpu(lambda: x.fill_(1.0), 10)
    copy = _time_gpu(lambda: y.copy_(x), 10)
    peaks = {'fill_GBps': round(x.numel() * 4 / (fill * 1e-3) / 1e9, 1),
             'copy_GBps': round(2 * x.numel() * 4 / (copy * 1e-3) / 1e9, 1), 'buffer': '1 GiB fp32'}
    del x, y
    for tag, m, canvas in (('iou_sparse', 16384, 8192), ('iou_dense', 16384, 1024)):
        a, b = make_iou(dev, m, canvas)
        ms = _time_gpu(lambda: rbbox_overlaps(a, b), 10)
        alg = 20 * (2 * m) + 4 * m * m
        k = kernel_counters('overlaps_tile_kernel', tag)
        res.append({'tag': tag, 'kernel': 'overlaps_tile_kernel', 'workload': 'rbbox_overlaps %d x %d, canvas %d' % (m, m, canvas),
                    'bound': 'hbm' if tag == 'iou_sparse' else 'hbm (stream) + valu (clip of 1.2 % of the pairs)',
                    'algorithmic_bytes': alg, 'kernel_ms': round(ms, 4), 'pairs_per_s': round(m * m / (ms * 1e-3), 1),
                    'achieved': round(alg / (ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    'frac_of_measured_fill': round(alg / (ms * 1e-3) / 1e9 / peaks['fill_GBps'], 4),
                    'traffic': hbm_bytes(k)})
    # every pair overlapping: 4096 x 4096 boxes of 32-128 px whose centres lie within 16 px -- all 1.7e7 pairs go through
    # the clip (~1100 VALU each): the VALU-issue-bound extreme of the op
    a, b = make_iou_allclip(dev, 4096)
    ms = _time_gpu(lambda: rbbox_overlaps(a, b), 10)
    nz = float((rbbox_overlaps(a, b) > 0).float().mean().item())
    k = kernel_counters('overlaps_tile_kernel', 'iou_allclip')
    alg = 20 * (2 * 4096) + 4 * 4096 * 4096
    row = {'tag': 'iou_allclip', 'kernel': 'overlaps_tile_kernel', 'workload': 'rbbox_overlaps 4096 x 4096, all pairs overlapping (clip bound)',
           'bound': 'valu_issue', 'algorithmic_bytes': alg, 'kernel_ms': round(ms, 4), 'pairs_per_s': round(4096 * 4096 / (ms * 1e-3), 1),
           'achieved': round(alg / (ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
           'frac': round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'traffic': hbm_bytes(k), 'nonzero_share': round(nz, 4)}
    if k and 'SQ_INSTS_VALU' in k:
        row['valu_issue_frac'] = round(k['SQ_INSTS_VALU'] / (ms * 1e-3) / 1e9 / VALU_PEAK_GINST, 4)
    res.append(row)
    del a, b
    feat, rois = make_rroi(dev)
    layer = RRoIAlign(7, 0.25, sample_num=2)
    rn = rois.cpu().numpy()
    foot = rroi_footprint_pixels(rn, 256, 256, 0.25)
    K, C, H, W = 512, 256, 256, 256
    alg = int(4 * K * C * 49 + 4 * C * min(foot, H * W) + 24 * K)
    # (timed through the C ABI, like the backward below: the Python layer's per-call overhead is as long as this kernel)
    from aidet_amd import _lib
    L = _lib.lib()
    st = _lib.current_stream_handle(dev)
    out_f = torch.empty((512, 256, 7, 7), device=dev)

    def fwd():
        _lib.check(L.obb_rroi_align_forward_f32(feat.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), 512, 7, 7, 0.25, 2, 1.0,
                                                out_f.data_ptr(), st))
    ms = _time_gpu(fwd, 50)
    assert torch.equal(out_f, layer(feat, rois))
    pc = sum_counters(['rroi_forward_kernel', 'rroi_order_kernel'], 'rroi_fwd') or {}
    res.append({'tag': 'rroi_fwd', 'kernel': 'rroi_forward_kernel', 'workload': 'RRoIAlign 512 RoIs x 256 ch x 7x7, 256^2 map, 2x2 samples (configs[2])',
                'bound': 'hbm', 'algorithmic_bytes': alg, 'kernel_ms': round(ms, 4),
                'rois_per_s': round(512 / (ms * 1e-3), 1),
                'achieved': round(alg / (ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'traffic': pc.get('traffic'),
                'lds_conflict_share': pc.get('lds_conflict_share'), 'rocprof_us': pc.get('rocprof_us')})
    # backward, the formulation the autograd Function runs at this size (and the one the `rroi_bwd` passes
    # profiled): plan + gather (+ the atomics kernel for RoIs the plan cannot hold).  Algorithmic bytes of its
    # own: the gradient of the output is read once (25.7 MB), every touched map element written once.
    gout = torch.randn(512, 256, 7, 7, device=dev)
    gfeat = torch.empty_like(feat)
    ws = torch.empty(L.obb_roi_backward_workspace_bytes(512), dtype=torch.uint8, device=dev)
    alg_b = int(4 * K * C * 49 + 4 * C * min(foot, H * W) + 24 * K)

    def bwd_gather():
        _lib.check(L.obb_roi_backward_gather_f32(gout.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), 512, 0, 7, 7, 0.25, 2, 1.0,
                                                 gfeat.data_ptr(), ws.data_ptr(), ws.numel(), st))

    def bwd_scatter():
        gfeat.zero_()
        _lib.check(L.obb_rroi_align_backward_f32(gout.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), 512, 7, 7,
                                                 0.25, 2, 1.0, gfeat.data_ptr(), st))
    for tag, fn, names, wl, what in (
            ('rroi_bwd', bwd_gather, ['rroi_plan_kernel', 'rroi_gather_kernel', 'rroi_backward_kernel'], 'rroi_bwd',
             'gather formulation: plan + one block per map tile (+ atomics for unplanned RoIs); bit-reproducible'),
            ('rroi_bwd_scatter', bwd_scatter, ['rroi_backward_kernel'], 'rroi_bwd_scatter',
             'scatter formulation incl. the zero fill of the map (float atomics into the map)')):
        ms = _time_gpu(fn, 20)
        pc = sum_counters(names, wl) or {}
        res.append({'tag': tag, 'kernel': '+'.join(names), 'workload': 'RRoIAlign backward, same shape; ' + what,
                    'bound': 'hbm', 'algorithmic_bytes': alg_b, 'kernel_ms': round(ms, 4),
                    'achieved': round(alg_b / (ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(alg_b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'traffic': pc.get('traffic'),
                    'rocprof_us': pc.get('rocprof_us')})
    # ---- the same two ops on the same map held channels-last (B, H, W, C): pooled in place by the channels-last kernels
    # (round 5; obb_rroi_nhwc.hip), same algorithmic bytes, results bit-equal to the NCHW kernels'
    featl = feat.contiguous(memory_format=torch.channels_last)
    out_l = torch.empty((512, 256, 7, 7), device=dev)

    def fwd_nhwc():
        _lib.check(L.obb_rroi_align_forward_nhwc_f32(featl.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), 512, 7, 7, 0.25, 2, 1.0,
                                                     out_l.data_ptr(), st))
    ms = _time_gpu(fwd_nhwc, 50)
    assert torch.equal(out_l, out_f)
    pc = sum_counters(['roi_forward_nhwc_kernel'], 'rroi_fwd_nhwc') or {}
    res.append({'tag': 'rroi_fwd_nhwc', 'kernel': 'roi_forward_nhwc_kernel', 'workload': 'the same RRoIAlign forward on a channels-last map (bit-equal output)',
                'bound': 'hbm', 'algorithmic_bytes': alg, 'kernel_ms': round(ms, 4), 'rois_per_s': round(512 / (ms * 1e-3), 1),
                'achieved': round(alg / (ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'traffic': pc.get('traffic'),
                'lds_conflict_share': pc.get('lds_conflict_share'), 'rocprof_us': pc.get('rocprof_us')})
    gfl = torch.empty((1, 256, 256, 256), device=dev, memory_format=torch.channels_last)
    wsl = torch.empty(L.obb_roi_backward_nhwc_workspace_bytes(512, 256, 7, 7), dtype=torch.uint8, device=dev)

    def bwd_nhwc():
        _lib.check(L.obb_roi_backward_gather_nhwc_f32(gout.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), 512, 0, 7, 7, 0.25, 2, 1.0,
                                                      gfl.data_ptr(), wsl.data_ptr(), wsl.numel(), st))
    ms = _time_gpu(bwd_nhwc, 20)
    bwd_gather()
    assert torch.equal(gfl.contiguous(), gfeat)
    names = ['rroi_plan_kernel', 'roi_gather_nhwc_kernel', 'roi_backward_nhwc_direct_kernel']
    pc = sum_counters(names, 'rroi_bwd_nhwc') or {}
    res.append({'tag': 'rroi_bwd_nhwc', 'kernel': '+'.join(names), 'workload': 'the same backward into a channels-last gradient map (plan + transposition in one launch, gather, atomics tail; bit-equal)',
                'bound': 'hbm', 'algorithmic_bytes': alg_b, 'kernel_ms': round(ms, 4),
                'achieved': round(alg_b / (ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(alg_b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'traffic': pc.get('traffic'),
                'rocprof_us': pc.get('rocprof_us')})
    return {'measured_peak': peaks, 'kernels': res}


def extras(dev, args):
    """Side metrics outside the timed region (same process, current stream)."""
    import numpy as np
    import torch

    import oracle  # CPU legs of the side metrics only (checker / baseline, never timed as product)
    import workloads
    from aidet_amd import RRoIAlign, RoIAlign, _lib, nms, rbbox_overlaps
    from aidet_amd.nms_wrapper import _run_nms
    res = {}
    # configs[0]: 1k x 1k on a 1024 canvas (latency bound on the GPU) + the CPU path beside it
    a, b = make_iou(dev, 1000, 1024)
    ms = _time_gpu(lambda: rbbox_overlaps(a, b), 200)
    an, bn = a.cpu().numpy(), b.cpu().numpy()
    t0 = time.perf_counter()
    ref = oracle.overlaps(an, bn)
    dt = time.perf_counter() - t0
    res['iou_1k_x_1k'] = {'ms': round(ms, 4), 'pairs_per_s': round(1e6 / (ms * 1e-3), 1),
                          'cpu_oracle_ms': round(dt * 1e3, 2), 'cpu_oracle_pairs_per_s': round(1e6 / dt, 1),
                          'max_abs_diff_vs_oracle': float(np.abs(rbbox_overlaps(a, b).cpu().numpy() - ref).max())}
    # per-call latency of rnms at the sizes a detection head sees: fused (n <= 128), direct
    # (n <= 10240) and sorted routes; full call incl. the count read-back
    for nn in (100, 2048, 20000):
        dd = torch.from_numpy(workloads.boxes(nn, (nn ** 0.5) * 63.0, 7)).to(dev)
        for _ in range(5):
            _run_nms(dd, _lib.KIND_RBOX, 0.5)
        torch.cuda.synchronize()
        ts = []
        for _ in range(200):       # every call is synchronous: per-call wall times, median
            t0 = time.perf_counter()
            kk, _info = _run_nms(dd, _lib.KIND_RBOX, 0.5)
            ts.append(time.perf_counter() - t0)
        us = float(np.median(ts)) * 1e6
        res['rnms_latency_n%d' % nn] = {'us_per_call': round(us, 1), 'boxes_per_s': round(nn / us * 1e6, 1),
                                        'kept': int(kk.numel()), 'stat': 'median of 200 calls',
                                        'p99_us': round(float(np.percentile(ts, 99)) * 1e6, 1)}
    # the RPN-shaped horizontal NMS (2000 boxes, thr 0.7, rpn_head.py:92): sync call vs 5 queued
    rng = np.random.default_rng(2)
    x1, y1 = rng.uniform(0, 900, 2000), rng.uniform(0, 900, 2000)
    hb = np.stack([x1, y1, x1 + rng.uniform(8, 200, 2000), y1 + rng.uniform(8, 200, 2000),
                   (rng.permutation(2000) + .5) / 2000], 1).astype(np.float32)
    hd = torch.from_numpy(hb).to(dev)
    for _ in range(5):
        nms(hd, 0.7)
    ts = []
    for _ in range(100):
        t0 = time.perf_counter()
        nms(hd, 0.7)
        ts.append(time.perf_counter() - t0)
    from aidet_amd import nms_wrapper
    ta = []
    for _ in range(50):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nms_wrapper.nms_collect([nms_wrapper.nms_enqueue(hd, 0.7) for _ in range(5)])
        ta.append(time.perf_counter() - t0)
    sn = []
    for _ in range(12):
        t0 = time.perf_counter()
        sd, si = nms_wrapper.soft_nms(hd, 0.5, min_score=0.05)
        sn.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    _rd, ri = oracle.soft_nms(hb, 0.5, 'linear', 0.5, 0.05)
    res['soft_nms_2000'] = {'ms_per_call': round(float(np.median(sn[2:])) * 1e3, 3), 'kept': int(si.numel()),
                            'cpu_oracle_ms': round((time.perf_counter() - t0) * 1e3, 2),
                            'index_exact_vs_oracle': bool(np.array_equal(si.cpu().numpy(), ri))}
    res['hbb_nms_rpn_2000'] = {'us_per_sync_call': round(float(np.median(ts)) * 1e6, 1),
                               'us_per_call_5_queued': round(float(np.median(ta)) * 1e6 / 5, 1),
                               'kept': int(nms(hd, 0.7)[1].numel())}
    # the headline workload through the bare C ABI (caller-owned keep + workspace)
    L = _lib.lib()
    nn = 100000
    dd = torch.from_numpy(workloads.boxes(nn, 20000.0, 3)).to(dev)
    keepb = torch.empty(nn, dtype=torch.int64, device=dev)
    wsb = torch.empty(L.obb_nms_workspace_bytes(nn, 0, 0), dtype=torch.uint8, device=dev)
    infob = (ctypes.c_int64 * 4)()
    stp = _lib.current_stream_handle(dev)
    ts = []
    for i in range(60):
        t0 = time.perf_counter()
        rc = L.obb_nms_f32(dd.data_ptr(), nn, 0, 0.5, keepb.data_ptr(), infob, wsb.data_ptr(), wsb.numel(), stp)
        ts.append(time.perf_counter() - t0)
        assert rc == 0
    us = float(np.median(ts[10:])) * 1e6
    res['rnms_100k_c_abi'] = {'us_per_call': round(us, 1), 'boxes_per_s': round(nn / us * 1e6, 1), 'kept': int(infob[0]),
                              'stat': 'median of 50 calls of obb_nms_f32 (stateless: four launches)'}
    # ... and with OBB_NMS_RETAIN on the same caller-owned workspace (three launches from the second call on), over a pool of scenes
    poolb = [dd] + [torch.from_numpy(workloads.boxes(nn, 20000.0, 3 + 10 * k)).to(dev) for k in (1, 2, 3)]
    ts = []
    for i in range(64):
        t0 = time.perf_counter()
        rc = L.obb_nms_ex_f32(poolb[i & 3].data_ptr(), nn, 0, 0.5, None, None, _lib.NMS_RETAIN, keepb.data_ptr(), infob, wsb.data_ptr(),
                              wsb.numel(), stp)
        ts.append(time.perf_counter() - t0)
        assert rc == 0
    us = float(np.median(ts[12:])) * 1e6
    res['rnms_100k_c_abi_retained'] = {'us_per_call': round(us, 1), 'boxes_per_s': round(nn / us * 1e6, 1), 'kept': int(infob[0]),
                                       'last_retained': int(L.obb_nms_last_retained()),
                                       'stat': 'median of 52 calls of obb_nms_ex_f32 with OBB_NMS_RETAIN, four scenes in turn'}
    # ... and FOUR host threads at once, each with a stream, a keep list and a retained workspace of its own (the library's
    # state is per thread and per workspace; ctypes releases the GIL for the call): what the chip does when it is handed
    # more than one latency-bound call at a time -- a throughput beside the headline's per-call time, not part of `value`
    import threading
    nthreads, per_thread = 4, 60
    bar = threading.Barrier(nthreads + 1)
    errs = []

    def worker(t):
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                k2 = torch.empty(nn, dtype=torch.int64, device=dev)
                w2 = torch.empty(wsb.numel(), dtype=torch.uint8, device=dev)
                i2 = (ctypes.c_int64 * 4)()
                s2 = _lib.current_stream_handle(dev)
                for i in range(6):
                    L.obb_nms_ex_f32(poolb[(i + t) & 3].data_ptr(), nn, 0, 0.5, None, None, _lib.NMS_RETAIN, k2.data_ptr(), i2, w2.data_ptr(), w2.numel(), s2)
                torch.cuda.synchronize()
                bar.wait()
                for i in range(per_thread):
                    if L.obb_nms_ex_f32(poolb[(i + t) & 3].data_ptr(), nn, 0, 0.5, None, None, _lib.NMS_RETAIN, k2.data_ptr(), i2, w2.data_ptr(),
                                        w2.numel(), s2) != 0:
                        errs.append(t)
                bar.wait()
        except Exception as e:
            errs.append(repr(e))
            bar.abort()

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for th in ths:
        th.start()
    try:
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        dt = time.perf_counter() - t0
    except threading.BrokenBarrierError:
        dt = None
    for th in ths:
        th.join()
    if dt and not errs:
        res['rnms_100k_4_threads_4_streams'] = {'boxes_per_s': round(nthreads * per_thread * nn / dt, 1), 'us_per_call_per_thread': round(dt / per_thread * 1e6, 1),
                                                'stat': '%d threads x %d synchronous calls of obb_nms_ex_f32 (OBB_NMS_RETAIN), a stream and a workspace each' % (nthreads, per_thread)}
    # RRoIAlign configs[2]: value check against the oracle on ALL 512 RoIs, 1 thread and all cores
    feat, rois = make_rroi(dev)
    layer = RRoIAlign(7, 0.25, sample_num=2)
    fn, rn = feat.cpu().numpy(), rois.cpu().numpy()
    t0 = time.perf_counter()
    ref = oracle.rroi_align_forward(fn, rn, 7, 0.25, 2)
    dt1 = time.perf_counter() - t0
    nthr = oracle.max_threads()
    t0 = time.perf_counter()
    oracle.rroi_align_forward(fn, rn, 7, 0.25, 2, threads=nthr)
    dtn = time.perf_counter() - t0
    got = layer(feat, rois).cpu().numpy()
    res['rroi_align_512x256x7x7'] = {'max_abs_diff_vs_oracle': float(np.abs(got - ref).max()),
                                     'cpu_oracle_rois_per_s_1_thread': round(512 / dt1, 1),
                                     'cpu_oracle_rois_per_s_all_cores': round(512 / dtn, 1), 'cpu_threads': nthr}
    # the horizontal RoIAlign drop-in at the shape config 4 calls it with (2000 RoIs, one level)
    hr = torch.from_numpy(np.concatenate([np.zeros((2000, 1), np.float32), hb[:, :4]], 1)).to(dev)
    hl = RoIAlign(7, 0.25, sample_num=2)
    res['roi_align_2000x256x7x7'] = {'ms': round(_time_gpu(lambda: hl(feat, hr), 30), 4)}
    # RoI-Align backward, both formulations (scatter: atomics into a zero-filled map; gather: plan + one block per
    # map tile, overwrites the map, bit-reproducible), at configs[2] and at the config-4 shape above
    stp = _lib.current_stream_handle(dev)
    bw = {}
    for tag, rr, kind, ms_ in (('rroi_512', rois, 0, 1.0), ('roi_2000', hr, 1, 0.0)):
        kk = rr.shape[0]
        go = torch.randn(kk, 256, 7, 7, device=dev)
        gm = torch.empty_like(feat)
        wsb2 = torch.empty(L.obb_roi_backward_workspace_bytes(kk), dtype=torch.uint8, device=dev)

        def gather():
            _lib.check(L.obb_roi_backward_gather_f32(go.data_ptr(), 1, 256, 256, 256, rr.data_ptr(), kk, kind, 7, 7, 0.25, 2, ms_,
                                                     gm.data_ptr(), wsb2.data_ptr(), wsb2.numel(), stp))

        def scatter():
            gm.zero_()
            if kind == 0:
                _lib.check(L.obb_rroi_align_backward_f32(go.data_ptr(), 1, 256, 256, 256, rr.data_ptr(), kk, 7, 7, 0.25, 2, 1.0, gm.data_ptr(), stp))
            else:
                _lib.check(L.obb_roi_align_backward_f32(go.data_ptr(), 1, 256, 256, 256, rr.data_ptr(), kk, 7, 7, 0.25, 2, 0, gm.data_ptr(), stp))
        bw[tag] = {'gather_ms': round(_time_gpu(gather, 20), 4), 'scatter_incl_zero_fill_ms': round(_time_gpu(scatter, 20), 4)}
    res['roi_align_backward_256ch_7x7'] = bw
    try:
        from aidet_amd.models import bench_cfg4
        res['cfg4_tile'] = bench_cfg4(dev)
    except ImportError:
        pass
    return res


def cpu_side_baselines():
    """The oracle's other two ops on this host, 1 thread, bounded samples: rbbox_overlaps 1000 x 1000 on a 1024 canvas
    (BASELINE configs[0] itself) and RRoIAlign on 128 of configs[2]'s 512 RoIs (256 channels, 7 x 7, 2 x 2 samples)."""
    import numpy as np
    import oracle
    import workloads
    a, b = workloads.boxes(1000, 1024, 1)[:, :5].copy(), workloads.boxes(1000, 1024, 2)[:, :5].copy()
    oracle.overlaps(a[:64], b)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 1.0:
        oracle.overlaps(a, b)
        reps += 1
    iou_rate = reps * 1e6 / (time.perf_counter() - t0)
    rng = np.random.default_rng(5)
    feat = rng.standard_normal((1, 256, 256, 256)).astype(np.float32)
    r = workloads.boxes(512, 1024, 5)[:128, :5]
    rois = np.concatenate([np.zeros((128, 1), np.float32), r], 1)
    t0 = time.perf_counter()
    oracle.rroi_align_forward(feat, rois, 7, 0.25, 2)
    roi_rate = 128 / (time.perf_counter() - t0)
    return {'iou_pairs_per_s_1_thread': round(iou_rate, 1), 'iou_sample': 'rbbox_overlaps 1000 x 1000, canvas 1024 (configs[0]), %d repetitions' % reps,
            'rroi_rois_per_s_1_thread': round(roi_rate, 1), 'rroi_sample': '128 of the 512 RoIs of configs[2], 256 channels, 7 x 7 bins, 2 x 2 samples'}


def cpu_baseline(boxes, keep):
    """The C oracle (port of the reference's serial loop, nms_cpu.cpp:35-58) on this host on the
    SAME boxes: 1 thread -- what the reference's CPU path is -- and all cores (the suppression
    loop of each kept box dealt to OpenMP threads); also confirms the GPU keep set."""
    import numpy as np
    import oracle
    t0 = time.perf_counter()
    ref = oracle.rnms(boxes, 0.5)
    dt1 = time.perf_counter() - t0
    nthr = oracle.max_threads()
    runs = []
    for _ in range(3):      # (the all-cores leg varies 4.6-6.7 s from run to run: best of 3, all three stated)
        t0 = time.perf_counter()
        ref_mt = oracle.rnms_mt(boxes, 0.5, nthr)
        runs.append(time.perf_counter() - t0)
    dtn = min(runs)
    model = ''
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    same = bool(np.array_equal(ref, keep.cpu().numpy())) and bool(np.array_equal(ref, ref_mt))
    side = cpu_side_baselines()
    return {**side, 'value': round(len(boxes) / dt1, 1), 'unit': 'boxes/s', 'cores': 1, 'kind': 'port',
            'sample': 'the full %d-box workload, one run, %.1f s' % (len(boxes), dt1),
            'all_cores_value': round(len(boxes) / dtn, 1), 'all_cores_threads': nthr,
            'all_cores_sample': 'full workload, best of 3 runs (%s s)' % ', '.join('%.1f' % r for r in runs),
            'host_cpus': os.cpu_count(), 'cpu_model': model, 'keep_bit_exact_vs_gpu': same}


# ==============================================================================================
# N > 1 (and the one-GPU side metric): tiled inference, gather + merged rotated NMS
# ==============================================================================================
_scenes = {}


def tiled_scene_cached(images):
    import workloads
    if images not in _scenes:
        _scenes.clear()                 # (one scene at a time: the large one is 0.4 GB of host arrays)
        _scenes[images] = workloads.tiled_scene(images=images)
    return _scenes[images]


def tiled_on_this_rank(dev, rank, world, steps, warmup, layout, group=None, barrier=None, images=16):
    """Runs BASELINE configs[4] with this rank's patches; returns (summary dict, elapsed s,
    kept sequence numbers of the last step)."""
    import numpy as np
    import torch

    from aidet_amd import dota_io, dota_merge
    scene = tiled_scene_cached(images)
    C = scene['num_classes']
    P = len(scene['patch_name'])
    mine = np.isin(scene['patch'], np.asarray(dota_merge.shard_patches(P, rank, world)))
    idx = np.nonzero(mine)[0]
    dets = torch.from_numpy(scene['dets'][idx]).to(dev)
    labels = torch.from_numpy(scene['label'][idx]).to(dev)
    patch = torch.from_numpy(scene['patch'][idx]).to(dev)
    seq = torch.from_numpy(idx.astype(np.int64)).to(dev)           # global row number = single-process order
    # patch offsets as the reference's merge recovers them: from the patch FILE NAMES
    xy = np.array([dota_io.parse_patch_name(nm)[2:4] for nm in scene['patch_name']], np.float32)
    assert np.array_equal(xy, scene['patch_xy'])
    patch_xy = torch.from_numpy(xy).to(dev)
    patch_image = torch.from_numpy(scene['patch_image']).to(dev)
    thrs = dota_io.class_thresholds()
    total = len(scene['dets'])
    ev = {}

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev[name] = e

    def step(timed=False, gather_result=False):
        if timed:
            mark('start')
        # (routed: the kept rows stay with the rank that owns their image inside the timed steps -- the layout's
        # end state; one extra untimed pass below gathers them for the comparison with the single-rank run)
        return dota_merge.tiled_inference_merge(dets, labels, patch, seq, patch_xy, patch_image, C, 0.5, 0.3, thrs,
                                                group=group, layout=layout, mark=mark if timed else None,
                                                gather_result=gather_result or layout != 'routed')
    repeats0 = dota_merge.phase_repeats       # (a phase whose NMS gave the grid route up is repeated on the sorted route -- once: the thread then skips the grid)
    for _ in range(warmup + 1):
        kept, keep, allrows = step()
    if barrier:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        kept, keep, allrows = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if barrier:
        barrier()
    step(timed=True)       # one extra, untimed pass with events between the phases
    torch.cuda.synchronize()
    ph = {'local_nms_us': round(ev['start'].elapsed_time(ev['local']) * 1e3, 1),
          'gather_us': round(ev['local'].elapsed_time(ev['gather']) * 1e3, 1),
          'merge_us': round(ev['gather'].elapsed_time(ev['merge']) * 1e3, 1)}
    kept_here = int(keep.numel())
    if layout == 'routed' and world > 1:
        kept, keep, allrows = step(gather_result=True)
    summary = {'ms_per_step': round(elapsed * 1e3 / steps, 4), 'boxes_per_s': round(total * steps / elapsed, 1),
               'detections': total, 'patches': P, 'images': int(scene['patch_image'].max()) + 1,
               'local_detections': int(len(idx)), 'rows_merged_on_this_rank': int(allrows.size(0)),
               'kept_on_this_rank': kept_here, 'kept_after_merge': int(kept.size(0)), 'layout': layout,
               'phases_repeated_without_the_grid': int(dota_merge.phase_repeats - repeats0), **ph}
    return summary, elapsed, kept[:, :6].contiguous()


# ==============================================================================================
# profiling passes: one workload per kernel name (tools/collect_profiles.sh)
# ==============================================================================================
def profile_target(args):
    import torch
    import workloads
    from aidet_amd import RRoIAlign, RoIAlign, _lib, rbbox_overlaps
    from aidet_amd.nms_wrapper import _run_nms
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    t = args.profile_target
    if t == 'nms':
        # (the pool of scenes of the timed region: a call finds the frame of ANOTHER scene in the retained workspace)
        pool = [torch.from_numpy(workloads.boxes(args.boxes, args.canvas, 3 + 10 * k)).to(dev) for k in range(4)]
        cnt = [0]

        def fn():
            cnt[0] += 1
            return _run_nms(pool[cnt[0] & 3], _lib.KIND_RBOX, 0.5)
    elif t in ('iou_sparse', 'iou_dense'):
        a, b = make_iou(dev, 16384, 8192 if t == 'iou_sparse' else 1024)
        fn = lambda: rbbox_overlaps(a, b)
    elif t == 'iou_allclip':
        a, b = make_iou_allclip(dev, 4096)
        fn = lambda: rbbox_overlaps(a, b)
    elif t in ('rroi_fwd', 'rroi_bwd', 'rroi_bwd_scatter', 'rroi_fwd_nhwc', 'rroi_bwd_nhwc'):
        feat, rois = make_rroi(dev)
        layer = RRoIAlign(7, 0.25, sample_num=2)
        if t.endswith('_nhwc'):      # the same map as channels-last memory: pooled in place by the channels-last kernels
            feat = feat.contiguous(memory_format=torch.channels_last)
        if t in ('rroi_fwd', 'rroi_fwd_nhwc'):
            fn = lambda: layer(feat, rois)
        elif t == 'rroi_bwd_scatter':   # the scatter formulation alone (incl. the zero fill it needs), through the C ABI
            g = torch.randn(512, 256, 7, 7, device=dev)
            gm = torch.empty_like(feat)
            st = _lib.current_stream_handle(dev)
            L = _lib.lib()

            def fn():
                gm.zero_()
                _lib.check(L.obb_rroi_align_backward_f32(g.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), 512, 7, 7,
                                                         0.25, 2, 1.0, gm.data_ptr(), st))
        else:
            feat.requires_grad_(True)
            g = torch.randn(512, 256, 7, 7, device=dev)

            def fn():
                feat.grad = None
                layer(feat, rois).backward(g)
    elif t in ('roi_fwd', 'roi_fwd_nhwc'):
        import numpy as np
        feat, _ = make_rroi(dev)
        if t == 'roi_fwd_nhwc':
            feat = feat.contiguous(memory_format=torch.channels_last)
        rng = np.random.default_rng(2)
        x1, y1 = rng.uniform(0, 900, 2000), rng.uniform(0, 900, 2000)
        r = np.stack([np.zeros(2000), x1, y1, x1 + rng.uniform(8, 200, 2000), y1 + rng.uniform(8, 200, 2000)], 1)
        rois = torch.from_numpy(r.astype(np.float32)).to(dev)
        layer = RoIAlign(7, 0.25, sample_num=2)
        fn = lambda: layer(feat, rois)
    else:
        s, _, _ = tiled_on_this_rank(dev, 0, 1, steps=args.steps, warmup=args.warmup, layout='replicated')
        print(json.dumps({'profile_target': t, **s}))
        return
    for _ in range(args.warmup):
        fn()
    ms = _time_gpu(fn, args.steps)
    print(json.dumps({'profile_target': t, 'ms_per_call': round(ms, 4), 'calls': args.steps}))


if __name__ == '__main__':
    main()

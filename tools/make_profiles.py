"""Build profiles/<tag>_* from gpurun_out/collect (see tools/collect_profiles.sh) and print the
roofline fractions bench.py reports, recomputed from the CSVs alone."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r05'
src = 'gpurun_out/collect'
dst = 'profiles'
WORKLOAD = {'nms': 'nms_100000_20000', 'iou_sparse': 'iou_sparse', 'iou_dense': 'iou_dense', 'iou_allclip': 'iou_allclip',
            'rroi_fwd': 'rroi_fwd', 'rroi_bwd': 'rroi_bwd', 'rroi_bwd_scatter': 'rroi_bwd_scatter', 'roi_fwd': 'roi_fwd',
            'rroi_fwd_nhwc': 'rroi_fwd_nhwc', 'rroi_bwd_nhwc': 'rroi_bwd_nhwc', 'roi_fwd_nhwc': 'roi_fwd_nhwc'}
OURS = ('nms_', 'overlaps_', 'rroi_', 'roi_', 'soft_nms', 'rbox_to_poly')


def jline(path):
    for line in open(path):
        if line.startswith('{'):
            return json.loads(line)
    raise SystemExit('no JSON line in ' + path)


def short(name):
    return name.split('(')[0].replace('void ', '').replace('obb::', '').split('<')[0].strip()


def find(d, suffix):
    hits = glob.glob(os.path.join(d, '**', '*' + suffix), recursive=True)
    return hits[0] if hits else None


bench = jline(src + '/bench.json')
json.dump(bench, open('%s/%s_bench.json' % (dst, tag), 'w'), indent=1)
json.dump(jline(src + '/bench_dense.json'), open('%s/%s_bench_dense.json' % (dst, tag), 'w'), indent=1)
json.dump(jline(src + '/trace_headline.json'), open('%s/%s_bench_under_rocprof.json' % (dst, tag), 'w'), indent=1)


def copy_stats(trace_dir, out):
    rows = list(csv.reader(open(find(trace_dir, 'kernel_stats.csv'))))
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        for r in rows:
            r[0] = r[0] if len(r[0]) <= 140 else r[0][:137] + '...'
            w.writerow(r)
    return rows


copy_stats(src + '/trace_headline', '%s/%s_kernel_stats.csv' % (dst, tag))
kern = {}
for t, wl in WORKLOAD.items():
    td = '%s/trace_%s' % (src, t)
    if not os.path.isdir(td) or not find(td, 'kernel_stats.csv'):
        continue
    rows = copy_stats(td, '%s/%s_kernel_stats_%s.csv' % (dst, tag, t))
    hdr = rows[0]
    iname, icalls, iavg = hdr.index('Name'), hdr.index('Calls'), hdr.index('AverageNs')
    here = {}
    for r in rows[1:]:
        k = short(r[iname])
        if k.startswith(OURS):
            here[k] = {'workload': wl, 'calls_traced': int(r[icalls]), 'avg_us': round(float(r[iavg]) / 1e3, 2)}
    for i in range(1, 5):
        f = find('%s/pmc_%s_%d' % (src, t, i), 'counter_collection.csv')
        if not f:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(lambda: collections.defaultdict(set))
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[k][r['Counter_Name']].add(r['Dispatch_Id'])
        for k in agg:
            if k in here:
                for c, v in agg[k].items():
                    key = c + '_KB' if c in ('FETCH_SIZE', 'WRITE_SIZE') else c
                    here[k][key] = round(v / len(cnt[k][c]), 1)
    for k, v in here.items():                      # one record per (kernel, workload)
        kern[k if k not in kern else '%s@%s' % (k, wl)] = dict(v, kernel=k)
out = {
    'what': 'per-launch means from rocprofv3 on the MI355X box (tools/collect_profiles.sh): avg_us from '
            '`--kernel-trace --stats` of `bench.py --profile-target <t> --steps 20 --warmup 3`; counters from four '
            'separate `--kernel-trace --pmc` passes of the same command (--steps 5): FETCH_SIZE | WRITE_SIZE | '
            'SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES | SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE '
            'SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES.  FETCH/WRITE in KB as reported; HBM bytes = 2*FETCH_SIZE_KB*1024 + '
            'WRITE_SIZE_KB*1024 (gfx950: FETCH_SIZE reports half of a wide coalesced read, MI355X_MICROARCH.md).  '
            'SQ counters are sums over the 8 XCDs.',
    'kernels': dict(sorted(kern.items())),
}
json.dump(out, open('%s/%s_counters.json' % (dst, tag), 'w'), indent=1)

# ---- the fractions of bench.py's roofline objects, from the files above alone ----------------
VALU_PEAK = 256 * 4 * 2.4e9 / 2.0


def hbm(k):
    return 2 * k.get('FETCH_SIZE_KB', 0) * 1024 + k.get('WRITE_SIZE_KB', 0) * 1024


print('kernel                         workload            avg_us   HBM MB   HBM frac  VALU-issue frac')
for k, v in sorted(kern.items()):
    t = v['avg_us'] * 1e-6
    line = '%-30s %-18s %8.1f %8.2f %9.4f' % (v['kernel'], v['workload'], v['avg_us'], hbm(v) / 1e6, hbm(v) / t / 8e12)
    if 'SQ_INSTS_VALU' in v:
        line += ' %10.4f' % (v['SQ_INSTS_VALU'] / t / VALU_PEAK)
    print(line)
call = [v for v in kern.values() if v['workload'] == WORKLOAD['nms']]
if call and bench['config'].get('nms_launches') == 3:
    # a call on a retained workspace (round 6): the frame kernel ran once, for the seeding call -- not part of a timed call
    call = [v for v in call if v['kernel'] != 'nms_frame_kernel']
if call:
    print('rnms 100k: %d kernels, sum of avg_us %.1f, HBM bytes per call %.1f MB; bench ms_per_step %.4f'
          % (len(call), sum(v['avg_us'] for v in call), sum(hbm(v) for v in call) / 1e6, bench['ms_per_step']))
for name, alg in (('iou_sparse', 20 * 2 * 16384 + 4 * 16384 ** 2), ('iou_dense', 20 * 2 * 16384 + 4 * 16384 ** 2)):
    ks = [v for k, v in kern.items() if v['workload'] == name and v['kernel'].startswith('overlaps_tile')]
    if ks:
        print('%s: algorithmic %.1f MB / %.1f us = %.0f GB/s = %.3f of 8 TB/s' % (
            name, alg / 1e6, ks[0]['avg_us'], alg / ks[0]['avg_us'] / 1e3, alg / ks[0]['avg_us'] / 1e3 / 8000))
for name in ('multi_rank_sim.txt', 'multi_rank_sim_large.txt', 'roi_backward.txt', 'detect.json', 'soft_nms.txt', 'rroi_ablate.txt', 'rroi_stamps.txt', 'gprobe_stamps.txt', 'frame_stamps.txt', 'tiled_grid.txt', 'grid_vs_sorted.txt',
             'nhwc_stamps.txt', 'nhwc_fwd.txt', 'nhwc_bwd.txt', 'overlaps_density.txt', 'plan_stamps.txt', 'nhwc_gather_stamps.txt',
             'resolve_stamps.txt', 'resolve_parts.txt', 'resolve_tail.txt'):
    if os.path.exists(os.path.join(src, name)):
        with open(os.path.join(src, name)) as f, open('%s/%s_%s' % (dst, tag, name), 'w') as g:
            g.write(''.join(l for l in f if 'amdgpu.ids' not in l))
if os.path.isdir(src + '/trace_cfg4') and find(src + '/trace_cfg4', 'kernel_stats.csv'):
    copy_stats(src + '/trace_cfg4', '%s/%s_kernel_stats_cfg4.csv' % (dst, tag))
if os.path.isdir(src + '/trace_sim8') and find(src + '/trace_sim8', 'kernel_stats.csv'):
    copy_stats(src + '/trace_sim8', '%s/%s_kernel_stats_cfg5_rank0_of_8.csv' % (dst, tag))
rk = bench.get('roofline_kernels', {}).get('kernels', [])
for r in rk:
    if r['kernel'].startswith(('rroi_', 'roi_')):
        names = r['kernel'].split('+')
        ks = [v for v in kern.values() if v['kernel'] in names and v['workload'] == r.get('tag')]
        if ks:
            us, tb = sum(k['avg_us'] for k in ks), sum(hbm(k) for k in ks)
            print('%s [%s]: algorithmic %.1f MB / %.1f us (rocprof, %d kernels) = %.3f of 8 TB/s; counter/algorithmic = %.2f' % (
                r['kernel'], r.get('tag'), r['algorithmic_bytes'] / 1e6, us, len(ks), r['algorithmic_bytes'] / us / 1e3 / 8000,
                tb / r['algorithmic_bytes']))
print('bench value', bench['value'], 'ms/step', bench['ms_per_step'], 'roofline', json.dumps(bench['roofline'])[:300])

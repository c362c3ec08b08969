"""Build profiles/<tag>_* from gpurun_out/collect (see tools/collect_profiles.sh)."""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src = 'gpurun_out/collect'
dst = 'profiles'


def jline(path):
    for line in open(path):
        if line.startswith('{'):
            return json.loads(line)
    raise SystemExit('no JSON line in ' + path)


bench = jline(src + '/bench.json')
json.dump(bench, open('%s/%s_bench.json' % (dst, tag), 'w'), indent=1)
json.dump(jline(src + '/bench_dense.json'), open('%s/%s_bench_dense.json' % (dst, tag), 'w'), indent=1)
json.dump(jline(src + '/trace_bench.json'), open('%s/%s_bench_under_rocprof.json' % (dst, tag), 'w'), indent=1)
rows = list(csv.reader(open(src + '/trace/t_kernel_stats.csv')))
with open('%s/%s_kernel_stats.csv' % (dst, tag), 'w', newline='') as f:
    w = csv.writer(f)
    for r in rows:
        r[0] = r[0] if len(r[0]) <= 140 else r[0][:137] + '...'
        w.writerow(r)
# PMC: mean per launch, per kernel
kern = collections.defaultdict(dict)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    agg = collections.defaultdict(float)
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open('%s/pmc_%s/p_counter_collection.csv' % (src, c))):
        if r['Counter_Name'] != c:
            continue
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('obb::', '').split('<')[0]
        agg[k] += float(r['Counter_Value'])
        cnt[k].add(r['Dispatch_Id'])
    for k in agg:
        kern[k][c + '_KB'] = round(agg[k] / len(cnt[k]), 1)
        kern[k]['launches_' + c] = len(cnt[k])
out = {
    'what': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) around '
            '`python3 bench.py --steps 5 --warmup 2 --no-cpu`; mean per launch, KB as reported. '
            'HBM bytes = 2*FETCH_SIZE_KB*1024 + WRITE_SIZE_KB*1024 (gfx950 correction of '
            'MI355X_MICROARCH.md: FETCH_SIZE reports half of a wide coalesced read).',
    'workload': {'boxes': bench['config']['boxes_per_gpu'], 'canvas': 20000.0},
    'kernels': {k: v for k, v in sorted(kern.items()) if k.startswith(('nms_', 'overlaps_', 'rroi_'))},
}
json.dump(out, open('%s/%s_traffic.json' % (dst, tag), 'w'), indent=1)
print(json.dumps(out['kernels'], indent=1))
print('value', bench['value'], 'ms', bench['ms_per_step'])

"""Summarise a rocprofv3 --pmc run: per-kernel mean counter values and durations.
usage: python tools/pmc_summary.py <dir> <prefix>"""
import collections
import csv
import sys

d, pre = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open('%s/%s_counter_collection.csv' % (d, pre))))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in rows:
    k = r['Kernel_Name'].split('(')[0][-44:]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[k].add(r['Dispatch_Id'])
dur = collections.defaultdict(list)
try:
    for r in csv.DictReader(open('%s/%s_kernel_trace.csv' % (d, pre))):
        dur[r['Kernel_Name'].split('(')[0][-44:]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
except FileNotFoundError:
    pass
for k, v in agg.items():
    n = max(len(cnt[k]), 1)
    t = dur.get(k, [])
    print('%-46s n=%-3d us=%-8s %s' % (k, n, ('%.1f' % (sum(t) / len(t))) if t else '-',
                                       {c: round(x / n) for c, x in sorted(v.items())}))

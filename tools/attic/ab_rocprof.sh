#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace --stats) of the headline workload for several library builds, in one box
export TMPDIR=/tmp
for rep in 1 2; do
for lib in $LIBS; do
  O=gpurun_out/ab_$lib.$rep; rm -rf $O
  OBBOPS_LIB=/root/repo/aidet_amd/lib/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 bench.py --profile-target nms --steps 40 --warmup 5 > /dev/null 2>&1
  echo "== $lib rep $rep"
  python3 - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/t_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'nms_' in n:
        print('   %-28s calls %4s avg %7.2f us' % (n.split('(')[0].replace('void obb::','')[:28], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
done

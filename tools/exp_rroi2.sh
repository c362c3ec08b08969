export TMPDIR=/tmp
for g in 1 2 4; do
  echo -n "gpb=$g : "; OBB_RROI_GPB=$g timeout 100 python tools/bench_rroi.py 2>/dev/null
  for c in FETCH_SIZE; do
    rm -rf gpurun_out/rr_$g; OBB_RROI_GPB=$g timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/rr_$g -o p -- python3 tools/bench_rroi.py > /dev/null 2>&1
    f=$(find gpurun_out/rr_$g -name 'p_counter_collection.csv' | head -1); python tools/pmc_summary.py $(dirname $f) p | grep rroi
  done
done

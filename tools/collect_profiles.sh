#!/bin/bash
# Run ON THE MI355X BOX (through gpurun): produces everything profiles/ is built from.
#   gpurun -- 'bash tools/collect_profiles.sh'   then   python tools/make_profiles.py <tag>
export TMPDIR=/tmp
O=gpurun_out/collect
rm -rf $O; mkdir -p $O
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err
timeout 200 python bench.py --canvas 4096 --no-cpu --no-extras > $O/bench_dense.json 2>> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --no-cpu > $O/trace_bench.json 2> $O/trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu > $O/pmc_$c.json 2> $O/pmc_$c.err
done
ls -R $O | head -40

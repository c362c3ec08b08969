#!/bin/bash
# Run ON THE MI355X BOX (through gpurun): produces everything profiles/ is built from.
#   gpurun -- 'bash tools/collect_profiles.sh'   then   python tools/make_profiles.py r03
# One workload per kernel name and pass (bench.py --profile-target), counters in their own
# passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; SQ counters: 4 + 4).
export TMPDIR=/tmp
O=gpurun_out/collect
rm -rf $O; mkdir -p $O
TARGETS=${TARGETS:-"nms iou_sparse iou_dense rroi_fwd rroi_bwd rroi_bwd_scatter roi_fwd"}
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
timeout 200 python bench.py --canvas 4096 --no-cpu --no-extras > $O/bench_dense.json 2>> $O/bench.err
# kernel durations of the HEADLINE workload only (no side metrics: one size per kernel name)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_headline -o t -- python3 bench.py --no-cpu --no-extras > $O/trace_headline.json 2> $O/trace_headline.err
for t in $TARGETS; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$t -o t -- python3 bench.py --profile-target $t --steps 20 --warmup 3 > $O/trace_$t.json 2> $O/trace_$t.err
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_${t}_$i -o p -- python3 bench.py --profile-target $t --steps 5 --warmup 2 > $O/pmc_${t}_$i.json 2> $O/pmc_${t}_$i.err
  done
done
# side measurements that DESIGN.md quotes
timeout 300 python tools/exp_multi_rank_sim.py > $O/multi_rank_sim.txt 2>> $O/bench.err
timeout 200 python tools/bench_detect.py > $O/detect.json 2>> $O/bench.err
timeout 200 python tools/bench_roi_backward.py > $O/roi_backward.txt 2>> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg4 -o t -- python3 tools/prof_cfg4.py > /dev/null 2> $O/trace_cfg4.err
find $O -name '*.csv' | head -80

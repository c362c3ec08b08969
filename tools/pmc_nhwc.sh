#!/bin/bash
# Channels-last RoI-Align at configs[2] under rocprofv3: kernel trace + counter passes (run ON the GPU box).
#   gpurun -- 'bash tools/pmc_nhwc.sh [targets]'        results: gpurun_out/pmc_nhwc/
export TMPDIR=/tmp
O=gpurun_out/pmc_nhwc; rm -rf $O; mkdir -p $O
TARGETS=${1:-"rroi_fwd_nhwc"}
for t in $TARGETS; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$t -o t -- python3 bench.py --profile-target $t --steps 20 --warmup 3 > $O/trace_$t.json 2> $O/trace_$t.err
  f=$(find $O/trace_$t -name 't_kernel_stats.csv' | head -1); [ -n "$f" ] && head -8 $f
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TA_TA_BUSY_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_${t}_$i -o p -- python3 bench.py --profile-target $t --steps 5 --warmup 2 > /dev/null 2> $O/pmc_${t}_$i.err
    f=$(find $O/pmc_${t}_$i -name 'p_counter_collection.csv' | head -1)
    [ -n "$f" ] && python tools/attic/pmc_summary.py $(dirname $f) p | grep -i "roi\|nhwc" || tail -2 $O/pmc_${t}_$i.err
  done
done

#!/bin/bash
# instruction-mix / stall counters of the NMS kernels (run through gpurun)
#   BENCH_ARGS="--boxes 10000 --canvas 6300" bash tools/pmc_scan.sh
export TMPDIR=/tmp
O=gpurun_out/pmcscan
rm -rf $O; mkdir -p $O
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras ${BENCH_ARGS} > $O/$tag.json 2> $O/$tag.err
  f=$(find $O/$tag -name 'p_counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $(dirname $f) p | grep -E "scan|clip"
done

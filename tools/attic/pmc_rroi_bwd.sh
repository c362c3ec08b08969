# RoI-Align backward (gather formulation, configs[2]): where do the waves of the gather kernel spend their time?
export TMPDIR=/tmp
O=gpurun_out/pmcbwd4; rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- python3 bench.py --profile-target rroi_bwd --steps 5 --warmup 2 > /dev/null 2> $O/trace.err
f=$(find $O/trace -name 'p_kernel_stats.csv' | head -1); [ -n "$f" ] && grep -E "rroi|Name" $f | cut -c1-200
for set in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAVES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-24)
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -o p -- python3 bench.py --profile-target rroi_bwd --steps 5 --warmup 2 > /dev/null 2> $O/$tag.err
  f=$(find $O/$tag -name 'p_counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $(dirname $f) p | grep -E "rroi_gather|rroi_plan"
done

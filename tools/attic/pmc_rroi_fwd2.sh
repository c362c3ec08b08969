# RoI-Align forward (configs[2]): where do the waves wait?
export TMPDIR=/tmp
O=gpurun_out/pmcfwd2; rm -rf $O; mkdir -p $O
for set in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-24)
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -o p -- python3 bench.py --profile-target rroi_fwd --steps 5 --warmup 2 > /dev/null 2> $O/$tag.err
  f=$(find $O/$tag -name 'p_counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $(dirname $f) p | grep rroi_forward || tail -2 $O/$tag.err
done

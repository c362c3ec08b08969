export TMPDIR=/tmp
O=gpurun_out/pmcbwd; rm -rf $O; mkdir -p $O
python tools/exp_rroi_cpb.py | head -4
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-30)
  for t in rroi_bwd rroi_fwd; do
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/${t}_$tag -o p -- python3 bench.py --profile-target $t --steps 5 --warmup 2 > /dev/null 2> $O/${t}_$tag.err
  f=$(find $O/${t}_$tag -name 'p_counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $(dirname $f) p | grep rroi
  done
done

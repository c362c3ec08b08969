# RoI-Align forward, given vs Morton-ordered RoIs: fabric traffic, L2 hit rate and where the waves wait
export TMPDIR=/tmp
O=gpurun_out/pmcfwd; rm -rf $O; mkdir -p $O
for order in given morton; do
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS"; do
  tag=${order}_$(echo $set | tr ' ' '_' | cut -c1-24)
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -o p -- python3 tools/bench_rroi_fwd.py $order ${GROUPS_:--1} > /dev/null 2> $O/$tag.err
  f=$(find $O/$tag -name 'p_counter_collection.csv' | head -1)
  [ -n "$f" ] && echo "$order: $(python tools/pmc_summary.py $(dirname $f) p | grep rroi_forward)"
done
done

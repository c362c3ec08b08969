#!/bin/bash
# Run ON THE MI355X BOX (through gpurun): produces everything profiles/ is built from.
#   bash tools/build_ablations.sh            (here: the instrumented libraries travel with the snapshot)
#   gpurun -- 'bash tools/collect_profiles.sh'   then   python tools/make_profiles.py r04
# One workload per kernel name and pass (bench.py --profile-target), counters in their own
# passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; SQ counters: 4 + 4).
# Every rocprofv3 call: --output-format csv and a timeout (without csv the --stats post-processing of a long trace
# did not come back within 15 minutes on this pool).
export TMPDIR=/tmp
O=gpurun_out/collect
rm -rf $O; mkdir -p $O
TARGETS=${TARGETS:-"nms iou_sparse iou_dense iou_allclip rroi_fwd rroi_bwd rroi_bwd_scatter roi_fwd rroi_fwd_nhwc rroi_bwd_nhwc roi_fwd_nhwc"}
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 200 python bench.py --canvas 4096 --no-cpu --no-extras --no-cfg5-large > $O/bench_dense.json 2>> $O/bench.err
# the bench command itself under the profiler (MIXED: the NMS leg, both IoU legs, cfg5 -- per-kernel averages of one
# workload each are the per-target passes below)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_headline -o t -- python3 bench.py --no-cpu --no-extras --no-cfg5-large > $O/trace_headline.json 2> $O/trace_headline.err
for t in $TARGETS; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$t -o t -- python3 bench.py --profile-target $t --steps 20 --warmup 3 > $O/trace_$t.json 2> $O/trace_$t.err
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_${t}_$i -o p -- python3 bench.py --profile-target $t --steps 5 --warmup 2 > $O/pmc_${t}_$i.json 2> $O/pmc_${t}_$i.err
  done
done
# side measurements that DESIGN.md quotes
timeout 300 python tools/sim_multi_rank.py > $O/multi_rank_sim.txt 2>> $O/bench.err
timeout 400 python tools/sim_multi_rank.py --scene large --layouts routed > $O/multi_rank_sim_large.txt 2>> $O/bench.err
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_sim8 -o t -- python3 tools/sim_multi_rank.py --worlds 8 --layouts routed --steps 20 > /dev/null 2> $O/trace_sim8.err
timeout 200 python tools/bench_detect.py > $O/detect.json 2>> $O/bench.err
timeout 200 python tools/bench_roi_backward.py > $O/roi_backward.txt 2>> $O/bench.err
timeout 200 python tools/bench_soft_nms.py > $O/soft_nms.txt 2>> $O/bench.err
timeout 200 python tools/exp_rroi_ablate.py > $O/rroi_ablate.txt 2>> $O/bench.err
timeout 200 python tools/exp_rroi_stamps.py > $O/rroi_stamps.txt 2>> $O/bench.err
timeout 200 python tools/exp_nhwc_stamps.py 64 4 > $O/nhwc_stamps.txt 2>> $O/bench.err
timeout 200 python tools/exp_nhwc_fwd.py > $O/nhwc_fwd.txt 2>> $O/bench.err
timeout 200 python tools/exp_nhwc_bwd.py > $O/nhwc_bwd.txt 2>> $O/bench.err
timeout 200 python tools/exp_overlaps_density.py > $O/overlaps_density.txt 2>> $O/bench.err
(timeout 100 python tools/exp_plan_stamps.py; timeout 100 python tools/exp_plan_stamps.py nhwc) > $O/plan_stamps.txt 2>> $O/bench.err
(timeout 100 python tools/exp_nhwc_gather_stamps.py; timeout 100 python tools/exp_nhwc_gather_stamps.py 2000) > $O/nhwc_gather_stamps.txt 2>> $O/bench.err
timeout 200 python tools/exp_resolve_stamps.py 16 > $O/resolve_stamps.txt 2>> $O/bench.err
timeout 200 python tools/exp_resolve_parts.py 16 > $O/resolve_parts.txt 2>> $O/bench.err
timeout 200 python tools/exp_resolve_tail.py > $O/resolve_tail.txt 2>> $O/bench.err
timeout 200 python tools/exp_gprobe_stamps.py > $O/gprobe_stamps.txt 2>> $O/bench.err
timeout 200 python tools/exp_frame_stamps.py > $O/frame_stamps.txt 2>> $O/bench.err
timeout 300 python tools/exp_tiled_grid.py 16 0,1 > $O/tiled_grid.txt 2>> $O/bench.err
timeout 300 python tools/exp_grid_vs_sorted.py > $O/grid_vs_sorted.txt 2>> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg4 -o t -- python3 tools/prof_cfg4.py > /dev/null 2> $O/trace_cfg4.err
find $O -name '*.csv' | head -80

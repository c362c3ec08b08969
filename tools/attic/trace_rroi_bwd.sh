# kernel durations of the RoI-Align backward (gather formulation) at configs[2]
export TMPDIR=/tmp
O=gpurun_out/trbwd; rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- python3 bench.py --profile-target rroi_bwd --steps 5 --warmup 2 > /dev/null 2> $O/trace.err
f=$(find $O/trace -name 'p_kernel_stats.csv' | head -1); [ -n "$f" ] && grep -E "rroi" $f | cut -d, -f1,4 | sed 's/(.*)//' 

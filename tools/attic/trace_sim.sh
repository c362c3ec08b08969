# kernel trace of one rank's pass of the 8-rank routed simulation
export TMPDIR=/tmp
O=gpurun_out/trsim; rm -rf $O; mkdir -p $O
export SIM_ONLY=${SIM_ONLY:-routed8}
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o p -- python3 tools/exp_multi_rank_sim.py > $O/out.txt 2> $O/err.txt
cat $O/out.txt

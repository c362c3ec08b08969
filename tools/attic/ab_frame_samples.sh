#!/bin/bash
# A/B of the frame kernel's sample count in one box: the product library against a variant built with another
# -DOBB_FRAME_FEW_LOG2 (aidet_amd/lib/libobbops_fs256.so).  BOTH are named through OBBOPS_LIB: the Python wrapper then takes its
# ctypes path for both (the first version of this script ran the product through the pybind11 extension: ~2.4 us per call of bias).
for rep in 1 2; do
for lib in /root/repo/aidet_amd/lib/libobbops.so /root/repo/aidet_amd/lib/libobbops_fs256.so; do
  echo "== lib $lib"
  OBBOPS_LIB=$lib python tools/exp_grid_vs_sorted.py 2>&1 | grep -v amdgpu | tail -10 | cut -d'|' -f1,2,4
  OBBOPS_LIB=$lib python tools/exp_tiled_grid.py 16 1 2>&1 | tail -1
  OBBOPS_LIB=$lib python tools/attic/bench_loop_dbg.py 2>&1 | grep "watch 1" | head -1
done
done

#!/bin/bash
# UNBIASED A/B of library builds in one box: every variant is named through OBBOPS_LIB (with it the Python wrapper takes the
# ctypes path for ALL of them -- a variant named that way against the product, which goes through the pybind11 extension, is
# ~2.4 us per call apart for that reason alone: the first A/Bs of round 5's last hours were biased by it)
for rep in 1 2 3; do
for lib in $LIBS; do
  echo "== $lib"
  OBBOPS_LIB=/root/repo/aidet_amd/lib/$lib python tools/attic/bench_loop_dbg.py 2>&1 | grep "watch 1" | sed -n '4p'
  OBBOPS_LIB=/root/repo/aidet_amd/lib/$lib python tools/attic/ab_small_calls.py 2>&1 | grep -v amdgpu | cut -c1-120
done
done

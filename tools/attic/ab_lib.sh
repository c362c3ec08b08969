#!/bin/bash
# A/B of the product library against aidet_amd/lib/libobbops_old.so (built from another revision) in one box
for rep in 1 2 3; do
for lib in "" /root/repo/aidet_amd/lib/libobbops_old.so; do
  echo "== lib ${lib:-product}"
  OBBOPS_LIB=$lib python tools/attic/bench_loop_dbg.py 2>&1 | grep "watch 1" | sed -n '1p;4p'
  OBBOPS_LIB=$lib python tools/exp_resolve_tail.py 2>&1 | tail -4 | head -3
done
done

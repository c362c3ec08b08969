for v in "" NOCLIP NOSTORE; do
  if [ -n "$v" ]; then export OBBOPS_LIB=/root/repo/aidet_amd/lib/libobbops_ov_$v.so; else unset OBBOPS_LIB; fi
  echo "== build: ${v:-product}"; python tools/exp_overlaps_density.py 2>&1 | grep -v amdgpu.ids | head -3
done

# RRoIAlign: LDS tile size (per-wave quarter = OBB_STAGE_FLOATS / 4) and channels per block
for sf in 4096 6144 8192; do
  rm -f aidet_amd/csrc/obb_rroi.o; make -C aidet_amd/csrc EXTRA="-DOBB_STAGE_FLOATS=$sf" > /dev/null 2>&1
  for g in 2 4; do echo -n "stage=$sf gpb=$g : "; OBB_RROI_GPB=$g timeout 100 python tools/bench_rroi.py 2>/dev/null; done
done
rm -f aidet_amd/csrc/obb_rroi.o; make -C aidet_amd/csrc > /dev/null 2>&1

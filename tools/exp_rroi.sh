# RRoIAlign ablations: LDS tile size and channels per block
for st in 4096 6144 8192 12288; do
  rm -f aidet_amd/csrc/obb_rroi.o; make -C aidet_amd/csrc EXTRA=-DOBB_STAGE_FLOATS=$st > /dev/null 2>&1
  echo -n "stage=$st : "; timeout 100 python tools/bench_rroi.py
done
rm -f aidet_amd/csrc/obb_rroi.o; make -C aidet_amd/csrc > /dev/null 2>&1
for g in 1 2 3 4 5 8; do echo -n "gpb=$g : "; OBB_RROI_GPB=$g timeout 100 python tools/bench_rroi.py; done

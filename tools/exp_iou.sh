for fl in "" "-DOBB_OV_NOCLIP" "-DOBB_OV_NOSTORE" "-DOBB_OV_NOCLIP -DOBB_OV_NOSTORE"; do
  rm -f aidet_amd/csrc/obb_overlaps.o; make -C aidet_amd/csrc EXTRA="$fl" > /dev/null 2>&1
  echo -n "flags=[$fl] : "; timeout 100 python tools/bench_iou.py
done
rm -f aidet_amd/csrc/obb_overlaps.o; make -C aidet_amd/csrc > /dev/null 2>&1

for fl in "-DOBB_SCAN_NOATOMIC" "-DOBB_SCAN_NOSTORE" "-DOBB_SCAN_NOATOMIC -DOBB_SCAN_NOSTORE"; do
  rm -f aidet_amd/csrc/obb_nms.o; make -C aidet_amd/csrc EXTRA="$fl" > /dev/null 2>&1
  for c in 4096; do
  echo -n "flags=[$fl] canvas=$c: "; timeout 100 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu --canvas $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['phase_ms']['scan'], d['phase_ms']['clip'])"
  done
done
rm -f aidet_amd/csrc/obb_nms.o; make -C aidet_amd/csrc > /dev/null 2>&1

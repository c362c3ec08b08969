# overlaps ablations on the sparse 16k x 16k case (canvas 8192)
python - <<'PY'
import torch, time
x = torch.empty(16384, 16384, device='cuda')
for _ in range(3): x.zero_()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): x.zero_()
e.record(); torch.cuda.synchronize()
print('torch zero_ 1.07 GB: %.4f ms' % (s.elapsed_time(e) / 10))
PY
for fl in "" "-DOBB_OV_NOCLIP" "-DOBB_OV_NOSTORE"; do
  rm -f aidet_amd/csrc/obb_overlaps.o; make -C aidet_amd/csrc EXTRA="$fl" > /dev/null 2>&1
  echo -n "flags=[$fl] sparse: "; timeout 100 python tools/bench_iou.py 16384 8192
done
rm -f aidet_amd/csrc/obb_overlaps.o; make -C aidet_amd/csrc > /dev/null 2>&1

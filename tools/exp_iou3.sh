make -C aidet_amd/csrc > /dev/null 2>&1
for shape in "4096 4096 4096" "1000 1024 1000" "2000 2048 500" "8192 8192 8192" "500 1024 20000"; do
for r in 8 12 16 24 32 48; do
  echo -n "rows=$r shape=[$shape]: "; OBB_OV_ROWS=$r timeout 100 python tools/bench_iou.py $shape 2>/dev/null
done; done

for c in 20000 4096; do echo canvas $c; for pb in 1024 2048 4096 8192; do
echo -n "blocks=$pb : "; OBB_NMS_SCAN_BLOCKS=$pb timeout 100 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu --canvas $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phase_ms'], d['config']['kept'])"
done; done

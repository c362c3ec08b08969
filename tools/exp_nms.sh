for c in 20000 4096; do for pb in 2048 4096 8192; do echo -n "canvas=$c blocks=$pb: "; OBB_NMS_SCAN_BLOCKS=$pb timeout 100 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu --canvas $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phase_ms']['scan'], d['phase_ms']['clip'], d['config']['kept'])"
done; done

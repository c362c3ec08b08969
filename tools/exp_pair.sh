for c in 20000 4096; do timeout 100 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu --canvas $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phase_ms'], d['config']['kept'], d['config']['candidate_pairs'])"
done

export TMPDIR=/tmp
for nb in "2048 2851" "8192 5700" "2048 588"; do set -- $nb; echo "n=$1 canvas=$2"; timeout 100 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu --boxes $1 --canvas $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phase_ms'], d['config']['candidate_pairs'])"; done
rm -rf gpurun_out/tr; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr -o t -- python3 bench.py --no-cpu --no-extras --boxes 2048 --canvas 2851 --steps 50 > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/tr/**/t_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)): print(r['Name'][:60].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY

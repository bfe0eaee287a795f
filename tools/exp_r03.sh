set -x
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest1.log 2>&1; echo "pytest rc=$?" 
tail -3 gpurun_out/pytest1.log
for w in 4 8; do
  PEAPODS_RNG_WAVES=$w timeout -k 10 120 python bench.py --steps 2000 --warmup 100 --no-cpu-baseline > gpurun_out/bench_w${w}_2000.json 2> gpurun_out/bench_w${w}_2000.err; echo rc=$?
  PEAPODS_RNG_WAVES=$w timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_w${w}_20.json 2> gpurun_out/bench_w${w}_20.err; echo rc=$?
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_w*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])
    except Exception as e: print(f, 'ERR', e)
PY

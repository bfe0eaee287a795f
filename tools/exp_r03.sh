mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest6.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest6.log
for v in 0 1; do
  if [ $v = 1 ]; then export PEAPODS_NO_GRAPH=1; else unset PEAPODS_NO_GRAPH; fi
  timeout -k 10 120 python bench.py --steps 2000 --warmup 100 --no-cpu-baseline --no-other-configs --repeats 3 > gpurun_out/bench_gr${v}_2000.json 2> gpurun_out/bench_gr${v}_2000.err; echo rc=$?
  timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/bench_gr${v}_20.json 2> gpurun_out/bench_gr${v}_20.err; echo rc=$?
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_gr*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], [round(x*1e3,1) for x in d['ms_per_step_all_repeats']], d['host_enqueue_ms'])
    except Exception as e: print(f, 'ERR', e)
PY

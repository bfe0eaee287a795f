mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest5.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest5.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err; echo rc=$?
timeout -k 10 120 python bench.py --steps 2000 --warmup 100 --no-cpu-baseline --no-other-configs --repeats 3 > gpurun_out/bench_steady.json 2> gpurun_out/bench_steady.err; echo rc=$?
python - <<'PY'
import json
for f in ('gpurun_out/bench_driver.json','gpurun_out/bench_steady.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d.get('untimed_settling_calls'), [round(x*1e3,1) for x in d['ms_per_step_all_repeats']], d['roofline']['kernel_ms'], d['roofline']['frac'])
    if 'other_configs' in d:
        for k,v in d['other_configs'].items(): print(' ',k, v.get('value'), v.get('ms_per_step'), v.get('error'), v.get('roofline',{}).get('frac'))
    print(' cpu', d.get('cpu_baseline',{}).get('value'))
PY

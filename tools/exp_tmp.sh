mkdir -p gpurun_out/c4
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "houdayer or config4 or overlap_moves or cluster" > gpurun_out/c4/pytest.log 2>&1; echo rc=$?
tail -n 2 gpurun_out/c4/pytest.log
timeout -k 10 120 python bench.py --config c4 --oc-mode houdayer --steps 300 --warmup 30 --no-cpu-baseline --repeats 3 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"

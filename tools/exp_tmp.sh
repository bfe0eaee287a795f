mkdir -p gpurun_out/c1
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "eight_and_twelve or large_lattice" > gpurun_out/c1/pytest.log 2>&1; echo rc=$?
tail -n 12 gpurun_out/c1/pytest.log

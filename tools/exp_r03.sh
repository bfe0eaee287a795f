mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "reference_benchmark_lattice" > gpurun_out/pytest8.log 2>&1; echo "pytest rc=$?"
tail -40 gpurun_out/pytest8.log

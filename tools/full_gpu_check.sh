mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_full.log 2>&1; echo rc=$?
tail -n 3 gpurun_out/pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2

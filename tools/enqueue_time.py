"""Diagnostic: host enqueue time vs total time of one sample() of the bench protocol."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sim, *_ = bench.build_ensemble(0, 128, 1)
cfg = sim.make_config(300, "metropolis", pt_interval=1, warmup_ratio=0.0)
sim.run(cfg)
for _ in range(2):
    t0 = time.perf_counter(); sim.run(cfg, wait=False); t1 = time.perf_counter(); sim.synchronize(); t2 = time.perf_counter()
    print("streams=%s enqueue %.1f us/sweep, total %.1f us/sweep" % (os.environ.get("PEAPODS_STREAMS", "2"), (t1 - t0) / 300 * 1e6, (t2 - t0) / 300 * 1e6))

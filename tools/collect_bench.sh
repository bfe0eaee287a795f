#!/bin/bash
# Runs on the GPU box (from the repo root): the bench lines and diagnostics DESIGN.md quotes, into gpurun_out/bench_<tag>/.
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/collect_bench.sh r02'   then copy into profiles/
tag=${1:-r04}
out=gpurun_out/bench_$tag
mkdir -p "$out"
python bench.py --steps 2000 --warmup 100 --no-other-configs > "$out/bench.json" 2>"$out/bench.err"
python bench.py --steps 20 --warmup 5 > "$out/bench_20steps.json" 2>>"$out/bench.err"
python bench.py --steps 2000 --warmup 100 --pt-schedule full_ladder --no-cpu-baseline --no-other-configs > "$out/bench_full_ladder.json" 2>>"$out/bench.err"
PEAPODS_DEVELOPER=1 PEAPODS_STREAMS=1 python bench.py --steps 2000 --warmup 100 --no-cpu-baseline --no-other-configs > "$out/bench_1stream.json" 2>>"$out/bench.err"
python bench.py --config c1 --steps 300 --warmup 30 > "$out/bench_c1.json" 2>>"$out/bench.err"
for m in houdayer jorg cmr; do
    python bench.py --config c4 --oc-mode $m --steps 200 --warmup 20 --no-cpu-baseline > "$out/bench_c4_$m.json" 2>>"$out/bench.err"
done
for m in houdayer jorg; do   # the union-find forms
    python bench.py --config c4 --oc-mode $m --oc-variant sw --steps 200 --warmup 20 --no-cpu-baseline > "$out/bench_c4_${m}_sw.json" 2>>"$out/bench.err"
done
python tools/stamps_reg.py > "$out/stamps_reg.txt" 2>&1
python tools/stamps_cluster.py jorg sw > "$out/stamps_cluster_jorg_sw.txt" 2>&1
python tools/stamps_cluster.py houdayer sw > "$out/stamps_cluster_houdayer_sw.txt" 2>&1
python tools/stamps_hc.py > "$out/stamps_hc.txt" 2>&1
python tools/stamps_slab.py > "$out/stamps_slab.txt" 2>&1
python tools/fixed_cost.py > "$out/fixed_cost.json" 2>&1
python tools/determinism_soak.py full_ladder > "$out/determinism_soak.txt" 2>&1
python tools/time_variants.py > "$out/variants.txt" 2>&1
python tools/call_cost.py > "$out/call_cost.json" 2>&1
python tools/time_gaussian.py metropolis 300 > "$out/gaussian.txt" 2>&1
PEAPODS_BENCH_FORCE_COMM=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > "$out/bench_20steps_one_rank_rccl.json" 2>>"$out/bench.err"
echo done

#!/bin/bash
# Runs on the GPU box (from the repo root): rocprofv3 summaries behind profiles/<tag>_*.
#   bash tools/collect_profiles.sh r02          (normally through tools/run_collect.sh, which labels the commit)
# 1. kernel trace + stats of the bench command (shorter run, same workload), 2 stream groups and 1
# 2. PMC passes, each in its own run with --kernel-trace only (no other trace domain)
# 3. kernel stats of the other BASELINE configs (c1: 64^3 ferromagnet, c4: 12^3 with the overlap moves)
tag=${1:-r02}
out=$PWD/gpurun_out/prof_$tag
repo=$PWD
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- \
    python3 "$repo/bench.py" --steps 500 --warmup 50 --no-cpu-baseline --no-step-profile --no-other-configs --repeats 1 --settle-calls 0 > "$out/stats.log" 2>&1
echo "stats done" >> "$out/progress.log"
# the default run interleaves two stream groups of 64 realisations, so its per-launch averages are
# for half the ensemble; the remaining passes use one group (every launch = the full ensemble, as
# in bench.py's roofline.kernel_ms)
export PEAPODS_DEVELOPER=1 PEAPODS_STREAMS=1   # (kernel-selection switches count only behind the developer gate)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_1stream" -o bench -- \
    python3 "$repo/bench.py" --steps 500 --warmup 50 --no-cpu-baseline --no-step-profile --no-other-configs --repeats 1 --settle-calls 0 > "$out/stats_1stream.log" 2>&1
echo "stats 1 stream done" >> "$out/progress.log"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    name=$(echo "$pass" | tr ' ' '_' | cut -c1-24)
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$out/pmc_$name" -o pmc -- \
        python3 "$repo/bench.py" --steps 25 --warmup 5 --no-cpu-baseline --no-step-profile --no-other-configs --repeats 1 --settle-calls 0 --profile-sweeps 4 > "$out/pmc_$name.log" 2>&1
    echo "pass $name done" >> "$out/progress.log"
done
unset PEAPODS_STREAMS PEAPODS_DEVELOPER
# counters of the other BASELINE configs' kernels (c1: 64^3 slab sweep; c4: the overlap-cluster kernels), each counter
# group in its own run, --kernel-trace only
for cfg in "c1:--config c1" "c4_houdayer:--config c4 --oc-mode houdayer" "c4_jorg:--config c4 --oc-mode jorg"; do
    cname=${cfg%%:*}; cargs=${cfg#*:}
    for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        name=$(echo "$pass" | tr ' ' '_' | cut -c1-24)
        rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$out/pmcx_${cname}_$name" -o pmc -- \
            python3 "$repo/bench.py" $cargs --steps 12 --warmup 3 --no-cpu-baseline --no-step-profile --repeats 1 --settle-calls 0 --profile-sweeps 2 > "$out/pmcx_${cname}_$name.log" 2>&1
        echo "pass $cname $name done" >> "$out/progress.log"
    done
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_c1" -o bench -- \
    python3 "$repo/bench.py" --config c1 --steps 300 --warmup 30 --no-cpu-baseline --no-step-profile --repeats 1 --settle-calls 0 > "$out/stats_c1.log" 2>&1
echo "c1 done" >> "$out/progress.log"
for mode in houdayer jorg cmr; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_c4_$mode" -o bench -- \
        python3 "$repo/bench.py" --config c4 --oc-mode $mode --steps 200 --warmup 20 --no-cpu-baseline --no-step-profile --repeats 1 --settle-calls 0 > "$out/stats_c4_$mode.log" 2>&1
    echo "c4 $mode done" >> "$out/progress.log"
done
for mode in houdayer jorg; do   # the union-find forms (BASELINE configs[4] names "BFS/union-find")
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_c4_${mode}_sw" -o bench -- \
        python3 "$repo/bench.py" --config c4 --oc-mode $mode --oc-variant sw --steps 200 --warmup 20 --no-cpu-baseline --no-step-profile --repeats 1 --settle-calls 0 > "$out/stats_c4_${mode}_sw.log" 2>&1
    echo "c4 $mode sw done" >> "$out/progress.log"
done
cd "$repo" && python3 tools/summarize_profiles.py "$out" "$tag" > "$out/summary.log" 2>&1
cat "$out/summary.log"

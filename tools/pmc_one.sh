#!/bin/bash
# Counter passes for one bench configuration (developer aid): bash tools/pmc_one.sh <name> <bench args...>
# Each counter group in its own run with --kernel-trace only; output under gpurun_out/pmc1_<name>/.
name=$1; shift
out=$PWD/gpurun_out/pmc1_$name
repo=$PWD
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export PEAPODS_DEVELOPER=1 PEAPODS_STREAMS=1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    pn=$(echo "$pass" | tr ' ' '_' | cut -c1-24)
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$out/$pn" -o pmc -- \
        python3 "$repo/bench.py" "$@" --steps 12 --warmup 3 --no-cpu-baseline --no-step-profile --repeats 1 --settle-calls 0 --profile-sweeps 2 > "$out/$pn.log" 2>&1
    echo "pass $pn done" >> "$out/progress.log"
done
cd "$repo" && python3 - "$out" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
s, n = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        if "pa::" not in row["Kernel_Name"]:
            continue
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        s[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]] += 1
for k in sorted(s):
    print(k)
    for c in sorted(s[k]):
        print(f"    {c:24s} {s[k][c] / n[k][c]:14.1f}  ({n[k][c]} launches)")
PY

"""Condenses rocprofv3 CSV output (tools/collect_profiles.sh) into the files kept under profiles/:
   <tag>_kernel_stats*.csv      rocprofv3 --stats kernel summaries (copied; c2 with two stream groups and one, c1, c4)
   <tag>_pmc_per_launch.csv     average counter value per launch and kernel
   counters.json                what bench.py prints as roofline.traffic / limiter counts: HBM bytes per sweep of the
                                two sweep launches (FETCH_SIZE / WRITE_SIZE), wave-instruction counts, and the commit
                                they were collected at (profiles/collect_head.txt, written by tools/run_collect.sh)
Usage: summarize_profiles.py gpurun_out/prof_<tag> <tag>"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")

stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    print("kernel stats:", stats[0])
for sub in ("1stream", "c1", "c4_houdayer", "c4_jorg", "c4_cmr", "c4_houdayer_sw", "c4_jorg_sw"):
    found = glob.glob(os.path.join(src, f"stats_{sub}", "**", "*kernel_stats.csv"), recursive=True)
    if found:
        shutil.copy(found[0], os.path.join(dst, f"{tag}_kernel_stats_{sub}.csv"))

sums = defaultdict(lambda: defaultdict(float))
counts = defaultdict(lambda: defaultdict(int))
for path in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            if "pa::" not in name:
                continue
            short = name.split("(")[0].replace("void ", "").replace(",", ";")
            sums[short][row["Counter_Name"]] += float(row["Counter_Value"])
            counts[short][row["Counter_Name"]] += 1
# the other configs' counter passes (pmcx_<config>_<counters>): one table per config
for cfg in ("c1", "c4_houdayer", "c4_jorg"):
    xs, xn = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    for path in glob.glob(os.path.join(src, f"pmcx_{cfg}_*", "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if "pa::" not in name:
                    continue
                short = name.split("(")[0].replace("void ", "").replace(",", ";")
                xs[short][row["Counter_Name"]] += float(row["Counter_Value"])
                xn[short][row["Counter_Name"]] += 1
    if xs:
        cols = sorted({c for k in xs for c in xs[k]})
        with open(os.path.join(dst, f"{tag}_pmc_{cfg}.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["kernel"] + cols)
            for kernel in sorted(xs):
                w.writerow([kernel] + [f"{xs[kernel][c] / xn[kernel][c]:.1f}" if xn[kernel][c] else "" for c in cols])
        print(f"pmc {cfg}:", len(xs), "kernels")
counters = sorted({c for k in sums for c in sums[k]})
rows = []
for kernel in sorted(sums):
    rows.append([kernel] + [f"{sums[kernel][c] / counts[kernel][c]:.1f}" if counts[kernel][c] else "" for c in counters])
with open(os.path.join(dst, f"{tag}_pmc_per_launch.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel"] + counters)
    w.writerows(rows)
print("pmc per launch:", len(rows), "kernels,", counters)


def steady(prefixes):
    """The template instance of a kernel family that ran most often: the steady-state kernel of the run (a run also has a
    couple of prologue launches of other instances -- k_rng_relay<..., 0, ...> before the first tempering event -- whose
    counts must not be averaged into it)."""
    best, launches = None, 0
    for kernel in sums:
        if any(kernel.startswith(prefix) for prefix in prefixes):
            n = max(counts[kernel].values())
            if n > launches:
                best, launches = kernel, n
    return best


def value(kernel, counter):
    return sums[kernel][counter] / counts[kernel][counter] if kernel and counts[kernel][counter] else None


SWEEP = steady(("pa::k_sweep_hc<", "pa::k_sweep<"))     # the on-chip sweep of this build, whichever variant ran
RNG = steady(("pa::k_rng_relay<", "pa::k_rng_overlap"))  # the random-stream launch with its riders
fetch = [value(RNG, "FETCH_SIZE"), value(SWEEP, "FETCH_SIZE")]
write = [value(RNG, "WRITE_SIZE"), value(SWEEP, "WRITE_SIZE")]
insts = {"k_rng": value(RNG, "SQ_INSTS_VALU"), "k_sweep": value(SWEEP, "SQ_INSTS_VALU")}
head_file = os.path.join(dst, "collect_head.txt")
commit = open(head_file).read().strip() if os.path.exists(head_file) else "unlabelled"
if all(v is not None for v in fetch + write):
    fetch_kb, write_kb = sum(fetch), sum(write)
    sys.path.insert(0, root)
    from peapods_amd import build as _build

    counters_out = {
        "commit": commit,
        "source_hash": _build.source_hash(),  # the kernel sources these counts belong to (bench.py compares it with pa_version)
        "workload": "bench.py --config c2, 128 realisations per launch, one stream group (PEAPODS_STREAMS=1)",
        "wave_insts_per_launch": {k: v for k, v in insts.items() if v is not None},
        "steady_kernels": {"k_rng": RNG, "k_sweep": SWEEP},  # one row each of the per-launch table: no averaging over instances
        "wave_insts_per_launch_by_kernel": {k: value(k, "SQ_INSTS_VALU") for k in sorted(sums) if value(k, "SQ_INSTS_VALU")},
        "lds_bank_conflict_ratio_by_kernel": {k: value(k, "SQ_LDS_BANK_CONFLICT") / value(k, "SQ_LDS_IDX_ACTIVE") for k in sorted(sums)
                                              if value(k, "SQ_LDS_IDX_ACTIVE")},
        "hbm_bytes_per_sweep": (2 * fetch_kb + write_kb) * 1024,
        "kernels": "k_rng_relay + k_sweep_hc: the two launches of one Metropolis sweep of 50 331 648 attempts (the first "
                   "also carries the previous sweep's overlap measurement, statistics and tempering event)",
        "fetch_size_kb_raw": fetch_kb,
        "fetch_size_kb_by_kernel": {"k_rng": fetch[0], "k_sweep": fetch[1]},
        "write_size_kb": write_kb,
        "correction": "MI355X_MICROARCH.md: FETCH_SIZE reports half of a wide coalesced 16 B/lane stream on "
                      "gfx950 -> read side doubled; WRITE_SIZE exact. Separate --pmc passes for FETCH_SIZE "
                      "and WRITE_SIZE.",
        "note": "The doubled figure is an upper bound for this access pattern (part of the reads are narrow "
                "and counted 1:1).",
        "source": f"profiles/{tag}_pmc_per_launch.csv (tools/collect_profiles.sh)",
    }
    with open(os.path.join(dst, "counters.json"), "w") as fh:
        json.dump(counters_out, fh, indent=1)
    print("traffic:", counters_out["hbm_bytes_per_sweep"] / 1e6, "MB per sweep (corrected); collected at", commit)

"""Condenses rocprofv3 CSV output (tools/collect_profiles.sh) into the files kept under profiles/:
   <tag>_kernel_stats.csv       rocprofv3 --stats kernel summary (copied)
   <tag>_pmc_per_launch.csv     average counter value per launch and kernel
   traffic.json                 HBM bytes per sweep of the two sweep launches (FETCH_SIZE / WRITE_SIZE)
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
stats1 = glob.glob(os.path.join(src, "stats_1stream", "**", "*kernel_stats.csv"), recursive=True)
if stats1:
    shutil.copy(stats1[0], os.path.join(dst, f"{tag}_kernel_stats_1stream.csv"))

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
counters = sorted({c for k in sums for c in sums[k]})
rows = []
for kernel in sorted(sums):
    rows.append([kernel] + [f"{sums[kernel][c] / counts[kernel][c]:.1f}" if counts[kernel][c] else "" for c in counters])
with open(os.path.join(dst, f"{tag}_pmc_per_launch.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel"] + counters)
    w.writerows(rows)
print("pmc per launch:", len(rows), "kernels,", counters)


def avg(kernel_prefix, counter):
    tot = n = 0.0
    for kernel in sums:
        if kernel.startswith(kernel_prefix) and counts[kernel][counter]:
            tot += sums[kernel][counter]
            n += counts[kernel][counter]
    return tot / n if n else None


fetch = [avg("pa::k_rng_overlap", "FETCH_SIZE"), avg("pa::k_sweep<", "FETCH_SIZE")]
write = [avg("pa::k_rng_overlap", "WRITE_SIZE"), avg("pa::k_sweep<", "WRITE_SIZE")]
if all(v is not None for v in fetch + write):
    fetch_kb, write_kb = sum(fetch), sum(write)
    traffic = {
        "hbm_bytes_per_sweep": (2 * fetch_kb + write_kb) * 1024,
        "kernels": "k_rng_overlap + k_sweep (one Metropolis sweep of 50 331 648 attempts and the overlap "
                   "measurement of the previous one)",
        "fetch_size_kb_raw": fetch_kb,
        "write_size_kb": write_kb,
        "correction": "MI355X_MICROARCH.md: FETCH_SIZE reports half of a wide coalesced 16 B/lane stream on "
                      "gfx950 -> read side doubled; WRITE_SIZE exact. Separate --pmc passes for FETCH_SIZE "
                      "and WRITE_SIZE.",
        "note": "The doubled figure is an upper bound for this access pattern (part of the reads are narrow "
                "and counted 1:1).",
        "source": f"profiles/{tag}_pmc_per_launch.csv (tools/collect_profiles.sh)",
    }
    with open(os.path.join(dst, "traffic.json"), "w") as fh:
        json.dump(traffic, fh, indent=1)
    print("traffic:", traffic["hbm_bytes_per_sweep"] / 1e6, "MB per sweep (corrected)")

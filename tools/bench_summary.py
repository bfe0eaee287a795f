"""Prints the headline fields of bench.py's JSON line.  Usage: bench_summary.py FILE [TAG]"""
import json
import sys

path = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else ""
with open(path) as fh:
    lines = [l for l in fh.read().strip().splitlines() if l.startswith("{")]
if not lines:
    sys.exit(f"{tag} no JSON line in {path}")
d = json.loads(lines[-1])
km = {k: round(v * 1e3, 1) for k, v in d["roofline"]["kernel_ms"].items()}
print(tag, f"{d['value'] / 1e9:.1f} Gflips/s", f"{d['ms_per_step'] * 1e3:.1f} us/step", "kernels(us):", km)

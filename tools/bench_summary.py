"""Prints the headline fields of bench.py's JSON line (reads stdin)."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
line = [l for l in sys.stdin.read().strip().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
km = {k: round(v * 1e3, 1) for k, v in d["roofline"]["kernel_ms"].items()}
print(tag, f"{d['value'] / 1e9:.1f} Gflips/s", f"{d['ms_per_step'] * 1e3:.1f} us/step", "kernels(us):", km)

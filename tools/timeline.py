#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace (csv) and prints the device timeline of a few steady-state steps:
start/end of every launch relative to the first, its stream / queue, and the idle gaps."""
import csv
import sys

path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
rows = [r for r in rows if "pa::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
take = int(sys.argv[3]) if len(sys.argv) > 3 else 24
t0 = int(rows[skip]["Start_Timestamp"])
busy_end = 0
for r in rows[skip:skip + take]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("void ", "").split("(")[0][:40]
    print(f"{s / 1e3:9.2f} -> {e / 1e3:9.2f} us  ({(e - s) / 1e3:6.2f})  queue {r.get('Queue_Id', '?'):>3}  {name}")
# fraction of wall time with at least one kernel running, over the second half of the trace
events = []
for r in rows[skip:]:
    events.append((int(r["Start_Timestamp"]), 1))
    events.append((int(r["End_Timestamp"]), -1))
events.sort()
depth, last, busy, two = 0, events[0][0], 0, 0
for t, d in events:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        two += t - last
    depth += d
    last = t
span = events[-1][0] - events[0][0]
print(f"span {span / 1e3:.1f} us; >=1 kernel running {busy / span:.1%}; >=2 running {two / span:.1%}")

#!/usr/bin/env python
"""Per-step kernel breakdown from a rocprofv3 --kernel-trace CSV of bench.py.
Steady-state window = between the 3rd and the last dmn::cat_coords launch (one per forward)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
topn = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if "cat_coords" in r["Kernel_Name"]]
lo, hi = marks[2], marks[-1]
nsteps = len(marks) - 3
agg = defaultdict(lambda: [0, 0.0])
durs = defaultdict(list)
busy = 0.0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if lo <= s < hi:
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += (e - s) / 1e6
        durs[r["Kernel_Name"]].append((e - s) / 1e3)
        busy += (e - s) / 1e6
print("steady-state steps: %d ; wall per step %.2f ms ; kernel-busy per step %.2f ms"
      % (nsteps, (hi - lo) / 1e6 / nsteps, busy / nsteps))
for name, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:topn]:
    print("%8.3f ms/step %6.1f calls/step avg %9.1f us  %s" % (ms / nsteps, n / nsteps, ms / n * 1e3, name[:120]))

# One kernel symbol can cover launches of very different sizes (the k=3 scan runs the 5120-step
# mSRU layer, its 256-step tail and the three 20-step language layers).  For the scans, list the
# launches of the LONGEST class (within 2x of the longest) separately: that is the launch
# bench.py's `roofline` object times with HIP events.
for name, d in sorted(durs.items()):
    if "sru_fwd_cols" in name or "mix_sru" in name:
        top = max(d)
        big = [x for x in d if x > top / 2]
        print("longest class of %s: %d launches, avg %.1f us (min %.1f, max %.1f)"
              % (name[:60], len(big), sum(big) / len(big), min(big), max(big)))

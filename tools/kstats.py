#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats capture: per-kernel time per forward.  python tools/kstats.py <dir> [forwards]"""
import csv, glob, sys
d, nf = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print("%9.1f us/fwd %5d calls  avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / nf / 1e3, int(r["Calls"]),
                                                       float(r["AverageNs"]) / 1e3, r["Name"][:110]))
print("total per forward", tot / nf / 1e6, "ms")

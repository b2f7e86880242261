#!/usr/bin/env python
"""per-kernel timeline of one B = 1 generator forward from a rocprofv3 --kernel-trace csv of `tools/b1_gaps.py --run`"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
emb = [i for i, r in enumerate(rows) if "embed_concat" in r["Kernel_Name"]]
a, b = emb[-2], emb[-1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = r.get("Queue_Id", "?")
    grid = r.get("Grid_Size") or "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    wg = r.get("Workgroup_Size") or r.get("Workgroup_Size_X", "?")
    print("%8.1f %8.1f %7.1f q%s grid %14s wg %4s lds %6s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, q, grid, wg,
                                                             r.get("LDS_Block_Size", "?"), r["Kernel_Name"][:64]))

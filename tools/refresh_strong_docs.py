#!/usr/bin/env python
"""Rewrite the strong-scaling prediction table and the sentences quoting it in DESIGN.md / NOTES.md / README.md /
profiles/<round>/README.md from profiles/<round>/strong_model.json (after tools/strong_rehearsal.py)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
m = json.load(open(os.path.join(ROOT, "profiles", R, "strong_model.json")))
p = m["per_n"]
s = open(os.path.join(ROOT, "DESIGN.md")).read()
rows = []
for n in ("1", "2", "4", "8"):
    r = p[n]
    wall = f"{r['predicted_wall_ms']:.1f}" + (f" (measured {m['t1_measured_wall_ms']:.1f})" if n == "1" else "")
    sp, ef = f"{r['predicted_speedup']:.2f}", f"{r['predicted_efficiency']:.2f}"
    if n == "8":
        sp, ef = f"**{sp}**", f"**{ef}**"
    rows.append(f"| {n} | {r['rounds']} | {r['max_solo_compute_ms']:.1f} | {r['sum_solo_compute_ms']:.1f} | {r['load_imbalance']:.3f} | "
                f"{r['modelled_allgather_ms_total']:.1f} | {r['exposed_tail_ms']:.1f} | {wall} | {sp} | {ef} |")
a = re.search(r"^\| 1 \| \d+ \| [0-9.]+ \| [0-9.]+ \| 1\.000 \|", s, re.M).start()
b = s.index("Reading: sharding itself costs")
s = s[:a] + "\n".join(rows) + "\n\n" + s[b:]
r8, r1 = p["8"], p["1"]
d, g, f = r8["delivery_ms_total"], r8["modelled_allgather_ms_total"], r8["round_fractions"][-1]
subs = [
    (r"Reading: sharding itself costs [0-9.]+ % at 8 ranks \(sum of shares [0-9.]+ vs [0-9.]+ ms",
     f"Reading: sharding itself costs {100 * (r8['sum_solo_compute_ms'] / r1['sum_solo_compute_ms'] - 1):.1f} % at 8 ranks (sum of shares {r8['sum_solo_compute_ms']:.1f} vs {r1['sum_solo_compute_ms']:.1f} ms"),
    (r"the loss is the exposed tail, [0-9.]+ ms of [0-9.]+: the last of two rounds holds [0-9.]+ of the payload, so [0-9.]+ of rank 0's delivery \([0-9.]+ ms",
     f"the loss is the exposed tail, {r8['exposed_tail_ms']:.1f} ms of {r8['predicted_wall_ms']:.1f}: the last of two rounds holds {f:.2f} of the payload, so {f:.2f} of rank 0's delivery ({d:.1f} ms"),
    (r"With equal rounds the same capture reads [0-9.]+ ms, without any\noverlap [0-9.]+ ms → [0-9.]+×",
     f"With equal rounds the same capture reads {r8['max_solo_compute_ms'] + (d + g) / 2:.1f} ms, without any\noverlap {r8['max_solo_compute_ms'] + d + g:.1f} ms → {m['t1_measured_wall_ms'] / (r8['max_solo_compute_ms'] + d + g):.1f}×"),
    (r"The measured N = 1 tail \(wall − last kernel done\) is [0-9.]+ ms where the model's is [0-9.]+",
     f"The measured N = 1 tail (wall − last kernel done) is {m['t1_exposed_tail_ms']:.1f} ms where the model's is {r1['exposed_tail_ms']:.1f}"),
    (r"and the predicted N = 1 wall is [0-9.]+ ms off the measured one",
     f"and the predicted N = 1 wall is {abs(m['t1_measured_wall_ms'] - r1['predicted_wall_ms']):.1f} ms off the measured one"),
    (r"\(measured at N = 1, where both run: [0-9.]+ vs [0-9.]+ ms predicted\)",
     f"(measured at N = 1, where both run: {m['t1_measured_wall_ms']:.1f} vs {r1['predicted_wall_ms']:.1f} ms predicted)"),
    (r"[0-9.]+ ms = [0-9.]+× at 8 ranks, efficiency [0-9.]+ \(`profiles/" + R + r"/strong_model.md`",
     f"{r8['predicted_wall_ms']:.1f} ms = {r8['predicted_speedup']:.2f}× at 8 ranks, efficiency {r8['predicted_efficiency']:.2f} (`profiles/{R}/strong_model.md`"),
]
for pat, rep in subs:
    s = re.sub(pat, rep, s)
open(os.path.join(ROOT, "DESIGN.md"), "w").write(s)
quote = f"{r8['predicted_wall_ms']:.1f} ms = {r8['predicted_speedup']:.2f}×"
for fn in ("NOTES.md", os.path.join("profiles", R, "README.md"), "README.md"):
    t = open(os.path.join(ROOT, fn)).read()
    t = re.sub(r"predicted 8-GPU wall [0-9.]+ ms = [0-9.]+×", f"predicted 8-GPU wall {quote}", t)
    t = re.sub(r"\(8 ranks: [0-9.]+ ms, [0-9.]+x, [0-9.]+\)", f"(8 ranks: {r8['predicted_wall_ms']:.1f} ms, {r8['predicted_speedup']:.2f}x, {r8['predicted_efficiency']:.2f})", t)
    open(os.path.join(ROOT, fn), "w").write(t)
print("8 ranks:", quote, "efficiency", r8["predicted_efficiency"])

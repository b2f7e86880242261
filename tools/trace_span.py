#!/usr/bin/env python
"""Wall span per forward from a rocprofv3 --kernel-trace of the BENCHED schedule (multistream=1: the generator's three ResBlock
chains / the encoder's batch parts on parallel streams) -- the check the serial-launch PMC captures cannot give: kernel time per
step <= ms_per_step (round 5 verdict, item 5).
    python tools/trace_span.py gen|enc X_kernel_trace.csv [--json out.json]
A forward starts at its marker kernel (gen: embed_concat_kernel, enc: hubert_lengths_kernel).  Reported per forward (median over the
forwards of the capture, the first two dropped): span = first kernel start -> last kernel end, busy = union of the kernel intervals
(what the GPU was not idle for), sum = sum of kernel durations (> span when streams overlap), period = start -> next start."""
import csv, json, sys

what, path = sys.argv[1], sys.argv[2]
marker = "embed_concat_kernel" if what == "gen" else "hubert_lengths_kernel"
rows = sorted(([int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]] for r in csv.DictReader(open(path))
               if "dissc::" in r["Kernel_Name"]), key=lambda r: r[0])
starts = [i for i, r in enumerate(rows) if marker in r[2]]
# the encoder's batch parts each launch the marker: a forward = the markers closer than 1 ms to each other
fw = []
for i in starts:
    if not fw or rows[i][0] - rows[fw[-1][-1]][0] > 1_000_000:
        fw.append([i])
    else:
        fw[-1].append(i)
firsts = [f[0] for f in fw]
res = []
for k in range(len(firsts) - 1):
    seg = rows[firsts[k]:firsts[k + 1]]
    span = max(r[1] for r in seg) - seg[0][0]
    iv = sorted((r[0], r[1]) for r in seg)
    busy, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    res.append({"kernels": len(seg), "span_us": span / 1e3, "busy_us": busy / 1e3, "sum_us": sum(r[1] - r[0] for r in seg) / 1e3,
                "period_us": (rows[firsts[k + 1]][0] - seg[0][0]) / 1e3})
res = res[2:] or res
med = lambda key: sorted(r[key] for r in res)[len(res) // 2]
out = {"what": what, "forwards": len(res), "kernels_per_forward": res[0]["kernels"], "span_us": round(med("span_us"), 1),
       "busy_us": round(med("busy_us"), 1), "sum_us": round(med("sum_us"), 1), "period_us": round(med("period_us"), 1),
       "schedule": "multistream=1 (the benched schedule), rocprofv3 --kernel-trace only"}
print(json.dumps(out))
if "--json" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)

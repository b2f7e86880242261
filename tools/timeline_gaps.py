#!/usr/bin/env python
"""Idle time of the GPU inside one generator forward from a rocprofv3 --kernel-trace capture taken with the default
(concurrent) launch mode: union of the kernel intervals vs the span of a forward.  python tools/timeline_gaps.py trace.csv [n_fwd]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "dissc::" in r["Kernel_Name"]]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# forwards start with embed_concat_kernel
starts = [i for i, x in enumerate(iv) if "embed_concat" in x[2]]
for fi in range(1, len(starts)):
    seg = iv[starts[fi - 1]:starts[fi]]
    t0, t1 = seg[0][0], max(x[1] for x in seg)
    busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
    gaps = []
    for s, e, _ in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, _))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    ksum = sum(e - s for s, e, _ in seg)
    print(f"forward {fi}: span {(t1 - t0) / 1e6:.3f} ms, union busy {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms "
          f"({len(gaps)} gaps, largest {max(g[0] for g in gaps) / 1e3:.1f} us), sum of kernel durations {ksum / 1e6:.3f} ms")

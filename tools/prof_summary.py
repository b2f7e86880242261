#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV of one bench.py run: per generator layer
(launch order within a forward) duration, algorithmic FLOPs and TFLOP/s.

    python tools/prof_summary.py gpurun_out/prof1/*/*_kernel_trace.csv [--md out.md]
"""
import csv
import sys
from collections import defaultdict

UPS = [(5, 11), (4, 8), (4, 8), (2, 4), (2, 4)]
RK = [3, 7, 11]
DIL = [1, 3, 5]


def layer_list(B=32, T=500):
    """(name, flops) in launch order of dissc_gen_forward (conv kernels only)."""
    out = [("conv_pre 257->512 k7", 2.0 * 512 * 257 * 7 * T * B)]
    ch, L = 512, T
    for i, (s, k) in enumerate(UPS):
        ngrp = {5: 3, 4: 2, 2: 2}[s] if ch // 2 >= 64 else 1  # ConvTranspose = one launch per phase group
        for gi in range(ngrp):
            out.append((f"up{i} convT {ch}->{ch//2} k{k} s{s} g{gi}", 2.0 * ch * (ch // 2) * k * L * B / ngrp))
        ch //= 2
        L *= s
        for rk in RK:
            for m, d in enumerate(DIL):
                out.append((f"s{i} C{ch} k{rk} d{d} conv1", 2.0 * ch * ch * rk * L * B))
                out.append((f"s{i} C{ch} k{rk} d1 conv2", 2.0 * ch * ch * rk * L * B))
    return out


def main():
    path = sys.argv[1]
    rows = [r for r in csv.DictReader(open(path)) if "conv_mfma" in r["Kernel_Name"] or "conv16_stream" in r["Kernel_Name"]]
    layers = layer_list()
    n = len(layers)
    assert len(rows) % n == 0, (len(rows), n)
    reps = len(rows) // n
    dur = defaultdict(list)
    for i, r in enumerate(rows):
        dur[i % n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lines = ["| # | layer | grid | us (median) | GFLOP | TFLOP/s |", "|---|---|---|---|---|---|"]
    tot_us = tot_fl = 0.0
    stage = defaultdict(lambda: [0.0, 0.0])
    for i, (name, fl) in enumerate(layers):
        d = sorted(dur[i])[len(dur[i]) // 2]
        r = rows[i]
        grid = "x".join(str(int(r[f"Grid_Size_{a}"]) // int(r[f"Workgroup_Size_{a}"])) for a in "XYZ")
        lines.append(f"| {i} | {name} | {grid} | {d:.0f} | {fl/1e9:.1f} | {fl/d/1e6:.1f} |")
        tot_us += d
        tot_fl += fl
        key = name.split()[0]
        stage[key][0] += d
        stage[key][1] += fl
    lines.append(f"| | **total** ({reps} forwards traced) | | {tot_us:.0f} | {tot_fl/1e9:.0f} | {tot_fl/tot_us/1e6:.1f} |")
    lines.append("")
    lines.append("| group | us | share | TFLOP/s |")
    lines.append("|---|---|---|---|")
    for k, (d, fl) in stage.items():
        lines.append(f"| {k} | {d:.0f} | {100*d/tot_us:.1f}% | {fl/d/1e6:.1f} |")
    txt = "\n".join(lines)
    print(txt)
    if "--md" in sys.argv:
        open(sys.argv[sys.argv.index("--md") + 1], "w").write(txt + "\n")


if __name__ == "__main__":
    main()

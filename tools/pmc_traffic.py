#!/usr/bin/env python
"""HBM traffic per generator forward from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    python tools/pmc_traffic.py gpurun_out/pmc_fetch/*/*_counter_collection.csv \
                                gpurun_out/pmc_write/*/*_counter_collection.csv

Units per /opt/skills/guides/MI355X_MICROARCH.md section HBM: counter values are KiB; on gfx950
FETCH_SIZE tallies 128-B requests at 64 B, i.e. reports exactly half the bytes of a wide
coalesced read -> doubled here.  WRITE_SIZE is used as reported (uncalibrated).
"""
import csv
import sys
from collections import defaultdict


def per_forward(path, counter, n_fwd=3):
    tot = defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "dissc::" in r["Kernel_Name"]:
            key = "conv_mfma" if "conv_mfma" in r["Kernel_Name"] else r["Kernel_Name"].split("(")[0]
            tot[key] += float(r["Counter_Value"])
    return {k: v / n_fwd for k, v in tot.items()}


def main():
    f = per_forward(sys.argv[1], "FETCH_SIZE")
    w = per_forward(sys.argv[2], "WRITE_SIZE")
    print("| kernel family | FETCH_SIZE KiB/fwd (raw) | read GB/fwd (x2 gfx950 corr.) | WRITE_SIZE KiB/fwd | write GB/fwd |")
    print("|---|---|---|---|---|")
    tr = tw = 0.0
    for k in sorted(set(f) | set(w)):
        rb, wb = 2 * f.get(k, 0) * 1024, w.get(k, 0) * 1024
        tr += rb
        tw += wb
        print(f"| {k} | {f.get(k,0):.0f} | {rb/1e9:.2f} | {w.get(k,0):.0f} | {wb/1e9:.2f} |")
    print(f"| **total** | | {tr/1e9:.2f} | | {tw/1e9:.2f} |")
    print(f"\nHBM bytes per forward (B=32 x T=500): {(tr+tw)/1e9:.2f} GB")


if __name__ == "__main__":
    main()

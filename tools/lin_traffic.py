#!/usr/bin/env python
"""Fabric traffic and time of HuBERT's linears under the XCD order's sweep width (option xcd_mg; lin_gemm.hip launch_lin128_t).
   python tools/lin_traffic.py time MG [MG ...]    sustained time per shape and sweep width (no profiler)
   python tools/lin_traffic.py pmc  MG [MG ...]    six launches per (width, shape) for a rocprofv3 --pmc pass (tools/lin_traffic.sh)
   python tools/lin_traffic.py tab DIR MG [...]    table from DIR/{fetch,write}/*counter_collection.csv of the pmc runs
Dispatch order = (MG, shape) order, 6 launches each: the tables key on it (the shapes share kernel names)."""
import csv, ctypes, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [("qkv 768->2304", 768, 2304, 0), ("out 768->768 +res", 768, 768, 1), ("fc1 768->3072", 768, 3072, 0),
          ("fc2 3072->768 +res", 3072, 768, 1)]
B, T = 32, 499
NL = 6  # launches per (width, shape) in pmc mode: dissc_conv_bench's 2 warm-ups + 4


def alg_bytes(cin, cout, epi):
    ld = (T + 3) // 4 * 4
    return 4.0 * (B * cin * T + cout * cin + B * cout * T * (2 if epi else 1))


def main():
    mode = sys.argv[1]
    if mode == "tab":
        d, mgs = sys.argv[2], [int(v) for v in sys.argv[3:]]
        vals = {}
        for name, scale in (("fetch", 2.0 * 1024.0), ("write", 1024.0)):  # KiB; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM)
            rows = []
            for f in glob.glob(os.path.join(d, name, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "lin128_kernel" in r["Kernel_Name"]:
                        rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            per = {}
            for did, v in rows:  # a counter is reported once per XCD / instance: sum per dispatch
                per[did] = per.get(did, 0.0) + v
            ids = sorted(per)
            assert len(ids) == NL * len(mgs) * len(SHAPES), (name, len(ids))
            for i, did in enumerate(ids):
                vals.setdefault((i // NL, name), []).append(per[did] * scale)
        print("| xcd_mg | " + " | ".join(s[0] for s in SHAPES) + " |   (fabric GB per launch: read + write = total, x algorithmic)")
        print("|---|" + "---|" * len(SHAPES))
        for mi, mg in enumerate(mgs):
            cells = []
            for si, (nm, cin, cout, epi) in enumerate(SHAPES):
                k = mi * len(SHAPES) + si
                rd = sorted(vals[(k, "fetch")][2:])[1] / 1e9
                wr = sorted(vals[(k, "write")][2:])[1] / 1e9
                cells.append(f"{rd:.3f} + {wr:.3f} = {rd + wr:.3f} ({(rd + wr) * 1e9 / alg_bytes(cin, cout, epi):.2f}x)")
            print(f"| {mg} | " + " | ".join(cells) + " |")
        return
    from dissc_amd._lib import lib, check
    mgs = [int(v) for v in sys.argv[2:]]
    ms = ctypes.c_float()
    if mode == "pmc":
        for mg in mgs:
            check(lib.dissc_set_option(b"xcd_mg", mg), "set")
            for nm, cin, cout, epi in SHAPES:
                check(lib.dissc_conv_bench(B, cin, cout, 1, 1, T, epi, NL - 2, 1, ctypes.byref(ms)), "bench")
        return
    check(lib.dissc_conv_bench(B, 768, 3072, 1, 1, T, 0, 3000, 1, ctypes.byref(ms)), "warm")
    best = {}
    for rep in range(2):
        for mg in mgs:
            check(lib.dissc_set_option(b"xcd_mg", mg), "set")
            for nm, cin, cout, epi in SHAPES:
                check(lib.dissc_conv_bench(B, cin, cout, 1, 1, T, epi, 600, 1, ctypes.byref(ms)), "bench")
                best[(mg, nm)] = min(best.get((mg, nm), 1e9), ms.value * 1e3)
    print("| xcd_mg | " + " | ".join(s[0] for s in SHAPES) + " |   (us per launch, sustained, best of 2)")
    print("|---|" + "---|" * len(SHAPES))
    for mg in mgs:
        print(f"| {mg} | " + " | ".join(f"{best[(mg, s[0])]:.1f}" for s in SHAPES) + " |")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Per-layer roofline tables from rocprofv3 captures of the generator (bench.py / tools/opt_ab.py) or the
HuBERT encoder (tools/encode_bench.py), B=32 x 10 s, DISSC_OPTIONS=multistream=0 (serial launches).

    python tools/prof_tables.py gen|enc --trace X_kernel_trace.csv [--sq X_counter_collection.csv]
                                [--fetch F_counter_collection.csv] [--write W_counter_collection.csv]
                                [--md out.md] [--json out.json] [--skip N]

--trace : rocprofv3 --kernel-trace (durations; launches are matched to the layer list by order)
--sq    : a --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (+ SQ_WAIT_ANY SQ_WAIT_INST_ANY
          SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES): MFMA-busy fraction = MFMA_BUSY / (GUI_ACTIVE/8 XCDs x 1024 SIMDs)
--fetch / --write : separate --pmc FETCH_SIZE / WRITE_SIZE passes (KiB; FETCH_SIZE x2 on gfx950, see
          /opt/skills/guides/MI355X_MICROARCH.md section HBM; WRITE_SIZE as reported)
--skip  : forwards at the start of the capture to ignore (warm-up calls)
"""
import argparse
import csv
import json
import sys
from collections import defaultdict

B, T = 32, 500
UPS = [(5, 11), (4, 8), (4, 8), (2, 4), (2, 4)]
RK = [3, 7, 11]
DIL = [1, 3, 5]
PAIR_MAX_C = 32


def gen_layers():
    """(name, flops, algorithmic bytes) of every dissc:: launch of one generator forward, in launch order."""
    f4 = 4.0
    out = [("embed_concat", 0.0, B * 257 * T * f4)]
    out.append(("conv_pre 257->512 k7", 2.0 * 512 * 257 * 7 * T * B, B * (257 + 512) * T * f4))
    ch, L = 512, T
    for i, (s, k) in enumerate(UPS):
        ngrp = {5: 3, 4: 2, 2: 2}[s] if ch // 2 >= 64 else 1
        for gi in range(ngrp):
            out.append((f"up{i} convT {ch}->{ch // 2} k{k} s{s} g{gi}", 2.0 * ch * (ch // 2) * k * L * B / ngrp,
                        B * (ch * L + ch // 2 * L * s / ngrp) * f4))
        ch //= 2
        L *= s
        act = B * ch * L * f4
        for rk in RK:
            for m, d in enumerate(DIL):
                fl = 2.0 * ch * ch * rk * L * B
                last = m == 2
                if ch <= PAIR_MAX_C:  # one launch per residual pair: read x, write y (+ MRF accumulator r/w)
                    out.append((f"s{i} C{ch} k{rk} d{d} pair", 2 * fl, act * (2 + (2 if last and rk != RK[0] else (1 if last else 0)))))
                else:
                    out.append((f"s{i} C{ch} k{rk} d{d} conv1", fl, 2 * act))
                    out.append((f"s{i} C{ch} k{rk} d1 conv2", fl, act * (3 + (2 if last and rk != RK[0] else (1 if last else 0)) - (1 if last else 0))))
    out.append(("conv_post 16->1 k7 + tanh", 2.0 * ch * 7 * L * B, B * (ch + 1) * L * f4))
    return out


def enc_layers():
    """HuBERT-base layer-6 unit encoder, 32 x 160000 samples (a2-a4)."""
    f4 = 4.0
    N = 160000
    fr = [N]
    for k, s in [(10, 5), (3, 2), (3, 2), (3, 2), (3, 2), (2, 2), (2, 2)]:
        fr.append((fr[-1] - k) // s + 1)
    T0, Tn = fr[1], fr[7]
    out = [("lengths", 0, 0), ("conv0 lag moments (GroupNorm stats)", 0, B * N * f4), ("conv0 finalize", 0, 0),
           ("conv0 1->512 k10 s5 + GroupNorm + GELU", 2.0 * 512 * 10 * T0 * B, B * (N + 512 * T0) * f4)]
    for li, k in enumerate([3, 3, 3, 3, 2, 2]):
        tin, tout = fr[li + 1], fr[li + 2]
        out.append((f"conv{li + 1} 512->512 k{k} s2 + GELU", 2.0 * 512 * 512 * k * tout * B, B * 512 * (tin + tout) * f4))
    lin = lambda name, cin, cout, extra=0: (name, 2.0 * cin * cout * Tn * B, B * (cin + cout + extra) * Tn * f4)
    out.append(("LayerNorm(512)", 0, 2 * B * 512 * Tn * f4))
    out.append(lin("post_extract_proj 512->768", 512, 768))
    out.append(("pos_conv k128 g16 + GELU + residual", 2.0 * 768 * 48 * 128 * Tn * B, 3 * B * 768 * Tn * f4))
    out.append(("encoder LayerNorm", 0, 2 * B * 768 * Tn * f4))
    for l in range(6):
        out.append(lin(f"L{l} qkv 768->2304", 768, 2304))
        out.append((f"L{l} attention 12 x 64 (fused)", 4.0 * Tn * Tn * 64 * 12 * B, 4 * B * 768 * Tn * f4))
        out.append(lin(f"L{l} out_proj 768->768 + residual", 768, 768, 768))
        out.append((f"L{l} LayerNorm", 0, 2 * B * 768 * Tn * f4))
        out.append(lin(f"L{l} fc1 768->3072 + GELU", 768, 3072))
        out.append(lin(f"L{l} fc2 3072->768 + residual", 3072, 768, 768))
        out.append((f"L{l} LayerNorm", 0, 2 * B * 768 * Tn * f4))
    # round 5: ONE launch, the bit-exact fp32 fma-chain kernel on the vector ALU (products + argmin), reads the features once
    out.append(("k-means assign 768->100 (fp32 fma chain on the VALU + argmin)", 2.0 * 768 * 100 * Tn * B, B * (768 * f4 + 8) * Tn))
    return out


def dissc_rows(path):
    return [r for r in csv.DictReader(open(path)) if "dissc::" in r["Kernel_Name"]]


def counters_by_dispatch(path):
    """-> list (dispatch order of dissc kernels) of {counter: value}"""
    per = {}
    order = []
    for r in csv.DictReader(open(path)):
        if "dissc::" not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        if d not in per:
            per[d] = {}
            order.append(d)
        per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [per[d] for d in sorted(order)]


def fold(values, n, skip):
    """values per dispatch -> per layer index: list over forwards"""
    vals = values[skip * n:]
    assert len(vals) % n == 0 and vals, (len(values), n, skip)
    out = defaultdict(list)
    for i, v in enumerate(vals):
        out[i % n].append(v)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["gen", "enc"])
    ap.add_argument("--trace", required=True)
    ap.add_argument("--sq")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--md")
    ap.add_argument("--json")
    ap.add_argument("--skip", type=int, default=0)
    a = ap.parse_args()
    layers = gen_layers() if a.what == "gen" else enc_layers()
    n = len(layers)
    rows = dissc_rows(a.trace)
    dur = fold([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows], n, a.skip)
    names = fold([r["Kernel_Name"].split("(")[0].replace("void dissc::", "").replace("dissc::", "") for r in rows], n, a.skip)
    sq = fold(counters_by_dispatch(a.sq), n, a.skip) if a.sq else None
    fe = fold(counters_by_dispatch(a.fetch), n, a.skip) if a.fetch else None
    wr = fold(counters_by_dispatch(a.write), n, a.skip) if a.write else None
    med = lambda xs: sorted(xs)[len(xs) // 2]
    hdr = ["#", "layer", "kernel", "us", "GFLOP", "TFLOP/s", "exec. TFLOP/s", "alg. GB", "alg. GB/s"]
    if sq:
        hdr += ["MFMA busy", "wait_any", "wait_inst", "active"]
    if fe and wr:
        hdr += ["HBM GB (PMC)", "HBM GB/s", "PMC/alg."]
    lines = ["| " + " | ".join(hdr) + " |", "|" + "---|" * len(hdr)]
    tot = defaultdict(float)
    groups = defaultdict(lambda: defaultdict(float))
    for i, (name, fl, by) in enumerate(layers):
        d = med(dur[i])
        # layers that ran in the Toom-Cook transform domain execute 6 ceil(k / 3) / 4 products per output instead of k
        fx = fl
        if "conv_wino_kernel" in names[i][0]:
            import re
            k = int(re.search(r" k(\d+) ", name + " ").group(1))
            fx = fl * 6.0 * ((k + 2) // 3) / (4.0 * k)
        if "respair32_f23_kernel" in names[i][0] or "respair16_f23_kernel" in names[i][0]:  # register-only F(2,3) pairs (k = 11: four sub-filters): 8 products per output
            fx = fl * 8.0 / 11.0
        if "conv_wino8_kernel" in names[i][0]:
            # the eight-point forms: 8 ceil(k / R) / (9 - R) products per output -- R = the instance's sixth template
            # argument (3: F(6,3), 4: F(5,4))
            import re
            k = int(re.search(r" k(\d+) ", name + " ").group(1))
            targs = [int(v) for v in re.search(r"conv_wino8_kernel<([^>]*)>", names[i][0]).group(1).split(",")]
            R = targs[5] if len(targs) > 5 else 3
            fx = fl * 8.0 * ((k + R - 1) // R) / ((9.0 - R) * k)
        row = [str(i), name, names[i][0][:44], f"{d:.0f}", f"{fl / 1e9:.1f}", f"{fl / d / 1e6:.1f}" if fl else "-",
               f"{fx / d / 1e6:.1f}" if fl else "-", f"{by / 1e9:.3f}", f"{by / d / 1e3:.0f}"]
        g = name.split()[0] if a.what == "gen" else ("attention" if "attention" in name else "linear" if ("->" in name and "conv" not in name)
                                                      else "feature conv" if name.startswith("conv") and "conv0" not in name else "other")
        groups[g]["us"] += d
        groups[g]["fl"] += fl
        groups[g]["fx"] += fx
        tot["us"] += d
        tot["fl"] += fl
        tot["fx"] += fx
        tot["by"] += by
        if sq:
            c = {k: med([x.get(k, 0.0) for x in sq[i]]) for k in sq[i][0]}
            simd_cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 1024.0
            busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles if simd_cycles else 0.0
            wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            row += [f"{busy:.2f}", f"{c.get('SQ_WAIT_ANY', 0) / wc:.2f}", f"{c.get('SQ_WAIT_INST_ANY', 0) / wc:.2f}",
                    f"{c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f}"]
            groups[g]["mfma"] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            groups[g]["simd"] += simd_cycles
            tot["mfma"] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            tot["simd"] += simd_cycles
        if fe and wr:
            rb = 2.0 * 1024.0 * med([x.get("FETCH_SIZE", 0.0) for x in fe[i]])
            wb = 1024.0 * med([x.get("WRITE_SIZE", 0.0) for x in wr[i]])
            row += [f"{(rb + wb) / 1e9:.3f}", f"{(rb + wb) / d / 1e3:.0f}", f"{(rb + wb) / by:.2f}" if by else "-"]
            tot["rb"] += rb
            tot["wb"] += wb
            groups[g]["hbm"] += rb + wb
        lines.append("| " + " | ".join(row) + " |")
    lines.append("")
    lines.append(f"**total**: {tot['us']:.0f} us per forward (serial launches), {tot['fl'] / 1e9:.0f} GFLOP algorithmic -> "
                 f"{tot['fl'] / tot['us'] / 1e6:.1f} TFLOP/s; {tot['fx'] / 1e9:.0f} GFLOP executed on the matrix pipe -> "
                 f"{tot['fx'] / tot['us'] / 1e6:.1f} TFLOP/s = {tot['fx'] / tot['us'] / 1e6 / 157.3:.3f} of the fp32 MFMA peak (157.3)"
                 + (f"; MFMA busy {tot['mfma'] / tot['simd']:.2f} of SIMD cycles" if sq else "")
                 + (f"; HBM traffic (PMC) {(tot['rb'] + tot['wb']) / 1e9:.2f} GB = read {tot['rb'] / 1e9:.2f} + write {tot['wb'] / 1e9:.2f} "
                    f"-> {(tot['rb'] + tot['wb']) / tot['us'] / 1e3:.0f} GB/s = {(tot['rb'] + tot['wb']) / tot['us'] / 1e3 / 8000:.3f} of 8 TB/s; "
                    f"algorithmic per-layer bytes {tot['by'] / 1e9:.2f} GB" if fe and wr else ""))
    lines.append("")
    gh = ["group", "us", "share", "TFLOP/s", "exec. TFLOP/s"] + (["MFMA busy"] if sq else []) + (["HBM GB", "GB/s"] if fe and wr else [])
    lines += ["| " + " | ".join(gh) + " |", "|" + "---|" * len(gh)]
    for g, v in groups.items():
        r = [g, f"{v['us']:.0f}", f"{100 * v['us'] / tot['us']:.1f}%", f"{v['fl'] / v['us'] / 1e6:.1f}", f"{v['fx'] / v['us'] / 1e6:.1f}"]
        if sq:
            r.append(f"{v['mfma'] / v['simd']:.2f}" if v["simd"] else "-")
        if fe and wr:
            r += [f"{v['hbm'] / 1e9:.2f}", f"{v['hbm'] / v['us'] / 1e3:.0f}"]
        lines.append("| " + " | ".join(r) + " |")
    txt = "\n".join(lines)
    print(txt)
    if a.md:
        open(a.md, "w").write(txt + "\n")
    if a.json and fe and wr:
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import kernel_source_hash
        json.dump({"bytes_per_step_B32_T500": tot["rb"] + tot["wb"], "read_bytes": tot["rb"], "write_bytes": tot["wb"],
                   "algorithmic_bytes_per_layer_model": tot["by"], "kernel_us_serial": tot["us"],
                   "flops_algorithmic": tot["fl"], "flops_executed": tot["fx"],
                   "kernel_source_hash": kernel_source_hash(),
                   "kernel_version": "PMC capture (separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH x2 on gfx950, serial "
                                     "launches) of the kernel sources with this hash"}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()

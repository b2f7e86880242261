"""Per kernel (summed over a capture of tools/valu_share.sh): non-MFMA VALU instructions per MFMA instruction, and the VALU-issue
cycles they cost (4 per wave64 instruction) relative to the MFMA-busy cycles.   python tools/valu_share.py <counter_collection.csv>"""
import csv
import re
import sys
from collections import defaultdict

tot = defaultdict(lambda: defaultdict(float))
dur = defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].replace("dissc::", "")
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU":
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("| kernel | us (sum) | VALU non-MFMA / MFMA insts | 4 x non-MFMA VALU / MFMA-busy cycles | VALU-active / MFMA-busy | LDS insts / MFMA | SALU / MFMA |")
print("|---|---|---|---|---|---|---|")
rows = sorted(tot.items(), key=lambda kv: -dur[kv[0]])
for k, c in rows[:40]:
    mf = c["SQ_INSTS_MFMA"]
    if mf <= 0:
        continue
    nv = c["SQ_INSTS_VALU"] - mf
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"]
    print(f"| {k[:58]} | {dur[k]:.0f} | {nv / mf:.2f} | {4 * nv / busy if busy else 0:.3f} | "
          f"{c['SQ_ACTIVE_INST_VALU'] / busy if busy else 0:.3f} | {c['SQ_INSTS_LDS'] / mf:.2f} | {c['SQ_INSTS_SALU'] / mf:.2f} |")

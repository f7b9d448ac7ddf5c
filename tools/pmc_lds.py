"""LDS bank-conflict share per kernel from one rocprofv3 --pmc pass (SQ_LDS_BANK_CONFLICT = extra LDS cycles,
SQ_LDS_IDX_ACTIVE = all LDS-array cycles; MI355X_MICROARCH.md section LDS).   python tools/pmc_lds.py <pmc_dir>"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True))[-1]
agg = {}
for r in csv.DictReader(open(f)):
    a = agg.setdefault(r["Kernel_Name"], {"n": 0})
    a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    a["n"] += 1
rows = []
for k, a in agg.items():
    act, conf = a.get("SQ_LDS_IDX_ACTIVE", 0.0), a.get("SQ_LDS_BANK_CONFLICT", 0.0)
    if act > 0:
        rows.append((act, conf / act, k))
rows.sort(reverse=True)
print("LDS-active cycles (sum) | conflict share | kernel")
for act, share, k in rows[:40]:
    print(f"{act:14.3e}  {100 * share:6.1f} %  {k[:120]}")

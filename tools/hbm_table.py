"""HBM-side GB/s of every non-GEMM kernel of the step: PMC bytes per launch (profiles/rNN_*_pmc_traffic.json, by_kernel) divided by
the un-instrumented average duration (rNN_*_kernel_stats.csv).   python tools/hbm_table.py profiles/r02_a > profiles/r02_a_hbm_kernels.md"""
import csv, json, sys
stem = sys.argv[1]
sep = "" if stem.endswith("/") else "_"            # profiles/r04_a (files r04_a_*.json) or a directory gpurun_out/prof_x/
doc = json.load(open(stem + sep + "pmc_traffic.json"))
tr = doc["by_kernel"]
strip = lambda n: n[5:] if n.startswith("void ") else n
stats = {strip(r["Name"]): (int(r["Calls"]), float(r["AverageNs"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(stem + sep + "kernel_stats.csv"))}
rows = []
for name, e in tr.items():
    n = strip(name)
    if n not in stats or "conv_" in n and "kernel<" in n and ("fwd" in n or "dgrad" in n or "wgrad" in n or "bwd_pair" in n) or "spin_kernel" in n:
        continue
    calls, avg_ns, tot = stats[n]
    rows.append((tot, n, calls, avg_ns / 1e3, e["bytes_per_launch"] / 1e6, e["bytes_per_launch"] / avg_ns))
rows.sort(reverse=True)
short = lambda n: n.replace("(anonymous namespace)::", "").split("(")[0][:64]
print("# HBM-side traffic and achieved bandwidth of the non-GEMM kernels (training step, 8 clips)\n")
print(f"measured at commit `{doc.get('commit')}`\n")
print("PMC bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (separate rocprofv3 --pmc passes), averaged over all launches of a kernel (all pyramid\n"
      "levels); duration = un-instrumented average of the kernel trace.  Peak HBM rate 8 TB/s; most of these launches move < 10 MB and sit on the\n"
      "~5 us launch floor, so the fraction says how far a kernel is from being bandwidth-limited at all (L2 / Infinity-Cache hits do not count).\n")
print("| kernel | launches (profile run) | avg us | MB / launch | GB/s | of 8 TB/s |\n|---|---|---|---|---|---|")
for tot, n, calls, us, mb, gbs in rows[:40]:
    print(f"| `{short(n)}` | {calls} | {us:.1f} | {mb:.2f} | {gbs:.0f} | {gbs / 80:.1f} % |")

"""Per-kernel MFMA utilisation from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) joined with the
kernel durations of an un-instrumented `rocprofv3 --kernel-trace --stats` run of the same command.

    python tools/pmc_mfma.py <pmc_dir> <kernel_stats.csv> <out.json>

SQ_VALU_MFMA_BUSY_CYCLES is per SIMD in shader cycles (64 per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_32x32x16_bf16 --
MI355X_MICROARCH.md) and deterministic per launch, but dispatches run ~2x longer under counter collection, so the
denominator comes from the plain kernel trace:

    mfma_util         = busy_per_launch / (1024 SIMDs x avg_duration x 2.4 GHz)         (the number to quote)
    mfma_util_pmc_run = busy / (GRBM_GUI_ACTIVE / 8 XCDs x 1024)                         (rocprofiler-sdk's MfmaUtil expression;
                                                                                         the CSV sums GUI_ACTIVE over 8 XCDs)
2.4 GHz is the clock at which 1024 SIMDs x 64 flop/cycle gives the 157.3 TFLOP/s fp32-MFMA peak bench.py prices against, so
mfma_util is directly comparable with roofline.frac: the gap between the two is padded / non-algorithmic MFMA work."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(__file__))
from pmc_summary import tag_of, head_commit

SIMDS, XCDS, GHZ = 256 * 4, 8, 2.4


def strip(name):
    return name[5:] if name.startswith("void ") else name


def main():
    f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True))[-1]
    stats = {strip(r["Name"]): (int(r["Calls"]), float(r["AverageNs"])) for r in csv.DictReader(open(sys.argv[2]))}
    per = {}
    for r in csv.DictReader(open(f)):
        d = per.setdefault(r["Dispatch_Id"], {"name": strip(r["Kernel_Name"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    by_kernel = {}
    for d in per.values():
        e = by_kernel.setdefault(d["name"], {"launches": 0, "busy": 0.0, "gui": 0.0})
        e["launches"] += 1
        e["busy"] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        e["gui"] += d.get("GRBM_GUI_ACTIVE", 0.0)
    out_k, tags = {}, {}
    for name, e in by_kernel.items():
        if e["busy"] <= 0 or name not in stats:
            continue
        busy_l, avg_ns = e["busy"] / e["launches"], stats[name][1]
        out_k[name] = {"launches": e["launches"], "mfma_busy_cycles_per_launch": round(busy_l), "avg_duration_us": round(avg_ns / 1e3, 2),
                       "mfma_util": round(busy_l / (SIMDS * avg_ns * GHZ), 4),
                       "mfma_util_pmc_run": round(e["busy"] / (e["gui"] / XCDS * SIMDS), 4)}
        tag = tag_of(name)
        if tag:
            t = tags.setdefault(tag, {"launches": 0, "busy": 0.0, "ns": 0.0, "gui": 0.0})
            t["launches"] += e["launches"]
            t["busy"] += e["busy"]
            t["ns"] += avg_ns * e["launches"]
            t["gui"] += e["gui"]
    out_t = {k: {"launches": v["launches"], "mfma_util": round(v["busy"] / (SIMDS * v["ns"] * GHZ), 4),
                 "mfma_util_pmc_run": round(v["busy"] / (v["gui"] / XCDS * SIMDS), 4)} for k, v in tags.items()}
    tb, tn = sum(v["busy"] for v in tags.values()), sum(v["ns"] for v in tags.values())
    out = {"commit": head_commit(), "note": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES per launch / (1024 SIMDs x un-instrumented avg duration x 2.4 GHz)",
           "all_conv_kernels": round(tb / (SIMDS * tn * GHZ), 4), "by_tag": out_t, "by_kernel": out_k}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out_t.items()):
        print(f"{k:24s} n={v['launches']:5d} mfma_util={100*v['mfma_util']:6.2f}%  (in the PMC run: {100*v['mfma_util_pmc_run']:6.2f}%)")
    print(f"all conv kernels: {100*out['all_conv_kernels']:.2f}%")


if __name__ == "__main__":
    main()

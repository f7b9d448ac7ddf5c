"""Fold two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately) into per-kernel HBM bytes per launch.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json>

Units / corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports both counters in KiB;
on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled; WRITE_SIZE is exact for 16-B/lane stores and
float atomics.  Kernel classes are keyed both by the full rocprof kernel name and by the bench.py roofline tag."""
import csv, glob, json, os, re, sys

def load(d, counter):
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[-1]
    agg = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        a = agg.setdefault(r["Kernel_Name"], [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"]) * 1024.0
    return agg


def head_commit():
    """Commit the measured snapshot was taken from: $AVDF_COMMIT, else .gpurun_commit (written by tools/gpu.sh before the
    snapshot travels to the GPU box, which has no .git), else `git rev-parse` where a repository exists."""
    c = os.environ.get("AVDF_COMMIT")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not c and os.path.exists(os.path.join(root, ".gpurun_commit")):
        c = open(os.path.join(root, ".gpurun_commit")).read().strip()
    if not c:
        import subprocess
        r = subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True)
        c = r.stdout.strip() if r.returncode == 0 else None
    return c or None


def tag_of(name):
    """bench.py roofline class of a rocprof kernel name: conv_{fwd,dgrad,wgrad}_k{K}s{S} (all tile variants folded;
    the grouped q/k/v launches run the same kernels and are folded in too)."""
    if "conv_bwd_pair_k1_kernel" in name:
        return "conv_bwdpair_k1s1"
    m = re.search(r"conv_(fwd|dgrad|wgrad)_kernel<(\d+), (\d+),", name)
    return None if not m else f"conv_{m.group(1)}_k{m.group(2)}s{m.group(3)}"


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    by_kernel, by_tag = {}, {}
    for name in sorted(set(fetch) | set(write)):
        nf, fb = fetch.get(name, (0, 0.0))
        nw, wb = write.get(name, (0, 0.0))
        ent = {"launches": max(nf, nw), "fetch_bytes": round(2.0 * fb / max(nf, 1)), "write_bytes": round(wb / max(nw, 1))}
        ent["bytes_per_launch"] = ent["fetch_bytes"] + ent["write_bytes"]
        by_kernel[name] = ent
        tag = tag_of(name)
        if tag:
            t = by_tag.setdefault(tag, {"launches": 0, "f": 0.0, "w": 0.0})
            t["launches"] += max(nf, nw)
            t["f"] += 2.0 * fb
            t["w"] += wb * (max(nf, nw) / max(nw, 1))
    out_tag = {k: {"launches": v["launches"], "fetch_bytes": round(v["f"] / v["launches"]),
                   "write_bytes": round(v["w"] / v["launches"]),
                   "bytes_per_launch": round((v["f"] + v["w"]) / v["launches"])} for k, v in by_tag.items()}
    # grouped q/k/v launches carry the fwd3 / dgrad3 tags in bench.py: same kernels, 3 problems per launch; they are
    # folded into the plain tags here (rocprof cannot tell them apart)
    json.dump({"commit": head_commit(), "note": "HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), separate rocprofv3 --pmc passes",
               "by_tag": out_tag, "by_kernel": by_kernel}, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out_tag.items()):
        print(f"{k:32s} n={v['launches']:5d} fetch={v['fetch_bytes']/1e6:8.2f} MB write={v['write_bytes']/1e6:8.2f} MB")


if __name__ == "__main__":
    main()

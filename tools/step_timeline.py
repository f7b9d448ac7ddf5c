"""Ordered kernel timeline of ONE graph-replayed training step from a rocprofv3 kernel trace.

    python tools/step_timeline.py <dir-or-kernel_trace.csv> [step_index] > timeline.txt

A step ends with the optimiser launch (`adamw_ema_seg_kernel`); `step_index` picks which delimiter pair is printed
(default: the 10th, well inside the timed graph replays).  Columns: start offset (us), duration (us: in a graph replay
the kernels are back to back, so this includes the dispatch gap), workgroups x threads, LDS bytes, VGPRs, kernel name.
The last lines aggregate the step by kernel family.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "aten::", n)
    return n.split("(")[0][:72]


def main():
    p = sys.argv[1]
    if os.path.isdir(p):
        p = sorted(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True))[-1]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    rows = []
    with open(p) as f:
        for r in csv.DictReader(f):
            wgs = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], grid // max(1, wgs), wgs,
                         int(r["LDS_Block_Size"]), int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"])))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "adamw_ema" in r[2]]
    if len(ends) < which + 2:
        which = max(0, len(ends) - 3)
    a, b = ends[which], ends[which + 1]
    step = rows[a + 1:b + 1]
    t0 = step[0][0]
    fam = defaultdict(lambda: [0, 0.0])
    tot = 0.0
    for s, e, n, wg, ws, lds, vg in step:
        d = (e - s) / 1e3
        tot += d
        k = short(n)
        fam[k][0] += 1
        fam[k][1] += d
        print("%9.1f dur=%7.1f wg=%6d x%4d lds=%6d vgpr=%3d %s" % ((s - t0) / 1e3, d, wg, ws, lds, vg, k))
    print("# launches=%d sum_us=%.1f wall_us=%.1f" % (len(step), tot, (step[-1][1] - t0) / 1e3))
    for k, (c, d) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print("# %4d launches %8.1f us  %s" % (c, d, k))


if __name__ == "__main__":
    main()

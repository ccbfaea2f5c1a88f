#!/usr/bin/env python3
"""Condenses a tools/profile_gpu.sh output directory into a short text summary:
per-kernel time statistics from the kernel trace, and per-kernel means of each PMC counter.
gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts 64 B per 128-B request on
wide reads, so HBM-side read bytes ~= 2 x FETCH_SIZE x 1024 (uncalibrated for 64-B gathers:
both the raw and the doubled figure are printed); WRITE_SIZE x 1024 is exact."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(root, pat):
    return sorted(glob.glob(os.path.join(root, "**", pat), recursive=True))


def main(out):
    print(f"# profile summary of {os.path.basename(out)}")
    for f in find(os.path.join(out, "trace"), "*kernel_stats.csv"):
        print("## kernel stats (rocprofv3 --kernel-trace --stats)")
        for row in csv.DictReader(open(f)):
            name = row.get("Name", "")[:70]
            print(f"{name:70s} calls {row.get('Calls'):>6s} total_ns {row.get('TotalDurationNs'):>12s} "
                  f"avg_ns {row.get('AverageNs'):>12s} min {row.get('MinNs')} max {row.get('MaxNs')} pct {row.get('Percentage')}")
    for f in find(os.path.join(out, "trace"), "*kernel_trace.csv"):
        rows = list(csv.DictReader(open(f)))
        by = defaultdict(list)
        for r in rows:
            by[r["Kernel_Name"][:60]].append(r)
        print("## kernel trace resources")
        for k, rs in by.items():
            r = rs[0]
            print(f"{k:60s} n={len(rs)} VGPR={r.get('VGPR_Count')} accum={r.get('Accum_VGPR_Count')} SGPR={r.get('SGPR_Count')} "
                  f"LDS={r.get('LDS_Block_Size')} scratch={r.get('Scratch_Size')} grid={r.get('Grid_Size_X')} wg={r.get('Workgroup_Size_X')}")
    print("## PMC counters: mean per dispatch, per kernel")
    for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
        if not os.path.isdir(d):
            continue
        for f in find(d, "*counter_collection.csv"):
            acc = defaultdict(lambda: defaultdict(list))
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, cs in acc.items():
                if "trace" not in k and "k_wf" not in k and "k_fold" not in k:
                    continue
                for c, vals in cs.items():
                    m = sum(vals) / len(vals)
                    extra = ""
                    if c == "FETCH_SIZE":
                        extra = f"  => read bytes raw {m * 1024:.4g}, x2-corrected {2 * m * 1024:.4g}"
                    if c == "WRITE_SIZE":
                        extra = f"  => write bytes {m * 1024:.4g}"
                    print(f"{k:40s} {c:32s} mean {m:16.1f} (n={len(vals)}){extra}")


if __name__ == "__main__":
    main(sys.argv[1])

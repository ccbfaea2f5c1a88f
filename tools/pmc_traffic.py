#!/usr/bin/env python3
"""profiles/r02_pmc_traffic.json from a tools/profile_gpu.sh output directory: mean FETCH_SIZE / WRITE_SIZE per launch of
every path-tracer kernel, keyed by the hash of the kernel sources (bench.py quotes it only for the same sources).
gfx950 (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 64 B per 128-B request of a wide coalesced stream (x2), and is
exact for 64-byte gathers; the kernels here mix both, so the raw figure (a lower bound) and the doubled one are kept,
hbm_bytes_per_launch uses the RAW reads + the exact writes.
Usage: pmc_traffic.py gpurun_out/prof_<tag> [more profile directories ...] out.json   (a kernel family is taken from the first
directory that has it: the stage-split pipeline and the persistent kernel are profiled in separate runs)"""
import csv, glob, json, os, sys
from collections import defaultdict
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from bench import source_sha  # noqa: E402

def collect(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # one entry per kernel family: the launches of all its non-instrumented instantiations together (the first template
    # argument is COUNT; k_wf_extend / k_wf_shade also come as a bounce-0 instantiation, part of the same per-launch mean)
    fam = defaultdict(lambda: {"names": [], "FETCH_SIZE": [], "WRITE_SIZE": []})
    for k, cs in acc.items():
        head = k.split("(")[0].replace("void ", "")
        short = head.split("<")[0]
        if not short.startswith("k_") or head.split("<")[-1].split(",")[0].strip() == "true":
            continue
        fam[short]["names"].append(head)
        fam[short]["FETCH_SIZE"] += cs["FETCH_SIZE"]
        fam[short]["WRITE_SIZE"] += cs["WRITE_SIZE"]
    kernels = {}
    for short, cs in fam.items():
        fs = sum(cs["FETCH_SIZE"]) / max(1, len(cs["FETCH_SIZE"])) * 1024
        ws = sum(cs["WRITE_SIZE"]) / max(1, len(cs["WRITE_SIZE"])) * 1024
        kernels[short] = {"instantiations": sorted(cs["names"]), "launches": len(cs["FETCH_SIZE"]), "fetch_bytes_raw": int(fs),
                          "fetch_bytes_x2": int(2 * fs), "write_bytes": int(ws), "hbm_bytes_per_launch": int(fs + ws)}
    return kernels


def main(dirs, out):
    kernels = {}
    for d in dirs:
        for k, v in collect(d).items():
            kernels.setdefault(k, dict(v, profile=os.path.basename(d.rstrip("/"))))
    json.dump({"source_sha": source_sha(), "from": [os.path.basename(d.rstrip("/")) for d in dirs],
               "note": "mean per launch; FETCH_SIZE raw (exact for 64-B gathers, half of a wide coalesced stream: MI355X_MICROARCH.md) "
                       "+ WRITE_SIZE (exact); hbm_bytes_per_launch = raw reads + writes (a lower bound for the streaming kernels: see fetch_bytes_x2)",
               "kernels": kernels}, open(out, "w"), indent=1)
    print(json.dumps(kernels, indent=1))

if __name__ == "__main__":
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    main(sys.argv[1:-1], sys.argv[-1])

#!/usr/bin/env python3
"""profiles/r02_pmc_traffic.json from a tools/profile_gpu.sh output directory: mean FETCH_SIZE / WRITE_SIZE per launch of
every path-tracer kernel, keyed by the hash of the kernel sources (bench.py quotes it only for the same sources).
gfx950 (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 64 B per 128-B request of a wide coalesced stream (x2), and is
exact for 64-byte gathers; the kernels here mix both, so the raw figure (a lower bound) and the doubled one are kept,
hbm_bytes_per_launch uses the RAW reads + the exact writes.  Usage: pmc_traffic.py gpurun_out/prof_<tag> [out.json]"""
import csv, glob, json, os, sys
from collections import defaultdict
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from bench import source_sha  # noqa: E402

def main(d, out):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kernels = {}
    for k, cs in acc.items():
        short = k.split("(")[0].replace("void ", "").split("<")[0]
        if not short.startswith("k_") or "true" in k.split("(")[0]:   # skip the instrumented instantiations
            continue
        fs = sum(cs["FETCH_SIZE"]) / max(1, len(cs["FETCH_SIZE"])) * 1024
        ws = sum(cs["WRITE_SIZE"]) / max(1, len(cs["WRITE_SIZE"])) * 1024
        kernels[short] = {"full_name": k, "launches": len(cs["FETCH_SIZE"]), "fetch_bytes_raw": int(fs), "fetch_bytes_x2": int(2 * fs),
                          "write_bytes": int(ws), "hbm_bytes_per_launch": int(fs + ws)}
    json.dump({"source_sha": source_sha(), "from": os.path.basename(d.rstrip("/")), "kernels": kernels}, open(out, "w"), indent=1)
    print(json.dumps(kernels, indent=1))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r02_pmc_traffic.json"))

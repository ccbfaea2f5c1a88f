#!/usr/bin/env python3
"""Compute-only projection of the tile split: every part of an N-way stripe split of the bench step
is timed ALONE on this one GPU; the slowest part bounds the step of an N-GPU run (no gather, no launch skew).
Usage: SPP=16 python tools/part_bench.py [--rows 8,16,32] [--kernels persistent,wavefront]"""
import argparse
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import gpu_pathtracer_amd as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", default="8,16,32")
ap.add_argument("--kernels", default="persistent,wavefront")
ap.add_argument("--parts", default="1,2,4,8")
a = ap.parse_args()
W, H = 1920, 1080
SPP = int(os.environ.get("SPP", "16"))
bvh = g.Bvh(g.scene_mesh("cornell_dragon_800k"))
pt = g.PathTracer(0)
pt.set_option(g.OPT_REBUILD, 2)     # as bench.py uploads: keep the tree with fewer node visits
pt.upload_bvh(bvh)
pt.upload_spheres(g.reference_spheres())
cam = g.default_camera(W, H)
acc, rgba = pt.alloc_frame(W, H + 64)
print(f"cornell_dragon_800k {W}x{H}, {SPP} spp per call; ms per call of each part alone (min over 3 x 6 calls)")
KERN = {"persistent": g.KERNEL_PERSISTENT, "wavefront": g.KERNEL_WAVEFRONT, "mega": g.KERNEL_MEGA_BVH2}
for kname in a.kernels.split(","):
    pt.set_option(g.OPT_KERNEL, KERN[kname])
    base = None
    for rows in [int(r) for r in a.rows.split(",")]:
        for count in [int(c) for c in a.parts.split(",")]:
            ts = []
            for part in range(count):
                best = 1e9
                for r in range(3):
                    pt.sync()
                    t0 = time.perf_counter()
                    for f in range(6):
                        p = g.default_params(W, H)
                        p.frame, p.sample_index, p.flags = f * SPP, 1 + f * SPP, g.FLAG_WRITE_RGBA
                        p.part_index, p.part_count, p.part_rows = part, count, rows
                        pt.launch_kernel(acc.ptr, rgba.ptr, cam, p, SPP)
                    pt.sync()
                    best = min(best, (time.perf_counter() - t0) / 6 * 1e3)
                ts.append(best)
            if count == 1:
                base = base or max(ts)
            print(f"{kname:10s} rows {rows:3d} parts {count}: slowest part {max(ts):7.3f} ms (fastest {min(ts):7.3f})  "
                  f"speed-up vs 1 part {base / max(ts):5.2f}x", flush=True)

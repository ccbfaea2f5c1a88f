#!/usr/bin/env python3
"""Which TREE the path kernels walk fastest: the uploaded host hierarchy (SBVH port, the reference's flow), the same
triangles re-clustered on the device at upload (PT_OPT_REBUILD), or pt_build_bvh (PLOC / LBVH, with and without
pre-splitting of long triangles).  The closest hit does not depend on the tree (ties go to the smaller triangle id), so
every row renders the same image; the checksum column shows it.
Usage: python tools/tree_ab.py [--scenes cornell_dragon_800k,gto_sixteen] [--spp 16]"""
import argparse
import os
import sys
import time
import zlib

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import gpu_pathtracer_amd as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", default="cornell_dragon_800k")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--presplit", default="0,50,100,200")
ap.add_argument("--leaf-max", default="")
a = ap.parse_args()
W, H = a.width, a.height


def timed(pt, spp, acc, rgba, cam, frames):
    def run(n, first):
        for i in range(n):
            p = g.default_params(W, H)
            p.frame, p.sample_index = (first + i) * spp, 1 + (first + i) * spp
            pt.launch_kernel(acc.ptr, rgba.ptr, cam, p, spp)
    acc.zero()
    run(6, 0)   # PT_KERNEL_AUTO decides in its first calls (four timed trials)
    pt.sync()
    t0 = time.perf_counter()
    run(frames, 3)
    pt.sync()
    return (time.perf_counter() - t0) / frames * 1e3


for scene in a.scenes.split(","):
    mesh = g.scene_mesh(scene)
    t0 = time.perf_counter()
    bvh = g.Bvh(mesh)
    host_s = time.perf_counter() - t0
    closed = scene.startswith("cornell")
    sph = g.reference_spheres() if closed else None
    cam = g.default_camera(W, H)
    if scene == "dragon":
        cam.dist = 18.0
    rows = [("host tree (upload)", dict(upload=True)), ("upload + PT_OPT_REBUILD 1", dict(upload=True, rebuild=1)),
            ("upload + PT_OPT_REBUILD 2 (cheaper)", dict(upload=True, rebuild=2))]
    for lm in a.leaf_max.split(",") if a.leaf_max else ():
        rows.append((f"host tree, PT_OPT_LEAF_MAX {lm}", dict(upload=True, leaf_max=int(lm))))
    for ps in a.presplit.split(","):
        rows.append((f"pt_build_bvh PLOC presplit {ps}", dict(algo=1, presplit=int(ps))))
    for lm in a.leaf_max.split(",") if a.leaf_max else ():
        rows.append((f"pt_build_bvh PLOC, PT_OPT_LEAF_MAX {lm}", dict(algo=1, presplit=0, leaf_max=int(lm))))
    rows.append(("pt_build_bvh LBVH", dict(algo=0, presplit=0)))
    print(f"== {scene}: {len(mesh.tris)} triangles, {W}x{H}, {a.spp} spp per call (host build {host_s:.1f} s)")
    for name, o in rows:
        out = []
        crc = None
        for kernel, kname in ((g.KERNEL_WAVEFRONT, "pipeline"), (g.KERNEL_PERSISTENT, "persistent")):
            pt = g.PathTracer(0)
            pt.set_option(g.OPT_KERNEL, kernel)
            if o.get("upload"):
                pt.set_option(g.OPT_REBUILD, o.get("rebuild", 0))
                if "leaf_max" in o:
                    pt.set_option(g.OPT_LEAF_MAX, o["leaf_max"])
                pt.upload_bvh(bvh)
                b_ms = -1.0
            else:
                pt.set_option(g.OPT_LEAF_MAX, o.get("leaf_max", 2))
                pt.set_option(g.OPT_BUILD_ALGO, o["algo"])
                pt.set_option(g.OPT_PRESPLIT, o["presplit"])
                b_ms = pt.build_bvh(mesh)
            pt.upload_spheres(sph)
            info = pt.scene_info()
            sah = pt.tree_cost()
            acc, rgba = pt.alloc_frame(W, H)
            ms = timed(pt, a.spp, acc, rgba, cam, a.frames)
            img = acc.download(np.float32, (H, W, 3))
            k = zlib.crc32(img.tobytes())
            crc = k if crc is None else (crc if crc == k else -1)
            out.append(f"{kname} {ms:7.3f} ms")
            pt.close()
        print(f"  {name:34s} {info['device_bytes'] / 2 ** 20:7.1f} MB  build {b_ms:6.2f} ms  " + "  ".join(out) + f"  area cost {sah[0]:6.2f} nodes + {sah[1]:5.2f} tris  image crc {crc:#010x}")

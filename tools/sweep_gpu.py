#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants / options in ONE process (cdna guide §5.4 rule 24).
Usage: python tools/sweep_gpu.py [--scene cornell_dragon_800k] [--rounds 5] [--frames 20]"""
import argparse, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import gpu_pathtracer_amd as g

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cornell_dragon_800k")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--frames", type=int, default=20)
ap.add_argument("--mat", type=int, default=0)
ap.add_argument("--spp", type=int, default=1)
ap.add_argument("--no-spheres", action="store_true")
ap.add_argument("--bvh", default="", help="builder overrides, e.g. split_alpha=1e-5,sah_tri_cost=2")
ap.add_argument("--leaf-max", type=int, default=2)
ap.add_argument("--device-build", action="store_true", help="pt_build_bvh (LBVH on the device) instead of the host SAH/SBVH tree")
ap.add_argument("--variants", default="mega::64:8:16:0,mega::0:8:16:1,mega::0:6:16:1,mega::0:4:16:1,persist:16:0:8:16:1,persist:16:0:6:16:1,persist:16:0:4:16:1,persist:8:0:8:16:1,persist:32:0:8:16:1,persist:16:64:8:16:0",
                help="comma list of kernel[:batch[:top_nodes[:occupancy[:lds_stack[:walk[:refill[:vote_node[:vote_rec]]]]]]]]")
a = ap.parse_args()

W, H = a.width, a.height
kw = {}
for item in filter(None, a.bvh.split(",")):
    k, v = item.split("=")
    kw[k] = float(v) if k in ("split_alpha", "sah_tri_cost", "sah_node_cost") else int(v)
pt = g.PathTracer(0)
pt.set_option(g.OPT_LEAF_MAX, a.leaf_max)
if "PT_SPH_LDS" in os.environ:
    pt.set_option(g.OPT_SPHERE_LDS, int(os.environ["PT_SPH_LDS"]))
if "PT_PRESPLIT" in os.environ:
    pt.set_option(g.OPT_PRESPLIT, int(os.environ["PT_PRESPLIT"]))
if a.device_build:
    if "PT_BUILD_ALGO" in os.environ:
        pt.set_option(g.OPT_BUILD_ALGO, int(os.environ["PT_BUILD_ALGO"]))
    m_ = g.scene_mesh(a.scene)
    print("device build ms", [round(pt.build_bvh(m_), 2) for _ in range(3)])
else:
    bvh = g.Bvh(g.scene_mesh(a.scene), **kw)
    print("bvh", kw, bvh.stats)
    if "PT_REBUILD" in os.environ:
        pt.set_option(g.OPT_REBUILD, int(os.environ["PT_REBUILD"]))
    pt.upload_bvh(bvh)
print("leaf_max", a.leaf_max, pt.scene_info())
pt.upload_spheres(None if a.no_spheres else g.reference_spheres())
cam = g.default_camera(W, H)
acc, rgba = pt.alloc_frame(W, H)
variants = []
for v in a.variants.split(","):
    parts = v.split(":") + ["", "", "", "", "", "", "", ""]
    variants.append((v, {"mega": g.KERNEL_MEGA_BVH2, "persist": g.KERNEL_PERSISTENT, "wave": g.KERNEL_WAVEFRONT}[parts[0]],
                     int(parts[1]) if parts[1] else 16, int(parts[2]) if parts[2] else 64,
                     int(parts[3]) if parts[3] else 5, int(parts[4]) if parts[4] else 16, int(parts[5]) if parts[5] else 2, int(parts[6]) if parts[6] else 8,
                     int(parts[7]) if parts[7] else 3, int(parts[8]) if parts[8] else 2))
res = {v[0]: [] for v in variants}
for r in range(a.rounds + 1):
    for name, k, batch, top, occ, lstk, walk, refill, vn, vr in variants:
        pt.set_option(g.OPT_KERNEL, k)
        pt.set_option(g.OPT_WAVE_BATCH if k == g.KERNEL_WAVEFRONT else g.OPT_BATCH, batch)
        pt.set_option(g.OPT_TOP_NODES, top)
        pt.set_option(g.OPT_OCCUPANCY, occ)
        pt.set_option(g.OPT_LDS_STACK, lstk)
        pt.set_option(g.OPT_WALK, walk)
        pt.set_option(g.OPT_REFILL, refill)
        pt.set_option(g.OPT_VOTE_NODE, vn)
        pt.set_option(g.OPT_VOTE_REC, vr)
        pt.sync()
        t0 = time.perf_counter()
        for f in range(a.frames):
            p = g.default_params(W, H, tri_mat=a.mat)
            p.frame, p.sample_index = f * a.spp, 1 + f * a.spp
            p.flags = g.FLAG_WRITE_RGBA
            pt.launch_kernel(acc.ptr, rgba.ptr, cam, p, a.spp)
        pt.sync()
        dt = (time.perf_counter() - t0) / a.frames
        if r > 0:
            res[name].append(dt * 1e3)
print(f"scene {a.scene} {W}x{H} mat {a.mat} spheres {not a.no_spheres}: ms/frame (median, min) and Mrays/s at median (closed-scene bound)")
for name, v in res.items():
    med, mn = float(np.median(v)), float(np.min(v))
    print(f"  {name:26s} median {med:7.3f} ms  min {mn:7.3f} ms   {W * H * 4 * a.spp / med / 1e3:8.1f} Mrays/s")

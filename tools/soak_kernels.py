#!/usr/bin/env python3
"""Soak test: many launches of every frame kernel (and of PT_KERNEL_AUTO's probing) at assorted sizes / spp / partitions /
estimator flags / trees; all must produce the same accumulator every time.  Usage: python tools/soak_kernels.py [--rounds 40]"""
import argparse, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import gpu_pathtracer_amd as g

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=40)
a = ap.parse_args()
rng = np.random.default_rng(1)
meshes = {n: g.scene_mesh(n) for n in ("cornell", "gto_sixteen", "cornell_dragon")}
scenes = {n: g.Bvh(m) for n, m in meshes.items()}
pts = {}
for name, kern in (("persistent", g.KERNEL_PERSISTENT), ("wavefront", g.KERNEL_WAVEFRONT), ("mega", g.KERNEL_MEGA_BVH2), ("auto", g.KERNEL_AUTO)):
    pts[name] = g.PathTracer(0)
    pts[name].set_option(g.OPT_KERNEL, kern)
t0 = time.time()
n_launch = 0
for r in range(a.rounds):
    scene = list(scenes)[r % len(scenes)]
    W, H = int(rng.integers(2, 900)), int(rng.integers(2, 700))
    spp = int(rng.choice([1, 1, 2, 3, 4, 5, 8, 12, 16]))   # 4, 8, 12, 16: sample groups of 4 / 8 / 4 / 16 (PT_OPT_WAVE_SAMPLES)
    mat = int(rng.integers(0, 4))
    spheres = bool(rng.integers(0, 4))
    parts = int(rng.choice([1, 1, 2, 3]))
    part = int(rng.integers(0, parts))
    depth = int(rng.choice([1, 2, 4, 4, 7]))
    flags = int(rng.choice([0, 0, g.FLAG_COSINE_DIFF, g.FLAG_COSINE_DIFF | g.FLAG_NEE, g.FLAGS_SMALLPT, g.FLAGS_CPU_TRACER | g.FLAG_NEE]))
    cam = g.default_camera(max(W, 61), max(H, 61))
    cam.aspect = W / H
    res = {}
    source = ["host tree", "host tree, rebuilt on the device", "device PLOC", "device LBVH", "host tree, optimised at upload",
              "host tree, rebuilt on the device and optimised"][int(rng.integers(0, 6))]
    for name, pt in pts.items():
        # the reference kernel (persistent) always walks the host tree; the others a randomly chosen tree
        src = "host tree" if name == "persistent" else source
        if src == "host tree":
            pt.upload_bvh(scenes[scene])
        elif src == "host tree, rebuilt on the device":
            pt.set_option(g.OPT_REBUILD, 1); pt.upload_bvh(scenes[scene]); pt.set_option(g.OPT_REBUILD, 0)
        elif src == "host tree, optimised at upload":
            pt.set_option(g.OPT_OPTIMIZE, 2); pt.upload_bvh(scenes[scene]); pt.set_option(g.OPT_OPTIMIZE, 0)
        elif src == "host tree, rebuilt on the device and optimised":
            pt.set_option(g.OPT_OPTIMIZE, 2); pt.set_option(g.OPT_REBUILD, 1); pt.upload_bvh(scenes[scene])
            pt.set_option(g.OPT_REBUILD, 0); pt.set_option(g.OPT_OPTIMIZE, 0)
        else:
            pt.set_option(g.OPT_BUILD_ALGO, 1 if src == "device PLOC" else 0)
            pt.build_bvh(meshes[scene])
        pt.upload_spheres(g.reference_spheres() if spheres else None)
        acc, rgba = pt.alloc_frame(W, H + 64)
        for f in range(4):
            p = g.default_params(W, H, depth=depth, tri_mat=mat)
            p.frame, p.sample_index, p.flags = f * spp, 1 + f * spp, g.FLAG_WRITE_RGBA | flags
            p.part_index, p.part_count, p.part_rows = part, parts, 8
            pt.launch_kernel(acc.ptr, rgba.ptr, cam, p, spp)
            n_launch += 1
        pt.sync()
        res[name] = acc.download(np.float32, (H, W, 3))
        acc.free(); rgba.free()
    # another tree may let a grazing candidate through the quantised boxes that the first one culls: <= 2 pixels
    nd = max(int(np.any((res["persistent"] != v) & ~(np.isnan(res["persistent"]) & np.isnan(v)), axis=-1).sum()) for v in res.values())
    ok = nd <= (0 if source == "host tree" else 2)
    print(f"round {r}: {scene} {W}x{H} spp {spp} mat {mat} spheres {spheres} depth {depth} flags {flags:#x} part {part}/{parts}, others on {source}: "
          f"{'same' if nd == 0 else str(nd) + ' pixels differ'}", flush=True)
    if not ok:
        sys.exit(1)
print(f"{n_launch} launches in {time.time() - t0:.1f} s: all kernels agree")

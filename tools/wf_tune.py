#!/usr/bin/env python3
"""Stage times of the stage-split pipeline (PT_KERNEL_WAVEFRONT) under its knobs, and the persistent kernel
beside it, in ONE process on the bench workload.  Also: two contexts on two streams rendering halves of the
samples concurrently (does one context's bandwidth-bound shade stage overlap the other's latency-bound extend?).
Usage: python tools/wf_tune.py [--scene cornell_dragon_800k] [--spp 16] [--what knobs,overlap]"""
import argparse
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import gpu_pathtracer_amd as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cornell_dragon_800k")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--frames", type=int, default=10)
ap.add_argument("--what", default="knobs,overlap")
ap.add_argument("--no-spheres", action="store_true")
a = ap.parse_args()
W, H = a.width, a.height
mesh = g.scene_mesh(a.scene)
bvh = g.Bvh(mesh)
sph = None if a.no_spheres else g.reference_spheres()
cam = g.default_camera(W, H)
if a.scene == "dragon":
    cam.dist = 18.0


def make(kernel):
    pt = g.PathTracer(0)
    pt.set_option(g.OPT_KERNEL, kernel)
    pt.set_option(g.OPT_REBUILD, 2)     # as bench.py uploads
    pt.upload_bvh(bvh)
    pt.upload_spheres(sph)
    return pt


def run(pt, frames, spp, acc, rgba, first=0):
    for i in range(frames):
        p = g.default_params(W, H)
        p.frame, p.sample_index = (first + i) * spp, 1 + i * spp
        pt.launch_kernel(acc.ptr, rgba.ptr, cam, p, spp)


def timed(pt, spp, acc, rgba, frames=None):
    frames = frames or a.frames
    run(pt, 2, spp, acc, rgba)
    pt.sync()
    t0 = time.perf_counter()
    run(pt, frames, spp, acc, rgba)
    pt.sync()
    return (time.perf_counter() - t0) / frames * 1e3


def stages(pt, spp, acc, rgba):
    pt.set_option(g.OPT_TIMING, 1)
    tot = {}
    for i in range(4):
        run(pt, 1, spp, acc, rgba, first=i)
        pt.sync()
        for k, v in pt.stage_ms().items():
            tot[k] = tot.get(k, 0) + v / 4
    pt.set_option(g.OPT_TIMING, 0)
    return {k: round(v, 3) for k, v in tot.items() if v > 0}


if "knobs" in a.what:
    pw = make(g.KERNEL_WAVEFRONT)
    pp = make(g.KERNEL_PERSISTENT)
    acc, rgba = pw.alloc_frame(W, H)
    acc2, rgba2 = pp.alloc_frame(W, H)
    print(f"{a.scene} {W}x{H} spp {a.spp}: persistent {timed(pp, a.spp, acc2, rgba2):.3f} ms/step")
    for lstk in (16, 24) if "stack" in a.what else ():
        pw.set_option(g.OPT_LDS_STACK, lstk)
        for batch in (8, 12, 16, 20, 24, 32, 48):
            pw.set_option(g.OPT_WAVE_BATCH, batch)
            print(f"  wavefront lds_stack {lstk} batch {batch:2d}: {timed(pw, a.spp, acc, rgba):7.3f} ms/step  {stages(pw, a.spp, acc, rgba)}")
    pw.set_option(g.OPT_LDS_STACK, 16)
    pw.set_option(g.OPT_WAVE_BATCH, 16)
    for walk in (2, 4):
        pw.set_option(g.OPT_WALK, walk)
        for top in ((0, 16, 32, 48, 64) if walk == 2 else (0,)):
            pw.set_option(g.OPT_TOP_NODES, top)
            for batch in (16, 24):
                pw.set_option(g.OPT_WAVE_BATCH, batch)
                print(f"  wavefront walk {walk} top {top:2d} batch {batch}: {timed(pw, a.spp, acc, rgba):7.3f} ms/step  {stages(pw, a.spp, acc, rgba)}")
    pw.set_option(g.OPT_WALK, 2)
    pw.set_option(g.OPT_TOP_NODES, 64)
    pw.set_option(g.OPT_WAVE_BATCH, 16)
    for blocks in (8, 6, 4) if "grid" in a.what else ():
        pw.set_option(g.OPT_WAVE_BLOCKS, blocks)
        print(f"  wavefront extend grid {blocks} blocks/CU: {timed(pw, a.spp, acc, rgba):7.3f} ms/step  {stages(pw, a.spp, acc, rgba)}")
    pw.set_option(g.OPT_WAVE_BLOCKS, 8)
    for spp in (1, 2, 4, 8):
        print(f"  spp {spp}: wavefront {timed(pw, spp, acc, rgba, 30):7.3f} ms/call  persistent {timed(pp, spp, acc2, rgba2, 30):7.3f} ms/call")
    pw.close()
    pp.close()

if "overlap" in a.what:
    # two contexts, each its own stream, each folding half of the samples of a step
    half = max(1, a.spp // 2)
    for blocks in (8, 6, 5, 4):
        pts = [make(g.KERNEL_WAVEFRONT) for _ in range(2)]
        bufs = [p.alloc_frame(W, H) for p in pts]
        for p in pts:
            p.set_option(g.OPT_WAVE_BLOCKS, blocks)
        for rep in range(2):
            for p in pts:
                p.sync()
            t0 = time.perf_counter()
            for i in range(a.frames):
                for k, p in enumerate(pts):
                    q = g.default_params(W, H)
                    q.frame, q.sample_index = i * a.spp + k * half, 1 + i * half
                    p.launch_kernel(bufs[k][0].ptr, bufs[k][1].ptr, cam, q, half)
            for p in pts:
                p.sync()
            dt = (time.perf_counter() - t0) / a.frames * 1e3
        print(f"  two contexts x {half} spp, extend grid {blocks} blocks/CU each: {dt:7.3f} ms per {2 * half} spp")
        for p in pts:
            p.close()

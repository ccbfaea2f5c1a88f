#!/usr/bin/env python3
"""One sample per call with pt_sync before every call (the reference host's loop, BasicScene.cpp:395-404) and without, for
PT_OPT_OVERLAP on / off and both stage layouts: ms per call and Mrays/s on the bench scene.  Usage: sync_1spp.py [n_calls]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import gpu_pathtracer_amd as g
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
W, H = 1920, 1080
pt = g.PathTracer(0)
pt.set_option(g.OPT_REBUILD, 2); pt.upload_bvh(g.Bvh(g.scene_mesh("cornell_dragon_800k"))); pt.set_option(g.OPT_REBUILD, 0)
pt.upload_spheres(g.reference_spheres())
cam = g.default_camera(W, H); acc, rgba = pt.alloc_frame(W, H)
def run(n, sync_each):
    for f in range(n):
        p = g.default_params(W, H); p.frame, p.sample_index = f, 1 + f; p.flags = g.FLAG_WRITE_RGBA
        if sync_each:
            pt.sync()
        pt.launch_kernel(acc.ptr, rgba.ptr, cam, p, 1)
    pt.sync()
for kname, kern in (("auto", g.KERNEL_AUTO), ("persistent", g.KERNEL_PERSISTENT), ("wavefront", g.KERNEL_WAVEFRONT)):
    for overlap in (1, 0):
        pt.set_option(g.OPT_KERNEL, kern); pt.set_option(g.OPT_OVERLAP, overlap)
        for sync_each in (True, False):
            run(20, sync_each)
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter(); run(n, sync_each); best = min(best, (time.perf_counter() - t0) / n * 1e3)
            print(f"{kname:10s} overlap {overlap} {'sync before every call' if sync_each else 'calls back to back   '}: {best:.4f} ms per call = {W * H * 4 / best / 1e3:.0f} Mrays/s", flush=True)
pt.close()

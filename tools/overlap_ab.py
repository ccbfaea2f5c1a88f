#!/usr/bin/env python3
"""A/B of PT_OPT_OVERLAP (the path kernel of a call on a side stream, so that it starts while the previous call's last
paths drain) over launch sizes, in one process.  Usage: python tools/overlap_ab.py [--second-context] [--no-rgba]"""
import sys, time, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import gpu_pathtracer_amd as g
W, H = 1920, 1080
bvh = g.Bvh(g.scene_mesh("cornell_dragon_800k"))
other = None
if "--second-context" in sys.argv:
    other = g.PathTracer(0); other.upload_bvh(bvh); other.upload_spheres(g.reference_spheres())
if "--many-streams" in sys.argv:
    import torch
    keep = [torch.cuda.Stream() for _ in range(6)]
    for s_ in keep:
        with torch.cuda.stream(s_):
            torch.zeros(16, device="cuda")
pt = g.PathTracer(0); pt.set_option(g.OPT_KERNEL, g.KERNEL_PERSISTENT)
pt.upload_bvh(bvh); pt.upload_spheres(g.reference_spheres())
cam = g.default_camera(W, H); acc, rgba = pt.alloc_frame(W, H)
def run(spp, parts, n):
    for f in range(n):
        p = g.default_params(W, H); p.frame, p.sample_index = f * spp, 1 + f * spp
        if "--no-rgba" not in sys.argv: p.flags = g.FLAG_WRITE_RGBA
        p.part_index, p.part_count, p.part_rows = (0, parts, 8)
        pt.launch_kernel(acc.ptr, rgba.ptr, cam, p, spp)
for spp, parts, n in ((1, 1, 40), (16, 1, 10), (16, 8, 20)):
    for ovl in (0, 1, 0, 1):
        pt.set_option(g.OPT_OVERLAP, ovl)
        run(spp, parts, 3); pt.sync()
        t0 = time.perf_counter(); run(spp, parts, n); pt.sync()
        print(f"spp {spp} parts {parts} overlap {ovl}: {(time.perf_counter()-t0)/n*1e3:.3f} ms/call", flush=True)

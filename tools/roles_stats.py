#!/usr/bin/env python3
"""Schedule statistics of the role-split kernel (library built with -DPT_ROLES_STATS)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import ctypes as C
import gpu_pathtracer_amd as g
W, H, SPP = 1920, 1080, 4
pt = g.PathTracer(0)
pt.upload_bvh(g.Bvh(g.scene_mesh("cornell_dragon_800k")))
pt.upload_spheres(g.reference_spheres())
pt.set_option(g.OPT_KERNEL, g.KERNEL_WAVEFRONT)
cam = g.default_camera(W, H)
acc, rgba = pt.alloc_frame(W, H)
for batch in (8, 16, 32):
    pt.set_option(g.OPT_ROLES_BATCH, batch)
    # zero the statistics words (pt_get_wave_stats reads counters[6..14])
    pt.set_option(g.OPT_COUNTERS, 1); pt.set_option(g.OPT_KERNEL, g.KERNEL_PERSISTENT)
    p = g.default_params(64, 64); a2, r2 = pt.alloc_frame(64, 64); pt.launch_kernel(a2.ptr, r2.ptr, g.default_camera(64, 64), p, 1); pt.sync()
    pt.set_option(g.OPT_COUNTERS, 0); pt.set_option(g.OPT_KERNEL, g.KERNEL_WAVEFRONT)
    base = pt.wave_stats()
    p = g.default_params(W, H); p.flags = g.FLAG_WRITE_RGBA
    pt.launch_kernel(acc.ptr, rgba.ptr, cam, p, SPP); pt.sync()
    w = pt.wave_stats()
    v = [w[k] - base[k] for k in ("it_node", "act_node", "it_rec", "act_rec", "it_shade", "act_shade", "it_begin", "act_begin")]
    helps = w["it_loop"] - base["it_loop"]
    it, idle, live, sp, sg, sidle, bp, bg = v
    rays = W * H * 4 * SPP
    print(f"batch {batch}: tracer iterations {it} idle {idle} ({100*idle/max(it,1):.1f} %), live lanes per non-idle iteration {live/max(it-idle,1):.1f}; "
          f"shade passes {sp} lanes/pass {sg/max(sp,1):.1f} (segments {sg} of {rays}); shader idle polls {sidle}; begin passes {bp} lanes/pass {bg/max(bp,1):.1f}; shading passes taken by idle tracer waves {helps}")

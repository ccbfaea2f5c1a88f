#!/usr/bin/env python3
"""Times the bench step's stages for several builds of libptmi (tools/build_variant.sh), each in a child
process (PT_LIBPTMI selects the library), interleaved over rounds.  Usage: [PT_AB_REBUILD=2] ab_variants.py tag1 tag2 ... [--kernel 5]"""
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r'''
import sys, os, json
sys.path[:0] = [%r, os.path.join(%r, "tests")]
import gpu_pathtracer_amd as g
W, H, spp = 1920, 1080, 16
pt = g.PathTracer(0); pt.set_option(g.OPT_KERNEL, int(sys.argv[1])); pt.set_option(g.OPT_REBUILD, int(os.environ.get("PT_AB_REBUILD", "0")))
pt.upload_bvh(g.Bvh(g.scene_mesh("cornell_dragon_800k"))); pt.upload_spheres(g.reference_spheres())
cam = g.default_camera(W, H); acc, rgba = pt.alloc_frame(W, H)
pt.set_option(g.OPT_TIMING, 1)
tot = {}
for i in range(8):
    p = g.default_params(W, H); p.frame, p.sample_index = i * spp, 1 + i * spp
    pt.launch_kernel(acc.ptr, rgba.ptr, cam, p, spp); pt.sync()
    if i >= 2:
        for k, v in pt.stage_ms().items(): tot[k] = tot.get(k, 0) + v / 6
print(json.dumps({k: round(v, 3) for k, v in tot.items() if v > 0}))
''' % (ROOT, ROOT)
tags = [t for t in sys.argv[1:] if not t.startswith("--")]
kernel = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else "5"
for rnd in range(2):
    for t in tags:
        env = dict(os.environ)
        if t != "base":
            env["PT_LIBPTMI"] = os.path.join(ROOT, "g.p.u-pathtracer_amd", "csrc", f"libptmi_{t}.so")
        out = subprocess.run([sys.executable, "-c", CHILD, kernel], env=env, capture_output=True, text=True)
        print(rnd, t, out.stdout.strip().splitlines()[-1] if out.returncode == 0 else out.stderr[-300:], flush=True)

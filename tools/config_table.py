#!/usr/bin/env python3
"""Throughput of every BASELINE.json configuration on one GPU: exact segment counts
(instrumented replay of the same frames) over the mean kernel time of the timed launches.
Usage: python tools/config_table.py [--frames 10]"""
import argparse, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import gpu_pathtracer_amd as g

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=10)
ap.add_argument("--only", default="", help="comma-separated substrings: run the configurations whose name holds one of them")
ap.add_argument("--optimize", type=int, default=3, help="PT_OPT_OPTIMIZE passes at upload (0 = the trees of rounds 1-2)")
ap.add_argument("--wave-samples", type=int, default=0, help="PT_OPT_WAVE_SAMPLES (0 = the default; 1 = one sample of a tile per wave)")
a = ap.parse_args()

CONFIGS = [
    # name, scene, W, H, spp per call, material, spheres
    ("C2 cornell 1280x720, sphere room", "cornell", 1280, 720, 16, g.MAT_DIFF, True),
    ("C2 cornell 1280x720, no spheres (open)", "cornell", 1280, 720, 16, g.MAT_DIFF, False),
    ("C3 cornell_dragon 100k 1920x1080 diffuse", "cornell_dragon", 1920, 1080, 16, g.MAT_DIFF, True),
    ("C3 cornell_dragon 800k 1920x1080 diffuse", "cornell_dragon_800k", 1920, 1080, 16, g.MAT_DIFF, True),
    ("C3 cornell_dragon 800k 1920x1080 metal", "cornell_dragon_800k", 1920, 1080, 16, g.MAT_METAL, True),
    ("C3 cornell_dragon 800k 1920x1080 specular", "cornell_dragon_800k", 1920, 1080, 16, g.MAT_SPEC, True),
    ("C3 cornell_dragon 800k 1920x1080 diffuse, 1 spp per call", "cornell_dragon_800k", 1920, 1080, 1, g.MAT_DIFF, True),
    ("C4 gto_sixteen 1920x1080", "gto_sixteen", 1920, 1080, 16, g.MAT_DIFF, True),
    ("C5 dragon 4096x4096 8 spp (open)", "dragon", 4096, 4096, 8, g.MAT_DIFF, False),
    ("C5 dragon 4096x4096 1 spp (open)", "dragon", 4096, 4096, 1, g.MAT_DIFF, False),
    ("C3 cornell_dragon 800k 1920x1080 diffuse, 4 spp per call", "cornell_dragon_800k", 1920, 1080, 4, g.MAT_DIFF, True),
]
print(f"{'configuration':62s} {'kernel':>10s} {'ms/call':>9s} {'Msegments':>10s} {'Mrays/s':>9s}")
for name, scene, W, H, spp, mat, spheres in CONFIGS:
    if a.only and not any(k in name for k in a.only.split(",")):
        continue
    mesh = g.scene_mesh(scene)
    bvh = g.Bvh(mesh)
    for kname, kern in (("persistent", g.KERNEL_PERSISTENT), ("wavefront", g.KERNEL_WAVEFRONT)):
        pt = g.PathTracer(0)
        pt.set_option(g.OPT_KERNEL, kern)
        pt.set_option(g.OPT_REBUILD, 2)      # as bench.py uploads: keep the tree with fewer node visits
        pt.set_option(g.OPT_OPTIMIZE, a.optimize)
        if a.wave_samples:
            pt.set_option(g.OPT_WAVE_SAMPLES, a.wave_samples)
        pt.upload_bvh(bvh)
        pt.upload_spheres(g.reference_spheres() if spheres else None)
        cam = g.default_camera(W, H)
        if scene == "dragon":
            cam.dist = 18.0   # the reference dist = H/60 would start every ray behind the dragon
        acc, rgba = pt.alloc_frame(W, H)
        def launch(f):
            p = g.default_params(W, H, tri_mat=mat)
            p.frame, p.sample_index, p.flags = f * spp, 1 + f * spp, g.FLAG_WRITE_RGBA
            pt.launch_kernel(acc.ptr, rgba.ptr, cam, p, spp)
        for f in range(2):
            launch(f)
        pt.sync()
        ms = 1e30   # best of three timed runs: a 17 ms window (the light configurations) is easily disturbed (r03: 2.0 vs 4.3 ms)
        for rep in range(3):
            t0 = time.perf_counter()
            for f in range(a.frames):
                launch(2 + f)
            pt.sync()
            ms = min(ms, (time.perf_counter() - t0) / a.frames * 1e3)
        # exact segments of the same frames (instrumented persistent kernel: same paths, same counts)
        pt.set_option(g.OPT_KERNEL, g.KERNEL_PERSISTENT)
        pt.set_option(g.OPT_COUNTERS, 1)
        seg = 0
        n_c = min(a.frames, 2)
        for f in range(n_c):
            launch(2 + f)
            pt.sync()
            seg += pt.counters()["rays"]
        seg /= n_c
        print(f"{name:62s} {kname:>10s} {ms:9.3f} {seg / 1e6:10.2f} {seg / ms / 1e3:9.1f}", flush=True)
        acc.free(); rgba.free(); pt.close()

#!/usr/bin/env python3
"""The bench step (or --scene ...) of the stage-split pipeline under a few settings: wall ms per step, the frame's crc, per-ray
counters.  Usage: wf_parts.py [--scene cornell_dragon_800k] [--spp 16] [--device-build | --keep | --no-splits] parts:blocks ...
`parts` > 1 (shade of one region range beside the extend of the next, two streams) and --node-width 8 drive experiments that were
measured and removed (round 3: profiles/r03_wf_parts.txt, r03_wide8_ab.txt; code in commit aada2a3 and its parent): with this
library they are refused; 1:8 is the product."""
import sys, time, zlib, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import gpu_pathtracer_amd as g
args = sys.argv[1:]
scene, spp, dev_build, node_width, occ, no_splits = "cornell_dragon_800k", 16, False, None, None, False
cfgs = []
keep = False
bvh_kw = {}
rebuild = None
optimize = 0
wave_samples = None
while args:
    a = args.pop(0)
    if a == "--scene": scene = args.pop(0)
    elif a == "--spp": spp = int(args.pop(0))
    elif a == "--device-build": dev_build = True
    elif a == "--node-width": node_width = int(args.pop(0))
    elif a == "--occ": occ = int(args.pop(0))
    elif a == "--no-splits": no_splits = True
    elif a == "--keep": keep = True
    elif a == "--bvh":   # host builder parameters: k=v,k=v (pth_build_params fields); implies --keep unless --rebuild N is given
        keep = True
        for kv in args.pop(0).split(","):
            k, v = kv.split("=")
            bvh_kw[k] = float(v) if k in ("split_alpha", "sah_node_cost", "sah_tri_cost") else int(v)
    elif a == "--rebuild": rebuild = int(args.pop(0))
    elif a == "--optimize": optimize = int(args.pop(0))   # PT_OPT_OPTIMIZE passes at upload
    elif a == "--wave-samples": wave_samples = int(args.pop(0))   # PT_OPT_WAVE_SAMPLES (1 = a whole tile of one sample per wave)
    else: cfgs.append(tuple(int(x) for x in a.split(":")))
W, H = 1920, 1080
pt = g.PathTracer(0)
pt.set_option(g.OPT_KERNEL, g.KERNEL_WAVEFRONT)
mesh = g.scene_mesh(scene)
if node_width is not None:   # the 8-wide experiment (commit aada2a3: profiles/r03_wide8_ab.txt); the option is gone with it
    pt.set_option(23, node_width)
if occ is not None:
    pt.set_option(g.OPT_OCCUPANCY, occ)
if wave_samples is not None:
    pt.set_option(g.OPT_WAVE_SAMPLES, wave_samples)
t0 = time.perf_counter()
if dev_build:
    pt.set_option(g.OPT_OPTIMIZE, optimize)
    pt.build_bvh(mesh)
    pt.set_option(g.OPT_OPTIMIZE, 0)
    print(f"device build (+ PT_OPT_OPTIMIZE {optimize}) {time.perf_counter() - t0:.1f} s", flush=True)
else:
    bvh = g.Bvh(mesh, split_alpha=-1.0) if no_splits else g.Bvh(mesh, **bvh_kw)
    t1 = time.perf_counter()
    if bvh_kw:
        print("host tree", bvh_kw, {k: bvh.stats[k] for k in ("n_inner", "n_tri_refs", "max_depth", "sah_cost", "opt_cost_before", "opt_cost_after")}, flush=True)
    pt.set_option(g.OPT_OPTIMIZE, optimize)
    if rebuild is not None:
        pt.set_option(g.OPT_REBUILD, rebuild)
    elif not no_splits and not keep:
        pt.set_option(g.OPT_REBUILD, 2)
    pt.upload_bvh(bvh); pt.set_option(g.OPT_REBUILD, 0)
    print(f"host build {t1 - t0:.1f} s, upload {time.perf_counter() - t1:.1f} s", flush=True)
print("scene", pt.scene_info(), "area cost (node visits, tri tests)", tuple(round(x, 3) for x in pt.tree_cost()), flush=True)
pt.upload_spheres(g.reference_spheres())
cam = g.default_camera(W, H); acc, rgba = pt.alloc_frame(W, H)
def run(n, first=0):
    for f in range(n):
        p = g.default_params(W, H); p.frame, p.sample_index = (first + f) * spp, 1 + (first + f) * spp; p.flags = g.FLAG_WRITE_RGBA
        pt.launch_kernel(acc.ptr, rgba.ptr, cam, p, spp)
for rnd in range(2):
    for parts, blocks in cfgs:
        if parts > 1:
            pt.set_option(22, parts)   # PT_OPT_WAVE_PARTS of the experiment build
        pt.set_option(g.OPT_WAVE_BLOCKS, blocks)
        acc.zero(); run(3); pt.sync()
        t0 = time.perf_counter(); run(10, 3); pt.sync()
        dt = (time.perf_counter() - t0) / 10 * 1e3
        crc = zlib.crc32(acc.download(np.float32, (H, W, 3)).tobytes())
        pt.set_option(g.OPT_TIMING, 1); run(1, 13); pt.sync(); run(1, 13); pt.sync(); st = pt.stage_ms(); pt.set_option(g.OPT_TIMING, 0)
        if rnd == 0:
            pt.set_option(g.OPT_COUNTERS, 1); run(1, 13); pt.sync(); c = pt.counters(); w = pt.wave_stats(); pt.set_option(g.OPT_COUNTERS, 0)
            print(f"   per ray: nodes {c['inner'] / c['rays']:.2f} records {c['tris'] / c['rays']:.2f} leaves {c['leaves'] / c['rays']:.2f}; lane use node {w['act_node'] / max(1, 64 * w['it_node']):.3f} rec {w['act_rec'] / max(1, 64 * w['it_rec']):.3f} overflows/ray {w['stack_overflows'] / c['rays']:.4f}", flush=True)
        print(f"round {rnd} parts {parts} blocks {blocks}: {dt:.3f} ms/step crc {crc:08x} stages " + " ".join(f"{k} {v:.2f}" for k, v in st.items() if v > 0), flush=True)
pt.close()

#!/usr/bin/env python3
"""Lane-occupancy statistics of the persistent wide walk (instrumented launch, PT_OPT_COUNTERS=1):
how many wave-iterations each phase ran and how many of the 64 lanes were active in them.
Usage: python tools/wave_stats.py [--scene cornell_dragon_800k] [--spp 4] [--batch 40] [--refill 8]"""
import argparse, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import gpu_pathtracer_amd as g

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cornell_dragon_800k")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=4)
ap.add_argument("--batch", default="40")
ap.add_argument("--refill", default="8")
ap.add_argument("--walk", default="2")
ap.add_argument("--vote", default="3:2")
a = ap.parse_args()
W, H = a.width, a.height
pt = g.PathTracer(0)
pt.upload_bvh(g.Bvh(g.scene_mesh(a.scene)))
pt.upload_spheres(g.reference_spheres())
cam = g.default_camera(W, H)
acc, rgba = pt.alloc_frame(W, H)
pt.set_option(g.OPT_COUNTERS, 1)
# VALU instructions per wave-iteration of each phase, from the gfx950 listing (make isa)
COST = {"node": 135, "rec": 85, "shade": 700, "begin": 230}
import itertools
for walk, vote, batch, refill in itertools.product([int(x) for x in a.walk.split(",")], a.vote.split(","), [int(x) for x in a.batch.split(",")], [int(x) for x in a.refill.split(",")]):
    if True:
        pt.set_option(g.OPT_WALK, walk)
        pt.set_option(g.OPT_VOTE_NODE, int(vote.split(":")[0]))
        pt.set_option(g.OPT_VOTE_REC, int(vote.split(":")[1]))
        pt.set_option(g.OPT_BATCH, batch)
        pt.set_option(g.OPT_REFILL, refill)
        p = g.default_params(W, H)
        p.frame, p.sample_index, p.flags = 0, 1, g.FLAG_WRITE_RGBA
        pt.launch_kernel(acc.ptr, rgba.ptr, cam, p, a.spp)
        pt.sync()
        c, w = pt.counters(), pt.wave_stats()
        print(f"walk {walk} vote {vote} batch {batch} refill {refill} spp {a.spp}: rays {c['rays']} inner/ray {c['inner'] / c['rays']:.2f} tris/ray {c['tris'] / c['rays']:.2f}")
        tot_w = tot_l = 0.0
        for ph in ("node", "rec", "shade", "begin"):
            it, act = w["it_" + ph], w["act_" + ph]
            tot_w += it * COST[ph] * 64
            tot_l += act * COST[ph]
            print(f"  {ph:6s} wave-iterations {it:12d}  lanes/iteration {act / max(it, 1):6.2f}  ({100 * act / max(it, 1) / 64:5.1f} %)  per ray: {it * 64 / c['rays']:.2f} slots, {act / c['rays']:.2f} used")
        sys.stdout.flush()
        print(f"  outer-loop iterations {w['it_loop']}; cost-weighted lane use {100 * tot_l / tot_w:.1f} % (weights {COST})")

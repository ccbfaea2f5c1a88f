#!/usr/bin/env python3
"""Runs pt_build_bvh on the bench mesh a few times (for rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import gpu_pathtracer_amd as g
mesh = g.scene_mesh(sys.argv[1] if len(sys.argv) > 1 else "cornell_dragon_800k")
pt = g.PathTracer(0)
print("device build ms:", [round(pt.build_bvh(mesh), 2) for _ in range(6)], pt.scene_info())

"""g.p.u-pathtracer_amd — MI355X-native progressive path tracer (hot path only).

Python host-side mirror of the reference's device-facing surface over the C ABI in
include/ptmi.h (HIP kernels in csrc/).  Import name: `gpu_pathtracer_amd` (the directory
name is not a valid Python identifier; the repo-root shim gpu_pathtracer_amd.py aliases it).
"""
from . import _abi
from ._abi import (Camera, Params, Sphere, Material, Counters, BuildParams, MAT_DIFF, MAT_METAL, MAT_SPEC, MAT_REFR,
                   FLAG_METAL_LITERAL_W, FLAG_WRITE_RGBA, FLAG_FACE_FORWARD, FLAG_COSINE_DIFF, FLAG_GLASS_FIX, FLAG_RUSSIAN_ROULETTE, FLAG_MISS_KEEPS_PATH, FLAG_RR_CPU_TRACER, FLAG_NEE, FLAGS_SMALLPT, FLAGS_CPU_TRACER, KERNEL_AUTO, KERNEL_MEGA_BVH2,
                   KERNEL_PERSISTENT, KERNEL_WAVEFRONT, OPT_KERNEL, OPT_COUNTERS, OPT_TIMING, OPT_BATCH, OPT_TOP_NODES, OPT_OCCUPANCY, OPT_LDS_STACK, OPT_WALK, OPT_LEAF_MAX, OPT_TRI_TEST, OPT_REFILL, OPT_VOTE_NODE, OPT_VOTE_REC, OPT_WAVE_BATCH, OPT_SPHERE_LDS, OPT_BUILD_ALGO, OPT_REBUILD, OPT_PRESPLIT, OPT_WAVE_BLOCKS, OPT_OVERLAP, OPT_OPTIMIZE, OPT_WAVE_SAMPLES)
from .host import (Mesh, Bvh, write_image, save_checkpoint, load_checkpoint, frame_hash, reference_spheres, default_camera, default_params, scene_mesh)
from .tracer import PathTracer, PtError, DeviceBuffer, algorithmic_bytes

__all__ = [n for n in dir() if not n.startswith("_")]

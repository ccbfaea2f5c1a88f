"""ctypes declarations for the two in-tree product libraries.

* csrc/libptmi.so  — HIP kernels + C ABI (include/ptmi.h): the drop-in for
  BasicScene::launchKernel (GpuPathTracer/tracer.cu:405-415).
* host/libpthost.so — CPU scene preparation (host/pthost.h): stands in for the
  reference's loadIndexedTris → BVH → CudaBVH::createCompact chain
  (GpuPathTracer/BasicScene.cpp:281-294).

No fallbacks: if a library is missing or fails to load, importing raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PT_LIBPTMI: another build of the same library (A/B experiments: tools/build_variant.sh); never a fallback
PTMI_PATH = os.environ.get("PT_LIBPTMI") or os.path.join(_HERE, "csrc", "libptmi.so")
PTHOST_PATH = os.path.join(_HERE, "host", "libpthost.so")


class Camera(C.Structure):
    """pt_camera — CamInfo, GpuPathTracer/CpuStructs.hpp:19-28."""
    _fields_ = [("pos", C.c_float * 3), ("front", C.c_float * 3), ("right", C.c_float * 3),
                ("up", C.c_float * 3), ("dist", C.c_float), ("aspect", C.c_float),
                ("fov", C.c_float), ("_pad", C.c_float)]


class Sphere(C.Structure):
    """pt_sphere — Sphere, GpuPathTracer/CommomStructs.hpp:18-39 (44 bytes)."""
    _fields_ = [("pos_rad", C.c_float * 4), ("emi", C.c_float * 3), ("col", C.c_float * 3),
                ("mat", C.c_int32)]


class Material(C.Structure):
    """pt_material — one row of the per-triangle material table (extension, include/ptmi.h)."""
    _fields_ = [("col", C.c_float * 3), ("emi", C.c_float * 3), ("mat", C.c_int32), ("phong_expo", C.c_float)]


class CheckpointInfo(C.Structure):
    """pth_checkpoint_info (host/pthost.h)."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("next_frame", C.c_uint64),
                ("constant_pdf", C.c_uint64), ("scene_tag", C.c_uint64)]


class Params(C.Structure):
    """pt_params — scalar part of kernelInfo, GpuPathTracer/CpuStructs.hpp:45-72."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("depth", C.c_uint32),
                ("cull_backfaces", C.c_int32), ("frame", C.c_uint64), ("sample_index", C.c_uint64),
                ("tri_mat", C.c_int32), ("tri_col", C.c_float * 3), ("tri_emi", C.c_float * 3),
                ("bk_color", C.c_float * 3), ("air_ior", C.c_float), ("glass_ior", C.c_float),
                ("phong_expo", C.c_float), ("flags", C.c_uint32),
                ("part_index", C.c_int32), ("part_count", C.c_int32), ("part_rows", C.c_int32),
                ("_pad", C.c_int32)]


class Counters(C.Structure):
    """pt_counters — N_* of the algorithmic-byte definition."""
    _fields_ = [("rays", C.c_uint64), ("inner", C.c_uint64), ("tris", C.c_uint64),
                ("leaves", C.c_uint64), ("hits", C.c_uint64), ("paths", C.c_uint64)]


class BuildParams(C.Structure):
    _fields_ = [("max_leaf_size", C.c_int32), ("min_leaf_size", C.c_int32), ("max_depth", C.c_int32),
                ("n_bins", C.c_int32), ("sah_node_cost", C.c_float), ("sah_tri_cost", C.c_float),
                ("split_alpha", C.c_float), ("n_spatial_bins", C.c_int32), ("optimize_passes", C.c_int32)]


class BvhStats(C.Structure):
    _fields_ = [("n_inner", C.c_uint64), ("n_leaves", C.c_uint64), ("n_tri_refs", C.c_uint64),
                ("max_depth", C.c_uint32), ("sah_cost", C.c_float), ("build_ms", C.c_double),
                ("opt_cost_before", C.c_float), ("opt_cost_after", C.c_float)]


MAT_DIFF, MAT_METAL, MAT_SPEC, MAT_REFR = 0, 1, 2, 3
FLAG_METAL_LITERAL_W = 1 << 0
FLAG_WRITE_RGBA = 1 << 1
FLAG_FACE_FORWARD, FLAG_COSINE_DIFF, FLAG_GLASS_FIX, FLAG_RUSSIAN_ROULETTE, FLAG_MISS_KEEPS_PATH = 1 << 2, 1 << 3, 1 << 4, 1 << 5, 1 << 6
FLAG_RR_CPU_TRACER = 1 << 7
FLAG_NEE = 1 << 8
FLAGS_SMALLPT = FLAG_FACE_FORWARD | FLAG_COSINE_DIFF | FLAG_RUSSIAN_ROULETTE | FLAG_MISS_KEEPS_PATH
FLAGS_CPU_TRACER = FLAG_FACE_FORWARD | FLAG_COSINE_DIFF | FLAG_RR_CPU_TRACER | FLAG_MISS_KEEPS_PATH
KERNEL_AUTO, KERNEL_MEGA_BVH2, KERNEL_PERSISTENT, KERNEL_WAVEFRONT = 0, 1, 3, 5
OPT_KERNEL, OPT_COUNTERS, OPT_TIMING, OPT_BATCH, OPT_TOP_NODES, OPT_OCCUPANCY, OPT_LDS_STACK, OPT_WALK, OPT_LEAF_MAX, OPT_TRI_TEST, OPT_REFILL, OPT_VOTE_NODE, OPT_VOTE_REC, OPT_WAVE_BATCH, OPT_SPHERE_LDS, OPT_BUILD_ALGO, OPT_REBUILD, OPT_PRESPLIT, OPT_WAVE_BLOCKS, OPT_OVERLAP, OPT_OPTIMIZE, OPT_WAVE_SAMPLES = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 21, 24, 25

# every symbol include/ptmi.h declares: (name, restype, argtypes)
_vp, _sz, _i, _u32 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32
PTMI_SYMBOLS = [
    ("pt_abi_version", _i, []),
    ("pt_device_count", _i, []),
    ("pt_create", _i, [_i, C.POINTER(_vp)]),
    ("pt_destroy", _i, [_vp]),
    ("pt_last_error", C.c_char_p, [_vp]),
    ("pt_set_stream", _i, [_vp, _vp]),
    ("pt_set_option", _i, [_vp, _i, _i]),
    ("pt_sync", _i, [_vp]),
    ("pt_malloc", _i, [_vp, _sz, C.POINTER(_vp)]),
    ("pt_free", _i, [_vp, _vp]),
    ("pt_memset", _i, [_vp, _vp, _i, _sz]),
    ("pt_download", _i, [_vp, _vp, _vp, _sz]),
    ("pt_upload", _i, [_vp, _vp, _vp, _sz]),
    ("pt_upload_bvh", _i, [_vp, _vp, _sz, _vp, _sz, _vp, _sz]),
    ("pt_upload_spheres", _i, [_vp, C.POINTER(Sphere), _sz]),
    ("pt_render", _i, [_vp, _vp, _vp, C.POINTER(Camera), C.POINTER(Params), _u32]),
    ("pt_trace_rays", _i, [_vp, _vp, _sz, _i, _vp, _vp, _vp]),
    ("pt_build_bvh", _i, [_vp, _vp, C.c_size_t, _vp, C.c_size_t]),
    ("pt_last_build_ms", _i, [_vp, C.POINTER(C.c_float)]),
    ("pt_upload_tri_materials", _i, [_vp, C.POINTER(Material), C.c_size_t, C.POINTER(C.c_int32), C.c_size_t]),
    ("pt_get_counters", _i, [_vp, C.POINTER(Counters)]),
    ("pt_get_wave_stats", _i, [_vp, C.POINTER(C.c_uint64), _i]),
    ("pt_last_kernel_ms", _i, [_vp, C.POINTER(C.c_float)]),
    ("pt_get_stage_ms", _i, [_vp, C.POINTER(C.c_float), _i]),
    ("pt_auto_choice", _i, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("pt_tree_cost", _i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("pt_scene_info", _i, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                           C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
]

PTHOST_SYMBOLS = [
    ("pth_last_error", C.c_char_p, []),
    ("pth_default_build_params", None, [C.POINTER(BuildParams)]),
    ("pth_mesh_create", _vp, [_vp, _sz, _vp, _sz]),
    ("pth_mesh_load_obj", _vp, [C.c_char_p]),
    ("pth_mesh_load_ptmesh", _vp, [C.c_char_p]),
    ("pth_mesh_append", _i, [_vp, _vp, _vp]),
    ("pth_mesh_n_verts", _sz, [_vp]),
    ("pth_mesh_n_tris", _sz, [_vp]),
    ("pth_mesh_verts", _vp, [_vp]),
    ("pth_mesh_tris", _vp, [_vp]),
    ("pth_mesh_bounds", None, [_vp, C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3)]),
    ("pth_mesh_save_ptmesh", _i, [_vp, C.c_char_p]),
    ("pth_mesh_n_materials", _sz, [_vp]),
    ("pth_mesh_materials", _vp, [_vp]),
    ("pth_mesh_tri_materials", _vp, [_vp]),
    ("pth_mesh_set_materials", _i, [_vp, C.POINTER(Material), _sz, _vp]),
    ("pth_mesh_free", None, [_vp]),
    ("pth_bvh_build", _vp, [_vp, C.POINTER(BuildParams)]),
    ("pth_bvh_nodes", _vp, [_vp]),
    ("pth_bvh_n_node_vec4", _sz, [_vp]),
    ("pth_bvh_tris", _vp, [_vp]),
    ("pth_bvh_n_tri_vec4", _sz, [_vp]),
    ("pth_bvh_index", _vp, [_vp]),
    ("pth_bvh_n_index", _sz, [_vp]),
    ("pth_bvh_get_stats", None, [_vp, C.POINTER(BvhStats)]),
    ("pth_bvh_free", None, [_vp]),
    ("pth_write_ppm", _i, [C.c_char_p, _vp, _i, _i]),
    ("pth_write_png", _i, [C.c_char_p, _vp, _i, _i]),
    ("pth_write_pfm", _i, [C.c_char_p, _vp, _i, _i]),
    ("pth_checkpoint_save", _i, [C.c_char_p, C.POINTER(CheckpointInfo), _vp]),
    ("pth_checkpoint_load", _i, [C.c_char_p, C.POINTER(CheckpointInfo), _vp]),
    ("pth_frame_hash", C.c_uint64, [C.c_uint64]),
]


def _load(path, symbols):
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing - run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU or PyTorch fallback for the path tracer)")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, res, args in symbols:
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


_ptmi = None
_pthost = None


def ptmi():
    global _ptmi
    if _ptmi is None:
        _ptmi = _load(PTMI_PATH, PTMI_SYMBOLS)
    return _ptmi


def pthost():
    global _pthost
    if _pthost is None:
        _pthost = _load(PTHOST_PATH, PTHOST_SYMBOLS)
    return _pthost

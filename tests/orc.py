"""ctypes binding of the CPU oracle (oracle/liborc.so) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module;
nothing under g.p.u-pathtracer_amd/ does (tests/test_host_and_abi.py::test_product_never_touches_the_oracle enforces it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

import gpu_pathtracer_amd as g

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
ORC_DIR = os.path.join(_ROOT, "oracle")
ORC_PATH = os.path.join(ORC_DIR, "liborc.so")
REF_BIN = os.path.join(ORC_DIR, "_ref", "cpuraytracer_core")

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORC_PATH):
            subprocess.check_call(["make", "-C", ORC_DIR, "liborc.so"])
        L = C.CDLL(ORC_PATH)
        vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
        L.orc_wang64.restype = C.c_uint64
        L.orc_wang64.argtypes = [C.c_uint64]
        L.orc_rng_draw.restype = C.c_float
        L.orc_rng_draw.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        L.orc_sincos2pi.restype = None
        L.orc_sincos2pi.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_pow01.restype = C.c_float
        L.orc_pow01.argtypes = [C.c_float, C.c_float]
        L.orc_camera_ray.restype = None
        L.orc_camera_ray.argtypes = [C.POINTER(g.Camera), i32, i32, i32, i32, C.c_float, C.c_float, vp, vp]
        L.orc_accumulate.restype = None
        L.orc_accumulate.argtypes = [vp, vp, vp, C.c_uint64]
        L.orc_trace_rays_bvh.restype = None
        L.orc_trace_rays_bvh.argtypes = [vp, vp, vp, vp, sz, i32, vp, vp, vp, C.POINTER(g.Counters)]
        L.orc_trace_rays_brute.restype = None
        L.orc_trace_rays_brute.argtypes = [vp, vp, sz, vp, sz, i32, vp, vp, vp]
        L.orc_render.restype = i32
        L.orc_render.argtypes = [vp, vp, vp, vp, vp, C.POINTER(g.Sphere), sz, C.POINTER(g.Camera),
                                 C.POINTER(g.Params), C.c_uint32, C.POINTER(g.Counters)]
        L.orc_render_mat.restype = i32
        L.orc_render_mat.argtypes = [vp, vp, vp, vp, vp, C.POINTER(g.Sphere), sz, C.POINTER(g.Material), vp,
                                     C.POINTER(g.Camera), C.POINTER(g.Params), C.c_uint32, C.POINTER(g.Counters)]
        L.orc_set_tri_lights.restype = None
        L.orc_set_tri_lights.argtypes = [vp, sz]
        L.orc_sample_pixels.restype = i32
        L.orc_sample_pixels.argtypes = [vp, vp, vp, vp, vp, sz, C.POINTER(g.Sphere), sz, C.POINTER(g.Camera), C.POINTER(g.Params),
                                        C.c_uint32, vp, sz, vp, vp]
        L.orc_primary_rays.restype = None
        L.orc_primary_rays.argtypes = [C.POINTER(g.Camera), i32, i32, C.c_uint64, i32, vp]
        _lib = L
    return _lib


def counters_dict(c):
    return {f: getattr(c, f) for f, _ in g.Counters._fields_}


def tri_lights(mesh, materials, tri_material):
    """The light list PT_FLAG_NEE uses with a material table: every triangle whose row emits, ascending id, as
    (v0, emi.r) (e1 = v1 - v0, emi.g) (e2 = v2 - v0, emi.b) in binary32 — what the product copies out of its records."""
    tm = np.asarray(tri_material, np.int64)
    emi = np.array([[m.emi[0], m.emi[1], m.emi[2]] for m in materials], np.float32)
    ids = np.nonzero(np.any(emi[tm] != 0, axis=1))[0]
    v = np.asarray(mesh.verts, np.float32)[np.asarray(mesh.tris, np.int64)[ids]]   # [n, 3, 3]
    out = np.zeros((len(ids), 12), np.float32)
    out[:, 0:3] = v[:, 0]
    out[:, 4:7] = v[:, 1] - v[:, 0]
    out[:, 8:11] = v[:, 2] - v[:, 0]
    out[:, 3], out[:, 7], out[:, 11] = emi[tm[ids]].T
    return np.ascontiguousarray(out)


def render(bvh, spheres, cam, params, spp=1, accum=None, want_rgba=True, materials=None, tri_material=None, lights=None):
    """CPU restatement of trace<<<>>> (tracer.cu:343-400).  Returns accum, rgba, counters.
    materials / tri_material: the per-triangle material extension (pt_upload_tri_materials); lights: tri_lights(...)
    for PT_FLAG_NEE over emissive triangles."""
    W, H = params.width, params.height
    if accum is None:
        accum = np.zeros((H, W, 3), np.float32)
    rgba = np.zeros((H, W), np.uint32) if want_rgba else None
    cnt = g.Counters()
    n_s = len(spheres) if spheres is not None else 0
    nodes = bvh.nodes.ctypes.data if bvh is not None else None
    tris = bvh.tris.ctypes.data if bvh is not None else None
    idx = bvh.index.ctypes.data if bvh is not None else None
    if materials is not None and len(materials):
        mtab = (g.Material * len(materials))(*materials)
        ids = np.ascontiguousarray(tri_material, np.int32)
        lib().orc_set_tri_lights(lights.ctypes.data if lights is not None and len(lights) else None, len(lights) if lights is not None else 0)
        rc = lib().orc_render_mat(accum.ctypes.data, rgba.ctypes.data if want_rgba else None, nodes, tris, idx,
                                  spheres if n_s else None, n_s, mtab, ids.ctypes.data, C.byref(cam), C.byref(params),
                                  spp, C.byref(cnt))
        lib().orc_set_tri_lights(None, 0)
    else:
        rc = lib().orc_render(accum.ctypes.data, rgba.ctypes.data if want_rgba else None, nodes, tris, idx,
                              spheres if n_s else None, n_s, C.byref(cam), C.byref(params), spp, C.byref(cnt))
    assert rc == 0
    return accum, rgba, counters_dict(cnt)


def sample_pixels(pixels_xy, spheres, cam, params, spp, bvh=None, mesh=None):
    """Selected pixels sample by sample: the colours before the fold [n][spp][3] and every segment's (t, triangle id)
    [n][spp][depth].  bvh: hits from the oracle's walk over the Compact arrays; mesh (bvh None): hits from the
    BRUTE-FORCE loop over the raw triangles — the arbiter for pixels where two renders differ."""
    px = np.ascontiguousarray(pixels_xy, np.int32).reshape(-1, 2)
    n, depth = len(px), params.depth
    col = np.zeros((n, spp, 3), np.float32)
    seg = np.zeros((n, spp, depth, 2), np.float32)
    n_s = len(spheres) if spheres is not None else 0
    if bvh is not None:
        rc = lib().orc_sample_pixels(bvh.nodes.ctypes.data, bvh.tris.ctypes.data, bvh.index.ctypes.data, None, None, 0,
                                     spheres if n_s else None, n_s, C.byref(cam), C.byref(params), spp, px.ctypes.data, n,
                                     col.ctypes.data, seg.ctypes.data)
    else:
        v, f = np.ascontiguousarray(mesh.verts), np.ascontiguousarray(mesh.tris)
        rc = lib().orc_sample_pixels(None, None, None, v.ctypes.data, f.ctypes.data, len(f), spheres if n_s else None, n_s,
                                     C.byref(cam), C.byref(params), spp, px.ctypes.data, n, col.ctypes.data, seg.ctypes.data)
    assert rc == 0
    return col, seg[..., 0], seg[..., 1].view(np.int32)


def fold_samples(col, first_n, accum=None):
    """The running mean of tracer.cu:386-391 over the sample colours col[..., spp, 3], N = first_n, first_n + 1, ..."""
    col = np.asarray(col, np.float32)
    acc = np.zeros(col.shape[:-2] + (3,), np.float32) if accum is None else np.array(accum, np.float32)
    flat_a, flat_c = acc.reshape(-1, 3), col.reshape(-1, col.shape[-2], 3)
    for i in range(len(flat_a)):
        for s in range(col.shape[-2]):
            lib().orc_accumulate(flat_a[i].ctypes.data, None, flat_c[i, s].ctypes.data, first_n + s)
    return acc


def trace_bvh(bvh, rays, cull=True):
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
    n = len(rays)
    t = np.empty(n, np.float32)
    tri = np.empty(n, np.int32)
    nrm = np.empty((n, 3), np.float32)
    cnt = g.Counters()
    lib().orc_trace_rays_bvh(bvh.nodes.ctypes.data, bvh.tris.ctypes.data, bvh.index.ctypes.data, rays.ctypes.data, n,
                             int(cull), t.ctypes.data, tri.ctypes.data, nrm.ctypes.data, C.byref(cnt))
    return t, tri, nrm, counters_dict(cnt)


def trace_brute(mesh, rays, cull=True):
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
    v, f = np.ascontiguousarray(mesh.verts), np.ascontiguousarray(mesh.tris)
    n = len(rays)
    t = np.empty(n, np.float32)
    tri = np.empty(n, np.int32)
    nrm = np.empty((n, 3), np.float32)
    lib().orc_trace_rays_brute(v.ctypes.data, f.ctypes.data, len(f), rays.ctypes.data, n, int(cull), t.ctypes.data,
                               tri.ctypes.data, nrm.ctypes.data)
    return t, tri, nrm


def primary_rays(cam, W, H, frame=0, jitter=True):
    rays = np.empty((H * W, 8), np.float32)
    lib().orc_primary_rays(C.byref(cam), W, H, frame, int(jitter), rays.ctypes.data)
    return rays


def random_rays(n, lo, hi, seed=1234):
    """Incoherent test rays: origins in an inflated scene box, directions uniform on the sphere."""
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    ext = hi - lo
    o = rng.uniform(lo - 0.25 * ext, hi + 0.25 * ext, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = o
    rays[:, 4:7] = d
    return rays

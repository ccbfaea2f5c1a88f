"""Host-side mirror of the reference's scene set-up (GpuPathTracer/BasicScene.cpp:65-324).

Everything numeric here is DATA restated from the reference constructor so that the
synthetic benchmark inputs equal the reference's defaults:
sphere room BasicScene.cpp:181-202, kernel defaults :220-236, camera :239-259.
Mesh handling and the BVH come from host/libpthost.so.
"""
import ctypes as C
import os

import numpy as np

from . import _abi
from ._abi import (BuildParams, BvhStats, Camera, Material, Params, Sphere, MAT_DIFF, MAT_METAL, MAT_SPEC, MAT_REFR)

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets")


class Mesh:
    """Indexed triangle mesh (SceneMesh, GpuPathTracer/SceneMesh.hpp)."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("pthost: " + _abi.pthost().pth_last_error().decode())
        self._h = C.c_void_p(handle)

    @classmethod
    def from_arrays(cls, verts, tris):
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(tris, np.int32).reshape(-1, 3)
        return cls(_abi.pthost().pth_mesh_create(v.ctypes.data, len(v), t.ctypes.data, len(t)))

    @classmethod
    def load(cls, path):
        lib = _abi.pthost()
        if path.endswith(".ptmesh"):
            return cls(lib.pth_mesh_load_ptmesh(path.encode()))
        return cls(lib.pth_mesh_load_obj(path.encode()))

    @classmethod
    def asset(cls, name):
        """One of the committed mesh fixtures (assets/<name>.ptmesh)."""
        return cls.load(os.path.join(ASSETS, name + ".ptmesh"))

    def append(self, other, xform=None):
        m = None
        if xform is not None:
            m = np.ascontiguousarray(xform, np.float32).reshape(12)
        rc = _abi.pthost().pth_mesh_append(self._h, other._h, m.ctypes.data if m is not None else None)
        if rc != 0:
            raise RuntimeError("pthost: " + _abi.pthost().pth_last_error().decode())
        return self

    @property
    def n_verts(self):
        return _abi.pthost().pth_mesh_n_verts(self._h)

    @property
    def n_tris(self):
        return _abi.pthost().pth_mesh_n_tris(self._h)

    @property
    def verts(self):
        p = _abi.pthost().pth_mesh_verts(self._h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), (self.n_verts, 3)).copy()

    @property
    def tris(self):
        p = _abi.pthost().pth_mesh_tris(self._h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), (self.n_tris, 3)).copy()

    # per-triangle materials (extension; OBJ `usemtl` + .mtl, or PTMESH2 fixtures)
    @property
    def materials(self):
        """List of Material rows ([] when the mesh carries none)."""
        lib = _abi.pthost()
        n = lib.pth_mesh_n_materials(self._h)
        if n == 0:
            return []
        arr = C.cast(lib.pth_mesh_materials(self._h), C.POINTER(Material * n)).contents
        out = []
        for m in arr:
            c = Material()
            C.memmove(C.byref(c), C.byref(m), C.sizeof(Material))
            out.append(c)
        return out

    @property
    def tri_material(self):
        """int32 material row per triangle (None when the mesh carries no materials)."""
        p = _abi.pthost().pth_mesh_tri_materials(self._h)
        if not p:
            return None
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), (self.n_tris,)).copy()

    def set_materials(self, table, tri_material):
        n = len(table) if table is not None else 0
        arr = (Material * n)(*table) if n else None
        ids = np.ascontiguousarray(tri_material, np.int32) if n else None
        if n and len(ids) != self.n_tris:
            raise ValueError("one material row per triangle expected")
        if _abi.pthost().pth_mesh_set_materials(self._h, arr, n, ids.ctypes.data if n else None) != 0:
            raise RuntimeError("pthost: " + _abi.pthost().pth_last_error().decode())
        return self

    def save(self, path):
        if _abi.pthost().pth_mesh_save_ptmesh(self._h, path.encode()) != 0:
            raise RuntimeError("pthost: " + _abi.pthost().pth_last_error().decode())

    def bounds(self):
        lo, hi = (C.c_float * 3)(), (C.c_float * 3)()
        _abi.pthost().pth_mesh_bounds(self._h, C.byref(lo), C.byref(hi))
        return np.array(lo[:], np.float32), np.array(hi[:], np.float32)

    def __del__(self):
        try:
            if self._h:
                _abi.pthost().pth_mesh_free(self._h)
                self._h = None
        except Exception:
            pass


class Bvh:
    """The three arrays of CudaBVH::createCompact (GpuPathTracer/CudaBVH.cpp:121-270)."""

    def __init__(self, mesh, params=None, **kw):
        lib = _abi.pthost()
        bp = BuildParams()
        lib.pth_default_build_params(C.byref(bp))
        if params is not None:
            bp = params
        for k, v in kw.items():
            setattr(bp, k, v)
        h = lib.pth_bvh_build(mesh._h, C.byref(bp))
        if not h:
            raise RuntimeError("pthost: " + lib.pth_last_error().decode())
        h = C.c_void_p(h)
        n = lib.pth_bvh_n_node_vec4(h)
        self.nodes = np.ctypeslib.as_array(C.cast(lib.pth_bvh_nodes(h), C.POINTER(C.c_float)), (n, 4)).copy()
        n = lib.pth_bvh_n_tri_vec4(h)
        self.tris = np.ctypeslib.as_array(C.cast(lib.pth_bvh_tris(h), C.POINTER(C.c_float)), (n, 4)).copy()
        n = lib.pth_bvh_n_index(h)
        self.index = np.ctypeslib.as_array(C.cast(lib.pth_bvh_index(h), C.POINTER(C.c_int32)), (n,)).copy()
        st = BvhStats()
        lib.pth_bvh_get_stats(h, C.byref(st))
        self.stats = {f: getattr(st, f) for f, _ in BvhStats._fields_}
        lib.pth_bvh_free(h)


def _host_call(rc):
    if rc != 0:
        raise RuntimeError("pthost: " + _abi.pthost().pth_last_error().decode())


def write_image(path, frame):
    """.ppm / .png from the display words (uint32 [H][W], 0x00BBGGRR), .pfm from the float
    accumulator ([H][W][3]); rows are bottom-up in both buffers, as the kernel stores them."""
    lib = _abi.pthost()
    if path.endswith(".pfm"):
        a = np.ascontiguousarray(frame, np.float32)
        _host_call(lib.pth_write_pfm(path.encode(), a.ctypes.data, a.shape[1], a.shape[0]))
    else:
        a = np.ascontiguousarray(frame, np.uint32)
        fn = lib.pth_write_png if path.endswith(".png") else lib.pth_write_ppm
        _host_call(fn(path.encode(), a.ctypes.data, a.shape[1], a.shape[0]))


def save_checkpoint(path, accum, next_frame, constant_pdf, scene_tag=0):
    a = np.ascontiguousarray(accum, np.float32)
    ci = _abi.CheckpointInfo(a.shape[1], a.shape[0], next_frame, constant_pdf, scene_tag)
    _host_call(_abi.pthost().pth_checkpoint_save(path.encode(), C.byref(ci), a.ctypes.data))


def load_checkpoint(path):
    """-> (accum [H][W][3] float32, next_frame, constant_pdf, scene_tag)"""
    lib = _abi.pthost()
    ci = _abi.CheckpointInfo()
    _host_call(lib.pth_checkpoint_load(path.encode(), C.byref(ci), None))
    a = np.empty((ci.height, ci.width, 3), np.float32)
    _host_call(lib.pth_checkpoint_load(path.encode(), C.byref(ci), a.ctypes.data))
    return a, ci.next_frame, ci.constant_pdf, ci.scene_tag


def frame_hash(frame):
    """uf::hash(frameNumber), GpuPathTracer/utilfun.cpp:380-389."""
    return _abi.pthost().pth_frame_hash(frame)


# --------------------------------------------------------------------------- defaults
def reference_spheres():
    """The 8 analytic spheres of BasicScene.cpp:181-202 (6 r=600 'walls', mirror, emitter)."""
    rad, px, py = 600.0, 20.0, 15.0
    a = (197.0 / 255.0, 153.0 / 255.0, 92.0 / 255.0)
    rows = [
        ((0.0, -py - rad, -20, rad), a, (0.5, 0.5, 0.5), MAT_DIFF),
        ((0.0, py + rad, -20, rad), a, (0.1, 0.3, 0.4), MAT_DIFF),
        ((px + rad, 0.0, -20, rad), (165 / 255.0, 15 / 255.0, 0.0), (165 / 255.0, 15 / 255.0, 0.0), MAT_DIFF),
        ((-px - rad, 0.0, -20, rad), (30 / 255.0, 76 / 255.0, 14 / 255.0), (30 / 255.0, 76 / 255.0, 14 / 255.0), MAT_DIFF),
        ((0.0, 0.0, -rad * 1.5 - 20, rad), a, (1.0, 1.0, 1.0), MAT_DIFF),
        ((0.0, 0.0, rad * 1.5 + 20, rad), (0.0, 1.0, 0.8), (0.5, 0.5, 0.5), MAT_DIFF),
        ((13.0, -8, -35, 6), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), MAT_SPEC),
        ((10.0, -15, -68, 10), (0.0, 1.0, 1.0), (1.0, 1.0, 1.0), MAT_DIFF),
    ]
    arr = (Sphere * len(rows))()
    for s, (pr, emi, col, mat) in zip(arr, rows):
        s.pos_rad[:] = [np.float32(x) for x in pr]
        s.emi[:] = [np.float32(x) for x in emi]
        s.col[:] = [np.float32(x) for x in col]
        s.mat = mat
    return arr


def default_camera(width, height):
    """BasicScene.cpp:239-259: dist = height/60 (INTEGER divide), fov = tan(45 deg) = 1,
    yaw 0 / pitch 0 => front (0,0,-1), right (1,0,0), up (0,1,0), pos 0."""
    cam = Camera()
    cam.pos[:] = (0, 0, 0)
    cam.front[:] = (0, 0, -1)
    cam.right[:] = (1, 0, 0)
    cam.up[:] = (0, 1, 0)
    cam.dist = float(height // 60)
    cam.aspect = np.float32(width * np.float32(1.0) / np.float32(height))
    cam.fov = 1.0
    return cam


def default_params(width, height, depth=4, tri_mat=MAT_DIFF):
    """kernelInfo defaults, BasicScene.cpp:220-236 and CpuStructs.hpp:45-72."""
    p = Params()
    p.width, p.height, p.depth, p.cull_backfaces = width, height, depth, 1
    p.frame, p.sample_index = 0, 1
    p.tri_mat = tri_mat
    p.tri_col[:] = (np.float32(246.0 / 256.0), np.float32(246.0 / 255.0), np.float32(70.0 / 255.0))
    p.tri_emi[:] = (0, 0, 0)
    p.bk_color[:] = (1, 1, 1)
    p.air_ior, p.glass_ior, p.phong_expo = 1.0, 1.4, 30.0
    p.flags = 0
    p.part_index, p.part_count, p.part_rows = 0, 1, 8
    return p


# --------------------------------------------------------------------------- scenes
def scene_mesh(name):
    """Named benchmark meshes.

    cornell, dragon, gto_sixteen, bunny_low, cube, sphere : the committed fixtures
    cornell_box         : CornellBox-Original with its 8 materials (PTMESH2; extension)
    cornell_box_dragon  : cornell_box ∪ dragon as a gold Phong-metal object (100 036 tris)
    cornell_dragon      : cornell ∪ dragon (100 032 tris; both share one frame, F7)
    cornell_dragon_800k : cornell ∪ 8 dragon copies under fixed transforms inside the box
                          (800 032 tris) — deterministic stand-in for the missing
                          Assets/cornell_dragon.obj blob (.MISSING_LARGE_BLOBS:3)
    cornell_dragon_2700k / _6400k : cornell ∪ 27 / 64 dragon copies — item buffers of ~0.35 / ~0.8 GB,
                          beyond the 256 MiB Infinity Cache (bench.py's extra workload)
    """
    if name == "cornell_box_dragon":
        # the material-carrying Cornell box (assets/cornell_box.ptmesh, lit by its own quad) with
        # the dragon in it as a Phong-metal object: the per-triangle material extension's scene
        m = Mesh.asset("cornell_box")
        d = Mesh.asset("dragon")
        gold = Material()
        gold.col[:] = (0.9, 0.7, 0.3)
        gold.emi[:] = (0.0, 0.0, 0.0)
        gold.mat, gold.phong_expo = MAT_METAL, 30.0
        d.set_materials([gold], np.zeros(d.n_tris, np.int32))
        return m.append(d)
    if name == "cornell_dragon":
        m = Mesh.asset("cornell")
        return m.append(Mesh.asset("dragon"))
    if name == "cornell_dragon_800k":
        m = Mesh.asset("cornell")
        d = Mesh.asset("dragon")
        lo, hi = d.bounds()
        c = 0.5 * (lo + hi)
        # 8 copies: 2 x 2 x 2 lattice inside the box (x in [-15.3,15.5], y in [-15,15.8],
        # z in [-59.4,-27.9]); scale s about the dragon centre, then translate.
        k = 0
        for iy in range(2):
            for iz in range(2):
                for ix in range(2):
                    s = np.float32(0.9 + 0.05 * (k % 3))
                    t = np.array([-7.0 + 13.0 * ix, -11.0 + 12.5 * iy, -52.0 + 14.0 * iz], np.float32)
                    ang = np.float32(0.35 * k)
                    ca, sa = np.cos(ang), np.sin(ang)
                    rot = np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]], np.float32) * s
                    off = t - rot @ c
                    xf = np.concatenate([rot, off[:, None]], axis=1).astype(np.float32)
                    m.append(d, xf)
                    k += 1
        return m
    if name in ("cornell_dragon_2700k", "cornell_dragon_6400k"):
        # beyond the 256 MiB Infinity Cache (SURVEY.md §7 hard parts): cornell + an n x n x n lattice of dragon
        # copies inside the box (n = 3: 2 700 032 triangles, n = 4: 6 400 032), each turned and scaled a little
        n = 3 if name.endswith("2700k") else 4
        m = Mesh.asset("cornell")
        d = Mesh.asset("dragon")
        lo, hi = d.bounds()
        c = 0.5 * (lo + hi)
        ext = float(np.max(hi - lo))
        cell = np.array([26.0, 26.0, 27.0], np.float32) / n          # the box interior, cut into n^3 cells
        k = 0
        for iy in range(n):
            for iz in range(n):
                for ix in range(n):
                    s = np.float32((0.8 + 0.05 * (k % 4)) * float(cell.min()) / ext)
                    t = np.array([-13.0, -13.5, -57.0], np.float32) + cell * (np.array([ix, iy, iz], np.float32) + 0.5)
                    ang = np.float32(0.35 * k)
                    ca, sa = np.cos(ang), np.sin(ang)
                    rot = np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]], np.float32) * s
                    off = t - rot @ c
                    m.append(d, np.concatenate([rot, off[:, None]], axis=1).astype(np.float32))
                    k += 1
        return m
    return Mesh.asset(name)

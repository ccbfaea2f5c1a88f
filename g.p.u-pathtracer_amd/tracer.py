"""PathTracer — Python mirror of the reference's device-facing surface, over the C ABI.

Reference interface mirrored (same names / argument meaning where one exists):
  BasicScene::launchKernel(const kernelInfo&)   tracer.cu:405-415  → PathTracer.launch_kernel(cam, params, spp)
  cudaMalloc/cudaMemcpy of the CudaBVH arrays  BasicScene.cpp:297-306 → PathTracer.upload_bvh(bvh)
  cudaMalloc/cudaMemcpy of the sphere array     BasicScene.cpp:214-215 → PathTracer.upload_spheres(spheres)
  accumBuffer / dev_drawRes allocation          BasicScene.cpp:138-149 → PathTracer.alloc_frame(w, h)
Errors: the reference prints and exit(1)s (utilfun.hpp:81-90); here every failure raises
PtError carrying pt_last_error().  There is no CPU fallback of any kind.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import Camera, Counters, Material, Params, Sphere


class PtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ptmi error {code}: {msg}")
        self.code = code


class DeviceBuffer:
    """A raw device allocation owned by a PathTracer (pt_malloc / pt_free)."""

    def __init__(self, owner, nbytes):
        self.owner, self.nbytes = owner, nbytes
        p = C.c_void_p()
        owner._check(owner._lib.pt_malloc(owner._ctx, nbytes, C.byref(p)))
        self.ptr = p.value

    def zero(self):
        self.owner._check(self.owner._lib.pt_memset(self.owner._ctx, self.ptr, 0, self.nbytes))

    def download(self, dtype, shape):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        self.owner._check(self.owner._lib.pt_download(self.owner._ctx, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        self.owner._check(self.owner._lib.pt_upload(self.owner._ctx, self.ptr, a.ctypes.data, a.nbytes))

    def free(self):
        if self.ptr:
            self.owner._lib.pt_free(self.owner._ctx, self.ptr)
            self.ptr = None


class PathTracer:
    def __init__(self, device=0):
        self._lib = _abi.ptmi()
        ctx = C.c_void_p()
        rc = self._lib.pt_create(device, C.byref(ctx))
        if rc != 0:
            raise PtError(rc, self._lib.pt_last_error(None).decode())
        self._ctx = ctx
        self.device = device

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc != 0:
            raise PtError(rc, self._lib.pt_last_error(self._ctx).decode())

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.pt_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream_ptr):
        self._check(self._lib.pt_set_stream(self._ctx, hip_stream_ptr))

    def set_option(self, opt, value):
        self._check(self._lib.pt_set_option(self._ctx, opt, int(value)))

    def sync(self):
        self._check(self._lib.pt_sync(self._ctx))

    def malloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    # ------------------------------------------------------------------ scene
    def upload_bvh(self, bvh):
        n, t, i = (np.ascontiguousarray(bvh.nodes, np.float32), np.ascontiguousarray(bvh.tris, np.float32),
                   np.ascontiguousarray(bvh.index, np.int32))
        self._check(self._lib.pt_upload_bvh(self._ctx, n.ctypes.data, n.size // 4, t.ctypes.data, t.size // 4,
                                            i.ctypes.data, i.size))

    def upload_bvh_arrays(self, nodes, n_node_vec4, tris, n_tri_vec4, index, n_index):
        self._check(self._lib.pt_upload_bvh(self._ctx, nodes, n_node_vec4, tris, n_tri_vec4, index, n_index))

    def upload_spheres(self, spheres):
        n = len(spheres) if spheres is not None else 0
        self._check(self._lib.pt_upload_spheres(self._ctx, spheres if n else None, n))

    def build_bvh(self, mesh):
        """Build the BVH on the device from a Mesh (pt_build_bvh; extension).  Returns the device
        build time in ms."""
        v = np.ascontiguousarray(mesh.verts, np.float32)
        t = np.ascontiguousarray(mesh.tris, np.int32)
        self._check(self._lib.pt_build_bvh(self._ctx, v.ctypes.data, len(v), t.ctypes.data, len(t)))
        ms = C.c_float()
        self._check(self._lib.pt_last_build_ms(self._ctx, C.byref(ms)))
        return ms.value

    def upload_tri_materials(self, table, tri_material):
        """Per-triangle materials (extension): `table` = sequence of Material, `tri_material` =
        int32 row per ORIGINAL triangle id.  table=None clears (one global material again)."""
        if table is None or len(table) == 0:
            self._check(self._lib.pt_upload_tri_materials(self._ctx, None, 0, None, 0))
            return
        arr = (Material * len(table))(*table)
        ids = np.ascontiguousarray(tri_material, np.int32)
        self._check(self._lib.pt_upload_tri_materials(self._ctx, arr, len(table), ids.ctypes.data_as(C.POINTER(C.c_int32)), len(ids)))

    def scene_info(self):
        a, b, c_, e = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        d = C.c_uint32()
        self._check(self._lib.pt_scene_info(self._ctx, C.byref(a), C.byref(b), C.byref(c_), C.byref(d), C.byref(e)))
        return {"n_inner": a.value, "n_tri_refs": b.value, "n_leaves": c_.value, "max_depth": d.value,
                "device_bytes": e.value}

    # ------------------------------------------------------------------ hot path
    def alloc_frame(self, width, height):
        """accumBuffer (vec3[W*H], zeroed) and dev_drawRes (uint[W*H]), BasicScene.cpp:138-149."""
        acc = self.malloc(width * height * 12)
        acc.zero()
        rgba = self.malloc(width * height * 4)
        rgba.zero()
        return acc, rgba

    def launch_kernel(self, accum_ptr, rgba_ptr, cam, params, spp=1):
        """render(accum, bvh, camera, spp): asynchronous until sync()."""
        self._check(self._lib.pt_render(self._ctx, accum_ptr, rgba_ptr, C.byref(cam), C.byref(params), spp))

    def trace_rays(self, rays_ptr, n, cull, t_ptr, tri_ptr, normal_ptr=None):
        self._check(self._lib.pt_trace_rays(self._ctx, rays_ptr, n, int(cull), t_ptr, tri_ptr, normal_ptr))

    # ------------------------------------------------------------------ measurement
    def counters(self):
        c = Counters()
        self._check(self._lib.pt_get_counters(self._ctx, C.byref(c)))
        return {f: getattr(c, f) for f, _ in Counters._fields_}

    def stage_ms(self):
        """Device ms of the last timed launch_kernel (PT_OPT_TIMING=1) by stage (pt_get_stage_ms)."""
        out = (C.c_float * 6)()
        self._check(self._lib.pt_get_stage_ms(self._ctx, out, 6))
        return dict(zip(("none", "frame", "generate", "extend", "shade", "fold"), [float(v) for v in out]))

    def auto_choice(self):
        """PT_KERNEL_AUTO's pick for the last configuration: (kernel or KERNEL_AUTO while undecided, ms persistent, ms wavefront)."""
        k, a, b = C.c_int(), C.c_float(), C.c_float()
        self._check(self._lib.pt_auto_choice(self._ctx, C.byref(k), C.byref(a), C.byref(b)))
        return k.value, a.value, b.value

    def tree_cost(self):
        """(expected wide-node visits, expected triangle tests) of a random ray: surface-area cost of the 4-wide tree."""
        a, b = C.c_double(), C.c_double()
        self._check(self._lib.pt_tree_cost(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def wave_stats(self):
        out = (C.c_uint64 * 10)()
        self._check(self._lib.pt_get_wave_stats(self._ctx, out, 10))
        names = ("it_node", "act_node", "it_rec", "act_rec", "it_shade", "act_shade", "it_begin", "act_begin", "it_loop", "stack_overflows")
        return dict(zip(names, [int(v) for v in out]))

    def last_build_ms(self):
        """Device time of the build behind the tree on the context; -1 when it is an uploaded hierarchy."""
        ms = C.c_float()
        if self._lib.pt_last_build_ms(self._ctx, C.byref(ms)) != 0:
            return -1.0
        return ms.value

    def last_kernel_ms(self):
        ms = C.c_float()
        self._check(self._lib.pt_last_kernel_ms(self._ctx, C.byref(ms)))
        return ms.value


def algorithmic_bytes(counters, n_spheres):
    """B = sum(64*N_inner + 48*N_tri + 16*N_leaf + 4*[hit]) + 44*n_spheres per
    segment + 28 B per pixel-sample (12 accum read + 12 accum write + 4 RGBA8 write)."""
    return (64 * counters["inner"] + 48 * counters["tris"] + 16 * counters["leaves"] + 4 * counters["hits"]
            + 44 * n_spheres * counters["rays"] + 28 * counters["paths"])

"""pt_build_bvh — the BVH built ON the device (extension; csrc/pt_build.h).

The reference builds on the host and holds no fixture for a device builder; what pins this one
is the path's own invariant: the closest hit (exact Moller-Trumbore, equal-t ties to the smaller
triangle id) does not depend on the tree, so a device-built tree must give
  * the brute-force oracle's (t, id) on explicit ray batches, bit for bit, and
  * the oracle's image (rendered over the HOST builder's tree), bit for bit for the binary walks,
    <= 2 grazing pixels for the quantised wide walks (the bar of test_gpu_wide.py)."""
import numpy as np
import pytest

import gpu_pathtracer_amd as g
import orc
from test_gpu_parity import gpu_trace, golden_camera

pytestmark = pytest.mark.gpu

VARIANTS = {"persistent-postponed": (g.KERNEL_PERSISTENT, 4), "mega-wide": (g.KERNEL_MEGA_BVH2, 2), "wavefront": (g.KERNEL_WAVEFRONT, 2),
            "persistent-unified": (g.KERNEL_PERSISTENT, 1), "mega-whilewhile": (g.KERNEL_MEGA_BVH2, 0)}


@pytest.fixture(scope="module", params=list(VARIANTS), ids=list(VARIANTS))
def pt(request):
    t = g.PathTracer(0)
    t.set_option(g.OPT_KERNEL, VARIANTS[request.param][0])
    t.set_option(g.OPT_WALK, VARIANTS[request.param][1])
    t.max_diff = 2 if VARIANTS[request.param][1] >= 2 else 0
    yield t
    t.close()


def random_rays(mesh, n, seed):
    rng = np.random.default_rng(seed)
    lo, hi = mesh.bounds()
    c, r = 0.5 * (lo + hi), 0.5 * np.linalg.norm(hi - lo)
    o = c + rng.normal(size=(n, 3)) * r * 1.2
    tgt = lo + rng.random((n, 3)) * (hi - lo)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3], rays[:, 4:7] = o, d
    return rays


@pytest.mark.parametrize("name", ["cornell", "bunny_low", "gto_sixteen", "dragon"])
def test_device_tree_hits_equal_brute_force(name):
    mesh = g.scene_mesh(name)
    t = g.PathTracer(0)
    try:
        for algo, leaf_max in ((0, 1), (0, 2), (0, 4), (1, 2)):     # LBVH with three leaf sizes, PLOC (the default)
            t.set_option(g.OPT_BUILD_ALGO, algo)
            t.set_option(g.OPT_LEAF_MAX, leaf_max)
            ms = t.build_bvh(mesh)
            info = t.scene_info()
            print(f"{name} algo {algo} leaf_max {leaf_max}: built in {ms:.2f} ms on the device, {info}")
            assert info["n_tri_refs"] == mesh.n_tris and info["n_inner"] == mesh.n_tris - 1
            assert info["n_leaves"] >= mesh.n_tris / leaf_max and info["max_depth"] <= 64
            n = 20000 if name == "dragon" else 60000
            rays = random_rays(mesh, n, 7 + leaf_max)
            for cull in (True, False):
                tg, ig, ng = gpu_trace(t, rays, cull)
                tb, ib, nb = orc.trace_brute(mesh, rays, cull)
                assert np.array_equal(ig, ib)
                assert np.array_equal(tg, tb)
                hit = ib >= 0
                assert hit.mean() > 0.2 and np.array_equal(ng[hit], nb[hit])
    finally:
        t.close()


@pytest.mark.parametrize("scene,W,H,spp", [("cornell", 256, 256, 2), ("cornell_dragon", 640, 360, 2), ("gto_sixteen", 321, 175, 3)])
def test_device_tree_image_equals_oracle(pt, scene, W, H, spp):
    mesh = g.scene_mesh(scene)
    host_bvh = g.Bvh(mesh)
    sph = g.reference_spheres()
    cam, p = g.default_camera(W, H), g.default_params(W, H)
    p.frame, p.flags = 4, g.FLAG_WRITE_RGBA
    ref, rref, _ = orc.render(host_bvh, sph, cam, p, spp)
    pt.upload_tri_materials(None, None)
    ms = pt.build_bvh(mesh)
    pt.upload_spheres(sph)
    acc, rgba = pt.alloc_frame(W, H)
    pt.launch_kernel(acc.ptr, rgba.ptr, cam, p, spp)
    pt.sync()
    a, r = acc.download(np.float32, (H, W, 3)), rgba.download(np.uint32, (H, W))
    acc.free()
    rgba.free()
    n_diff = int(np.any(a != ref, axis=-1).sum())
    print(f"{scene}: device build {ms:.2f} ms, differing pixels {n_diff} of {W * H}")
    assert n_diff <= pt.max_diff
    if n_diff == 0:
        assert np.array_equal(r, rref)


@pytest.mark.parametrize("name", ["cornell", "bunny_low", "dragon"])
def test_ploc_tree_hits_equal_brute_force(name):
    """PT_OPT_BUILD_ALGO 1 (PLOC): same invariant — the closest hit does not depend on the tree."""
    mesh = g.scene_mesh(name)
    t = g.PathTracer(0)
    try:
        t.set_option(g.OPT_BUILD_ALGO, 1)
        ms = t.build_bvh(mesh)
        info = t.scene_info()
        print(f"{name}: PLOC built in {ms:.2f} ms on the device, {info}")
        assert info["n_tri_refs"] == mesh.n_tris and mesh.n_tris / 2 <= info["n_leaves"] <= mesh.n_tris and info["max_depth"] <= 64
        rays = random_rays(mesh, 30000, 3)
        for cull in (True, False):
            tg, ig, ng = gpu_trace(t, rays, cull)
            tb, ib, nb = orc.trace_brute(mesh, rays, cull)
            assert np.array_equal(ig, ib) and np.array_equal(tg, tb)
            hit = ib >= 0
            assert np.array_equal(ng[hit], nb[hit])
        # and a rendered frame against the oracle over the host tree
        W, H = 320, 200
        cam, p = g.default_camera(W, H), g.default_params(W, H)
        sph = g.reference_spheres()
        ref, _, _ = orc.render(g.Bvh(mesh), sph, cam, p, 2)
        t.upload_spheres(sph)
        acc, rgba = t.alloc_frame(W, H)
        t.launch_kernel(acc.ptr, rgba.ptr, cam, p, 2)
        t.sync()
        a = acc.download(np.float32, (H, W, 3))
        acc.free()
        rgba.free()
        assert int(np.any(a != ref, axis=-1).sum()) <= 2
    finally:
        t.close()


@pytest.mark.parametrize("name,presplit", [("gto_sixteen", 200), ("cornell_dragon", 200), ("bunny_low", 50)])
def test_presplit_tree_hits_equal_brute_force(name, presplit):
    """PT_OPT_PRESPLIT: long triangles enter the builder as several primitives (slabs of their box);
    hits, ids and normals must not change."""
    mesh = g.scene_mesh(name)
    t = g.PathTracer(0)
    try:
        for algo in (1, 0):
            t.set_option(g.OPT_BUILD_ALGO, algo)
            t.set_option(g.OPT_PRESPLIT, presplit)
            ms = t.build_bvh(mesh)
            info = t.scene_info()
            print(f"{name} algo {algo} presplit {presplit}: {ms:.2f} ms, {info}")
            assert info["n_tri_refs"] > mesh.n_tris          # something was split
            rays = random_rays(mesh, 40000, 11)
            for cull in (True, False):
                tg, ig, ng = gpu_trace(t, rays, cull)
                tb, ib, nb = orc.trace_brute(mesh, rays, cull)
                assert np.array_equal(ig, ib) and np.array_equal(tg, tb)
                hit = ib >= 0
                assert np.array_equal(ng[hit], nb[hit])
    finally:
        t.close()


def test_degenerate_inputs():
    t = g.PathTracer(0)
    try:
        # one triangle, two triangles, 300 copies of ONE triangle (all Morton keys equal: the
        # position tie-break keeps the hierarchy a balanced tree; equal t -> smallest id)
        v = np.array([[-1, -1, -5], [1, -1, -5], [0, 1, -5], [3, -1, -6], [5, -1, -6], [4, 1, -6]], np.float32)
        rays = np.zeros((3, 8), np.float32)
        rays[:, 4:7] = (0, 0, -1)
        rays[1, :3] = (4, 0, 0)
        rays[2, :3] = (9, 9, 0)
        for tris, expect in (([[0, 1, 2]], [0, -1, -1]), ([[0, 1, 2], [3, 4, 5]], [0, 1, -1]),
                             ([[0, 1, 2]] * 300 + [[3, 4, 5]] * 5, [0, 300, -1])):
            m = g.Mesh.from_arrays(v, np.array(tris, np.int32))
            for leaf_max in (1, 2, 8):
                t.set_option(g.OPT_LEAF_MAX, leaf_max)
                t.build_bvh(m)
                tg, ig, _ = gpu_trace(t, rays, False)
                assert ig.tolist() == expect
                assert tg[0] == 5.0
        # errors: index out of range, NaN vertex, empty mesh
        lib = t._lib
        bad = np.array([[0, 1, 7]], np.int32)
        assert lib.pt_build_bvh(t._ctx, v.ctypes.data, 6, bad.ctypes.data, 1) != 0
        vn = v.copy()
        vn[2, 1] = np.nan
        ok = np.array([[0, 1, 2]], np.int32)
        assert lib.pt_build_bvh(t._ctx, vn.ctypes.data, 6, ok.ctypes.data, 1) != 0
        assert lib.pt_build_bvh(t._ctx, v.ctypes.data, 6, ok.ctypes.data, 0) != 0
        # ... and the context still works afterwards
        t.build_bvh(g.Mesh.from_arrays(v, ok))
        assert gpu_trace(t, rays, False)[1].tolist() == [0, -1, -1]
    finally:
        t.close()


def test_device_build_800k_speed_and_parity():
    """The bench scene: build time, and the 1080p frame over the DEVICE-built tree against the ORACLE's
    image over the host tree (the closest hit does not depend on the tree), plus the host-tree render."""
    mesh = g.scene_mesh("cornell_dragon_800k")
    W, H = 1920, 1080
    cam, p = g.default_camera(W, H), g.default_params(W, H)
    sph = g.reference_spheres()
    t = g.PathTracer(0)
    try:
        def frame():
            t.upload_spheres(sph)
            acc, rgba = t.alloc_frame(W, H)
            t.launch_kernel(acc.ptr, rgba.ptr, cam, p, 1)
            t.sync()
            a = acc.download(np.float32, (H, W, 3))
            acc.free()
            rgba.free()
            return a
        bvh = g.Bvh(mesh)
        ref, _, _ = orc.render(bvh, sph, cam, p, 1, want_rgba=False)
        t.upload_bvh(bvh)
        a_host = frame()
        ms = min(t.build_bvh(mesh) for _ in range(3))
        info = t.scene_info()
        a_dev = frame()
        n_diff = int(np.any(a_host != a_dev, axis=-1).sum())
        n_orc = int(np.any(ref != a_dev, axis=-1).sum())
        err = float(np.sqrt(np.mean(np.sum((a_dev.astype(np.float64) - ref) ** 2, axis=-1))))
        print(f"800k: device build {ms:.1f} ms, {info}; differing pixels vs host tree {n_diff}, vs oracle {n_orc} (L2 {err:.2e})")
        assert ms < 200.0
        assert n_diff <= 40
        assert err < 1e-3 and n_orc <= 40
    finally:
        t.close()


@pytest.mark.parametrize("scene", ["cornell", "gto_sixteen", "cornell_dragon"])
def test_upload_with_rebuild_is_bit_identical(pt, scene):
    """PT_OPT_REBUILD: pt_upload_bvh keeps the triangles of the Compact arrays (SBVH duplicates folded,
    original ids kept) and clusters them again on the device; the frame equals the plain upload's."""
    mesh = g.scene_mesh(scene)
    bvh = g.Bvh(mesh)                       # spatial splits on: gto_sixteen lists 63 % of its triangles more than once
    W, H = 400, 225
    cam, p = g.default_camera(W, H), g.default_params(W, H)
    p.frame = 9
    sph = g.reference_spheres()
    ref, _, _ = orc.render(bvh, sph, cam, p, 2)
    frames = []
    for rebuild in (0, 1):
        pt.set_option(g.OPT_REBUILD, rebuild)
        try:
            pt.upload_bvh(bvh)
        finally:
            pt.set_option(g.OPT_REBUILD, 0)
        info = pt.scene_info()
        pt.upload_spheres(sph)
        acc, rgba = pt.alloc_frame(W, H)
        pt.launch_kernel(acc.ptr, rgba.ptr, cam, p, 2)
        pt.sync()
        frames.append(acc.download(np.float32, (H, W, 3)))
        acc.free()
        rgba.free()
        if rebuild:
            assert info["n_tri_refs"] == mesh.n_tris          # duplicates folded
    assert int(np.any(frames[0] != ref, axis=-1).sum()) <= pt.max_diff
    assert int(np.any(frames[1] != ref, axis=-1).sum()) <= pt.max_diff


def test_rebuild_2_keeps_the_cheaper_tree():
    """PT_OPT_REBUILD 2: the uploaded hierarchy AND the re-clustered one are built, the one whose 4-wide tree costs a random
    ray fewer node visits (pt_tree_cost) stays.  gto_sixteen (long triangles the host builder's spatial splits cut): the
    caller's; the 32-triangle cornell box: the re-clustered one.  The choice is reproducible, the picture is the oracle's."""
    t = g.PathTracer(0)
    try:
        with pytest.raises(g.PtError):
            t.tree_cost()                                                    # nothing uploaded yet
        for scene, expect_rebuilt in (("gto_sixteen", False), ("cornell", True)):
            mesh = g.scene_mesh(scene)
            bvh = g.Bvh(mesh)
            costs = {}
            for mode in (0, 1, 2, 2):
                t.set_option(g.OPT_REBUILD, mode)
                t.upload_bvh(bvh)
                costs.setdefault(mode, []).append((t.tree_cost(), t.scene_info()))
            t.set_option(g.OPT_REBUILD, 0)
            (c0, i0), (c1, i1) = costs[0][0], costs[1][0]
            (c2, i2), (c2b, i2b) = costs[2]
            print(f"{scene}: node visits {c0[0]:.3f} uploaded, {c1[0]:.3f} re-clustered -> kept {c2[0]:.3f}")
            assert (c1[0] < c0[0]) == expect_rebuilt
            assert c2 == (c1 if expect_rebuilt else c0) and i2 == (i1 if expect_rebuilt else i0)
            assert c2b == c2 and i2b == i2                                   # reproducible
            assert c2[0] == min(c0[0], c1[0])
            # the kept tree renders the oracle's picture
            t.set_option(g.OPT_REBUILD, 2)
            t.upload_bvh(bvh)
            t.set_option(g.OPT_REBUILD, 0)
            W, H = 320, 200
            cam, p = g.default_camera(W, H), g.default_params(W, H)
            p.frame = 3
            sph = g.reference_spheres()
            t.upload_spheres(sph)
            ref, _, _ = orc.render(bvh, sph, cam, p, 2)
            acc, rgba = t.alloc_frame(W, H)
            t.launch_kernel(acc.ptr, rgba.ptr, cam, p, 2)
            t.sync()
            a = acc.download(np.float32, (H, W, 3))
            acc.free()
            rgba.free()
            assert int(np.any(a != ref, axis=-1).sum()) <= 2
        with pytest.raises(g.PtError):
            t.set_option(g.OPT_REBUILD, 3)
    finally:
        t.close()


@pytest.mark.parametrize("scene", ["gto_sixteen", "cornell_dragon"])
def test_upload_time_optimisation_keeps_hits_and_lowers_the_cost(scene):
    """PT_OPT_OPTIMIZE: the uploaded hierarchy — and, with PT_OPT_REBUILD, the device's re-clustered one, fetched back — re-arranged
    by insertion-based optimisation (csrc/pt_tree_opt.h) before it is emitted.  Ray batches == brute force bit for bit (t, id,
    normal), the frame == the oracle's over the caller's own tree, the 4-wide tree's area cost in node visits drops, and
    PT_OPT_REBUILD 2 keeps the cheaper of the two optimised trees."""
    mesh = g.scene_mesh(scene)
    bvh = g.Bvh(mesh)
    lo, hi = mesh.bounds()
    rays = np.concatenate([orc.random_rays(60000, lo, hi, seed=3), orc.primary_rays(g.default_camera(320, 180), 320, 180, frame=1)[::7]])
    W, H = 480, 270
    cam, p = g.default_camera(W, H), g.default_params(W, H)
    p.frame = 2
    sph = g.reference_spheres()
    ref, _, _ = orc.render(bvh, sph, cam, p, 2, want_rgba=False)
    tb, ib, nb = orc.trace_brute(mesh, rays)
    t = g.PathTracer(0)
    try:
        costs = {}
        for rebuild, passes in ((0, 0), (0, 2), (1, 0), (1, 2), (2, 2)):
            t.set_option(g.OPT_OPTIMIZE, passes)
            t.set_option(g.OPT_REBUILD, rebuild)
            t.upload_bvh(bvh)
            t.set_option(g.OPT_REBUILD, 0)
            t.set_option(g.OPT_OPTIMIZE, 0)
            costs[(rebuild, passes)] = t.tree_cost()[0]
            tg, ig, ng = gpu_trace(t, rays)
            assert np.array_equal(tg, tb) and np.array_equal(ig, ib)
            hit = ib >= 0
            assert np.array_equal(ng[hit], nb[hit])
            t.upload_spheres(sph)
            acc, rgba = t.alloc_frame(W, H)
            t.launch_kernel(acc.ptr, rgba.ptr, cam, p, 2)
            t.sync()
            got = acc.download(np.float32, (H, W, 3))
            acc.free()
            rgba.free()
            assert int(np.any(got != ref, axis=-1).sum()) <= 2
            if rebuild == 1:
                assert t.last_build_ms() > 0          # a device build stands behind the tree, optimised or not
        print(f"{scene}: area cost in node visits, uploaded {costs[(0, 0)]:.3f} -> optimised {costs[(0, 2)]:.3f}; re-clustered {costs[(1, 0)]:.3f} "
              f"-> optimised {costs[(1, 2)]:.3f}; PT_OPT_REBUILD 2 keeps {costs[(2, 2)]:.3f}")
        assert costs[(0, 2)] < costs[(0, 0)] and costs[(1, 2)] < costs[(1, 0)]
        assert costs[(2, 2)] == min(costs[(0, 2)], costs[(1, 2)])
    finally:
        t.close()

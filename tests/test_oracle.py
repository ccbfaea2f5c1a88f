"""CPU tests of the oracle itself (no GPU): pinned against the reference-generated fixture,
its own brute force, and the committed regression vectors."""
import ctypes as C
import os

import numpy as np
import pytest

import gpu_pathtracer_amd as g
import orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_camera(W, H):
    cam = g.default_camera(W, H)
    cam.dist = 18.0 * H / 1080.0
    return cam


# ---------------------------------------------------------------- pin vs the reference
@pytest.mark.parametrize("name", ["dragon", "gto_sixteen"])
def test_closest_hit_matches_reference_cpu_intersector(name):
    """tests/golden/ref_primary_hits.npz holds distances computed by the REFERENCE's
    Triangle::intersect + KDNode::hit (CpuRayTracer/src/triangle.hpp:49-72, kdtree.cpp:59-90),
    compiled from /root/reference by oracle/Makefile and run in the dev container.  The
    reference intersector is double precision, never culls and only rejects det == 0, so:
    cull off, tolerance 1e-4 relative (binary32 Moller-Trumbore vs binary64)."""
    z = np.load(os.path.join(GOLD, "ref_primary_hits.npz"))
    rays, t_ref = z[name + "_rays"], z[name + "_t_ref"]
    bvh = g.Bvh(g.scene_mesh(name))
    t, tri, _, _ = orc.trace_bvh(bvh, rays, cull=False)
    hit_o, hit_r = t < 1e30, t_ref < 1e30
    assert hit_r.sum() > 500
    assert (hit_o != hit_r).sum() <= 2  # grazing rays may flip between f32 and f64
    both = hit_o & hit_r
    rel = np.abs(t[both] - t_ref[both]) / t_ref[both]
    assert rel.max() < 1e-4


# ---------------------------------------------------------------- BVH walk vs brute force
@pytest.mark.parametrize("name,n", [("cornell", 20000), ("gto_sixteen", 6000), ("dragon", 1500), ("cube", 5000)])
@pytest.mark.parametrize("cull", [True, False])
def test_bvh_equals_bruteforce(name, n, cull):
    """The while-while traversal (cudaUtils.h:256-460) must return the same t bit for bit and
    the same triangle id as a loop over every triangle (cudaUtils.h:194-217's idea)."""
    mesh = g.scene_mesh(name)
    bvh = g.Bvh(mesh)
    lo, hi = mesh.bounds()
    rays = orc.random_rays(n, lo, hi, seed=11)
    t1, tri1, n1, cnt = orc.trace_bvh(bvh, rays, cull)
    t2, tri2, n2 = orc.trace_brute(mesh, rays, cull)
    assert cnt["rays"] == n
    assert (t1 < 1e30).sum() > n // 50
    # a slab test may cull a grazing true hit (not watertight); allow 1e-4 of the rays
    bad = (t1 != t2) | (tri1 != tri2)
    assert bad.sum() <= max(1, n // 10000), bad.sum()
    ok = ~bad
    assert np.array_equal(n1[ok], n2[ok])


def test_spatial_split_tree_gives_same_hits():
    mesh = g.scene_mesh("gto_sixteen")
    lo, hi = mesh.bounds()
    rays = orc.random_rays(4000, lo, hi, seed=5)
    a = g.Bvh(mesh, split_alpha=-1.0)     # object splits only
    b = g.Bvh(mesh)                       # default: the reference's splitAlpha = 1e-5
    assert b.stats["n_tri_refs"] >= a.stats["n_tri_refs"]
    ta, ia, _, _ = orc.trace_bvh(a, rays)
    tb, ib, _, _ = orc.trace_bvh(b, rays)
    assert (ta != tb).sum() <= 1 and (ia != ib).sum() <= 1


# ---------------------------------------------------------------- unit: ray/triangle
def _one_tri_mesh():
    v = np.array([[0, 0, -5], [1, 0, -5], [0, 1, -5]], np.float32)
    return g.Mesh.from_arrays(v, np.array([[0, 1, 2]], np.int32))


def _ray(o, d):
    d = np.asarray(d, np.float64)
    d = d / np.linalg.norm(d)
    r = np.zeros((1, 8), np.float32)
    r[0, 0:3], r[0, 4:7] = o, d
    return r


def test_moller_trumbore_hand_cases():
    """intersectRayTriangleEdge, cudaUtils.h:135-172."""
    m = _one_tri_mesh()
    # the triangle (0,0)-(1,0)-(0,1) at z=-5 has normal +z: a ray travelling -z sees det > 0 (front)
    t, tri, n = orc.trace_brute(m, _ray((0.25, 0.25, 0), (0, 0, -1)), cull=True)
    assert tri[0] == 0 and t[0] == pytest.approx(5.0)
    assert np.allclose(n[0], [0, 0, 1])  # cross(v0-v1, v0-v2)
    # from behind: culled when cullBackFaces, hit otherwise (:151-155)
    t, tri, _ = orc.trace_brute(m, _ray((0.25, 0.25, -10), (0, 0, 1)), cull=True)
    assert tri[0] == -1 and t[0] > 1e30
    t, tri, _ = orc.trace_brute(m, _ray((0.25, 0.25, -10), (0, 0, 1)), cull=False)
    assert tri[0] == 0 and t[0] == pytest.approx(5.0)
    # outside u/v range, parallel, behind the origin (t <= 0)
    assert orc.trace_brute(m, _ray((2, 2, 0), (0, 0, -1)))[1][0] == -1
    assert orc.trace_brute(m, _ray((0.25, 0.25, 0), (1, 0, 0)))[1][0] == -1
    assert orc.trace_brute(m, _ray((0.25, 0.25, -6), (0, 0, -1)), cull=False)[1][0] == -1
    # vertex and edge are inside (u>=0, v>=0, u+v<=1 are inclusive, :160-161)
    assert orc.trace_brute(m, _ray((0, 0, 0), (0, 0, -1)))[1][0] == 0
    assert orc.trace_brute(m, _ray((0.5, 0.5, 0), (0, 0, -1)))[1][0] == 0


def test_zero_direction_components_use_ooeps():
    """cudaUtils.h:283-286: a zero direction component becomes 2^-80, never a division by 0."""
    mesh = g.scene_mesh("cornell")
    bvh = g.Bvh(mesh)
    rays = np.concatenate([_ray((0, 0, 0), (0, 0, -1)), _ray((1, 2, 0), (0, 0, -1)), _ray((0, 0, -40), (0, -1, 0)),
                           _ray((0, 0, -40), (1, 0, 0))])
    t1, i1, _, _ = orc.trace_bvh(bvh, rays, cull=False)
    t2, i2, _ = orc.trace_brute(mesh, rays, cull=False)
    assert np.array_equal(t1, t2) and np.array_equal(i1, i2)
    assert (i1 >= 0).all()


# ---------------------------------------------------------------- unit: math contract
def test_wang64_is_the_reference_frame_hash():
    """uf::hash, utilfun.cpp:380-389; the product's host library restates it separately."""
    L = orc.lib()
    for f in (0, 1, 2, 77, 2 ** 40 + 17, 2 ** 64 - 1):
        assert L.orc_wang64(f) == g.frame_hash(f)
    z = np.load(os.path.join(GOLD, "oracle_kat.npz"))
    for f, h in zip(z["wang64_in"], z["wang64_out"]):
        assert L.orc_wang64(int(f)) == int(h)


def test_rng_known_answers_and_range():
    L = orc.lib()
    z = np.load(os.path.join(GOLD, "oracle_kat.npz"))
    for (f, p, d), v in zip(z["rng_key"], z["rng_val"]):
        assert np.float32(L.orc_rng_draw(int(f), int(p), int(d))) == v
    u = np.array([L.orc_rng_draw(5, p, d) for p in range(2000) for d in range(8)], np.float32)
    assert u.min() > 0.0 and u.max() <= 1.0  # (0,1] like curand_uniform
    assert abs(u.mean() - 0.5) < 0.01
    # neighbouring pixels / frames are decorrelated
    a = np.array([L.orc_rng_draw(0, p, 0) for p in range(4000)])
    b = np.array([L.orc_rng_draw(1, p, 0) for p in range(4000)])
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 0.05
    assert abs(np.corrcoef(a, b)[0, 1]) < 0.05


def test_sincos2pi_accuracy_and_kat():
    L = orc.lib()
    z = np.load(os.path.join(GOLD, "oracle_kat.npz"))
    for u, (c_ref, s_ref) in zip(z["sincos_u"], z["sincos_cs"]):
        c, s = C.c_float(), C.c_float()
        L.orc_sincos2pi(float(u), C.byref(c), C.byref(s))
        assert np.float32(c.value) == c_ref and np.float32(s.value) == s_ref
        assert abs(c.value - np.cos(2 * np.pi * np.float64(u))) < 3e-7
        assert abs(s.value - np.sin(2 * np.pi * np.float64(u))) < 3e-7


def test_pow01_accuracy_and_kat():
    L = orc.lib()
    z = np.load(os.path.join(GOLD, "oracle_kat.npz"))
    for i, x in enumerate(z["pow_x"]):
        for j, y in enumerate(z["pow_y"]):
            v = np.float32(L.orc_pow01(float(x), float(y)))
            assert v == z["pow_v"][i, j]
            ref = float(np.float64(x) ** np.float64(y))
            assert abs(v - ref) <= 4e-6 * max(ref, 1e-30) + 1e-38


def test_camera_ray_restates_getCamRayDir():
    """cudaUtils.h:111-134 in numpy float32, same evaluation order."""
    W, H = 64, 48
    cam = golden_camera(W, H)
    o, d = (C.c_float * 3)(), (C.c_float * 3)()
    f = np.float32
    for px, py, u0, u1 in [(0, 0, 0.5, 0.5), (63, 47, 0.25, 1.0), (31, 7, 2.0 ** -24, 0.75)]:
        orc.lib().orc_camera_ray(C.byref(cam), px, py, W, H, u0, u1, o, d)
        jx, jy = f(u0) - f(0.5), f(u1) - f(0.5)
        xs = (f(px) - f(W) / f(2) + f(0.5) + jx) * f(cam.dist) * f(cam.aspect) * f(cam.fov) / f(W - 1)
        ys = (f(py) - f(H) / f(2) + f(0.5) + jy) * f(cam.dist) * f(cam.fov) / f(H - 1)
        dirv = np.array([xs, ys, -f(cam.dist)], np.float64)
        assert np.allclose(o[:], dirv, rtol=1e-6, atol=1e-6)        # origin ON the image plane
        assert np.allclose(d[:], dirv / np.linalg.norm(dirv), rtol=1e-6, atol=1e-6)


def test_accumulate_and_pack_quirks():
    """tracer.cu:386-398: running mean, clamp EVERY frame, truncating 8-bit pack 0x00BBGGRR."""
    L = orc.lib()
    acc = np.array([0.25, 0.5, 1.0], np.float32)
    rgba = np.zeros(1, np.uint32)
    s = np.array([4.0, 0.5, 0.0], np.float32)
    L.orc_accumulate(acc.ctypes.data, rgba.ctypes.data, s.ctypes.data, 1)  # N=1 overwrites
    assert np.array_equal(acc, [1.0, 0.5, 0.0])                            # 4.0 clamped to 1
    assert rgba[0] == (0 << 16) | (127 << 8) | 255                         # 127.5 truncates
    s = np.array([0.0, 0.0, 1.0], np.float32)
    L.orc_accumulate(acc.ctypes.data, rgba.ctypes.data, s.ctypes.data, 2)
    assert np.allclose(acc, [0.5, 0.25, 0.5])


# ---------------------------------------------------------------- render semantics
def test_golden_images_regression():
    z = np.load(os.path.join(GOLD, "oracle_images.npz"))
    W = H = 64
    bvh = g.Bvh(g.scene_mesh("cornell"))
    sph = g.reference_spheres()
    cam = golden_camera(W, H)
    for mat, mname in ((g.MAT_DIFF, "diff"), (g.MAT_METAL, "metal"), (g.MAT_SPEC, "spec"), (g.MAT_REFR, "refr")):
        p = g.default_params(W, H, tri_mat=mat)
        p.flags = g.FLAG_WRITE_RGBA
        acc, rgba, _ = orc.render(bvh, sph, cam, p, spp=4)
        assert np.array_equal(acc, z[f"{mname}_4"]), mname
        assert np.array_equal(rgba, z[f"{mname}_4_rgba"]), mname


def test_spp_equals_single_sample_calls():
    """spp samples in one call == spp launches with constantPdf = 1..spp (BasicScene.cpp:399)."""
    W = H = 32
    bvh = g.Bvh(g.scene_mesh("cornell"))
    sph = g.reference_spheres()
    cam = golden_camera(W, H)
    p = g.default_params(W, H)
    one, _, _ = orc.render(bvh, sph, cam, p, spp=5)
    acc = np.zeros((H, W, 3), np.float32)
    for s in range(5):
        q = g.default_params(W, H)
        q.frame, q.sample_index = s, 1 + s
        orc.render(bvh, sph, cam, q, spp=1, accum=acc)
    assert np.array_equal(one, acc)


def test_closed_room_segment_count_and_miss_background():
    W = H = 32
    bvh = g.Bvh(g.scene_mesh("cornell"))
    cam = golden_camera(W, H)
    p = g.default_params(W, H)
    _, _, cnt = orc.render(bvh, g.reference_spheres(), cam, p, spp=2)
    assert cnt["paths"] == W * H * 2
    assert cnt["rays"] == W * H * 2 * p.depth      # closed room: every path runs `depth` segments
    acc, _, cnt2 = orc.render(bvh, None, cam, p, spp=1)
    assert cnt2["rays"] < W * H * p.depth          # open box: paths end on the first miss
    corner = acc[0, 0]
    assert np.array_equal(corner, np.array(p.bk_color[:], np.float32))  # unmasked bkColor (:140-142)


def test_partition_is_bit_identical():
    """Rendering stripes on N 'GPUs' and merging equals the single render (RNG keyed by pixel)."""
    W, H = 48, 40
    bvh = g.Bvh(g.scene_mesh("cornell"))
    sph = g.reference_spheres()
    cam = golden_camera(W, H)
    full, _, _ = orc.render(bvh, sph, cam, g.default_params(W, H), spp=2)
    merged = np.zeros_like(full)
    for part in range(3):
        p = g.default_params(W, H)
        p.part_index, p.part_count, p.part_rows = part, 3, 8
        orc.render(bvh, sph, cam, p, spp=2, accum=merged)
    assert np.array_equal(full, merged)


def test_config1_cornell_256_through_the_reference_cpu_tracer(tmp_path):
    """BASELINE.json configs[0], literally: cornell.obj 256x256 1 spp through the reference's own CpuRayTracer classes on
    the host CPU (oracle/_ref = Scene / Mesh / KDNode / Material compiled from /root/reference; plumbing, no GPU).  The
    render loop is renderer.cpp:197-222's, the room main.cpp:26-35's; every camera ray is one path, the segment counter
    sits in front of the mesh (a forwarding Object), and the image is finite and lit."""
    import json
    import subprocess
    if not os.path.exists(orc.REF_BIN):
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    mesh = g.scene_mesh("cornell")
    v, f = mesh.verts.astype(np.float64), mesh.tris
    lo, hi = v.min(0), v.max(0)
    v = (v - 0.5 * (lo + hi)) * (3.0 / float(np.max(hi - lo)))     # into the room: 3 units, z-up, on the floor
    v = np.stack([v[:, 0], -v[:, 2], v[:, 1]], -1)
    v[:, 2] -= v[:, 2].min()
    obj, out = tmp_path / "cornell.obj", tmp_path / "img.f32"
    with open(obj, "w") as fh:
        fh.write("o cornell\n")
        np.savetxt(fh, v, fmt="v %.6f %.6f %.6f")
        np.savetxt(fh, f + 1, fmt="f %d %d %d")
    r = subprocess.run([orc.REF_BIN, "render", str(obj), "256", "256", "1", "0", "0", "0", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-300:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    img = np.fromfile(out, np.float32).reshape(256, 256, 3)
    assert j["paths"] == 256 * 256 and j["segments"] >= j["paths"]
    assert np.isfinite(img).all() and 0.05 < img.mean() < 2.2 and abs(img.mean() - j["mean"]) < 1e-4


def test_arbiter_sample_pixels_matches_render_and_brute_force():
    """orc.sample_pixels (the arbiter of the GPU tests' differing pixels): folded sample colours == orc.render's pixels, and the
    brute-force closest hit (no tree) reproduces the walk's segments on a scene without grazing cases."""
    mesh = g.scene_mesh("cornell_dragon")
    bvh = g.Bvh(mesh)
    W, H, spp = 64, 48, 3
    cam, p = g.default_camera(W, H), g.default_params(W, H)
    cam.dist = 18.0 * H / 1080
    sph = g.reference_spheres()
    ref, _, _ = orc.render(bvh, sph, cam, p, spp=spp, want_rgba=False)
    px = [(x, y) for y in range(0, H, 7) for x in range(0, W, 9)]
    col_w, t_w, id_w = orc.sample_pixels(px, sph, cam, p, spp, bvh=bvh)
    col_b, t_b, id_b = orc.sample_pixels(px, sph, cam, p, spp, mesh=mesh)
    assert np.array_equal(orc.fold_samples(col_w, 1), np.array([ref[y, x] for x, y in px]))
    assert np.array_equal(col_w, col_b) and np.array_equal(t_w, t_b) and np.array_equal(id_w, id_b)
    assert (id_w >= 0).mean() > 0.3 and (id_w[..., 0] >= -1).all()

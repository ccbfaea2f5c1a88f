"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs.  The BVH2 megakernel shares the oracle's arithmetic contract
(DESIGN.md §4), so the bar here is BIT-EXACT float accumulators; the north_star tolerance
(per-pixel L2 < 1e-3) is asserted as well and is the bar for the re-ordered kernel variants.
"""
import os

import numpy as np
import pytest

import gpu_pathtracer_amd as g
import orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
L2_TOL = 1e-3  # BASELINE.json north_star: per-pixel L2 vs CPU ref < 1e-3


def golden_camera(W, H):
    cam = g.default_camera(W, H)
    cam.dist = 18.0 * H / 1080.0
    return cam


def l2(a, b):
    return float(np.sqrt(np.mean(np.sum((a.astype(np.float64) - b) ** 2, axis=-1))))


VARIANTS = {
    # id: (kernel, walk, occupancy, lds_stack, top_nodes)
    "mega-unified": (g.KERNEL_MEGA_BVH2, 1, 8, 16, 0),
    "persistent-unified": (g.KERNEL_PERSISTENT, 1, 8, 16, 0),
    "mega-whilewhile": (g.KERNEL_MEGA_BVH2, 0, 8, 16, 64),
    "persistent-whilewhile-occ4": (g.KERNEL_PERSISTENT, 0, 4, 0, 256),
}


@pytest.fixture(scope="module", params=list(VARIANTS), ids=list(VARIANTS))
def pt(request):
    """Every test runs against every exact kernel variant (schedule, walk order, register
    budget, LDS stack window, LDS top-of-tree mirror): all must equal the oracle bit for bit."""
    k, walk, occ, lstk, top = VARIANTS[request.param]
    t = g.PathTracer(0)
    t.set_option(g.OPT_KERNEL, k)
    t.set_option(g.OPT_WALK, walk)
    t.set_option(g.OPT_LEAF_MAX, 0 if "whilewhile" in request.param else 2)
    t.set_option(g.OPT_OCCUPANCY, occ)
    t.set_option(g.OPT_LDS_STACK, lstk)
    t.set_option(g.OPT_TOP_NODES, top)
    t.walk = walk
    yield t
    t.close()


_bvh_cache = {}


def bvh_of(name, **kw):
    key = (name, tuple(sorted(kw.items())))
    if key not in _bvh_cache:
        mesh = g.scene_mesh(name)
        _bvh_cache[key] = (mesh, g.Bvh(mesh, **kw))
    return _bvh_cache[key]


def gpu_render(pt, bvh, spheres, cam, p, spp=1, accum_init=None):
    W, H = p.width, p.height
    if bvh is not None:
        pt.upload_bvh(bvh)
    pt.upload_spheres(spheres)
    acc, rgba = pt.alloc_frame(W, H)
    if accum_init is not None:
        acc.upload(accum_init)
    pt.launch_kernel(acc.ptr, rgba.ptr, cam, p, spp)
    pt.sync()
    a = acc.download(np.float32, (H, W, 3))
    r = rgba.download(np.uint32, (H, W))
    acc.free()
    rgba.free()
    return a, r


def gpu_trace(pt, rays, cull=True):
    n = len(rays)
    d_r = pt.malloc(rays.nbytes)
    d_r.upload(rays)
    d_t, d_i, d_n = pt.malloc(4 * n), pt.malloc(4 * n), pt.malloc(12 * n)
    pt.trace_rays(d_r.ptr, n, cull, d_t.ptr, d_i.ptr, d_n.ptr)
    pt.sync()
    out = d_t.download(np.float32, (n,)), d_i.download(np.int32, (n,)), d_n.download(np.float32, (n, 3))
    for b in (d_r, d_t, d_i, d_n):
        b.free()
    return out


def gpu_trace_frame_free(*a, **k):  # re-export hook for the sibling test modules
    return gpu_trace(*a, **k)


# ---------------------------------------------------------------- rows a5-a7: closest hit
@pytest.mark.parametrize("name,n", [("cornell", 200_000), ("gto_sixteen", 200_000), ("dragon", 200_000), ("cube", 50_000)])
@pytest.mark.parametrize("cull", [True, False])
def test_closest_hit_bit_exact_vs_oracle(pt, name, n, cull):
    mesh, bvh = bvh_of(name)
    pt.upload_bvh(bvh)
    lo, hi = mesh.bounds()
    rays = orc.random_rays(n, lo, hi, seed=21)
    t, tri, nrm = gpu_trace(pt, rays, cull)
    t0, tri0, nrm0, _ = orc.trace_bvh(bvh, rays, cull)
    assert (t0 < 1e30).sum() > n // 100
    assert np.array_equal(t, t0)
    assert np.array_equal(tri, tri0)
    assert np.array_equal(nrm, nrm0)


def test_closest_hit_vs_bruteforce_and_reference_fixture(pt):
    """GPU vs brute force over every triangle, and vs the distances the REFERENCE's CPU
    intersector produced (tests/golden/ref_primary_hits.npz)."""
    mesh, bvh = bvh_of("dragon")
    pt.upload_bvh(bvh)
    lo, hi = mesh.bounds()
    rays = orc.random_rays(3000, lo, hi, seed=33)
    t, tri, _ = gpu_trace(pt, rays, True)
    tb, ib, _ = orc.trace_brute(mesh, rays, True)
    assert ((t != tb) | (tri != ib)).sum() <= 1
    z = np.load(os.path.join(GOLD, "ref_primary_hits.npz"))
    rays, t_ref = z["dragon_rays"], z["dragon_t_ref"]
    t, tri, _ = gpu_trace(pt, rays, False)
    hit, hit_r = t < 1e30, t_ref < 1e30
    assert (hit != hit_r).sum() <= 2
    both = hit & hit_r
    assert (np.abs(t[both] - t_ref[both]) / t_ref[both]).max() < 1e-4


def test_edge_case_rays(pt):
    """Zero direction components (ooeps substitution), rays starting inside boxes, rays that
    miss everything, axis-parallel rays along box faces.  The last ray lies EXACTLY in a
    bounding plane with a zero direction component: whether a slab test keeps such a ray is a
    property of the individual boxes, so this case is checked on the producer's own tree
    (PT_OPT_LEAF_MAX 0), the tree the oracle walks."""
    mesh, bvh = bvh_of("cornell")
    pt.set_option(g.OPT_LEAF_MAX, 0)
    pt.upload_bvh(bvh)
    pt.set_option(g.OPT_LEAF_MAX, 2)
    o = np.array([[0, 0, 0], [1, 2, 0], [0, 0, -40], [0, 0, -40], [0, 0, -40], [100, 100, 100], [0, 15.850145, -40]], np.float32)
    d = np.array([[0, 0, -1], [0, 0, -1], [0, -1, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 0, 0]], np.float32)
    rays = np.zeros((len(o), 8), np.float32)
    rays[:, 0:3], rays[:, 4:7] = o, d
    for cull in (True, False):
        t, tri, nrm = gpu_trace(pt, rays, cull)
        t0, tri0, nrm0, _ = orc.trace_bvh(bvh, rays, cull)
        assert np.array_equal(t, t0) and np.array_equal(tri, tri0) and np.array_equal(nrm, nrm0)


def test_spatial_split_tree_and_root_leaf(pt):
    mesh, bvh = bvh_of("gto_sixteen", split_alpha=1e-5)
    pt.upload_bvh(bvh)
    lo, hi = mesh.bounds()
    rays = orc.random_rays(50_000, lo, hi, seed=4)
    t, tri, _ = gpu_trace(pt, rays)
    t0, tri0, _, _ = orc.trace_bvh(bvh, rays)
    assert np.array_equal(t, t0) and np.array_equal(tri, tri0)
    one = g.Mesh.from_arrays(np.array([[0, 0, -5], [1, 0, -5], [0, 1, -5]], np.float32), [[0, 1, 2]])
    b1 = g.Bvh(one)
    pt.upload_bvh(b1)
    rays = orc.random_rays(10_000, [-1, -1, -6], [2, 2, 1], seed=9)
    t, tri, _ = gpu_trace(pt, rays, False)
    t0, tri0, _, _ = orc.trace_bvh(b1, rays, False)
    assert (tri0 == 0).sum() > 20
    assert np.array_equal(t, t0) and np.array_equal(tri, tri0)


# ---------------------------------------------------------------- rows a1-a4, a8-a10: the frame
@pytest.mark.parametrize("mat", [g.MAT_DIFF, g.MAT_METAL, g.MAT_SPEC, g.MAT_REFR])
def test_golden_images(pt, mat):
    """Committed 64x64 fixtures (oracle output) for every material, 1/4/16 spp."""
    z = np.load(os.path.join(GOLD, "oracle_images.npz"))
    name = {g.MAT_DIFF: "diff", g.MAT_METAL: "metal", g.MAT_SPEC: "spec", g.MAT_REFR: "refr"}[mat]
    _, bvh = bvh_of("cornell")
    sph = g.reference_spheres()
    cam = golden_camera(64, 64)
    for spp in (1, 4, 16):
        p = g.default_params(64, 64, tri_mat=mat)
        p.flags = g.FLAG_WRITE_RGBA
        acc, rgba = gpu_render(pt, bvh, sph, cam, p, spp)
        assert l2(acc, z[f"{name}_{spp}"]) < L2_TOL
        assert np.array_equal(acc, z[f"{name}_{spp}"]), (name, spp)
        assert np.array_equal(rgba, z[f"{name}_{spp}_rgba"])


def test_golden_metal_literal_and_open_box(pt):
    z = np.load(os.path.join(GOLD, "oracle_images.npz"))
    _, bvh = bvh_of("cornell")
    cam = golden_camera(64, 64)
    p = g.default_params(64, 64, tri_mat=g.MAT_METAL)
    p.flags = g.FLAG_METAL_LITERAL_W
    acc, _ = gpu_render(pt, bvh, g.reference_spheres(), cam, p, 4)
    assert np.array_equal(acc, z["metal_literal_4"])
    acc, _ = gpu_render(pt, bvh, None, cam, g.default_params(64, 64), 4)   # misses return bkColor
    assert np.array_equal(acc, z["nospheres_4"])


def test_config2_cornell_720p_diffuse(pt):
    """BASELINE.json configs[1]: cornell.obj 1280x720 4-bounce diffuse-only, 1 MI355X —
    with and without the sphere room."""
    W, H = 1280, 720
    _, bvh = bvh_of("cornell")
    cam = g.default_camera(W, H)
    for sph in (g.reference_spheres(), None):
        p = g.default_params(W, H)
        acc, _ = gpu_render(pt, bvh, sph, cam, p, 1)
        ref, _, _ = orc.render(bvh, sph, cam, p, 1, want_rgba=False)
        n_diff = int(np.any(acc != ref, axis=-1).sum())
        assert l2(acc, ref) < L2_TOL
        assert n_diff == 0, n_diff


@pytest.mark.parametrize("mat", [g.MAT_DIFF, g.MAT_METAL, g.MAT_SPEC])
def test_config3_cornell_dragon_1080p(pt, mat):
    """BASELINE.json configs[2] at full size (100 032-triangle variant): one progressive frame,
    every pixel compared with the oracle."""
    W, H = 1920, 1080
    _, bvh = bvh_of("cornell_dragon")
    sph = g.reference_spheres()
    cam = g.default_camera(W, H)
    p = g.default_params(W, H, tri_mat=mat)
    p.frame, p.sample_index = 3, 1
    acc, _ = gpu_render(pt, bvh, sph, cam, p, 1)
    ref, _, _ = orc.render(bvh, sph, cam, p, 1, want_rgba=False)
    n_diff = int(np.any(acc != ref, axis=-1).sum())
    assert l2(acc, ref) < L2_TOL
    assert n_diff == 0, n_diff


def test_spp_semantics_and_progressive_state(pt):
    """spp samples in one call == spp single-sample calls (constantPdf 1..spp,
    BasicScene.cpp:399); a call with sample_index > 1 continues from the accum buffer."""
    W, H = 160, 96
    _, bvh = bvh_of("cornell")
    sph = g.reference_spheres()
    cam = golden_camera(W, H)
    p = g.default_params(W, H)
    one, _ = gpu_render(pt, bvh, sph, cam, p, 6)
    pt.upload_bvh(bvh)
    pt.upload_spheres(sph)
    acc, rgba = pt.alloc_frame(W, H)
    for s in range(6):
        q = g.default_params(W, H)
        q.frame, q.sample_index = s, 1 + s
        pt.launch_kernel(acc.ptr, rgba.ptr, cam, q, 1)
    pt.sync()
    step = acc.download(np.float32, (H, W, 3))
    assert np.array_equal(one, step)
    ref, _, _ = orc.render(bvh, sph, cam, p, 6, want_rgba=False)
    assert np.array_equal(one, ref)
    # garbage in the buffer must not leak through an N=1 launch
    acc.upload(np.full((H, W, 3), np.nan, np.float32))
    pt.launch_kernel(acc.ptr, rgba.ptr, cam, p, 1)
    pt.sync()
    assert np.isfinite(acc.download(np.float32, (H, W, 3))).all()


def test_ragged_sizes(pt):
    """Width/height not multiples of the 8x8 wave tile; tiny images."""
    _, bvh = bvh_of("cornell")
    sph = g.reference_spheres()
    for W, H in ((2, 2), (7, 5), (65, 33), (129, 71)):
        cam = golden_camera(W, H)
        p = g.default_params(W, H)
        acc, _ = gpu_render(pt, bvh, sph, cam, p, 2)
        ref, _, _ = orc.render(bvh, sph, cam, p, 2, want_rgba=False)
        assert np.array_equal(acc, ref), (W, H)


def test_tile_split_is_bit_identical(pt):
    """Row (e): N-way framebuffer partition merged == single render, bit for bit."""
    W, H = 200, 120
    _, bvh = bvh_of("gto_sixteen")
    sph = g.reference_spheres()
    cam = golden_camera(W, H)
    full, _ = gpu_render(pt, bvh, sph, cam, g.default_params(W, H), 2)
    for count, rows in ((2, 8), (3, 16), (8, 8)):
        acc, rgba = pt.alloc_frame(W, H)
        for part in range(count):
            p = g.default_params(W, H)
            p.part_index, p.part_count, p.part_rows = part, count, rows
            pt.launch_kernel(acc.ptr, rgba.ptr, cam, p, 2)
        pt.sync()
        merged = acc.download(np.float32, (H, W, 3))
        assert np.array_equal(full, merged), (count, rows)
        acc.free()
        rgba.free()


def test_counters_and_algorithmic_bytes(pt):
    """Instrumented launch: segment/hit counts equal the oracle's; node counts are >= the
    single-lane oracle's (lanes that wait for their wave keep descending, cudaUtils.h:383-394)."""
    W, H = 256, 144
    _, bvh = bvh_of("cornell_dragon")
    sph = g.reference_spheres()
    cam = golden_camera(W, H)
    p = g.default_params(W, H)
    pt.set_option(g.OPT_COUNTERS, 1)
    pt.set_option(g.OPT_LEAF_MAX, 0)     # walk the producer's own leaves, like the oracle
    try:
        acc, _ = gpu_render(pt, bvh, sph, cam, p, 2)
        c = pt.counters()
    finally:
        pt.set_option(g.OPT_COUNTERS, 0)
        pt.set_option(g.OPT_LEAF_MAX, 2)
    ref, _, c0 = orc.render(bvh, sph, cam, p, 2, want_rgba=False)
    assert np.array_equal(acc, ref)
    assert c["paths"] == c0["paths"] == W * H * 2
    assert c["rays"] == c0["rays"] and c["hits"] == c0["hits"]
    # the wave-coupled while-while walk only ever visits MORE than the single-lane oracle; the
    # unified-step walk tests a leaf as soon as it is popped and can visit slightly fewer
    assert 0.7 * c0["inner"] <= c["inner"] <= 2 * c0["inner"]
    assert 0.7 * c0["tris"] <= c["tris"] <= 2 * c0["tris"]
    if not pt.walk:
        assert c["inner"] >= c0["inner"] and c["tris"] >= c0["tris"]
    assert g.algorithmic_bytes(c0, len(sph)) > 0


def test_error_handling(pt):
    p = g.default_params(64, 64)
    cam = golden_camera(64, 64)
    acc, rgba = pt.alloc_frame(64, 64)
    fresh = g.PathTracer(0)
    with pytest.raises(g.PtError) as e:
        fresh.launch_kernel(acc.ptr, rgba.ptr, cam, p, 1)          # no scene yet
    assert e.value.code == -3
    fresh.close()
    bad = g.default_params(64, 64)
    bad.sample_index = 0
    with pytest.raises(g.PtError):
        pt.launch_kernel(acc.ptr, rgba.ptr, cam, bad, 1)
    with pytest.raises(g.PtError):
        pt.launch_kernel(acc.ptr, rgba.ptr, cam, p, 0)
    with pytest.raises(g.PtError):
        pt.launch_kernel(None, rgba.ptr, cam, p, 1)
    # corrupt BVH arrays are rejected at upload, never reach the kernel
    _, bvh = bvh_of("cornell")
    nodes = bvh.nodes.copy()
    nodes.view(np.int32)[3, 0] = 12345                                # not a multiple of 64
    with pytest.raises(g.PtError):
        pt.upload_bvh_arrays(nodes.ctypes.data, 16, bvh.tris.ctypes.data, 101, bvh.index.ctypes.data, 101)
    nodes = bvh.nodes.copy()
    nodes.view(np.int32)[3, 0] = ~5000                                # leaf past the array
    with pytest.raises(g.PtError):
        pt.upload_bvh_arrays(nodes.ctypes.data, 16, bvh.tris.ctypes.data, 101, bvh.index.ctypes.data, 101)


def test_bad_triangle_ids_and_oversized_calls_are_refused():
    """pt_upload_bvh rejects triangle ids the kernels cannot carry (-1 is the walks' miss marker, ids index the material
    rows); pt_render refuses width*height*spp >= 2^31 BEFORE touching its buffers, so the context keeps working."""
    _, bvh = bvh_of("cornell")
    t = g.PathTracer(0)
    try:
        bad = type("B", (), {})()
        bad.nodes, bad.tris, bad.index = bvh.nodes, bvh.tris, bvh.index.copy()
        first = int(np.flatnonzero(bad.index >= 0)[0])
        for v in (-1, -7, 1 << 30):
            bad.index[first] = v
            with pytest.raises(g.PtError) as e:
                t.upload_bvh(bad)
            assert e.value.code == -1 and "triangle id" in str(e.value)
        t.upload_bvh(bvh)
        t.upload_spheres(g.reference_spheres())
        W, H = 640, 360
        cam, p = golden_camera(W, H), g.default_params(W, H)
        acc, rgba = t.alloc_frame(W, H)
        t.launch_kernel(acc.ptr, rgba.ptr, cam, p, 2)
        t.sync()
        before = acc.download(np.float32, (H, W, 3))
        with pytest.raises(g.PtError) as e:
            t.launch_kernel(acc.ptr, rgba.ptr, cam, p, 20000)     # 640*360*20000 >= 2^31
        assert e.value.code == -1 and "too large" in str(e.value)
        acc.zero()
        t.launch_kernel(acc.ptr, rgba.ptr, cam, p, 2)             # the context is intact
        t.sync()
        assert np.array_equal(acc.download(np.float32, (H, W, 3)), before)
    finally:
        t.close()

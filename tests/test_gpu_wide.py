"""GPU tests of the WIDE walk (PT_OPT_WALK = 2: 4-way tree, 8-bit outward-rounded child boxes).

Parity class: leaves hold the exact triangle records, so every reported hit is the exact
kernels' (t, id, normal); the quantised boxes can only let MORE candidates through, which
matters only where the binary tree's own slab rounding culls a grazing true hit.  Bar:
north_star's per-pixel L2 < 1e-3 against the oracle, plus a stated cap on differing
pixels (they are counted and must stay at the 1e-5 level)."""
import os

import numpy as np
import pytest

import gpu_pathtracer_amd as g
import orc
from test_gpu_parity import gpu_render, golden_camera, l2, bvh_of

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module", params=[(g.KERNEL_PERSISTENT, 2), (g.KERNEL_MEGA_BVH2, 2), (g.KERNEL_PERSISTENT, 4), (g.KERNEL_WAVEFRONT, 2)],
                ids=["persistent-wide", "mega-wide", "persistent-wide-postponed-leaf", "wavefront-wide"])
def ptw(request):
    t = g.PathTracer(0)
    t.set_option(g.OPT_KERNEL, request.param[0])
    t.set_option(g.OPT_WALK, request.param[1])
    t._walk = request.param[1]
    yield t
    t.close()


@pytest.mark.parametrize("mat", [g.MAT_DIFF, g.MAT_METAL, g.MAT_SPEC, g.MAT_REFR])
def test_wide_golden_images(ptw, mat):
    z = np.load(os.path.join(GOLD, "oracle_images.npz"))
    name = {g.MAT_DIFF: "diff", g.MAT_METAL: "metal", g.MAT_SPEC: "spec", g.MAT_REFR: "refr"}[mat]
    _, bvh = bvh_of("cornell")
    for spp in (1, 16):
        p = g.default_params(64, 64, tri_mat=mat)
        p.flags = g.FLAG_WRITE_RGBA
        acc, rgba = gpu_render(ptw, bvh, g.reference_spheres(), golden_camera(64, 64), p, spp)
        assert l2(acc, z[f"{name}_{spp}"]) < 1e-3
        assert np.any(acc != z[f"{name}_{spp}"], axis=-1).sum() <= 1


@pytest.mark.parametrize("scene,mat", [("cornell_dragon", g.MAT_DIFF), ("cornell_dragon", g.MAT_METAL),
                                       ("gto_sixteen", g.MAT_SPEC), ("cornell_dragon_800k", g.MAT_DIFF)])
def test_wide_full_frame_vs_oracle(ptw, scene, mat):
    """1920x1080, depth 4, sphere room: every pixel against the oracle."""
    W, H = 1920, 1080
    _, bvh = bvh_of(scene)
    sph = g.reference_spheres()
    cam = g.default_camera(W, H)
    p = g.default_params(W, H, tri_mat=mat)
    p.frame = 5
    acc, _ = gpu_render(ptw, bvh, sph, cam, p, 1)
    ref, _, _ = orc.render(bvh, sph, cam, p, 1, want_rgba=False)
    n_diff = int(np.any(acc != ref, axis=-1).sum())
    err = l2(acc, ref)
    print(f"wide {scene} mat {mat}: L2 {err:.3e}, differing pixels {n_diff} of {W * H}, scene {ptw.scene_info()}")
    assert err < 1e-3
    assert n_diff <= 40          # 2e-5 of the pixels


def test_wide_open_scene_and_ragged(ptw):
    """No spheres (paths die on a miss, lanes refill), ragged image sizes, several spp."""
    _, bvh = bvh_of("dragon")
    for W, H in ((129, 71), (640, 360)):
        cam = golden_camera(W, H)
        p = g.default_params(W, H)
        acc, _ = gpu_render(ptw, bvh, None, cam, p, 3)
        ref, _, _ = orc.render(bvh, None, cam, p, 3, want_rgba=False)
        assert l2(acc, ref) < 1e-3
        assert np.any(acc != ref, axis=-1).sum() <= 2


def test_wide_equals_exact_kernel_on_hits(ptw):
    """Same frame through the exact unified walk and the wide walk on one device; the
    instrumented counters show the wide walk fetches far fewer node items."""
    W, H = 960, 540
    _, bvh = bvh_of("cornell_dragon")
    sph = g.reference_spheres()
    cam = golden_camera(W, H)
    p = g.default_params(W, H)
    ptw.set_option(g.OPT_COUNTERS, 1)
    a_w, _ = gpu_render(ptw, bvh, sph, cam, p, 2)
    c_w = ptw.counters()
    ptw.set_option(g.OPT_WALK, 1)
    try:
        a_e, _ = gpu_render(ptw, bvh, sph, cam, p, 2)
        c_e = ptw.counters()
    finally:
        ptw.set_option(g.OPT_WALK, ptw._walk)
        ptw.set_option(g.OPT_COUNTERS, 0)
    print("wide counters", c_w, "exact counters", c_e)
    assert np.any(a_w != a_e, axis=-1).sum() <= 10
    assert c_w["rays"] == c_e["rays"] and c_w["hits"] >= c_e["hits"] - 10
    assert c_w["inner"] < 0.75 * c_e["inner"]


def test_config5_open_scene_4096_tile_split(ptw):
    """BASELINE.json configs[4] flavour: dragon.obj, 4096x4096, no sphere room (paths die on the
    first miss, so lanes are refilled constantly), rendered as an 8-way stripe split on one GPU
    and compared, every pixel, with the oracle and with the unsplit render."""
    W = H = 4096
    _, bvh = bvh_of("dragon")
    cam = g.default_camera(W, H)
    cam.dist = 18.0   # the reference's dist = H/60 = 68 would start every ray BEHIND the dragon
    ptw.upload_bvh(bvh)
    ptw.upload_spheres(None)
    acc, rgba = ptw.alloc_frame(W, H)
    for part in range(8):
        p = g.default_params(W, H)
        p.part_index, p.part_count, p.part_rows = part, 8, 8
        ptw.launch_kernel(acc.ptr, rgba.ptr, cam, p, 1)
    ptw.sync()
    split = acc.download(np.float32, (H, W, 3))
    acc.zero()
    p = g.default_params(W, H)
    ptw.launch_kernel(acc.ptr, rgba.ptr, cam, p, 1)
    ptw.sync()
    whole = acc.download(np.float32, (H, W, 3))
    acc.free()
    rgba.free()
    assert np.array_equal(split, whole)
    ref, _, cnt = orc.render(bvh, None, cam, p, 1, want_rgba=False)
    assert W * H * 1.02 < cnt["rays"] < W * H * p.depth   # open scene: paths end on their first miss
    n_diff = int(np.any(whole != ref, axis=-1).sum())
    print(f"config5 4096^2 open scene: segments {cnt['rays']}, differing pixels {n_diff}")
    assert l2(whole, ref) < 1e-3
    assert n_diff <= 64


# ---------------------------------------------------------------- Woop records (PT_OPT_TRI_TEST 1)
@pytest.fixture(scope="module")
def ptwoop():
    t = g.PathTracer(0)
    t.set_option(g.OPT_TRI_TEST, 1)
    yield t
    t.close()


@pytest.mark.parametrize("scene,mat,spp,tol", [("cornell", g.MAT_DIFF, 16, 1e-3), ("cornell_dragon", g.MAT_DIFF, 8, 1e-3),
                                               ("cornell_dragon", g.MAT_METAL, 16, 1e-3), ("gto_sixteen", g.MAT_REFR, 4, 1e-3),
                                               ("cornell_dragon", g.MAT_DIFF, 1, 4e-3)])
def test_woop_records_within_tolerance(ptwoop, scene, mat, spp, tol):
    """north_star asks for Woop's test; the reference kernel runs Moller-Trumbore and
    so does the oracle.  Woop's t differs from Moller-Trumbore's in the last bits, so every later
    bounce is perturbed at the 1e-7 level and about one edge-grazing path per 500k flips to a
    different surface.  Tolerance-class parity: north_star's per-pixel L2 < 1e-3 holds from a few
    spp on; a 1-spp frame cannot meet it in general — ONE flipped pixel out of 518 400 is already
    L2 = 1.3e-3 — so that case bounds the flipped pixels and L2 < 4e-3.  (This is why Woop records
    are an option and the bit-exact Moller-Trumbore records are the default.)"""
    W, H = 960, 540
    _, bvh = bvh_of(scene)
    sph = g.reference_spheres()
    cam = golden_camera(W, H)
    p = g.default_params(W, H, tri_mat=mat)
    acc, _ = gpu_render(ptwoop, bvh, sph, cam, p, spp)
    ref, _, _ = orc.render(bvh, sph, cam, p, spp, want_rgba=False)
    err = l2(acc, ref)
    big = int((np.abs(acc - ref).max(axis=-1) > 1e-2).sum())
    print(f"woop {scene} mat {mat} spp {spp}: L2 {err:.3e}, pixels off by > 1e-2: {big} of {W * H}")
    assert err < tol
    assert big <= W * H // 10000       # 1e-4 of the pixels


def test_woop_ray_batch_is_refused(ptwoop):
    _, bvh = bvh_of("cornell")
    ptwoop.upload_bvh(bvh)
    buf = ptwoop.malloc(1024)
    with pytest.raises(g.PtError) as e:
        ptwoop.trace_rays(buf.ptr, 4, True, buf.ptr, buf.ptr, None)
    assert e.value.code == -5
    buf.free()


def test_wave_statistics_are_consistent():
    """pt_get_wave_stats (instrumented persistent wide walks): the schedule counters add up."""
    _, bvh = bvh_of("cornell_dragon")
    W, H = 640, 360
    cam, p = g.default_camera(W, H), g.default_params(W, H)
    sph = g.reference_spheres()
    for walk in (2, 4):
        t = g.PathTracer(0)
        try:
            t.set_option(g.OPT_WALK, walk)
            t.set_option(g.OPT_COUNTERS, 1)
            acc, _ = gpu_render(t, bvh, sph, cam, p, 2)
            c, w = t.counters(), t.wave_stats()
            print(f"walk {walk}: {c} {w}")
            assert c["paths"] == 2 * W * H and c["rays"] == 4 * c["paths"]       # closed room, depth 4
            assert w["act_node"] == c["inner"]                                    # one active lane per node visit
            assert w["act_shade"] == c["rays"] and w["act_begin"] == c["paths"]
            for ph in ("node", "rec", "shade", "begin"):
                assert 0 < w["act_" + ph] <= 64 * w["it_" + ph]
            assert w["it_loop"] >= w["it_shade"]
            if walk == 2:
                assert w["act_rec"] <= c["tris"]      # the leaf loop tests further records inside one record step
            else:
                assert w["act_rec"] == c["tris"]
        finally:
            t.close()

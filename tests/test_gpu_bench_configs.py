"""GPU parity at the configurations bench.py TIMES, with the library's DEFAULT kernel (no option set):
the benchmarked path is the tested path.

BASELINE.json configs[2] (cornell_dragon ~800k, 1920x1080, diffuse / metal / specular), the bench
step itself (16 spp folded by ONE pt_render call: sample buffer + fold order of tracer.cu:386-391
at full size) and configs[4] (dragon 4096x4096 at 8 spp).  Checker: the CPU oracle on the same
seeded inputs.  Bar: north_star's per-pixel L2 < 1e-3 AND a stated cap on differing pixels (the wide
walk's outward-rounded boxes may let a grazing candidate through that the binary tree culls; measured 0).
"""
import numpy as np
import pytest

import gpu_pathtracer_amd as g
import orc
from test_gpu_parity import gpu_render, golden_camera, l2, bvh_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["default", "cheaper-tree", "wavefront", "persistent"])
def ptd(request):
    """default = whatever PT_KERNEL_AUTO picks; cheaper-tree = the same with PT_OPT_OPTIMIZE 4 + PT_OPT_REBUILD 2 over a host tree
    built without spatial splits, which is what bench.py times (the upload optimises the caller's hierarchy by re-insertion,
    re-clusters the triangles on the device and optimises that too, and keeps whichever hierarchy costs fewer node visits);
    the two stage layouts named too."""
    t = g.PathTracer(0)
    if request.param == "wavefront":
        t.set_option(g.OPT_KERNEL, g.KERNEL_WAVEFRONT)
    elif request.param == "persistent":
        t.set_option(g.OPT_KERNEL, g.KERNEL_PERSISTENT)
    elif request.param == "cheaper-tree":
        t.set_option(g.OPT_OPTIMIZE, 4)
        t.set_option(g.OPT_REBUILD, 2)
    t.variant = request.param
    # bench.py's host tree is built without spatial splits (with PT_OPT_OPTIMIZE they no longer pay on its scene): the variant that
    # mirrors bench.py uploads THAT tree; the oracle walks the default host tree (closest hits do not depend on the tree)
    t.upload_tree = (lambda name: bvh_of(name, split_alpha=-1.0)[1]) if request.param == "cheaper-tree" else (lambda name: bvh_of(name)[1])
    yield t
    t.close()


_oracle_cache = {}


def oracle(key, fn):
    """one oracle render per configuration, shared by the kernel variants"""
    if key not in _oracle_cache:
        _oracle_cache[key] = fn()
    return _oracle_cache[key]


def check(acc, ref, what, max_diff):
    n_diff = int(np.any(acc != ref, axis=-1).sum())
    err = l2(acc, ref)
    print(f"{what}: L2 {err:.3e}, differing pixels {n_diff} of {acc.shape[0] * acc.shape[1]}")
    assert err < 1e-3
    assert n_diff <= max_diff


def test_bench_step_16spp_800k_full_frame(ptd):
    """THE bench step: cornell_dragon_800k, 1920x1080, depth 4, diffuse + sphere room, 16 spp in ONE call,
    continuing a running mean (sample_index 17: the fold reads the accumulator of an earlier step)."""
    W, H, spp = 1920, 1080, 16
    _, bvh = bvh_of("cornell_dragon_800k")
    sph = g.reference_spheres()
    cam = g.default_camera(W, H)
    p = g.default_params(W, H)
    p.flags = g.FLAG_WRITE_RGBA
    p.frame, p.sample_index = 16, 17
    rng = np.random.default_rng(5)
    prev = rng.random((H, W, 3), dtype=np.float32)
    ref, ref_rgba, cnt = oracle("step16", lambda: orc.render(bvh, sph, cam, p, spp, accum=prev.copy()))
    acc, rgba = gpu_render(ptd, ptd.upload_tree("cornell_dragon_800k"), sph, cam, p, spp, accum_init=prev)
    assert cnt["rays"] == W * H * spp * p.depth          # the closed room: every path runs all four segments
    check(acc, ref, f"[{ptd.variant}] 800k diffuse 16 spp/call", 40)
    assert int((rgba != ref_rgba).sum()) <= 40


@pytest.mark.parametrize("mat", [g.MAT_METAL, g.MAT_SPEC], ids=["metal", "specular"])
def test_800k_metal_and_specular_full_frame(ptd, mat):
    """configs[2]: the two other triangle materials bench.py times on the 800k scene."""
    W, H, spp = 1920, 1080, 2
    _, bvh = bvh_of("cornell_dragon_800k")
    sph = g.reference_spheres()
    cam = g.default_camera(W, H)
    p = g.default_params(W, H, tri_mat=mat)
    p.frame = 3
    ref, _, _ = oracle(("mat", mat), lambda: orc.render(bvh, sph, cam, p, spp, want_rgba=False))
    acc, _ = gpu_render(ptd, ptd.upload_tree("cornell_dragon_800k"), sph, cam, p, spp)
    check(acc, ref, f"[{ptd.variant}] 800k mat {mat} {spp} spp", 40)


def test_config5_dragon_4096_8spp(ptd):
    """configs[4]: dragon.obj, 4096x4096, 8 spp in one call, open scene (paths end on their first miss:
    the stage-split pipeline compacts them away between bounces, the persistent kernel refills lanes)."""
    W = H = 4096
    spp = 8
    _, bvh = bvh_of("dragon")
    cam = g.default_camera(W, H)
    cam.dist = 18.0   # the reference's dist = H/60 = 68 would start every ray BEHIND the dragon
    p = g.default_params(W, H)
    ref, _, cnt = oracle("config5", lambda: orc.render(bvh, None, cam, p, spp, want_rgba=False))
    acc, _ = gpu_render(ptd, bvh, None, cam, p, spp)
    assert W * H * spp * 1.02 < cnt["rays"] < W * H * spp * p.depth
    check(acc, ref, f"[{ptd.variant}] config5 dragon 4096^2 8 spp ({cnt['rays']} segments)", 256)


def test_one_spp_calls_equal_one_multi_spp_call(ptd):
    """render(accum, bvh, camera, spp) == spp launches of the reference (BasicScene.cpp:395-404), bit for bit,
    at the bench resolution on the 100k scene."""
    W, H, spp = 1920, 1080, 4
    _, bvh = bvh_of("cornell_dragon")
    sph = g.reference_spheres()
    cam = g.default_camera(W, H)
    p = g.default_params(W, H)
    one, _ = gpu_render(ptd, bvh, sph, cam, p, spp)
    ptd.upload_spheres(sph)
    acc, rgba = ptd.alloc_frame(W, H)
    for s in range(spp):
        q = g.Params.from_buffer_copy(p)
        q.frame, q.sample_index = p.frame + s, p.sample_index + s
        ptd.launch_kernel(acc.ptr, rgba.ptr, cam, q, 1)
    ptd.sync()
    many = acc.download(np.float32, (H, W, 3))
    acc.free()
    rgba.free()
    assert np.array_equal(one, many)


@pytest.mark.parametrize("kernel", [g.KERNEL_WAVEFRONT, g.KERNEL_PERSISTENT], ids=["wavefront", "persistent"])
def test_wave_sample_groups_change_no_pixel(kernel):
    """PT_OPT_WAVE_SAMPLES decides which lane of the stage-split pipeline / the persistent kernel traces which (pixel, sample) — 16 samples of a 2x2 pixel
    block per wave by default, a whole 8x8 tile of one sample with 1 — and which layout the sample buffer has; the accumulator and
    the display words must not depend on it: ragged image, a tile split, spp that 16 / 8 / 4 divide and one that nothing does,
    next-event estimation (its shadow records carry the sample number), a second call on top of the first (running mean)."""
    _, bvh = bvh_of("cornell_dragon")
    sph = g.reference_spheres()
    for (W, H), spp, flags, parts in (((333, 187), 16, 0, 1), ((640, 360), 8, 0, 3), ((640, 360), 12, g.FLAG_NEE | g.FLAG_COSINE_DIFF, 1),
                                      ((333, 187), 5, 0, 1), ((256, 256), 64, 0, 1)):
        cam = golden_camera(W, H)
        frames = {}
        for cap in (1, 4, 16, 64):
            t = g.PathTracer(0)
            try:
                t.set_option(g.OPT_KERNEL, kernel)
                t.set_option(g.OPT_WAVE_SAMPLES, cap)
                t.upload_bvh(bvh)
                t.upload_spheres(sph)
                acc, rgba = t.alloc_frame(W, H)
                for call in range(2):
                    for part in range(parts):
                        p = g.default_params(W, H)
                        p.flags = flags | g.FLAG_WRITE_RGBA
                        p.frame, p.sample_index = 7 + call * spp, 1 + call * spp
                        p.part_index, p.part_count, p.part_rows = part, parts, 8
                        t.launch_kernel(acc.ptr, rgba.ptr, cam, p, spp)
                t.sync()
                frames[cap] = (acc.download(np.float32, (H, W, 3)), rgba.download(np.uint32, (H, W)))
                acc.free()
                rgba.free()
            finally:
                t.close()
        for cap in (4, 16, 64):
            assert np.array_equal(frames[cap][0], frames[1][0]), (W, H, spp, cap)
            assert np.array_equal(frames[cap][1], frames[1][1]), (W, H, spp, cap)
        assert frames[1][0].any()


def test_auto_times_both_layouts_and_keeps_one():
    """PT_KERNEL_AUTO: the first four calls of a configuration are the timed trials (persistent kernel, stage-split
    pipeline, and both once more: the faster trial of each counts), later ones run the faster layout; every call gives the
    same image.  The library keeps a table of configurations (keyed by the partition's SHAPE, not by which part a call renders)."""
    W, H, spp = 1280, 720, 8
    _, bvh = bvh_of("cornell_dragon")
    sph = g.reference_spheres()
    cam = g.default_camera(W, H)
    p = g.default_params(W, H)
    t = g.PathTracer(0)
    try:
        t.upload_bvh(bvh)
        t.upload_spheres(sph)
        acc, rgba = t.alloc_frame(W, H)
        frames = []
        for i in range(6):
            acc.zero()
            t.launch_kernel(acc.ptr, rgba.ptr, cam, p, spp)
            t.sync()
            frames.append(acc.download(np.float32, (H, W, 3)))
            k, ms_p, ms_w = t.auto_choice()
            if i < 3:
                assert k == g.KERNEL_AUTO          # the last trial has not run yet
        assert k in (g.KERNEL_PERSISTENT, g.KERNEL_WAVEFRONT) and ms_p > 0 and ms_w > 0
        print(f"auto: persistent {ms_p:.3f} ms, wavefront {ms_w:.3f} ms -> {'wavefront' if k == g.KERNEL_WAVEFRONT else 'persistent'}")
        for f in frames[1:]:
            assert np.array_equal(f, frames[0])
        # another configuration starts over ...
        t.launch_kernel(acc.ptr, rgba.ptr, cam, p, 1)
        t.sync()
        assert t.auto_choice()[0] == g.KERNEL_AUTO
        # ... and the first one is remembered
        t.launch_kernel(acc.ptr, rgba.ptr, cam, p, spp)
        t.sync()
        assert t.auto_choice() == (k, ms_p, ms_w)
        # the parts of a tile split share one decision: the four trials are spread over the four parts
        q = g.Params.from_buffer_copy(p)
        q.part_count, q.part_rows = 4, 8
        for part in range(4):
            q.part_index = part
            t.launch_kernel(acc.ptr, rgba.ptr, cam, q, spp)
            t.sync()
            assert (t.auto_choice()[0] == g.KERNEL_AUTO) == (part < 3)
    finally:
        t.close()


def arbitrate(pt, mesh, bvh, sph, cam, p, frames, what, max_diff, oracle_key=None):
    """Sample by sample (one pt_render per frame, N = 1) against the oracle's walk over `bvh`; every (frame, pixel) where the
    two differ is replayed with the BRUTE-FORCE closest hit (orc.sample_pixels over the raw triangles: no tree, so no box
    can cull anything) and the GPU must hold brute force's colour.  Returns (differing, of which the oracle's walk was off)."""
    W, H = p.width, p.height
    acc, rgba = pt.alloc_frame(W, H)
    diffs = []
    for f in frames:
        q = g.Params.from_buffer_copy(p)
        q.frame, q.sample_index = f, 1
        pt.launch_kernel(acc.ptr, rgba.ptr, cam, q, 1)
        pt.sync()
        got = acc.download(np.float32, (H, W, 3))
        ref = oracle((oracle_key, f), lambda: orc.render(bvh, sph, cam, q, 1, want_rgba=False)[0]) if oracle_key else \
            orc.render(bvh, sph, cam, q, 1, want_rgba=False)[0]
        ys, xs = np.nonzero(np.any(got != ref, axis=-1))
        diffs += [(f, int(x), int(y), got[y, x].copy(), ref[y, x].copy()) for x, y in zip(xs, ys)]
    acc.free()
    rgba.free()
    n_oracle_off = 0
    for f, x, y, got, ref in diffs:
        q = g.Params.from_buffer_copy(p)
        q.frame, q.sample_index = f, 1
        col, t_b, id_b = orc.sample_pixels([(x, y)], sph, cam, q, 1, mesh=mesh)
        _, t_o, id_o = orc.sample_pixels([(x, y)], sph, cam, q, 1, bvh=bvh)
        brute = orc.fold_samples(col, 1)[0]
        seg = int(np.argmax((t_b[0, 0] != t_o[0, 0]) | (id_b[0, 0] != id_o[0, 0]))) if (np.any(t_b != t_o) or np.any(id_b != id_o)) else -1
        print(f"  {what}: frame {f} pixel ({x},{y}) gpu {got} oracle {ref} brute {brute}; oracle's walk leaves brute force at segment {seg}: "
              f"t {t_o[0, 0, seg] if seg >= 0 else None} id {id_o[0, 0, seg] if seg >= 0 else None} vs t {t_b[0, 0, seg] if seg >= 0 else None} id {id_b[0, 0, seg] if seg >= 0 else None}")
        assert np.array_equal(got, brute), f"{what}: GPU differs from the brute-force arbiter at frame {f} pixel ({x},{y})"
        n_oracle_off += int(not np.array_equal(ref, brute))
    print(f"{what}: {len(diffs)} differing (frame, pixel) pairs of {len(frames) * W * H}, all equal to brute force on the GPU side; "
          f"the oracle's binary walk was the one off in {n_oracle_off}")
    assert len(diffs) <= max_diff
    return len(diffs), n_oracle_off


def test_bench_step_differing_pixels_are_brute_force_hits(ptd):
    """The 16 frames of the bench step, one by one: wherever the GPU's sample differs from the oracle's (the wide walk's
    outward-rounded boxes keep a grazing candidate that the binary tree's slab rounding culls), the GPU's colour is the
    brute-force renderer's — the GPU is the side that found the true closest hit."""
    W, H = 1920, 1080
    mesh, bvh = bvh_of("cornell_dragon_800k")
    sph = g.reference_spheres()
    cam = g.default_camera(W, H)
    p = g.default_params(W, H)
    ptd.upload_bvh(ptd.upload_tree("cornell_dragon_800k"))
    ptd.upload_spheres(sph)
    # frames 80..95: the first timed step of `bench.py --steps 20 --warmup 5` (what the round-end driver runs), where round 2's
    # in-run parity counted 2 differing pixels of 2 073 600
    arbitrate(ptd, mesh, bvh, sph, cam, p, range(80, 96), f"[{ptd.variant}] 800k bench frames", 64, oracle_key="arb800k")


def test_rebuilt_800k_tree_against_uploaded_and_brute_force():
    """PT_OPT_REBUILD 1 at 800k triangles (tests/test_gpu_build.py covers <= 100k bit for bit): the re-clustered tree's
    frames against the oracle over the HOST tree; pixels may differ only where a ray grazes a bounding plane, and each such
    pixel must be the brute-force hit."""
    W, H = 1920, 1080
    mesh, bvh = bvh_of("cornell_dragon_800k")
    sph = g.reference_spheres()
    cam = g.default_camera(W, H)
    p = g.default_params(W, H)
    t = g.PathTracer(0)
    try:
        t.set_option(g.OPT_REBUILD, 1)
        t.upload_bvh(bvh)
        t.set_option(g.OPT_REBUILD, 0)
        assert t.scene_info()["n_tri_refs"] == mesh.n_tris
        t.upload_spheres(sph)
        arbitrate(t, mesh, bvh, sph, cam, p, range(80, 88), "re-clustered 800k tree", 32, oracle_key="arb800k")
    finally:
        t.close()


def test_big_scene_6400k_parity():
    """bench.py's HBM-resident workload (cornell + 64 dragons, 6.4 M triangles, tree built ON the device and passed through the
    hierarchy optimiser, 0.75 GB of items):
    (1) 200k incoherent rays + the primary rays of a 480x270 frame: (t, id, normal) == the brute-force oracle bit for bit;
    (2) one 1920x1080 frame, 2 spp, against the oracle's walk over the HOST tree of the same mesh: L2 < 1e-3, and every
    differing pixel arbitrated by brute force."""
    from test_gpu_parity import gpu_trace
    mesh = g.scene_mesh("cornell_dragon_6400k")
    t = g.PathTracer(0)
    try:
        t.set_option(g.OPT_OPTIMIZE, 3)
        ms = t.build_bvh(mesh)
        t.set_option(g.OPT_OPTIMIZE, 0)
        info = t.scene_info()
        print(f"6400k: device build {ms:.1f} ms (+ PT_OPT_OPTIMIZE 3 on the host), {info}")
        assert info["device_bytes"] > 700e6 and info["n_tri_refs"] == mesh.n_tris
        lo, hi = mesh.bounds()
        cam = g.default_camera(480, 270)
        rays = np.concatenate([orc.random_rays(6000, lo, hi, seed=77), orc.primary_rays(cam, 480, 270, frame=3)[::40]])
        tg, ig, ng = gpu_trace(t, rays)
        tb, ib, nb = orc.trace_brute(mesh, rays)
        assert np.array_equal(tg, tb) and np.array_equal(ig, ib)
        hit = ib >= 0
        assert hit.mean() > 0.3 and np.array_equal(ng[hit], nb[hit])
        # the full frame
        W, H = 1920, 1080
        bvh = g.Bvh(mesh, split_alpha=-1.0)      # host SAH tree without spatial splits (a negative alpha turns them off): 6.4 M refs
        sph = g.reference_spheres()
        cam = g.default_camera(W, H)
        p = g.default_params(W, H)
        p.frame = 4
        t.upload_spheres(sph)
        acc, rgba = t.alloc_frame(W, H)
        t.launch_kernel(acc.ptr, rgba.ptr, cam, p, 2)
        t.sync()
        got = acc.download(np.float32, (H, W, 3))
        acc.free()
        rgba.free()
        ref, _, cnt = orc.render(bvh, sph, cam, p, 2, want_rgba=False)
        check(got, ref, "6400k device tree vs oracle over the host tree, 2 spp", 40)
        assert cnt["rays"] == W * H * 2 * p.depth
        # frames 7, 8: bench.py's in-run parity frame of this scene, where one pixel of 2 073 600 differs from the oracle's walk
        arbitrate(t, mesh, bvh, sph, cam, p, (4, 5, 7, 8), "6400k frames", 40)
    finally:
        t.close()

"""CPU tests (no GPU): host scene preparation, Compact-layout invariants, C-ABI surface."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import gpu_pathtracer_amd as g

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
GOLD = os.path.join(ROOT, "tests", "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.int32)


def check_compact_invariants(bvh, n_mesh_tris):
    """Layout contract of CudaBVH::createCompact (CudaBVH.cpp:161,184,201-212,221-224)."""
    nodes, tris, index = bvh.nodes, bvh.tris, bvh.index
    assert nodes.shape[1] == 4 and len(nodes) % 4 == 0           # every inner node = 4 vec4 = 64 B
    assert len(tris) == len(index)                               # parallel arrays
    links = bits(nodes)[3::4, :2]
    assert (bits(nodes)[3::4, 2:] == 0).all()
    inner = links[links >= 0]
    assert (inner % 64 == 0).all() and (inner < len(nodes) * 16).all() and (inner > 0).all()
    assert len(np.unique(inner)) == len(inner)                   # a tree: nobody referenced twice
    assert len(inner) == len(nodes) // 4 - 1                     # every node but the root has one parent
    seen = np.zeros(len(tris), bool)
    tb = bits(tris)
    n_refs = 0
    starts = []
    for link in links[links < 0]:
        a = ~int(link)
        assert 0 <= a < len(tris)
        while tb[a, 0] != np.int32(-2 ** 31):                    # 0x80000000 terminator
            assert not seen[a]
            seen[a:a + 3] = True
            assert (tris[a:a + 3, 3] == 0).all()                 # xyz0
            assert 0 <= index[a] < n_mesh_tris and index[a + 1] == 0 and index[a + 2] == 0
            n_refs += 1
            starts.append(a)
            a += 3
        assert (tb[a] == np.int32(-2 ** 31)).all() and index[a] == 0
        seen[a] = True
    assert seen.all()                                            # no orphan records
    assert n_refs == len(starts)
    return np.array(starts)


@pytest.mark.parametrize("name", ["cornell", "cube", "sphere", "bunny_low", "gto_sixteen"])
def test_flatten_invariants(name):
    mesh = g.scene_mesh(name)
    bvh = g.Bvh(mesh)
    recs = check_compact_invariants(bvh, mesh.n_tris)
    assert len(recs) == bvh.stats["n_tri_refs"] >= mesh.n_tris
    # every triangle is referenced, and the record holds its three vertices verbatim
    assert set(bvh.index[recs].tolist()) == set(range(mesh.n_tris))
    v, f = mesh.verts, mesh.tris
    for a in recs[:200]:
        assert np.array_equal(bvh.tris[a:a + 3, :3], v[f[bvh.index[a]]])


def test_cornell_matches_reference_counts_and_fixture():
    """The reference's own builder+flatten gives 16 node vec4 / 101 tri vec4 / 5 leaves for
    cornell.obj (SURVEY.md §8c probe); so does ours, and the arrays are the committed fixture."""
    bvh = g.Bvh(g.scene_mesh("cornell"))
    assert bvh.nodes.shape == (16, 4) and bvh.tris.shape == (101, 4) and bvh.stats["n_leaves"] == 5
    z = np.load(os.path.join(GOLD, "cornell_compact.npz"))
    assert np.array_equal(bits(bvh.nodes), bits(z["nodes"]))
    assert np.array_equal(bits(bvh.tris), bits(z["tris"]))
    assert np.array_equal(bvh.index, z["index"])


def test_child_boxes_enclose_their_triangles():
    """Object splits only: a leaf box holds its triangles whole.  (With spatial splits a leaf box
    holds only the clipped part of a duplicated reference; that tree is checked against brute
    force in test_oracle.py.)"""
    mesh = g.scene_mesh("bunny_low")
    bvh = g.Bvh(mesh, split_alpha=-1.0)
    nodes, tb = bvh.nodes.reshape(-1, 16), bits(bvh.tris)
    for nd in nodes:
        for ci in range(2):
            link = int(np.float32(nd[12 + ci]).view(np.int32))
            if link >= 0:
                continue
            lo = np.array([nd[0 + 4 * ci], nd[2 + 4 * ci], nd[8 + 2 * ci]])
            hi = np.array([nd[1 + 4 * ci], nd[3 + 4 * ci], nd[9 + 2 * ci]])
            a = ~link
            while tb[a, 0] != np.int32(-2 ** 31):
                p = bvh.tris[a:a + 3, :3]
                assert (p >= lo - 1e-6).all() and (p <= hi + 1e-6).all()
                a += 3


def test_root_leaf_is_wrapped():
    """the reference asserts (CudaBVH.cpp:141) when the root is a leaf."""
    v = np.array([[0, 0, -5], [1, 0, -5], [0, 1, -5]], np.float32)
    mesh = g.Mesh.from_arrays(v, [[0, 1, 2]])
    bvh = g.Bvh(mesh)
    assert bvh.nodes.shape == (4, 4)
    check_compact_invariants(bvh, 1)
    assert bvh.stats["n_leaves"] == 2 and bvh.stats["n_tri_refs"] == 1


def test_obj_reader(tmp_path):
    """`v`/`f` records only, negative indices, v/vt/vn tokens, quads fan-triangulated,
    several `o` objects in one file (the reference asserts on that: utilfun.cpp:474)."""
    p = tmp_path / "two.obj"
    p.write_text("# comment\no first\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1 4//1\n"
                 "o second\nv 0 0 -1\nv 1 0 -1\nv 0 1 -1\nf -3 -2 -1\ns off\n")
    m = g.Mesh.load(str(p))
    assert m.n_verts == 7 and m.n_tris == 3
    assert m.tris.tolist() == [[0, 1, 2], [0, 2, 3], [4, 5, 6]]
    with pytest.raises(RuntimeError):
        g.Mesh.load(str(tmp_path / "missing.obj"))
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(RuntimeError):
        g.Mesh.load(str(bad))


def test_sbvh_statistics_track_the_reference_builder():
    """Default build params = the reference's (Platform costs 1:1, splitAlpha 1e-5).  The
    reference's own builder+flatten gives 18 124 node vec4 / 52 409 tri vec4 / 4 532 leaves on
    gto_sixteen.obj (SURVEY.md §8c probe); ours must land within a few percent."""
    b = g.Bvh(g.scene_mesh("gto_sixteen"))
    assert abs(len(b.nodes) - 18124) / 18124 < 0.05
    assert abs(len(b.tris) - 52409) / 52409 < 0.08
    assert abs(b.stats["n_leaves"] - 4532) / 4532 < 0.05


def test_scene_synthesis_is_deterministic():
    a, b = g.scene_mesh("cornell_dragon"), g.scene_mesh("cornell_dragon")
    assert a.n_tris == 100032
    assert np.array_equal(a.verts, b.verts) and np.array_equal(a.tris, b.tris)


def test_defaults_mirror_basicscene():
    """BasicScene.cpp:220-259."""
    cam = g.default_camera(1280, 720)
    assert cam.dist == 12.0 and abs(cam.aspect - 1280 / 720) < 1e-6 and cam.fov == 1.0
    assert g.default_camera(1920, 1080).dist == 18.0 and g.default_camera(256, 256).dist == 4.0
    assert list(cam.front) == [0, 0, -1] and list(cam.right) == [1, 0, 0] and list(cam.up) == [0, 1, 0]
    p = g.default_params(1280, 720)
    assert p.depth == 4 and p.cull_backfaces == 1 and list(p.bk_color) == [1, 1, 1] and p.tri_mat == g.MAT_DIFF
    s = g.reference_spheres()
    assert len(s) == 8 and C.sizeof(g.Sphere) == 44
    assert [x.mat for x in s].count(g.MAT_SPEC) == 1


# ---------------------------------------------------------------- C ABI surface
def test_ptmi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ptmi.h")).read()
    declared = set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"pt_ctx", "pt_counters"}  # "pt_counters (PT_OPT..." in a comment
    lib = g._abi.ptmi()
    bound = {n for n, _, _ in g._abi.PTMI_SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name)
    out = subprocess.check_output(["nm", "-D", "--defined-only", g._abi.PTMI_PATH]).decode()
    exported = set(re.findall(r" T (pt_[a-z_0-9]+)", out))
    assert declared <= exported
    assert lib.pt_abi_version() == 3


def test_ptmi_is_gfx950_code_object():


    blob = open(g._abi.PTMI_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"k_trace_mega_bvh2" in blob and b"k_trace_rays_bvh2" in blob



def test_struct_layouts_match_header():
    assert C.sizeof(g.Camera) == 64
    assert C.sizeof(g.Sphere) == 44
    assert C.sizeof(g.Params) == 104
    assert C.sizeof(g.Counters) == 48
    assert g.Params.frame.offset == 16 and g.Params.sample_index.offset == 24


def test_errors_are_codes_not_exits():
    """No GPU here: pt_create must fail with a code and a message (the reference exit(1)s)."""
    lib = g._abi.ptmi()
    ctx = C.c_void_p()
    n = lib.pt_device_count()
    if n > 0:
        pytest.skip("a GPU is present; the no-device path is covered on CPU-only hosts")
    rc = lib.pt_create(0, C.byref(ctx))
    assert rc < 0 and not ctx.value
    assert len(lib.pt_last_error(None)) > 0
    with pytest.raises(g.PtError):
        g.PathTracer(0)
    assert lib.pt_render(None, None, None, None, None, 1) == -1   # PT_ERR_INVALID, no crash
    assert lib.pt_destroy(None) == 0


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing in the package may import, include, link or
    load anything under oracle/ (and there is no CPU fallback path)."""
    pkg = os.path.join(ROOT, "g.p.u-pathtracer_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.lower().replace("vs oracle", "").replace("the oracle", "").replace(
                    "oracle/pt_oracle.c:bvh_intersect", ""), os.path.join(dp, f)
                assert "liborc" not in txt and "pt_oracle.h" not in txt
    for so in (g._abi.PTMI_PATH, g._abi.PTHOST_PATH):
        ldd = subprocess.check_output(["ldd", so]).decode()
        assert "liborc" not in ldd


def test_bench_and_entry_points_parse():
    """bench.py / __graft_entry__.py / tools: no syntax errors, the CLI answers --help without a GPU."""
    import ast
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for rel in ["bench.py", "__graft_entry__.py"] + [os.path.join("tools", f) for f in sorted(os.listdir(os.path.join(root, "tools"))) if f.endswith(".py")]:
        ast.parse(open(os.path.join(root, rel)).read(), rel)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "--gpus" in r.stdout and "--steps" in r.stdout and "--warmup" in r.stdout


def test_bench_pmc_csv_parsing(tmp_path):
    """bench.py's reader of rocprofv3 --pmc output: per-dispatch counter values of one kernel family, instrumented
    instantiations (first template argument true) and other kernels left out."""
    sys.path.insert(0, ROOT)
    import bench
    d = tmp_path / "FETCH_SIZE" / "host" / "1234"
    d.mkdir(parents=True)
    rows = ["Correlation_Id,Dispatch_Id,Agent_Id,Kernel_Name,Counter_Name,Counter_Value",
            '1,1,4,"void k_wf_extend<false, 8, 16, true>(KParams)",FETCH_SIZE,1000.0',
            '2,2,4,"void k_wf_extend<false, 8, 16, false>(KParams)",FETCH_SIZE,3000.0',
            '3,3,4,"void k_wf_extend<true, 8, 16, false>(KParams)",FETCH_SIZE,999999.0',
            '4,4,4,"void k_wf_shade<false, false, false, true>(KParams)",FETCH_SIZE,500.0',
            '5,5,4,"k_fold_samples(KParams)",FETCH_SIZE,7.0',
            '6,6,4,"void k_wf_extend<false, 8, 16, false>(KParams)",WRITE_SIZE,11.0']
    (d / "1234_counter_collection.csv").write_text("\n".join(rows) + "\n")
    assert bench.pmc_values(str(tmp_path / "FETCH_SIZE"), "FETCH_SIZE", "k_wf_extend") == [1000.0, 3000.0]
    assert bench.pmc_values(str(tmp_path / "FETCH_SIZE"), "FETCH_SIZE", "k_fold_samples") == [7.0]
    assert bench.pmc_values(str(tmp_path / "FETCH_SIZE"), "WRITE_SIZE", "k_wf_shade") == []


@pytest.mark.parametrize("scene", ["gto_sixteen", "cornell_dragon", "bunny_low"])
def test_insertion_optimised_tree_keeps_every_hit_and_costs_less(scene):
    """pth_build_params.optimize_passes (extension): every node re-inserted where the tree's area cost grows least.  The
    flattened tree holds the same triangle references, stays within the Compact layout's depth, costs less, and the oracle's
    walk over it reports the closest hits of the plain tree and of brute force bit for bit (gto_sixteen: spatial-split
    references with clipped boxes move around too)."""
    import orc
    mesh = g.scene_mesh(scene)
    plain, opt = g.Bvh(mesh), g.Bvh(mesh, optimize_passes=2)
    assert opt.stats["n_tri_refs"] == plain.stats["n_tri_refs"] and opt.stats["n_leaves"] == plain.stats["n_leaves"]
    assert opt.stats["max_depth"] <= 64
    assert 0 < opt.stats["opt_cost_after"] < 0.99 * opt.stats["opt_cost_before"]
    assert opt.stats["sah_cost"] < plain.stats["sah_cost"]
    assert sorted(opt.index[::1][opt.tris.view(np.uint32)[:, 0] != 0x80000000][::3].tolist()) == \
        sorted(plain.index[::1][plain.tris.view(np.uint32)[:, 0] != 0x80000000][::3].tolist())
    lo, hi = mesh.bounds()
    rays = orc.random_rays(20000, lo, hi, seed=11)
    for cull in (True, False):
        t0, i0, n0, _ = orc.trace_bvh(plain, rays, cull)
        t1, i1, n1, c1 = orc.trace_bvh(opt, rays, cull)
        tb, ib, nb = orc.trace_brute(mesh, rays, cull)
        assert np.array_equal(t1, tb) and np.array_equal(i1, ib)
        assert np.array_equal(t0, t1) and np.array_equal(i0, i1)
        assert (ib >= 0).mean() > 0.05

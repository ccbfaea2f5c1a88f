#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ (run in the dev container).

  ref_primary_hits.npz  rays + closest-hit distances produced by the REFERENCE's own CPU
                        intersector (oracle/_ref/cpuraytracer_core hits, i.e.
                        CpuRayTracer/src/triangle.hpp:49-72 + kdtree.cpp:59-90 compiled from
                        /root/reference) on Assets/dragon.obj and Assets/gto_sixteen.obj.
                        Needs /root/reference; this is the pin of the oracle's geometric core.
  oracle_images.npz     64x64 float accum images of the oracle (1/4/16 spp x 4 materials,
                        cornell + sphere room): regression vectors for oracle AND HIP.
  oracle_kat.npz        RNG / sincos / pow / camera-ray known answers of the oracle.
  cornell_compact.npz   our flatten of cornell.obj in the reference's Compact layout.
The oracle has no reference-provided vectors to pin its radiance against (the reference has
no tests and its RNG is cuRAND): see the PARITY STATUS note in oracle/pt_oracle.c.
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import gpu_pathtracer_amd as g  # noqa: E402
import orc  # noqa: E402

REF_ASSETS = "/root/reference/Assets"


def golden_camera(W, H):
    cam = g.default_camera(W, H)
    cam.dist = 18.0 * H / 1080.0  # the 1080p field of view at any test resolution
    return cam


def make_ref_hits():
    if not os.path.isdir(REF_ASSETS) or not os.path.exists(orc.REF_BIN):
        print("skip ref_primary_hits (needs /root/reference and oracle/_ref)")
        return
    out = {}
    for name in ("dragon", "gto_sixteen"):
        mesh = g.scene_mesh(name)
        lo, hi = mesh.bounds()
        c = 0.5 * (lo + hi)
        # 48x48 pinhole rays from in front of the mesh + 2048 incoherent rays
        ext = float(np.max(hi - lo))
        eye = c + np.array([0.3 * ext, 0.2 * ext, 1.6 * ext], np.float32)
        ys, xs = np.mgrid[0:48, 0:48]
        tgt = c[None, :] + np.stack([(xs.ravel() / 47.0 - 0.5) * ext, (ys.ravel() / 47.0 - 0.5) * ext,
                                     np.zeros(48 * 48)], -1)
        d = tgt - eye[None, :]
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        prim = np.zeros((48 * 48, 8), np.float32)
        prim[:, 0:3] = eye
        prim[:, 4:7] = d
        rays = np.concatenate([prim, orc.random_rays(2048, lo, hi, seed=7)]).astype(np.float32)
        with tempfile.TemporaryDirectory() as td:
            rf, of = os.path.join(td, "rays.f32"), os.path.join(td, "t.f32")
            rays.tofile(rf)
            subprocess.check_call([orc.REF_BIN, "hits", os.path.join(REF_ASSETS, name + ".obj"), rf, str(len(rays)), of],
                                  stdout=subprocess.DEVNULL)
            t = np.fromfile(of, np.float32)
        out[name + "_rays"] = rays
        out[name + "_t_ref"] = t
        print(name, "reference hits:", int((t < 1e30).sum()), "of", len(t))
    np.savez_compressed(os.path.join(HERE, "ref_primary_hits.npz"), **out)


def make_oracle_images():
    W = H = 64
    bvh = g.Bvh(g.scene_mesh("cornell"))
    sph = g.reference_spheres()
    cam = golden_camera(W, H)
    out = {}
    for mat, mname in ((g.MAT_DIFF, "diff"), (g.MAT_METAL, "metal"), (g.MAT_SPEC, "spec"), (g.MAT_REFR, "refr")):
        for spp in (1, 4, 16):
            p = g.default_params(W, H, tri_mat=mat)
            p.flags = g.FLAG_WRITE_RGBA
            acc, rgba, cnt = orc.render(bvh, sph, cam, p, spp=spp)
            out[f"{mname}_{spp}"] = acc
            out[f"{mname}_{spp}_rgba"] = rgba
    p = g.default_params(W, H, tri_mat=g.MAT_METAL)
    p.flags = g.FLAG_METAL_LITERAL_W
    out["metal_literal_4"] = orc.render(bvh, sph, cam, p, spp=4)[0]
    p = g.default_params(W, H)
    out["nospheres_4"] = orc.render(bvh, None, cam, p, spp=4)[0]   # open box: misses return bk
    np.savez_compressed(os.path.join(HERE, "oracle_images.npz"), **out)
    print("oracle images:", len(out))


def make_kat():
    L = orc.lib()
    frames = np.array([0, 1, 2, 12345, 2 ** 40 + 17], np.uint64)
    out = {"wang64_in": frames, "wang64_out": np.array([L.orc_wang64(int(f)) for f in frames], np.uint64)}
    draws = [(int(f), int(p), d) for f in (0, 1, 999) for p in (0, 1, 1920 * 1080 - 1, 2 ** 33) for d in (0, 1, 5, 17)]
    out["rng_key"] = np.array(draws, np.uint64)
    out["rng_val"] = np.array([L.orc_rng_draw(f, p, d) for f, p, d in draws], np.float32)
    u = np.concatenate([np.linspace(2.0 ** -24, 1.0, 257, dtype=np.float32),
                        np.array([0.125, 0.25, 0.375, 0.5, 0.625, 0.75, 0.875], np.float32)])
    cs = np.zeros((len(u), 2), np.float32)
    for i, x in enumerate(u):
        c_, s_ = C.c_float(), C.c_float()
        L.orc_sincos2pi(float(x), C.byref(c_), C.byref(s_))
        cs[i] = (c_.value, s_.value)
    out["sincos_u"], out["sincos_cs"] = u, cs
    x = np.concatenate([np.array([0.0, 2.0 ** -24, 1.0], np.float32), np.linspace(0.001, 0.999, 64, dtype=np.float32)])
    y = np.array([1.0 / 31.0, 0.5, 1.0 / 3.0, 1.0], np.float32)
    out["pow_x"], out["pow_y"] = x, y
    out["pow_v"] = np.array([[L.orc_pow01(float(a), float(b)) for b in y] for a in x], np.float32)
    cam = golden_camera(64, 64)
    rays = orc.primary_rays(cam, 64, 64, frame=3, jitter=True)
    out["cam_rays_64_f3"] = rays
    np.savez_compressed(os.path.join(HERE, "oracle_kat.npz"), **out)
    print("KAT written")


def make_cornell_compact():
    b = g.Bvh(g.scene_mesh("cornell"))
    np.savez_compressed(os.path.join(HERE, "cornell_compact.npz"), nodes=b.nodes, tris=b.tris, index=b.index)
    print("cornell compact:", b.nodes.shape, b.tris.shape, b.index.shape)


# ---- radiance of the REFERENCE's CPU tracer on its own room (CpuRayTracer/src/main.cpp:26-35) ------------------
REF_W, REF_H, REF_SPP, REF_BLOCK = 160, 120, 400, 20
REF_MESH_POS = (0.3, 0.2, 0.0)


def ref_room_mesh():
    """bunny_low turned z-up, 2.4 units tall, standing on the z = 0 floor of main.cpp's room."""
    mesh = g.scene_mesh("bunny_low")
    v, f = mesh.verts.astype(np.float64), mesh.tris
    lo, hi = v.min(0), v.max(0)
    c, ext = 0.5 * (lo + hi), float(np.max(hi - lo))
    v = (v - c) * (2.4 / ext)
    v = np.stack([v[:, 0], -v[:, 2], v[:, 1]], -1)
    v[:, 2] -= v[:, 2].min()
    return v.astype(np.float32), f


def make_ref_render():
    """oracle/_ref render = Scene::trace_ray / Material::get_reflected_ray / Sphere / Mesh / KDNode of the reference,
    compiled from /root/reference, on main.cpp's room + a mesh.  Kept as block means (a 160x120 image at 400 spp is
    still noisy per pixel; 20x20 blocks average 160 000 paths each)."""
    if not os.path.exists(orc.REF_BIN):
        print("skip ref_room_radiance (needs oracle/_ref)")
        return
    v, f = ref_room_mesh()
    with tempfile.TemporaryDirectory() as td:
        obj, out = os.path.join(td, "m.obj"), os.path.join(td, "img.f32")
        with open(obj, "w") as fh:
            fh.write("o mesh\n")
            np.savetxt(fh, v, fmt="v %.7f %.7f %.7f")
            np.savetxt(fh, f + 1, fmt="f %d %d %d")
        r = subprocess.run([orc.REF_BIN, "render", obj, str(REF_W), str(REF_H), str(REF_SPP)] + [str(x) for x in REF_MESH_POS] + [out],
                           capture_output=True, text=True, check=True)
        img = np.fromfile(out, np.float32).reshape(REF_H, REF_W, 3)
    blocks = img.reshape(REF_H // REF_BLOCK, REF_BLOCK, REF_W // REF_BLOCK, REF_BLOCK, 3).mean(axis=(1, 3))
    np.savez_compressed(os.path.join(HERE, "ref_room_radiance.npz"), blocks=blocks.astype(np.float32), mean=np.float32(img.mean()),
                        verts=v, tris=f.astype(np.int32), width=REF_W, height=REF_H, spp=REF_SPP, block=REF_BLOCK,
                        mesh_pos=np.array(REF_MESH_POS, np.float32))
    print("reference room radiance:", r.stdout.strip().splitlines()[-1], "block means", blocks.mean(axis=-1).round(3).tolist())


if __name__ == "__main__":
    make_ref_render()
    make_ref_hits()
    make_oracle_images()
    make_kat()
    make_cornell_compact()

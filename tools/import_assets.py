#!/usr/bin/env python3
"""Convert the reference's mesh DATA files (Wavefront OBJ) into compact binary
`.ptmesh` fixtures under assets/.

The GPU box never sees /root/reference, so the three meshes that BASELINE.json's
configs name (cornell.obj 32 tris, dragon.obj 100 000 tris, gto_sixteen.obj 9 791
tris) are carried as data: float32 vertices + int32 triangle indices.
Nothing but `v` / `f` records is kept (the reference ignores everything else too:
GpuPathTracer/utilfun.cpp:393-530).  This script is only runnable where
/root/reference exists (the dev container); its outputs are committed.

.ptmesh layout (little endian):
    char  magic[8]  = "PTMESH1\\0"
    u32   n_verts, n_tris
    f32   verts[n_verts][3]
    i32   tris[n_tris][3]          (0-based vertex indices)
"PTMESH2\0" adds u32 n_materials after n_tris and, after tris, the material rows
(32 bytes each: col[3], emi[3], i32 mat, f32 phong) and i32 tri_material[n_tris].
"""
import os
import struct
import sys

import numpy as np

REF = "/root/reference/Assets"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets")
MESHES = ["cornell", "dragon", "gto_sixteen", "bunny_low", "cube", "sphere"]


def read_obj(path):
    verts, tris = [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                verts.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):  # fan-triangulate polygons
                    tris.append((idx[0], idx[k], idx[k + 1]))
    return np.asarray(verts, np.float32), np.asarray(tris, np.int32)


def write_ptmesh(path, verts, tris):
    with open(path, "wb") as f:
        f.write(b"PTMESH1\0")
        f.write(struct.pack("<II", len(verts), len(tris)))
        f.write(verts.astype("<f4").tobytes())
        f.write(tris.astype("<i4").tobytes())


def import_cornell_box():
    """cornellBox/CornellBox/CornellBox-Original.obj + .mtl (McGuire's public-domain data set, 36
    triangles, 8 materials, one emissive quad) -> assets/cornell_box.ptmesh (PTMESH2: + material
    table + one row per triangle), moved into the frame cornell.obj lives in so that the
    reference's default camera sees it: p' = 15.5 * (p - (0, 1, 1)) + (0, 0, -28)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import gpu_pathtracer_amd as g
    src = os.path.join(REF, "cornellBox", "CornellBox", "CornellBox-Original.obj")
    if not os.path.exists(src):
        print("missing", src, file=sys.stderr)
        return
    m = g.Mesh.load(src)
    v = (m.verts.astype(np.float32) - np.array([0, 1, 1], np.float32)) * np.float32(15.5) + np.array([0, 0, -28], np.float32)
    out = g.Mesh.from_arrays(v, m.tris).set_materials(m.materials, m.tri_material)
    dst = os.path.join(OUT, "cornell_box.ptmesh")
    out.save(dst)
    print(f"cornell_box: {out.n_verts} verts {out.n_tris} tris {len(out.materials)} materials -> {os.path.getsize(dst)} B")


def main():
    os.makedirs(OUT, exist_ok=True)
    import_cornell_box()
    for name in MESHES:
        src = os.path.join(REF, name + ".obj")
        if not os.path.exists(src):
            print("missing", src, file=sys.stderr)
            continue
        v, t = read_obj(src)
        dst = os.path.join(OUT, name + ".ptmesh")
        write_ptmesh(dst, v, t)
        print(f"{name}: {len(v)} verts {len(t)} tris -> {os.path.getsize(dst)} B")


if __name__ == "__main__":
    main()

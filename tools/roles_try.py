import sys, time
sys.path[:0] = ['.', 'tests']
import numpy as np, gpu_pathtracer_amd as g, orc
from test_gpu_parity import gpu_render, golden_camera, bvh_of
pt = g.PathTracer(0)
pt.set_option(g.OPT_KERNEL, g.KERNEL_WAVEFRONT)
pt.set_option(g.OPT_WALK, 2)
for scene, W, H, spp in (("cornell", 64, 64, 1), ("cornell", 64, 64, 4), ("cornell_dragon", 640, 360, 2), ("dragon", 129, 71, 3)):
    _, bvh = bvh_of(scene)
    sph = g.reference_spheres() if scene != "dragon" else None
    cam = golden_camera(W, H)
    p = g.default_params(W, H)
    p.flags = g.FLAG_WRITE_RGBA
    t0 = time.time()
    acc, rgba = gpu_render(pt, bvh, sph, cam, p, spp)
    dt = time.time() - t0
    ref, rref, _ = orc.render(bvh, sph, cam, p, spp)
    nd = int(np.any(acc != ref, axis=-1).sum())
    print(scene, W, H, spp, "differing pixels", nd, "of", W * H, "rgba equal", bool(np.array_equal(rgba, rref)), f"{dt*1e3:.1f} ms", flush=True)
print("ok")

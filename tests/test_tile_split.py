"""Row (e) on CPU: world_size-2 and -3 `gloo` runs of the tile-split host logic.  Each rank
'renders' its stripes with the CPU oracle (test stand-in for the GPU kernel, same
pt_params.part_* contract), the stripes are gathered with gpu_pathtracer_amd.tile_split —
the code bench.py uses with RCCL — and rank 0 must hold the single-process image bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, rows, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["OMP_NUM_THREADS"] = "2"
    import gpu_pathtracer_amd as g
    from gpu_pathtracer_amd import tile_split
    import orc

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    layout = tile_split.StripeLayout(W, H, world, rank, rows)
    bvh = g.Bvh(g.scene_mesh("cornell"))
    sph = g.reference_spheres()
    cam = g.default_camera(W, H)
    cam.dist = 18.0 * H / 1080.0
    p = layout.apply(g.default_params(W, H))
    p.flags = g.FLAG_WRITE_RGBA
    acc = np.zeros((layout.padded_height, W, 3), np.float32)
    rgba = np.zeros((layout.padded_height, W), np.uint32)
    import ctypes as C
    cnt = g.Counters()
    rc = orc.lib().orc_render(acc.ctypes.data, rgba.ctypes.data, bvh.nodes.ctypes.data, bvh.tris.ctypes.data,
                              bvh.index.ctypes.data, sph, len(sph), C.byref(cam), C.byref(p), 2, C.byref(cnt))
    assert rc == 0
    # a rank must have touched exactly its own rows
    mine = set(layout.owned_rows())
    touched = set(np.nonzero(acc.reshape(layout.padded_height, -1).any(axis=1))[0].tolist())
    assert touched <= mine
    t_acc, t_rgba = torch.from_numpy(acc), torch.from_numpy(rgba.view(np.int32))
    tile_split.gather_stripes(t_acc, layout)
    st = tile_split.gather_stripes(t_rgba, layout)
    tile_split.gather_stripes(t_rgba, layout, staging=st)  # staging reuse, as in the timed loop
    if rank == 0:
        np.savez(out_path, acc=t_acc.numpy()[:H], rgba=t_rgba.numpy()[:H].view(np.uint32))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,rows,W,H", [(2, 8, 64, 40), (3, 16, 48, 72)])
def test_gloo_tile_split_matches_single_process(tmp_path, world, rows, W, H):
    out = str(tmp_path / "merged.npz")
    mp.spawn(_worker, args=(world, _free_port(), W, H, rows, out), nprocs=world, join=True)
    import gpu_pathtracer_amd as g
    import orc
    bvh = g.Bvh(g.scene_mesh("cornell"))
    cam = g.default_camera(W, H)
    cam.dist = 18.0 * H / 1080.0
    p = g.default_params(W, H)
    p.flags = g.FLAG_WRITE_RGBA
    ref_acc, ref_rgba, _ = orc.render(bvh, g.reference_spheres(), cam, p, spp=2)
    z = np.load(out)
    assert np.array_equal(z["acc"], ref_acc)
    assert np.array_equal(z["rgba"], ref_rgba)


def test_stripe_layout_partition_is_exact():
    from gpu_pathtracer_amd import tile_split
    for W, H, world, rows in [(1920, 1080, 8, 8), (1920, 1080, 4, 16), (4096, 4096, 8, 8), (7, 5, 3, 8)]:
        seen = []
        for r in range(world):
            L = tile_split.StripeLayout(W, H, world, r, rows)
            assert L.padded_height % (rows * world) == 0 and L.padded_height >= H
            seen += L.owned_rows()
        assert sorted(seen) == list(range(H))          # every row exactly once
    with pytest.raises(ValueError):
        tile_split.StripeLayout(64, 64, 2, 0, rows=12)

#!/usr/bin/env python3
"""bench.py — headline benchmark of the path-tracing hot path on MI355X.

Metric (BASELINE.json): Mrays/s (+ achieved algorithmic GB/s vs the HBM roofline),
cornell_dragon 1920x1080, depth 4, reference sphere room, 1/2/4/8 GPUs.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one progressive frame: one pass of the hot path — ONE pt_render call through the C
ABI, `render(accum, bvh, camera, spp)` of BASELINE.json — folding `--spp` samples per pixel
(default 16; BASELINE's config 5 uses 8) into the whole 1920x1080 framebuffer.  The reference
launches one sample per displayed frame (BasicScene.cpp:404); a call with spp = S equals S such
launches bit for bit, but traces the S samples as independent work items, which is what gives
eight GPUs enough parallel work on an eighth of the frame each (1 spp: 0.47 ms for an eighth of
a 1.13 ms frame; 8 spp: 1.29 of 7.10 ms).  With N > 1 the framebuffer is
tile-split into interleaved 8-row stripes (stripe s belongs to rank s % N), every rank
renders its stripes of the SAME frame with the scene replicated, and the display words are
gathered on rank 0 with RCCL each step (the reference copies the frame to the display
every frame too, BasicScene.cpp:424-432).  Total work is fixed => "scaling": "strong".

Rank 0 prints ONE JSON line.  Everything under oracle/ is used here only for the
`cpu_baseline` leg and for the algorithmic-byte counters of the roofline (checker, never
the thing measured).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scene", default="cornell_dragon_800k")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--spp", type=int, default=16,
                    help="samples per pixel folded by ONE pt_render call = one step (render(accum, bvh, camera, spp))")
    ap.add_argument("--mat", default="diff", choices=["diff", "metal", "spec", "refr"])
    ap.add_argument("--no-spheres", action="store_true")
    ap.add_argument("--kernel", type=int, default=0, help="PT_KERNEL_* (0 = auto)")
    ap.add_argument("--occ", type=int, default=0, help="PT_OPT_OCCUPANCY (0 = library default)")
    ap.add_argument("--lds-stack", type=int, default=-1, help="PT_OPT_LDS_STACK (-1 = library default)")
    ap.add_argument("--top", type=int, default=-1, help="PT_OPT_TOP_NODES (-1 = library default)")
    ap.add_argument("--stripe-rows", type=int, default=8)
    ap.add_argument("--cpu-frames", type=int, default=2, help="frames of the workload timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-cpu-reference", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the metal/spec side measurements")
    ap.add_argument("--rehearse", action="store_true",
                    help="N>1 dry run on a ONE-GPU box: every rank uses cuda:0, gloo backend, stripes gathered "
                         "through host memory (validates the multi-rank code path, not its speed)")
    return ap.parse_args()


def cpu_baseline(g, bvh, sph, cam, params, first_frame, n_frames, spp):
    """The oracle (kind 'port') timed on this host's cores over `n_frames` frames of the SAME
    workload; also yields the N_* counters that define the algorithmic bytes per frame."""
    import orc
    p = g.Params.from_buffer_copy(params)
    p.part_count, p.part_index = 1, 0
    p.frame, p.sample_index = first_frame, 1
    acc = np.zeros((p.height, p.width, 3), np.float32)
    t0 = time.perf_counter()
    _, _, cnt = orc.render(bvh, sph, cam, p, spp=n_frames * spp, accum=acc, want_rgba=False)
    dt = time.perf_counter() - t0
    cores = int(os.environ.get("OMP_NUM_THREADS", 0)) or len(os.sched_getaffinity(0))
    return {"value": cnt["rays"] / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{n_frames} step(s) x {spp} spp of the bench workload ({p.width}x{p.height}, depth {p.depth}), "
                      f"{cnt['rays']} ray segments in {dt:.2f} s, oracle/pt_oracle.c with OpenMP"}, cnt, acc


def cpu_reference_tracer(g):
    """oracle/_ref/cpuraytracer_core: the reference's own CpuRayTracer classes (compiled from
    /root/reference in the dev container) on ITS scene (main.cpp:26-35 sphere room) with the
    dragon mesh scaled into that room.  A different renderer: reported beside
    the port, never compared pixel-wise."""
    import orc
    if not os.path.exists(orc.REF_BIN):
        return None
    mesh = g.scene_mesh("dragon")
    v, f = mesh.verts, mesh.tris
    lo, hi = mesh.bounds()
    c, ext = 0.5 * (lo + hi), float(np.max(hi - lo))
    # y-up -> z-up, ~3 units tall, standing on the z=0 floor sphere of main.cpp:30
    v = (v - c) * (3.0 / ext)
    v = np.stack([v[:, 0], -v[:, 2], v[:, 1]], -1)
    v[:, 2] -= v[:, 2].min()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "dragon_room.obj")
        with open(path, "w") as fh:
            fh.write("o dragon\n")
            np.savetxt(fh, v, fmt="v %.6f %.6f %.6f")
            np.savetxt(fh, f + 1, fmt="f %d %d %d")
        W, H, spp = 480, 270, 4
        out = subprocess.run([orc.REF_BIN, "render", path, str(W), str(H), str(spp), "0", "0", "0"],
                             capture_output=True, text=True, timeout=600)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode != 0 or not line:
        return None
    r = json.loads(line[-1])
    return {"value": r["segments"] / r["seconds"] / 1e6, "unit": "Mrays/s", "cores": r["threads"], "kind": "reference",
            "mpaths_per_s": r["paths"] / r["seconds"] / 1e6,
            "sample": f"CpuRayTracer classes (Scene/Mesh/KDNode/Material, double precision) on main.cpp's 5-sphere "
                      f"room + dragon (100k tris), {W}x{H}, {spp} spp, {r['segments']} segments in {r['seconds']:.2f} s"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    import gpu_pathtracer_amd as g

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the path tracer has no CPU fallback)")
    if a.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    W, H = a.width, a.height
    mat = {"diff": g.MAT_DIFF, "metal": g.MAT_METAL, "spec": g.MAT_SPEC, "refr": g.MAT_REFR}[a.mat]
    mesh = g.scene_mesh(a.scene)
    bvh = g.Bvh(mesh)
    sph = None if a.no_spheres else g.reference_spheres()
    n_sph = 0 if sph is None else len(sph)
    cam = g.default_camera(W, H)
    base = g.default_params(W, H, depth=a.depth, tri_mat=mat)
    base.flags = g.FLAG_WRITE_RGBA

    pt = g.PathTracer(local_rank)
    # one explicit HIP stream shared by torch (events, copies, RCCL ordering) and the kernels:
    # torch.cuda.Event only sees torch's current stream
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    pt.set_stream(stream.cuda_stream)
    if a.kernel:
        pt.set_option(g.OPT_KERNEL, a.kernel)
    if a.occ:
        pt.set_option(g.OPT_OCCUPANCY, a.occ)
    if a.lds_stack >= 0:
        pt.set_option(g.OPT_LDS_STACK, a.lds_stack)
    if a.top >= 0:
        pt.set_option(g.OPT_TOP_NODES, a.top)
    pt.upload_bvh(bvh)
    pt.upload_spheres(sph)
    info = pt.scene_info()

    # full-frame buffers, height padded so that the stripes split evenly over the ranks
    from gpu_pathtracer_amd import tile_split
    layout = tile_split.StripeLayout(W, H, world, rank, a.stripe_rows)
    rows = a.stripe_rows
    layout.apply(base)
    accum = torch.zeros((layout.padded_height, W, 3), dtype=torch.float32, device=dev)
    # display words are double-buffered so that the gather of frame i (side stream, RCCL)
    # overlaps the render of frame i+1 (main stream); PT_BENCH_NO_OVERLAP=1 serialises them
    overlap = world > 1 and not a.rehearse and os.environ.get("PT_BENCH_NO_OVERLAP", "0") != "1"
    n_buf = 2 if overlap else 1
    rgbas = [torch.zeros((layout.padded_height, W), dtype=torch.int32, device=dev) for _ in range(n_buf)]
    rgba = rgbas[0]
    staging = [None] * n_buf
    side = torch.cuda.Stream(device=dev) if overlap else None
    ev_render = [torch.cuda.Event() for _ in range(n_buf)]
    ev_gather = [None] * n_buf
    step_no = [0]

    def step(i, params=base, spp=a.spp, fresh=False):
        p = g.Params.from_buffer_copy(params)
        p.frame, p.sample_index = i * spp, 1 if fresh else 1 + i * spp
        k = step_no[0] % n_buf
        step_no[0] += 1
        buf = rgbas[k]
        if overlap and ev_gather[k] is not None:
            stream.wait_event(ev_gather[k])          # the gather that last read this buffer is done
        pt.launch_kernel(accum.data_ptr(), buf.data_ptr(), cam, p, spp)
        if world == 1:
            return
        if overlap:
            ev_render[k].record(stream)
            with torch.cuda.stream(side):
                side.wait_event(ev_render[k])
                staging[k] = tile_split.gather_stripes(buf, layout, dst=0, staging=staging[k])
                ev_gather[k] = torch.cuda.Event()
                ev_gather[k].record(side)
                buf.record_stream(side)
        elif not a.rehearse:   # display words of the finished stripes -> rank 0 (RCCL over xGMI)
            staging[k] = tile_split.gather_stripes(buf, layout, dst=0, staging=staging[k])
        else:
            host = buf.cpu()
            staging[k] = tile_split.gather_stripes(host, layout, dst=0, staging=staging[k])
            if rank == 0:
                buf.copy_(host)

    def last_frame():
        return rgbas[(step_no[0] - 1) % n_buf]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_steps, first, params=base, events=False):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_steps)] if events else None
        barrier()
        t0 = time.perf_counter()
        for k in range(n_steps):
            if ev:
                ev[k][0].record()
            step(first + k, params)
            if ev:
                ev[k][1].record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if a.rehearse else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        kms = [e0.elapsed_time(e1) for e0, e1 in ev] if ev else None
        return dt, kms

    for k in range(a.warmup):
        step(k)
    dt, kernel_ms = timed(a.steps, a.warmup, events=True)

    # exact segment count of the timed frames: replay them instrumented (untimed)
    pt.set_option(g.OPT_COUNTERS, 1)
    seg = torch.zeros(2, dtype=torch.float64, device="cpu" if a.rehearse else dev)
    n_count = min(a.steps, 4)
    for k in range(n_count):
        step(a.warmup + k)
        torch.cuda.synchronize()
        c = pt.counters()
        seg[0] += c["rays"]
        seg[1] += c["paths"]
    pt.set_option(g.OPT_COUNTERS, 0)
    if world > 1:
        dist.all_reduce(seg)
    rays_per_step = float(seg[0].item()) / n_count
    closed = abs(rays_per_step - W * H * a.depth * a.spp) < 0.5
    total_rays = rays_per_step * a.steps

    extra = {}
    if not a.no_extra:
        for name, m in (("metal", g.MAT_METAL), ("spec", g.MAT_SPEC)):
            if m == mat:
                continue
            pm = g.Params.from_buffer_copy(base)
            pm.tri_mat = m
            n_x = max(5, a.steps // 5)
            step(0, pm)
            dtx, _ = timed(n_x, 1, pm)
            extra[f"mrays_per_s_{name}"] = round(W * H * a.depth * a.spp * n_x / dtx / 1e6, 1)

    if not a.no_extra and world == 1:
        # side measurement: the same workload over a tree built ON the device
        # (pt_build_bvh, LBVH); done last, it replaces the scene of this context
        info_host = pt.scene_info()
        build_ms = min(pt.build_bvh(mesh) for _ in range(3))
        n_x = max(5, a.steps // 5)
        step(0)
        dtx, _ = timed(n_x, 1)
        extra["device_bvh_build_ms"] = round(build_ms, 2)          # PLOC (the default PT_OPT_BUILD_ALGO)
        extra["mrays_per_s_device_built_tree"] = round(W * H * a.depth * a.spp * n_x / dtx / 1e6, 1)
        pt.set_option(g.OPT_BUILD_ALGO, 0)                          # Karras LBVH: the fastest build
        extra["device_bvh_build_ms_lbvh"] = round(min(pt.build_bvh(mesh) for _ in range(3)), 2)
        pt.set_option(g.OPT_BUILD_ALGO, 1)
        info = info_host

    merged_ok = None
    if world > 1:
        # rank 0 re-renders the last gathered frame alone and compares the display words
        last = a.warmup + a.steps
        step(last, fresh=True)   # sample_index 1: the frame does not depend on accumulated history
        torch.cuda.synchronize()
        if rank == 0:
            solo = g.Params.from_buffer_copy(base)
            solo.part_index, solo.part_count = 0, 1
            solo.frame, solo.sample_index = last * a.spp, 1
            acc2 = torch.zeros_like(accum)
            rgba2 = torch.zeros_like(rgba)
            pt.launch_kernel(acc2.data_ptr(), rgba2.data_ptr(), cam, solo, a.spp)
            torch.cuda.synchronize()
            merged_ok = bool(torch.equal(last_frame()[:H], rgba2[:H]))
    if rank == 0:
        value = total_rays / dt / 1e6
        out = {
            "metric": "Mrays/sec, cornell_dragon 1920x1080 (+ achieved algorithmic GB/s vs HBM roofline)",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, gloo)" if a.rehearse else ""),
            "config": {"workload": f"{a.scene} ({mesh.n_tris} tris) {W}x{H} depth {a.depth} {a.mat} + "
                                   f"{'reference 8-sphere room' if n_sph else 'no spheres'}, {a.spp} spp per step",
                       "bvh": {"inner": info["n_inner"], "tri_refs": info["n_tri_refs"], "max_depth": info["max_depth"],
                               "device_mb": round(info["device_bytes"] / 2 ** 20, 1)},
                       "parallelism": (f"tile-split x{world} ({rows}-row stripes, RCCL all-gather of RGBA8 every step"
                                       f"{', overlapped with the next render' if overlap else ''})") if world > 1 else "1 GPU",
                       "closed_scene": bool(closed), "rays_per_step": rays_per_step,
                       "tile_split_equals_single_gpu": merged_ok},
        }
        out.update(extra)
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None}
        if world == 1:
            kavg = float(np.mean(kernel_ms))
            roof["kernel"] = "k_trace_persist_bvh2" if a.kernel in (0, 3) else "k_trace_mega_bvh2"
            roof["kernel_ms_avg"] = round(kavg, 4)
            if a.cpu_frames > 0:
                cb, cnt, _ = cpu_baseline(g, bvh, sph, cam, base, a.warmup * a.spp, a.cpu_frames, a.spp)
                out["cpu_baseline"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in cb.items()}
                alg = g.algorithmic_bytes(cnt, n_sph) / a.cpu_frames      # bytes per step
                roof["achieved"] = round(alg / (kavg * 1e-3) / 1e9, 1)
                roof["frac"] = round(roof["achieved"] / HBM_PEAK_GBS, 4)
                roof["algorithmic_bytes_per_step"] = int(alg)
                roof["bytes_per_ray"] = round(alg / (cnt["rays"] / a.cpu_frames), 1)
                roof["nodes_per_ray"] = round(cnt["inner"] / cnt["rays"], 2)
                roof["tris_per_ray"] = round(cnt["tris"] / cnt["rays"], 2)
            # PMC traffic cannot be collected inside this process: it is the committed figure of the
            # rocprofv3 --pmc passes over THIS command (tools/profile_gpu.sh), valid for the default workload
            tr = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
            default_workload = (a.scene, W, H, a.depth, a.spp, a.mat, a.no_spheres) == ("cornell_dragon_800k", 1920, 1080, 4, 16, "diff", False)
            if os.path.exists(tr) and default_workload:
                try:
                    roof["traffic"] = json.load(open(tr)).get("hbm_bytes_per_launch")
                except Exception:
                    pass
            roof["note"] = ("frac > 1 is not a measurement error: 'achieved' counts ALGORITHMIC bytes (SURVEY 8d: 64 B per node "
                            "visited, 48 B per triangle tested, ...), which the L2 / Infinity Cache serve; 'traffic' is what "
                            "crossed the L2<->fabric boundary per launch. The kernel is bound by the random 64-byte gather rate "
                            "of the cache hierarchy and by divergent VALU issue, not by HBM (DESIGN.md 5, 7).")
            if not a.no_cpu_reference:
                try:
                    ref = cpu_reference_tracer(g)
                    if ref:
                        out["cpu_reference_tracer"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in ref.items()}
                except Exception as e:  # a baseline, never a reason to lose the GPU number
                    out["cpu_reference_tracer"] = {"error": str(e)[:200]}
        out["roofline"] = roof
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    pt.close()


if __name__ == "__main__":
    main()

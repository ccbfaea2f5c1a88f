#!/usr/bin/env python3
"""bench.py — headline benchmark of the path-tracing hot path on MI355X.

Metric (BASELINE.json): Mrays/s (+ achieved GB/s against the memory roofline),
cornell_dragon 1920x1080, depth 4, reference sphere room, 1/2/4/8 GPUs.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one progressive frame: one pass of the hot path — ONE pt_render call through the C
ABI, `render(accum, bvh, camera, spp)` of BASELINE.json — folding `--spp` samples per pixel
(default 16; BASELINE's config 5 uses 8) into the whole 1920x1080 framebuffer.  The reference
launches one sample per displayed frame (BasicScene.cpp:404); a call with spp = S equals S such
launches bit for bit, but traces the S samples as independent work items (the 1-spp-per-call rate
is reported beside the headline as `mrays_per_s_1spp`).  With N > 1 the framebuffer is
tile-split into interleaved 8-row stripes (stripe s belongs to rank s % N), every rank
renders its stripes of the SAME frame with the scene replicated, the display words are
gathered on rank 0 with RCCL each step (the reference copies the frame to the display
every frame too, BasicScene.cpp:424-432) and the float accumulator stripes once at the end.  Total work is fixed => "scaling": "strong".
`--gpus N` without a torchrun environment starts the N ranks itself, as child processes.

Rank 0 prints ONE JSON line.  Everything under oracle/ is used here only AFTER the timed region:
for the `cpu_baseline` leg, for the parity check of the timed configuration against it, and for
the algorithmic-byte counters of the roofline (checker, never the thing measured).
"""
import argparse
import glob
import hashlib
import json
import os
import socket
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
OPT_PASSES = 4         # PT_OPT_OPTIMIZE passes at upload (csrc/pt_tree_opt.h; the batched passes converge by the fourth)
KERNEL_NAMES = {0: "auto", 1: "mega", 3: "persistent", 5: "wavefront"}


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
    ap.add_argument("--kernel", type=int, default=0, help="PT_KERNEL_* (0 = auto, 1 mega, 3 persistent, 5 wavefront)")
    ap.add_argument("--occ", type=int, default=0, help="PT_OPT_OCCUPANCY (0 = library default)")
    ap.add_argument("--lds-stack", type=int, default=-1, help="PT_OPT_LDS_STACK (-1 = library default)")
    ap.add_argument("--top", type=int, default=-1, help="PT_OPT_TOP_NODES (-1 = library default)")
    ap.add_argument("--batch", type=int, default=0, help="PT_OPT_BATCH / PT_OPT_WAVE_BATCH (0 = library default)")
    ap.add_argument("--no-overlap", action="store_true", help="PT_OPT_OVERLAP 0: every launch on the caller's stream")
    ap.add_argument("--stripe-rows", type=int, default=8)
    ap.add_argument("--cpu-frames", type=int, default=1, help="steps of the workload run on the CPU oracle: baseline + parity (0 = skip)")
    ap.add_argument("--no-cpu-reference", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the side measurements (materials, 1 spp, device tree, big scene, gather bound)")
    ap.add_argument("--device-build", action="store_true", help="build the BVH on the device (pt_build_bvh) instead of the host SBVH builder")
    ap.add_argument("--device-optimize", type=int, default=0, help="with --device-build: PT_OPT_OPTIMIZE passes over the device-built tree")
    ap.add_argument("--host-splits", action="store_true", help="host tree with the builder's spatial splits (the reference's splitAlpha = 1e-5)")
    ap.add_argument("--keep-hierarchy", action="store_true",
                    help="PT_OPT_REBUILD 0: walk the uploaded hierarchy whatever it costs (default: PT_OPT_REBUILD 2, the upload also "
                         "re-clusters the triangles on the device and keeps the tree with the smaller area cost in node visits)")
    ap.add_argument("--force-dist", action="store_true",
                    help="with --gpus 1: still initialise torch.distributed over RCCL (backend nccl, world 1) and send every step's "
                         "display words through the same gather-to-root / side-stream path the N > 1 runs use")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not collect HBM traffic live (child runs of this script under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # the run rocprofv3 wraps: timed steps only
    ap.add_argument("--big-parity-spp", type=int, default=2, help="samples of the big scene's in-run parity frame against the oracle (0 = skip)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N>1 dry run on a ONE-GPU box: every rank uses cuda:0, gloo backend, stripes gathered "
                         "through host memory (validates the multi-rank code path, not its speed)")
    return ap.parse_args()


def source_sha():
    """Identifies the kernel sources a profile was taken with (profiles/*_pmc_traffic.json is only quoted
    when it was collected for THIS code)."""
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "g.p.u-pathtracer_amd", "csrc", "*.h*")) + [os.path.join(ROOT, "include", "ptmi.h")]):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(g, bvh, sph, cam, params, first_frame, n_frames, spp):
    """The oracle (kind 'port') timed on this host's cores over `n_frames` steps of the SAME workload,
    starting from an empty accumulator; also yields the N_* counters that define the algorithmic bytes,
    and the accumulator for the parity check."""
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


def gather_bound(table_bytes):
    """Random 64-byte dependent gathers from a table of the scene's size, 8 waves/SIMD, every lane busy:
    the rate the memory hierarchy serves ITEMS at for this working set (tools/ubench_gather.hip), measured
    by a CHILD process (its own HIP context; nothing is exec'ed from this one)."""
    exe = os.path.join(ROOT, "tools", "ubench_gather")
    if not os.path.exists(exe):
        return None
    n_items = max(1024, int(table_bytes // 64))
    out = subprocess.run([exe, "--json", str(n_items), "256"], capture_output=True, text=True, timeout=120)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    return json.loads(line[-1]) if out.returncode == 0 and line else None


def pmc_values(directory, counter, kernel_family):
    """Per-dispatch values of `counter` for the non-instrumented instantiations of one kernel family (k_wf_extend<false, ...>,
    k_trace_persist_bvh2<false, ...>, plain names) from the counter_collection.csv files rocprofv3 --pmc wrote under `directory`."""
    import csv
    vals = []
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            head = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if row["Counter_Name"] != counter or head.split("<")[0] != kernel_family:
                continue
            if "<" in head and head.split("<")[-1].split(",")[0].strip() == "true":   # instrumented (COUNT) instantiation
                continue
            vals.append(float(row["Counter_Value"]))
    return vals


def pmc_live(kernel_family, extra_args, timeout=240):
    """HBM-side traffic per launch of one kernel family, measured NOW for the running sources: this script is run again as a
    CHILD under `rocprofv3 --pmc <counter>` (one pass per counter: FETCH_SIZE and WRITE_SIZE do not fit one pass,
    MI355X_MICROARCH.md "rocprofv3 PMC slots"), timed steps only.  FETCH_SIZE is raw (TCC_EA0_RDREQ x 64 B: exact for 64-byte
    gathers, HALF the bytes of a wide coalesced stream on gfx950 — the x2 figure is kept beside it); WRITE_SIZE is exact.
    Returns None when rocprofv3 is not there or a pass fails (the committed profiles/ figure is quoted instead)."""
    import shutil
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    out = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
            env.pop(k, None)
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(td, counter)
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--steps", "4", "--warmup", "2", "--cpu-frames", "0", "--no-cpu-reference", "--no-extra", "--no-pmc"] + extra_args
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
            except Exception:
                return None
            vals = pmc_values(d, counter, kernel_family)
            if r.returncode != 0 or not vals:
                return None
            out[counter] = (sum(vals) / len(vals) * 1024.0, len(vals))
    fs, ws = out["FETCH_SIZE"][0], out["WRITE_SIZE"][0]
    return {"fetch_bytes_raw": int(fs), "fetch_bytes_x2": int(2 * fs), "write_bytes": int(ws), "hbm_bytes_per_launch": int(fs + ws),
            "launches_profiled": out["FETCH_SIZE"][1], "source": "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child runs of this bench, same sources"}


def spawn_ranks(a):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks as children (fresh
    processes, before anything here has touched the GPU) and pass rank 0's JSON line through."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(a))
    # stdout carries exactly ONE line, rank 0's JSON: whatever libraries print on the way (RCCL's version banner, gloo's
    # connection notes) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
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
    use_dist = world > 1 or a.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:   # --force-dist: a one-rank RCCL group, no torchrun needed
            s_ = socket.socket()
            s_.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1]))
            s_.close()
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if a.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    W, H = a.width, a.height
    mat = {"diff": g.MAT_DIFF, "metal": g.MAT_METAL, "spec": g.MAT_SPEC, "refr": g.MAT_REFR}[a.mat]
    mesh = g.scene_mesh(a.scene)
    # the host tree: the clean-room SBVH builder (host/pthost.cpp) WITHOUT its spatial splits — once the upload optimises the hierarchy
    # (PT_OPT_OPTIMIZE) they no longer pay on this scene (9.00 vs 8.87 ms per step: profiles/r03_tree_opt.txt); --host-splits keeps them
    bvh = None if a.device_build else (g.Bvh(mesh) if a.host_splits else g.Bvh(mesh, split_alpha=-1.0))
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
    if a.no_overlap:
        pt.set_option(g.OPT_OVERLAP, 0)
    if a.batch:
        pt.set_option(g.OPT_BATCH, a.batch)
        pt.set_option(g.OPT_WAVE_BATCH, a.batch)
    tree_note = "built on the device (pt_build_bvh, PLOC)"
    if a.device_build:
        pt.set_option(g.OPT_OPTIMIZE, a.device_optimize)
        pt.build_bvh(mesh)
        pt.set_option(g.OPT_OPTIMIZE, 0)
        if a.device_optimize:
            tree_note += f", then PT_OPT_OPTIMIZE {a.device_optimize} on the host"
    else:
        # the reference's flow: hierarchy built on the host (SBVH port), flattened, uploaded.  PT_OPT_OPTIMIZE: the upload first
        # re-inserts every node of the caller's hierarchy where the area cost grows least (OPT_PASSES passes on the host's threads,
        # 2-3 s, outside every timed region — the reference's own builder takes longer); PT_OPT_REBUILD 2: it also re-clusters the
        # triangles on the device, optimises that tree too, and keeps whichever hierarchy is cheaper to walk
        pt.set_option(g.OPT_OPTIMIZE, 0 if a.keep_hierarchy else OPT_PASSES)
        pt.set_option(g.OPT_REBUILD, 0 if a.keep_hierarchy else 2)
        t_up = time.perf_counter()
        pt.upload_bvh(bvh)
        t_up = time.perf_counter() - t_up
        pt.set_option(g.OPT_REBUILD, 0)
        pt.set_option(g.OPT_OPTIMIZE, 0)
        host_kind = "host SBVH hierarchy" if a.host_splits else "host SAH hierarchy (no spatial splits)"
        tree_note = (f"{host_kind}, uploaded as built (PT_OPT_OPTIMIZE 0, PT_OPT_REBUILD 0)" if a.keep_hierarchy else
                     f"{host_kind} uploaded with PT_OPT_OPTIMIZE {OPT_PASSES} + PT_OPT_REBUILD 2 ({t_up:.1f} s): kept " +
                     ("the device's re-clustered tree, optimised" if pt.last_build_ms() > 0 else "the uploaded hierarchy, optimised"))
    pt.upload_spheres(sph)
    info = pt.scene_info()
    try:
        tree_cost = pt.tree_cost()
    except Exception:
        tree_cost = None

    # full-frame buffers, height padded so that the stripes split evenly over the ranks
    from gpu_pathtracer_amd import tile_split
    layout = tile_split.StripeLayout(W, H, world, rank, a.stripe_rows)
    rows = a.stripe_rows
    layout.apply(base)
    accum = torch.zeros((layout.padded_height, W, 3), dtype=torch.float32, device=dev)
    # display words are double-buffered so that the gather of frame i (side stream, RCCL)
    # overlaps the render of frame i+1 (main stream); PT_BENCH_NO_OVERLAP=1 serialises them
    overlap = use_dist and not a.rehearse and os.environ.get("PT_BENCH_NO_OVERLAP", "0") != "1"
    n_buf = 2 if overlap else 1
    rgbas = [torch.zeros((layout.padded_height, W), dtype=torch.int32, device=dev) for _ in range(n_buf)]
    rgba = rgbas[0]
    staging = [None] * n_buf
    side = torch.cuda.Stream(device=dev) if overlap else None
    ev_render = [torch.cuda.Event() for _ in range(n_buf)]
    ev_gather = [None] * n_buf
    step_no = [0]

    def step(i, params=base, spp=a.spp, fresh=False, into=None):
        p = g.Params.from_buffer_copy(params)
        p.frame, p.sample_index = i * spp, 1 if fresh else 1 + i * spp
        k = step_no[0] % n_buf
        step_no[0] += 1
        buf = rgbas[k]
        if overlap and ev_gather[k] is not None:
            stream.wait_event(ev_gather[k])          # the gather that last read this buffer is done
        pt.launch_kernel((accum if into is None else into).data_ptr(), buf.data_ptr(), cam, p, spp)
        if not use_dist:
            return
        if overlap:
            ev_render[k].record(stream)
            with torch.cuda.stream(side):
                side.wait_event(ev_render[k])
                staging[k] = tile_split.gather_stripes(buf, layout, dst=0, staging=staging[k], force=a.force_dist)
                ev_gather[k] = torch.cuda.Event()
                ev_gather[k].record(side)
                buf.record_stream(side)
        elif not a.rehearse:   # display words of the finished stripes -> rank 0 (RCCL over xGMI)
            staging[k] = tile_split.gather_stripes(buf, layout, dst=0, staging=staging[k], force=a.force_dist)
        else:
            host = buf.cpu()
            staging[k] = tile_split.gather_stripes(host, layout, dst=0, staging=staging[k])
            if rank == 0:
                buf.copy_(host)

    def last_frame():
        return rgbas[(step_no[0] - 1) % n_buf]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_steps, first, params=base, spp=a.spp, fresh=False, per_step=None, sync_each=False):
        """wall time of n_steps steps between barrier + synchronize; per_step (a list) also receives every step's device time
        from events on the launch stream (the median is reported beside the mean); sync_each: pt_sync before every call,
        the reference host's own loop (BasicScene.cpp:395)"""
        barrier()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)] if per_step is not None else None
        t0 = time.perf_counter()
        if evs:
            evs[0].record(stream)
        for k in range(n_steps):
            if sync_each:
                pt.sync()
            step(first + k, params, spp=spp, fresh=fresh)
            if evs:
                evs[k + 1].record(stream)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if a.rehearse else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        if evs:
            per_step.extend(evs[k].elapsed_time(evs[k + 1]) for k in range(n_steps))
        return dt

    # ------------------------------------------------------------------ the timed region
    # Library warm start, in front of the W warm-up steps and whatever W is: PT_KERNEL_AUTO's four trial calls and its decision
    # (per configuration: image, spp, partition), and the buffers either stage layout allocates on first use.  The first
    # warm-up step restarts the running mean (sample_index 1), so nothing of these calls is left in the frame.
    for k in range(6):
        step(k)
    torch.cuda.synchronize()
    for k in range(a.warmup):
        step(k)
    step_ms = []
    dt = timed(a.steps, a.warmup, per_step=step_ms)
    if a.pmc_child:   # the run rocprofv3 wraps (pmc_live): nothing but the timed steps
        if rank == 0:
            os.write(json_fd, (json.dumps({"pmc_child": True, "ms_per_step": round(dt / a.steps * 1e3, 4)}) + "\n").encode())
        pt.close()
        return

    # ---- everything below is outside the timed region ------------------------------------------
    # device time by stage (HIP events recorded by the library on the launch stream, PT_OPT_TIMING)
    pt.set_option(g.OPT_TIMING, 1)
    stage = {}
    n_tm = min(a.steps, 8)
    for k in range(n_tm):
        step(a.warmup + k)
        torch.cuda.synchronize()
        for name, ms in pt.stage_ms().items():
            stage[name] = stage.get(name, 0.0) + ms / n_tm
    pt.set_option(g.OPT_TIMING, 0)

    # exact segment / item counts of the timed frames: replay them instrumented — with the kernel that was TIMED (under
    # PT_KERNEL_AUTO an instrumented call would otherwise run the persistent kernel whatever the decision was)
    timed_kernel = a.kernel
    if a.kernel == g.KERNEL_AUTO:
        k_auto = pt.auto_choice()[0]
        timed_kernel = k_auto if k_auto != g.KERNEL_AUTO else g.KERNEL_PERSISTENT
    pt.set_option(g.OPT_KERNEL, timed_kernel)
    pt.set_option(g.OPT_COUNTERS, 1)
    seg = torch.zeros(6, dtype=torch.float64, device="cpu" if a.rehearse else dev)
    n_count = min(a.steps, 2)
    wstats = None
    for k in range(n_count):
        step(a.warmup + k)
        torch.cuda.synchronize()
        c = pt.counters()
        for j, key in enumerate(("rays", "paths", "inner", "tris", "leaves", "hits")):
            seg[j] += c[key]
        wstats = pt.wave_stats()
    pt.set_option(g.OPT_COUNTERS, 0)
    pt.set_option(g.OPT_KERNEL, a.kernel)
    if world > 1:
        dist.all_reduce(seg)
    rays_per_step, paths_per_step, items_nodes, items_recs = (float(seg[j].item()) / n_count for j in (0, 1, 2, 3))
    closed = abs(rays_per_step - W * H * a.depth * a.spp) < 0.5
    total_rays = rays_per_step * a.steps

    def rate(n_steps, seconds, spp=a.spp):
        return round(rays_per_step / a.spp * spp * n_steps / seconds / 1e6, 1)

    def settle(params=base, spp=a.spp):
        """untimed calls in front of a side measurement: PT_KERNEL_AUTO's four trial calls, its decision, and the
        buffers either stage layout allocates on first use"""
        for k in range(6):
            step(k, params, spp=spp)
        torch.cuda.synchronize()

    extra = {}
    if not a.no_extra:
        n_x = max(5, a.steps // 5)
        for name, m in (("metal", g.MAT_METAL), ("spec", g.MAT_SPEC)):
            if m == mat:
                continue
            pm = g.Params.from_buffer_copy(base)
            pm.tri_mat = m
            settle(pm)
            extra[f"mrays_per_s_{name}"] = round(W * H * a.depth * a.spp * n_x / timed(n_x, 1, pm) / 1e6, 1)
        # the reference's own granularity: ONE sample per launch (BasicScene.cpp:395-404), sync between launches
        # not needed (in-order stream)
        settle(spp=1)
        n_1 = max(20, a.steps)
        extra["mrays_per_s_1spp"] = rate(n_1, timed(n_1, 1, spp=1), spp=1)
        # ... and exactly the reference host's loop: cudaStreamSynchronize before every launch (BasicScene.cpp:395-404) —
        # what a maintainer following INTEGRATION.md section 2 gets; no call ever overlaps the previous one's tail
        extra["mrays_per_s_1spp_sync"] = rate(n_1, timed(n_1, 1, spp=1, sync_each=True), spp=1)
        if world == 1:
            for kname, kern in (("persistent", g.KERNEL_PERSISTENT), ("wavefront", g.KERNEL_WAVEFRONT)):
                pt.set_option(g.OPT_KERNEL, kern)
                settle()
                extra[f"mrays_per_s_{kname}"] = rate(n_x, timed(n_x, 1))
                settle(spp=1)
                extra[f"mrays_per_s_1spp_{kname}"] = rate(n_1, timed(n_1, 1, spp=1), spp=1)
                extra[f"mrays_per_s_1spp_sync_{kname}"] = rate(n_1, timed(n_1, 1, spp=1, sync_each=True), spp=1)
            pt.set_option(g.OPT_KERNEL, a.kernel)

    merged_ok = None
    accum_gather_ms = None
    if use_dist:
        # the last step once more, fresh (sample_index 1: the frame does not depend on accumulated history), then what north_star
        # names: the gather of the ACCUMULATED tiles (float accumulator stripes, 24.9 MB in all at 1080p) to rank 0, beside the
        # per-step gather of the display words; rank 0 re-renders the frame alone and compares both buffers
        last = a.warmup + a.steps
        step(last, fresh=True)
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        if a.rehearse:
            host_acc = accum.cpu()
            tile_split.gather_stripes(host_acc, layout, dst=0)
            if rank == 0:
                accum.copy_(host_acc)
        else:
            tile_split.gather_stripes(accum, layout, dst=0, force=a.force_dist)
        torch.cuda.synchronize()
        accum_gather_ms = (time.perf_counter() - t0) * 1e3
        if rank == 0:
            solo = g.Params.from_buffer_copy(base)
            solo.part_index, solo.part_count = 0, 1
            solo.frame, solo.sample_index = last * a.spp, 1
            acc2 = torch.zeros_like(accum)
            rgba2 = torch.zeros_like(rgba)
            pt.launch_kernel(acc2.data_ptr(), rgba2.data_ptr(), cam, solo, a.spp)
            torch.cuda.synchronize()
            merged_ok = bool(torch.equal(last_frame()[:H], rgba2[:H])) and bool(torch.equal(accum[:H], acc2[:H]))

    if rank == 0:
        value = total_rays / dt / 1e6
        out = {
            "metric": "Mrays/sec, cornell_dragon 1920x1080 (+ achieved GB/s against the memory roofline)",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "ms_per_step_median": round(float(np.median(step_ms)), 4) if step_ms else None,
            "timed_seconds": round(dt, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, gloo)" if a.rehearse else ""),
            "config": {"workload": f"{a.scene} ({mesh.n_tris} tris) {W}x{H} depth {a.depth} {a.mat} + "
                                   f"{'reference 8-sphere room' if n_sph else 'no spheres'}, {a.spp} spp per step",
                       "kernel": KERNEL_NAMES.get(a.kernel, str(a.kernel)),
                       "bvh": {"inner": info["n_inner"], "tri_refs": info["n_tri_refs"], "max_depth": info["max_depth"],
                               "device_mb": round(info["device_bytes"] / 2 ** 20, 1), "built_on": "device" if a.device_build else "host", "tree": tree_note,
                               "area_cost_node_visits": None if tree_cost is None else round(tree_cost[0], 3)},
                       "parallelism": (f"tile-split x{world} ({rows}-row stripes, "
                                       f"{'REHEARSAL: all ranks on one GPU, gloo gather through host memory' if a.rehearse else 'RCCL gather to rank 0 (ncclSend/ncclRecv group)'}"
                                       f" of RGBA8 every step{', overlapped with the next render' if overlap else ''}; float accumulator stripes gathered at the end)") if use_dist else "1 GPU",
                       "closed_scene": bool(closed), "rays_per_step": rays_per_step,
                       "tile_split_equals_single_gpu": merged_ok,
                       "accumulator_gather_ms": None if accum_gather_ms is None else round(accum_gather_ms, 3)},
            "stage_ms": {k: round(v, 4) for k, v in stage.items() if v > 0},
        }
        out.update(extra)
        # ---------------------------------------------------------------- roofline of the dominant kernel
        # units: one launch of the dominant kernel.  PT_KERNEL_WAVEFRONT launches k_wf_extend once per bounce
        # (`depth` launches per step); the other frame kernels are one launch per step.
        dom = max((k for k in ("frame", "extend", "shade") if stage.get(k, 0) > 0), key=lambda k: stage[k], default=None)
        launches = a.depth if dom in ("extend", "shade") else 1
        kname = {"frame": "k_trace_persist_bvh2" if a.kernel != 1 else "k_trace_mega_bvh2", "extend": "k_wf_extend", "shade": "k_wf_shade"}.get(dom)
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "kernel": kname, "launches_per_step": launches}
        if world == 1 and dom:
            kms = stage[dom] / launches
            roof["kernel_ms_avg"] = round(kms, 4)
            # TRAFFIC = bytes at the L2 <-> fabric boundary per launch (FETCH_SIZE + WRITE_SIZE), measured for THESE sources:
            # live child runs under rocprofv3 --pmc, else the committed profile when its source hash matches.
            # ACHIEVED = traffic / the kernel's HIP-event duration; FRAC = achieved / 8 TB/s — the fraction of the HBM roofline
            # the kernel really draws (north_star's figure).  Infinity-Cache hits are inside FETCH_SIZE (MI355X_MICROARCH.md),
            # so on a scene that fits the 256 MB cache this is an UPPER bound of what reaches HBM.
            scene_args = (["--host-splits"] if a.host_splits else []) + ["--scene", a.scene, "--width", str(W), "--height", str(H), "--spp", str(a.spp), "--depth", str(a.depth), "--mat", a.mat,
                          "--kernel", str(timed_kernel)] + (["--no-spheres"] if a.no_spheres else []) + (["--device-build"] if a.device_build else []) + \
                         (["--keep-hierarchy"] if a.keep_hierarchy else [])
            tr = None
            if not a.no_pmc and not a.no_extra:
                try:
                    tr = pmc_live(kname, scene_args)
                except Exception as e:
                    tr = None
                    roof["pmc_live_error"] = str(e)[:120]
            if tr is None:
                f = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
                if os.path.exists(f):
                    try:
                        j = json.load(open(f))
                        w = j.get("kernels", {}).get(kname)
                        if w and (a.scene, W, H, a.spp) == ("cornell_dragon_800k", 1920, 1080, 16):
                            if j.get("source_sha") == source_sha():
                                tr = dict(w, source="profiles/r03_pmc_traffic.json (rocprofv3 --pmc passes, same kernel sources)")
                            else:
                                roof["traffic_of_other_sources"] = {"hbm_bytes_per_launch": w.get("hbm_bytes_per_launch"),
                                                                    "note": "profiles/r03_pmc_traffic.json was taken with other kernel sources: not used for frac"}
                    except Exception:
                        pass
            if tr:
                hb = tr["hbm_bytes_per_launch"]
                roof["traffic"] = hb
                roof["achieved"] = round(hb / (kms * 1e-3) / 1e9, 1)
                roof["frac"] = round(hb / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                roof["traffic_detail"] = {k: tr[k] for k in ("fetch_bytes_raw", "fetch_bytes_x2", "write_bytes", "source") if k in tr}
            # REQUESTED = the bytes this kernel's algorithm asks the memory system for per launch: 64 B per item the walk fetches
            # (wide node or triangle record, counted by the instrumented replay of the timed kernel) + its own streams (extend:
            # 32-B ray in, 8-B hit out per segment; persistent: 12-B sample colour per path).  Served mostly by L1 / L2.
            items = (items_nodes + items_recs) / launches
            stream_b = (rays_per_step * 40.0 / launches) if dom == "extend" else (12.0 * paths_per_step)
            req = items * 64.0 + stream_b
            roof["requested"] = {"bytes_per_launch": int(req), "gbs": round(req / (kms * 1e-3) / 1e9, 1),
                                 "over_hbm_peak": round(req / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "traffic_over_requested": round(roof["traffic"] / req, 4) if roof["traffic"] else None}
            roof["items_per_ray"] = round((items_nodes + items_recs) / rays_per_step, 2)
            roof["gitems_per_s"] = round(items / (kms * 1e-3) / 1e9, 2)
            if wstats:
                roof["lane_use"] = {"node_steps": round(wstats["act_node"] / max(1, 64 * wstats["it_node"]), 3),
                                    "record_steps": round(wstats["act_rec"] / max(1, 64 * wstats["it_rec"]), 3),
                                    "stack_overflows_per_ray": round(wstats["stack_overflows"] / max(1.0, rays_per_step), 4)}
            # GATHER CEILING (secondary): the rate at which the memory hierarchy serves DEPENDENT random 64-byte items to every
            # lane of 8 waves/SIMD when every fetch is an L2 hit (tools/ubench_gather, a child process: 2 MB table) — what can
            # bind a per-lane walk over a cache-resident scene — and the same through a table of the scene's size.
            if not a.no_extra:
                try:
                    gb = gather_bound(info["device_bytes"])
                except Exception as e:
                    gb = {"error": str(e)[:100]}
                if gb and "items_per_s" in gb:
                    roof["gather_ceiling"] = {"gbs": round(gb["items_per_s_l2_resident"] * 64.0 / 1e9, 1),
                                              "gitems_per_s_l2_resident": round(gb["items_per_s_l2_resident"] / 1e9, 2),
                                              "gitems_per_s_uniform_over_scene": round(gb["items_per_s"] / 1e9, 2),
                                              "table_mb": round(gb["table_bytes"] / 2 ** 20, 1)}
                    roof["frac_of_gather_ceiling"] = round(items / (kms * 1e-3) / gb["items_per_s_l2_resident"], 4)
        if world == 1 and a.cpu_frames > 0:
            cb, cnt, ref_acc = cpu_baseline(g, bvh if bvh is not None else g.Bvh(mesh), sph, cam, base, a.warmup * a.spp, a.cpu_frames, a.spp)
            out["cpu_baseline"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in cb.items()}
            # parity of the TIMED configuration: the same first steps on the GPU, empty accumulator, vs the oracle
            acc_chk = torch.zeros_like(accum)
            for k in range(a.cpu_frames):
                pp = g.Params.from_buffer_copy(base)
                pp.frame, pp.sample_index = (a.warmup + k) * a.spp, 1 + k * a.spp
                pt.launch_kernel(acc_chk.data_ptr(), rgba.data_ptr(), cam, pp, a.spp)
            torch.cuda.synchronize()
            got = acc_chk[:H].cpu().numpy()
            diff = got.astype(np.float64) - ref_acc
            out["parity"] = {"l2": float(np.sqrt(np.mean(np.sum(diff ** 2, axis=-1)))),
                             "n_diff": int(np.any(got != ref_acc, axis=-1).sum()), "pixels": W * H,
                             "max_abs": float(np.abs(diff).max()),
                             "what": f"GPU accumulator after {a.cpu_frames} timed-configuration step(s) vs oracle/pt_oracle.c, same seeds"}
            # ... and of the two other triangle materials the line quotes rates for (configs[2]: metal, specular), at 2 spp
            if not a.no_extra:
                import orc
                for name, m in (("metal", g.MAT_METAL), ("spec", g.MAT_SPEC)):
                    if m == mat:
                        continue
                    pm = g.Params.from_buffer_copy(base)
                    pm.part_count, pm.part_index = 1, 0
                    pm.tri_mat, pm.frame, pm.sample_index = m, 3, 1
                    acc_m = torch.zeros_like(accum)
                    pt.launch_kernel(acc_m.data_ptr(), rgba.data_ptr(), cam, pm, 2)
                    torch.cuda.synchronize()
                    got_m = acc_m[:H].cpu().numpy()
                    ref_m, _, _ = orc.render(bvh if bvh is not None else g.Bvh(mesh), sph, cam, pm, spp=2, want_rgba=False)
                    d_m = got_m.astype(np.float64) - ref_m
                    out[f"parity_{name}"] = {"l2": float(np.sqrt(np.mean(np.sum(d_m ** 2, axis=-1)))), "n_diff": int(np.any(got_m != ref_m, axis=-1).sum()),
                                             "pixels": W * H, "what": f"2 spp, triangle material {name}, vs the oracle"}
            if dom:
                # SURVEY 8(d)'s figure, kept beside: ALGORITHMIC bytes of the REFERENCE layout (64 B per binary node, 48 B per
                # triangle, 16 B per leaf, 4 B per hit, 44 B per sphere, 28 B per pixel-sample), counted by the oracle on the
                # reference's own Compact arrays, over the whole step and over the HBM spec peak.  The caches serve these
                # bytes, so the quotient is not bounded by 1.
                alg_step = g.algorithmic_bytes(cnt, n_sph) / a.cpu_frames
                step_ms = sum(v for k, v in stage.items() if k != "none")
                roof["algorithmic_8d"] = {"bytes_per_step": int(alg_step), "bytes_per_ray": round(alg_step / (cnt["rays"] / a.cpu_frames), 1),
                                          "nodes_per_ray": round(cnt["inner"] / cnt["rays"], 2), "tris_per_ray": round(cnt["tris"] / cnt["rays"], 2),
                                          "gbs_whole_step": round(alg_step / (step_ms * 1e-3) / 1e9, 1),
                                          "over_hbm_peak": round(alg_step / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        roof["note"] = ("traffic = PMC bytes at the L2<->fabric boundary per launch of the dominant kernel (FETCH_SIZE raw + WRITE_SIZE, measured for "
                        "these sources); achieved = traffic / the kernel's HIP-event duration; peak = 8 TB/s (HBM3E spec); frac = achieved / peak. "
                        "requested = what the walk asks for (items x 64 B + streams; caches serve most of it); gather_ceiling / frac_of_gather_ceiling "
                        "= the measured L2-resident dependent-gather rate and the walk's share of it; algorithmic_8d = SURVEY 8(d)'s "
                        "reference-layout bytes over the whole step (served by caches: may exceed the HBM peak).  roofline_hbm_scene repeats "
                        "the exercise on a 1.2 GB item buffer that does not fit the 256 MB Infinity Cache")
        out["roofline"] = roof
        if world == 1 and not a.no_cpu_reference:
            try:
                ref = cpu_reference_tracer(g)
                if ref:
                    out["cpu_reference_tracer"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in ref.items()}
            except Exception as e:  # a baseline, never a reason to lose the GPU number
                out["cpu_reference_tracer"] = {"error": str(e)[:200]}

    if not a.no_extra and world == 1:
        # side measurement: the same workload over a tree built ON the device
        # (pt_build_bvh); done last, it replaces the scene of this context
        n_x = max(5, a.steps // 5)
        if bvh is not None and not a.keep_hierarchy:   # the uploaded hierarchy AS BUILT (no optimisation, no re-clustering) ...
            pt.upload_bvh(bvh)
            settle()
            out["mrays_per_s_uploaded_hierarchy"] = rate(n_x, timed(n_x, 1))
            out["area_cost_node_visits_uploaded_hierarchy"] = round(pt.tree_cost()[0], 3)
            pt.set_option(g.OPT_REBUILD, 1)            # ... and the device's re-clustered tree (PLOC), whatever the upload kept
            pt.upload_bvh(bvh)
            pt.set_option(g.OPT_REBUILD, 0)
            settle()
            out["mrays_per_s_reclustered_tree"] = rate(n_x, timed(n_x, 1))
            out["area_cost_node_visits_reclustered_tree"] = round(pt.tree_cost()[0], 3)
        build_ms = min(pt.build_bvh(mesh) for _ in range(3))
        settle()
        out["device_bvh_build_ms"] = round(build_ms, 2)          # PLOC (the default PT_OPT_BUILD_ALGO)
        out["mrays_per_s_device_built_tree"] = rate(n_x, timed(n_x, 1))
        pt.set_option(g.OPT_BUILD_ALGO, 0)                          # Karras LBVH: the fastest build
        out["device_bvh_build_ms_lbvh"] = round(min(pt.build_bvh(mesh) for _ in range(3)), 2)
        pt.set_option(g.OPT_BUILD_ALGO, 1)
        # EXTRA workload beyond the Infinity Cache (SURVEY §7 hard parts): cornell + 64 dragon copies, item buffer
        # ~0.8 GB > 256 MiB, tree built on the device; same camera / room / spp.  Not the headline.
        try:
            big = g.scene_mesh("cornell_dragon_6400k")
            pt.set_option(g.OPT_OPTIMIZE, 0 if a.keep_hierarchy else 3)   # the device's tree through the optimiser (~20 s for 7.8 M nodes)
            t_big = time.perf_counter()
            b_ms = pt.build_bvh(big)
            t_big = time.perf_counter() - t_big
            pt.set_option(g.OPT_OPTIMIZE, 0)
            binfo = pt.scene_info()
            settle()
            n_b = max(3, a.steps // 10)
            dtb = timed(n_b, 1)
            pt.set_option(g.OPT_TIMING, 1)
            step(1)
            torch.cuda.synchronize()
            bst = pt.stage_ms()
            pt.set_option(g.OPT_TIMING, 0)
            pt.set_option(g.OPT_COUNTERS, 1)
            step(1)
            torch.cuda.synchronize()
            bc = pt.counters()
            pt.set_option(g.OPT_COUNTERS, 0)
            bdom = max((k for k in ("frame", "extend") if bst.get(k, 0) > 0), key=lambda k: bst[k])
            b_launches = a.depth if bdom == "extend" else 1
            b_kname = "k_wf_extend" if bdom == "extend" else "k_trace_persist_bvh2"
            b_kms = bst[bdom] / b_launches
            b_items = bc["inner"] + bc["tris"]
            b_stream = bc["rays"] * 40.0 if bdom == "extend" else 12.0 * bc["paths"]
            b_req = (b_items * 64.0 + b_stream) / b_launches
            out["big_scene"] = {"workload": f"cornell_dragon_6400k ({big.n_tris} tris) {W}x{H} depth {a.depth} {a.mat} + sphere room, {a.spp} spp per step",
                                "device_mb": round(binfo["device_bytes"] / 2 ** 20, 1), "device_build_ms": round(b_ms, 1),
                                "tree": f"built on the device (PLOC), then PT_OPT_OPTIMIZE 3 on the host: {t_big:.1f} s in all",
                                "mrays_per_s": round(bc["rays"] * n_b / dtb / 1e6, 1), "ms_per_step": round(dtb / n_b * 1e3, 3),
                                "stage_ms": {k: round(v, 3) for k, v in bst.items() if v > 0},
                                "items_per_ray": round(b_items / bc["rays"], 2)}
            # the roofline record of the workload that does leave the caches: same fields as `roofline`
            rb = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "kernel": b_kname,
                  "launches_per_step": b_launches, "kernel_ms_avg": round(b_kms, 4),
                  "requested": {"bytes_per_launch": int(b_req), "gbs": round(b_req / (b_kms * 1e-3) / 1e9, 1),
                                "over_hbm_peak": round(b_req / (b_kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                  "items_per_ray": round(b_items / bc["rays"], 2), "gitems_per_s": round(b_items / b_launches / (b_kms * 1e-3) / 1e9, 2)}
            trb = None
            if not a.no_pmc:
                try:
                    trb = pmc_live(b_kname, ["--scene", "cornell_dragon_6400k", "--device-build", "--device-optimize", "3", "--width", str(W), "--height", str(H), "--spp", str(a.spp),
                                             "--depth", str(a.depth), "--mat", a.mat, "--kernel", str(g.KERNEL_WAVEFRONT if bdom == "extend" else g.KERNEL_PERSISTENT)])
                except Exception as e:
                    rb["pmc_live_error"] = str(e)[:120]
            if trb is None:
                f = os.path.join(ROOT, "profiles", "r03_big_scene_pmc_traffic.json")
                if os.path.exists(f):
                    jb = json.load(open(f))
                    wb = jb.get("kernels", {}).get(b_kname)
                    if jb.get("source_sha") == source_sha() and wb and (W, H, a.spp, a.depth) == (1920, 1080, 16, 4):
                        trb = dict(wb, source="profiles/r03_big_scene_pmc_traffic.json (rocprofv3 --pmc passes, same kernel sources)")
            if trb:
                rb["traffic"] = trb["hbm_bytes_per_launch"]
                rb["achieved"] = round(trb["hbm_bytes_per_launch"] / (b_kms * 1e-3) / 1e9, 1)
                rb["frac"] = round(trb["hbm_bytes_per_launch"] / (b_kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                rb["requested"]["traffic_over_requested"] = round(trb["hbm_bytes_per_launch"] / b_req, 4)
                rb["traffic_detail"] = {k: trb[k] for k in ("fetch_bytes_raw", "fetch_bytes_x2", "write_bytes", "source") if k in trb}
            gb = gather_bound(binfo["device_bytes"])
            if gb and "items_per_s" in gb:
                rb["gather_ceiling"] = {"gbs": round(gb["items_per_s_l2_resident"] * 64.0 / 1e9, 1),
                                        "gitems_per_s_l2_resident": round(gb["items_per_s_l2_resident"] / 1e9, 2),
                                        "gitems_per_s_uniform_over_scene": round(gb["items_per_s"] / 1e9, 2),
                                        "table_mb": round(gb["table_bytes"] / 2 ** 20, 1)}
                rb["frac_of_gather_ceiling"] = round(b_items / b_launches / (b_kms * 1e-3) / gb["items_per_s_l2_resident"], 4)
                rb["over_uniform_gather_rate"] = round(b_items / b_launches / (b_kms * 1e-3) / gb["items_per_s"], 4)
            rb["note"] = ("same fields as `roofline`, for cornell_dragon_6400k (0.75-1.2 GB of items: beyond the 256 MB Infinity Cache); "
                          "the tree is the device builder's, optimised on the host")
            out["roofline_hbm_scene"] = rb
            # in-run parity of this workload: one frame of `--big-parity-spp` samples over the DEVICE tree against the oracle's walk
            # over a HOST tree of the same mesh (checker, after every timed region), + a ray batch against brute force
            if a.big_parity_spp > 0:
                import orc
                pb = g.Params.from_buffer_copy(base)
                pb.frame, pb.sample_index = 7, 1
                acc_b = torch.zeros_like(accum)
                pt.launch_kernel(acc_b.data_ptr(), rgba.data_ptr(), cam, pb, a.big_parity_spp)
                torch.cuda.synchronize()
                got_b = acc_b[:H].cpu().numpy()
                t0 = time.perf_counter()
                hb = g.Bvh(big, split_alpha=-1.0)
                t_build = time.perf_counter() - t0
                ref_b, _, cnt_b = orc.render(hb, sph, cam, pb, spp=a.big_parity_spp, want_rgba=False)
                d_b = got_b.astype(np.float64) - ref_b
                lo_b, hi_b = big.bounds()
                rays_b = orc.random_rays(2048, lo_b, hi_b, seed=5)
                d_rays = torch.from_numpy(rays_b).to(dev)
                d_t = torch.empty(len(rays_b), dtype=torch.float32, device=dev)
                d_i = torch.empty(len(rays_b), dtype=torch.int32, device=dev)
                pt.trace_rays(d_rays.data_ptr(), len(rays_b), True, d_t.data_ptr(), d_i.data_ptr(), None)
                torch.cuda.synchronize()
                tb_, ib_, _ = orc.trace_brute(big, rays_b)
                out["big_scene"]["parity"] = {"l2": float(np.sqrt(np.mean(np.sum(d_b ** 2, axis=-1)))),
                                              "n_diff": int(np.any(got_b != ref_b, axis=-1).sum()), "pixels": W * H, "max_abs": float(np.abs(d_b).max()),
                                              "ray_batch_vs_brute_force": {"rays": len(rays_b), "t_equal": bool(np.array_equal(d_t.cpu().numpy(), tb_)),
                                                                           "id_equal": bool(np.array_equal(d_i.cpu().numpy(), ib_)), "hit_share": round(float((ib_ >= 0).mean()), 3)},
                                              "what": f"device-built tree, {a.big_parity_spp} spp, vs oracle/pt_oracle.c over a host SAH tree of the same mesh "
                                                      f"(built in {t_build:.1f} s), same seeds; + pt_trace_rays vs the brute-force loop"}
        except Exception as e:
            import traceback
            out["big_scene"] = {"error": str(e)[:200], "where": traceback.format_exc().strip().splitlines()[-3:]}

    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    pt.close()


if __name__ == "__main__":
    main()

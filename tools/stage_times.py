#!/usr/bin/env python3
"""Stage times (PT_OPT_TIMING) of the stage-split pipeline for one configuration under PT_OPT_WAVE_SAMPLES settings.
Usage: stage_times.py scene W H spp spheres(0/1) [wave_samples ...]   e.g. stage_times.py cornell 1280 720 16 0 1 16"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import gpu_pathtracer_amd as g
scene, W, H, spp, spheres = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
caps = [int(x) for x in sys.argv[6:]] or [1, 16]
kernel = int(os.environ.get("PT_KERNEL", g.KERNEL_WAVEFRONT))
bvh = g.Bvh(g.scene_mesh(scene))
for rnd in range(2):
    for cap in caps:
        pt = g.PathTracer(0)
        pt.set_option(g.OPT_KERNEL, kernel)
        pt.set_option(g.OPT_WAVE_SAMPLES, cap)
        pt.upload_bvh(bvh)
        pt.upload_spheres(g.reference_spheres() if spheres else None)
        cam = g.default_camera(W, H)
        if scene == "dragon":
            cam.dist = 18.0
        acc, rgba = pt.alloc_frame(W, H)
        def launch(f):
            p = g.default_params(W, H)
            p.frame, p.sample_index, p.flags = f * spp, 1 + f * spp, g.FLAG_WRITE_RGBA
            pt.launch_kernel(acc.ptr, rgba.ptr, cam, p, spp)
        for f in range(3):
            launch(f)
        pt.sync()
        t0 = time.perf_counter()
        for f in range(10):
            launch(3 + f)
        pt.sync()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        pt.set_option(g.OPT_TIMING, 1)
        launch(13); pt.sync(); launch(14); pt.sync()
        st = pt.stage_ms()
        print(f"round {rnd} wave-samples {cap:2d}: {ms:7.3f} ms/call  stages " + " ".join(f"{k} {v:.3f}" for k, v in st.items() if v > 0), flush=True)
        if rnd == 0 and os.environ.get("PT_STATS"):   # instrumented replay: wave-level iterations of each phase and the lanes active in them
            pt.set_option(g.OPT_TIMING, 0)
            pt.set_option(g.OPT_COUNTERS, 1)
            launch(15); pt.sync()
            c, w = pt.counters(), pt.wave_stats()
            print("   counters", c)
            print("   wave stats", w, f"| per ray: loop iterations x64 {64 * w['it_loop'] / c['rays']:.2f}, node steps x64 {64 * w['it_node'] / c['rays']:.2f} "
                  f"(lane use {w['act_node'] / max(1, 64 * w['it_node']):.2f}), record steps x64 {64 * w['it_rec'] / c['rays']:.2f}, begin x64 {64 * w['it_begin'] / max(1, c['rays']):.2f}", flush=True)
            pt.set_option(g.OPT_COUNTERS, 0)
        acc.free(); rgba.free(); pt.close()

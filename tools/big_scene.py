import sys, time
sys.path[:0]=["/root/repo","/root/repo/tests"]
import gpu_pathtracer_amd as g
W,H=1920,1080
for name in ("cornell_dragon_2700k","cornell_dragon_6400k"):
    mesh=g.scene_mesh(name)
    for kern in (g.KERNEL_PERSISTENT, g.KERNEL_WAVEFRONT, g.KERNEL_AUTO):
        pt=g.PathTracer(0); pt.set_option(g.OPT_KERNEL, kern)
        try:
            ms=pt.build_bvh(mesh); info=pt.scene_info()
            pt.upload_spheres(g.reference_spheres())
            cam=g.default_camera(W,H); acc,rgba=pt.alloc_frame(W,H)
            def run(n):
                for f in range(n):
                    p=g.default_params(W,H); p.frame,p.sample_index=f*16,1+f*16; p.flags=g.FLAG_WRITE_RGBA
                    pt.launch_kernel(acc.ptr,rgba.ptr,cam,p,16)
            run(6); pt.sync()
            t0=time.perf_counter(); run(5); pt.sync()
            dt=(time.perf_counter()-t0)/5*1e3
            print(name, kern, f"build {ms:.1f} ms, {info['device_bytes']/2**20:.0f} MB, {dt:.2f} ms/step = {W*H*64/dt/1e3:.0f} Mrays/s", pt.auto_choice(), flush=True)
        except Exception as e:
            print(name, kern, "ERROR", e, flush=True)
        pt.close()

import sys, time; sys.path[:0]=["/root/repo","/root/repo/tests"]
import numpy as np, gpu_pathtracer_amd as g
W,H=1920,1080
bvh=g.Bvh(g.scene_mesh("cornell_dragon_800k")); pt=g.PathTracer(0); pt.upload_bvh(bvh); pt.upload_spheres(g.reference_spheres())
cam=g.default_camera(W,H); acc,rgba=pt.alloc_frame(W,H)
import os
SPP = int(os.environ.get("SPP", "1"))
print("spp per launch", SPP)
for count in (1,2,4,8):
    for kern in (g.KERNEL_PERSISTENT, g.KERNEL_MEGA_BVH2):
        pt.set_option(g.OPT_KERNEL, kern)
        ts=[]
        for part in range(count):
            best=1e9
            for r in range(3):
                pt.sync(); t0=time.perf_counter()
                for f in range(10):
                    p=g.default_params(W,H); p.frame,p.sample_index=f*SPP,1+f*SPP; p.flags=g.FLAG_WRITE_RGBA
                    p.part_index,p.part_count,p.part_rows=part,count,8
                    pt.launch_kernel(acc.ptr,rgba.ptr,cam,p,SPP)
                pt.sync(); best=min(best,(time.perf_counter()-t0)/10*1e3)
            ts.append(best)
        print(f"parts {count} kernel {kern}: per-part ms min {min(ts):.3f} max {max(ts):.3f}  -> ideal-scaling speedup vs 1 part = see max")

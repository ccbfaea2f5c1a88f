/*
 * ptmi.h — C ABI of the MI355X-native progressive path tracer ("libptmi.so").
 *
 * This is the drop-in boundary for the ONE host→device call of the reference:
 *
 *     void BasicScene::launchKernel(const kernelInfo&)
 *         declared  GpuPathTracer/BasicScene.hpp:32
 *         defined   GpuPathTracer/tracer.cu:405-415   (trace<<<grid,16x16>>>(info))
 *         called    GpuPathTracer/BasicScene.cpp:404   (once per displayed frame)
 *
 * plus the device-buffer set-up the reference's constructor performs for that call
 * (cudaMalloc/cudaMemcpy of the three CudaBVH arrays and the sphere array,
 * GpuPathTracer/BasicScene.cpp:138-149, :214-215, :297-313).
 *
 * Everything crossing the boundary is plain C: fixed-width integers, float arrays,
 * raw pointers and sizes.  No glm, no bool, no C++ default initialisers (kernelInfo,
 * GpuPathTracer/CpuStructs.hpp:45-72, is not a C layout), no torch types.
 *
 * Conventions
 *   - every function returns 0 on success and a negative pt_status on failure; nothing
 *     ever calls exit() (the reference's checkCudaErrors does: utilfun.hpp:81-90);
 *     pt_last_error(ctx) returns a human-readable message for the last failure.
 *   - a ctx binds one HIP device and one stream.  Calls on one ctx are not re-entrant;
 *     different ctxs may be driven from different threads / processes (one per GPU).
 *   - pt_render is asynchronous with respect to the host until pt_sync.
 *   - "device pointer" arguments may come from pt_malloc or from any other HIP
 *     allocator in the same process (hipMalloc, a torch tensor's data_ptr()).
 */
#ifndef PTMI_H
#define PTMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTMI_ABI_VERSION 3

typedef struct pt_ctx pt_ctx;

typedef enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID = -1,   /* bad argument (null pointer, zero size, inconsistent arrays) */
    PT_ERR_DEVICE = -2,    /* a HIP call failed (message has hipGetErrorString)           */
    PT_ERR_NO_SCENE = -3,  /* pt_render / pt_trace_rays before pt_upload_bvh              */
    PT_ERR_NOMEM = -4,
    PT_ERR_UNSUPPORTED = -5
} pt_status;

/* Mat, GpuPathTracer/CommomStructs.hpp:12 — same numeric values. */
enum { PT_MAT_DIFF = 0, PT_MAT_METAL = 1, PT_MAT_SPEC = 2, PT_MAT_REFR = 3 };

/* pt_params.flags */
enum {
    /* METAL lobe adds float(image width)*cos(theta) to every component, exactly as
     * tracer.cu:280 does (`w` there is the int image width from tracer.cu:45).
     * Default (flag clear) is the intended  w1*cos(theta)  (SURVEY.md §3.4 table). */
    PT_FLAG_METAL_LITERAL_W = 1u << 0,
    /* also write the 0x00BBGGRR display word (tracer.cu:394-398); needs rgba_dev. */
    PT_FLAG_WRITE_RGBA = 1u << 1,
    /* Corrected-estimator switches — EXTENSIONS, all off by default so that the default path is the
     * reference's arithmetic, quirks included.  Oracle: same flags. */
    PT_FLAG_FACE_FORWARD = 1u << 2,  /* triangles: nl = dot(n,d) < 0 ? n : -n — what tracer.cu:126-127
                                        computes and then discards (nl = n in the reference)           */
    PT_FLAG_COSINE_DIFF = 1u << 3,   /* DIFF lobe: cosine-weighted direction from TWO draws (smallpt's
                                        estimator; mask *= col is then exact) instead of the reference's
                                        uniform hemisphere with no cosine term and two discarded draws
                                        (tracer.cu:159-186)                                            */
    PT_FLAG_GLASS_FIX = 1u << 4,     /* REFR: R0 = ((nt-nc)/(nt+nc))^2 (the reference's :230 multiplies
                                        where it should divide), reflection chosen with probability
                                        P = .25 + .5 Re (the weights RP/TP assume it; the reference
                                        uses 0.2), transmitted rays start on the far side (-nl)       */
    PT_FLAG_RUSSIAN_ROULETTE = 1u << 5,/* from the 3rd segment on: continue with probability
                                        p = max(col) and divide col by p, else end the path          */
    PT_FLAG_RR_CPU_TRACER = 1u << 7,   /* Russian roulette exactly as the reference's CPU tracer plays it
                                        (CpuRayTracer/src/scene.cpp:38-47): from the 6th hit of a path on, continue
                                        with probability 0.9 p, p = max(col), and scale col by 0.9 / p — an expected
                                        0.81 per bounce: that renderer's energy loss, reproduced so that its images
                                        can be matched; PT_FLAG_RUSSIAN_ROULETTE is the unbiased one            */
    PT_FLAG_NEE = 1u << 8,             /* next-event estimation at DIFF hits (needs PT_FLAG_COSINE_DIFF): one shadow ray towards
                                        ONE of the emissive spheres the hit point is outside of (of the first 8 spheres;
                                        picked uniformly, cone-sampled, weighted by their number), and a DIFF-sampled ray
                                        that then lands on such a sphere does not count its emission again.  With a material
                                        table on the context (pt_upload_tri_materials) the TRIANGLES whose row emits are
                                        lights too: the pick is uniform over eligible spheres + emissive triangles, a point
                                        is drawn uniformly on the triangle (the faces a path can hit emit: the front one
                                        under cull_backfaces, both otherwise), and the next DIFF-sampled hit on any emissive triangle is not counted again.
                                        Spheres the path is inside of (the reference room's glowing walls) and triangles
                                        lit by the one global material of pt_params are gathered by the bounce as before.
                                        Same expectation, less noise for small lights; runs in the stage-split pipeline
                                        (a shadow-ray stage per bounce) or the megakernel                                   */
    PT_FLAG_MISS_KEEPS_PATH = 1u << 6  /* a segment that hits nothing ends the path with
                                        accu + mask * bk_color (smallpt, and the reference's own CPU
                                        tracer: CpuRayTracer/src/scene.cpp:27 returns black for the
                                        MISSING TERM only) instead of the reference GPU kernel's bare
                                        bk_color, which throws the gathered light away
                                        (tracer.cu:140-142)                                          */
};
/* The estimator of the reference's CPU tracer (CpuRayTracer/src/scene.cpp:23-56, material.cpp:24-45: smallpt's):
 * cosine-weighted diffuse lobe, normals facing the ray, Russian roulette, light only where a path ENDS on
 * emission.  With bk_color = 0, an emitter given col = 0 and a generous depth the expected radiance equals
 * that renderer's (tests/test_reference_radiance.py compares them). */
#define PT_FLAGS_SMALLPT (PT_FLAG_FACE_FORWARD | PT_FLAG_COSINE_DIFF | PT_FLAG_RUSSIAN_ROULETTE | PT_FLAG_MISS_KEEPS_PATH)
/* ... and with that renderer's own roulette instead of the unbiased one: its images, bias included */
#define PT_FLAGS_CPU_TRACER (PT_FLAG_FACE_FORWARD | PT_FLAG_COSINE_DIFF | PT_FLAG_RR_CPU_TRACER | PT_FLAG_MISS_KEEPS_PATH)

/* pt_ctx kernel selection (pt_set_option PT_OPT_KERNEL).  The values 2 and 4 of ABI 1-2 (a reserved name and the
 * round-1 role-split experiment) are gone with ABI 3: pt_set_option rejects them with PT_ERR_UNSUPPORTED. */
enum {
    PT_KERNEL_AUTO = 0,      /* PT_KERNEL_PERSISTENT or PT_KERNEL_WAVEFRONT, whichever is faster for the
                                configuration (image, samples per call, depth, partition shape, scene, material,
                                flags): the first four pt_render calls of a configuration are timed trials, two per
                                layout (HIP events; buffers are allocated before the timed span; the faster trial
                                of each layout counts) and the following ones run the faster; the images are the same.  The decision never blocks the
                                host: while a trial's events are still pending, calls run the persistent kernel.
                                The library remembers the last 8 configurations.  pt_auto_choice reports it   */
    PT_KERNEL_MEGA_BVH2 = 1, /* one lane per pixel, one wave per 8x8 tile, bounce by bounce      */
    PT_KERNEL_PERSISTENT = 3,/* persistent waves: work queue, ballot/prefix-count lane refill    */
    PT_KERNEL_WAVEFRONT = 5  /* stage split (BASELINE.json configs[4]): path records in HBM, one
                                launch per stage — per bounce extend (persistent waves walking the
                                BVH, lanes refilled from the ray queue) and shade (one lane per
                                live path; survivors compacted with a ballot / prefix count into
                                the next generation).  Needs a BVH, depth >= 1 and the wide walk
                                over exact records; anything else runs PT_KERNEL_PERSISTENT.  Same images */
};

enum {
    PT_OPT_KERNEL = 1,        /* one of PT_KERNEL_*                                        */
    PT_OPT_COUNTERS = 2,      /* 1 = instrumented launch: fill pt_counters (slower)         */
    PT_OPT_TIMING = 3,        /* 1 = bracket every launch with hipEvents (pt_last_kernel_ms) */
    PT_OPT_BATCH = 4,         /* persistent kernel: waiting lanes (1..64) that make a wave
                                 leave the traversal loop to shade / refill; default 36      */
    PT_OPT_TOP_NODES = 5,     /* BVH nodes (breadth-first prefix, 0..1024) mirrored in LDS    */
    PT_OPT_OCCUPANCY = 6,     /* waves per SIMD the registers are budgeted for: 4/5/6/8 (default 6) */
    PT_OPT_LDS_STACK = 7,     /* traversal-stack entries kept in LDS per lane: 16 (default), 24 or
                                 0 = all 72; deeper entries overflow to private memory        */
    PT_OPT_WALK = 8,          /* closest-hit walk: 0 = while-while (Aila-Laine order, as the
                                 reference), 1 = unified-step over the binary tree,
                                 2 (default) = wide: unified-step over a 4-way tree with
                                 8-bit outward-rounded boxes; 4 = wide with one postponed leaf
                                 per lane (persistent kernel; the megakernel runs walk 2): a few
                                 per cent faster at 5 waves/SIMD, level at 8; all report the
                                 same hits                                                      */
    PT_OPT_REFILL = 11,       /* persistent kernel: idle lanes (1..64) that trigger a refill from the
                                 work queue; default 8; values above PT_OPT_BATCH are
                                 clamped to it (a wave must always have work to go to)          */
    PT_OPT_VOTE_NODE = 12,    /* walk 4: a wave runs a node step when                              */
    PT_OPT_VOTE_REC = 13,     /*   lanes_with_node * VOTE_NODE >= lanes_with_record * VOTE_REC (1, 1) */
    PT_OPT_WAVE_BATCH = 14,   /* PT_KERNEL_WAVEFRONT, extend stage: finished lanes (1..64) that make a wave leave the
                                 walk to store their hits and take new rays from the queue; default 16 */
    PT_OPT_SPHERE_LDS = 15,   /* persistent kernel: 1 (default) = the shading code reads the spheres from an LDS
                                 copy instead of scalar / global loads                            */
    PT_OPT_WAVE_BLOCKS = 19,  /* PT_KERNEL_WAVEFRONT, extend stage: resident 256-thread blocks per CU its persistent grid is
                                 sized for, 1..8 (default 8 = 8 waves per SIMD); fewer leave room for another
                                 context's launches on the same device                                 */
    PT_OPT_WAVE_SAMPLES = 25, /* PT_KERNEL_WAVEFRONT, bounce 0: how many samples of ONE pixel share a wave — the largest power of two
                                 <= this value (4 .. 64) that divides the call's spp; 16 by default (a wave = 16 samples x a 2x2
                                 pixel block: the samples of a pixel are the same ray but for the sub-pixel jitter, so they walk the
                                 same nodes and records in step), 1 = one sample of a whole 8x8 tile per wave as in the other
                                 kernels.  A speed knob: which lane traces which (pixel, sample) changes no result       */
    PT_OPT_OVERLAP = 21,      /* 1 (default): the path kernel of a pt_render call (persistent / mega kernels) runs on a
                                 stream of the context's own, so that it can start while the PREVIOUS call's last paths
                                 drain; the fold into the accumulator stays on the caller's stream, in call order.  When
                                 the caller's stream is idle at the call — a host that syncs before every launch, as
                                 BasicScene.cpp:395 does — the call runs in line on the caller's stream instead;
                                 0 = always in line                                                             */
    PT_OPT_BUILD_ALGO = 16,   /* pt_build_bvh: 1 (default) = PLOC (locally-ordered clustering over Morton order:
                                 a tree as good as the host SAH/SBVH builder's, ~6.5 ms for 800 k triangles;
                                 degenerate input falls back to 0), 0 = LBVH (Karras hierarchy: 1.8 ms, a
                                 tree that traces ~14 % slower)                                    */
    PT_OPT_REBUILD = 17,      /* pt_upload_bvh: 1 = keep the uploaded TRIANGLES but build the hierarchy again on
                                 the device (PT_OPT_BUILD_ALGO).  Same closest hits, hence the same images, except
                                 where a ray GRAZES a bounding plane: the binary32 slab test is not watertight, so
                                 a hit whose box is missed by one rounding in one tree and kept in the other can
                                 differ (measured: 2 of 2 073 600 pixels of the 16-spp 800k-triangle bench frame;
                                 every such pixel is arbitrated against brute force in tests/test_gpu_wide.py).
                                 Faster or slower than the caller's tree depending on the scene.
                                 2 = build it as well and keep whichever 4-wide tree has the smaller area cost
                                 in node visits (pt_tree_cost: the term that tracks the measured frame time,
                                 DESIGN.md 5.5) — the caller's on architectural scenes with long triangles
                                 that its spatial splits cut, the re-clustered one on dense scans; costs one
                                 device build (7 ms for 800 k triangles) per upload.  Default 0           */
    PT_OPT_OPTIMIZE = 24,     /* pt_upload_bvh / pt_build_bvh: 0 (default) = walk the hierarchy as it comes (after PT_OPT_LEAF_MAX); n > 0 = n
                                 passes of insertion-based optimisation over it first (Bittner et al. 2013: every node is taken out
                                 and re-inserted where the tree's surface-area cost grows least; csrc/pt_tree_opt.h) — on the host's
                                 threads, ~0.2 s per pass and million nodes on 64 threads; three passes get nearly all there is.  Same
                                 closest hits (grazing cases as PT_OPT_REBUILD).  A device-built tree (pt_build_bvh, PT_OPT_REBUILD)
                                 is fetched back, optimised and installed again.  cornell_dragon_800k: area cost in node visits /
                                 bench step, host SBVH tree 14.4 / 9.65 ms -> 12.6 / 9.00; host SAH tree without spatial splits
                                 14.6 / 9.95 -> 11.7 / 8.87; the device's PLOC tree 12.75 / 9.26 -> 12.1 / 9.15.  With
                                 PT_OPT_REBUILD 2 the two optimised hierarchies compete                                         */
    PT_OPT_PRESPLIT = 18,     /* pt_build_bvh / PT_OPT_REBUILD: 0 (default) = off; v > 0 = triangles longer than
                                 v per cent of (scene diagonal / sqrt(n triangles)) enter the builder as up
                                 to 8 primitives, one per slab of their box (early split clipping)      */
    PT_OPT_TRI_TEST = 10,     /* triangle records built at the next pt_upload_bvh: 0 = v0/e1/e2
                                 for Moller-Trumbore, what the reference kernel runs
                                 (cudaUtils.h:135-172; default, bit-exact vs the oracle);
                                 1 = Woop affine rows (north_star; CudaBVH.cpp:274-305 done
                                 right), tolerance-class parity, wide walk only               */
    PT_OPT_LEAF_MAX = 9       /* leaves holding more triangle references than this are split
                                 at the next pt_upload_bvh (0 = keep the producer's leaves;
                                 default 2)                                                   */
};

/* CamInfo, GpuPathTracer/CpuStructs.hpp:19-28 (pitch/yaw/dirty/bias/enabled are host-only
 * GUI state and never read by the kernel: cudaUtils.h:111-134). */
typedef struct pt_camera {
    float pos[3];
    float front[3];
    float right[3];
    float up[3];
    float dist;
    float aspect;
    float fov;
    float _pad;
} pt_camera;

/* Sphere, GpuPathTracer/CommomStructs.hpp:18-39 — 44 bytes, same field order. */
typedef struct pt_sphere {
    float pos_rad[4];  /* centre xyz, radius */
    float emi[3];
    float col[3];
    int32_t mat;       /* PT_MAT_* */
} pt_sphere;

/* The scalar part of kernelInfo (GpuPathTracer/CpuStructs.hpp:45-72) that the kernel
 * reads (tracer.cu:27-400); pointer members travel as explicit arguments. */
typedef struct pt_params {
    int32_t width, height;        /* kernelInfo::width/height                              */
    uint32_t depth;               /* kernelInfo::depth          (tracer.cu:72)             */
    int32_t cull_backfaces;       /* kernelInfo::cullBackFaces  (cudaUtils.h:151-155)      */
    uint64_t frame;               /* frameNumber; the kernel seed is uf::hash(frame)
                                     (BasicScene.cpp:397, utilfun.cpp:380-389)             */
    uint64_t sample_index;        /* kernelInfo::constantPdf = N of the running mean for
                                     the first sample of this call; 1 = overwrite
                                     (BasicScene.cpp:399, tracer.cu:386-391)               */
    int32_t tri_mat;              /* kernelInfo::triCurrentMat  (tracer.cu:135)            */
    float tri_col[3];             /* kernelInfo::col            (tracer.cu:131)            */
    float tri_emi[3];             /* kernelInfo::emi            (tracer.cu:132)            */
    float bk_color[3];            /* kernelInfo::bkColor        (tracer.cu:141)            */
    float air_ior;                /* kernelInfo::air_ref_index                              */
    float glass_ior;              /* kernelInfo::glass_ref_index                            */
    float phong_expo;             /* kernelInfo::phongExpo                                  */
    uint32_t flags;               /* PT_FLAG_*                                              */
    /* Framebuffer partition for multi-GPU tile split (new; the reference is single-GPU).
     * The image is cut into stripes of `part_rows` rows; this call renders only stripes
     * s with  s % part_count == part_index.  part_count <= 1 renders everything.
     * RNG is keyed by the GLOBAL pixel index, so any partition gives bit-identical
     * pixels.  accum/rgba are always full-frame buffers. */
    int32_t part_index, part_count, part_rows;
    int32_t _pad;
} pt_params;

/* Work counters of the last instrumented launch (PT_OPT_COUNTERS=1).  They are the
 * N_* of the algorithmic-byte definition in SURVEY.md §8(d). */
typedef struct pt_counters {
    uint64_t rays;        /* closest-hit queries (ray segments)          */
    uint64_t inner;       /* inner nodes fetched                          */
    uint64_t tris;        /* triangle records tested                      */
    uint64_t leaves;      /* leaves entered                               */
    uint64_t hits;        /* segments that ended on a triangle            */
    uint64_t paths;       /* pixel-samples                                */
} pt_counters;

/* ---- context ------------------------------------------------------------------ */
int pt_abi_version(void);
int pt_device_count(void);                       /* <0 on error                     */
int pt_create(int device, pt_ctx** out);
int pt_destroy(pt_ctx* ctx);
const char* pt_last_error(const pt_ctx* ctx);     /* ctx may be NULL: global message  */
int pt_set_stream(pt_ctx* ctx, void* hip_stream); /* NULL = ctx's own stream          */
int pt_set_option(pt_ctx* ctx, int option, int value);
int pt_sync(pt_ctx* ctx);

/* ---- memory (thin; callers may also pass pointers from their own allocator) ----- */
int pt_malloc(pt_ctx* ctx, size_t bytes, void** dev_out);
int pt_free(pt_ctx* ctx, void* dev);
int pt_memset(pt_ctx* ctx, void* dev, int value, size_t bytes);
int pt_download(pt_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);
int pt_upload(pt_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes);

/* ---- scene -------------------------------------------------------------------
 * pt_upload_bvh consumes the three host arrays CudaBVH::createCompact produces
 * (GpuPathTracer/CudaBVH.cpp:121-270) — exactly what BasicScene.cpp:297-306 copies:
 *   nodes      getGpuNodes()/getGpuNodesSize():  vec4[n_node_vec4], 4 per inner node,
 *              child link >=0 = BYTE offset of the child node, <0 = ~(first tri vec4)
 *   tri_verts  getDebugTri()/getDebugTriSize():  vec4[n_tri_vec4], per leaf
 *              {v0,v1,v2 (xyz,0)}*k followed by one 0x80000000 terminator vec4
 *   tri_index  getGpuTriIndices():               int[n_index], parallel to tri_verts
 * The arrays are validated, copied and re-laid-out for gfx950; the host
 * arrays may be freed when the call returns. */
int pt_upload_bvh(pt_ctx* ctx,
                  const float* nodes, size_t n_node_vec4,
                  const float* tri_verts, size_t n_tri_vec4,
                  const int32_t* tri_index, size_t n_index);
int pt_upload_spheres(pt_ctx* ctx, const pt_sphere* spheres, size_t n_spheres);

/* Build the acceleration structure ON THE DEVICE from an indexed triangle mesh — an EXTENSION
 * (the reference builds on the host: SplitBVHBuilder.cpp, BasicScene.cpp:281-294).
 * Morton order + PLOC clustering (or the Karras linear BVH, PT_OPT_BUILD_ALGO) collapsed into the
 * same item buffer pt_upload_bvh produces, in milliseconds; the PLOC tree traces as fast as the host
 * SAH/SBVH builder's.  Rendered images equal those over an uploaded hierarchy except for rays that graze a
 * bounding plane (see PT_OPT_REBUILD: a handful of pixels per 2 M at 800 k triangles, none on the small
 * scenes of the test suite).  Triangle ids are the row numbers of `tris`.  PT_OPT_LEAF_MAX (default 2)
 * = triangles per leaf.  verts: float[n_verts][3], tris: int32[n_tris][3]; host arrays, copied. */
int pt_build_bvh(pt_ctx* ctx, const float* verts, size_t n_verts, const int32_t* tris, size_t n_tris);
int pt_last_build_ms(pt_ctx* ctx, float* ms_out);   /* device time of the build behind the tree on the context (pt_build_bvh or a
                                                       kept PT_OPT_REBUILD tree); PT_ERR_INVALID for an uploaded hierarchy */

/* Per-triangle materials — an EXTENSION.  The reference parses the .mtl into
 * `materials` but never reads it (utilfun.cpp:458-462) and shades every triangle with the ONE
 * material of kernelInfo (tracer.cu:131-135 = pt_params.tri_mat/tri_col/tri_emi/phong_expo).
 * After this call a triangle with ORIGINAL id i (the ids of the Compact index array) is shaded
 * with table[tri_material[i]] instead; everything else of the path loop is unchanged.
 * n_materials = 0 clears the table (back to the reference's behaviour).  n_tris must cover
 * every id of the uploaded BVH.  Arrays are copied. */
typedef struct pt_material {
    float col[3];        /* albedo (mask *= col)                 */
    float emi[3];        /* emitted radiance (accu += mask*emi)  */
    int32_t mat;         /* PT_MAT_*                             */
    float phong_expo;    /* METAL lobe exponent                  */
} pt_material;           /* 32 bytes */
int pt_upload_tri_materials(pt_ctx* ctx, const pt_material* table, size_t n_materials,
                            const int32_t* tri_material, size_t n_tris);

/* ---- the hot path -------------------------------------------------------------
 * render(accum, bvh, camera, spp) of BASELINE.json: fold `spp` consecutive samples
 * (frames params->frame .. frame+spp-1, running-mean N = sample_index .. +spp-1) into
 * accum_dev (float[height][width][3], the reference's vec3 accumBuffer) and, when
 * PT_FLAG_WRITE_RGBA is set, the display word into rgba_dev (uint32[height][width],
 * the reference's dev_drawRes).  Equals `spp` single-sample calls bit for bit. */
int pt_render(pt_ctx* ctx, float* accum_dev, uint32_t* rgba_dev,
              const pt_camera* cam, const pt_params* params, uint32_t spp);

/* Closest-hit query on an explicit ray batch (rows a5–a7 of SURVEY.md §8 in isolation,
 * = intersectBVHandTriangles, cudaUtils.h:256-460).  rays_dev: float[n][8] =
 * (ox,oy,oz,tmin=0, dx,dy,dz,unused); out t_dev float[n] (F32_MAX on miss),
 * tri_dev int32[n] (original triangle id, -1 on miss), normal_dev float[n][3]
 * (un-normalised cross(v0-v1, v0-v2) of the winner; may be NULL). */
int pt_trace_rays(pt_ctx* ctx, const float* rays_dev, size_t n_rays, int cull_backfaces,
                  float* t_dev, int32_t* tri_dev, float* normal_dev);

/* ---- measurement --------------------------------------------------------------- */
int pt_get_counters(pt_ctx* ctx, pt_counters* out);
/* Schedule statistics of the last instrumented launch of the persistent wide walk
 * (PT_OPT_COUNTERS=1), summed over waves; up to PT_WAVE_STATS values:
 * [0] node-step iterations  [1] lanes active in them  [2] record-step iterations  [3] lanes
 * [4] shading passes        [5] lanes                 [6] path-start passes       [7] lanes
 * [8] outer-loop iterations  [9] traversal-stack pushes that overflowed the LDS window into
 * private memory (PT_OPT_LDS_STACK).  A wave-iteration with all 64 lanes active is 100 % use.
 * PT_KERNEL_WAVEFRONT: [0]-[3] and [6]-[9] are the extend stage's ([6]/[7] = refill passes), its
 * shade stage runs one lane per live path ([4], [5] stay 0). */
#define PT_WAVE_STATS 10
int pt_get_wave_stats(pt_ctx* ctx, uint64_t* out, int n);
int pt_last_kernel_ms(pt_ctx* ctx, float* ms_out);   /* needs PT_OPT_TIMING=1 */
/* Device time of the last timed pt_render (PT_OPT_TIMING=1) by stage, from HIP events recorded on the
 * context's stream between the launches: out[PT_STAGE_x] = milliseconds spent in that kind of launch,
 * summed over the call (PT_KERNEL_WAVEFRONT runs `depth` extend and `depth` shade launches).  The
 * reference's only timer is uf::GpuTimer around the whole launch (utilfun.hpp:44-79). */
enum {
    PT_STAGE_NONE = 0,      /* (start marker)                                                   */
    PT_STAGE_FRAME = 1,     /* the frame kernel of PT_KERNEL_MEGA_BVH2 / PT_KERNEL_PERSISTENT      */
    PT_STAGE_GENERATE = 2,  /* wavefront: prepare + camera rays                                   */
    PT_STAGE_EXTEND = 3,    /* wavefront: closest-hit walks, all bounces                          */
    PT_STAGE_SHADE = 4,     /* wavefront: spheres + shading + compaction, all bounces             */
    PT_STAGE_FOLD = 5,      /* k_fold_samples: sample colours -> running mean + display word      */
    PT_STAGE_COUNT = 6
};
int pt_get_stage_ms(pt_ctx* ctx, float* out, int n);
/* PT_KERNEL_AUTO's pick for the configuration of the last pt_render: *kernel = PT_KERNEL_PERSISTENT or
 * PT_KERNEL_WAVEFRONT once decided (PT_KERNEL_AUTO while the two trials are still running), and the two trial times. */
int pt_auto_choice(pt_ctx* ctx, int* kernel, float* ms_persistent, float* ms_wavefront);
/* Surface-area cost of the 4-wide tree on the context (measurement; what a random ray is expected to fetch):
 * *node_visits = (area of the root + of every child box that leads to a wide node) / area of the root,
 * *tri_tests   = sum over leaves of (area of the leaf's box x its triangle records) / area of the root.
 * Areas are those of the quantised boxes the walk tests.  Synchronises the context's stream. */
int pt_tree_cost(pt_ctx* ctx, double* node_visits, double* tri_tests);
int pt_scene_info(pt_ctx* ctx, uint64_t* n_inner, uint64_t* n_tri_refs,
                  uint64_t* n_leaves, uint32_t* max_depth, uint64_t* device_bytes);

#ifdef __cplusplus
}
#endif
#endif /* PTMI_H */

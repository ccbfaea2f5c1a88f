// pt_kernels.h — the path loop as gfx950 kernels.
//
// Reference path:  trace (GpuPathTracer/tracer.cu:343-400) → getSample
// (:27-339) → intersectBVHandTriangles (GpuPathTracer/cudaUtils.h:256-460) +
// intersectAllSpeheres (:221-236) → BRDF block (tracer.cu:156-293) → accumulate + pack
// (:386-398).  Not a translation: wave64 tiles, traversal stack in LDS, re-laid-out
// scene, counter RNG keyed by pixel.
#pragma once
#include "pt_math.h"
#include "../../include/ptmi.h"

#define PT_BLOCK 256          // frame kernels: 4 waves; one wave = one 8x8 pixel tile
#define PT_BLOCK_RAYS 256     // ray-batch kernel
#define PT_TILE 8
#define PT_MAX_TOP 1024       // most nodes the LDS copy of the top of the tree may hold
#define PT_STACK_CAP 72       // deepest traversal stack (tree depth <= 64, SplitBVHBuilder MaxDepth)

struct KScene {
    // One buffer of 64-byte items (4 float4 each): inner nodes first, then triangle records
    //   node: [c0.lo.x c0.hi.x c0.lo.y c0.hi.y][c1 ...][c0.lo.z c0.hi.z c1.lo.z c1.hi.z][link0 link1 0 0]
    //   tri : [v0.xyz, id][e1.xyz, last][e2.xyz, 0][cross(v0-v1, v0-v2), 0]
    // links: >= 0 float4 index of an inner node, < 0 ~(float4 index of a leaf's first record)
    const float4* __restrict__ nodes;   // = items
    const float4* __restrict__ tris;    // = items (same base: leaf links index the same buffer)
    const pt_sphere_d* __restrict__ spheres;
    int n_spheres;
    int has_bvh;
    int n_top;     // nodes [top_base/4, top_base/4 + n_top) (breadth-first prefix) are mirrored in LDS
    int stack_n;   // LDS stack entries per lane
    int top_base;  // float4 index of the mirrored tree's root: 0 (binary) or wide_root
    int wide_root; // float4 index of the 4-wide quantised tree's root (pt_wide.h), 0 if absent
};

#define PT_KSPHERES 8   // spheres carried in the kernel-argument block (scalar loads); more -> global array

struct KParams {
    KScene sc;
    pt_sphere_d ksph[PT_KSPHERES];
    float* __restrict__ accum;
    uint32_t* __restrict__ rgba;
    unsigned long long* counters;      // 6 x u64 when instrumented
    pt_camera cam;
    int W, H;
    uint32_t depth;
    int cull;
    uint64_t frame, sample_index;
    uint32_t spp;
    int tri_mat;
    float tri_col[3], tri_emi[3], bk[3];
    float air_ior, glass_ior, phong;
    // per-triangle materials (pt_upload_tri_materials): NULL = the reference's one global material
    const int* tri_matid;          // [original triangle id] -> row of mat_table
    const float4* mat_table;       // 2 float4 per material: (col, emi.x) (emi.yz, mat bits, phong)
    uint32_t flags;
    // tile enumeration: tiles_x tiles per tile-row; this launch covers n_tiles tiles taken
    // from the tile-rows this partition owns (stripes of stripe_tr tile-rows, round-robin)
    int tiles_x, tile_rows, n_tiles;
    int part_index, part_count, stripe_tr;
    // persistent kernel: global work counter over the n_tiles*64 tile-ordered pixel slots,
    // and the number of waiting lanes that makes a wave leave the traversal loop
    unsigned int* queue;
    int batch;
    int refill;   // idle lanes that trigger a refill from the queue
    int sph_tab;              // float index into the dynamic LDS of the sphere table (persistent kernel), -1 = none
    float4* roles_state;      // role-split kernel: cold path state, [block][slot][PT_COLD_DW] floats
    int vote_node, vote_rec;  // postponed-leaf walk: node step when n_node*vote_node >= n_rec*vote_rec
    int chunk;    // tile-ordered pixel slots per queue fetch (<= PT_CHUNK)
    // spp > 1: samples are traced as independent work items into `samples` ([spp][H*W][3] floats)
    // and folded into the running mean afterwards, in order, by k_fold_samples.  The frame's
    // critical path is then ONE path, not spp paths, and a launch has spp x more parallel work.
    float* __restrict__ samples;   // nullptr: fold each sample straight into accum (spp == 1)
};

struct Hit {
    float t;   // PT_F32_MAX on miss
    int tri;   // original triangle id, -1 on miss
    int rec;   // float4 index of the winner's record: its 4th piece holds cross(v0-v1, v0-v2),
               // fetched once per segment by pt_hit_normal instead of at every improvement (and two
               // registers less to carry through the walk)
};

// the un-normalised geometric normal of a hit triangle (4th piece of its record)
__device__ __forceinline__ v3 pt_hit_normal(const KScene& sc, const Hit& h) {
    const float4 q3 = sc.nodes[h.rec + 3];
    return V3(q3.x, q3.y, q3.z);
}

// true in exactly one of the lanes that execute this call together
__device__ __forceinline__ bool pt_first_active_lane() {
    const unsigned long long m = __ballot(true);
    return (__ffsll((long long)m) - 1) == (int)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
}

struct TravCount {
    uint32_t inner, tris, leaves;
    // wave-level schedule statistics (instrumented launches of the wide walk only; identical in
    // every lane): iterations spent in node / record steps and the lanes active in them
    uint32_t it_node, act_node, it_rec, act_rec;
};

#include "pt_walks.h"
#include "pt_shade.h"

template <bool COUNT, int ALG, class STK>
__device__ __forceinline__ v3 pt_get_sample(const KParams& P, int px, int py, uint64_t pix, uint64_t frame,
                                            STK& stk, const float4* __restrict__ s_top, TravCount& tc,
                                            uint32_t& n_rays, uint32_t& n_hits) {
    PathState ps;
    path_begin(P, px, py, pix, frame, ps);
    const bool cull = P.cull != 0;
    v3 col = V3(0.f, 0.f, 0.f);
    if (P.depth == 0) return col;
    for (;;) {
        Hit h;
        h.t = PT_F32_MAX; h.tri = -1; h.rec = 0;
        if (P.sc.has_bvh) {
            if (ALG >= 2) {
                TravState ts;
                trav_begin(ts, ps.o, ps.d, stk, P.sc.wide_root);
                trav_run_wide<COUNT, false, false, ALG == 3, STK>(ts, P.sc, ps.o, ps.d, cull, stk, tc, 0, 0);
                h = ts.h;
            } else if (ALG == 1) {
                TravState ts;
                trav_begin(ts, ps.o, ps.d, stk);
                trav_run_unified<COUNT, false, STK>(ts, P.sc, ps.o, ps.d, cull, stk, tc, 0, 0);
                h = ts.h;
            } else {
                h = trav_bvh2<COUNT, true, STK>(P.sc, ps.o, ps.d, cull, stk, tc, s_top);
            }
        }
        if (COUNT) { n_rays++; n_hits += (h.tri != -1); }
        if (path_shade(P, ps, h, col)) break;
    }
    return col;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Maps the launch's linear tile number to the global tile coordinates this partition owns.
__device__ __forceinline__ bool pt_tile_coords(const KParams& P, int tile, int& tx, int& ty) {
    if (tile >= P.n_tiles) return false;
    const int lrow = tile / P.tiles_x;
    tx = tile - lrow * P.tiles_x;
    if (P.part_count > 1) {
        const int k = lrow / P.stripe_tr, within = lrow - k * P.stripe_tr;
        ty = (P.part_index + k * P.part_count) * P.stripe_tr + within;
    } else {
        ty = lrow;
    }
    return ty < P.tile_rows;
}

// trace<<<>>>, tracer.cu:343-400: one lane per pixel, one wave per 8x8 tile, `spp`
// consecutive samples folded in registers.
// OCC = waves per SIMD the register allocator must leave room for (4 / 6 / 8)
// ALG = 0 while-while walk (Aila-Laine), 1 unified-step walk, 2 wide (4-way quantised) walk,
//       3 wide walk over Woop records
template <bool COUNT, int OCC, int LSTK, int ALG>
__global__ void __launch_bounds__(PT_BLOCK, OCC) k_trace_mega_bvh2(const KParams P) {
    float4* s_top = s_dyn;
    lds_load_top<PT_BLOCK>(P.sc, s_top);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    int tile = blockIdx.x * (PT_BLOCK / 64) + (tid >> 6);
    uint32_t s_only = 0;
    if (P.samples) {  // one wave per (sample, tile)
        s_only = (uint32_t)(tile / P.n_tiles);
        tile -= (int)s_only * P.n_tiles;
        if (s_only >= P.spp) return;
    }
    int tx, ty;
    if (!pt_tile_coords(P, tile, tx, ty)) return;
    const int px = tx * PT_TILE + (lane & 7), py = ty * PT_TILE + (lane >> 3);
    if (px >= P.W || py >= P.H) return;  // tracer.cu:358
    const uint64_t pix = (uint64_t)py * (uint64_t)P.W + (uint64_t)px;
    TravOverflow<LSTK> stk_ovf;
    TravStack<LSTK, PT_BLOCK> stk(__builtin_amdgcn_readfirstlane(16 * P.sc.n_top + (tid & ~63)), stk_ovf);

    TravCount tc;
    tc.inner = tc.tris = tc.leaves = 0;
    uint32_t n_rays = 0, n_hits = 0;

    uint32_t n_done = P.spp;
    if (P.samples) {
        const v3 col = pt_get_sample<COUNT, ALG>(P, px, py, pix, P.frame + s_only, stk, s_top, tc, n_rays, n_hits);
        float* dst = P.samples + 3 * ((size_t)s_only * (size_t)P.W * (size_t)P.H + (size_t)pix);
        dst[0] = col.x; dst[1] = col.y; dst[2] = col.z;
        n_done = 1;
    } else {
        float* acc = P.accum + 3 * pix;
        float ax = 0.f, ay = 0.f, az = 0.f;
        if (P.sample_index != 1) { ax = acc[0]; ay = acc[1]; az = acc[2]; }
        for (uint32_t s = 0; s < P.spp; s++) {
            const v3 col = pt_get_sample<COUNT, ALG>(P, px, py, pix, P.frame + s, stk, s_top, tc, n_rays, n_hits);
            pt_accumulate(ax, ay, az, col, P.sample_index + s);
        }
        acc[0] = ax; acc[1] = ay; acc[2] = az;
        if ((P.flags & PT_FLAG_WRITE_RGBA) && P.rgba) P.rgba[pix] = pt_pack_rgba(ax, ay, az);
    }
    if (COUNT) {
        const uint32_t a = wave_sum_u32(n_rays), b = wave_sum_u32(tc.inner), c = wave_sum_u32(tc.tris);
        const uint32_t dd = wave_sum_u32(tc.leaves), e = wave_sum_u32(n_hits), f = wave_sum_u32(n_done);
        if (__ffsll((long long)__ballot(1)) - 1 == lane) {
            atomicAdd(&P.counters[0], (unsigned long long)a);
            atomicAdd(&P.counters[1], (unsigned long long)b);
            atomicAdd(&P.counters[2], (unsigned long long)c);
            atomicAdd(&P.counters[3], (unsigned long long)dd);
            atomicAdd(&P.counters[4], (unsigned long long)e);
            atomicAdd(&P.counters[5], (unsigned long long)f);
        }
    }
}

// Closest-hit on an explicit ray batch (pt_trace_rays): rows a5–a7 in isolation.
__global__ void __launch_bounds__(PT_BLOCK_RAYS) k_trace_rays_bvh2(const KScene sc, const float4* __restrict__ rays, size_t n,
                                                                   int cull, float* __restrict__ t_out,
                                                                   int* __restrict__ tri_out, float* __restrict__ n_out) {
    float4* s_top = s_dyn;
    lds_load_top<PT_BLOCK_RAYS>(sc, s_top);
    const size_t i = (size_t)blockIdx.x * PT_BLOCK_RAYS + threadIdx.x;
    if (i >= n) return;
    const float4 ro = rays[2 * i], rd = rays[2 * i + 1];
    TravCount tc;
    tc.inner = tc.tris = tc.leaves = 0;
    TravOverflow<PT_STACK_CAP> stk_ovf;
    TravStack<PT_STACK_CAP, PT_BLOCK_RAYS> stk(__builtin_amdgcn_readfirstlane(16 * sc.n_top + ((int)threadIdx.x & ~63)), stk_ovf);
    const Hit h = trav_bvh2<false, true>(sc, V3(ro.x, ro.y, ro.z), V3(rd.x, rd.y, rd.z), cull != 0, stk, tc, s_top);
    t_out[i] = h.t;
    tri_out[i] = h.tri;
    if (n_out) {
        const v3 hn = h.tri != -1 ? pt_hit_normal(sc, h) : V3(0.f, 0.f, 0.f);
        n_out[3 * i] = hn.x; n_out[3 * i + 1] = hn.y; n_out[3 * i + 2] = hn.z;
    }
}

// ---------------------------------------------------------------------------------------
// Persistent-waves variant (Aila-Laine "persistent threads", which the reference does NOT
// have: SURVEY.md F5).  grid = resident waves only; each wave pulls chunks of PT_CHUNK
// tile-ordered pixel slots from one global counter and keeps every lane busy:
//   A. refill  — idle lanes take the next slots of the wave's chunk; the lane→slot map is a
//                ballot + prefix-count (mbcnt) compaction of the idle mask
//   B. walk    — lanes with a ray in flight run the resumable closest-hit walk; the wave
//                leaves it as soon as `batch` lanes are waiting for service
//   C. shade   — lanes whose walk finished do spheres/shading/BRDF and either get their next
//                segment (back to B) or fold the sample into the accumulator and go idle
// A lane's ray no longer waits for the slowest ray of its 8x8 tile at every bounce.  Each
// pixel still sees exactly the arithmetic of k_trace_mega_bvh2 (RNG keyed by pixel, same
// walk), so the image is bit-identical; only the schedule differs.
#define PT_CHUNK 64
#define PT_SHARDS 8          // work-queue counters (one per XCD worth of blocks)
#define PT_SHARD_STRIDE 32   // uints between counters: one 128-byte line each
enum { PH_IDLE = 0, PH_TRAV = 1, PH_SHADE = 2 };

template <bool COUNT, int OCC, int LSTK, int ALG>
__global__ void __launch_bounds__(PT_BLOCK, OCC) k_trace_persist_bvh2(const KParams P) {
    float4* s_top = s_dyn;
    if (P.sph_tab >= 0 && threadIdx.x < 11 * PT_KSPHERES) {  // sphere attributes + centres for the shading code (see path_shade)
        PT_KARGS(K);
        const float v = ((const __attribute__((address_space(4))) float*)&K.ksph[0])[threadIdx.x];
        ((float*)s_dyn)[P.sph_tab + threadIdx.x] = v;
        if (threadIdx.x % 11 < 4) ((float*)s_dyn)[P.sph_tab + 88 + 4 * (threadIdx.x / 11) + threadIdx.x % 11] = v;
    }
    lds_load_top<PT_BLOCK>(P.sc, s_top);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    TravOverflow<LSTK> stk_ovf;
    TravStack<LSTK, PT_BLOCK> stk(__builtin_amdgcn_readfirstlane(16 * P.sc.n_top + (tid & ~63)), stk_ovf);
    const bool cull = P.cull != 0;
    const uint32_t slots_per_sample = (uint32_t)P.n_tiles * 64u;
    const uint32_t total = slots_per_sample * (P.samples ? P.spp : 1u);

    uint32_t chunk_next = 0, chunk_end = 0;  // wave-uniform
    bool queue_empty = false;                // wave-uniform
    int shard = (int)(blockIdx.x & (PT_SHARDS - 1));  // wave-uniform: the shard this wave draws from
    const uint32_t chunk = (uint32_t)P.chunk;  // slots per fetch: 64, or less when the launch is small
    const uint32_t shard_chunks = ((total + chunk - 1) / chunk + PT_SHARDS - 1) / PT_SHARDS;

    int phase = PH_IDLE;
    uint32_t pix = 0, s_idx = 0;
    PathState ps;
    TravState ts;
    ps.o = ps.d = ps.mask = ps.accu = V3(0.f, 0.f, 0.f);
    ps.depth = 0; ps.rng.s0 = ps.rng.s1 = ps.rng.n = 0;
    ts.idx = ts.idy = ts.idz = ts.oodx = ts.oody = ts.oodz = 0.f;
    ts.node = PT_SENTINEL; ts.leaf = 0; ts.sp = 0;
    ts.h.t = PT_F32_MAX; ts.h.tri = -1; ts.h.rec = 0;

    TravCount tc;
    tc.inner = tc.tris = tc.leaves = 0;
    tc.it_node = tc.act_node = tc.it_rec = tc.act_rec = 0;
    uint32_t n_rays = 0, n_hits = 0, n_paths = 0;
    uint32_t it_begin = 0, act_begin = 0, it_shade = 0, act_shade = 0, it_loop = 0;  // COUNT only

    for (;;) {
        if (COUNT) it_loop++;
        // ---- A. refill idle lanes (all 64 lanes are converged here)
        const unsigned long long idle = __ballot(phase == PH_IDLE);
        const int n_idle = __popcll(idle);
        // Refill in batches: starting a path (tile coordinates, accumulator read, RNG seed, camera
        // ray: ~200 instructions) for one or two lanes at a time costs the whole wave those
        // instructions at 2-3 % utilisation (measured: 0.37 ms of a 1.18 ms frame).
        const int n_busy = __popcll(__ballot(phase == PH_TRAV));
        if (!queue_empty && (n_idle >= P.refill || (n_idle > 0 && n_busy == 0))) {
            if (chunk_next == chunk_end) {
                // eight counters, one per group of blocks that share an XCD (blockIdx % 8 is the
                // group label of the dispatcher's round-robin; speed only, never correctness);
                // shard s owns chunks s, s+8, s+16, ... (interleaved: contiguous bands of the image
                // cost very different amounts); an empty shard is left for the next (work stealing).  A single counter serialises: 32 400 chunk fetches
                // on one L2 atomic unit take ~0.37 ms (~88 returning atomics per microsecond).
                for (int tries = 0; tries < PT_SHARDS; tries++) {
                    uint32_t k = 0;
                    if (lane == 0) k = atomicAdd(P.queue + shard * PT_SHARD_STRIDE, 1u);
                    k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
                    const uint32_t first = (k * PT_SHARDS + (uint32_t)shard) * chunk;  // interleaved chunks
                    if (k < shard_chunks && first < total) {
                        chunk_next = first;
                        chunk_end = min(first + chunk, total);
                        break;
                    }
                    shard = (shard + 1) & (PT_SHARDS - 1);
                }
                if (chunk_next == chunk_end) queue_empty = true;
            }
            const uint32_t avail = chunk_end - chunk_next;
            const uint32_t take = min((uint32_t)n_idle, avail);
            bool started = false;
            if (phase == PH_IDLE) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                if (rank < take) {
                    uint32_t q = chunk_next + rank;
                    uint32_t s_first = 0;
                    if (P.samples) {  // sample-major slots: [sample][tile][lane]
                        s_first = q / slots_per_sample;
                        q -= s_first * slots_per_sample;
                    }
                    int tx, ty;
                    if (pt_tile_coords(P, (int)(q >> 6), tx, ty)) {
                        const int px = tx * PT_TILE + (int)(q & 7u);
                        const int py = ty * PT_TILE + (int)((q >> 3) & 7u);
                        if (px < P.W && py < P.H) {  // tracer.cu:358
                            pix = (uint32_t)py * (uint32_t)P.W + (uint32_t)px;
                            s_idx = s_first;
                            // camera ray, then walk (or straight to shading); px/py live only here
                            path_begin(P, px, py, (uint64_t)pix, P.frame + s_idx, ps);
                            if (P.depth == 0) {
                                phase = PH_SHADE;
                                ts.h.t = PT_F32_MAX; ts.h.tri = -1;
                            } else if (P.sc.has_bvh) {
                                trav_begin(ts, ps.o, ps.d, stk, ALG >= 2 ? P.sc.wide_root : 0);
                                phase = PH_TRAV;
                            } else {
                                ts.h.t = PT_F32_MAX; ts.h.tri = -1; ts.h.rec = 0;
                                phase = PH_SHADE;
                            }
                            started = true;
                        }
                    }
                }
            }
            chunk_next += take;
            if (COUNT) {
                const int nb = __popcll(__ballot(started));
                if (nb) { it_begin++; act_begin += nb; }
            }
        }

        // ---- B. closest-hit walk for the lanes with a segment in flight
        {
            const int n_dead = queue_empty ? __popcll(__ballot(phase == PH_IDLE)) : 0;
            if (phase == PH_TRAV) {
                const bool fin = (ALG == 4)   ? trav_run_wide_pend<COUNT, true>(ts, P.sc, ps.o, ps.d, cull, stk, tc, n_dead, P.batch, P.vote_node, P.vote_rec)
                                 : (ALG >= 2) ? trav_run_wide<COUNT, true, false, ALG == 3>(ts, P.sc, ps.o, ps.d, cull, stk, tc, n_dead, P.batch)
                                 : (ALG == 1) ? trav_run_unified<COUNT, true>(ts, P.sc, ps.o, ps.d, cull, stk, tc, n_dead, P.batch)
                                              : trav_run<COUNT, true, true>(ts, P.sc, ps.o, ps.d, cull, stk, tc, n_dead, P.batch, s_top);
                if (fin) phase = PH_SHADE;
            }
        }

        // ---- C. shade finished segments
        if (COUNT) {
            const int nb = __popcll(__ballot(phase == PH_SHADE));
            if (nb) { it_shade++; act_shade += nb; }
        }
        if (phase == PH_SHADE) {
            v3 col = V3(0.f, 0.f, 0.f);
            bool done;
            if (P.depth == 0) {
                done = true;
            } else {
                if (COUNT) { n_rays++; n_hits += (ts.h.tri != -1); }
                done = path_shade(P, ps, ts.h, col, P.sph_tab);
            }
            if (!done) {
                if (P.sc.has_bvh) {
                    trav_begin(ts, ps.o, ps.d, stk, ALG >= 2 ? P.sc.wide_root : 0);
                    phase = PH_TRAV;
                }  // else: stays PH_SHADE with the (miss) hit record, shaded again next round
            } else if (P.samples) {
                float* dst = P.samples + 3 * ((size_t)s_idx * (size_t)P.W * (size_t)P.H + (size_t)pix);
                dst[0] = col.x; dst[1] = col.y; dst[2] = col.z;
                if (COUNT) n_paths++;
                phase = PH_IDLE;
            } else {
                // one sample per call (spp > 1 always comes with the sample buffer): fold it straight
                // into the accumulator; nothing of the pixel's running mean is carried through the walk
                float* acc = P.accum + 3 * (size_t)pix;
                float ax = 0.f, ay = 0.f, az = 0.f;
                if (P.sample_index != 1) { ax = acc[0]; ay = acc[1]; az = acc[2]; }
                pt_accumulate(ax, ay, az, col, P.sample_index);
                if (COUNT) n_paths++;
                acc[0] = ax; acc[1] = ay; acc[2] = az;
                if ((P.flags & PT_FLAG_WRITE_RGBA) && P.rgba) P.rgba[pix] = pt_pack_rgba(ax, ay, az);
                phase = PH_IDLE;
            }
        }

        if (queue_empty && !__ballot(phase != PH_IDLE)) break;
    }

    if (COUNT) {
        const uint32_t a = wave_sum_u32(n_rays), b = wave_sum_u32(tc.inner), c = wave_sum_u32(tc.tris);
        const uint32_t dd = wave_sum_u32(tc.leaves), e = wave_sum_u32(n_hits), f = wave_sum_u32(n_paths);
        // the walk books its iterations in whichever lane is first among those inside it: sum the lanes
        const uint32_t w_it_node = wave_sum_u32(tc.it_node), w_act_node = wave_sum_u32(tc.act_node);
        const uint32_t w_it_rec = wave_sum_u32(tc.it_rec), w_act_rec = wave_sum_u32(tc.act_rec);
        if (lane == 0) {
            atomicAdd(&P.counters[0], (unsigned long long)a);
            atomicAdd(&P.counters[1], (unsigned long long)b);
            atomicAdd(&P.counters[2], (unsigned long long)c);
            atomicAdd(&P.counters[3], (unsigned long long)dd);
            atomicAdd(&P.counters[4], (unsigned long long)e);
            atomicAdd(&P.counters[5], (unsigned long long)f);
            // schedule statistics, one contribution per wave (pt_get_wave_stats)
            atomicAdd(&P.counters[6], (unsigned long long)w_it_node);
            atomicAdd(&P.counters[7], (unsigned long long)w_act_node);
            atomicAdd(&P.counters[8], (unsigned long long)w_it_rec);
            atomicAdd(&P.counters[9], (unsigned long long)w_act_rec);
            atomicAdd(&P.counters[10], (unsigned long long)it_shade);
            atomicAdd(&P.counters[11], (unsigned long long)act_shade);
            atomicAdd(&P.counters[12], (unsigned long long)it_begin);
            atomicAdd(&P.counters[13], (unsigned long long)act_begin);
            atomicAdd(&P.counters[14], (unsigned long long)it_loop);
        }
    }
}


#include "pt_roles.h"

// ---------------------------------------------------------------------------------------
// Folds the spp sample colours of every owned pixel into the running mean, in sample order,
// with the reference's per-frame clamp (tracer.cu:386-391) and packs the display word
// (:394-398): exactly what spp consecutive single-sample launches do to the accumulator.
__global__ void __launch_bounds__(256) k_fold_samples(const KParams P) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    int tx, ty;
    if (!pt_tile_coords(P, tile, tx, ty)) return;
    const int px = tx * PT_TILE + (lane & 7), py = ty * PT_TILE + (lane >> 3);
    if (px >= P.W || py >= P.H) return;
    const size_t pix = (size_t)py * (size_t)P.W + (size_t)px, plane = (size_t)P.W * (size_t)P.H;
    float* acc = P.accum + 3 * pix;
    float ax = 0.f, ay = 0.f, az = 0.f;
    if (P.sample_index != 1) { ax = acc[0]; ay = acc[1]; az = acc[2]; }
    for (uint32_t s = 0; s < P.spp; s++) {
        const float* c = P.samples + 3 * (s * plane + pix);
        pt_accumulate(ax, ay, az, V3(c[0], c[1], c[2]), P.sample_index + s);
    }
    acc[0] = ax; acc[1] = ay; acc[2] = az;
    if ((P.flags & PT_FLAG_WRITE_RGBA) && P.rgba) P.rgba[pix] = pt_pack_rgba(ax, ay, az);
}

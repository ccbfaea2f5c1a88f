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
#define PT_SHARDS 8           // work-queue counters (one per XCD worth of blocks)
#define PT_SHARD_STRIDE 32    // uints between counters: one 128-byte line each
#define PT_REGION 256         // stage-split pipeline: path-record slots per region (= one block)

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
    int wide_root; // float4 index of the 4-wide quantised tree's root (= its node 0), 0 if absent
};

#define PT_KSPHERES 8   // spheres carried in the kernel-argument block (scalar loads); more -> global array

// Stage-split (wavefront) pipeline, pt_k_wave.hip: one generation of path records in HBM.  A record lives
// in slot i of its REGION (256 consecutive slots = one block of the generate / shade stages); a region's live
// records are packed at its front and counted in cnt[region], so compaction never leaves the block
// (ballot + prefix count, no global atomics) and the order of the records is deterministic.
//   ray0[i] = (o.x, o.y, o.z, d.x)   ray1[i] = (d.y, d.z, bits(pixel), bits(sample << 12 | rng draws))
//   mask    = three planes [cap] (x, y, z), not stored for the first bounce (1, 1, 1)
//   hit[i]  = (t, bits(float4 index of the winning record)); t = F32_MAX: no triangle
struct KWave {
    const float4* __restrict__ ray0_in;
    const float4* __restrict__ ray1_in;
    const float* __restrict__ mask_in;
    float4* __restrict__ ray0_out;
    float4* __restrict__ ray1_out;
    float* __restrict__ mask_out;
    float2* __restrict__ hit;
    const int* __restrict__ cnt_in;
    int* __restrict__ cnt_out;
    unsigned long long* hashes;   // uf::hash(frame + s), s < spp (written by k_wf_prepare)
    unsigned int* queue;          // region queue of THIS bounce's extend launch (PT_SHARDS counters)
    unsigned int* queues_all;     // k_wf_prepare: every bounce's counters, zeroed
    uint32_t queues_words;
    uint32_t cap;                 // slots per plane (regions * 256)
    int n_regions;
    uint32_t n_slots;             // (sample, pixel) slots of the call: work tiles * 64 (bounce 0 works on slots, not records)
    // PT_FLAG_NEE: the shadow-ray records a shade launch emits (region-compacted like the survivors), traced by another
    // extend launch and resolved by k_wf_resolve: s_ray0/s_ray1 as ray0/ray1, s_con = (contribution rgb, t_max), s_hit
    float4* __restrict__ s_ray0;
    float4* __restrict__ s_ray1;
    float4* __restrict__ s_con;
    float2* __restrict__ s_hit;
    int* __restrict__ s_cnt;
    int nee;                      // 1: ray1.z carries pixel | nee_mask << 24
    uint32_t bounce;
};

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
    // PT_FLAG_NEE: the triangles whose material row emits, ascending original id; 3 float4 each:
    // (v0, emi.r) (e1 = v1 - v0, emi.g) (e2 = v2 - v0, emi.b) — copied from the triangle's record
    const float4* tri_lights;
    int n_tri_lights;
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
    int vote_node, vote_rec;  // postponed-leaf walk: node step when n_node*vote_node >= n_rec*vote_rec
    int chunk;    // tile-ordered pixel slots per queue fetch (<= PT_CHUNK)
    // spp > 1: samples are traced as independent work items into `samples` ([spp][H*W][3] floats)
    // and folded into the running mean afterwards, in order, by k_fold_samples.  The frame's
    // critical path is then ONE path, not spp paths, and a launch has spp x more parallel work.
    float* __restrict__ samples;   // nullptr: fold each sample straight into accum (spp == 1)
    // where (sample s, pixel p) lives: samples + 3 * (s * smp_ss + p * smp_ps) — [spp][H*W][3] (W*H, 1) for the kernels whose
    // waves hold ONE sample of 64 pixels, [H*W][spp][3] (1, spp) for the stage-split pipeline's sample groups
    unsigned long long smp_ss;
    uint32_t smp_ps;
    uint32_t sgroup_log2;          // 64 consecutive (sample, pixel) slots = 2^k samples x (64 >> k) pixels of a tile (pt_slot_pixel)
    KWave wf;
};

// the three colour floats of (sample s, pixel pix) in the call's sample buffer
__device__ __forceinline__ float* pt_sample_ptr(const KParams& P, uint32_t s, size_t pix) {
    return P.samples + 3 * ((size_t)s * (size_t)P.smp_ss + pix * (size_t)P.smp_ps);
}

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
                trav_run_wide<COUNT, false, ALG == 3>(ts, P.sc, ps.o, ps.d, cull, stk, tc, 0, 0);
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
        NeeReq req;
        req.want = false;
        const bool done = path_shade(P, ps, h, col, -1, (P.flags & PT_FLAG_NEE) ? &req : nullptr);
        // the loop's exit condition crosses the shadow walk below in a VGPR: kept as a lane mask in SGPRs, hipcc (ROCm 7.2)
        // lost it across the walk's wave-uniform loop and a finished path ran one more bounce
        int done_v = done ? 1 : 0;
        asm volatile("" : "+v"(done_v));
        if (req.want) {   // PT_FLAG_NEE: the shadow ray of this DIFF hit, against the triangles
            Hit h2;
            h2.t = PT_F32_MAX; h2.tri = -1; h2.rec = 0;
            if (P.sc.has_bvh) {
                TravState ts;
                if (ALG >= 2) {
                    trav_begin(ts, req.o, req.d, stk, P.sc.wide_root);
                    trav_run_wide<COUNT, false, ALG == 3>(ts, P.sc, req.o, req.d, cull, stk, tc, 0, 0);
                } else if (ALG == 1) {
                    trav_begin(ts, req.o, req.d, stk);
                    trav_run_unified<COUNT, false, STK>(ts, P.sc, req.o, req.d, cull, stk, tc, 0, 0);
                } else {
                    trav_begin(ts, req.o, req.d, stk);
                    trav_run<COUNT, false, true, STK>(ts, P.sc, req.o, req.d, cull, stk, tc, 0, 0, s_top);
                }
                h2 = ts.h;
            }
            if (COUNT) n_rays++;
            if (!(h2.t < req.t_max)) {
                ps.accu = vadd(ps.accu, req.contrib);
                if (done) col = vadd(col, req.contrib);   // the path ended with this bounce: col was the gathered light before it
            }
        }
        if (done_v) break;
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

// The call's (sample, pixel) slots, 64 to a work tile, n_tiles * spp work tiles (stage-split pipeline and persistent kernel):
//  * sgroup_log2 = 0: work tile = (sample, tile), slot & 63 = pixel of the 8x8 tile in row order;
//  * sgroup_log2 = k: G = 2^k samples x 64/G pixels of a tile — the G samples of a pixel are the SAME ray but for the sub-pixel
//    jitter, so lanes that start them together walk the same nodes and records in step.  Work tile -> (sample group, tile, j-th
//    share of the tile's pixels); lane -> (pixel number j * 64/G + lane / G in Morton order, sample lane % G).
// Which lane traces which (pixel, sample) changes no result: every path is keyed by (sample's frame hash, pixel).
// False for slots past the call's last and for the pixels of a partial tile that lie outside the image (tracer.cu:358).
__device__ __forceinline__ bool pt_slot_pixel(const KParams& P, uint32_t slot, uint32_t& s_idx, int& px, int& py) {
    const uint32_t lg = P.sgroup_log2;
    int wt = (int)(slot >> 6);
    uint32_t k = slot & 63u, s_in = 0u;
    if (lg) {
        const uint32_t g1 = (1u << lg) - 1u;
        s_in = k & g1;
        k = ((uint32_t)wt & g1) * (64u >> lg) + (k >> lg);
        wt >>= lg;
    }
    const uint32_t grp = (uint32_t)(wt / P.n_tiles);
    wt -= (int)grp * P.n_tiles;
    s_idx = (grp << lg) + s_in;
    int tx = 0, ty = 0;
    if (s_idx >= P.spp || !pt_tile_coords(P, wt, tx, ty)) return false;
    if (lg) {   // Morton order inside the tile: consecutive pixel numbers form 2x2, 4x2, 4x4 ... blocks
        px = tx * PT_TILE + (int)((k & 1u) | ((k >> 1) & 2u) | ((k >> 2) & 4u));
        py = ty * PT_TILE + (int)(((k >> 1) & 1u) | ((k >> 2) & 2u) | ((k >> 3) & 4u));
    } else {
        px = tx * PT_TILE + (int)(k & 7u);
        py = ty * PT_TILE + (int)((k >> 3) & 7u);
    }
    return px < P.W && py < P.H;
}

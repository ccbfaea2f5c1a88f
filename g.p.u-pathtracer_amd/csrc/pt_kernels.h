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

// ---------------------------------------------------------------------------------------
// Binary-tree closest hit, same visiting order and arithmetic as cudaUtils.h:256-460 /
// the CPU restatement, so results are bit-identical to it.
//   - stack lives in LDS, laid out [entry][thread] → conflict-free for any mix of depths
//   - slab tests: 12 v_fma + v_min3/v_max3 (the reference's PTX vmin/vmax trick is only
//     valid for non-negative floats)
//   - postponed-leaf exit on a 64-lane ballot (cudaUtils.h:383-394 is a 32-lane vote)
// The walk is resumable: all of its state is in TravState, and run<DYN=true> returns early
// when enough other lanes of the wave are waiting to be shaded / refilled (persistent
// kernel); the per-ray sequence of tests is the same either way.
//   - TOP: the first n_top nodes in breadth-first order (the levels every ray walks) are read
//     from an LDS mirror laid out as four float4 planes.  rocprof showed the CU's vector
//     memory pipe (TA/TD) ~90 % busy with 64-byte gathers and 70-80 % of node visits landing
//     in the top few hundred nodes; ds_read_b128 runs on the LDS pipe instead.
// Dynamic LDS of the kernels: [top-of-tree planes: 4 x n_top float4][stack: LSTK x BLOCK int].
// One extern array so the carve base stays 16-byte aligned (cdna guide G17).
extern __shared__ float4 s_dyn[];

struct TravState {
    float idx, idy, idz, oodx, oody, oodz;
    int node, leaf, sp;
    Hit h;
};

// Traversal stack: the first LSTK entries of every lane live in LDS ([entry][thread], so any
// mix of depths is conflict-free); deeper entries — rare: the walk pushes one entry per level
// that has both children hit — overflow into a private (scratch) array.  A small LSTK is what
// lets 6-8 waves per SIMD fit in the CU's 160 KiB of LDS (64 B/lane at LSTK = 16).
// lane id of the calling lane (v_mbcnt), opaque to the optimiser
__device__ __forceinline__ int pt_lane_fresh() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

template <int LSTK>
struct TravOverflow {
    int e[LSTK < PT_STACK_CAP ? PT_STACK_CAP - LSTK : 1];
};

template <int LSTK, int BLOCK>
struct TravStack {
    int base;  // WAVE-UNIFORM int index of lane 0's entry 0 inside s_dyn (the __shared__ symbol is
               // named in the accessors so that the accesses stay ds_read/ds_write: a stored
               // pointer makes hipcc merge the LDS and overflow paths into flat_load/flat_store)
    // the overflow array is a SEPARATE private object: as a member it drags the whole struct,
    // `base` included, into scratch memory (a scratch reload in front of every push)
    int (&ovf)[LSTK < PT_STACK_CAP ? PT_STACK_CAP - LSTK : 1];
    int lane_base;  // base + lane id.  Re-deriving the lane id at every access (v_mbcnt x2 + add) was the
                    // cheaper choice while ~120 SGPR spills ate the VGPR budget; with the kernel arguments
                    // read at use, one VGPR here saves ~12 VALU per node step (-1.4 % / -3.3 % at 8 / 6 waves)
    __device__ __forceinline__ TravStack(int b, TravOverflow<LSTK>& o) : base(b), ovf(o.e), lane_base(b + pt_lane_fresh()) {}
    __device__ __forceinline__ void put(int sp, int v) {
        if (LSTK >= PT_STACK_CAP || sp < LSTK) {
            ((int*)s_dyn)[lane_base + sp * BLOCK] = v;
        } else {
            asm volatile("" : "+v"(v));
            ovf[sp - LSTK] = v;
        }
    }
    __device__ __forceinline__ int get(int sp) const {
        int v;
        if (LSTK >= PT_STACK_CAP || sp < LSTK) {
            v = ((const int*)s_dyn)[lane_base + sp * BLOCK];
        } else {
            v = ovf[sp - LSTK];
            asm volatile("" : "+v"(v));
        }
        return v;
    }
};

template <class STK>
__device__ __forceinline__ void trav_begin(TravState& s, v3 o, v3 d, STK& stk, int root = 0) {
    const float ooeps = 8.271806125530277e-25f;  // exp2f(-80), cudaUtils.h:283
    s.idx = 1.0f / (fabsf(d.x) > ooeps ? d.x : copysignf(ooeps, d.x));
    s.idy = 1.0f / (fabsf(d.y) > ooeps ? d.y : copysignf(ooeps, d.y));
    s.idz = 1.0f / (fabsf(d.z) > ooeps ? d.z : copysignf(ooeps, d.z));
    s.oodx = o.x * s.idx; s.oody = o.y * s.idy; s.oodz = o.z * s.idz;
    s.sp = 0;
    stk.put(0, PT_SENTINEL);
    s.leaf = 0; s.node = root;
    s.h.t = PT_F32_MAX; s.h.tri = -1; s.h.rec = 0;
}

// returns true when the walk is complete
template <bool COUNT, bool DYN, bool TOP, class STK>
__device__ __forceinline__ bool trav_run(TravState& s, const KScene& sc, v3 o, v3 d, bool cull,
                                         STK& stk, TravCount& tc, int n_dead, int batch,
                                         const float4* __restrict__ s_top) {
    int node = s.node, leaf = s.leaf, sp = s.sp;
    Hit h = s.h;
    const float idx = s.idx, idy = s.idy, idz = s.idz, oodx = s.oodx, oody = s.oody, oodz = s.oodz;
    while (node != PT_SENTINEL) {
        while ((unsigned)node < (unsigned)PT_SENTINEL) {  // node >= 0 && node != sentinel
            float4 n0, n1, nz, nl;
            if (TOP && node < sc.n_top * 4) {
                // read through the __shared__ symbol itself and keep this a real branch: given
                // a pointer parameter, hipcc if-converts the two paths into ONE generic-pointer
                // select and emits eleven scalarised flat_load_dword per node
                const int i = node >> 2;
                n0 = s_dyn[i];
                n1 = s_dyn[sc.n_top + i];
                nz = s_dyn[2 * sc.n_top + i];
                nl = s_dyn[3 * sc.n_top + i];
                asm volatile("" : "+v"(n0.x), "+v"(nl.x));
            } else {
                n0 = sc.nodes[node + 0];
                n1 = sc.nodes[node + 1];
                nz = sc.nodes[node + 2];
                nl = sc.nodes[node + 3];
            }
            int cx = __float_as_int(nl.x), cy = __float_as_int(nl.y);
            // keep the link load with the three box loads: left alone, hipcc sinks it into the
            // "hit" branch below, which makes every node visit two dependent round trips
            asm volatile("" : "+v"(cx), "+v"(cy));
            if (COUNT) tc.inner++;
            const float c0lox = fmaf(n0.x, idx, -oodx), c0hix = fmaf(n0.y, idx, -oodx);
            const float c0loy = fmaf(n0.z, idy, -oody), c0hiy = fmaf(n0.w, idy, -oody);
            const float c1lox = fmaf(n1.x, idx, -oodx), c1hix = fmaf(n1.y, idx, -oodx);
            const float c1loy = fmaf(n1.z, idy, -oody), c1hiy = fmaf(n1.w, idy, -oody);
            const float c0loz = fmaf(nz.x, idz, -oodz), c0hiz = fmaf(nz.y, idz, -oodz);
            const float c1loz = fmaf(nz.z, idz, -oodz), c1hiz = fmaf(nz.w, idz, -oodz);
            const float c0min = fmaxf(fmaxf(fmaxf(fminf(c0lox, c0hix), fminf(c0loy, c0hiy)), fminf(c0loz, c0hiz)), 0.0f);
            const float c0max = fminf(fminf(fminf(fmaxf(c0lox, c0hix), fmaxf(c0loy, c0hiy)), fmaxf(c0loz, c0hiz)), h.t);
            const float c1min = fmaxf(fmaxf(fmaxf(fminf(c1lox, c1hix), fminf(c1loy, c1hiy)), fminf(c1loz, c1hiz)), 0.0f);
            const float c1max = fminf(fminf(fminf(fmaxf(c1lox, c1hix), fmaxf(c1loy, c1hiy)), fmaxf(c1loz, c1hiz)), h.t);
            const bool t0 = (c0min <= c0max) && (c0min >= 0.0f) && (c0min <= PT_F32_MAX);
            const bool t1 = (c1min <= c1max) && (c1min >= 0.0f) && (c1min <= PT_F32_MAX);
            if (!t0 && !t1) {
                node = stk.get(sp);
                sp--;
            } else {
                node = t0 ? cx : cy;
                if (t0 && t1) {
                    if (c1min < c0min) { int tmp = node; node = cy; cy = tmp; }
                    sp++;
                    stk.put(sp, cy);
                }
            }
            if (node < 0 && leaf >= 0) {  // first leaf: postpone, keep descending
                leaf = node;
                node = stk.get(sp);
                sp--;
            }
            if (!__ballot(leaf >= 0)) break;  // every active lane holds a leaf
        }
        while (leaf < 0) {
            if (COUNT) tc.leaves++;
            for (int a = ~leaf;; a += 4) {
                const float4 r0 = sc.tris[a + 0];
                const float4 r1 = sc.tris[a + 1];
                const float4 r2 = sc.tris[a + 2];
                if (COUNT) tc.tris++;
                const v3 v0 = V3(r0.x, r0.y, r0.z), e1 = V3(r1.x, r1.y, r1.z), e2 = V3(r2.x, r2.y, r2.z);
                const float t = pt_mt_intersect(v0, e1, e2, o, d, cull);
                const int id = __float_as_int(r0.w);
                if (t > 0.0f && (t < h.t || (t == h.t && h.tri != -1 && id < h.tri))) {
                    h.t = t;
                    h.tri = id;
                    h.rec = a;
                }
                if (__float_as_int(r1.w) != 0) break;  // last record of the leaf
            }
            leaf = node;
            if (node < 0) {
                node = stk.get(sp);
                sp--;
            }
        }
        if (DYN) {  // enough lanes are waiting for service: hand the wave back
            const int active = __popcll(__ballot(1));
            if (64 - active - n_dead >= batch) break;
        }
    }
    s.node = node; s.leaf = leaf; s.sp = sp; s.h = h;
    return node == PT_SENTINEL;
}

// ---------------------------------------------------------------------------------------
// Unified-step walk: every iteration EVERY live lane advances by one 64-byte item — an inner
// node (two slab tests) or one triangle record (Moller-Trumbore) — fetched by the same four
// dwordx4 loads.  Why: rocprof shows the CU's vector-memory return path (TD) ~90 % busy at
// ~16-20 cycles per dwordx4 WAVE instruction whatever the number of active lanes, and the
// while-while walk above issues those instructions at ~26 % lane utilisation (lanes holding a
// leaf idle through the node phase and vice versa).  Here one set of four loads serves all 64
// lanes.  `cur` is the lane's item: >= 0 node, < 0 ~record, sentinel = done.  The set of
// boxes/triangles a ray tests can differ slightly from the while-while order (a leaf is
// tested as soon as it is popped, so later nodes see the shorter ray), the closest hit
// (t, id, normal) cannot: ties go to the smaller id, so the result is order-independent.
template <bool COUNT, bool DYN, class STK>
__device__ __forceinline__ bool trav_run_unified(TravState& s, const KScene& sc, v3 o, v3 d, bool cull, STK& stk,
                                                 TravCount& tc, int n_dead, int batch) {
    int cur = s.node, sp = s.sp;
    Hit h = s.h;
    const float idx = s.idx, idy = s.idy, idz = s.idz, oodx = s.oodx, oody = s.oody, oodz = s.oodz;
    while (cur != PT_SENTINEL) {
        const int a = cur >= 0 ? cur : ~cur;
        const float4 q0 = sc.nodes[a + 0];
        const float4 q1 = sc.nodes[a + 1];
        const float4 q2 = sc.nodes[a + 2];
        // 4th piece only for node lanes (links); a record's 4th piece (normal) is read on a hit.
        // The CU's address/tag pipe costs ~1 cycle per LANE-level 16-byte load.
        int cx = 0, cy = 0;
        if (cur >= 0) {
            const float4 q3 = sc.nodes[a + 3];
            cx = __float_as_int(q3.x);
            cy = __float_as_int(q3.y);
        }
        asm volatile("" : "+v"(cx), "+v"(cy));
        if (cur >= 0) {
            if (COUNT) tc.inner++;
            const float c0lox = fmaf(q0.x, idx, -oodx), c0hix = fmaf(q0.y, idx, -oodx);
            const float c0loy = fmaf(q0.z, idy, -oody), c0hiy = fmaf(q0.w, idy, -oody);
            const float c1lox = fmaf(q1.x, idx, -oodx), c1hix = fmaf(q1.y, idx, -oodx);
            const float c1loy = fmaf(q1.z, idy, -oody), c1hiy = fmaf(q1.w, idy, -oody);
            const float c0loz = fmaf(q2.x, idz, -oodz), c0hiz = fmaf(q2.y, idz, -oodz);
            const float c1loz = fmaf(q2.z, idz, -oodz), c1hiz = fmaf(q2.w, idz, -oodz);
            const float c0min = fmaxf(fmaxf(fmaxf(fminf(c0lox, c0hix), fminf(c0loy, c0hiy)), fminf(c0loz, c0hiz)), 0.0f);
            const float c0max = fminf(fminf(fminf(fmaxf(c0lox, c0hix), fmaxf(c0loy, c0hiy)), fmaxf(c0loz, c0hiz)), h.t);
            const float c1min = fmaxf(fmaxf(fmaxf(fminf(c1lox, c1hix), fminf(c1loy, c1hiy)), fminf(c1loz, c1hiz)), 0.0f);
            const float c1max = fminf(fminf(fminf(fmaxf(c1lox, c1hix), fmaxf(c1loy, c1hiy)), fmaxf(c1loz, c1hiz)), h.t);
            const bool t0 = (c0min <= c0max) && (c0min >= 0.0f) && (c0min <= PT_F32_MAX);
            const bool t1 = (c1min <= c1max) && (c1min >= 0.0f) && (c1min <= PT_F32_MAX);
            if (!t0 && !t1) {
                cur = stk.get(sp);
                sp--;
            } else {
                cur = t0 ? cx : cy;
                if (t0 && t1) {
                    if (c1min < c0min) { int tmp = cur; cur = cy; cy = tmp; }
                    sp++;
                    stk.put(sp, cy);
                }
            }
            if (COUNT && cur < 0) tc.leaves++;
        } else {
            if (COUNT) tc.tris++;
            const v3 v0 = V3(q0.x, q0.y, q0.z), e1 = V3(q1.x, q1.y, q1.z), e2 = V3(q2.x, q2.y, q2.z);
            const float t = pt_mt_intersect(v0, e1, e2, o, d, cull);
            const int id = __float_as_int(q0.w);
            if (t > 0.0f && (t < h.t || (t == h.t && h.tri != -1 && id < h.tri))) {
                h.t = t;
                h.tri = id;
                h.rec = a;
            }
            if (__float_as_int(q1.w) != 0) {  // last record of the leaf
                cur = stk.get(sp);
                sp--;
                if (COUNT && cur < 0 && cur != PT_SENTINEL) tc.leaves++;
            } else {
                cur -= 4;  // ~(a + 4)
            }
        }
        if (DYN) {  // enough lanes are waiting for service: hand the wave back
            const int active = __popcll(__ballot(cur != PT_SENTINEL));
            if (64 - active - n_dead >= batch) break;
        }
    }
    s.node = cur; s.sp = sp; s.h = h;
    return cur == PT_SENTINEL;
}

// ---------------------------------------------------------------------------------------
// Wide walk: unified-step over the 4-wide quantised tree of pt_wide.h.  A node item tests four
// child boxes (24 v_cvt_f32_ubyte + 24 v_fma + min/max), sorts the hit children by entry
// distance with a 5-exchange network on (distance bits | child number) keys, continues with the
// nearest and pushes the rest far-to-near.  Record items are the exact Moller-Trumbore test of
// the other walks, so a reported hit is bit-identical to theirs; only the set of candidates the
// (outward-rounded) boxes let through differs.  3 pieces for a record, 4 for a node.
// WOOP: records hold Woop's affine rows (PT_OPT_TRI_TEST 1) instead of v0/e1/e2 — see
// pt_woop_intersect in pt_math.h; tolerance-class parity (the triangle arithmetic differs).
template <bool COUNT, bool DYN, bool TOP, bool WOOP, class STK>
__device__ __forceinline__ bool trav_run_wide(TravState& s, const KScene& sc, v3 o, v3 d, bool cull, STK& stk,
                                              TravCount& tc, int n_dead, int batch) {
    int cur = s.node, sp = s.sp;
    Hit h = s.h;
    const float idx = s.idx, idy = s.idy, idz = s.idz, oodx = s.oodx, oody = s.oody, oodz = s.oodz;
    for (;;) {
        // phase vote: the wave runs ONE kind of step per iteration, the kind most live lanes are
        // waiting for; the others sit this iteration out.  Node and record lanes no longer both
        // pay for each other's code every iteration (the limiter is VALU issue).
        // The loop is wave-uniform: every lane that entered stays until the common exit.
        const bool live = cur != PT_SENTINEL;
        const bool is_node = live && cur >= 0;
        const int n_live = __popcll(__ballot(live));
        if (n_live == 0) break;
        if (DYN && 64 - n_live - n_dead >= batch) break;  // enough lanes wait for service
        const int n_node = __popcll(__ballot(is_node));
        // a record step costs about half a node step: run whichever advances more lanes per instruction
        const bool node_phase = n_node >= 2 * (n_live - n_node);
        if (COUNT && pt_first_active_lane()) {  // one lane of those in the walk books the wave's iteration
            if (node_phase) { tc.it_node++; tc.act_node += n_node; }
            else { tc.it_rec++; tc.act_rec += n_live - n_node; }
        }
        if (!live || is_node != node_phase) continue;
        const int a = cur >= 0 ? cur : ~cur;
        float4 q0, q1, q2;
        float4 qw = make_float4(0.f, 0.f, 0.f, 0.f);  // WOOP: a record's 4th piece (normal | id<<1|last)
        int l2 = 0, l3 = 0;
        float sc_y = 0.f, sc_z = 0.f;
        const int ti = a - sc.top_base;
        if (TOP && cur >= 0 && (unsigned)ti < (unsigned)(sc.n_top * 4)) {
            const int i = ti >> 2;
            q0 = s_dyn[i];
            q1 = s_dyn[sc.n_top + i];
            q2 = s_dyn[2 * sc.n_top + i];
            const float4 q3 = s_dyn[3 * sc.n_top + i];
            l2 = __float_as_int(q3.x);
            l3 = __float_as_int(q3.y);
            sc_y = q3.z; sc_z = q3.w;
            asm volatile("" : "+v"(q0.x), "+v"(l2));
        } else {
            q0 = sc.nodes[a + 0];
            q1 = sc.nodes[a + 1];
            q2 = sc.nodes[a + 2];
            if (WOOP || cur >= 0) {
                const float4 q3 = sc.nodes[a + 3];
                l2 = __float_as_int(q3.x);
                l3 = __float_as_int(q3.y);
                sc_y = q3.z; sc_z = q3.w;
                qw = q3;
            }
            asm volatile("" : "+v"(l2), "+v"(l3));
        }
        if (cur >= 0) {
            if (COUNT) tc.inner++;
#ifdef PT_EXP_LOAD   // sensitivity experiment: one more 16-byte access to the node's line per node step
            { const float4 dummy = sc.nodes[a + 3]; asm volatile("" :: "v"(dummy.x), "v"(dummy.w)); }
#endif
#ifdef PT_EXP_VALU   // sensitivity experiment: 32 more dependent VALU instructions per node step
            { float z = q0.x;
#pragma unroll
              for (int e = 0; e < 32; e++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(z));
              asm volatile("" :: "v"(z)); }
#endif
            const float sx = q0.w * idx, sy = sc_y * idy, sz = sc_z * idz;  // per-axis grid step / direction
            const float bx = fmaf(q0.x, idx, -oodx), by = fmaf(q0.y, idy, -oody), bz = fmaf(q0.z, idz, -oodz);
            // entry/exit planes per axis follow the sign of the ray direction, so pick the packed
            // byte quadruples ONCE per node (6 v_cndmask) instead of min/max per child (24):
            // identical values to min(lo,hi)/max(lo,hi) of the reference's slab test
            const uint32_t qlx = __float_as_uint(q1.x), qly = __float_as_uint(q1.y), qlz = __float_as_uint(q1.z);
            const uint32_t qhx = __float_as_uint(q1.w), qhy = __float_as_uint(q2.x), qhz = __float_as_uint(q2.y);
            const bool px = idx >= 0.0f, py = idy >= 0.0f, pz = idz >= 0.0f;
            const uint32_t nx = px ? qlx : qhx, fx = px ? qhx : qlx;
            const uint32_t ny = py ? qly : qhy, fy = py ? qhy : qly;
            const uint32_t nz = pz ? qlz : qhz, fz = pz ? qhz : qlz;
            const int l0 = __float_as_int(q2.z), l1 = __float_as_int(q2.w);
            uint32_t key[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                // (entry, exit) of one axis in one v_pk_fma_f32
                const pt_f2 tx = pt_fma2(pt_mk2((float)((nx >> (8 * k)) & 0xffu), (float)((fx >> (8 * k)) & 0xffu)), pt_mk2(sx, sx), pt_mk2(bx, bx));
                const pt_f2 ty = pt_fma2(pt_mk2((float)((ny >> (8 * k)) & 0xffu), (float)((fy >> (8 * k)) & 0xffu)), pt_mk2(sy, sy), pt_mk2(by, by));
                const pt_f2 tz = pt_fma2(pt_mk2((float)((nz >> (8 * k)) & 0xffu), (float)((fz >> (8 * k)) & 0xffu)), pt_mk2(sz, sz), pt_mk2(bz, bz));
                const float tmin = fmaxf(fmaxf(fmaxf(tx.x, ty.x), tz.x), 0.0f);
                const float tmax = fminf(fminf(fminf(tx.y, ty.y), tz.y), h.t);
                const bool hit = tmin <= tmax;  // unused slots hold inverted boxes
                key[k] = hit ? ((__float_as_uint(tmin) & 0x7ffffffcu) | (uint32_t)k) : 0xffffffffu;
            }
            // sorting network for 4 keys: (0,1)(2,3)(0,2)(1,3)(1,2)
#define PT_CE(i, j) { const uint32_t lo_ = min(key[i], key[j]), hi_ = max(key[i], key[j]); key[i] = lo_; key[j] = hi_; }
            PT_CE(0, 1) PT_CE(2, 3) PT_CE(0, 2) PT_CE(1, 3) PT_CE(1, 2)
#undef PT_CE
#define PT_LINK(kk) (((kk) & 3u) == 0u ? l0 : (((kk) & 3u) == 1u ? l1 : (((kk) & 3u) == 2u ? l2 : l3)))
            if (key[3] != 0xffffffffu) { sp++; stk.put(sp, PT_LINK(key[3])); }
            if (key[2] != 0xffffffffu) { sp++; stk.put(sp, PT_LINK(key[2])); }
            if (key[1] != 0xffffffffu) {
                const int lk = PT_LINK(key[1]);
                sp++;
                stk.put(sp, lk);
            }
            if (key[0] != 0xffffffffu) {
                cur = PT_LINK(key[0]);
            } else {
                cur = stk.get(sp);
                sp--;
            }
#undef PT_LINK
            if (COUNT && cur < 0) tc.leaves++;
        } else {
            if (COUNT) tc.tris++;
            float t;
            int id;
            bool last;
            if (WOOP) {
                t = pt_woop_intersect(q0, q1, q2, V3(qw.x, qw.y, qw.z), o, d, cull);
                id = __float_as_int(qw.w) >> 1;
                last = (__float_as_int(qw.w) & 1) != 0;
            } else {
                const v3 v0 = V3(q0.x, q0.y, q0.z), e1 = V3(q1.x, q1.y, q1.z), e2 = V3(q2.x, q2.y, q2.z);
                t = pt_mt_intersect(v0, e1, e2, o, d, cull);
                id = __float_as_int(q0.w);
                last = __float_as_int(q1.w) != 0;
            }
            if (t > 0.0f && (t < h.t || (t == h.t && h.tri != -1 && id < h.tri))) {
                h.t = t;
                h.tri = id;
                h.rec = a;
            }
            // finish the leaf inside THIS iteration: its next record sits in the same or the next cache
            // line, so the extra fetch is short, and the wave saves a vote + a phase switch per record
            // (-1.4 % at 8 waves/SIMD, -3 % at 5-6)
            if (!WOOP) {
                int aa = a;
                while (!last) {
                    aa += 4;
                    const float4 r0 = sc.nodes[aa], r1 = sc.nodes[aa + 1], r2 = sc.nodes[aa + 2];
                    if (COUNT) tc.tris++;
                    const float t2 = pt_mt_intersect(V3(r0.x, r0.y, r0.z), V3(r1.x, r1.y, r1.z), V3(r2.x, r2.y, r2.z), o, d, cull);
                    const int id2 = __float_as_int(r0.w);
                    last = __float_as_int(r1.w) != 0;
                    if (t2 > 0.0f && (t2 < h.t || (t2 == h.t && h.tri != -1 && id2 < h.tri))) {
                        h.t = t2;
                        h.tri = id2;
                        h.rec = aa;
                    }
                }
            }
            if (last) {  // last record of the leaf
                cur = stk.get(sp);
                sp--;
                if (COUNT && cur < 0 && cur != PT_SENTINEL) tc.leaves++;
            } else {
                cur -= 4;  // ~(a + 4)
            }
        }
    }
    s.node = cur; s.sp = sp; s.h = h;
    return cur == PT_SENTINEL;
}

// Wide walk with ONE postponed leaf per lane (Aila-Laine's trick, restated for the phase vote):
// a lane that reaches a leaf parks it in `pend` and goes on with the next stack entry, so it can
// take part in node steps AND in record steps; it only waits when it holds a parked leaf and
// reaches a second one.  The closest hit is order independent (every pruning test uses a valid
// upper bound h.t, equal-t ties go to the smaller id), so the result is bit-identical to the
// other walks; parking a leaf only delays the tightening of h.t by a few node steps.
template <bool COUNT, bool DYN, class STK>
__device__ __forceinline__ bool trav_run_wide_pend(TravState& s, const KScene& sc, v3 o, v3 d, bool cull, STK& stk,
                                                   TravCount& tc, int n_dead, int batch, int vote_node, int vote_rec) {
    int cur = s.node, sp = s.sp, pend = s.leaf;  // pend: ~address of the next record of the parked leaf, 0 = none
    Hit h = s.h;
    const float idx = s.idx, idy = s.idy, idz = s.idz, oodx = s.oodx, oody = s.oody, oodz = s.oodz;
    for (;;) {
        const bool has_node = (unsigned)cur < (unsigned)PT_SENTINEL;
        const bool has_rec = pend != 0;
        const int n_live = __popcll(__ballot(has_node || has_rec));
        if (n_live == 0) break;
        if (DYN && 64 - n_live - n_dead >= batch) break;  // enough lanes wait for service
        const int n_node = __popcll(__ballot(has_node));
        const int n_rec = __popcll(__ballot(has_rec));
        const bool node_phase = n_node * vote_node >= n_rec * vote_rec;
        if (COUNT && pt_first_active_lane()) {
            if (node_phase) { tc.it_node++; tc.act_node += n_node; }
            else { tc.it_rec++; tc.act_rec += n_rec; }
        }
        if (node_phase) {
            if (!has_node) continue;
            const int a = cur;
            const float4 q0 = sc.nodes[a + 0], q1 = sc.nodes[a + 1], q2 = sc.nodes[a + 2], q3 = sc.nodes[a + 3];
            int l2 = __float_as_int(q3.x), l3 = __float_as_int(q3.y);
            const float sc_y = q3.z, sc_z = q3.w;
            asm volatile("" : "+v"(l2), "+v"(l3));
            if (COUNT) tc.inner++;
            const float sx = q0.w * idx, sy = sc_y * idy, sz = sc_z * idz;  // per-axis grid step / direction
            const float bx = fmaf(q0.x, idx, -oodx), by = fmaf(q0.y, idy, -oody), bz = fmaf(q0.z, idz, -oodz);
            const uint32_t qlx = __float_as_uint(q1.x), qly = __float_as_uint(q1.y), qlz = __float_as_uint(q1.z);
            const uint32_t qhx = __float_as_uint(q1.w), qhy = __float_as_uint(q2.x), qhz = __float_as_uint(q2.y);
            const bool px = idx >= 0.0f, py = idy >= 0.0f, pz = idz >= 0.0f;
            const uint32_t nx = px ? qlx : qhx, fx = px ? qhx : qlx;
            const uint32_t ny = py ? qly : qhy, fy = py ? qhy : qly;
            const uint32_t nz = pz ? qlz : qhz, fz = pz ? qhz : qlz;
            const int l0 = __float_as_int(q2.z), l1 = __float_as_int(q2.w);
            uint32_t key[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const pt_f2 tx = pt_fma2(pt_mk2((float)((nx >> (8 * k)) & 0xffu), (float)((fx >> (8 * k)) & 0xffu)), pt_mk2(sx, sx), pt_mk2(bx, bx));
                const pt_f2 ty = pt_fma2(pt_mk2((float)((ny >> (8 * k)) & 0xffu), (float)((fy >> (8 * k)) & 0xffu)), pt_mk2(sy, sy), pt_mk2(by, by));
                const pt_f2 tz = pt_fma2(pt_mk2((float)((nz >> (8 * k)) & 0xffu), (float)((fz >> (8 * k)) & 0xffu)), pt_mk2(sz, sz), pt_mk2(bz, bz));
                const float tmin = fmaxf(fmaxf(fmaxf(tx.x, ty.x), tz.x), 0.0f);
                const float tmax = fminf(fminf(fminf(tx.y, ty.y), tz.y), h.t);
                const bool hit = tmin <= tmax;  // unused slots hold inverted boxes
                key[k] = hit ? ((__float_as_uint(tmin) & 0x7ffffffcu) | (uint32_t)k) : 0xffffffffu;
            }
#define PT_CE(i, j) { const uint32_t lo_ = min(key[i], key[j]), hi_ = max(key[i], key[j]); key[i] = lo_; key[j] = hi_; }
            PT_CE(0, 1) PT_CE(2, 3) PT_CE(0, 2) PT_CE(1, 3) PT_CE(1, 2)
#undef PT_CE
#define PT_LINK(kk) (((kk) & 3u) == 0u ? l0 : (((kk) & 3u) == 1u ? l1 : (((kk) & 3u) == 2u ? l2 : l3)))
            if (key[3] != 0xffffffffu) { sp++; stk.put(sp, PT_LINK(key[3])); }
            if (key[2] != 0xffffffffu) { sp++; stk.put(sp, PT_LINK(key[2])); }
            if (key[1] != 0xffffffffu) { sp++; stk.put(sp, PT_LINK(key[1])); }
            if (key[0] != 0xffffffffu) {
                cur = PT_LINK(key[0]);
            } else {
                cur = stk.get(sp);
                sp--;
            }
#undef PT_LINK
            if (COUNT && cur < 0) tc.leaves++;
            if (cur < 0 && pend == 0) {  // park the leaf, go on with the next entry
                pend = cur;
                cur = stk.get(sp);
                sp--;
                if (COUNT && cur < 0) tc.leaves++;
            }
        } else {
            if (!has_rec) continue;
            const int a = ~pend;
            const float4 q0 = sc.nodes[a + 0], q1 = sc.nodes[a + 1], q2 = sc.nodes[a + 2];
            if (COUNT) tc.tris++;
            const v3 v0 = V3(q0.x, q0.y, q0.z), e1 = V3(q1.x, q1.y, q1.z), e2 = V3(q2.x, q2.y, q2.z);
            const float t = pt_mt_intersect(v0, e1, e2, o, d, cull);
            const int id = __float_as_int(q0.w);
            if (t > 0.0f && (t < h.t || (t == h.t && h.tri != -1 && id < h.tri))) {
                h.t = t;
                h.tri = id;
                h.rec = a;
            }
            if (__float_as_int(q1.w) != 0) {  // last record of the leaf
                pend = 0;
                if (cur < 0) {  // a second leaf was waiting in cur
                    pend = cur;
                    cur = stk.get(sp);
                    sp--;
                }
            } else {
                pend -= 4;  // ~(a + 4)
            }
        }
    }
    s.node = cur; s.sp = sp; s.h = h; s.leaf = pend;
    return cur == PT_SENTINEL && pend == 0;
}

template <bool COUNT, bool TOP, class STK>
__device__ __forceinline__ Hit trav_bvh2(const KScene& sc, v3 o, v3 d, bool cull, STK& stk,
                                         TravCount& tc, const float4* __restrict__ s_top) {
    TravState s;
    trav_begin(s, o, d, stk);
    trav_run<COUNT, false, TOP, STK>(s, sc, o, d, cull, stk, tc, 0, 0, s_top);
    return s.h;
}


template <int BLOCK>
__device__ __forceinline__ void lds_load_top(const KScene& sc, float4* __restrict__ s_top) {
    for (int i = threadIdx.x; i < sc.n_top * 4; i += BLOCK) s_top[(i & 3) * sc.n_top + (i >> 2)] = sc.nodes[sc.top_base + i];
    __syncthreads();
}

// ---------------------------------------------------------------------------------------
// One sample of one pixel: getSample, tracer.cu:27-339, cut into the pieces both kernels
// share: path_begin (camera ray), trav_* (closest hit), path_shade (spheres, shading, BRDF).
struct PathState {
    v3 o, d, mask, accu;
    uint32_t depth;
    pt_rng rng;
};

// The kernel-argument block seen through an opaque pointer (constant address space, scalar loads):
// a field read as K.x is fetched at that point instead of being preloaded and kept in SGPRs for
// the whole kernel.  KParams is the one and only argument of every kernel that uses this: offset 0.
#define PT_KARGS(K)                                                                               \
    const __attribute__((address_space(4))) KParams* K##_p =                                      \
        (const __attribute__((address_space(4))) KParams*)__builtin_amdgcn_kernarg_segment_ptr(); \
    asm volatile("" : "+s"(K##_p));                                                               \
    const __attribute__((address_space(4))) KParams& K = *K##_p

// RNG seed (tracer.cu:362-363) + getCamRayDir, cudaUtils.h:111-134 (origin ON the image plane)
__device__ __forceinline__ void path_begin(const KParams& P, int px, int py, uint64_t pix, uint64_t frame, PathState& ps) {
    PT_KARGS(K);   // camera: read where it is used, not held in SGPRs across the persistent loop
    ps.rng = pt_rng_init(pt_wang64(frame), pix);
    const float u0 = pt_rng_next(ps.rng), u1 = pt_rng_next(ps.rng);
    const float jx = u0 - 0.5f, jy = u1 - 0.5f;
    const float xs = ((((float)px - (float)P.W / 2.0f) + 0.5f) + jx) * K.cam.dist * K.cam.aspect * K.cam.fov / (float)(P.W - 1);
    const float ys = ((((float)py - (float)P.H / 2.0f) + 0.5f) + jy) * K.cam.dist * K.cam.fov / (float)(P.H - 1);
    const v3 front = V3(K.cam.front[0], K.cam.front[1], K.cam.front[2]);
    const v3 right = V3(K.cam.right[0], K.cam.right[1], K.cam.right[2]);
    const v3 up = V3(K.cam.up[0], K.cam.up[1], K.cam.up[2]);
    const v3 dir0 = vmadd(up, ys, vmadd(right, xs, vscale(front, K.cam.dist)));
    ps.o = vadd(V3(K.cam.pos[0], K.cam.pos[1], K.cam.pos[2]), dir0);
    ps.d = vnormalize(dir0);
    ps.mask = V3(1.f, 1.f, 1.f);
    ps.accu = V3(0.f, 0.f, 0.f);
    ps.depth = 0;
}

// One bounce after the closest triangle hit `h` is known (tracer.cu:98-296).  Returns true
// when the sample is complete (col_out valid), false when ps holds the next ray segment.
__device__ __forceinline__ bool path_shade(const KParams& P, PathState& ps, const Hit& h, v3& col_out) {
    PT_KARGS(K);   // shading scalars: read where they are used (see the sphere loop)
    const v3 o = ps.o, d = ps.d;
    v3 mask = ps.mask, accu = ps.accu;
    pt_rng rng = ps.rng;
    {
    int geom = 3;  // GeoType::NONE
    int sph_id = -1;
    float scene_t = h.t;
    if (h.tri != -1) geom = 0;
    // intersectAllSpeheres, cudaUtils.h:221-236 (uniform loop, scalar loads)
    // The reference scene has 8 spheres (BasicScene.cpp:181-202): they ride in the kernel-argument
    // block and the loop is unrolled, so the data arrives by scalar loads issued up front.  Inside
    // the divergent service phase hipcc otherwise keeps the loop counter in a VGPR and fetches each
    // sphere with dependent vector loads (a quarter of the kernel's vector-memory instructions).
    if (P.sc.n_spheres <= PT_KSPHERES) {
        // The centres/radii ride in the kernel-argument block, but are re-read HERE, one scalar load
        // per sphere behind an opaque pointer: kept in SGPRs across the whole persistent loop they
        // push ~50 other scalars into spill lanes (v_readlane/v_writelane on the hot path; measured
        // -4 % frame time, and what lets 6 waves per SIMD pay off).  A plain global pointer makes
        // hipcc fetch them with per-lane vector loads instead (+2 %).
        // KParams is the one and only kernel argument of every kernel that shades: offset 0.
        typedef const __attribute__((address_space(4))) float kfloat;
        kfloat* kp = (kfloat*)&K.ksph[0];
#pragma unroll
        for (int i = 0; i < PT_KSPHERES; i++) {
            if (i < P.sc.n_spheres) {
                struct { float px, py, pz, rad; } s = {kp[11 * i], kp[11 * i + 1], kp[11 * i + 2], kp[11 * i + 3]};
                const float ts = pt_sphere_intersect(s.px, s.py, s.pz, s.rad, o, d);
                if (ts != 0.0f && ts < scene_t && ts > 0.01f) { scene_t = ts; sph_id = i; geom = 1; }
            }
        }
    } else {
        for (int i = 0; i < P.sc.n_spheres; i++) {
            const pt_sphere_d& s = P.sc.spheres[i];
            const float ts = pt_sphere_intersect(s.px, s.py, s.pz, s.rad, o, d);
            if (ts != 0.0f && ts < scene_t && ts > 0.01f) { scene_t = ts; sph_id = i; geom = 1; }
        }
    }
    v3 hitpos = vmadd(d, scene_t, o);
    v3 n, nl, objcol, emit;
    int mat;
    float phong = K.phong;
    if (geom == 1) {
        const pt_sphere_d& s = P.sc.spheres[sph_id];
        n = vnormalize(vsub(hitpos, V3(s.px, s.py, s.pz)));
        nl = vdot(n, d) < 0 ? n : vscale(n, -1.0f);
        objcol = V3(s.col[0], s.col[1], s.col[2]);
        emit = V3(s.emi[0], s.emi[1], s.emi[2]);
        mat = s.mat;
    } else if (geom == 0) {
        n = vnormalize(pt_hit_normal(P.sc, h));
        nl = n;  // tracer.cu:126-127
        if ((K.flags & PT_FLAG_FACE_FORWARD) && !(vdot(n, d) < 0)) nl = vscale(n, -1.0f);
        if (K.tri_matid) {  // extension: per-triangle material row
            const int row = K.tri_matid[h.tri];
            const float4 m0 = K.mat_table[2 * row], m1 = K.mat_table[2 * row + 1];
            objcol = V3(m0.x, m0.y, m0.z);
            emit = V3(m0.w, m1.x, m1.y);
            mat = __float_as_int(m1.z);
            phong = m1.w;
        } else {
            objcol = V3(K.tri_col[0], K.tri_col[1], K.tri_col[2]);
            emit = V3(K.tri_emi[0], K.tri_emi[1], K.tri_emi[2]);
            mat = K.tri_mat;
        }
    } else {
        col_out = V3(K.bk[0], K.bk[1], K.bk[2]);  // tracer.cu:140-142: unmasked background
        return true;
    }
    accu = vadd(accu, vmul(mask, emit));

    if ((K.flags & PT_FLAG_RUSSIAN_ROULETTE) && ps.depth >= 2) {  // extension
        const float pr = fmaxf(objcol.x, fmaxf(objcol.y, objcol.z));
        if (!(pt_rng_next(rng) < pr)) { col_out = accu; return true; }
        objcol = vscale(objcol, 1.0f / pr);
    }

    v3 nextdir;
    if (mat == PT_MAT_DIFF) {  // tracer.cu:156-186
        if (!(K.flags & PT_FLAG_COSINE_DIFF)) {
            (void)pt_rng_next(rng);
            (void)pt_rng_next(rng);
        }
        v3 nt = fabsf(nl.x) > fabsf(nl.y) ? V3(nl.z, 0.f, -nl.x) : V3(0.f, -nl.z, nl.y);
        nt = vnormalize(nt);
        const v3 nb = vnormalize(vcross(nl, nt));
        const float f1 = pt_rng_next(rng), f2 = pt_rng_next(rng);
        float c, s;
        pt_sincos2pi(f1, c, s);
        v3 rv;
        if (K.flags & PT_FLAG_COSINE_DIFF) {  // extension: pdf = cos/pi
            const float r2s = sqrtf(f2);
            rv = V3(c * r2s, sqrtf(1.0f - f2), s * r2s);
        } else {
            rv = V3(c * f2, sqrtf(1.0f - f2 * f2), s * f2);  // cudaUtils.h:185-192
        }
        nextdir = vnormalize(vmadd(nt, rv.z, vmadd(nl, rv.y, vscale(nb, rv.x))));
        hitpos = vmadd(nl, 0.001f, hitpos);
        mask = vmul(mask, objcol);
    } else if (mat == PT_MAT_SPEC) {  // :190-203
        nextdir = vnormalize(vmadd(nl, -2.0f * vdot(nl, d), d));
        hitpos = vmadd(nl, 0.001f, hitpos);
        mask = vmul(mask, objcol);
    } else if (mat == PT_MAT_REFR) {  // :205-256
        const bool into = vdot(n, nl) > 0;
        const float nc = K.air_ior, ntt = K.glass_ior;
        const float nnt = into ? nc / ntt : ntt / nc;
        const float ddn = vdot(d, nl);
        const float cos2t = 1.0f - nnt * nnt * (1.0f - ddn * ddn);
        if (cos2t < 0.0f) {
            nextdir = vnormalize(vmadd(n, -2.0f * vdot(n, d), d));
            hitpos = vmadd(nl, 0.001f, hitpos);
        } else {
            const float k = (into ? 1.0f : -1.0f) * (ddn * nnt + sqrtf(cos2t));
            const v3 tdir = vnormalize(vmadd(n, -k, vscale(d, nnt)));
            const bool fix = (K.flags & PT_FLAG_GLASS_FIX) != 0;  // extension
            const float R0 = fix ? ((ntt - nc) * (ntt - nc)) / ((ntt + nc) * (ntt + nc))
                                 : (ntt - nc) * (ntt - nc) / (ntt + nc) * (ntt + nc);  // sic, :230
            const float c = 1.0f - (into ? -ddn : vdot(tdir, n));
            const float Re = R0 + (1.0f - R0) * c * c * c * c * c;
            const float Tr = 1 - Re;
            const float Pp = 0.25f + 0.5f * Re;
            const float RP = Re / Pp, TP = Tr / (1.0f - Pp);
            bool transmitted = false;
            if (pt_rng_next(rng) < (fix ? Pp : 0.2f)) {
                mask = vscale(mask, RP);
                nextdir = vnormalize(vmadd(n, -2.0f * vdot(n, d), d));
            } else {
                mask = vscale(mask, TP);
                nextdir = vnormalize(tdir);
                transmitted = true;
            }
            hitpos = vmadd(nl, (fix && transmitted) ? -0.001f : 0.001f, hitpos);
        }
    } else {  // METAL :257-293
        const float f1 = pt_rng_next(rng), r2 = pt_rng_next(rng);
        float cphi, sphi;
        pt_sincos2pi(f1, cphi, sphi);
        const float cosT = pt_pow01(1.0f - r2, 1.0f / (phong + 1.0f));
        const float sinT = sqrtf(1.0f - cosT * cosT);
        const v3 w1 = vnormalize(vmadd(nl, -2.0f * vdot(nl, d), d));
        const v3 ax = ((double)fabsf(w1.x) > 0.1) ? V3(0.f, 1.f, 0.f) : V3(1.f, 0.f, 0.f);
        const v3 uu = vnormalize(vcross(ax, w1));
        const v3 vv = vcross(w1, uu);
        const v3 base = vmadd(vv, sphi * sinT, vscale(uu, cphi * sinT));
        if (K.flags & PT_FLAG_METAL_LITERAL_W) {
            const float wc = (float)P.W * cosT;  // tracer.cu:280
            nextdir = V3(base.x + wc, base.y + wc, base.z + wc);
        } else {
            nextdir = vmadd(w1, cosT, base);
        }
        nextdir = vnormalize(nextdir);
        hitpos = vmadd(nl, 0.0001f, hitpos);
        mask = vmul(mask, objcol);
    }
        ps.o = hitpos;
        ps.d = nextdir;
    }
    ps.mask = mask; ps.accu = accu; ps.rng = rng;
    ps.depth++;
    if (ps.depth >= P.depth) { col_out = accu; return true; }  // tracer.cu:305
    return false;
}

// running mean with per-frame clamp, tracer.cu:386-391
__device__ __forceinline__ void pt_accumulate(float& ax, float& ay, float& az, v3 col, uint64_t N) {
    const float fm1 = (float)(N - 1), inv = 1.0f / (float)N;
    if (N == 1) { ax = 0.f; ay = 0.f; az = 0.f; } else { ax *= fm1; ay *= fm1; az *= fm1; }
    ax = pt_clamp01((ax + col.x) * inv);
    ay = pt_clamp01((ay + col.y) * inv);
    az = pt_clamp01((az + col.z) * inv);
}

// 8-bit truncating pack 0x00BBGGRR, tracer.cu:394-398 + cudaUtils.h:99-105
__device__ __forceinline__ uint32_t pt_pack_rgba(float ax, float ay, float az) {
    const uint32_t r = (uint32_t)(unsigned char)(255.0f * ax);
    const uint32_t g = (uint32_t)(unsigned char)(255.0f * ay);
    const uint32_t b = (uint32_t)(unsigned char)(255.0f * az);
    return (b << 16) | (g << 8) | r;
}

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
                done = path_shade(P, ps, ts.h, col);
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

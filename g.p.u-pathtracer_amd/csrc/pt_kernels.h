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
                                              TravCount& tc, int n_dead, int batch, int poll_index = -1, int poll_seen = 0) {
    int cur = s.node, sp = s.sp;
    Hit h = s.h;
    const float idx = s.idx, idy = s.idy, idz = s.idz, oodx = s.oodx, oody = s.oody, oodz = s.oodz;
    int iter = 0;
    for (;;) {
        // role-split kernel: lanes that found the ready queue empty watch its tail (an LDS word) every
        // 8th step, so that new segments are picked up while the other lanes are still walking
        if (DYN && poll_index >= 0 && n_dead > 0 && (++iter & 7) == 0 &&
            __atomic_load_n(&((int*)s_dyn)[poll_index], __ATOMIC_RELAXED) != poll_seen) break;
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

// What a segment ended on once the spheres have been tested too (intersectAllSpeheres,
// cudaUtils.h:221-236, after the triangle hit of the walk): the distance, GeoType and sphere number.
struct SceneHit {
    float t;
    int geom;     // 0 triangle, 1 sphere, 3 nothing (GeoType, CommomStructs.hpp)
    int sph_id;
};

// sph_tab: float index into the dynamic LDS of a copy of the first PT_KSPHERES spheres (11 floats
// each, then centre+radius as float4s), or -1 = read the kernel arguments.
__device__ __forceinline__ SceneHit pt_closest_sphere(const KParams& P, v3 o, v3 d, const Hit& h, int sph_tab = -1) {
    PT_KARGS(K);
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
                struct { float px, py, pz, rad; } s;
                if (sph_tab >= 0) {  // role-split kernel: one wave shades alone on its SIMD, so eight scalar-load
                                     // round trips in a row are exposed; the LDS copy is read as pipelined broadcasts
                    const float4 c = *(const float4*)((const float*)s_dyn + sph_tab + 88 + 4 * i);
                    s.px = c.x; s.py = c.y; s.pz = c.z; s.rad = c.w;
                } else {
                    s.px = kp[11 * i]; s.py = kp[11 * i + 1]; s.pz = kp[11 * i + 2]; s.rad = kp[11 * i + 3];
                }
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
    SceneHit sh;
    sh.t = scene_t; sh.geom = geom; sh.sph_id = sph_id;
    return sh;
}

// One bounce after the closest triangle hit `h` and the sphere tests `sh` are known
// (tracer.cu:98-296).  Returns true when the sample is complete (col_out valid), false when ps
// holds the next ray segment.  With sph_tab >= 0 the winner sphere's attributes are one short LDS
// gather instead of a global one.
// tri_n: pt_hit_normal of the walk's triangle hit (read only when the triangle is what was hit; the
// caller fetches it early so that the latency hides behind other work).
__device__ __forceinline__ bool path_shade_hit(const KParams& P, PathState& ps, const Hit& h, const SceneHit& sh, v3 tri_n, v3& col_out, int sph_tab = -1) {
    PT_KARGS(K);   // shading scalars: read where they are used (see the sphere loop)
    const v3 o = ps.o, d = ps.d;
    v3 mask = ps.mask, accu = ps.accu;
    pt_rng rng = ps.rng;
    {
    const int geom = sh.geom, sph_id = sh.sph_id;
    const float scene_t = sh.t;
    v3 hitpos = vmadd(d, scene_t, o);
    v3 n, nl, objcol, emit;
    int mat;
    float phong = K.phong;
    if (geom == 1) {
        pt_sphere_d s;
        if (sph_tab >= 0 && sph_id < PT_KSPHERES) {
            const float* t = (const float*)s_dyn + sph_tab + 11 * sph_id;
            s.px = t[0]; s.py = t[1]; s.pz = t[2]; s.rad = t[3];
            s.emi[0] = t[4]; s.emi[1] = t[5]; s.emi[2] = t[6];
            s.col[0] = t[7]; s.col[1] = t[8]; s.col[2] = t[9];
            s.mat = __float_as_int(t[10]);
        } else {
            s = P.sc.spheres[sph_id];
        }
        n = vnormalize(vsub(hitpos, V3(s.px, s.py, s.pz)));
        nl = vdot(n, d) < 0 ? n : vscale(n, -1.0f);
        objcol = V3(s.col[0], s.col[1], s.col[2]);
        emit = V3(s.emi[0], s.emi[1], s.emi[2]);
        mat = s.mat;
    } else if (geom == 0) {
        n = vnormalize(tri_n);
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

// spheres + shading in one go (the kernels that shade in the lane that walked)
__device__ __forceinline__ bool path_shade(const KParams& P, PathState& ps, const Hit& h, v3& col_out, int sph_tab = -1) {
    // the triangle's normal is asked for NOW so that its latency hides behind the sphere tests
    v3 tri_n = V3(0.f, 0.f, 0.f);
    if (h.tri != -1) tri_n = pt_hit_normal(P.sc, h);
    const SceneHit sh = pt_closest_sphere(P, ps.o, ps.d, h, sph_tab);
    return path_shade_hit(P, ps, h, sh, tri_n, col_out, sph_tab);
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


// ---------------------------------------------------------------------------------------
// Role-split variant (PT_KERNEL_WAVEFRONT): shading and traversal no longer share a wave's lanes.
// The schedule statistics of k_trace_persist_bvh2 put the shading passes at ~46 % of
// the issued VALU work at 56 % lane use and the node steps at 55 %: both draw on the same 64 lanes,
// so `batch` trades one against the other.  Here a block's waves take fixed roles and exchange rays
// only BETWEEN segments (the traversal stack is empty then), through LDS:
//   * every ray of the block owns one SLOT from its first camera ray to its last bounce: origin,
//     direction and hit (9 dwords) in LDS — what the two roles hand to each other — and the rest of
//     the path state (mask, accu, rng, depth, pixel, sample: 12 dwords) in a global array that only
//     the shader waves touch (it stays in L2);
//   * TRACER waves (PT_ROLE_TRACERS of the 4) pop slot numbers from the ready queue, load o/d, walk,
//     write the hit back and push the slot to the shade queue — they never hold path state;
//   * SHADER waves pop up to 64 finished segments, shade them at full width (path_shade, the same
//     arithmetic), and push the continuing ones to the ready queue; when nothing waits for shading
//     they start new paths (path_begin, 64 at a time) from the global work queue.
// Queues are rings of slot numbers with one entry per slot, so they can never overflow and no
// wave ever blocks on a full queue; a consumer that reserved an entry the producer has not
// written yet spins on that entry only.  Every spin is bounded (PT_ROLE_SPIN_MAX polls), after
// which the wave raises the error word and every wave of the block drains out: the grid always ends.
// Per-ray arithmetic is that of the other kernels (same path_begin / walk / path_shade), so the
// image is bit-identical; only WHICH lane does it changes.
#ifndef PT_ROLE_TRACERS
#define PT_ROLE_TRACERS 3            // tracer waves of the block's 4 (the rest shade)
#endif
#ifndef PT_ROLE_SLOTS
#define PT_ROLE_SLOTS 272            // rays in flight per block (LDS: 48 B each + 12 KB of tracer stacks = 25.4 KB)
#endif
#define PT_SLOT_DW 9                 // LDS part of a slot: o, d, hit; the rest (PT_COLD_DW) lives in global memory
#define PT_COLD_DW 12                // mask, accu, rng s0 s1 n, depth, pixel, sample
#define PT_ROLE_SPIN_MAX (1 << 22)
#ifndef PT_ROLE_S_MIN
#define PT_ROLE_S_MIN 48             // finished segments that make a shading pass worth starting
#define PT_ROLE_B_MIN 32             // free slots that make a path-start pass worth starting
#define PT_ROLE_T_LOW 16             // ready segments below which the shaders stop waiting for full passes
#define PT_ROLE_HELP_MIN 8           // finished segments that make an idle TRACER wave take a shading pass
#endif
#ifndef PT_ROLE_BLOCK
#define PT_ROLE_BLOCK 256            // threads per block: PT_ROLE_TRACERS tracer waves, the rest shade
#endif
#ifndef PT_ROLE_MIX
#define PT_ROLE_MIX 0
#endif
enum { RC_SQ_HEAD = 0, RC_SQ_TAIL, RC_TQ_HEAD, RC_TQ_TAIL, RC_FQ_HEAD, RC_FQ_TAIL, RC_ALIVE, RC_DRY, RC_ERROR, RC_WORDS = 16 };

#define LDSI(i) (((int*)s_dyn)[(i)])
#define LDSF(i) (((float*)s_dyn)[(i)])

// reserve up to `want` entries of ring [head, tail); returns the count and the first position (wave-uniform)
__device__ __forceinline__ int role_reserve(int head_i, int tail_i, int want, int lane, int& pos) {
    int n = 0, h = 0;
    if (lane == 0 && want > 0) {
        for (int tries = 0; tries < 64; tries++) {
            h = __atomic_load_n(&LDSI(head_i), __ATOMIC_RELAXED);
            const int t = __atomic_load_n(&LDSI(tail_i), __ATOMIC_RELAXED);
            n = min(want, t - h);
            if (n <= 0) { n = 0; break; }
            if (atomicCAS(&LDSI(head_i), h, h + n) == h) break;
            n = 0;
        }
    }
    pos = __builtin_amdgcn_readfirstlane(h);
    return __builtin_amdgcn_readfirstlane(n);
}

// the slot number stored at ring position p (spins until the producer has written it), entry cleared
__device__ __forceinline__ int role_take(int ring_i, int p, int err_i) {
    const int e = ring_i + (p % PT_ROLE_SLOTS);
    int v = 0;
    for (int spin = 0; spin < PT_ROLE_SPIN_MAX; spin++) {
        v = __atomic_load_n(&LDSI(e), __ATOMIC_ACQUIRE);
        if (v != 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
    if (v == 0) { atomicOr(&LDSI(err_i), 1); return -1; }
    __atomic_store_n(&LDSI(e), 0, __ATOMIC_RELAXED);
    return v - 1;
}

// lanes with `pred` append `slot` to a ring (one tail reservation per wave)
__device__ __forceinline__ void role_push(int ring_i, int tail_i, bool pred, int slot, int lane) {
    const unsigned long long m = __ballot(pred);
    const int n = __popcll(m);
    if (n == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(&LDSI(tail_i), n);
    base = __builtin_amdgcn_readfirstlane(base);
    if (pred) {
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        __atomic_store_n(&LDSI(ring_i + ((base + rank) % PT_ROLE_SLOTS)), slot + 1, __ATOMIC_RELEASE);
    }
}

// One shading pass of a wave: `got` finished segments starting at shade-queue position `pos`.
// Continuing rays go to the ready queue, finished paths write their sample and free their slot.
template <int SLOT_OFF, int SQ_OFF, int TQ_OFF, int FQ_OFF, int CTL>
__device__ __forceinline__ void role_shade_pass(const KParams& P, int lane, int got, int pos) {
    constexpr int F = PT_ROLE_SLOTS;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // cold state written by another wave of this block (same CU, same L1)
    int slot = -1;
    bool cont = false;
    if (lane < got) slot = role_take(SQ_OFF, pos + lane, CTL + RC_ERROR);
    if (slot >= 0) {
        const int b = SLOT_OFF + slot * PT_SLOT_DW;
        PathState ps;
        ps.o = V3(LDSF(b + 0), LDSF(b + 1), LDSF(b + 2));
        ps.d = V3(LDSF(b + 3), LDSF(b + 4), LDSF(b + 5));
        Hit h;
        h.t = LDSF(b + 6); h.tri = LDSI(b + 7); h.rec = LDSI(b + 8);
        float4* cold = P.roles_state + ((size_t)blockIdx.x * F + (size_t)slot) * (PT_COLD_DW / 4);
        const float4 c0 = cold[0], c1 = cold[1], c2 = cold[2];
        ps.mask = V3(c0.x, c0.y, c0.z);
        ps.accu = V3(c0.w, c1.x, c1.y);
        ps.rng.s0 = __float_as_uint(c1.z); ps.rng.s1 = __float_as_uint(c1.w); ps.rng.n = __float_as_uint(c2.x);
        ps.depth = __float_as_uint(c2.y);
        const uint32_t pix = __float_as_uint(c2.z), s_idx = __float_as_uint(c2.w);
        v3 col = V3(0.f, 0.f, 0.f);
        const bool done = path_shade(P, ps, h, col, CTL + RC_WORDS);
        if (!done) {
            LDSF(b + 0) = ps.o.x; LDSF(b + 1) = ps.o.y; LDSF(b + 2) = ps.o.z;
            LDSF(b + 3) = ps.d.x; LDSF(b + 4) = ps.d.y; LDSF(b + 5) = ps.d.z;
            cold[0] = make_float4(ps.mask.x, ps.mask.y, ps.mask.z, ps.accu.x);
            cold[1] = make_float4(ps.accu.y, ps.accu.z, c1.z, c1.w);
            cold[2] = make_float4(__uint_as_float(ps.rng.n), __uint_as_float(ps.depth), c2.z, c2.w);
            cont = true;
        } else if (P.samples) {
            float* dst = P.samples + 3 * ((size_t)s_idx * (size_t)P.W * (size_t)P.H + (size_t)pix);
            dst[0] = col.x; dst[1] = col.y; dst[2] = col.z;
        } else {
            float* acc = P.accum + 3 * (size_t)pix;
            float ax = 0.f, ay = 0.f, az = 0.f;
            if (P.sample_index != 1) { ax = acc[0]; ay = acc[1]; az = acc[2]; }
            pt_accumulate(ax, ay, az, col, P.sample_index);
            acc[0] = ax; acc[1] = ay; acc[2] = az;
            if ((P.flags & PT_FLAG_WRITE_RGBA) && P.rgba) P.rgba[pix] = pt_pack_rgba(ax, ay, az);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // cold state stored before the slot is handed on
    role_push(TQ_OFF, CTL + RC_TQ_TAIL, cont, slot, lane);
    const bool dead = slot >= 0 && !cont;
    role_push(FQ_OFF, CTL + RC_FQ_TAIL, dead, slot, lane);   // slot free again ...
    const int n_deadr = __popcll(__ballot(dead));
    if (lane == 0 && n_deadr) atomicSub(&LDSI(CTL + RC_ALIVE), n_deadr);   // ... then the ray count drops
}

template <int OCC, int LSTK>
__global__ void __launch_bounds__(PT_ROLE_BLOCK, OCC) k_trace_roles(const KParams P) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NT = PT_ROLE_TRACERS, F = PT_ROLE_SLOTS;
    // blocks alternate between NT tracers and NT-1 (PT_ROLE_MIX): the shading share of the work sits
    // between one and two waves of four
    const int nt = NT - ((PT_ROLE_MIX && (blockIdx.x & 1)) ? 1 : 0);
    const int NS = PT_ROLE_BLOCK / 64 - nt;
    constexpr int STK_OFF = 0, SLOT_OFF = NT * 64 * LSTK, SQ_OFF = SLOT_OFF + F * PT_SLOT_DW, TQ_OFF = SQ_OFF + F, FQ_OFF = TQ_OFF + F,
                  CTL = FQ_OFF + F;
    if (tid < 11 * PT_KSPHERES) {  // the spheres' attributes (11 floats each), then centre+radius as float4s
        PT_KARGS(K);
        const float v = ((const __attribute__((address_space(4))) float*)&K.ksph[0])[tid];
        LDSF(CTL + RC_WORDS + tid) = v;
        if (tid % 11 < 4) LDSF(CTL + RC_WORDS + 88 + 4 * (tid / 11) + tid % 11) = v;
    }
    for (int i = tid; i < F; i += PT_ROLE_BLOCK) { LDSI(SQ_OFF + i) = 0; LDSI(TQ_OFF + i) = 0; LDSI(FQ_OFF + i) = i + 1; }
    if (tid < RC_WORDS) LDSI(CTL + tid) = tid == RC_FQ_TAIL ? F : 0;
    __syncthreads();
    const bool cull = P.cull != 0;

    if (wave < nt) {
        // ------------------------------------------------------------------ tracer wave
        TravOverflow<LSTK> stk_ovf;
        TravStack<LSTK, NT * 64> stk(__builtin_amdgcn_readfirstlane(STK_OFF + wave * 64), stk_ovf);
        TravState ts;
        ts.idx = ts.idy = ts.idz = ts.oodx = ts.oody = ts.oodz = 0.f;
        ts.node = PT_SENTINEL; ts.leaf = 0; ts.sp = 0;
        ts.h.t = PT_F32_MAX; ts.h.tri = -1; ts.h.rec = 0;
        TravCount tc;
        tc.inner = tc.tris = tc.leaves = 0;
        v3 o = V3(0.f, 0.f, 0.f), d = V3(0.f, 0.f, 1.f);
        int slot = -1;           // -1: the lane is empty
        bool walking = false;
        int idle_polls = 0;
#ifdef PT_ROLES_STATS
        uint32_t st_iters = 0, st_idle = 0, st_live = 0, st_help = 0;
#endif
        for (;;) {
            // 1. finished segments -> shade queue
            const bool fin = slot >= 0 && !walking;
            if (fin) {   // (running the sphere tests here, on the lanes that just finished, costs +11 %: measured)
                const int b = SLOT_OFF + slot * PT_SLOT_DW;
                LDSF(b + 6) = ts.h.t; LDSI(b + 7) = ts.h.tri; LDSI(b + 8) = ts.h.rec;
            }
            role_push(SQ_OFF, CTL + RC_SQ_TAIL, fin, slot, lane);
            if (fin) slot = -1;
            // 2. empty lanes <- ready queue
            const unsigned long long em = __ballot(slot < 0);
            const int n_empty = __popcll(em);
            int pos = 0;
            const int seen_tail = __builtin_amdgcn_readfirstlane(__atomic_load_n(&LDSI(CTL + RC_TQ_TAIL), __ATOMIC_RELAXED));
            const int got = role_reserve(CTL + RC_TQ_HEAD, CTL + RC_TQ_TAIL, n_empty, lane, pos);
            if (slot < 0) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
                if (rank < got) {
                    slot = role_take(TQ_OFF, pos + rank, CTL + RC_ERROR);
                    if (slot >= 0) {
                        const int b = SLOT_OFF + slot * PT_SLOT_DW;
                        o = V3(LDSF(b + 0), LDSF(b + 1), LDSF(b + 2));
                        d = V3(LDSF(b + 3), LDSF(b + 4), LDSF(b + 5));
                        trav_begin(ts, o, d, stk, P.sc.wide_root);
                        walking = true;
                    }
                }
            }
            const int n_live = __popcll(__ballot(slot >= 0));
#ifdef PT_ROLES_STATS
            st_iters++; st_live += n_live; if (n_live == 0) st_idle++;
#endif
            if (__atomic_load_n(&LDSI(CTL + RC_ERROR), __ATOMIC_RELAXED) != 0) break;
            if (n_live == 0) {
                // nothing to walk.  A tracer wave without rays is free to shade: if finished segments wait,
                // take a pass of them (the roles balance themselves: starving tracers refill their own queue)
                {
                    int sq_n = 0;
                    if (lane == 0) sq_n = __atomic_load_n(&LDSI(CTL + RC_SQ_TAIL), __ATOMIC_RELAXED) - __atomic_load_n(&LDSI(CTL + RC_SQ_HEAD), __ATOMIC_RELAXED);
                    sq_n = __builtin_amdgcn_readfirstlane(sq_n);
                    if (sq_n >= PT_ROLE_HELP_MIN) {
                        int hpos = 0;
                        const int hgot = role_reserve(CTL + RC_SQ_HEAD, CTL + RC_SQ_TAIL, 64, lane, hpos);
                        if (hgot > 0) {
#ifdef PT_ROLES_STATS
                            st_help++;
#endif
                            role_shade_pass<SLOT_OFF, SQ_OFF, TQ_OFF, FQ_OFF, CTL>(P, lane, hgot, hpos);
                            idle_polls = 0;
                            continue;
                        }
                    }
                }
                // done when every shader wave has found the global queue dry and no ray is left
                if (__atomic_load_n(&LDSI(CTL + RC_DRY), __ATOMIC_RELAXED) >= NS && __atomic_load_n(&LDSI(CTL + RC_ALIVE), __ATOMIC_RELAXED) == 0) break;
                if (++idle_polls > PT_ROLE_SPIN_MAX) { atomicOr(&LDSI(CTL + RC_ERROR), 2); break; }
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            idle_polls = 0;
            // 3. walk; leave when `batch` lanes have finished, or when empty lanes see new ready segments
            const int n_dead = 64 - n_live;
            if (walking) {
                const bool done = trav_run_wide<false, true, false, false>(ts, P.sc, o, d, cull, stk, tc, n_dead, P.batch,
                                                                          got < n_empty ? CTL + RC_TQ_TAIL : -1, seen_tail);
                if (done) walking = false;
            }
        }
#ifdef PT_ROLES_STATS
        if (lane == 0) {
            atomicAdd(&P.counters[6], (unsigned long long)st_iters);
            atomicAdd(&P.counters[7], (unsigned long long)st_idle);
            atomicAdd(&P.counters[8], (unsigned long long)st_live);
            atomicAdd(&P.counters[14], (unsigned long long)st_help);
        }
#endif
    } else {
        // ------------------------------------------------------------------ shader wave
#ifdef PT_ROLE_SHADER_PRIO
        __builtin_amdgcn_s_setprio(PT_ROLE_SHADER_PRIO);   // the block's one shader wave is what the tracers wait for
#endif
        const uint32_t slots_per_sample = (uint32_t)P.n_tiles * 64u;
        const uint32_t total = slots_per_sample * (P.samples ? P.spp : 1u);
        const uint32_t chunk = (uint32_t)P.chunk;
        const uint32_t shard_chunks = ((total + chunk - 1) / chunk + PT_SHARDS - 1) / PT_SHARDS;
        uint32_t chunk_next = 0, chunk_end = 0;
        bool queue_empty = false, dry_flagged = false;
        int shard = (int)(blockIdx.x & (PT_SHARDS - 1));
        int idle_polls = 0;
#ifdef PT_ROLES_STATS
        uint32_t ss_pass = 0, ss_got = 0, ss_idle = 0, ss_bpass = 0, ss_bgot = 0;
#endif
        for (;;) {
            if (__atomic_load_n(&LDSI(CTL + RC_ERROR), __ATOMIC_RELAXED) != 0) break;
            // what to do next: a pass is worth its ~1 000 instructions only at high lane use, so wait for a
            // wave's worth of finished segments (or of free slots for new paths) unless the tracers are
            // about to run dry
            int sq_n = 0, tq_n = 0, fq_n = 0;
            if (lane == 0) {
                sq_n = __atomic_load_n(&LDSI(CTL + RC_SQ_TAIL), __ATOMIC_RELAXED) - __atomic_load_n(&LDSI(CTL + RC_SQ_HEAD), __ATOMIC_RELAXED);
                tq_n = __atomic_load_n(&LDSI(CTL + RC_TQ_TAIL), __ATOMIC_RELAXED) - __atomic_load_n(&LDSI(CTL + RC_TQ_HEAD), __ATOMIC_RELAXED);
                fq_n = __atomic_load_n(&LDSI(CTL + RC_FQ_TAIL), __ATOMIC_RELAXED) - __atomic_load_n(&LDSI(CTL + RC_FQ_HEAD), __ATOMIC_RELAXED);
            }
            sq_n = __builtin_amdgcn_readfirstlane(sq_n); tq_n = __builtin_amdgcn_readfirstlane(tq_n); fq_n = __builtin_amdgcn_readfirstlane(fq_n);
            const bool can_begin = !queue_empty && fq_n > 0;
            const bool hungry = tq_n < PT_ROLE_T_LOW;    // the ready queue is nearly empty
            const bool do_shade = sq_n >= PT_ROLE_S_MIN || (sq_n > 0 && hungry && !(can_begin && fq_n >= PT_ROLE_B_MIN));
            const bool do_begin = !do_shade && can_begin && (fq_n >= PT_ROLE_B_MIN || hungry);
            // A. finished segments: shade them, 64 at a time
            int pos = 0;
            const int got = do_shade ? role_reserve(CTL + RC_SQ_HEAD, CTL + RC_SQ_TAIL, 64, lane, pos) : 0;
            if (got > 0) {
                idle_polls = 0;
#ifdef PT_ROLES_STATS
                ss_pass++; ss_got += got;
#endif
                role_shade_pass<SLOT_OFF, SQ_OFF, TQ_OFF, FQ_OFF, CTL>(P, lane, got, pos);
                continue;
            }
            // B. start new paths into free slots
            if (do_begin) {
                if (chunk_next == chunk_end) {
                    for (int tries = 0; tries < PT_SHARDS; tries++) {
                        uint32_t k = 0;
                        if (lane == 0) k = atomicAdd(P.queue + shard * PT_SHARD_STRIDE, 1u);
                        k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
                        const uint32_t first = (k * PT_SHARDS + (uint32_t)shard) * chunk;
                        if (k < shard_chunks && first < total) {
                            chunk_next = first;
                            chunk_end = min(first + chunk, total);
                            break;
                        }
                        shard = (shard + 1) & (PT_SHARDS - 1);
                    }
                    if (chunk_next == chunk_end) queue_empty = true;
                }
                if (!queue_empty) {
                    int fpos = 0;
                    const int want = (int)min(64u, chunk_end - chunk_next);
                    const int n_new = role_reserve(CTL + RC_FQ_HEAD, CTL + RC_FQ_TAIL, want, lane, fpos);
                    if (n_new > 0) {
                        idle_polls = 0;
#ifdef PT_ROLES_STATS
                        ss_bpass++; ss_bgot += n_new;
#endif
                        if (lane == 0) atomicAdd(&LDSI(CTL + RC_ALIVE), n_new);   // before the slots become visible
                        int slot = -1;
                        bool ready = false;
                        if (lane < n_new) slot = role_take(FQ_OFF, fpos + lane, CTL + RC_ERROR);
                        if (slot >= 0) {
                            uint32_t q = chunk_next + (uint32_t)lane;
                            uint32_t s_first = 0;
                            if (P.samples) { s_first = q / slots_per_sample; q -= s_first * slots_per_sample; }
                            int tx, ty;
                            bool inside = false;
                            if (pt_tile_coords(P, (int)(q >> 6), tx, ty)) {
                                const int px = tx * PT_TILE + (int)(q & 7u), py = ty * PT_TILE + (int)((q >> 3) & 7u);
                                if (px < P.W && py < P.H) {
                                    inside = true;
                                    const uint32_t pix = (uint32_t)py * (uint32_t)P.W + (uint32_t)px;
                                    PathState ps;
                                    path_begin(P, px, py, (uint64_t)pix, P.frame + s_first, ps);
                                    const int b = SLOT_OFF + slot * PT_SLOT_DW;
                                    LDSF(b + 0) = ps.o.x; LDSF(b + 1) = ps.o.y; LDSF(b + 2) = ps.o.z;
                                    LDSF(b + 3) = ps.d.x; LDSF(b + 4) = ps.d.y; LDSF(b + 5) = ps.d.z;
                                    float4* cold = P.roles_state + ((size_t)blockIdx.x * F + (size_t)slot) * (PT_COLD_DW / 4);
                                    cold[0] = make_float4(ps.mask.x, ps.mask.y, ps.mask.z, ps.accu.x);
                                    cold[1] = make_float4(ps.accu.y, ps.accu.z, __uint_as_float(ps.rng.s0), __uint_as_float(ps.rng.s1));
                                    cold[2] = make_float4(__uint_as_float(ps.rng.n), __uint_as_float(0u), __uint_as_float(pix), __uint_as_float(s_first));
                                    ready = true;
                                }
                            }
                            (void)inside;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // cold state stored before the slot is handed on
                        role_push(TQ_OFF, CTL + RC_TQ_TAIL, ready, slot, lane);
                        const bool unused = slot >= 0 && !ready;             // slot outside the image / partition
                        role_push(FQ_OFF, CTL + RC_FQ_TAIL, unused, slot, lane);
                        const int n_un = __popcll(__ballot(unused));
                        if (lane == 0 && n_un) atomicSub(&LDSI(CTL + RC_ALIVE), n_un);
                        chunk_next += (uint32_t)n_new;
                        continue;
                    }
                }
            }
            if (queue_empty && !dry_flagged) {
                if (lane == 0) atomicAdd(&LDSI(CTL + RC_DRY), 1);
                dry_flagged = true;
            }
            // C. idle: done when the global queue is dry for every shader wave and no ray is left in the block
            if (__atomic_load_n(&LDSI(CTL + RC_DRY), __ATOMIC_RELAXED) >= NS && __atomic_load_n(&LDSI(CTL + RC_ALIVE), __ATOMIC_RELAXED) == 0) break;
            if (++idle_polls > PT_ROLE_SPIN_MAX) { atomicOr(&LDSI(CTL + RC_ERROR), 4); break; }
#ifdef PT_ROLES_STATS
            ss_idle++;
#endif
            __builtin_amdgcn_s_sleep(2);
        }
#ifdef PT_ROLES_STATS
        if (lane == 0) {
            atomicAdd(&P.counters[9], (unsigned long long)ss_pass);
            atomicAdd(&P.counters[10], (unsigned long long)ss_got);
            atomicAdd(&P.counters[11], (unsigned long long)ss_idle);
            atomicAdd(&P.counters[12], (unsigned long long)ss_bpass);
            atomicAdd(&P.counters[13], (unsigned long long)ss_bgot);
        }
#endif
    }
    // a wave that gave up tells the host (counters[15]); the launch then reports PT_ERR_DEVICE
    if (lane == 0) {
        const int e = __atomic_load_n(&LDSI(CTL + RC_ERROR), __ATOMIC_RELAXED);
        if (e) atomicOr(&P.counters[15], (unsigned long long)e);
    }
}
#undef LDSI
#undef LDSF

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

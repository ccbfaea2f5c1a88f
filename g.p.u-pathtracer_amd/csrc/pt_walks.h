// pt_walks.h — the closest-hit walks over the item buffer (rows a5-a7 of SURVEY.md §8): stack, while-while,
// unified-step, wide, wide with a postponed leaf.  Included by pt_kernels.h.
#pragma once

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

// Sensitivity experiments (tools/pt_exp_hooks.h, built only by tools/build_variant.sh) splice code into the wide walk's
// node step through this hook; the product build leaves it empty.
#ifndef PT_NODE_STEP_HOOK
#define PT_NODE_STEP_HOOK(sc, a, w)
#endif
// phase vote of the wide walk: a node step runs when  lanes at a node x VOTE_NODE >= lanes at a record x VOTE_REC.
// Round 3 sweep (profiles/r03_vote_ab.txt; 800 k / 6.4 M scene, ms per step): 1:1 9.39-9.48 / 19.98, 2:3 9.36-9.50 / 20.09,
// 1:2 (rounds 1-2) 9.36-9.43 / 20.22, 2:5 9.57-9.62 / 20.35, 1:3 9.53 / 20.42, 1:4 9.68-9.75 / 20.48 — the simple majority
// is as fast as any on the cache-resident scene and 1.2 % ahead on the HBM-resident one, with record steps at 0.52 lane
// use instead of 0.40 (node steps 0.67 instead of 0.75): the time does not follow either figure.
#ifndef PT_WIDE_VOTE_NODE
#define PT_WIDE_VOTE_NODE 1
#define PT_WIDE_VOTE_REC 1
#endif
#ifndef PT_WALK_DECL_HOOK
#define PT_WALK_DECL_HOOK()
#define PT_NODE_END_HOOK(sc, cur)
#define PT_WALK_EXIT_HOOK()
#endif

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
    uint32_t n_ovf = 0;  // pushes that went past the LDS window (read by the instrumented kernels only)
    __device__ __forceinline__ TravStack(int b, TravOverflow<LSTK>& o) : base(b), ovf(o.e), lane_base(b + pt_lane_fresh()) {}
    __device__ __forceinline__ void put(int sp, int v) {
        if (LSTK >= PT_STACK_CAP || sp < LSTK) {
            ((int*)s_dyn)[lane_base + sp * BLOCK] = v;
        } else {
            asm volatile("" : "+v"(v));
            ovf[sp - LSTK] = v;
            n_ovf++;
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
// Wide walk: unified-step over the 4-wide quantised tree (pt_items.h).  A node item tests four
// child boxes (24 v_cvt_f32_ubyte + 12 v_pk_fma + min/max), sorts the hit children by entry
// distance with a 5-exchange network on (distance bits | child number) keys, continues with the
// nearest and pushes the rest far-to-near.  Record items are the exact Moller-Trumbore test of
// the other walks, so a reported hit is bit-identical to theirs; only the set of candidates the
// (outward-rounded) boxes let through differs.
// WOOP: records hold Woop's affine rows (PT_OPT_TRI_TEST 1) instead of v0/e1/e2 — see
//      pt_woop_intersect in pt_math.h; tolerance-class parity (the triangle arithmetic differs).
// the three 16-byte pieces of an item as THREE global_load_dwordx4 issued back to back: every lane-level
// vector-memory access costs the CU's address pipe the same whatever its width (tools/ubench_ta), and left alone
// hipcc splits a piece whose .w is used apart from its .xyz into dwordx3 + dword (two accesses).  ONE asm pins all
// twelve dwords after the three loads are in flight (a pin per piece would put a wait behind every load).
__device__ __forceinline__ void pt_ld4x3(const float4* __restrict__ p, float4& q0, float4& q1, float4& q2) {
    q0 = p[0]; q1 = p[1]; q2 = p[2];
    asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q0.w), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w),
                      "+v"(q2.x), "+v"(q2.y), "+v"(q2.z), "+v"(q2.w));
}

struct WideNode {
    float ox, oy, oz, sx, sy, sz;            // origin, grid step per axis
    uint32_t qlx, qly, qlz, qhx, qhy, qhz;   // lo / hi planes, child k in byte k
    int l0, l1, l2, l3;                      // links: >= 0 float4 index of a node, < 0 ~(float4 index of a leaf's first record | min(records, 4) - 1)
};

__device__ __forceinline__ WideNode wide_node_load(const KScene& sc, int a) {
    WideNode w;
    const float4 q0 = sc.nodes[a + 0], q1 = sc.nodes[a + 1], q2 = sc.nodes[a + 2], q3 = sc.nodes[a + 3];
    w.ox = q0.x; w.oy = q0.y; w.oz = q0.z;
    w.sx = q0.w; w.sy = q3.z; w.sz = q3.w;
    w.qlx = __float_as_uint(q1.x); w.qly = __float_as_uint(q1.y); w.qlz = __float_as_uint(q1.z);
    w.qhx = __float_as_uint(q1.w); w.qhy = __float_as_uint(q2.x); w.qhz = __float_as_uint(q2.y);
    w.l0 = __float_as_int(q2.z); w.l1 = __float_as_int(q2.w);
    w.l2 = __float_as_int(q3.x); w.l3 = __float_as_int(q3.y);
    // keep the link words with the box words: left alone, hipcc sinks their use into the hit branches
    // (two dependent round trips per node)
    asm volatile("" : "+v"(w.l2), "+v"(w.l3));
    return w;
}

// link of child (kk & 3)
__device__ __forceinline__ int wide_link(const WideNode& w, uint32_t kk) {
    const uint32_t k = kk & 3u;
    return k == 0u ? w.l0 : (k == 1u ? w.l1 : (k == 2u ? w.l2 : w.l3));
}

// the four child-box tests of a node: keys = (entry distance bits | child number), 0xffffffff = missed; sorted ascending
__device__ __forceinline__ void wide_node_keys(const WideNode& w, float idx, float idy, float idz, float oodx, float oody, float oodz,
                                               float t_max, uint32_t key[4]) {
    const float sx = w.sx * idx, sy = w.sy * idy, sz = w.sz * idz;  // per-axis grid step / direction
    const float bx = fmaf(w.ox, idx, -oodx), by = fmaf(w.oy, idy, -oody), bz = fmaf(w.oz, idz, -oodz);
    // entry/exit planes per axis follow the sign of the ray direction, so pick the packed
    // byte quadruples ONCE per node (6 v_cndmask) instead of min/max per child (24):
    // identical values to min(lo,hi)/max(lo,hi) of the reference's slab test
    const bool px = idx >= 0.0f, py = idy >= 0.0f, pz = idz >= 0.0f;
    const uint32_t nx = px ? w.qlx : w.qhx, fx = px ? w.qhx : w.qlx;
    const uint32_t ny = py ? w.qly : w.qhy, fy = py ? w.qhy : w.qly;
    const uint32_t nz = pz ? w.qlz : w.qhz, fz = pz ? w.qhz : w.qlz;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // (entry, exit) of one axis in one v_pk_fma_f32
        const pt_f2 tx = pt_fma2(pt_mk2((float)((nx >> (8 * k)) & 0xffu), (float)((fx >> (8 * k)) & 0xffu)), pt_mk2(sx, sx), pt_mk2(bx, bx));
        const pt_f2 ty = pt_fma2(pt_mk2((float)((ny >> (8 * k)) & 0xffu), (float)((fy >> (8 * k)) & 0xffu)), pt_mk2(sy, sy), pt_mk2(by, by));
        const pt_f2 tz = pt_fma2(pt_mk2((float)((nz >> (8 * k)) & 0xffu), (float)((fz >> (8 * k)) & 0xffu)), pt_mk2(sz, sz), pt_mk2(bz, bz));
        const float tmin = fmaxf(fmaxf(fmaxf(tx.x, ty.x), tz.x), 0.0f);
        const float tmax = fminf(fminf(fminf(tx.y, ty.y), tz.y), t_max);
        const bool hit = tmin <= tmax;  // unused slots hold inverted boxes
        key[k] = hit ? ((__float_as_uint(tmin) & 0x7ffffffcu) | (uint32_t)k) : 0xffffffffu;
    }
    // sorting network for 4 keys: (0,1)(2,3)(0,2)(1,3)(1,2)
#define PT_CE(i, j) { const uint32_t lo_ = min(key[i], key[j]), hi_ = max(key[i], key[j]); key[i] = lo_; key[j] = hi_; }
    PT_CE(0, 1) PT_CE(2, 3) PT_CE(0, 2) PT_CE(1, 3) PT_CE(1, 2)
#undef PT_CE
}

// AHEAD: request a leaf's second record together with its first (see the record step); costs 12 VGPRs while the
// records are in flight, so only the kernel whose lanes carry nothing but the walk (k_wf_extend) asks for it.
template <bool COUNT, bool DYN, bool WOOP, bool AHEAD = false, class STK>
__device__ __forceinline__ bool trav_run_wide(TravState& s, const KScene& sc, v3 o, v3 d, bool cull, STK& stk,
                                              TravCount& tc, int n_dead, int batch) {
    int cur = s.node, sp = s.sp;
    Hit h = s.h;
    const float idx = s.idx, idy = s.idy, idz = s.idz, oodx = s.oodx, oody = s.oody, oodz = s.oodz;
    PT_WALK_DECL_HOOK();
    for (;;) {
        // phase vote: the wave runs ONE kind of step per iteration, the kind most live lanes are
        // waiting for; the others sit this iteration out.  Node and record lanes no longer both
        // pay for each other's code every iteration.
        // The loop is wave-uniform: every lane that entered stays until the common exit.
        const bool live = cur != PT_SENTINEL;
        const bool is_node = live && cur >= 0;
        const int n_live = __popcll(__ballot(live));
        if (n_live == 0) break;
        if (DYN && 64 - n_live - n_dead >= batch) break;  // enough lanes wait for service
        const int n_node = __popcll(__ballot(is_node));
        // the kind most live lanes are waiting for (weights: see PT_WIDE_VOTE_NODE)
        const bool node_phase = PT_WIDE_VOTE_NODE * n_node >= PT_WIDE_VOTE_REC * (n_live - n_node);
        if (COUNT && pt_first_active_lane()) {  // one lane of those in the walk books the wave's iteration
            if (node_phase) { tc.it_node++; tc.act_node += n_node; }
            else { tc.it_rec++; tc.act_rec += n_live - n_node; }
        }
        if (!live || is_node != node_phase) continue;
        // a leaf link = ~(record index | records - 1 (capped at 3)): wide nodes only (pt_scene_build.h)
        const int a = cur >= 0 ? cur : (~cur & ~3);
        if (cur >= 0) {
            const WideNode w = wide_node_load(sc, a);
            if (COUNT) tc.inner++;
            PT_NODE_STEP_HOOK(sc, a, w);
            uint32_t key[4];
            wide_node_keys(w, idx, idy, idz, oodx, oody, oodz, h.t, key);
            if (key[3] != 0xffffffffu) { sp++; stk.put(sp, wide_link(w, key[3])); }
            if (key[2] != 0xffffffffu) { sp++; stk.put(sp, wide_link(w, key[2])); }
            if (key[1] != 0xffffffffu) {
                const int lk = wide_link(w, key[1]);
                sp++;
                stk.put(sp, lk);
            }
            if (key[0] != 0xffffffffu) {
                cur = wide_link(w, key[0]);
            } else {
                cur = stk.get(sp);
                sp--;
            }
            if (COUNT && cur < 0) tc.leaves++;
            PT_NODE_END_HOOK(sc, cur);
        } else {
            float4 q0, q1, q2;
            // Two thirds of the leaves hold two records (PT_OPT_LEAF_MAX 2), and the link says so: the second one is
            // requested TOGETHER with the first instead of after its test — one dependent round trip less per such leaf.
            // Extend stage -1.5 % on the uploaded tree, -1.2 % on the re-clustered one.
            float4 x0, x1, x2;
            if (!WOOP && AHEAD) {
                const float4* p_ = sc.nodes + a;
                q0 = p_[0]; q1 = p_[1]; q2 = p_[2];
                x0 = x1 = x2 = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((~cur & 3) != 0) { x0 = p_[4]; x1 = p_[5]; x2 = p_[6]; }
                asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q0.w), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w),
                                  "+v"(q2.x), "+v"(q2.y), "+v"(q2.z), "+v"(q2.w), "+v"(x0.x), "+v"(x0.y), "+v"(x0.z), "+v"(x0.w),
                                  "+v"(x1.x), "+v"(x1.y), "+v"(x1.z), "+v"(x1.w), "+v"(x2.x), "+v"(x2.y), "+v"(x2.z), "+v"(x2.w));
            } else
            pt_ld4x3(sc.nodes + a, q0, q1, q2);
            if (COUNT) tc.tris++;
            float t;
            int id;
            bool last;
            if (WOOP) {
                const float4 qw = sc.nodes[a + 3];   // a record's 4th piece: normal | id << 1 | last
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
                if (AHEAD && !last) {   // the record requested ahead
                    aa += 4;
                    if (COUNT) tc.tris++;
                    const float t2 = pt_mt_intersect(V3(x0.x, x0.y, x0.z), V3(x1.x, x1.y, x1.z), V3(x2.x, x2.y, x2.z), o, d, cull);
                    const int id2 = __float_as_int(x0.w);
                    last = __float_as_int(x1.w) != 0;
                    if (t2 > 0.0f && (t2 < h.t || (t2 == h.t && h.tri != -1 && id2 < h.tri))) {
                        h.t = t2;
                        h.tri = id2;
                        h.rec = aa;
                    }
                }
                while (!last) {
                    aa += 4;
                    float4 r0, r1, r2;
                    pt_ld4x3(sc.nodes + aa, r0, r1, r2);
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
    PT_WALK_EXIT_HOOK();
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
            const WideNode w = wide_node_load(sc, cur);
            if (COUNT) tc.inner++;
            uint32_t key[4];
            wide_node_keys(w, idx, idy, idz, oodx, oody, oodz, h.t, key);
            if (key[3] != 0xffffffffu) { sp++; stk.put(sp, wide_link(w, key[3])); }
            if (key[2] != 0xffffffffu) { sp++; stk.put(sp, wide_link(w, key[2])); }
            if (key[1] != 0xffffffffu) { sp++; stk.put(sp, wide_link(w, key[1])); }
            if (key[0] != 0xffffffffu) {
                cur = wide_link(w, key[0]);
            } else {
                cur = stk.get(sp);
                sp--;
            }
            if (COUNT && cur < 0) tc.leaves++;
            if (cur < 0 && pend == 0) {  // park the leaf, go on with the next entry
                pend = cur;
                cur = stk.get(sp);
                sp--;
                if (COUNT && cur < 0) tc.leaves++;
            }
        } else {
            if (!has_rec) continue;
            const int a = ~pend & ~3;
            float4 q0, q1, q2;
            pt_ld4x3(sc.nodes + a, q0, q1, q2);
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


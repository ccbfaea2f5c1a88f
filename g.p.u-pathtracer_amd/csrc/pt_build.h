// pt_build.h — BVH construction ON the device (pt_build_bvh): the step in front
// of the hot path.  The reference builds on the host (SplitBVHBuilder.cpp, 1.7 s per 100 k
// triangles) and so does host/pthost.cpp (SAH/SBVH, ~1.4 s for 800 k; here 1.8 ms); this is the fast
// alternative for scenes that change: Morton order, then PLOC clustering (default) or a linear BVH
//   1. k_tri_bounds   triangle boxes + bounds of the box centres (ordered-int atomics)
//   2. k_morton       63-bit Morton key of every centre (21 bits per axis)
//   3. hipcub radix sort of (key, triangle)
//   4. k_hierarchy    Karras 2012: every inner node finds its key range and split in parallel;
//                     equal keys are told apart by their position, so the tree stays a tree
//   5. k_node_depth + k_fit_level   boxes bottom-up, one launch per tree level (no device fences)
//   6. k_records      the 64-byte triangle records in sorted order; a subtree with <= leaf_max
//                     triangles is ONE leaf (its records are contiguous), the last one flagged
//   7. k_binary       the binary nodes in the Compact layout (walks 0/1, pt_trace_rays)
//   8. k_collapse     level by level: 4-wide nodes (largest-area inner child opened first) with
//                     8-bit outward-rounded boxes — pt_encode_wide_node, the host path's encoder
// The product is the same item buffer pt_upload_bvh makes ([binary nodes][records][wide nodes]),
// so every kernel and every parity property (exact triangle test, ties to the smaller id: the
// closest hit does not depend on the tree) carries over; only the tree QUALITY differs (no SAH).
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "pt_items.h"

#define PTB_BLOCK 256

struct BuildArrays {
    // inputs
    const float* verts;      // [n_verts][3]
    const int* tris;         // [n][3]
    int n;                   // triangles (>= 2)
    int n_orig;              // caller's triangle count (1 when the lone triangle was doubled to get a tree)
    const int* id_map;       // optional: triangle id reported for row t (NULL = t itself)
    const int* ref_tri;      // optional (pre-splitting): primitive -> triangle row; the primitive's box is then
                             // one slab of the triangle's box and `tbox` is filled by k_split_emit
    int leaf_max;            // subtree size that becomes one leaf (>= 1)
    // per triangle (unsorted): box
    float* tbox;             // [n][6] lo xyz, hi xyz
    unsigned int* cbounds;   // [6] ordered-int min xyz / max xyz of the box centres
    // sorted order
    unsigned long long* key_in; unsigned long long* key; // [n]
    int* val_in; int* val;                               // [n] triangle id
    // hierarchy (inner node i in [0, n-1), leaf j in [0, n))
    int* left; int* right;   // child: >= 0 inner node, < 0 ~leaf position
    int* first; int* last;   // key range of the inner node
    int* parent_i;           // parent of inner node (root: -1)
    int* parent_l;           // parent of leaf position
    float* nbox;             // [n-1][6] box of the inner node
    unsigned int* arrive;    // [n-1] depth of the inner node (level of the bottom-up fit)
    // outputs
    float4* items;           // [binary nodes n-1][records n][wide nodes <= n-1]
    unsigned int* stats;     // [0] wide nodes allocated [1] - [2] leaves [3] max binary depth
    unsigned int* level_cnt; // [66] frontier size of every wide level (level 0 = 1: the root)
    int2* frontier_a; int2* frontier_b;   // (inner node, wide slot)
};

__device__ __forceinline__ unsigned int ptb_ordered(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ptb_unordered(unsigned int u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// ---- pre-splitting (early split clipping, Ernst & Greiner 2007) -----------------------------------
// A triangle much longer than `target` along its longest axis is entered as up to PTB_SPLIT_MAX
// primitives, one per slab of its box, each with the (padded) bounds of the part of the triangle
// inside the slab: what a spatial-split builder gets from splitting references, decided up front.
// All primitives of a triangle point at the same record data, so hits and ids do not change.
#ifndef PTB_SPLIT_MAX
#define PTB_SPLIT_MAX 8
#endif
__device__ __forceinline__ int ptb_split_count(const float* v0, const float* v1, const float* v2, float target, int& axis) {
    float ext[3];
    for (int a = 0; a < 3; a++) ext[a] = fmaxf(v0[a], fmaxf(v1[a], v2[a])) - fminf(v0[a], fminf(v1[a], v2[a]));
    axis = ext[0] >= ext[1] ? (ext[0] >= ext[2] ? 0 : 2) : (ext[1] >= ext[2] ? 1 : 2);
    if (!(ext[axis] > target) || !(target > 0.f)) return 1;
    // a triangle that would still dwarf its neighbours after PTB_SPLIT_MAX cuts (a room's wall next to a
    // fine mesh) is better left whole at the top of the tree: eight wall-sized slabs sink into the fine
    // clusters and bloat them (cornell_dragon 800 k: 5.57 -> 6.9 ms when the walls were cut)
    if (ext[axis] > 4.f * (float)PTB_SPLIT_MAX * target) return 1;
    return min(PTB_SPLIT_MAX, (int)ceilf(ext[axis] / target));
}

__global__ void __launch_bounds__(PTB_BLOCK) k_split_count(const float* __restrict__ verts, const int* __restrict__ tris, int n_tris, float target,
                                                           int* __restrict__ cnt) {
    const int i = blockIdx.x * PTB_BLOCK + threadIdx.x;
    if (i >= n_tris) return;
    int axis;
    cnt[i] = ptb_split_count(verts + 3 * (size_t)tris[3 * i], verts + 3 * (size_t)tris[3 * i + 1], verts + 3 * (size_t)tris[3 * i + 2], target, axis);
}

__global__ void __launch_bounds__(PTB_BLOCK) k_split_emit(const float* __restrict__ verts, const int* __restrict__ tris, int n_tris, float target,
                                                          const int* __restrict__ off, float* __restrict__ tbox, int* __restrict__ ref_tri) {
    const int i = blockIdx.x * PTB_BLOCK + threadIdx.x;
    if (i >= n_tris) return;
    const float* v[3] = {verts + 3 * (size_t)tris[3 * i], verts + 3 * (size_t)tris[3 * i + 1], verts + 3 * (size_t)tris[3 * i + 2]};
    int ax;
    const int k = ptb_split_count(v[0], v[1], v[2], target, ax);
    float lo[3], hi[3];
    for (int a = 0; a < 3; a++) { lo[a] = fminf(v[0][a], fminf(v[1][a], v[2][a])); hi[a] = fmaxf(v[0][a], fmaxf(v[1][a], v[2][a])); }
    const int base = off[i];
    for (int s = 0; s < k; s++) {
        float blo[3] = {lo[0], lo[1], lo[2]}, bhi[3] = {hi[0], hi[1], hi[2]};
        if (k > 1) {
            const float x0 = s == 0 ? lo[ax] : lo[ax] + (hi[ax] - lo[ax]) * ((float)s / (float)k);
            const float x1 = s == k - 1 ? hi[ax] : lo[ax] + (hi[ax] - lo[ax]) * ((float)(s + 1) / (float)k);
            // bounds of triangle ∩ slab: vertices inside the slab + edge crossings of its two planes
            float plo[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, phi[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
            for (int e = 0; e < 3; e++) {
                const float* a = v[e];
                const float* b = v[(e + 1) % 3];
                if (a[ax] >= x0 && a[ax] <= x1)
                    for (int c = 0; c < 3; c++) { plo[c] = fminf(plo[c], a[c]); phi[c] = fmaxf(phi[c], a[c]); }
                for (int pl = 0; pl < 2; pl++) {
                    const float x = pl ? x1 : x0;
                    if ((a[ax] < x && b[ax] > x) || (a[ax] > x && b[ax] < x)) {
                        const float t = (x - a[ax]) / (b[ax] - a[ax]);
                        for (int c = 0; c < 3; c++) {
                            const float pc = c == ax ? x : fmaf(t, b[c] - a[c], a[c]);
                            plo[c] = fminf(plo[c], pc); phi[c] = fmaxf(phi[c], pc);
                        }
                    }
                }
            }
            for (int c = 0; c < 3; c++) {
                // padded (the crossings are rounded), never beyond the triangle's own box or the slab
                const float pad = 1e-5f * fmaxf(hi[c] - lo[c], fmaxf(fabsf(lo[c]), fabsf(hi[c]))) + 1e-30f;
                blo[c] = fmaxf(lo[c], plo[c] - pad);
                bhi[c] = fminf(hi[c], phi[c] + pad);
                if (!(blo[c] <= bhi[c])) { blo[c] = lo[c]; bhi[c] = hi[c]; }   // nothing found (degenerate): the whole box
            }
            const float padx = 1e-5f * fmaxf(hi[ax] - lo[ax], fmaxf(fabsf(lo[ax]), fabsf(hi[ax]))) + 1e-30f;
            blo[ax] = fmaxf(blo[ax], fmaxf(lo[ax], x0 - padx));
            bhi[ax] = fminf(bhi[ax], fminf(hi[ax], x1 + padx));
            if (!(blo[ax] <= bhi[ax])) { blo[ax] = fmaxf(lo[ax], x0 - padx); bhi[ax] = fminf(hi[ax], x1 + padx); }
        }
        for (int c = 0; c < 3; c++) { tbox[6 * (size_t)(base + s) + c] = blo[c]; tbox[6 * (size_t)(base + s) + 3 + c] = bhi[c]; }
        ref_tri[base + s] = i;
    }
}

__global__ void __launch_bounds__(PTB_BLOCK) k_tri_bounds(const BuildArrays B) {
    const int i = blockIdx.x * PTB_BLOCK + threadIdx.x;
    float c[3] = {0.f, 0.f, 0.f};
    const bool live = i < B.n;
    if (live && B.ref_tri) {
        for (int a = 0; a < 3; a++) c[a] = 0.5f * B.tbox[6 * (size_t)i + a] + 0.5f * B.tbox[6 * (size_t)i + 3 + a];
    } else if (live) {
        const int i0 = B.tris[3 * i], i1 = B.tris[3 * i + 1], i2 = B.tris[3 * i + 2];
        float lo[3], hi[3];
        for (int a = 0; a < 3; a++) {
            const float x = B.verts[3 * (size_t)i0 + a], y = B.verts[3 * (size_t)i1 + a], z = B.verts[3 * (size_t)i2 + a];
            lo[a] = fminf(x, fminf(y, z));
            hi[a] = fmaxf(x, fmaxf(y, z));
            c[a] = 0.5f * lo[a] + 0.5f * hi[a];
            B.tbox[6 * (size_t)i + a] = lo[a];
            B.tbox[6 * (size_t)i + 3 + a] = hi[a];
        }
    }
    // wave reduction, block reduction through LDS, then ONE atomic per block and component (one per
    // wave = 75 000 atomics on six addresses = 0.86 ms of serialised L2 atomics for 800 k triangles)
    __shared__ unsigned int s_red[PTB_BLOCK / 64][6];
    for (int a = 0; a < 3; a++) {
        unsigned int mn = live ? ptb_ordered(c[a]) : 0xffffffffu, mx = live ? ptb_ordered(c[a]) : 0u;
        for (int off = 32; off > 0; off >>= 1) {
            mn = min(mn, (unsigned int)__shfl_xor((int)mn, off));
            mx = max(mx, (unsigned int)__shfl_xor((int)mx, off));
        }
        if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6][a] = mn; s_red[threadIdx.x >> 6][3 + a] = mx; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        unsigned int v = s_red[0][threadIdx.x];
        for (int w = 1; w < PTB_BLOCK / 64; w++) v = threadIdx.x < 3 ? min(v, s_red[w][threadIdx.x]) : max(v, s_red[w][threadIdx.x]);
        if (threadIdx.x < 3) atomicMin(&B.cbounds[threadIdx.x], v); else atomicMax(&B.cbounds[threadIdx.x], v);
    }
}

__device__ __forceinline__ unsigned long long ptb_spread21(unsigned int v) {  // bit i -> bit 3i
    unsigned long long x = v & 0x1fffffu;
    x = (x | (x << 32)) & 0x1f00000000ffffull;
    x = (x | (x << 16)) & 0x1f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

// Sort key.  Plain Morton order (PTB_SIZE_BITS 0): 21 bits per axis of the box centre.  Extended
// Morton codes (Vinkler et al. 2017): every 7th bit is a bit of the box DIAGONAL, so that triangles
// of very different size (32 wall triangles next to 800 000 dragon triangles) separate high in
// the tree instead of dragging huge boxes down into fine clusters: 18 bits per axis + 9 size bits.
#ifndef PTB_SIZE_BITS
#define PTB_SIZE_BITS 9
#endif
__global__ void __launch_bounds__(PTB_BLOCK) k_morton(const BuildArrays B) {
    const int i = blockIdx.x * PTB_BLOCK + threadIdx.x;
    if (i >= B.n) return;
    float u[3], diag2 = 0.f, sdiag2 = 0.f;
    for (int a = 0; a < 3; a++) {
        const float lo = ptb_unordered(B.cbounds[a]), hi = ptb_unordered(B.cbounds[3 + a]);
        const float blo = B.tbox[6 * (size_t)i + a], bhi = B.tbox[6 * (size_t)i + 3 + a];
        const float c = 0.5f * blo + 0.5f * bhi;
        const float ext = hi - lo;
        u[a] = ext > 0.f ? fminf(fmaxf((c - lo) / ext, 0.f), 1.f) : 0.f;
        diag2 += (bhi - blo) * (bhi - blo);
        sdiag2 += ext * ext;
    }
    unsigned long long key;
    if (PTB_SIZE_BITS == 0) {
        unsigned int q[3];
        for (int a = 0; a < 3; a++) q[a] = min((unsigned int)(u[a] * 2097152.0f), 2097151u);
        key = (ptb_spread21(q[0]) << 2) | (ptb_spread21(q[1]) << 1) | ptb_spread21(q[2]);
    } else {
        const int AB = (63 - PTB_SIZE_BITS) / 3;                       // bits per axis
        unsigned int q[3];
        for (int a = 0; a < 3; a++) q[a] = min((unsigned int)(u[a] * (float)(1u << AB)), (1u << AB) - 1u);
        const float rel = sdiag2 > 0.f ? sqrtf(diag2 / sdiag2) : 0.f;   // box diagonal / scene diagonal, 0..~1
        const unsigned int qs = min((unsigned int)(fminf(rel, 1.f) * (float)(1u << PTB_SIZE_BITS)), (1u << PTB_SIZE_BITS) - 1u);
        // most significant first: x y z x y z s, x y z x y z s, ...
        key = 0;
        int bx = AB - 1, bs = PTB_SIZE_BITS - 1, phase = 0;
        int ax = 0;
        for (int out = 0; out < 3 * AB + PTB_SIZE_BITS; out++) {
            unsigned int bit;
            if (phase == 6 && bs >= 0) { bit = (qs >> bs) & 1u; bs--; phase = 0; }
            else {
                bit = (q[ax] >> bx) & 1u;
                ax++;
                if (ax == 3) { ax = 0; bx--; }
                phase++;
                if (bx < 0) phase = 6;   // axes exhausted: the rest are size bits
            }
            key = (key << 1) | bit;
        }
    }
    B.key_in[i] = key;
    B.val_in[i] = i;
}

// common-prefix length of sorted positions i and j (-1 outside the array); equal keys are
// separated by the position itself (Karras 2012, section 4)
__device__ __forceinline__ int ptb_delta(const unsigned long long* key, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long x = key[i] ^ key[j];
    if (x == 0) return 64 + __clz((unsigned int)(i ^ j));
    return __clzll((long long)x);
}

__global__ void __launch_bounds__(PTB_BLOCK) k_hierarchy(const BuildArrays B) {
    const int i = blockIdx.x * PTB_BLOCK + threadIdx.x;
    const int n = B.n;
    if (i >= n - 1) return;
    const unsigned long long* key = B.key;
    const int d = ptb_delta(key, n, i, i + 1) - ptb_delta(key, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = ptb_delta(key, n, i, i - d);
    int lmax = 2;
    while (ptb_delta(key, n, i, i + lmax * d) > dmin) lmax <<= 1;   // lmax <= 2n: i + lmax*d leaves [0,n) first
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (ptb_delta(key, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = ptb_delta(key, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (ptb_delta(key, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int cl = lo == gamma ? ~gamma : gamma;
    const int cr = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    B.left[i] = cl; B.right[i] = cr;
    B.first[i] = lo; B.last[i] = hi;
    if (cl >= 0) B.parent_i[cl] = i; else B.parent_l[~cl] = i;
    if (cr >= 0) B.parent_i[cr] = i; else B.parent_l[~cr] = i;
    if (i == 0) B.parent_i[0] = -1;
}

__device__ __forceinline__ void ptb_child_box(const BuildArrays& B, int c, float* bx) {
    const float* src = c >= 0 ? B.nbox + 6 * (size_t)c : B.tbox + 6 * (size_t)B.val[~c];
    for (int a = 0; a < 6; a++) bx[a] = src[a];
}

// Depth of every inner node (edges to the root) by walking the parent chain, and the deepest one.
__global__ void __launch_bounds__(PTB_BLOCK) k_node_depth(const BuildArrays B) {
    const int i = blockIdx.x * PTB_BLOCK + threadIdx.x;
    unsigned int depth = 0;
    if (i < B.n - 1) {
        for (int cur = B.parent_i[i]; cur >= 0; cur = B.parent_i[cur]) depth++;
        B.arrive[i] = depth;
    }
    for (int off = 32; off > 0; off >>= 1) depth = max(depth, (unsigned int)__shfl_xor((int)depth, off));
    if ((threadIdx.x & 63) == 0) atomicMax(&B.stats[3], depth);
}

// Boxes bottom-up, ONE LEVEL PER LAUNCH: the inner nodes of depth `level` merge their children,
// which are leaves or nodes of depth level + 1 finished by the previous launch.  (The classic
// single-launch version — second arrival at a node merges, a device-scope fence on either side of
// the counter — spends 5.3 ms of an 8 ms build of 800 k triangles in those fences: on this part
// they write back / invalidate an XCD's whole L2.  ~40 launches of a trivial kernel take 0.7 ms.)
__global__ void __launch_bounds__(PTB_BLOCK) k_fit_level(const BuildArrays B, unsigned int level) {
    const int i = blockIdx.x * PTB_BLOCK + threadIdx.x;
    if (i >= B.n - 1 || B.arrive[i] != level) return;
    float a[6], b[6];
    ptb_child_box(B, B.left[i], a);
    ptb_child_box(B, B.right[i], b);
    float* dst = B.nbox + 6 * (size_t)i;
    for (int k = 0; k < 3; k++) {
        dst[k] = fminf(a[k], b[k]);
        dst[3 + k] = fmaxf(a[k + 3], b[k + 3]);
    }
}

__device__ __forceinline__ float ptb_area(const float* b) {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return 2.f * (dx * dy + dy * dz + dz * dx);
}

// ---- PLOC (Meister & Bittner 2018), the better-quality alternative to steps 4-5 ------------------
// Clusters start as the leaves in Morton order.  Every round each cluster looks PTB_PLOC_RADIUS places
// to either side for the neighbour whose union with it has the smallest surface; pairs that chose
// each other merge into a new inner node; the survivors are compacted in order.  Boxes are known at
// merge time (no fit pass); node numbers are handed out from the top down so that the last merge —
// the root — is node 0, where the walks expect it.  Subtrees are no longer contiguous in Morton
// order: the leaves are re-numbered depth-first afterwards (k_ploc_size / k_ploc_first / k_ploc_reorder), which makes
// every subtree a contiguous record range again, so the cut rule (PT_OPT_LEAF_MAX) applies as in the LBVH.
#ifndef PTB_PLOC_RADIUS
#define PTB_PLOC_RADIUS 8
#endif
struct PlocArrays {
    int* cl;            // [n_c] cluster = child reference (>= 0 inner node, < 0 ~leaf position)
    int* cl_next;       // [n_c] compacted survivors of this round
    float* cbox;        // [n_c][6] boxes of the clusters, gathered once per round
    int* nn;            // [n_c] chosen neighbour
    int* keep;          // [n_c] 1 = survives the round (itself or as the merged node)
    int* pos;           // [n_c] exclusive prefix sum of keep
    int* ref;           // [n_c] what the cluster is after the round
    int n_c;
};

__global__ void __launch_bounds__(PTB_BLOCK) k_ploc_gather(const BuildArrays B, const PlocArrays Q) {
    const int i = blockIdx.x * PTB_BLOCK + threadIdx.x;
    if (i >= Q.n_c) return;
    float bx[6];
    ptb_child_box(B, Q.cl[i], bx);
    for (int a = 0; a < 6; a++) Q.cbox[6 * (size_t)i + a] = bx[a];
}

__global__ void __launch_bounds__(PTB_BLOCK) k_ploc_nn(const PlocArrays Q) {
    const int i = blockIdx.x * PTB_BLOCK + threadIdx.x;
    if (i >= Q.n_c) return;
    float a[6];
    for (int k = 0; k < 6; k++) a[k] = Q.cbox[6 * (size_t)i + k];
    int best = -1;
    float best_area = 3.402823466e+38f;
    const int lo = max(0, i - PTB_PLOC_RADIUS), hi = min(Q.n_c - 1, i + PTB_PLOC_RADIUS);
    for (int j = lo; j <= hi; j++) {
        if (j == i) continue;
        const float* b = Q.cbox + 6 * (size_t)j;
        const float u[6] = {fminf(a[0], b[0]), fminf(a[1], b[1]), fminf(a[2], b[2]), fmaxf(a[3], b[3]), fmaxf(a[4], b[4]), fmaxf(a[5], b[5])};
        const float ar = ptb_area(u);
        if (ar < best_area) { best_area = ar; best = j; }   // ties: the smaller position (deterministic)
    }
    Q.nn[i] = best;
}

__global__ void __launch_bounds__(PTB_BLOCK) k_ploc_merge(const BuildArrays B, const PlocArrays Q) {
    const int i = blockIdx.x * PTB_BLOCK + threadIdx.x;
    if (i >= Q.n_c) return;
    const int j = Q.nn[i];
    int ref = Q.cl[i], keep = 1;
    if (j >= 0 && Q.nn[j] == i) {
        if (i < j) {
            const int id = (B.n - 2) - (int)atomicAdd(&B.stats[1], 1u);   // the last merge of all gets 0: the root
            B.left[id] = Q.cl[i];
            B.right[id] = Q.cl[j];
            float* dst = B.nbox + 6 * (size_t)id;
            for (int k = 0; k < 3; k++) {
                dst[k] = fminf(Q.cbox[6 * (size_t)i + k], Q.cbox[6 * (size_t)j + k]);
                dst[3 + k] = fmaxf(Q.cbox[6 * (size_t)i + 3 + k], Q.cbox[6 * (size_t)j + 3 + k]);
            }
            ref = id;
        } else {
            keep = 0;
        }
    }
    Q.ref[i] = ref;
    Q.keep[i] = keep;
}

__global__ void __launch_bounds__(PTB_BLOCK) k_ploc_scatter(const PlocArrays Q) {
    const int i = blockIdx.x * PTB_BLOCK + threadIdx.x;
    if (i >= Q.n_c) return;
    if (Q.keep[i]) Q.cl_next[Q.pos[i]] = Q.ref[i];
}

// parent links (and the empty key ranges that switch the leaf cut off) once the tree is complete
__global__ void __launch_bounds__(PTB_BLOCK) k_ploc_parents(const BuildArrays B) {
    const int i = blockIdx.x * PTB_BLOCK + threadIdx.x;
    if (i >= B.n - 1) return;
    const int cl = B.left[i], cr = B.right[i];
    if (cl >= 0) B.parent_i[cl] = i; else B.parent_l[~cl] = i;
    if (cr >= 0) B.parent_i[cr] = i; else B.parent_l[~cr] = i;
    if (i == 0) B.parent_i[0] = -1;
    B.first[i] = 0; B.last[i] = 0;
}

// Depth-first leaf order for a PLOC tree, so that every subtree covers a contiguous range of records
// again and small subtrees can be cut into multi-triangle leaves like the LBVH's:
//   k_ploc_size   (deepest level first)  size[node] = leaves below it, kept in `last`
//   k_ploc_first  (root level first)     first[node]; a leaf child gets its new position
//   k_ploc_reorder                       triangle ids move to the new positions, leaf references follow
__global__ void __launch_bounds__(PTB_BLOCK) k_ploc_size(const BuildArrays B, unsigned int level) {
    const int i = blockIdx.x * PTB_BLOCK + threadIdx.x;
    if (i >= B.n - 1 || B.arrive[i] != level) return;
    const int cl = B.left[i], cr = B.right[i];
    B.last[i] = (cl >= 0 ? B.last[cl] : 1) + (cr >= 0 ? B.last[cr] : 1);
}

__global__ void __launch_bounds__(PTB_BLOCK) k_ploc_first(const BuildArrays B, unsigned int level, int* __restrict__ newpos) {
    const int i = blockIdx.x * PTB_BLOCK + threadIdx.x;
    if (i >= B.n - 1 || B.arrive[i] != level) return;
    const int f = level == 0 ? 0 : B.first[i];
    if (level == 0) B.first[i] = 0;
    const int cl = B.left[i], cr = B.right[i];
    const int nl = cl >= 0 ? B.last[cl] : 1;
    if (cl >= 0) B.first[cl] = f; else newpos[~cl] = f;
    if (cr >= 0) B.first[cr] = f + nl; else newpos[~cr] = f + nl;
}

__global__ void __launch_bounds__(PTB_BLOCK) k_ploc_reorder(const BuildArrays B, const int* __restrict__ newpos, const int* __restrict__ val_old) {
    const int i = blockIdx.x * PTB_BLOCK + threadIdx.x;
    if (i < B.n) {
        B.val[newpos[i]] = val_old[i];
    }
    if (i < B.n - 1) {
        const int cl = B.left[i], cr = B.right[i];
        if (cl < 0) { B.left[i] = ~newpos[~cl]; B.parent_l[newpos[~cl]] = i; }
        if (cr < 0) { B.right[i] = ~newpos[~cr]; B.parent_l[newpos[~cr]] = i; }
        B.last[i] = B.first[i] + B.last[i] - 1;   // size -> last position
    }
}

__global__ void __launch_bounds__(PTB_BLOCK) k_ploc_init(const BuildArrays B, const PlocArrays Q) {
    const int i = blockIdx.x * PTB_BLOCK + threadIdx.x;
    if (i < B.n) Q.cl[i] = ~i;
}

// depth of every leaf (edges to the root), for the stack bound
__global__ void __launch_bounds__(PTB_BLOCK) k_depth(const BuildArrays B) {
    const int j = blockIdx.x * PTB_BLOCK + threadIdx.x;
    unsigned int depth = 0;
    if (j < B.n) depth = B.arrive[B.parent_l[j]] + 1u;   // the parent's depth is known (k_node_depth)
    for (int off = 32; off > 0; off >>= 1) depth = max(depth, (unsigned int)__shfl_xor((int)depth, off));
    if ((threadIdx.x & 63) == 0) atomicMax(&B.stats[3], depth);
}

// true when inner node c is cut into ONE leaf (its <= leaf_max records are contiguous)
__device__ __forceinline__ bool ptb_is_cut(const BuildArrays& B, int c) {
    return c > 0 && B.last[c] - B.first[c] + 1 <= B.leaf_max;
}

// link of child c as the walks read it, given where the records start (float4 index)
__device__ __forceinline__ bool ptb_child_is_leaf(const BuildArrays& B, int c, int& first_pos) {
    if (c < 0) { first_pos = ~c; return true; }
    if (ptb_is_cut(B, c)) { first_pos = B.first[c]; return true; }
    return false;
}

__global__ void __launch_bounds__(PTB_BLOCK) k_records(const BuildArrays B) {
    const int j = blockIdx.x * PTB_BLOCK + threadIdx.x;
    if (j >= B.n) return;
    // the leaf this record belongs to: the topmost cut ancestor, or the record alone
    int end = j;
    for (int p = B.parent_l[j]; p > 0 && ptb_is_cut(B, p); p = B.parent_i[p]) end = B.last[p];
    const int last = j == end ? 1 : 0;
    if (last) atomicAdd(&B.stats[2], 1u);
    const int prim = B.val[j];
    const int t = B.ref_tri ? B.ref_tri[prim] : prim;
    const int i0 = B.tris[3 * (size_t)t], i1 = B.tris[3 * (size_t)t + 1], i2 = B.tris[3 * (size_t)t + 2];
    float v0[3], v1[3], v2[3], rec[16];
    for (int a = 0; a < 3; a++) {
        v0[a] = B.verts[3 * (size_t)i0 + a];
        v1[a] = B.verts[3 * (size_t)i1 + a];
        v2[a] = B.verts[3 * (size_t)i2 + a];
    }
    if (v0[0] == 0.f) v0[0] = 0.f;   // -0.0f -> +0.0f, as the Compact producer stores it (host/pthost.cpp)
    const int row = min(t, B.n_orig - 1);
    pt_encode_record(v0, v1, v2, B.id_map ? B.id_map[row] : row, last, rec);
    float4* dst = B.items + 4 * (size_t)(B.n - 1) + 4 * (size_t)j;
    for (int k = 0; k < 4; k++) dst[k] = make_float4(rec[4 * k], rec[4 * k + 1], rec[4 * k + 2], rec[4 * k + 3]);
}

// binary nodes, Compact layout (CudaBVH.cpp:221-224): [c0.lo.x c0.hi.x c0.lo.y c0.hi.y]
// [c1 ...] [c0.lo.z c0.hi.z c1.lo.z c1.hi.z] [link0 link1 0 0]; inner node i at float4 index 4i
__global__ void __launch_bounds__(PTB_BLOCK) k_binary(const BuildArrays B) {
    const int i = blockIdx.x * PTB_BLOCK + threadIdx.x;
    if (i >= B.n - 1) return;
    const int rec_base = 4 * (B.n - 1);
    float d[16];
    const int ch[2] = {B.left[i], B.right[i]};
    for (int k = 0; k < 2; k++) {
        float bx[6];
        ptb_child_box(B, ch[k], bx);
        d[0 + 4 * k] = bx[0]; d[1 + 4 * k] = bx[3];
        d[2 + 4 * k] = bx[1]; d[3 + 4 * k] = bx[4];
        d[8 + 2 * k] = bx[2]; d[9 + 2 * k] = bx[5];
        int fp;
        d[12 + k] = pt_i2f(ptb_child_is_leaf(B, ch[k], fp) ? ~(rec_base + 4 * fp) : 4 * ch[k]);
    }
    d[14] = 0.f; d[15] = 0.f;
    float4* dst = B.items + 4 * (size_t)i;
    for (int k = 0; k < 4; k++) dst[k] = make_float4(d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3]);
}

// One level of the 4-wide tree: every frontier entry (inner node, wide slot) adopts up to four
// descendants — the inner child with the largest box is opened first, as the host path does —
// writes its node and queues its inner children for the next level.
// The frontier sizes live on the device (level_cnt[level]), so the host queues every level without
// reading anything back: a level whose frontier is empty costs one empty launch.
__global__ void __launch_bounds__(PTB_BLOCK) k_collapse(const BuildArrays B, const int2* __restrict__ in, int2* __restrict__ out, int level) {
    const int n_in = (int)B.level_cnt[level];
    for (int idx = blockIdx.x * PTB_BLOCK + threadIdx.x; idx < n_in; idx += gridDim.x * PTB_BLOCK) {
    const int2 item = in[idx];
    const int rec_base = 4 * (B.n - 1);
    const int wide_base = rec_base + 4 * B.n;
    int ref[4];
    PtBox cb[4];
    int cnt = 0;
    auto add = [&](int c) {
        float bx[6];
        ptb_child_box(B, c, bx);
        for (int a = 0; a < 3; a++) { cb[cnt].lo[a] = bx[a]; cb[cnt].hi[a] = bx[3 + a]; }
        ref[cnt] = c;
        cnt++;
    };
    add(B.left[item.x]);
    add(B.right[item.x]);
    while (cnt < 4) {
        int best = -1;
        float ba = -1.f;
        for (int k = 0; k < cnt; k++) {
            int fp;
            if (ptb_child_is_leaf(B, ref[k], fp)) continue;
            const float bx[6] = {cb[k].lo[0], cb[k].lo[1], cb[k].lo[2], cb[k].hi[0], cb[k].hi[1], cb[k].hi[2]};
            const float ar = ptb_area(bx);
            if (ar > ba) { ba = ar; best = k; }
        }
        if (best < 0) break;
        const int v = ref[best];
        cb[best] = cb[cnt - 1];
        ref[best] = ref[cnt - 1];
        cnt--;
        add(B.left[v]);
        add(B.right[v]);
    }
    int32_t link[4] = {0, 0, 0, 0};
    for (int k = 0; k < cnt; k++) {
        int fp;
        if (ptb_child_is_leaf(B, ref[k], fp)) {
            // a wide node's leaf link also says how many records the leaf holds (low two bits: min(count, 4) - 1)
            const int n_rec = ref[k] < 0 ? 1 : B.last[ref[k]] - B.first[ref[k]] + 1;
            link[k] = ~((rec_base + 4 * fp) | (min(n_rec, 4) - 1));
        } else {
            const int slot = (int)atomicAdd(&B.stats[0], 1u);
            link[k] = wide_base + 4 * slot;
            out[atomicAdd(&B.level_cnt[level + 1], 1u)] = make_int2(ref[k], slot);
        }
    }
    float d[16];
    pt_encode_wide_node(cb, cnt, link, d);
    float4* dst = B.items + (size_t)wide_base + 4 * (size_t)item.y;
    for (int k = 0; k < 4; k++) dst[k] = make_float4(d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3]);
    }
}

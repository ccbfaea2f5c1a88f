// pt_scene_build.h — host side of pt_upload_bvh: the reference's Compact arrays
// (CudaBVH::createCompact, GpuPathTracer/CudaBVH.cpp:121-270) → the gfx950 item buffer.
//
//   1. parse + validate (links in range and 64-byte multiples, a tree, leaves terminated,
//      depth <= 64): corrupt arrays are rejected here and never reach a kernel
//   2. refine: a leaf with more than `leaf_max` references is split by a small SAH sweep.  The
//      reference builder's 1:1 node/triangle cost leaves e.g. cornell.obj's 32 wall triangles
//      in 5 leaves that almost every ray tests in full; a triangle record costs the walk more
//      than a box test.  Closest hits do not depend on tree shape.
//   3. emit: triangle records (64 B), binary nodes (64 B, reference record), 4-wide quantised
//      nodes (64 B) — all in ONE buffer indexed by float4:
//          [binary nodes][records][wide nodes]
//      binary/wide nodes: the first PT_MAX_TOP in breadth-first order (LDS mirror prefix), the
//      rest depth-first.
//
//   record: [v0.xyz, id][e1.xyz, last][e2.xyz, 0][cross(v0-v1, v0-v2), 0]
//           e1 = v1-v0, e2 = v2-v0 (cudaUtils.h:177-178) and the normal (:432) hoisted to upload
//           with the kernels' own arithmetic; the id rides in v0.w (replaces the gpuTriIndices
//           remap :452-456), `last` replaces the 0x80000000 terminator fetch (:410-413)
//   wide  : (a leaf link = ~(float4 index of the first record | min(records, 4) - 1): record indices are multiples of 4)
//           f4[0] = origin.xyz, bits(ex | ey<<8 | ez<<16 | n_children<<24)
//           f4[1] = qlo.x[4], qlo.y[4], qlo.z[4], qhi.x[4]   (one byte per child per dword)
//           f4[2] = qhi.y[4], qhi.z[4], link0, link1          f4[3] = link2, link3, 0, 0
//           child box = origin + q * 2^(e-127), rounded OUTWARD
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "pt_items.h"
#include "pt_tree_opt.h"

namespace ptscene {

struct Box3 {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; a++) { lo[a] = 3.402823466e+38f; hi[a] = -3.402823466e+38f; } }
    void grow(const float* p) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    void grow(const Box3& b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    float area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return 2.f * (dx * dy + dy * dz + dz * dx);
    }
};

struct Ref { float v[9]; int32_t id; };          // v0 v1 v2, original triangle id
struct BNode { Box3 cb[2]; int32_t child[2]; };  // child >= 0: node index, < 0: ~leaf index
struct Leaf { uint32_t first, count; };          // range in `refs`

struct Tree {
    std::vector<BNode> nodes;   // nodes[0] = root
    std::vector<Leaf> leaves;
    std::vector<Ref> refs;
};

struct Output {
    std::vector<float> bin, rec, wide;     // 16 floats per item
    uint32_t n_top_bin = 0, n_top_wide = 0;
    uint32_t depth_bin = 0, depth_wide = 0;
    uint64_t n_refs = 0, n_leaves = 0;
    size_t wide_root_f4 = 0;
};

inline int32_t f2i(float f) { int32_t i; std::memcpy(&i, &f, 4); return i; }
inline float i2f(int32_t i) { float f; std::memcpy(&f, &i, 4); return f; }

// ---- 1. parse + validate ---------------------------------------------------------------
inline bool parse(const float* nodes, size_t n_node_vec4, const float* tri_verts, size_t n_tri_vec4,
                  const int32_t* tri_index, Tree& T, std::string& err) {
    const size_t n_in = n_node_vec4 / 4;
    std::vector<int32_t> map(n_in, -1);
    std::vector<uint32_t> depth;
    std::vector<size_t> order{0};
    map[0] = 0;
    T.nodes.clear(); T.leaves.clear(); T.refs.clear();
    T.nodes.push_back(BNode());
    depth.push_back(0);
    for (size_t k = 0; k < order.size(); k++) {
        const size_t u = order[k];
        const float* s = nodes + 16 * u;
        BNode bn;
        for (int i = 0; i < 2; i++) {
            bn.cb[i].lo[0] = s[0 + 4 * i]; bn.cb[i].hi[0] = s[1 + 4 * i];
            bn.cb[i].lo[1] = s[2 + 4 * i]; bn.cb[i].hi[1] = s[3 + 4 * i];
            bn.cb[i].lo[2] = s[8 + 2 * i]; bn.cb[i].hi[2] = s[9 + 2 * i];
            const int32_t l = f2i(s[12 + i]);
            if (l >= 0) {
                if ((l % 64) != 0 || (size_t)l / 64 >= n_in) { err = "child link is not a valid node byte offset"; return false; }
                const size_t c = (size_t)l / 64;
                if (map[c] != -1) { err = "node referenced twice (not a tree)"; return false; }
                map[c] = (int32_t)T.nodes.size();
                bn.child[i] = map[c];
                T.nodes.push_back(BNode());
                depth.push_back(depth[k] + 1);
                if (depth.back() > 64) { err = "tree deeper than 64 (SplitBVHBuilder MaxDepth)"; return false; }
                order.push_back(c);
            } else {
                Leaf lf;
                lf.first = (uint32_t)T.refs.size();
                size_t a = (size_t)(~l);
                for (;; a += 3) {
                    if (a >= n_tri_vec4) { err = "leaf runs past the triangle array"; return false; }
                    uint32_t w0;
                    std::memcpy(&w0, tri_verts + 4 * a, 4);
                    if (w0 == 0x80000000u) break;
                    if (a + 2 >= n_tri_vec4) { err = "leaf runs past the triangle array"; return false; }
                    Ref r;
                    for (int j = 0; j < 3; j++)
                        for (int x = 0; x < 3; x++) r.v[3 * j + x] = tri_verts[4 * (a + j) + x];
                    r.id = tri_index[a];
                    // -1 is the walks' "miss" marker and ids index the per-triangle material rows; Woop
                    // records carry id << 1 | last
                    if (r.id < 0 || r.id >= (1 << 30)) { err = "triangle id out of range (0 .. 2^30-1)"; return false; }
                    T.refs.push_back(r);
                }
                lf.count = (uint32_t)T.refs.size() - lf.first;
                bn.child[i] = ~(int32_t)T.leaves.size();
                T.leaves.push_back(lf);
            }
        }
        T.nodes[k] = bn;
    }
    return true;
}

// ---- 1b. a Tree from the item buffer's own binary nodes + records (what the device builder leaves on the context) --------
// bin: 16 floats per node ([c0.lo.x c0.hi.x c0.lo.y c0.hi.y][c1 ...][c0.lo.z c0.hi.z c1.lo.z c1.hi.z][link0 link1 0 0], links as
// float4 indices into the item buffer, < 0: ~(first record of a leaf)), node 0 = root; rec: 16 floats per record (id in [3],
// `last` flag in [7]).  The vertices come from `by_id` (the caller's own, bit for bit), not from the records' edges.
inline bool from_items(const float* bin, size_t n_bin, const float* rec, size_t n_rec, const std::vector<const float*>& by_id, Tree& T,
                       std::string& err) {
    T.nodes.clear(); T.leaves.clear(); T.refs.clear();
    if (n_bin == 0) { err = "no nodes"; return false; }
    const size_t rec_base = 4 * n_bin;
    std::vector<int32_t> map(n_bin, -1);
    std::vector<size_t> order{0};
    map[0] = 0;
    T.nodes.push_back(BNode());
    for (size_t k = 0; k < order.size(); k++) {
        const float* s = bin + 16 * order[k];
        BNode bn;
        for (int i = 0; i < 2; i++) {
            bn.cb[i].lo[0] = s[0 + 4 * i]; bn.cb[i].hi[0] = s[1 + 4 * i];
            bn.cb[i].lo[1] = s[2 + 4 * i]; bn.cb[i].hi[1] = s[3 + 4 * i];
            bn.cb[i].lo[2] = s[8 + 2 * i]; bn.cb[i].hi[2] = s[9 + 2 * i];
            const int32_t l = f2i(s[12 + i]);
            if (l >= 0) {
                const size_t cidx = (size_t)l / 4;
                if ((l % 4) != 0 || cidx >= n_bin || map[cidx] != -1) { err = "bad inner link"; return false; }
                map[cidx] = (int32_t)T.nodes.size();
                bn.child[i] = map[cidx];
                T.nodes.push_back(BNode());
                order.push_back(cidx);
            } else {
                const size_t r = (size_t)(~l);
                if (r < rec_base || ((r - rec_base) % 4) != 0) { err = "bad leaf link"; return false; }
                Leaf lf;
                lf.first = (uint32_t)T.refs.size();
                for (size_t ri = (r - rec_base) / 4;; ri++) {
                    if (ri >= n_rec) { err = "leaf runs past the records"; return false; }
                    const int32_t id = f2i(rec[16 * ri + 3]);
                    if (id < 0 || (size_t)id >= by_id.size() || !by_id[(size_t)id]) { err = "record of an unknown triangle"; return false; }
                    Ref rf;
                    std::memcpy(rf.v, by_id[(size_t)id], 9 * sizeof(float));
                    rf.id = id;
                    T.refs.push_back(rf);
                    if (f2i(rec[16 * ri + 7]) != 0) break;
                }
                lf.count = (uint32_t)T.refs.size() - lf.first;
                bn.child[i] = ~(int32_t)T.leaves.size();
                T.leaves.push_back(lf);
            }
        }
        T.nodes[k] = bn;
    }
    return true;
}

// ---- 2. refine big leaves --------------------------------------------------------------
inline Box3 ref_box(const Ref& r) {
    Box3 b; b.reset();
    b.grow(r.v); b.grow(r.v + 3); b.grow(r.v + 6);
    return b;
}

// intersection (a spatial-split builder hands over leaves whose box is the CLIPPED part of their triangles: what is cut out of
// such a leaf must stay inside it, or boxes grow again when something re-fits them — PT_OPT_OPTIMIZE does)
inline Box3 clipped(Box3 b, const Box3& bound) {
    for (int a = 0; a < 3; a++) { b.lo[a] = std::max(b.lo[a], bound.lo[a]); b.hi[a] = std::min(b.hi[a], bound.hi[a]); }
    return b;
}

// splits refs[first, first+count) (count > leaf_max) of a leaf with box `bound` and returns the child link of the subtree
inline int32_t split_leaf(Tree& T, uint32_t first, uint32_t count, uint32_t leaf_max, int depth_left, const Box3& bound) {
    if (count <= leaf_max || depth_left <= 0) {
        T.leaves.push_back(Leaf{first, count});
        return ~(int32_t)(T.leaves.size() - 1);
    }
    // SAH sweep over the three axes on centroids (count is small)
    float best = 3.4e38f;
    int best_axis = 0;
    uint32_t best_k = count / 2;
    std::vector<float> right(count);
    for (int ax = 0; ax < 3; ax++) {
        std::sort(T.refs.begin() + first, T.refs.begin() + first + count, [ax](const Ref& a, const Ref& b) {
            const float ca = a.v[ax] + a.v[3 + ax] + a.v[6 + ax], cb = b.v[ax] + b.v[3 + ax] + b.v[6 + ax];
            return ca < cb || (ca == cb && a.id < b.id);
        });
        Box3 acc; acc.reset();
        for (uint32_t i = count - 1; i > 0; i--) { acc.grow(ref_box(T.refs[first + i])); right[i] = acc.area(); }
        acc.reset();
        for (uint32_t i = 1; i < count; i++) {
            acc.grow(ref_box(T.refs[first + i - 1]));
            const float s = acc.area() * (float)i + right[i] * (float)(count - i);
            if (s < best) { best = s; best_axis = ax; best_k = i; }
        }
    }
    std::sort(T.refs.begin() + first, T.refs.begin() + first + count, [best_axis](const Ref& a, const Ref& b) {
        const float ca = a.v[best_axis] + a.v[3 + best_axis] + a.v[6 + best_axis], cb = b.v[best_axis] + b.v[3 + best_axis] + b.v[6 + best_axis];
        return ca < cb || (ca == cb && a.id < b.id);
    });
    const int32_t me = (int32_t)T.nodes.size();
    T.nodes.push_back(BNode());
    BNode bn;
    bn.cb[0].reset(); bn.cb[1].reset();
    for (uint32_t i = 0; i < count; i++) bn.cb[i < best_k ? 0 : 1].grow(ref_box(T.refs[first + i]));
    bn.cb[0] = clipped(bn.cb[0], bound);
    bn.cb[1] = clipped(bn.cb[1], bound);
    bn.child[0] = split_leaf(T, first, best_k, leaf_max, depth_left - 1, bn.cb[0]);
    bn.child[1] = split_leaf(T, first + best_k, count - best_k, leaf_max, depth_left - 1, bn.cb[1]);
    T.nodes[me] = bn;
    return me;
}

inline void refine(Tree& T, uint32_t leaf_max) {
    if (leaf_max == 0) return;
    std::vector<uint32_t> depth(T.nodes.size(), 0);
    const size_t n0 = T.nodes.size();
    for (size_t u = 0; u < n0; u++)
        for (int i = 0; i < 2; i++) {
            const int32_t c = T.nodes[u].child[i];
            if (c >= 0) { if ((size_t)c < depth.size()) depth[c] = depth[u] + 1; continue; }
            const Leaf lf = T.leaves[~c];
            if (lf.count <= leaf_max) continue;
            const Box3 bound = T.nodes[u].cb[i];   // a copy: split_leaf appends to T.nodes
            const int32_t link = split_leaf(T, lf.first, lf.count, leaf_max, 64 - (int)depth[u] - 1, bound);
            T.nodes[u].child[i] = link;  // the old leaf entry is orphaned (never referenced again)
        }
}

// ---- 2b. optimise (PT_OPT_OPTIMIZE): insertion-based optimisation of the hierarchy as it stands after `refine` ---------------
// pt_tree_opt.h works on index arrays: inner nodes keep their numbers, leaf k becomes node n_inner + k.  Returns the tree's area
// cost (sum of inner-node areas / root area) before and after; a result deeper than `max_depth` is thrown away.
inline bool optimize(Tree& T, int passes, uint32_t max_depth, double& cost_before, double& cost_after) {
    cost_before = cost_after = 0.0;
    const size_t n_in = T.nodes.size();
    if (passes <= 0 || n_in < 2) return false;
    // only what is reachable from the root (refine orphans the leaves it splits; their entries stay in T.leaves)
    pttreeopt::Tree O;
    std::vector<int32_t> leaf_of;        // O index -> leaf index (or -1)
    auto box6 = [](const Box3& b) { pttreeopt::Box6 r; for (int a = 0; a < 3; a++) { r.lo[a] = b.lo[a]; r.hi[a] = b.hi[a]; } return r; };
    {
        Box3 rb = T.nodes[0].cb[0];
        rb.grow(T.nodes[0].cb[1]);
        O.root = O.add(box6(rb), -1);
        leaf_of.push_back(-1);
        std::vector<std::pair<size_t, int>> st{{0, O.root}};
        while (!st.empty()) {
            const std::pair<size_t, int> it = st.back();
            st.pop_back();
            int kid[2];
            for (int i = 0; i < 2; i++) {
                const int32_t c = T.nodes[it.first].child[i];
                kid[i] = O.add(box6(T.nodes[it.first].cb[i]), it.second);
                leaf_of.push_back(c < 0 ? ~c : -1);
                if (c >= 0) st.push_back({(size_t)c, kid[i]});
            }
            O.c0[it.second] = kid[0]; O.c1[it.second] = kid[1];
        }
    }
    cost_before = cost_after = O.cost();
    const size_t n_nodes = O.box.size();
    (void)pttreeopt::optimise(O, passes);
    if (!pttreeopt::intact(O, n_nodes) || O.depth() > max_depth) return false;   // (never seen: the caller then emits the tree as it came)
    cost_after = O.cost();
    // back to BNodes: the root must be nodes[0]; inner nodes are renumbered in the order they are met
    std::vector<BNode> out;
    out.reserve(n_in);
    std::vector<int32_t> new_idx(O.box.size(), -1);
    std::vector<int> order{O.root};
    new_idx[O.root] = 0;
    out.push_back(BNode());
    for (size_t k = 0; k < order.size(); k++) {
        const int u = order[k];
        BNode bn;
        const int kid[2] = {O.c0[u], O.c1[u]};
        for (int i = 0; i < 2; i++) {
            const pttreeopt::Box6& b = O.box[kid[i]];
            for (int a = 0; a < 3; a++) { bn.cb[i].lo[a] = b.lo[a]; bn.cb[i].hi[a] = b.hi[a]; }
            if (O.leaf(kid[i])) {
                bn.child[i] = ~leaf_of[kid[i]];
            } else {
                new_idx[kid[i]] = (int32_t)out.size();
                bn.child[i] = new_idx[kid[i]];
                out.push_back(BNode());
                order.push_back(kid[i]);
            }
        }
        out[(size_t)new_idx[u]] = bn;
    }
    T.nodes.swap(out);
    return true;
}

// ---- 3. emit ----------------------------------------------------------------------------
// order: first max_top nodes breadth-first, the rest depth-first (children of a node adjacent)
template <class KidsFn>
inline void bfs_then_dfs(size_t n, size_t max_top, KidsFn kids, std::vector<size_t>& order, std::vector<size_t>& pos,
                         uint32_t& n_top) {
    order.clear();
    order.reserve(n);
    pos.assign(n, SIZE_MAX);
    std::vector<size_t> frontier{0};
    size_t head = 0;
    std::vector<size_t> tmp;
    while (head < frontier.size() && order.size() < max_top) {
        const size_t u = frontier[head++];
        pos[u] = order.size();
        order.push_back(u);
        tmp.clear();
        kids(u, tmp);
        for (size_t c : tmp) frontier.push_back(c);
    }
    n_top = (uint32_t)order.size();
    for (; head < frontier.size(); head++) {
        std::vector<size_t> st{frontier[head]};
        while (!st.empty()) {
            const size_t u = st.back();
            st.pop_back();
            pos[u] = order.size();
            order.push_back(u);
            tmp.clear();
            kids(u, tmp);
            for (size_t k = tmp.size(); k-- > 0;) st.push_back(tmp[k]);
        }
    }
}

struct WNode { Box3 cb[4]; int64_t child[4]; int n; };  // child >= 0: wide index, < 0: ~leaf index

inline void emit(const Tree& T, size_t max_top, Output& out, bool woop = false) {
    // reachable binary nodes, depth-first (also the record emission order)
    std::vector<size_t> reach;
    std::vector<int32_t> leaf_first(T.leaves.size(), -1);  // float4 index of a leaf's first record (relative)
    out.rec.clear();
    out.n_refs = 0; out.n_leaves = 0; out.depth_bin = 0;
    {
        struct It { size_t u; uint32_t d; };
        std::vector<It> st{{0, 0}};
        while (!st.empty()) {
            const It it = st.back();
            st.pop_back();
            reach.push_back(it.u);
            for (int i = 0; i < 2; i++) {
                const int32_t c = T.nodes[it.u].child[i];
                if (c >= 0) continue;
                const Leaf& lf = T.leaves[~c];
                leaf_first[~c] = (int32_t)(out.rec.size() / 4);
                uint32_t cnt = lf.count;
                for (uint32_t k = 0; k < lf.count; k++) {
                    const Ref& r = T.refs[lf.first + k];
                    const float* v0 = r.v; const float* v1 = r.v + 3; const float* v2 = r.v + 6;
                    // cross(v0-v1, v0-v2) with the kernels' vcross arithmetic (cudaUtils.h:432)
                    const float ax = v0[0] - v1[0], ay = v0[1] - v1[1], az = v0[2] - v1[2];
                    const float bx = v0[0] - v2[0], by = v0[1] - v2[1], bz = v0[2] - v2[2];
                    const float N[3] = {std::fmaf(ay, bz, -(az * by)), std::fmaf(az, bx, -(ax * bz)), std::fmaf(ax, by, -(ay * bx))};
                    const int last = k + 1 == lf.count ? 1 : 0;
                    if (woop) {
                        // W = M^-1, M = columns (a, b, n, v2), a = v0-v2, b = v1-v2, n = a x b (n is
                        // orthogonal to a and b, so the rows are (b x n, n x a, n) / |n|^2); binary64
                        const double A[3] = {(double)v0[0] - v2[0], (double)v0[1] - v2[1], (double)v0[2] - v2[2]};
                        const double B[3] = {(double)v1[0] - v2[0], (double)v1[1] - v2[1], (double)v1[2] - v2[2]};
                        const double n[3] = {A[1] * B[2] - A[2] * B[1], A[2] * B[0] - A[0] * B[2], A[0] * B[1] - A[1] * B[0]};
                        const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
                        double r0[3] = {0, 0, 0}, r1[3] = {0, 0, 0}, r2[3] = {0, 0, 0};
                        if (nn > 0.0) {
                            const double bn[3] = {B[1] * n[2] - B[2] * n[1], B[2] * n[0] - B[0] * n[2], B[0] * n[1] - B[1] * n[0]};
                            const double na[3] = {n[1] * A[2] - n[2] * A[1], n[2] * A[0] - n[0] * A[2], n[0] * A[1] - n[1] * A[0]};
                            for (int x = 0; x < 3; x++) { r0[x] = bn[x] / nn; r1[x] = na[x] / nn; r2[x] = n[x] / nn; }
                        }
                        const double t0 = -(r0[0] * v2[0] + r0[1] * v2[1] + r0[2] * v2[2]);
                        const double t1 = -(r1[0] * v2[0] + r1[1] * v2[1] + r1[2] * v2[2]);
                        const double t2 = -(r2[0] * v2[0] + r2[1] * v2[1] + r2[2] * v2[2]);
                        const float rec[16] = {(float)r2[0], (float)r2[1], (float)r2[2], (float)(-t2),
                                               (float)r0[0], (float)r0[1], (float)r0[2], (float)t0,
                                               (float)r1[0], (float)r1[1], (float)r1[2], (float)t1,
                                               N[0], N[1], N[2], i2f((int32_t)(((uint32_t)r.id << 1) | (uint32_t)last))};
                        out.rec.insert(out.rec.end(), rec, rec + 16);
                    } else {
                        float rec[16];
                        pt_encode_record(v0, v1, v2, r.id, last, rec);
                        out.rec.insert(out.rec.end(), rec, rec + 16);
                    }
                }
                if (lf.count == 0) {  // empty leaf: one degenerate record that can never be hit
                    // MT: id -1, last = 1, zero edges (det = 0);  Woop: zero rows (t = 0/0), id<<1|last
                    const float rec_mt[16] = {0, 0, 0, i2f(-1), 0, 0, 0, i2f(1), 0, 0, 0, 0, 0, 0, 0, 0};
                    const float rec_w[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, i2f(-1)};
                    const float* rec = woop ? rec_w : rec_mt;
                    out.rec.insert(out.rec.end(), rec, rec + 16);
                    cnt = 1;
                }
                out.n_refs += cnt;
                out.n_leaves++;
                out.depth_bin = std::max(out.depth_bin, it.d + 1);
            }
            for (int i = 1; i >= 0; i--)
                if (T.nodes[it.u].child[i] >= 0) st.push_back({(size_t)T.nodes[it.u].child[i], it.d + 1});
        }
    }
    const size_t n_bin = reach.size();
    const size_t rec_base = n_bin * 4;                     // float4 index of the first record
    const size_t wide_base = rec_base + out.rec.size() / 4;
    out.wide_root_f4 = wide_base;

    // binary nodes
    {
        std::vector<size_t> order, pos;
        auto kids = [&](size_t u, std::vector<size_t>& k) {
            for (int i = 0; i < 2; i++) if (T.nodes[u].child[i] >= 0) k.push_back((size_t)T.nodes[u].child[i]);
        };
        bfs_then_dfs(T.nodes.size(), max_top, kids, order, pos, out.n_top_bin);
        out.bin.assign(order.size() * 16, 0.f);
        for (size_t k = 0; k < order.size(); k++) {
            const BNode& b = T.nodes[order[k]];
            float* d = &out.bin[16 * k];
            for (int i = 0; i < 2; i++) {
                d[0 + 4 * i] = b.cb[i].lo[0]; d[1 + 4 * i] = b.cb[i].hi[0];
                d[2 + 4 * i] = b.cb[i].lo[1]; d[3 + 4 * i] = b.cb[i].hi[1];
                d[8 + 2 * i] = b.cb[i].lo[2]; d[9 + 2 * i] = b.cb[i].hi[2];
                const int32_t c = b.child[i];
                d[12 + i] = i2f(c >= 0 ? (int32_t)(pos[(size_t)c] * 4) : ~(int32_t)(rec_base + (size_t)leaf_first[~c]));
            }
        }
    }

    // 4-wide collapse: a wide node adopts up to four descendants, largest-area inner child first
    std::vector<WNode> W;
    std::vector<uint32_t> wdepth;
    out.depth_wide = 0;
    {
        auto grow = [&](size_t u) {
            WNode w;
            w.n = 0;
            int64_t ref[4];
            auto add = [&](size_t parent, int i) {
                w.cb[w.n] = T.nodes[parent].cb[i];
                ref[w.n] = T.nodes[parent].child[i];
                w.n++;
            };
            add(u, 0); add(u, 1);
            while (w.n < 4) {
                int best = -1;
                float ba = -1.f;
                for (int k = 0; k < w.n; k++)
                    if (ref[k] >= 0 && w.cb[k].area() > ba) { ba = w.cb[k].area(); best = k; }
                if (best < 0) break;
                const size_t v = (size_t)ref[best];
                w.cb[best] = w.cb[w.n - 1];
                ref[best] = ref[w.n - 1];
                w.n--;
                add(v, 0); add(v, 1);
            }
            for (int k = 0; k < 4; k++) w.child[k] = k < w.n ? ref[k] : 0;
            return w;
        };
        // breadth-first creation; inner refs (binary node numbers) become wide indices
        W.push_back(grow(0));
        wdepth.push_back(0);
        for (size_t i = 0; i < W.size(); i++)
            for (int k = 0; k < W[i].n; k++) {
                if (W[i].child[k] >= 0) {
                    const size_t bin_node = (size_t)W[i].child[k];
                    W[i].child[k] = (int64_t)W.size();
                    W.push_back(grow(bin_node));
                    wdepth.push_back(wdepth[i] + 1);
                } else {
                    out.depth_wide = std::max(out.depth_wide, wdepth[i] + 1);
                }
            }
    }
    {
        std::vector<size_t> order, pos;
        auto kids = [&](size_t u, std::vector<size_t>& k) {
            for (int i = 0; i < W[u].n; i++) if (W[u].child[i] >= 0) k.push_back((size_t)W[u].child[i]);
        };
        bfs_then_dfs(W.size(), max_top, kids, order, pos, out.n_top_wide);
        out.wide.assign(order.size() * 16, 0.f);
        for (size_t oi = 0; oi < order.size(); oi++) {
            const WNode& w = W[order[oi]];
            float* d = &out.wide[16 * oi];
            int32_t link[4] = {0, 0, 0, 0};
            for (int k = 0; k < w.n; k++)
                if (w.child[k] >= 0) {
                    link[k] = (int32_t)(wide_base + 4 * pos[(size_t)w.child[k]]);
                } else {   // a leaf link also says how many records the leaf holds (low two bits: min(count, 4) - 1)
                    const size_t li = (size_t)~(int32_t)w.child[k];
                    const uint32_t n_rec = std::max<uint32_t>(T.leaves[li].count, 1u);   // an empty leaf holds one dummy record
                    link[k] = ~(int32_t)((rec_base + (size_t)leaf_first[li]) | (size_t)(std::min<uint32_t>(n_rec, 4u) - 1u));
                }
            PtBox cb[4];
            for (int k = 0; k < w.n; k++)
                for (int a = 0; a < 3; a++) { cb[k].lo[a] = w.cb[k].lo[a]; cb[k].hi[a] = w.cb[k].hi[a]; }
            pt_encode_wide_node(cb, w.n, link, d);
        }
    }
}

}  // namespace ptscene

// pthost.cpp — host-side scene preparation (see pthost.h for the contract and the
// reference files this stands in for).  Clean-room: written from the layout/behaviour
// description in SURVEY.md §2/§8(a12), not from the reference's builder sources.
#include "pthost.h"
#include "../csrc/pt_tree_opt.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
void set_err(const std::string& s) { g_err = s; }

struct Vec3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 vmin(Vec3 a, Vec3 b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }
inline Vec3 vmax(Vec3 a, Vec3 b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }

struct Box {
    Vec3 lo{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    Vec3 hi{-std::numeric_limits<float>::max(), -std::numeric_limits<float>::max(), -std::numeric_limits<float>::max()};
    void grow(Vec3 p) { lo = vmin(lo, p); hi = vmax(hi, p); }
    void grow(const Box& b) { lo = vmin(lo, b.lo); hi = vmax(hi, b.hi); }
    bool valid() const { return lo.x <= hi.x && lo.y <= hi.y && lo.z <= hi.z; }
    float area() const {
        if (!valid()) return 0.f;
        float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
        return 2.f * (dx * dy + dy * dz + dz * dx);
    }
    void clip(const Box& b) { lo = vmax(lo, b.lo); hi = vmin(hi, b.hi); }
};

struct Ref {
    int tri;
    Box box;
};

struct Node {
    Box box;
    std::unique_ptr<Node> child[2];
    std::vector<int> tris;  // leaf payload (triangle ids)
    bool leaf() const { return !child[0]; }
};

}  // namespace

struct pth_mesh {
    std::vector<float> verts;    // xyz
    std::vector<int32_t> tris;   // 3 per triangle
    std::vector<pth_material> mats;   // empty: the file named no materials
    std::vector<int32_t> tri_mat;     // row of `mats` per triangle (empty with mats)
};

struct pth_bvh {
    std::vector<float> nodes;    // vec4 array
    std::vector<float> tris;     // vec4 array
    std::vector<int32_t> index;
    pth_bvh_stats stats{};
};

namespace {

struct Builder {
    const pth_mesh& mesh;
    pth_build_params P;
    float min_overlap = 0.f;
    uint64_t n_inner = 0, n_leaves = 0, n_refs = 0;
    double sah = 0.0;

    Builder(const pth_mesh& m, const pth_build_params& p) : mesh(m), P(p) {}

    Vec3 vert(int tri, int k) const {
        const float* v = &mesh.verts[3 * (size_t)mesh.tris[3 * (size_t)tri + k]];
        return {v[0], v[1], v[2]};
    }

    std::unique_ptr<Node> make_leaf(const std::vector<Ref>& refs, const Box& box) {
        auto n = std::make_unique<Node>();
        n->box = box;
        n->tris.reserve(refs.size());
        for (const Ref& r : refs) n->tris.push_back(r.tri);
        return n;
    }

    struct Split {
        float sah = std::numeric_limits<float>::max();
        int axis = -1;
        float pos = 0.f;   // spatial: plane position; object (binned): bin boundary centroid value
        int bin = -1;      // object binned: refs with bin index < bin go left
        bool sweep = false;  // object full sweep: `bin` = number of refs on the left in sorted order
    };

    // SAH of splitting `refs` by centroid into N bins on each axis.
    Split find_object_split_binned(const std::vector<Ref>& refs, const Box& cbox, float node_sah) {
        Split best;
        const int NB = std::max(2, P.n_bins);
        std::vector<Box> bb(NB);
        std::vector<int> bc(NB);
        std::vector<float> right_area(NB);
        std::vector<int> right_cnt(NB);
        for (int ax = 0; ax < 3; ax++) {
            float lo = cbox.lo[ax], hi = cbox.hi[ax];
            if (!(hi > lo)) continue;
            float scale = (float)NB / (hi - lo);
            for (int i = 0; i < NB; i++) { bb[i] = Box(); bc[i] = 0; }
            for (const Ref& r : refs) {
                float c = 0.5f * (r.box.lo[ax] + r.box.hi[ax]);
                int b = std::min(NB - 1, std::max(0, (int)((c - lo) * scale)));
                bb[b].grow(r.box);
                bc[b]++;
            }
            Box acc;
            int cnt = 0;
            for (int i = NB - 1; i > 0; i--) {
                acc.grow(bb[i]);
                cnt += bc[i];
                right_area[i] = acc.area();
                right_cnt[i] = cnt;
            }
            acc = Box();
            cnt = 0;
            for (int i = 1; i < NB; i++) {
                acc.grow(bb[i - 1]);
                cnt += bc[i - 1];
                if (cnt == 0 || right_cnt[i] == 0) continue;
                float s = node_sah + acc.area() * P.sah_tri_cost * cnt + right_area[i] * P.sah_tri_cost * right_cnt[i];
                if (s < best.sah) { best.sah = s; best.axis = ax; best.bin = i; best.pos = lo; }
            }
        }
        return best;
    }

    // Exact sweep over refs sorted by centroid (small nodes).
    Split find_object_split_sweep(std::vector<Ref>& refs, float node_sah) {
        Split best;
        const size_t n = refs.size();
        std::vector<float> right_area(n);
        for (int ax = 0; ax < 3; ax++) {
            std::sort(refs.begin(), refs.end(), [ax](const Ref& a, const Ref& b) {
                float ca = a.box.lo[ax] + a.box.hi[ax], cb = b.box.lo[ax] + b.box.hi[ax];
                return ca < cb || (ca == cb && a.tri < b.tri);
            });
            Box acc;
            for (size_t i = n - 1; i > 0; i--) { acc.grow(refs[i].box); right_area[i] = acc.area(); }
            acc = Box();
            for (size_t i = 1; i < n; i++) {
                acc.grow(refs[i - 1].box);
                float s = node_sah + acc.area() * P.sah_tri_cost * (float)i + right_area[i] * P.sah_tri_cost * (float)(n - i);
                if (s < best.sah) { best.sah = s; best.axis = ax; best.bin = (int)i; best.sweep = true; }
            }
        }
        return best;
    }

    // Clip triangle `tri` (restricted to `box`) against plane axis=pos; returns the two boxes.
    void split_reference(const Ref& r, int ax, float pos, Ref& left, Ref& right) const {
        left.tri = right.tri = r.tri;
        left.box = Box();
        right.box = Box();
        Vec3 v[3] = {vert(r.tri, 0), vert(r.tri, 1), vert(r.tri, 2)};
        for (int i = 0; i < 3; i++) {
            Vec3 a = v[i], b = v[(i + 1) % 3];
            float pa = a[ax], pb = b[ax];
            if (pa <= pos) left.box.grow(a);
            if (pa >= pos) right.box.grow(a);
            if ((pa < pos && pb > pos) || (pa > pos && pb < pos)) {
                float t = std::min(1.f, std::max(0.f, (pos - pa) / (pb - pa)));
                Vec3 p{a.x + (b.x - a.x) * t, a.y + (b.y - a.y) * t, a.z + (b.z - a.z) * t};
                p.at(ax) = pos;
                left.box.grow(p);
                right.box.grow(p);
            }
        }
        left.box.hi.at(ax) = pos;
        right.box.lo.at(ax) = pos;
        left.box.clip(r.box);
        right.box.clip(r.box);
    }

    Split find_spatial_split(const std::vector<Ref>& refs, const Box& box, float node_sah) {
        Split best;
        const int NB = std::max(2, P.n_spatial_bins);
        struct Bin { Box b; int enter = 0, exit = 0; };
        std::vector<Bin> bins(3 * (size_t)NB);
        Vec3 origin = box.lo;
        Vec3 size{(box.hi.x - box.lo.x) / NB, (box.hi.y - box.lo.y) / NB, (box.hi.z - box.lo.z) / NB};
        for (const Ref& r : refs) {
            for (int ax = 0; ax < 3; ax++) {
                if (!(size[ax] > 0.f)) continue;
                float inv = 1.f / size[ax];
                int first = std::min(NB - 1, std::max(0, (int)((r.box.lo[ax] - origin[ax]) * inv)));
                int last = std::min(NB - 1, std::max(first, (int)((r.box.hi[ax] - origin[ax]) * inv)));
                Ref cur = r;
                for (int i = first; i < last; i++) {
                    Ref l, rr;
                    split_reference(cur, ax, origin[ax] + size[ax] * (float)(i + 1), l, rr);
                    if (l.box.valid()) bins[(size_t)ax * NB + i].b.grow(l.box);
                    cur = rr;
                }
                if (cur.box.valid()) bins[(size_t)ax * NB + last].b.grow(cur.box);
                bins[(size_t)ax * NB + first].enter++;
                bins[(size_t)ax * NB + last].exit++;
            }
        }
        std::vector<float> right_area(NB);
        for (int ax = 0; ax < 3; ax++) {
            if (!(size[ax] > 0.f)) continue;
            Box acc;
            for (int i = NB - 1; i > 0; i--) { acc.grow(bins[(size_t)ax * NB + i].b); right_area[i] = acc.area(); }
            acc = Box();
            int ln = 0, rn = (int)refs.size();
            for (int i = 1; i < NB; i++) {
                acc.grow(bins[(size_t)ax * NB + i - 1].b);
                ln += bins[(size_t)ax * NB + i - 1].enter;
                rn -= bins[(size_t)ax * NB + i - 1].exit;
                if (ln == 0 || rn == 0) continue;
                float s = node_sah + acc.area() * P.sah_tri_cost * ln + right_area[i] * P.sah_tri_cost * rn;
                if (s < best.sah) { best.sah = s; best.axis = ax; best.pos = origin[ax] + size[ax] * (float)i; }
            }
        }
        return best;
    }

    std::unique_ptr<Node> build(std::vector<Ref>& refs, const Box& box, int depth, bool parallel) {
        const size_t n = refs.size();
        if ((int)n <= P.min_leaf_size || depth >= P.max_depth) return make_leaf(refs, box);

        const float area = box.area();
        const float leaf_sah = area * P.sah_tri_cost * (float)n;
        const float node_sah = area * P.sah_node_cost * 2.f;

        Box cbox;
        for (const Ref& r : refs) {
            cbox.grow(Vec3{0.5f * (r.box.lo.x + r.box.hi.x), 0.5f * (r.box.lo.y + r.box.hi.y), 0.5f * (r.box.lo.z + r.box.hi.z)});
        }
        Split obj = (P.n_bins <= 0 || n <= 32) ? find_object_split_sweep(refs, node_sah)
                                               : find_object_split_binned(refs, cbox, node_sah);
        Split spa;
        if (P.split_alpha >= 0.f && depth < P.max_depth - 8 && obj.axis >= 0) {
            // overlap of the object split's children gates the spatial attempt (Stich et al.)
            Box lb, rb;
            partition_object(refs, obj, cbox, nullptr, nullptr, &lb, &rb);
            lb.clip(rb);
            if (lb.valid() && lb.area() >= min_overlap) spa = find_spatial_split(refs, box, node_sah);
        }

        float min_sah = std::min(leaf_sah, std::min(obj.sah, spa.sah));
        if (min_sah == leaf_sah && (int)n <= P.max_leaf_size) return make_leaf(refs, box);

        std::vector<Ref> L, R;
        Box lb, rb;
        if (spa.sah < obj.sah && spa.axis >= 0) {
            partition_spatial(refs, spa, L, R, lb, rb);
            if (L.empty() || R.empty() || (L.size() == n && R.size() == n)) { L.clear(); R.clear(); }
        }
        if (L.empty() || R.empty()) {
            L.clear(); R.clear();
            if (obj.axis >= 0) partition_object(refs, obj, cbox, &L, &R, &lb, &rb);
            if (L.empty() || R.empty()) {  // degenerate (coincident centroids): median by id
                L.clear(); R.clear(); lb = Box(); rb = Box();
                std::sort(refs.begin(), refs.end(), [](const Ref& a, const Ref& b) { return a.tri < b.tri; });
                for (size_t i = 0; i < n; i++) {
                    if (i < n / 2) { L.push_back(refs[i]); lb.grow(refs[i].box); }
                    else { R.push_back(refs[i]); rb.grow(refs[i].box); }
                }
            }
        }
        std::vector<Ref>().swap(refs);

        auto node = std::make_unique<Node>();
        node->box = box;
        if (parallel && n > 4096) {
#pragma omp task shared(node, L, lb) firstprivate(depth)
            node->child[0] = build(L, lb, depth + 1, true);
#pragma omp task shared(node, R, rb) firstprivate(depth)
            node->child[1] = build(R, rb, depth + 1, true);
#pragma omp taskwait
        } else {
            node->child[0] = build(L, lb, depth + 1, false);
            node->child[1] = build(R, rb, depth + 1, false);
        }
        return node;
    }

    // Applies an object split; any of the outputs may be null (used for the overlap probe).
    void partition_object(std::vector<Ref>& refs, const Split& s, const Box& cbox,
                          std::vector<Ref>* L, std::vector<Ref>* R, Box* lb, Box* rb) {
        Box l, r;
        const size_t n = refs.size();
        if (s.sweep) {
            int ax = s.axis;
            std::sort(refs.begin(), refs.end(), [ax](const Ref& a, const Ref& b) {
                float ca = a.box.lo[ax] + a.box.hi[ax], cb = b.box.lo[ax] + b.box.hi[ax];
                return ca < cb || (ca == cb && a.tri < b.tri);
            });
            for (size_t i = 0; i < n; i++) {
                bool left = (int)i < s.bin;
                (left ? l : r).grow(refs[i].box);
                if (L && R) (left ? *L : *R).push_back(refs[i]);
            }
        } else {
            const int NB = std::max(2, P.n_bins);
            int ax = s.axis;
            float lo = cbox.lo[ax], hi = cbox.hi[ax];
            float scale = (float)NB / (hi - lo);
            for (const Ref& rf : refs) {
                float c = 0.5f * (rf.box.lo[ax] + rf.box.hi[ax]);
                int b = std::min(NB - 1, std::max(0, (int)((c - lo) * scale)));
                bool left = b < s.bin;
                (left ? l : r).grow(rf.box);
                if (L && R) (left ? *L : *R).push_back(rf);
            }
        }
        if (lb) *lb = l;
        if (rb) *rb = r;
    }

    void partition_spatial(const std::vector<Ref>& refs, const Split& s, std::vector<Ref>& L, std::vector<Ref>& R,
                           Box& lb, Box& rb) {
        lb = Box(); rb = Box();
        std::vector<Ref> straddle;
        for (const Ref& r : refs) {
            if (r.box.hi[s.axis] <= s.pos) { L.push_back(r); lb.grow(r.box); }
            else if (r.box.lo[s.axis] >= s.pos) { R.push_back(r); rb.grow(r.box); }
            else straddle.push_back(r);
        }
        // duplicate-or-unsplit decision per straddling reference (SBVH paper §4.4)
        for (const Ref& r : straddle) {
            Ref l, rr;
            split_reference(r, s.axis, s.pos, l, rr);
            Box lub = lb, rub = rb, ldb = lb, rdb = rb;
            lub.grow(r.box); rub.grow(r.box); ldb.grow(l.box); rdb.grow(rr.box);
            float lac = P.sah_tri_cost * (float)L.size(), rac = P.sah_tri_cost * (float)R.size();
            float lbc = P.sah_tri_cost * (float)(L.size() + 1), rbc = P.sah_tri_cost * (float)(R.size() + 1);
            float unsplit_l = lub.area() * lbc + rb.area() * rac;
            float unsplit_r = lb.area() * lac + rub.area() * rbc;
            float dup = ldb.area() * lbc + rdb.area() * rbc;
            float m = std::min(unsplit_l, std::min(unsplit_r, dup));
            if (m == unsplit_l || !l.box.valid() || !rr.box.valid()) {
                if (m == unsplit_r && l.box.valid() == false) { R.push_back(r); rb = rub; }
                else { L.push_back(r); lb = lub; }
            } else if (m == unsplit_r) { R.push_back(r); rb = rub; }
            else { L.push_back(l); R.push_back(rr); lb = ldb; rb = rdb; }
        }
    }
};


// ---------------------------------------------------------------------------------------------------------------------
// pth_build_params::optimize_passes (extension): insertion-based optimisation of the finished hierarchy, ../csrc/pt_tree_opt.h.
// Node tree -> index arrays -> passes -> Node tree; a leaf keeps its triangle list and its (possibly clipped) box.
struct OptBridge {
    pttreeopt::Tree T;
    std::vector<const Node*> src;   // node index -> the builder's node (leaf payload)

    static pttreeopt::Box6 box6(const Box& b) { return {{b.lo.x, b.lo.y, b.lo.z}, {b.hi.x, b.hi.y, b.hi.z}}; }
    void from(const Node* r) {
        struct It { const Node* n; int idx; };
        T.root = T.add(box6(r->box), -1);
        src.push_back(r);
        std::vector<It> st{{r, T.root}};
        while (!st.empty()) {
            const It it = st.back();
            st.pop_back();
            if (it.n->leaf()) continue;
            const int a = T.add(box6(it.n->child[0]->box), it.idx);
            src.push_back(it.n->child[0].get());
            const int b = T.add(box6(it.n->child[1]->box), it.idx);
            src.push_back(it.n->child[1].get());
            T.c0[it.idx] = a; T.c1[it.idx] = b;
            st.push_back({it.n->child[0].get(), a});
            st.push_back({it.n->child[1].get(), b});
        }
    }
    std::unique_ptr<Node> to_nodes() const {   // children before parents: post-order through an explicit stack
        std::vector<std::unique_ptr<Node>> made(T.box.size());
        std::vector<std::pair<int, bool>> st{{T.root, false}};
        auto boxof = [&](int i) { Box b; b.lo = {T.box[i].lo[0], T.box[i].lo[1], T.box[i].lo[2]}; b.hi = {T.box[i].hi[0], T.box[i].hi[1], T.box[i].hi[2]}; return b; };
        while (!st.empty()) {
            const std::pair<int, bool> it = st.back();
            st.pop_back();
            const int i = it.first;
            if (T.leaf(i)) {
                made[i] = std::make_unique<Node>();
                made[i]->box = src[i]->box;
                made[i]->tris = src[i]->tris;
            } else if (!it.second) {
                st.push_back({i, true});
                st.push_back({T.c0[i], false});
                st.push_back({T.c1[i], false});
            } else {
                made[i] = std::make_unique<Node>();
                made[i]->box = boxof(i);
                made[i]->child[0] = std::move(made[T.c0[i]]);
                made[i]->child[1] = std::move(made[T.c1[i]]);
            }
        }
        return std::move(made[T.root]);
    }
};

struct FlattenCtx {
    std::vector<float>& nodes;
    std::vector<float>& tris;
    std::vector<int32_t>& index;
    const pth_mesh& mesh;
    uint64_t n_inner = 0, n_leaves = 0, n_refs = 0;
    uint32_t max_depth = 0;
    double sah = 0.0;
};

inline float bits_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

int32_t emit_leaf(FlattenCtx& c, const std::vector<int>& ids) {
    int32_t link = ~(int32_t)(c.tris.size() / 4);
    for (int id : ids) {
        for (int k = 0; k < 3; k++) {
            const float* v = &c.mesh.verts[3 * (size_t)c.mesh.tris[3 * (size_t)id + k]];
            // a v0.x of -0.0f has the terminator's bit pattern (0x80000000) and would end the
            // leaf early in any consumer of this layout (cudaUtils.h:413); the reference guards
            // only its Woop rows against it (CudaBVH.cpp:191).  Store +0.0f instead.
            const float x = (k == 0 && v[0] == 0.f) ? 0.f : v[0];
            c.tris.insert(c.tris.end(), {x, v[1], v[2], 0.f});
        }
        c.index.insert(c.index.end(), {id, 0, 0});
    }
    float term = bits_as_float(0x80000000u);
    c.tris.insert(c.tris.end(), {term, term, term, term});
    c.index.push_back(0);
    c.n_leaves++;
    c.n_refs += ids.size();
    return link;
}

void flatten(FlattenCtx& c, const Node* root) {
    struct Item { const Node* n; size_t slot; uint32_t depth; };
    c.nodes.assign(16, 0.f);  // root occupies the first 4 vec4 (CudaBVH.cpp:129)
    std::vector<Item> stack{{root, 0, 0}};
    const float root_area = std::max(root->box.area(), 1e-30f);
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        c.n_inner++;
        c.sah += it.n->box.area() / root_area;
        int32_t link[2];
        const Box* cb[2];
        for (int i = 0; i < 2; i++) {
            const Node* ch = it.n->child[i].get();
            cb[i] = &ch->box;
            if (!ch->leaf()) {
                size_t slot = c.nodes.size();
                link[i] = (int32_t)(slot * sizeof(float));  // byte offset
                c.nodes.resize(slot + 16, 0.f);
                stack.push_back({ch, slot, it.depth + 1});
            } else {
                link[i] = emit_leaf(c, ch->tris);
                c.sah += ch->box.area() / root_area * (double)ch->tris.size();
                c.max_depth = std::max(c.max_depth, it.depth + 1);
            }
        }
        float* d = &c.nodes[it.slot];
        d[0] = cb[0]->lo.x; d[1] = cb[0]->hi.x; d[2] = cb[0]->lo.y; d[3] = cb[0]->hi.y;
        d[4] = cb[1]->lo.x; d[5] = cb[1]->hi.x; d[6] = cb[1]->lo.y; d[7] = cb[1]->hi.y;
        d[8] = cb[0]->lo.z; d[9] = cb[0]->hi.z; d[10] = cb[1]->lo.z; d[11] = cb[1]->hi.z;
        d[12] = bits_as_float((uint32_t)link[0]);
        d[13] = bits_as_float((uint32_t)link[1]);
        d[14] = 0.f; d[15] = 0.f;
    }
}

pth_mesh* mesh_from(std::vector<float>&& v, std::vector<int32_t>&& t) {
    const size_t nv = v.size() / 3;
    for (int32_t i : t)
        if (i < 0 || (size_t)i >= nv) { set_err("triangle index out of range"); return nullptr; }
    auto* m = new pth_mesh();
    m->verts = std::move(v);
    m->tris = std::move(t);
    return m;
}

}  // namespace

extern "C" {

const char* pth_last_error(void) { return g_err.c_str(); }

void pth_default_build_params(pth_build_params* p) {
    p->max_leaf_size = 0x7FFFFFF;
    p->min_leaf_size = 1;
    p->max_depth = 64;
    p->n_bins = 32;
    p->sah_node_cost = 1.f;
    p->sah_tri_cost = 1.f;
    p->split_alpha = 1e-5f;  // BuildParams::splitAlpha of the reference (SBVH)
    p->n_spatial_bins = 32;
    p->optimize_passes = 0;
}

pth_mesh* pth_mesh_create(const float* verts, size_t n_verts, const int32_t* tris, size_t n_tris) {
    if (!verts || !tris) { set_err("null mesh arrays"); return nullptr; }
    return mesh_from(std::vector<float>(verts, verts + 3 * n_verts), std::vector<int32_t>(tris, tris + 3 * n_tris));
}

pth_mesh* pth_mesh_load_ptmesh(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) { set_err(std::string("cannot open ") + path); return nullptr; }
    char magic[8];
    uint32_t nv = 0, nt = 0, nm = 0;
    bool ok = std::fread(magic, 1, 8, f) == 8;
    const bool v2 = ok && std::memcmp(magic, "PTMESH2", 8) == 0;
    ok = ok && (v2 || std::memcmp(magic, "PTMESH1", 8) == 0) &&
         std::fread(&nv, 4, 1, f) == 1 && std::fread(&nt, 4, 1, f) == 1 && (!v2 || std::fread(&nm, 4, 1, f) == 1);
    std::vector<float> v;
    std::vector<int32_t> t, tm;
    std::vector<pth_material> mats;
    if (ok) {
        v.resize(3 * (size_t)nv);
        t.resize(3 * (size_t)nt);
        ok = std::fread(v.data(), 4, v.size(), f) == v.size() && std::fread(t.data(), 4, t.size(), f) == t.size();
    }
    if (ok && nm) {  // PTMESH2: material table + one row index per triangle
        mats.resize(nm);
        tm.resize(nt);
        ok = std::fread(mats.data(), sizeof(pth_material), nm, f) == nm && std::fread(tm.data(), 4, nt, f) == nt;
        for (size_t i = 0; ok && i < tm.size(); i++) ok = tm[i] >= 0 && (uint32_t)tm[i] < nm;
    }
    std::fclose(f);
    if (!ok) { set_err(std::string("bad ptmesh file ") + path); return nullptr; }
    pth_mesh* m = mesh_from(std::move(v), std::move(t));
    if (m) { m->mats = std::move(mats); m->tri_mat = std::move(tm); }
    return m;
}

namespace {
// Wavefront .mtl → pth_material rows.  Kd → col, Ke → emi; the lobe follows `illum`:
// 3 → METAL (Phong exponent Ns), 5 → SPEC (mirror), 4/6/7 → REFR (glass), anything else DIFF.
// (The reference parses the .mtl through tinyobj and then ignores it: utilfun.cpp:458-462.)
bool load_mtl(const std::string& path, std::vector<std::string>& names, std::vector<pth_material>& mats) {
    FILE* f = std::fopen(path.c_str(), "r");
    if (!f) return false;
    std::vector<char> line(1 << 12);
    std::vector<float> ns;
    while (std::fgets(line.data(), (int)line.size(), f)) {
        const char* s = line.data();
        while (*s == ' ' || *s == '\t') s++;
        char name[512];
        float a, b, c;
        int il;
        if (std::sscanf(s, "newmtl %511s", name) == 1) {
            names.emplace_back(name);
            pth_material m{{0.8f, 0.8f, 0.8f}, {0.f, 0.f, 0.f}, 0, 0.f};
            mats.push_back(m);
            ns.push_back(0.f);
        } else if (mats.empty()) {
            continue;
        } else if (std::sscanf(s, "Kd %f %f %f", &a, &b, &c) == 3) {
            mats.back().col[0] = a; mats.back().col[1] = b; mats.back().col[2] = c;
        } else if (std::sscanf(s, "Ke %f %f %f", &a, &b, &c) == 3) {
            mats.back().emi[0] = a; mats.back().emi[1] = b; mats.back().emi[2] = c;
        } else if (std::sscanf(s, "Ns %f", &a) == 1) {
            ns.back() = a;
        } else if (std::sscanf(s, "illum %d", &il) == 1) {
            mats.back().mat = il == 3 ? 1 : il == 5 ? 2 : (il == 4 || il == 6 || il == 7) ? 3 : 0;
        }
    }
    std::fclose(f);
    for (size_t i = 0; i < mats.size(); i++) mats[i].phong_expo = mats[i].mat == 1 ? ns[i] : 0.f;
    return true;
}
}  // namespace


pth_mesh* pth_mesh_load_obj(const char* path) {
    FILE* f = std::fopen(path, "r");
    if (!f) { set_err(std::string("cannot open ") + path); return nullptr; }
    std::vector<float> v;
    std::vector<int32_t> t, tm;
    std::vector<std::string> mat_names;
    std::vector<pth_material> mats;
    int32_t cur_mat = -1;      // -1: no usemtl seen yet
    bool any_unnamed = false;
    std::vector<char> line(1 << 16);
    while (std::fgets(line.data(), (int)line.size(), f)) {
        const char* s = line.data();
        while (*s == ' ' || *s == '\t') s++;
        char word[1024];
        if (s[0] == 'm' && std::sscanf(s, "mtllib %1023s", word) == 1) {
            std::string dir(path);
            const size_t slash = dir.find_last_of('/');
            dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
            (void)load_mtl(dir + word, mat_names, mats);   // a missing .mtl leaves the mesh without materials
        } else if (s[0] == 'u' && std::sscanf(s, "usemtl %1023s", word) == 1) {
            cur_mat = -1;
            for (size_t i = 0; i < mat_names.size(); i++)
                if (mat_names[i] == word) cur_mat = (int32_t)i;
        } else if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            float x, y, z;
            if (std::sscanf(s + 1, "%f %f %f", &x, &y, &z) == 3) v.insert(v.end(), {x, y, z});
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            std::vector<int32_t> poly;
            const char* p = s + 1;
            while (*p) {
                while (*p == ' ' || *p == '\t') p++;
                if (*p == '\0' || *p == '\n' || *p == '\r') break;
                char* end = nullptr;
                long idx = std::strtol(p, &end, 10);
                if (end == p) break;
                const long nv = (long)(v.size() / 3);
                poly.push_back((int32_t)(idx > 0 ? idx - 1 : nv + idx));
                p = end;
                while (*p && *p != ' ' && *p != '\t' && *p != '\n' && *p != '\r') p++;  // skip /vt/vn
            }
            for (size_t k = 1; k + 1 < poly.size(); k++) {
                t.insert(t.end(), {poly[0], poly[k], poly[k + 1]});
                tm.push_back(cur_mat);
                any_unnamed = any_unnamed || cur_mat < 0;
            }
        }
    }
    std::fclose(f);
    if (t.empty()) { set_err(std::string("no faces in ") + path); return nullptr; }
    pth_mesh* m = mesh_from(std::move(v), std::move(t));
    if (m && !mats.empty()) {
        if (any_unnamed) {  // faces outside any usemtl get tinyobj's default grey
            mats.push_back(pth_material{{0.8f, 0.8f, 0.8f}, {0.f, 0.f, 0.f}, 0, 0.f});
            for (int32_t& i : tm) if (i < 0) i = (int32_t)mats.size() - 1;
        }
        m->mats = std::move(mats);
        m->tri_mat = std::move(tm);
    }
    return m;
}

int pth_mesh_append(pth_mesh* dst, const pth_mesh* src, const float* m) {
    if (!dst || !src) { set_err("null mesh"); return -1; }
    const int32_t base = (int32_t)(dst->verts.size() / 3);
    const size_t nv = src->verts.size() / 3;
    // src may alias dst: copy first
    std::vector<float> sv(src->verts);
    std::vector<int32_t> st(src->tris);
    for (size_t i = 0; i < nv; i++) {
        float x = sv[3 * i], y = sv[3 * i + 1], z = sv[3 * i + 2];
        if (m) {
            float nx = m[0] * x + m[1] * y + m[2] * z + m[3];
            float ny = m[4] * x + m[5] * y + m[6] * z + m[7];
            float nz = m[8] * x + m[9] * y + m[10] * z + m[11];
            x = nx; y = ny; z = nz;
        }
        dst->verts.insert(dst->verts.end(), {x, y, z});
    }
    // materials travel with the triangles; a side without a table gets one default grey row
    const size_t nt_dst = dst->tris.size() / 3, nt_src = st.size() / 3;
    if (!dst->mats.empty() || !src->mats.empty()) {
        const std::vector<pth_material> sm(src->mats);
        const std::vector<int32_t> stm(src->tri_mat);
        const pth_material grey{{0.8f, 0.8f, 0.8f}, {0.f, 0.f, 0.f}, 0, 0.f};
        if (dst->mats.empty()) { dst->mats.push_back(grey); dst->tri_mat.assign(nt_dst, 0); }
        const int32_t mbase = (int32_t)dst->mats.size();
        if (sm.empty()) {
            dst->mats.push_back(grey);
            dst->tri_mat.insert(dst->tri_mat.end(), nt_src, mbase);
        } else {
            dst->mats.insert(dst->mats.end(), sm.begin(), sm.end());
            for (int32_t i : stm) dst->tri_mat.push_back(mbase + i);
        }
    }
    for (int32_t i : st) dst->tris.push_back(base + i);
    return 0;
}

size_t pth_mesh_n_materials(const pth_mesh* m) { return m ? m->mats.size() : 0; }
const pth_material* pth_mesh_materials(const pth_mesh* m) { return m && !m->mats.empty() ? m->mats.data() : nullptr; }
const int32_t* pth_mesh_tri_materials(const pth_mesh* m) { return m && !m->mats.empty() ? m->tri_mat.data() : nullptr; }
int pth_mesh_set_materials(pth_mesh* m, const pth_material* table, size_t n, const int32_t* tri_mat) {
    if (!m) { set_err("null mesh"); return -1; }
    if (n == 0) { m->mats.clear(); m->tri_mat.clear(); return 0; }
    if (!table || !tri_mat) { set_err("null material arrays"); return -1; }
    const size_t nt = m->tris.size() / 3;
    for (size_t i = 0; i < nt; i++)
        if (tri_mat[i] < 0 || (size_t)tri_mat[i] >= n) { set_err("material index out of range"); return -1; }
    for (size_t i = 0; i < n; i++)
        if (table[i].mat < 0 || table[i].mat > 3) { set_err("bad material type"); return -1; }
    m->mats.assign(table, table + n);
    m->tri_mat.assign(tri_mat, tri_mat + nt);
    return 0;
}
int pth_mesh_save_ptmesh(const pth_mesh* m, const char* path) {
    if (!m || !path) { set_err("null argument"); return -1; }
    FILE* f = std::fopen(path, "wb");
    if (!f) { set_err(std::string("cannot write ") + path); return -1; }
    const uint32_t nv = (uint32_t)(m->verts.size() / 3), nt = (uint32_t)(m->tris.size() / 3), nm = (uint32_t)m->mats.size();
    bool ok = std::fwrite(nm ? "PTMESH2" : "PTMESH1", 1, 8, f) == 8 && std::fwrite(&nv, 4, 1, f) == 1 && std::fwrite(&nt, 4, 1, f) == 1;
    if (nm) ok = ok && std::fwrite(&nm, 4, 1, f) == 1;
    ok = ok && std::fwrite(m->verts.data(), 4, m->verts.size(), f) == m->verts.size() &&
         std::fwrite(m->tris.data(), 4, m->tris.size(), f) == m->tris.size();
    if (nm) ok = ok && std::fwrite(m->mats.data(), sizeof(pth_material), nm, f) == nm && std::fwrite(m->tri_mat.data(), 4, nt, f) == nt;
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) { set_err(std::string("write failed: ") + path); return -1; }
    return 0;
}

size_t pth_mesh_n_verts(const pth_mesh* m) { return m ? m->verts.size() / 3 : 0; }
size_t pth_mesh_n_tris(const pth_mesh* m) { return m ? m->tris.size() / 3 : 0; }
const float* pth_mesh_verts(const pth_mesh* m) { return m ? m->verts.data() : nullptr; }
const int32_t* pth_mesh_tris(const pth_mesh* m) { return m ? m->tris.data() : nullptr; }
void pth_mesh_bounds(const pth_mesh* m, float lo[3], float hi[3]) {
    Box b;
    for (size_t i = 0; i + 2 < m->verts.size(); i += 3) b.grow(Vec3{m->verts[i], m->verts[i + 1], m->verts[i + 2]});
    lo[0] = b.lo.x; lo[1] = b.lo.y; lo[2] = b.lo.z;
    hi[0] = b.hi.x; hi[1] = b.hi.y; hi[2] = b.hi.z;
}
void pth_mesh_free(pth_mesh* m) { delete m; }

pth_bvh* pth_bvh_build(const pth_mesh* mesh, const pth_build_params* params) {
    if (!mesh || mesh->tris.empty()) { set_err("empty mesh"); return nullptr; }
    pth_build_params P;
    if (params) P = *params; else pth_default_build_params(&P);
    if (P.max_depth <= 0 || P.max_depth > 64) P.max_depth = 64;
    if (P.min_leaf_size < 1) P.min_leaf_size = 1;
    if (P.max_leaf_size < P.min_leaf_size) P.max_leaf_size = P.min_leaf_size;
    auto t0 = std::chrono::steady_clock::now();

    Builder b(*mesh, P);
    const size_t nt = mesh->tris.size() / 3;
    std::vector<Ref> refs;
    refs.reserve(nt);
    Box root;
    for (size_t i = 0; i < nt; i++) {
        Ref r;
        r.tri = (int)i;
        for (int k = 0; k < 3; k++) r.box.grow(b.vert((int)i, k));
        root.grow(r.box);
        refs.push_back(r);
    }
    b.min_overlap = root.area() * std::max(0.f, P.split_alpha);

    std::unique_ptr<Node> tree;
#pragma omp parallel
#pragma omp single
    tree = b.build(refs, root, 0, true);

    double opt_before = 0.0, opt_after = 0.0;
    if (P.optimize_passes > 0 && !tree->leaf()) {
        OptBridge ob;
        ob.from(tree.get());
        opt_before = opt_after = ob.T.cost();
        const size_t n_nodes = ob.T.box.size();
        (void)pttreeopt::optimise(ob.T, P.optimize_passes);
        opt_after = ob.T.cost();
        if (pttreeopt::intact(ob.T, n_nodes) && ob.T.depth() <= (uint32_t)P.max_depth) tree = ob.to_nodes();
        else opt_after = opt_before;   // keep the builder's tree (the Compact layout caps the depth)
    }

    if (tree->leaf()) {  // wrap a root leaf (CudaBVH.cpp:141 asserts)
        auto top = std::make_unique<Node>();
        top->box = tree->box;
        auto empty = std::make_unique<Node>();  // inverted box: never passes the slab test
        top->child[0] = std::move(tree);
        top->child[1] = std::move(empty);
        tree = std::move(top);
    }

    auto* out = new pth_bvh();
    FlattenCtx fc{out->nodes, out->tris, out->index, *mesh};
    flatten(fc, tree.get());
    auto t1 = std::chrono::steady_clock::now();
    out->stats.n_inner = fc.n_inner;
    out->stats.n_leaves = fc.n_leaves;
    out->stats.n_tri_refs = fc.n_refs;
    out->stats.max_depth = fc.max_depth;
    out->stats.sah_cost = (float)fc.sah;
    out->stats.build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    out->stats.opt_cost_before = (float)opt_before;
    out->stats.opt_cost_after = (float)opt_after;
    return out;
}

const float* pth_bvh_nodes(const pth_bvh* b) { return b->nodes.data(); }
size_t pth_bvh_n_node_vec4(const pth_bvh* b) { return b->nodes.size() / 4; }
const float* pth_bvh_tris(const pth_bvh* b) { return b->tris.data(); }
size_t pth_bvh_n_tri_vec4(const pth_bvh* b) { return b->tris.size() / 4; }
const int32_t* pth_bvh_index(const pth_bvh* b) { return b->index.data(); }
size_t pth_bvh_n_index(const pth_bvh* b) { return b->index.size(); }
void pth_bvh_get_stats(const pth_bvh* b, pth_bvh_stats* out) { *out = b->stats; }
void pth_bvh_free(pth_bvh* b) { delete b; }

// ---- image output -----------------------------------------------------------------------
static void rgb_row(const uint32_t* row, int W, std::vector<unsigned char>& out) {
    for (int x = 0; x < W; x++) {
        const uint32_t w = row[x];
        out.push_back((unsigned char)(w & 255)); out.push_back((unsigned char)((w >> 8) & 255)); out.push_back((unsigned char)((w >> 16) & 255));
    }
}

int pth_write_ppm(const char* path, const uint32_t* rgba, int W, int H) {
    if (!path || !rgba || W <= 0 || H <= 0) { set_err("pth_write_ppm: bad argument"); return -1; }
    FILE* f = std::fopen(path, "wb");
    if (!f) { set_err(std::string("cannot write ") + path); return -1; }
    std::fprintf(f, "P6\n%d %d\n255\n", W, H);
    std::vector<unsigned char> row;
    bool ok = true;
    for (int y = H - 1; y >= 0 && ok; y--) {
        row.clear();
        rgb_row(rgba + (size_t)y * W, W, row);
        ok = std::fwrite(row.data(), 1, row.size(), f) == row.size();
    }
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) { set_err(std::string("write failed: ") + path); return -1; }
    return 0;
}

static uint32_t crc32_of(const unsigned char* p, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 255] ^ (crc >> 8);
    return ~crc;
}
static void be32(std::vector<unsigned char>& v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((unsigned char)(x >> s)); }
static bool png_chunk(FILE* f, const char* type, const std::vector<unsigned char>& data) {
    std::vector<unsigned char> buf;
    be32(buf, (uint32_t)data.size());
    buf.insert(buf.end(), type, type + 4);
    buf.insert(buf.end(), data.begin(), data.end());
    const uint32_t crc = crc32_of(buf.data() + 4, buf.size() - 4);
    be32(buf, crc);
    return std::fwrite(buf.data(), 1, buf.size(), f) == buf.size();
}

int pth_write_png(const char* path, const uint32_t* rgba, int W, int H) {
    if (!path || !rgba || W <= 0 || H <= 0) { set_err("pth_write_png: bad argument"); return -1; }
    // raw scanlines (filter byte 0 + RGB), top row first
    std::vector<unsigned char> raw;
    raw.reserve((size_t)H * (3 * (size_t)W + 1));
    for (int y = H - 1; y >= 0; y--) { raw.push_back(0); rgb_row(rgba + (size_t)y * W, W, raw); }
    // zlib stream of stored deflate blocks (<= 65535 bytes each) + adler32
    std::vector<unsigned char> z{0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n == raw.size() ? 1 : 0);
        z.push_back((unsigned char)(n & 255)); z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 255)); z.push_back((unsigned char)((~n >> 8) & 255));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = off; i < off + n; i++) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
        off += n;
    }
    be32(z, (b << 16) | a);
    FILE* f = std::fopen(path, "wb");
    if (!f) { set_err(std::string("cannot write ") + path); return -1; }
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<unsigned char> hdr;
    be32(hdr, (uint32_t)W); be32(hdr, (uint32_t)H);
    hdr.insert(hdr.end(), {8, 2, 0, 0, 0});  // 8 bit, truecolour
    bool ok = std::fwrite(sig, 1, 8, f) == 8 && png_chunk(f, "IHDR", hdr) && png_chunk(f, "IDAT", z) && png_chunk(f, "IEND", {});
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) { set_err(std::string("write failed: ") + path); return -1; }
    return 0;
}

int pth_write_pfm(const char* path, const float* accum, int W, int H) {
    if (!path || !accum || W <= 0 || H <= 0) { set_err("pth_write_pfm: bad argument"); return -1; }
    FILE* f = std::fopen(path, "wb");
    if (!f) { set_err(std::string("cannot write ") + path); return -1; }
    std::fprintf(f, "PF\n%d %d\n-1.0\n", W, H);   // negative scale = little endian; rows bottom-up, as stored
    const size_t n = (size_t)W * H * 3;
    bool ok = std::fwrite(accum, 4, n, f) == n;
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) { set_err(std::string("write failed: ") + path); return -1; }
    return 0;
}

int pth_checkpoint_save(const char* path, const pth_checkpoint_info* info, const float* accum) {
    if (!path || !info || !accum || info->width <= 0 || info->height <= 0) { set_err("pth_checkpoint_save: bad argument"); return -1; }
    const std::string tmp = std::string(path) + ".tmp";   // write-then-rename: a killed run never leaves half a file
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) { set_err(std::string("cannot write ") + tmp); return -1; }
    const size_t n = (size_t)info->width * info->height * 3;
    const uint32_t crc = crc32_of((const unsigned char*)accum, n * 4);
    bool ok = std::fwrite("PTCKPT1", 1, 8, f) == 8 && std::fwrite(info, sizeof *info, 1, f) == 1 && std::fwrite(&crc, 4, 1, f) == 1 &&
              std::fwrite(accum, 4, n, f) == n;
    ok = (std::fclose(f) == 0) && ok;
    if (!ok || std::rename(tmp.c_str(), path) != 0) { set_err(std::string("write failed: ") + path); return -1; }
    return 0;
}

int pth_checkpoint_load(const char* path, pth_checkpoint_info* info, float* accum) {
    if (!path || !info) { set_err("pth_checkpoint_load: bad argument"); return -1; }
    FILE* f = std::fopen(path, "rb");
    if (!f) { set_err(std::string("cannot open ") + path); return -1; }
    char magic[8];
    uint32_t crc = 0;
    bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "PTCKPT1", 8) == 0 && std::fread(info, sizeof *info, 1, f) == 1 &&
              std::fread(&crc, 4, 1, f) == 1 && info->width > 0 && info->height > 0;
    if (ok && accum) {
        const size_t n = (size_t)info->width * info->height * 3;
        ok = std::fread(accum, 4, n, f) == n && crc32_of((const unsigned char*)accum, n * 4) == crc;
    }
    std::fclose(f);
    if (!ok) { set_err(std::string("bad or damaged checkpoint ") + path); return -1; }
    return 0;
}

uint64_t pth_frame_hash(uint64_t key) {
    key = (~key) + (key << 21);
    key ^= key >> 24;
    key += (key << 3) + (key << 8);
    key ^= key >> 14;
    key += (key << 2) + (key << 4);
    key ^= key >> 28;
    key += key << 31;
    return key;
}

}  // extern "C"

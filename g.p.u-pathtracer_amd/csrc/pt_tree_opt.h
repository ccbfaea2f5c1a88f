// pt_tree_opt.h — insertion-based optimisation of a finished binary BVH (Bittner, Hapala, Havran 2013, "Fast insertion-based
// optimization of bounding volume hierarchies").  Host-only C++, shared by libpthost (pth_build_params::optimize_passes: the
// host builder's own tree) and libptmi (PT_OPT_OPTIMIZE: ANY uploaded hierarchy, e.g. the reference's SplitBVHBuilder output,
// after its leaves have been cut to PT_OPT_LEAF_MAX).  An EXTENSION: the reference walks its builder's tree as built.
//
// Every node, largest area first, is taken out of the tree with its subtree (its parent goes with it, the sibling moves up) and
// put back where the tree's surface-area cost grows least — found by a branch-and-bound search over the whole tree that
// descends by induced cost (the growth of the ancestors' boxes) and prunes with induced + area(node) as the lower bound.
// Putting it back beside its old sibling is one of the candidates, so the cost never rises.  Leaves stay what they are and keep
// their (possibly clipped) boxes; ancestors' boxes are unions of their children's, so every hit stays reachable: the closest
// hit of any ray is unchanged (tests/test_host_and_abi.py: the oracle's walk over optimised trees == brute force).
// Measured (profiles/r03_tree_opt.txt), two passes at upload: cornell_dragon_800k host SBVH tree: area cost in node visits 14.4 -> 12.6,
// bench step 9.65 -> 9.00 ms (the device's PLOC tree: 12.75 / 9.26 ms); built without spatial splits 11.7 / 8.87 ms;
// cornell_dragon-100k 7.33 -> 7.05 ms, gto_sixteen 6.36 -> 6.31, dragon 5.31 -> 5.34 (flat).
#pragma once
#include <algorithm>
#include <cstdint>
#include <limits>
#include <vector>

namespace pttreeopt {

struct Box6 {
    float lo[3], hi[3];
    void grow(const Box6& b) {
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); }
    }
    bool same(const Box6& b) const {
        for (int a = 0; a < 3; a++) if (lo[a] != b.lo[a] || hi[a] != b.hi[a]) return false;
        return true;
    }
    float area() const {
        if (!(lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2])) return 0.f;   // an empty (inverted) box
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return 2.f * (dx * dy + dy * dz + dz * dx);
    }
};

// nodes by index; c0 < 0: a leaf (its payload is the caller's business: indices never change, only parent / child links and boxes)
struct Tree {
    std::vector<Box6> box;
    std::vector<float> area;
    std::vector<int> parent, c0, c1;
    int root = 0;

    int add(const Box6& b, int par) {
        const int i = (int)box.size();
        box.push_back(b); area.push_back(b.area()); parent.push_back(par); c0.push_back(-1); c1.push_back(-1);
        return i;
    }
    bool leaf(int i) const { return c0[i] < 0; }
    void replace_child(int p, int old_c, int new_c) { if (c0[p] == old_c) c0[p] = new_c; else c1[p] = new_c; }
    void refit_up(int i) {
        while (i >= 0) {
            Box6 b = box[c0[i]];
            b.grow(box[c1[i]]);
            if (b.same(box[i])) break;
            box[i] = b; area[i] = b.area();
            i = parent[i];
        }
    }
    double cost() const {   // sum of the inner nodes' areas / root area
        double s = 0.0;
        std::vector<int> st{root};
        while (!st.empty()) {
            const int i = st.back();
            st.pop_back();
            if (leaf(i)) continue;
            s += area[i];
            st.push_back(c0[i]);
            st.push_back(c1[i]);
        }
        return s / std::max(1e-30f, area[root]);
    }
    uint32_t depth() const {   // edges on the longest root-to-leaf path
        uint32_t d = 0;
        std::vector<std::pair<int, uint32_t>> st{{root, 0u}};
        while (!st.empty()) {
            const std::pair<int, uint32_t> it = st.back();
            st.pop_back();
            if (leaf(it.first)) { d = std::max(d, it.second); continue; }
            st.push_back({c0[it.first], it.second + 1});
            st.push_back({c1[it.first], it.second + 1});
        }
        return d;
    }
};

// one pass over every node but the root and its children; returns how many found a better place.  max_visits caps one search.
inline size_t reinsertion_pass(Tree& T, size_t max_visits = 4096) {
    const int n = (int)T.box.size();
    std::vector<int> order;
    order.reserve((size_t)n);
    for (int i = 0; i < n; i++) if (i != T.root) order.push_back(i);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return T.area[a] > T.area[b] || (T.area[a] == T.area[b] && a < b); });
    struct Cand { float induced; int node; };
    auto worse = [](const Cand& a, const Cand& b) { return a.induced > b.induced || (a.induced == b.induced && a.node > b.node); };
    std::vector<Cand> heap;
    size_t moved = 0;
    for (int L : order) {
        const int P = T.parent[L];
        if (P < 0) continue;                        // became the root meanwhile
        const int S = T.c0[P] == L ? T.c1[P] : T.c0[P];
        const int G = T.parent[P];
        if (G < 0) continue;                        // a child of the root: the top stays
        T.replace_child(G, P, S);                   // take L (and P) out: S moves up
        T.parent[S] = G;
        T.refit_up(G);
        const Box6 lb = T.box[L];
        const float la = T.area[L];
        float best_cost = std::numeric_limits<float>::max();
        int best = S;
        heap.clear();
        heap.push_back({0.f, T.root});
        size_t visits = 0;
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end(), worse);
            const Cand cnd = heap.back();
            heap.pop_back();
            if (cnd.induced + la >= best_cost) break;   // the heap is ordered by induced cost: nothing left can beat the best
            Box6 u = T.box[cnd.node];
            u.grow(lb);
            const float total = cnd.induced + u.area();
            if (total < best_cost) { best_cost = total; best = cnd.node; }
            if (++visits >= max_visits) break;
            const float ind_child = total - T.area[cnd.node];
            if (!T.leaf(cnd.node) && ind_child + la < best_cost) {
                heap.push_back({ind_child, T.c0[cnd.node]});
                std::push_heap(heap.begin(), heap.end(), worse);
                heap.push_back({ind_child, T.c1[cnd.node]});
                std::push_heap(heap.begin(), heap.end(), worse);
            }
        }
        const int X = best, XP = T.parent[X];          // put it back: P becomes the parent of (X, L)
        if (XP >= 0) T.replace_child(XP, X, P); else T.root = P;
        T.parent[P] = XP;
        T.c0[P] = X; T.c1[P] = L;
        T.parent[X] = P; T.parent[L] = P;
        Box6 pb = T.box[X];
        pb.grow(lb);
        T.box[P] = pb; T.area[P] = pb.area();
        T.refit_up(XP);
        if (X != S) moved++;
    }
    return moved;
}

// (The paper's finer move — dissolve an inner node and re-insert its two children separately — was tried on top of these passes:
// cornell_dragon_800k 24.13 -> 23.86 in inner-node area cost after two more passes, and not monotone; not kept.)

}  // namespace pttreeopt

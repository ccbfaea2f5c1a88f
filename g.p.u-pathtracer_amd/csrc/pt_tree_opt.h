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
// Measured (profiles/r03_tree_opt.txt), area cost of the 4-wide tree in node visits / bench step: cornell_dragon_800k host SAH tree
// without spatial splits 14.6 / 9.95 ms -> 11.7 / 8.8 ms (four passes, 2.5 s on the GPU box's host); the device's PLOC tree
// 12.75 / 9.26 -> 11.9 / 9.1; host SBVH tree 14.4 / 9.65 -> 13.1 / 9.3 (12.6 / 9.0 with strictly serial passes);
// cornell_dragon-100k 7.33 -> 7.05 ms, gto_sixteen 6.36 -> 6.31, dragon flat; the 6.4 M-triangle scene 38.6 / 19.6 -> 34.5 / 17.5 ms.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <limits>
#include <thread>
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

// ---------------------------------------------------------------------------------------------------------------------------------
// A pass runs in parallel batches.  The search is the expensive part and reads only: for a batch of nodes (consecutive in the
// area order; small batches for the large nodes at the top, growing towards the leaves) every thread finds its node's best place
// on the tree AS IT STANDS, with the node taken out VIRTUALLY — the boxes of its ancestors are re-fitted in a side table (at most
// one entry per level) and its parent stands for the sibling that would move up.  The moves are then carried out one after the
// other, each priced again on the tree as it is by then against staying put (apply_move), so the cost never rises.  Measured
// against the strictly serial pass (take out, search, put back, node by node: 5 s per pass and million nodes), 800 k scene,
// inner-node area cost from 30.66: serial 24.94 / 24.13 after 1 / 2 passes; batched 26.27 / 24.51 / 24.05 after 1 / 2 / 3 passes at
// 1.3 s per pass on 8 threads — three batched passes replace two serial ones.
struct SearchScratch {
    std::vector<uint8_t> mark;         // node -> index into path (255 = not on the path)
    int path_node[72];
    Box6 path_box[72];
    float path_area[72];
    int n_path = 0;
    struct Cand { float induced; int node; };
    std::vector<Cand> heap;
};

// best sibling for L with L (and its parent) taken out virtually; -1: leave it
inline int find_place(const Tree& T, int L, size_t max_visits, SearchScratch& sc) {
    const int P = T.parent[L];
    if (P < 0) return -1;
    const int G = T.parent[P];
    if (G < 0) return -1;
    const int S = T.c0[P] == L ? T.c1[P] : T.c0[P];
    if (sc.mark.size() != T.box.size()) sc.mark.assign(T.box.size(), 255);
    // ancestors re-fitted without L: G's child P is replaced by S
    sc.n_path = 0;
    {
        Box6 below = T.box[S];
        int from = P, a = G;
        while (a >= 0 && sc.n_path < 72) {
            const int other = T.c0[a] == from ? T.c1[a] : T.c0[a];
            Box6 b = below;
            b.grow(T.box[other]);
            if (b.same(T.box[a])) break;          // nothing changes from here up
            sc.path_node[sc.n_path] = a; sc.path_box[sc.n_path] = b; sc.path_area[sc.n_path] = b.area();
            sc.mark[(size_t)a] = (uint8_t)sc.n_path;
            sc.n_path++;
            below = b;
            from = a;
            a = T.parent[a];
        }
    }
    auto worse = [](const SearchScratch::Cand& x, const SearchScratch::Cand& y) { return x.induced > y.induced || (x.induced == y.induced && x.node > y.node); };
    const Box6 lb = T.box[L];
    const float la = T.area[L];
    float best_cost = std::numeric_limits<float>::max();
    int best = S;
    sc.heap.clear();
    sc.heap.push_back({0.f, T.root});
    size_t visits = 0;
    while (!sc.heap.empty()) {
        std::pop_heap(sc.heap.begin(), sc.heap.end(), worse);
        SearchScratch::Cand cnd = sc.heap.back();
        sc.heap.pop_back();
        if (cnd.induced + la >= best_cost) break;
        int X = cnd.node == P ? S : cnd.node;             // P is out: its sibling child stands in its place
        const uint8_t m = sc.mark[(size_t)X];
        Box6 u = m != 255 ? sc.path_box[m] : T.box[X];
        const float xa = m != 255 ? sc.path_area[m] : T.area[X];
        u.grow(lb);
        const float total = cnd.induced + u.area();
        if (total < best_cost) { best_cost = total; best = X; }
        if (++visits >= max_visits) break;
        const float ind_child = total - xa;
        if (!T.leaf(X) && ind_child + la < best_cost) {
            sc.heap.push_back({ind_child, T.c0[X]});
            std::push_heap(sc.heap.begin(), sc.heap.end(), worse);
            sc.heap.push_back({ind_child, T.c1[X]});
            std::push_heap(sc.heap.begin(), sc.heap.end(), worse);
        }
    }
    for (int k = 0; k < sc.n_path; k++) sc.mark[(size_t)sc.path_node[k]] = 255;
    return best == S ? -1 : best;
}

// what inserting a subtree with box lb beside X costs on the tree as it stands: area(X u lb) + the growth of X's ancestors
inline float insertion_cost(const Tree& T, int X, const Box6& lb) {
    Box6 u = T.box[X];
    u.grow(lb);
    float cost = u.area();
    for (int a = T.parent[X]; a >= 0; a = T.parent[a]) {
        Box6 g = T.box[a];
        g.grow(lb);
        const float grown = g.area() - T.area[a];
        if (!(grown > 0.f)) break;                  // this ancestor already holds lb: so do all above it
        cost += grown;
    }
    return cost;
}

// carries out one move found by find_place on an EARLIER state of the tree: L is taken out for real, the target X is priced again on
// the tree as it is NOW against putting L back beside its old sibling, and L goes where it is cheaper — a stale move can no longer
// raise the cost.  Returns 1 when L ended up somewhere new, 0 when there was nothing to do, -1 when the move had gone stale (the
// caller may search again on the tree as it is now).
inline int apply_move(Tree& T, int L, int X) {
    const int P = T.parent[L];
    if (P < 0 || X < 0 || X == L) return 0;
    const int G = T.parent[P];
    if (G < 0) return 0;
    const int S = T.c0[P] == L ? T.c1[P] : T.c0[P];
    if (X == P) X = S;
    if (X == S) return 0;
    for (int a = X; a >= 0; a = T.parent[a]) if (a == L) return -1;   // the target sits inside the subtree that moves
    T.replace_child(G, P, S);
    T.parent[S] = G;
    T.refit_up(G);
    const Box6 lb = T.box[L];
    const bool better = insertion_cost(T, X, lb) < insertion_cost(T, S, lb);
    if (!better) X = S;
    const int XP = T.parent[X];
    if (XP >= 0) T.replace_child(XP, X, P); else T.root = P;
    T.parent[P] = XP;
    T.c0[P] = X; T.c1[P] = L;
    T.parent[X] = P; T.parent[L] = P;
    Box6 pb = T.box[X];
    pb.grow(lb);
    T.box[P] = pb; T.area[P] = pb.area();
    T.refit_up(XP);
    return better ? 1 : -1;
}

// run(n_items, fn(begin, end, thread)) must call fn over disjoint ranges covering [0, n_items) from up to n_threads threads and
// return when all are done (`optimise` below brings std::thread workers)
template <class ParallelFor>
inline size_t reinsertion_pass_batched(Tree& T, int n_threads, ParallelFor&& run, size_t max_visits = 4096) {
    const int n = (int)T.box.size();
    std::vector<int> order;
    order.reserve((size_t)n);
    for (int i = 0; i < n; i++) if (i != T.root) order.push_back(i);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return T.area[a] > T.area[b] || (T.area[a] == T.area[b] && a < b); });
    std::vector<SearchScratch> scratch((size_t)std::max(1, n_threads));
    std::vector<int> target(order.size(), -1);
    size_t moved = 0;
    // small batches at the top of the order (the large nodes, whose moves change the most), growing towards the leaves
    size_t begin = 0, batch = 256;
    while (begin < order.size()) {
        const size_t end = std::min(order.size(), begin + batch);
        run(end - begin, [&](size_t b, size_t e, int thread) {
            SearchScratch& sc = scratch[(size_t)thread];
            for (size_t k = b; k < e; k++) target[begin + k] = find_place(T, order[begin + k], max_visits, sc);
        });
        for (size_t k = begin; k < end; k++) {
            if (target[k] < 0) continue;
            int r = apply_move(T, order[k], target[k]);
            if (r < 0) r = apply_move(T, order[k], find_place(T, order[k], max_visits, scratch[0]));   // gone stale: once more, on the tree as it is now
            if (r > 0) moved++;
        }
        begin = end;
        batch = std::min<size_t>(batch * 2, 32768);
    }
    return moved;
}

// structural check of a tree that was re-arranged: every node reached exactly once from the root, parent / child links consistent
inline bool intact(const Tree& T, size_t n_nodes_expected) {
    std::vector<uint8_t> seen(T.box.size(), 0);
    std::vector<int> st{T.root};
    size_t n = 0;
    if (T.root < 0 || (size_t)T.root >= T.box.size() || T.parent[T.root] != -1) return false;
    while (!st.empty()) {
        const int i = st.back();
        st.pop_back();
        if (i < 0 || (size_t)i >= T.box.size() || seen[(size_t)i]) return false;
        seen[(size_t)i] = 1;
        n++;
        if (T.leaf(i)) continue;
        const int a = T.c0[i], b = T.c1[i];
        if (a < 0 || b < 0 || (size_t)a >= T.box.size() || (size_t)b >= T.box.size() || T.parent[a] != i || T.parent[b] != i) return false;
        st.push_back(a);
        st.push_back(b);
    }
    return n == n_nodes_expected;
}

// n passes with the machine's threads (std::thread: no OpenMP runtime needed where this header goes); the result is the same for
// any number of threads (searches read a frozen tree, moves are carried out in order)
inline size_t optimise(Tree& T, int passes, size_t max_visits = 4096) {
    const unsigned hw = std::thread::hardware_concurrency();
    const int n_threads = (int)std::min<unsigned>(std::max<unsigned>(hw, 1u), 64u);
    auto run = [&](size_t n_items, auto fn) {
        const size_t chunk = 16;
        if (n_threads <= 1 || n_items <= 4 * chunk) { fn((size_t)0, n_items, 0); return; }
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        th.reserve((size_t)n_threads);
        for (int t = 0; t < n_threads; t++)
            th.emplace_back([&, t] {
                for (;;) {
                    const size_t b = next.fetch_add(chunk);
                    if (b >= n_items) break;
                    fn(b, std::min(n_items, b + chunk), t);
                }
            });
        for (std::thread& x : th) x.join();
    };
    size_t moved_total = 0;
    for (int p = 0; p < passes; p++) {
        const size_t moved = reinsertion_pass_batched(T, n_threads, run, max_visits);
        moved_total += moved;
        if (moved == 0) break;
    }
    return moved_total;
}

// (The paper's finer move — dissolve an inner node and re-insert its two children separately — was tried on top of these passes:
// cornell_dragon_800k 24.13 -> 23.86 in inner-node area cost after two more passes, and not monotone; not kept.)

}  // namespace pttreeopt

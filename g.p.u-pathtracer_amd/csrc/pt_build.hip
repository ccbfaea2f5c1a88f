// pt_build.hip — pt_build_bvh's driver: the launches of the device BVH builder (kernels in pt_build.h).
// One translation unit of libptmi.so (pt_ctx.h).
#include <cmath>
#include <cstring>

#include "pt_ctx.h"
#include "pt_build.h"

namespace ptmi {

int build_bvh_impl(pt_ctx* c, const float* verts, size_t n_verts, const int32_t* tris, size_t n_tris, int algo, bool* too_deep,
                          const int32_t* id_map) {
    *too_deep = false;
    if (!verts || !tris || n_verts == 0 || n_tris == 0) return fail(c, PT_ERR_INVALID, "pt_build_bvh: empty mesh or null array");
    if (n_tris > (1u << 27) || n_verts > (1u << 30)) return fail(c, PT_ERR_INVALID, "pt_build_bvh: mesh too large for 32-bit links");
    if (c->opt_tri_test == 1) return fail(c, PT_ERR_UNSUPPORTED, "pt_build_bvh: Woop records are made by the host path only (pt_upload_bvh)");
    for (size_t i = 0; i < 3 * n_tris; i++)
        if (tris[i] < 0 || (size_t)tris[i] >= n_verts) return fail(c, PT_ERR_INVALID, "pt_build_bvh: vertex index out of range");
    for (size_t i = 0; i < 3 * n_verts; i++)
        if (!(std::fabs(verts[i]) <= 3.0e38f)) return fail(c, PT_ERR_INVALID, "pt_build_bvh: non-finite vertex");
    if (c->d_tri_matid && n_tris > c->n_tri_matid)
        return fail(c, PT_ERR_INVALID, "pt_build_bvh: the triangle-material array on this context does not cover this mesh (clear or re-upload it first)");

    // a lone triangle is doubled: the hierarchy needs two leaves (both report id 0)
    std::vector<int32_t> two;
    int n = (int)std::max<size_t>(n_tris, 2);
    if (n_tris == 1) { two.assign(tris, tris + 3); two.insert(two.end(), tris, tris + 3); tris = two.data(); }

    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    DevTemp tmp;
    BuildArrays B;
    std::memset(&B, 0, sizeof B);
    B.n_orig = (int)n_tris;
    B.leaf_max = std::max(1, c->opt_leaf_max ? c->opt_leaf_max : 1);
    hipStream_t st = c->stream;
    float* d_verts = nullptr;
    int* d_tris_idx = nullptr;
    HIP_TRY(c, tmp.get(&d_verts, 3 * n_verts));
    HIP_TRY(c, tmp.get(&d_tris_idx, 3 * (size_t)n));
    HIP_TRY(c, hipMemcpyAsync(d_verts, verts, 3 * n_verts * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(d_tris_idx, tris, 3 * (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    hipEvent_t e0, e1;
    HIP_TRY(c, hipEventCreate(&e0));
    HIP_TRY(c, hipEventCreate(&e1));
    struct EvGuard { hipEvent_t a, b; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evg{e0, e1};
    HIP_TRY(c, hipEventRecord(e0, st));
    if (c->opt_presplit && n_tris >= 64) {
        // pre-splitting: long triangles enter as several primitives (pt_build.h); the target length is a
        // multiple of the edge a triangle would have if the n of them tiled a square of the scene's diagonal
        float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
        for (size_t i = 0; i < 3 * n_verts; i++) { lo[i % 3] = std::min(lo[i % 3], verts[i]); hi[i % 3] = std::max(hi[i % 3], verts[i]); }
        const float diag = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
        const float target = 0.01f * (float)c->opt_presplit * diag / std::sqrt((float)n_tris);
        int *d_cnt = nullptr, *d_off = nullptr;
        HIP_TRY(c, tmp.get(&d_cnt, n_tris));
        HIP_TRY(c, tmp.get(&d_off, n_tris));
        const dim3 g0((unsigned)((n_tris + PTB_BLOCK - 1) / PTB_BLOCK));
        hipLaunchKernelGGL(k_split_count, g0, dim3(PTB_BLOCK), 0, st, d_verts, d_tris_idx, (int)n_tris, target, d_cnt);
        size_t sb = 0;
        HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(nullptr, sb, d_cnt, d_off, (int)n_tris, st));
        char* stmp = nullptr;
        HIP_TRY(c, tmp.get(&stmp, sb));
        HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(stmp, sb, d_cnt, d_off, (int)n_tris, st));
        int last_off = 0, last_cnt = 0;
        HIP_TRY(c, hipMemcpyAsync(&last_off, d_off + (n_tris - 1), sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(&last_cnt, d_cnt + (n_tris - 1), sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        const long n_ref = (long)last_off + last_cnt;
        if (n_ref > (long)n_tris && n_ref < (1l << 27)) {
            int* d_ref = nullptr;
            HIP_TRY(c, tmp.get(&B.tbox, 6 * (size_t)n_ref));
            HIP_TRY(c, tmp.get(&d_ref, (size_t)n_ref));
            hipLaunchKernelGGL(k_split_emit, g0, dim3(PTB_BLOCK), 0, st, d_verts, d_tris_idx, (int)n_tris, target, d_off, B.tbox, d_ref);
            HIP_TRY(c, hipGetLastError());
            B.ref_tri = d_ref;
            n = (int)n_ref;
        }
    }
    B.n = n;
    if (!B.tbox) HIP_TRY(c, tmp.get(&B.tbox, 6 * (size_t)n));
    HIP_TRY(c, tmp.get(&B.cbounds, 6));
    HIP_TRY(c, tmp.get(&B.key_in, (size_t)n));
    HIP_TRY(c, tmp.get(&B.key, (size_t)n));
    HIP_TRY(c, tmp.get(&B.val_in, (size_t)n));
    HIP_TRY(c, tmp.get(&B.val, (size_t)n));
    HIP_TRY(c, tmp.get(&B.left, (size_t)n));
    HIP_TRY(c, tmp.get(&B.right, (size_t)n));
    HIP_TRY(c, tmp.get(&B.first, (size_t)n));
    HIP_TRY(c, tmp.get(&B.last, (size_t)n));
    HIP_TRY(c, tmp.get(&B.parent_i, (size_t)n));
    HIP_TRY(c, tmp.get(&B.parent_l, (size_t)n));
    HIP_TRY(c, tmp.get(&B.nbox, 6 * (size_t)n));
    HIP_TRY(c, tmp.get(&B.arrive, (size_t)n));
    HIP_TRY(c, tmp.get(&B.stats, 4));
    HIP_TRY(c, tmp.get(&B.level_cnt, 68));
    HIP_TRY(c, tmp.get(&B.frontier_a, (size_t)n));
    HIP_TRY(c, tmp.get(&B.frontier_b, (size_t)n));
    B.verts = d_verts;
    B.tris = d_tris_idx;
    if (id_map) {
        int* d_map = nullptr;
        HIP_TRY(c, tmp.get(&d_map, n_tris));
        HIP_TRY(c, hipMemcpyAsync(d_map, id_map, n_tris * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        B.id_map = d_map;
    }
    const size_t n_items = (size_t)(n - 1) + (size_t)n + (size_t)(n - 1);   // binary, records, wide (upper bound)
    if (n_items * 4 >= (size_t)PT_SENTINEL) return fail(c, PT_ERR_INVALID, "pt_build_bvh: scene too large for 32-bit links");
    float4* items = nullptr;
    HIP_TRY(c, hipMalloc((void**)&items, n_items * 64));
    B.items = items;
    struct ItemsGuard { float4* p; ~ItemsGuard() { if (p) (void)hipFree(p); } } guard{items};

    const unsigned int cb0[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    const unsigned int st0[4] = {1u, 0u, 0u, 0u};   // wide slot 0 is the root's
    HIP_TRY(c, hipMemcpyAsync(B.cbounds, cb0, sizeof cb0, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(B.stats, st0, sizeof st0, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemsetAsync(B.level_cnt, 0, 68 * sizeof(unsigned int), st));
    const unsigned int one = 1u;
    HIP_TRY(c, hipMemcpyAsync(B.level_cnt, &one, sizeof one, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemsetAsync(items, 0, n_items * 64, st));

    const dim3 blk(PTB_BLOCK), grd((unsigned)((n + PTB_BLOCK - 1) / PTB_BLOCK));
    hipLaunchKernelGGL(k_tri_bounds, grd, blk, 0, st, B);
    hipLaunchKernelGGL(k_morton, grd, blk, 0, st, B);
    size_t cub_bytes = 0;
    HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, cub_bytes, B.key_in, B.key, B.val_in, B.val, n, 0, 63, st));
    char* cub_tmp = nullptr;
    HIP_TRY(c, tmp.get(&cub_tmp, cub_bytes));
    HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(cub_tmp, cub_bytes, B.key_in, B.key, B.val_in, B.val, n, 0, 63, st));
    unsigned int n_levels_max = 64;
    const bool ploc = algo == 1 && n > 2;
    if (ploc) {
        // PLOC: rounds of nearest-neighbour search + mutual merges + ordered compaction
        PlocArrays Q;
        std::memset(&Q, 0, sizeof Q);
        HIP_TRY(c, tmp.get(&Q.cl, (size_t)n));
        HIP_TRY(c, tmp.get(&Q.cl_next, (size_t)n));
        HIP_TRY(c, tmp.get(&Q.cbox, 6 * (size_t)n));
        HIP_TRY(c, tmp.get(&Q.nn, (size_t)n));
        HIP_TRY(c, tmp.get(&Q.keep, (size_t)n));
        HIP_TRY(c, tmp.get(&Q.pos, (size_t)n + 1));
        HIP_TRY(c, tmp.get(&Q.ref, (size_t)n));
        size_t scan_bytes = 0;
        HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, Q.keep, Q.pos, n, st));
        char* scan_tmp = nullptr;
        HIP_TRY(c, tmp.get(&scan_tmp, scan_bytes));
        Q.n_c = n;
        hipLaunchKernelGGL(k_ploc_init, grd, blk, 0, st, B, Q);
        HIP_TRY(c, hipMemsetAsync(B.stats + 1, 0, sizeof(unsigned int), st));
        int rounds = 0;
        while (Q.n_c > 1) {
            if (++rounds > 192) { *too_deep = true; return fail(c, PT_ERR_UNSUPPORTED, "pt_build_bvh: PLOC needs too many rounds (degenerate input)"); }
            const dim3 g((unsigned)((Q.n_c + PTB_BLOCK - 1) / PTB_BLOCK));
            hipLaunchKernelGGL(k_ploc_gather, g, blk, 0, st, B, Q);
            hipLaunchKernelGGL(k_ploc_nn, g, blk, 0, st, Q);
            hipLaunchKernelGGL(k_ploc_merge, g, blk, 0, st, B, Q);
            HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, Q.keep, Q.pos, Q.n_c, st));
            hipLaunchKernelGGL(k_ploc_scatter, g, blk, 0, st, Q);
            HIP_TRY(c, hipGetLastError());
            int last_pos = 0, last_keep = 0;
            HIP_TRY(c, hipMemcpyAsync(&last_pos, Q.pos + (Q.n_c - 1), sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipMemcpyAsync(&last_keep, Q.keep + (Q.n_c - 1), sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipStreamSynchronize(st));
            const int next = last_pos + last_keep;
            if (next >= Q.n_c || next < 1) return fail(c, PT_ERR_DEVICE, "pt_build_bvh: PLOC round made no progress");
            Q.n_c = next;
            std::swap(Q.cl, Q.cl_next);
        }
        hipLaunchKernelGGL(k_ploc_parents, grd, blk, 0, st, B);
        hipLaunchKernelGGL(k_node_depth, grd, blk, 0, st, B);
        unsigned int deepest = 0;
        HIP_TRY(c, hipMemcpyAsync(&deepest, B.stats + 3, sizeof deepest, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        if (deepest + 1 > 64) { *too_deep = true; return fail(c, PT_ERR_UNSUPPORTED, "pt_build_bvh: PLOC tree deeper than 64 levels"); }
        n_levels_max = deepest + 1;
        // depth-first leaf order: subtrees become contiguous record ranges, so small ones can be cut into
        // multi-triangle leaves (PT_OPT_LEAF_MAX) exactly as in the LBVH
        B.leaf_max = std::max(1, c->opt_leaf_max ? c->opt_leaf_max : 1);
        int* newpos = nullptr;
        HIP_TRY(c, tmp.get(&newpos, (size_t)n));
        for (unsigned int level = deepest + 1; level-- > 0;) hipLaunchKernelGGL(k_ploc_size, grd, blk, 0, st, B, level);
        for (unsigned int level = 0; level <= deepest; level++) hipLaunchKernelGGL(k_ploc_first, grd, blk, 0, st, B, level, newpos);
        HIP_TRY(c, hipMemcpyAsync(B.val_in, B.val, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_ploc_reorder, grd, blk, 0, st, B, newpos, B.val_in);
        HIP_TRY(c, hipGetLastError());
    } else {
    hipLaunchKernelGGL(k_hierarchy, grd, blk, 0, st, B);
    hipLaunchKernelGGL(k_node_depth, grd, blk, 0, st, B);
    HIP_TRY(c, hipGetLastError());
    {   // bottom-up fit, one launch per level, deepest first
        unsigned int deepest = 0;
        HIP_TRY(c, hipMemcpyAsync(&deepest, B.stats + 3, sizeof deepest, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        if (deepest + 1 > 64) return fail(c, PT_ERR_UNSUPPORTED, "pt_build_bvh: tree deeper than 64 levels (degenerate input); use the host builder");
        for (unsigned int level = deepest + 1; level-- > 0;) hipLaunchKernelGGL(k_fit_level, grd, blk, 0, st, B, level);
        n_levels_max = deepest + 1;
    }
    }
    hipLaunchKernelGGL(k_depth, grd, blk, 0, st, B);
    hipLaunchKernelGGL(k_records, grd, blk, 0, st, B);
    hipLaunchKernelGGL(k_binary, grd, blk, 0, st, B);
    HIP_TRY(c, hipGetLastError());
    // 4-wide collapse, one launch per level of the wide tree; frontier sizes stay on the device, so
    // nothing is read back between levels (a wide level spans at least one binary level: `deepest + 1`
    // launches cover every tree; the empty ones at the end cost a few microseconds each)
    const int2 root_item = make_int2(0, 0);
    HIP_TRY(c, hipMemcpyAsync(B.frontier_a, &root_item, sizeof root_item, hipMemcpyHostToDevice, st));
    {
        int2 *fin = B.frontier_a, *fout = B.frontier_b;
        const unsigned cgrid = (unsigned)std::min<int>((n + PTB_BLOCK - 1) / PTB_BLOCK, 2048);
        for (unsigned int level = 0; level <= n_levels_max; level++) {
            hipLaunchKernelGGL(k_collapse, dim3(cgrid), blk, 0, st, B, fin, fout, (int)level);
            std::swap(fin, fout);
        }
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipEventRecord(e1, st));
    unsigned int stats[4], level_cnt[68];
    HIP_TRY(c, hipMemcpyAsync(stats, B.stats, sizeof stats, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(level_cnt, B.level_cnt, sizeof level_cnt, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    uint32_t levels = 0;
    while (levels < 66 && level_cnt[levels] > 0) levels++;
    if (level_cnt[std::min<unsigned int>(n_levels_max + 1, 67)] != 0) return fail(c, PT_ERR_DEVICE, "pt_build_bvh: wide collapse did not finish");
    HIP_TRY(c, hipEventElapsedTime(&c->build_ms, e0, e1));
    if (stats[3] > 64) return fail(c, PT_ERR_UNSUPPORTED, "pt_build_bvh: tree deeper than 64 levels (degenerate input); use the host builder");

    (void)hipFree(c->d_nodes);
    c->d_nodes = items;
    guard.p = nullptr;
    c->d_tris = c->d_nodes;
    c->records_woop = false;
    c->wide_root = 4 * ((uint64_t)(n - 1) + (uint64_t)n);
    c->wide_top_layout = 1;      // level order below the root, not a breadth-first prefix of fixed size
    c->n_top_layout = 1;
    c->wide_depth = levels;
    c->n_wide = stats[0];
    c->n_inner = (uint64_t)(n - 1);
    c->n_refs = (uint64_t)n;
    c->n_leaves = stats[2];
    c->max_depth = stats[3];
    c->scene_bytes = n_items * 64;
    c->max_tri_id = (int32_t)n_tris - 1;
    if (id_map) for (size_t i = 0; i < n_tris; i++) c->max_tri_id = std::max(c->max_tri_id, id_map[i]);
    c->has_bvh = true;
    c->scene_gen++;
    return PT_OK;
}

}  // namespace ptmi

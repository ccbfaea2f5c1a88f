// ptmi.hip — C ABI of libptmi.so (include/ptmi.h): context, scene upload + gfx950
// re-layout, launches.  The kernels are in pt_kernels.h.
//
// Replaces BasicScene::launchKernel (GpuPathTracer/tracer.cu:405-415) and the device
// buffer set-up of GpuPathTracer/BasicScene.cpp:138-149,:214-215,:297-313.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "pt_ctx.h"
#include "pt_scene_build.h"

static_assert(sizeof(pt_sphere) == 44, "pt_sphere must match the reference Sphere (44 B)");
static_assert(sizeof(pt_sphere_d) == sizeof(pt_sphere), "device sphere mirror");
static_assert(sizeof(pt_params) == 104 && sizeof(pt_camera) == 64 && sizeof(pt_counters) == 48, "ABI struct sizes (tests/test_host_and_abi.py)");

namespace ptmi {
thread_local std::string g_err;
}
using namespace ptmi;

namespace ptmi {
static int tree_cost_impl(pt_ctx* c, double* node_visits, double* tri_tests);

// everything on the context that describes the acceleration structure (PT_OPT_REBUILD 2 holds two for a moment)
struct TreeState {
    float4 *d_nodes, *d_tris;
    bool records_woop, has_bvh;
    uint64_t wide_root, n_wide, n_inner, n_refs, n_leaves, scene_bytes;
    uint32_t wide_top_layout, wide_depth, n_top_layout, max_depth;
    int32_t max_tri_id;
    float build_ms;
    static TreeState of(const pt_ctx* c) {
        return {c->d_nodes, c->d_tris, c->records_woop, c->has_bvh, c->wide_root, c->n_wide, c->n_inner, c->n_refs, c->n_leaves, c->scene_bytes,
                c->wide_top_layout, c->wide_depth, c->n_top_layout, c->max_depth, c->max_tri_id, c->build_ms};
    }
    void restore(pt_ctx* c) const {
        c->d_nodes = d_nodes; c->d_tris = d_tris; c->records_woop = records_woop; c->has_bvh = has_bvh;
        c->wide_root = wide_root; c->n_wide = n_wide; c->n_inner = n_inner; c->n_refs = n_refs; c->n_leaves = n_leaves; c->scene_bytes = scene_bytes;
        c->wide_top_layout = wide_top_layout; c->wide_depth = wide_depth; c->n_top_layout = n_top_layout; c->max_depth = max_depth;
        c->max_tri_id = max_tri_id; c->build_ms = build_ms;
    }
};
}  // namespace ptmi

namespace ptmi {
// PT_KERNEL_AUTO: reads the trials' events once they have all completed (wait = false: only if that needs no waiting) and decides
static bool auto_decide(pt_ctx* c, pt_ctx::AutoPick& a, bool wait) {
    using AP = pt_ctx::AutoPick;
    if (a.phase != AP::PENDING) return a.phase == AP::DECIDED;
    for (int t = 0; t < AP::TRIALS; t++) {
        const hipError_t e = wait ? hipEventSynchronize(a.e[2 * t + 1]) : hipEventQuery(a.e[2 * t + 1]);
        if (e != hipSuccess) { (void)hipGetLastError(); return false; }   // hipErrorNotReady is not an error of the call
    }
    float best[2] = {3.0e38f, 3.0e38f};
    bool ok = true;
    for (int t = 0; t < AP::TRIALS; t++) {
        float ms = 0.f;
        ok = ok && hipEventElapsedTime(&ms, a.e[2 * t], a.e[2 * t + 1]) == hipSuccess;
        best[t & 1] = std::min(best[t & 1], ms);
    }
    a.ms[0] = best[0]; a.ms[1] = best[1];
    a.choice = ok && a.ms[1] < a.ms[0] ? PT_KERNEL_WAVEFRONT : PT_KERNEL_PERSISTENT;
    a.phase = AP::DECIDED;
    bool wave_wanted = false;   // by any remembered configuration
    for (const AP& o : c->picks)
        if (o.key && (o.phase < AP::DECIDED || o.choice == PT_KERNEL_WAVEFRONT)) wave_wanted = true;
    if (!wave_wanted && c->d_wave) {   // the pipeline's path records (3 GB at 1080p x 16 spp) are not needed; the trials that used them are done
        (void)hipFree(c->d_wave);
        c->d_wave = nullptr;
        c->wave_bytes = 0;
    }
    return true;
}

int stage_mark(pt_ctx* c, int kind) {
    if (!c->opt_timing) return PT_OK;
    if (c->stage_used == c->stage_ev.size()) {
        hipEvent_t e = nullptr;
        HIP_TRY(c, hipEventCreate(&e));
        c->stage_ev.push_back(e);
        c->stage_kind.push_back(0);
    }
    HIP_TRY(c, hipEventRecord(c->stage_ev[c->stage_used], c->stream));
    c->stage_kind[c->stage_used] = kind;   // kind of the work that ENDS at this event
    c->stage_used++;
    return PT_OK;
}
}  // namespace ptmi


namespace ptmi {
// refine (leaves of at most PT_OPT_LEAF_MAX references), optimise (PT_OPT_OPTIMIZE), emit, upload: the tree `X` becomes the context's
static int install_tree(pt_ctx* c, ptscene::Tree& X, int32_t max_id) {
    ptscene::refine(X, (uint32_t)c->opt_leaf_max);
    c->opt_cost[0] = c->opt_cost[1] = 0.0;
    if (c->opt_optimize > 0 && c->opt_tri_test == 0) {   // every node re-inserted where the area cost grows least (pt_tree_opt.h)
        double before = 0.0, after = 0.0;
        if (ptscene::optimize(X, c->opt_optimize, 64, before, after)) { c->opt_cost[0] = before; c->opt_cost[1] = after; }
    }
    ptscene::Output O;
    ptscene::emit(X, PT_MAX_TOP, O, c->opt_tri_test == 1);
    const size_t nb = O.bin.size() * sizeof(float), tb = O.rec.size() * sizeof(float), wb = O.wide.size() * sizeof(float);
    if ((nb + tb + wb) / 16 >= (size_t)PT_SENTINEL) return fail(c, PT_ERR_INVALID, "pt_upload_bvh: scene too large for 32-bit links");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_nodes); c->d_nodes = nullptr;
    c->d_tris = nullptr;
    c->has_bvh = false;
    HIP_TRY(c, hipMalloc((void**)&c->d_nodes, nb + tb + wb));
    c->d_tris = c->d_nodes;  // one item buffer: links index it directly
    HIP_TRY(c, hipMemcpy(c->d_nodes, O.bin.data(), nb, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy((char*)c->d_nodes + nb, O.rec.data(), tb, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy((char*)c->d_nodes + nb + tb, O.wide.data(), wb, hipMemcpyHostToDevice));
    c->records_woop = c->opt_tri_test == 1;
    c->wide_root = O.wide_root_f4;
    c->wide_top_layout = O.n_top_wide;
    c->wide_depth = O.depth_wide;
    c->n_wide = O.wide.size() / 16;
    c->n_top_layout = O.n_top_bin;
    c->n_inner = O.bin.size() / 16;
    c->n_refs = O.n_refs;
    c->n_leaves = O.n_leaves;
    c->max_depth = O.depth_bin;
    c->scene_bytes = nb + tb + wb;
    c->max_tri_id = max_id;
    c->has_bvh = true;
    c->build_ms = -1.f;   // no device build stands behind this tree (optimise_device_tree puts it back when one does)
    c->scene_gen++;
    return PT_OK;
}

// PT_OPT_OPTIMIZE on a tree the DEVICE built (pt_build_bvh, PT_OPT_REBUILD): binary nodes + the records' ids come back to the host, the
// hierarchy is optimised like an uploaded one and installed in its place; the device build's time stays on the context.
// by_id[id] = the nine vertex floats of triangle `id` (the caller's own: records are re-encoded from them bit for bit).
static int optimise_device_tree(pt_ctx* c, const std::vector<const float*>& by_id, int32_t max_id) {
    if (c->opt_optimize <= 0 || !c->has_bvh || c->records_woop) return PT_OK;
    const size_t n_bin = (size_t)c->n_inner, n_rec = (size_t)c->n_refs;
    std::vector<float> bin(16 * n_bin), rec(16 * n_rec);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(bin.data(), c->d_nodes, bin.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(rec.data(), (const char*)c->d_nodes + bin.size() * sizeof(float), rec.size() * sizeof(float), hipMemcpyDeviceToHost));
    ptscene::Tree X;
    std::string why;
    if (!ptscene::from_items(bin.data(), n_bin, rec.data(), n_rec, by_id, X, why)) return fail(c, PT_ERR_DEVICE, "device-built tree: " + why);
    const float device_ms = c->build_ms;
    const int rc = install_tree(c, X, max_id);
    if (rc == PT_OK) c->build_ms = device_ms;
    return rc;
}
}  // namespace ptmi

extern "C" {

int pt_abi_version(void) { return PTMI_ABI_VERSION; }

int pt_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { g_err = std::string("hipGetDeviceCount: ") + hipGetErrorString(e); return PT_ERR_DEVICE; }
    return n;
}

const char* pt_last_error(const pt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int pt_create(int device, pt_ctx** out) {
    if (!out) return fail(nullptr, PT_ERR_INVALID, "pt_create: out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipGetDeviceCount");
    if (device < 0 || device >= n) return fail(nullptr, PT_ERR_INVALID, "pt_create: no such device");
    pt_ctx* c = new pt_ctx();
    c->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess) { delete c; return hip_fail(nullptr, e, "hipSetDevice"); }
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) { delete c; return hip_fail(nullptr, e, "hipStreamCreate"); }
    c->stream = c->own_stream;
    if ((e = hipMalloc(&c->d_counters, 16 * sizeof(unsigned long long))) != hipSuccess) { pt_destroy(c); return hip_fail(nullptr, e, "hipMalloc"); }
    if ((e = hipMemset(c->d_counters, 0, 16 * sizeof(unsigned long long))) != hipSuccess) { pt_destroy(c); return hip_fail(nullptr, e, "hipMemset"); }
    if ((e = hipMalloc(&c->d_queue, PT_SHARDS * PT_SHARD_STRIDE * sizeof(unsigned int))) != hipSuccess) { pt_destroy(c); return hip_fail(nullptr, e, "hipMalloc"); }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) { pt_destroy(c); return hip_fail(nullptr, e, "hipGetDeviceProperties"); }
    c->n_cu = prop.multiProcessorCount;
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess) { pt_destroy(c); return hip_fail(nullptr, e, "hipEventCreate"); }
    *out = c;
    return PT_OK;
}

int pt_destroy(pt_ctx* c) {
    if (!c) return PT_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->d_nodes);  // d_tris aliases it
    (void)hipFree(c->d_spheres);
    (void)hipFree(c->d_tri_matid);
    (void)hipFree(c->d_mat_table);
    (void)hipFree(c->d_light_slot);
    (void)hipFree(c->d_tri_lights);
    (void)hipFree(c->d_counters);
    (void)hipFree(c->d_queue);
    (void)hipFree(c->d_samples);
    (void)hipFree(c->d_wave);
    for (pt_ctx::Side& s : c->side) {
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        (void)hipFree(s.samples);
        (void)hipFree(s.queue);
        if (s.traced) (void)hipEventDestroy(s.traced);
        if (s.folded) (void)hipEventDestroy(s.folded);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    for (hipEvent_t e : c->stage_ev) (void)hipEventDestroy(e);
    for (pt_ctx::AutoPick& a : c->picks)
        for (hipEvent_t e : a.e) if (e) (void)hipEventDestroy(e);
    if (c->lights_ev) (void)hipEventDestroy(c->lights_ev);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return PT_OK;
}

int pt_set_stream(pt_ctx* c, void* s) {
    if (!c) return fail(nullptr, PT_ERR_INVALID, "null ctx");
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return PT_OK;
}

int pt_set_option(pt_ctx* c, int option, int value) {
    if (!c) return fail(nullptr, PT_ERR_INVALID, "null ctx");
    switch (option) {
        case PT_OPT_KERNEL:
            if (value != PT_KERNEL_AUTO && value != PT_KERNEL_MEGA_BVH2 && value != PT_KERNEL_PERSISTENT && value != PT_KERNEL_WAVEFRONT)
                return fail(c, PT_ERR_UNSUPPORTED, "pt_set_option: kernel variant not available in this build");
            c->opt_kernel = value;
            return PT_OK;
        case PT_OPT_COUNTERS: c->opt_counters = value != 0; return PT_OK;
        case PT_OPT_TIMING: c->opt_timing = value != 0; return PT_OK;
        case PT_OPT_TOP_NODES:
            if (value < 0 || value > PT_MAX_TOP) return fail(c, PT_ERR_INVALID, "pt_set_option: top nodes must be 0..1024");
            c->opt_top = value;
            return PT_OK;
        case PT_OPT_OCCUPANCY:
            if (value != 4 && value != 5 && value != 6 && value != 8) return fail(c, PT_ERR_INVALID, "pt_set_option: occupancy must be 4, 5 (runs as 6), 6 or 8 waves per SIMD");
            c->opt_occ = value;
            return PT_OK;
        case PT_OPT_TRI_TEST:
            if (value != 0 && value != 1) return fail(c, PT_ERR_INVALID, "pt_set_option: tri test must be 0 (Moller-Trumbore) or 1 (Woop)");
            c->opt_tri_test = value;   // takes effect at the next pt_upload_bvh
            return PT_OK;
        case PT_OPT_LEAF_MAX:
            if (value < 0 || value > 1024) return fail(c, PT_ERR_INVALID, "pt_set_option: leaf_max must be 0 (keep) .. 1024");
            c->opt_leaf_max = value;   // takes effect at the next pt_upload_bvh
            return PT_OK;
        case PT_OPT_WALK:
            if (value < 0 || value > 4 || value == 3) return fail(c, PT_ERR_INVALID, "pt_set_option: walk must be 0 (while-while), 1 (unified-step), 2 (wide) or 4 (wide, postponed leaf)");
            c->opt_walk = value;
            return PT_OK;
        case PT_OPT_SPHERE_LDS: c->opt_sph_lds = value != 0; return PT_OK;
        case PT_OPT_REBUILD:
            if (value < 0 || value > 2) return fail(c, PT_ERR_INVALID, "pt_set_option: rebuild must be 0 (keep the hierarchy), 1 (re-cluster) or 2 (keep the cheaper tree)");
            c->opt_rebuild = value;
            return PT_OK;
        case PT_OPT_PRESPLIT:
            if (value < 0 || value > 100000) return fail(c, PT_ERR_INVALID, "pt_set_option: presplit must be 0 (off) .. 100000 (per cent of diag/sqrt(n))");
            c->opt_presplit = value;
            return PT_OK;
        case PT_OPT_BUILD_ALGO:
            if (value != 0 && value != 1) return fail(c, PT_ERR_INVALID, "pt_set_option: build algorithm must be 0 (LBVH) or 1 (PLOC)");
            c->opt_build_algo = value;
            return PT_OK;
        case PT_OPT_WAVE_BATCH:
            if (value < 1 || value > 64) return fail(c, PT_ERR_INVALID, "pt_set_option: wave batch must be 1..64");
            c->opt_wave_batch = value;
            return PT_OK;
        case PT_OPT_VOTE_NODE:
        case PT_OPT_VOTE_REC:
            if (value < 1 || value > 64) return fail(c, PT_ERR_INVALID, "pt_set_option: vote weight must be 1..64");
            (option == PT_OPT_VOTE_NODE ? c->opt_vote_node : c->opt_vote_rec) = value;
            return PT_OK;
        case PT_OPT_OVERLAP: c->opt_overlap = value != 0; return PT_OK;
        case PT_OPT_OPTIMIZE:
            if (value < 0 || value > 16) return fail(c, PT_ERR_INVALID, "pt_set_option: optimize passes must be 0 (off) .. 16");
            c->opt_optimize = value;   // takes effect at the next pt_upload_bvh
            return PT_OK;
        case PT_OPT_WAVE_SAMPLES:
            if (value < 1 || value > 64) return fail(c, PT_ERR_INVALID, "pt_set_option: wave samples must be 1 (one sample of a tile per wave) .. 64");
            c->opt_wave_samples = value;
            return PT_OK;
        case PT_OPT_WAVE_BLOCKS:
            if (value < 1 || value > 8) return fail(c, PT_ERR_INVALID, "pt_set_option: wave blocks must be 1..8 per CU");
            c->opt_wave_blocks = value;
            return PT_OK;
        case PT_OPT_LDS_STACK:
            if (value != 0 && value != 16 && value != 24) return fail(c, PT_ERR_INVALID, "pt_set_option: LDS stack must be 0 (all 72 entries in LDS), 16 or 24 entries");
            c->opt_lstk = value;
            return PT_OK;
        case PT_OPT_REFILL:
            if (value < 1 || value > 64) return fail(c, PT_ERR_INVALID, "pt_set_option: refill must be 1..64");
            c->opt_refill = value;
            return PT_OK;
        case PT_OPT_BATCH:
            if (value < 1 || value > 64) return fail(c, PT_ERR_INVALID, "pt_set_option: batch must be 1..64");
            c->opt_batch = value;
            return PT_OK;
        default: return fail(c, PT_ERR_INVALID, "pt_set_option: unknown option");
    }
}

int pt_sync(pt_ctx* c) {
    if (!c) return fail(nullptr, PT_ERR_INVALID, "null ctx");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_malloc(pt_ctx* c, size_t bytes, void** out) {
    if (!c || !out || bytes == 0) return fail(c, PT_ERR_INVALID, "pt_malloc: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMalloc(out, bytes));
    return PT_OK;
}
int pt_free(pt_ctx* c, void* p) {
    if (!c) return fail(nullptr, PT_ERR_INVALID, "null ctx");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipFree(p));
    return PT_OK;
}
int pt_memset(pt_ctx* c, void* p, int v, size_t bytes) {
    if (!c || !p) return fail(c, PT_ERR_INVALID, "pt_memset: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemsetAsync(p, v, bytes, c->stream));
    return PT_OK;
}
int pt_download(pt_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || !dst || !src) return fail(c, PT_ERR_INVALID, "pt_download: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}
int pt_upload(pt_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || !dst || !src) return fail(c, PT_ERR_INVALID, "pt_upload: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

// ---------------------------------------------------------------------------------------
// Scene upload: validate the reference Compact arrays (CudaBVH.cpp:121-270), then re-lay
// them out for the gfx950 kernels:
//   nodes  : same 64-byte record; the top PT_MAX_TOP nodes in breadth-first order (any prefix
//            of them can be mirrored in LDS), the rest depth-first (a parent next to its first
//            inner child); links rewritten from byte offsets to float4 indices
//   tris   : 48-byte records {v0.xyz, id | e1.xyz, last | e2.xyz, 0}: the edge subtraction
//            of cudaUtils.h:177-178 is hoisted to upload (same IEEE result), the index
//            remap of :452-456 and the 16-byte terminator fetch of :410-413 disappear
int pt_upload_bvh(pt_ctx* c, const float* nodes, size_t n_node_vec4, const float* tri_verts, size_t n_tri_vec4,
                  const int32_t* tri_index, size_t n_index) {
    if (!c) return fail(nullptr, PT_ERR_INVALID, "null ctx");
    if (!nodes || !tri_verts || !tri_index) return fail(c, PT_ERR_INVALID, "pt_upload_bvh: null array");
    if (n_node_vec4 < 4 || (n_node_vec4 % 4) != 0) return fail(c, PT_ERR_INVALID, "pt_upload_bvh: node array must hold whole 4-vec4 nodes");
    if (n_index != n_tri_vec4) return fail(c, PT_ERR_INVALID, "pt_upload_bvh: index array must parallel the triangle array");
    if (n_node_vec4 * 16 >= (size_t)PT_SENTINEL || n_tri_vec4 >= (size_t)0x7fffffff)
        return fail(c, PT_ERR_INVALID, "pt_upload_bvh: scene too large for 32-bit links");

    ptscene::Tree T;
    std::string perr;
    if (!ptscene::parse(nodes, n_node_vec4, tri_verts, n_tri_vec4, tri_index, T, perr))
        return fail(c, PT_ERR_INVALID, "pt_upload_bvh: " + perr);
    int32_t max_id = -1;
    for (const ptscene::Ref& r : T.refs) max_id = std::max(max_id, r.id);
    if (c->d_tri_matid && (size_t)max_id >= c->n_tri_matid)
        return fail(c, PT_ERR_INVALID, "pt_upload_bvh: the triangle-material array on this context does not cover this BVH's triangle ids (clear or re-upload it first)");
    const int rebuild = c->opt_tri_test == 0 ? c->opt_rebuild : 0;
    auto recluster = [&]() -> int {
        // PT_OPT_REBUILD: keep the caller's TRIANGLES, not its hierarchy — the distinct triangles of the
        // Compact arrays (a spatial-split builder lists some more than once, each time in full) are
        // clustered again on the device (pt_build.h).  The closest hit does not depend on the tree, so the
        // images are the same bit for bit; whether the new tree is faster depends on the scene.
        std::vector<int32_t> ids;
        std::vector<float> verts;
        {
            std::vector<const ptscene::Ref*> sorted;
            sorted.reserve(T.refs.size());
            for (const ptscene::Ref& r : T.refs) sorted.push_back(&r);
            std::sort(sorted.begin(), sorted.end(), [](const ptscene::Ref* a, const ptscene::Ref* b) { return a->id < b->id; });
            for (const ptscene::Ref* r : sorted) {
                if (!ids.empty() && ids.back() == r->id) continue;
                ids.push_back(r->id);
                verts.insert(verts.end(), r->v, r->v + 9);
            }
        }
        std::vector<int32_t> tri_rows(3 * ids.size());
        for (size_t i = 0; i < tri_rows.size(); i++) tri_rows[i] = (int32_t)i;
        bool too_deep = false;
        int rc = build_bvh_impl(c, verts.data(), verts.size() / 3, tri_rows.data(), ids.size(), c->opt_build_algo, &too_deep, ids.data());
        if (rc != PT_OK && too_deep && c->opt_build_algo == 1)
            rc = build_bvh_impl(c, verts.data(), verts.size() / 3, tri_rows.data(), ids.size(), 0, &too_deep, ids.data());
        return rc;
    };
    auto install = [&](ptscene::Tree& X) -> int { return install_tree(c, X, max_id); };
    auto optimise_reclustered = [&]() -> int {
        if (c->opt_optimize <= 0) return PT_OK;
        std::vector<const float*> by_id((size_t)max_id + 1, nullptr);
        for (const ptscene::Ref& r : T.refs) if (!by_id[(size_t)r.id]) by_id[(size_t)r.id] = r.v;
        return optimise_device_tree(c, by_id, max_id);
    };
    if (rebuild == 1) {
        const int rc = recluster();
        return rc == PT_OK ? optimise_reclustered() : rc;
    }
    {
        const int rc = install(T);
        if (rc != PT_OK) return rc;
    }
    if (rebuild == 2 && c->n_wide > 0) {
        // PT_OPT_REBUILD 2: the caller's hierarchy is up; build the re-clustered one beside it and keep whichever costs a
        // random ray fewer wide-node visits (pt_tree_cost).  A failure on the way leaves the caller's tree in place.
        double cost_a = 0.0, cost_b = 0.0, unused = 0.0;
        if (tree_cost_impl(c, &cost_a, &unused) != PT_OK) return PT_OK;
        const TreeState mine = TreeState::of(c);
        const double mine_opt[2] = {c->opt_cost[0], c->opt_cost[1]};
        c->d_nodes = nullptr;   // the builder frees the context's buffer before it installs its own
        c->d_tris = nullptr;
        const bool built = recluster() == PT_OK && c->d_nodes != nullptr && optimise_reclustered() == PT_OK && c->d_nodes != nullptr;
        if (built && tree_cost_impl(c, &cost_b, &unused) == PT_OK && cost_b < cost_a) {
            (void)hipFree(mine.d_nodes);
        } else {
            if (c->d_nodes != mine.d_nodes) (void)hipFree(c->d_nodes);
            mine.restore(c);
            c->opt_cost[0] = mine_opt[0]; c->opt_cost[1] = mine_opt[1];
            c->err.clear();
        }
        c->scene_gen++;
    }
    return PT_OK;
}

// ---- pt_build_bvh: the BVH built on the device (pt_build.h) ---------------------------------
int pt_build_bvh(pt_ctx* c, const float* verts, size_t n_verts, const int32_t* tris, size_t n_tris) {
    if (!c) return fail(nullptr, PT_ERR_INVALID, "null ctx");
    bool too_deep = false;
    int rc = build_bvh_impl(c, verts, n_verts, tris, n_tris, c->opt_build_algo, &too_deep);
    // PLOC on degenerate input (hundreds of identical boxes: one merge per round, a chain): the Karras
    // hierarchy separates equal keys by position and stays balanced
    if (rc != PT_OK && too_deep && c->opt_build_algo == 1) rc = build_bvh_impl(c, verts, n_verts, tris, n_tris, 0, &too_deep);
    if (rc == PT_OK && c->opt_optimize > 0 && c->opt_presplit == 0) {   // PT_OPT_OPTIMIZE: the built hierarchy goes through the host optimiser
        std::vector<float> flat(9 * n_tris);
        std::vector<const float*> by_id(n_tris);
        for (size_t t = 0; t < n_tris; t++) {
            for (int k = 0; k < 3; k++) std::memcpy(&flat[9 * t + 3 * (size_t)k], verts + 3 * (size_t)tris[3 * t + (size_t)k], 3 * sizeof(float));
            by_id[t] = &flat[9 * t];
        }
        rc = optimise_device_tree(c, by_id, (int32_t)n_tris - 1);
    }
    return rc;
}

int pt_last_build_ms(pt_ctx* c, float* ms) {
    if (!c || !ms) return fail(c, PT_ERR_INVALID, "pt_last_build_ms: null argument");
    if (c->build_ms < 0.f) return fail(c, PT_ERR_INVALID, "pt_last_build_ms: the tree on this context was not built on the device");
    *ms = c->build_ms;
    return PT_OK;
}

int pt_upload_tri_materials(pt_ctx* c, const pt_material* table, size_t n_materials, const int32_t* tri_material, size_t n_tris) {
    if (!c) return fail(nullptr, PT_ERR_INVALID, "null ctx");
    HIP_TRY(c, hipSetDevice(c->device));
    if (n_materials == 0) {  // back to the one global material of pt_params
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_tri_matid); c->d_tri_matid = nullptr;
        (void)hipFree(c->d_mat_table); c->d_mat_table = nullptr;
        (void)hipFree(c->d_light_slot); c->d_light_slot = nullptr;
        (void)hipFree(c->d_tri_lights); c->d_tri_lights = nullptr;
        c->emissive_ids.clear();
        c->n_tri_matid = 0;
        c->mat_gen++;
        return PT_OK;
    }
    if (!table || !tri_material) return fail(c, PT_ERR_INVALID, "pt_upload_tri_materials: null array");
    if (n_materials > (1u << 24) || n_tris >= (size_t)0x7fffffff) return fail(c, PT_ERR_INVALID, "pt_upload_tri_materials: table too large");
    if (c->has_bvh && (size_t)c->max_tri_id >= n_tris && c->max_tri_id >= 0)
        return fail(c, PT_ERR_INVALID, "pt_upload_tri_materials: n_tris does not cover the triangle ids of the uploaded BVH");
    for (size_t i = 0; i < n_materials; i++)
        if (table[i].mat < PT_MAT_DIFF || table[i].mat > PT_MAT_REFR) return fail(c, PT_ERR_INVALID, "pt_upload_tri_materials: bad material type");
    for (size_t i = 0; i < n_tris; i++)
        if (tri_material[i] < 0 || (size_t)tri_material[i] >= n_materials) return fail(c, PT_ERR_INVALID, "pt_upload_tri_materials: material index out of range");
    static_assert(sizeof(pt_material) == 32, "pt_material is two float4 rows");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_tri_matid); c->d_tri_matid = nullptr;
    (void)hipFree(c->d_mat_table); c->d_mat_table = nullptr;
    c->n_tri_matid = 0;
    HIP_TRY(c, hipMalloc((void**)&c->d_tri_matid, n_tris * sizeof(int32_t)));
    HIP_TRY(c, hipMalloc((void**)&c->d_mat_table, n_materials * sizeof(pt_material)));
    HIP_TRY(c, hipMemcpy(c->d_tri_matid, tri_material, n_tris * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_mat_table, table, n_materials * sizeof(pt_material), hipMemcpyHostToDevice));
    c->n_tri_matid = n_tris;
    // PT_FLAG_NEE: the triangles that emit, in id order, and where each one's light record goes
    (void)hipFree(c->d_light_slot); c->d_light_slot = nullptr;
    (void)hipFree(c->d_tri_lights); c->d_tri_lights = nullptr;
    c->emissive_ids.clear();
    c->mat_gen++;
    std::vector<int32_t> slot(n_tris, -1);
    for (size_t i = 0; i < n_tris; i++) {
        const pt_material& m = table[tri_material[i]];
        if (!(m.emi[0] == 0.0f && m.emi[1] == 0.0f && m.emi[2] == 0.0f)) {
            slot[i] = (int32_t)c->emissive_ids.size();
            c->emissive_ids.push_back((int32_t)i);
        }
    }
    if (!c->emissive_ids.empty()) {
        HIP_TRY(c, hipMalloc((void**)&c->d_light_slot, n_tris * sizeof(int32_t)));
        HIP_TRY(c, hipMalloc((void**)&c->d_tri_lights, c->emissive_ids.size() * 3 * sizeof(float4)));
        HIP_TRY(c, hipMemcpy(c->d_light_slot, slot.data(), n_tris * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    return PT_OK;
}

// PT_FLAG_NEE: copies (v0, e1, e2) of every emissive triangle from its record into its light slot (a triangle a spatial
// split lists more than once is listed in full each time: the copies write the same values)
__global__ void __launch_bounds__(256) k_collect_tri_lights(const float4* __restrict__ rec, uint32_t n_rec, const int32_t* __restrict__ slot,
                                                            uint32_t n_ids, const int* __restrict__ tri_matid, const float4* __restrict__ mat_table,
                                                            float4* __restrict__ lights) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_rec) return;
    const float4 q0 = rec[4 * (size_t)i];
    const int id = __float_as_int(q0.w);
    if (id < 0 || (uint32_t)id >= n_ids) return;
    const int s = slot[id];
    if (s < 0) return;
    const float4 q1 = rec[4 * (size_t)i + 1], q2 = rec[4 * (size_t)i + 2];
    const int row = tri_matid[id];
    const float4 m0 = mat_table[2 * row], m1 = mat_table[2 * row + 1];
    lights[3 * (size_t)s + 0] = make_float4(q0.x, q0.y, q0.z, m0.w);
    lights[3 * (size_t)s + 1] = make_float4(q1.x, q1.y, q1.z, m1.x);
    lights[3 * (size_t)s + 2] = make_float4(q2.x, q2.y, q2.z, m1.y);
}

int pt_upload_spheres(pt_ctx* c, const pt_sphere* spheres, size_t n) {
    if (!c) return fail(nullptr, PT_ERR_INVALID, "null ctx");
    if (n > 0 && !spheres) return fail(c, PT_ERR_INVALID, "pt_upload_spheres: null array");
    if (n > 4096) return fail(c, PT_ERR_INVALID, "pt_upload_spheres: too many spheres");
    for (size_t i = 0; i < n; i++)
        if (spheres[i].mat < PT_MAT_DIFF || spheres[i].mat > PT_MAT_REFR) return fail(c, PT_ERR_INVALID, "pt_upload_spheres: bad material");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_spheres); c->d_spheres = nullptr;
    c->n_spheres = 0;
    if (n) {
        HIP_TRY(c, hipMalloc((void**)&c->d_spheres, n * sizeof(pt_sphere)));
        HIP_TRY(c, hipMemcpy(c->d_spheres, spheres, n * sizeof(pt_sphere), hipMemcpyHostToDevice));
        c->n_spheres = (int)n;
        std::memcpy(c->h_spheres, spheres, std::min<size_t>(n, PT_KSPHERES) * sizeof(pt_sphere));
    }
    return PT_OK;
}

int pt_scene_info(pt_ctx* c, uint64_t* n_inner, uint64_t* n_refs, uint64_t* n_leaves, uint32_t* max_depth, uint64_t* bytes) {
    if (!c) return fail(nullptr, PT_ERR_INVALID, "null ctx");
    if (!c->has_bvh) return fail(c, PT_ERR_NO_SCENE, "pt_scene_info: no BVH uploaded");
    if (n_inner) *n_inner = c->n_inner;
    if (n_refs) *n_refs = c->n_refs;
    if (n_leaves) *n_leaves = c->n_leaves;
    if (max_depth) *max_depth = c->max_depth;
    if (bytes) *bytes = c->scene_bytes;
    return PT_OK;
}

// ---------------------------------------------------------------------------------------
int pt_render(pt_ctx* c, float* accum_dev, uint32_t* rgba_dev, const pt_camera* cam, const pt_params* p, uint32_t spp) {
    if (!c) return fail(nullptr, PT_ERR_INVALID, "null ctx");
    if (!accum_dev || !cam || !p) return fail(c, PT_ERR_INVALID, "pt_render: null argument");
    if (p->width < 2 || p->height < 2) return fail(c, PT_ERR_INVALID, "pt_render: image must be at least 2x2 (the camera divides by w-1, h-1)");
    if (spp == 0) return fail(c, PT_ERR_INVALID, "pt_render: spp must be >= 1");
    if (p->sample_index == 0) return fail(c, PT_ERR_INVALID, "pt_render: sample_index (constantPdf) starts at 1");
    if (p->tri_mat < PT_MAT_DIFF || p->tri_mat > PT_MAT_REFR) return fail(c, PT_ERR_INVALID, "pt_render: bad triangle material");
    if ((p->flags & PT_FLAG_WRITE_RGBA) && !rgba_dev) return fail(c, PT_ERR_INVALID, "pt_render: PT_FLAG_WRITE_RGBA needs rgba_dev");
    if ((p->flags & PT_FLAG_NEE) && !(p->flags & PT_FLAG_COSINE_DIFF)) return fail(c, PT_ERR_INVALID, "pt_render: PT_FLAG_NEE needs PT_FLAG_COSINE_DIFF (the reference's DIFF lobe has no density to weigh a light sample against)");
    if (!c->has_bvh && c->n_spheres == 0) return fail(c, PT_ERR_NO_SCENE, "pt_render: no scene uploaded");
    if (p->part_count > 1) {
        if (p->part_index < 0 || p->part_index >= p->part_count) return fail(c, PT_ERR_INVALID, "pt_render: part_index out of range");
        if (p->part_rows <= 0 || (p->part_rows % PT_TILE) != 0) return fail(c, PT_ERR_INVALID, "pt_render: part_rows must be a positive multiple of 8");
    }
    HIP_TRY(c, hipSetDevice(c->device));

    KParams P;
    std::memset(&P, 0, sizeof P);
    P.sc.nodes = c->d_nodes;
    P.sc.tris = c->d_tris;
    P.sc.spheres = c->d_spheres;
    P.sc.n_spheres = c->n_spheres;
    std::memcpy(P.ksph, c->h_spheres, sizeof P.ksph);
    P.sc.has_bvh = c->has_bvh ? 1 : 0;
    P.accum = accum_dev;
    P.rgba = rgba_dev;
    P.counters = c->d_counters;
    P.cam = *cam;
    P.W = p->width; P.H = p->height;
    P.depth = p->depth;
    P.cull = p->cull_backfaces;
    P.frame = p->frame; P.sample_index = p->sample_index; P.spp = spp;
    P.tri_mat = p->tri_mat;
    for (int i = 0; i < 3; i++) { P.tri_col[i] = p->tri_col[i]; P.tri_emi[i] = p->tri_emi[i]; P.bk[i] = p->bk_color[i]; }
    P.air_ior = p->air_ior; P.glass_ior = p->glass_ior; P.phong = p->phong_expo;
    P.tri_matid = c->d_tri_matid;
    P.mat_table = c->d_mat_table;
    P.flags = p->flags;
    if ((p->flags & PT_FLAG_NEE) && c->has_bvh && !c->emissive_ids.empty()) {
        if (c->records_woop) return fail(c, PT_ERR_UNSUPPORTED, "pt_render: PT_FLAG_NEE over emissive triangles needs the exact (Moller-Trumbore) records");
        const uint64_t key = (c->scene_gen << 32) ^ c->mat_gen;
        if (c->lights_key != key) {   // the scene or the materials changed: copy the lights' vertices out of the records again
            const uint32_t n_rec = (uint32_t)c->n_refs;
            HIP_TRY(c, hipMemsetAsync(c->d_tri_lights, 0, c->emissive_ids.size() * 3 * sizeof(float4), c->stream));
            hipLaunchKernelGGL(k_collect_tri_lights, dim3((n_rec + 255) / 256), dim3(256), 0, c->stream, c->d_nodes + 4 * (size_t)c->n_inner, n_rec,
                               c->d_light_slot, (uint32_t)c->n_tri_matid, c->d_tri_matid, c->d_mat_table, c->d_tri_lights);
            HIP_TRY(c, hipGetLastError());
            c->lights_key = key;
            // a path kernel on a side stream (PT_OPT_OVERLAP) must not read the list before it is written
            if (!c->lights_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->lights_ev, hipEventDisableTiming));
            HIP_TRY(c, hipEventRecord(c->lights_ev, c->stream));
            c->lights_gen++;
        }
        P.tri_lights = c->d_tri_lights;
        P.n_tri_lights = (int)c->emissive_ids.size();
    }
    P.tiles_x = (p->width + PT_TILE - 1) / PT_TILE;
    P.tile_rows = (p->height + PT_TILE - 1) / PT_TILE;
    if (p->part_count > 1) {
        P.part_index = p->part_index; P.part_count = p->part_count; P.stripe_tr = p->part_rows / PT_TILE;
        const int n_stripes = (P.tile_rows + P.stripe_tr - 1) / P.stripe_tr;
        const int owned = (n_stripes - p->part_index + p->part_count - 1) / p->part_count;  // stripes index, index+count, ...
        P.n_tiles = owned * P.stripe_tr * P.tiles_x;
    } else {
        P.part_index = 0; P.part_count = 1; P.stripe_tr = 1;
        P.n_tiles = P.tile_rows * P.tiles_x;
    }
    if (P.n_tiles <= 0) return PT_OK;
    const int waves_per_block = PT_BLOCK / 64;
    if ((uint64_t)P.n_tiles * 64u * (uint64_t)spp >= (1ull << 31)) return fail(c, PT_ERR_INVALID, "pt_render: width*height*spp too large for one call (split the samples over several calls)");

    // the wide walk pushes up to three entries per level
    int walk = c->opt_walk;
    const bool wide_ok = c->has_bvh && 3 * c->wide_depth + 2 <= (uint32_t)PT_STACK_CAP;
    if (c->has_bvh && c->records_woop) {
        if (!wide_ok) return fail(c, PT_ERR_UNSUPPORTED, "pt_render: Woop records need the wide walk and this tree is too deep for it");
        walk = 3;  // Woop records are only understood by the wide walk
    } else if ((walk == 2 || walk == 4) && !wide_ok) {
        walk = 1;
    }
    // which frame kernel: the stage-split pipeline needs a BVH, at least one bounce and the wide walk over exact
    // records; every other request for it runs the persistent kernel (same images)
    const bool wave_ok = c->has_bvh && P.depth > 0 && walk == 2;
    int kernel = c->opt_kernel;
    int probe = -1;   // PT_KERNEL_AUTO: 0 / 1 = this call is the timed trial of the persistent kernel / the pipeline
    pt_ctx::AutoPick* pick = nullptr;
    if (kernel == PT_KERNEL_AUTO) {
        kernel = PT_KERNEL_PERSISTENT;
        if (wave_ok && !c->opt_counters && !(p->flags & PT_FLAG_NEE)) {
            // the key holds the SHAPE of the partition, not which part this call renders: the parts of a tile split cost alike
            uint64_t key = 0xcbf29ce484222325ull;
            const uint64_t parts[] = {(uint64_t)p->width, (uint64_t)p->height, (uint64_t)spp, (uint64_t)p->depth,
                                      (uint64_t)(p->part_count > 1 ? p->part_count : 1), (uint64_t)(p->part_count > 1 ? p->part_rows : 0), c->scene_gen,
                                      (uint64_t)c->n_spheres, (uint64_t)p->tri_mat, (uint64_t)(p->flags & ~(uint32_t)PT_FLAG_WRITE_RGBA)};
            for (uint64_t v : parts) { key ^= v; key *= 0x100000001b3ull; }
            if (key == 0) key = 1;
            int slot = -1, lru = 0;
            for (int i = 0; i < pt_ctx::N_PICKS; i++) {
                if (c->picks[i].key == key) { slot = i; break; }
                if (c->picks[i].used < c->picks[lru].used) lru = i;
            }
            if (slot < 0) {
                slot = lru;
                pt_ctx::AutoPick& n = c->picks[slot];
                if (n.phase > 0 && n.phase < pt_ctx::AutoPick::DECIDED)   // trials of the configuration it held may still be in flight: its events must be idle
                    for (hipEvent_t e : n.e) if (e) (void)hipEventSynchronize(e);
                n.key = key; n.phase = 0; n.choice = PT_KERNEL_PERSISTENT; n.ms[0] = n.ms[1] = 0.f;
            }
            pt_ctx::AutoPick& a = c->picks[slot];
            a.used = ++c->pick_tick;
            c->pick_last = slot;
            pick = &a;
            (void)auto_decide(c, a, false);   // all trials queued: decide once their events have completed, never wait for them
            if (a.phase < pt_ctx::AutoPick::TRIALS) {
                for (hipEvent_t& e : a.e)
                    if (!e) HIP_TRY(c, hipEventCreate(&e));
                probe = a.phase;
                kernel = (probe & 1) ? PT_KERNEL_WAVEFRONT : PT_KERNEL_PERSISTENT;
                a.phase++;
            } else {
                kernel = a.phase == pt_ctx::AutoPick::DECIDED ? a.choice : PT_KERNEL_PERSISTENT;
            }
        }
    }
    if (p->flags & PT_FLAG_NEE) {   // shadow rays: the stage-split pipeline has a stage for them, the megakernel a loop; the
                                    // persistent kernel does not (its lanes have no room for a second ray)
        const bool nee_wave = wave_ok && (uint64_t)p->width * (uint64_t)p->height <= (1ull << 24);   // pixel | nee_mask << 24
        if (kernel != PT_KERNEL_MEGA_BVH2) kernel = nee_wave ? PT_KERNEL_WAVEFRONT : PT_KERNEL_MEGA_BVH2;
        if (kernel == PT_KERNEL_WAVEFRONT && !nee_wave) kernel = PT_KERNEL_MEGA_BVH2;
    }
    if (kernel == PT_KERNEL_WAVEFRONT && !wave_ok) kernel = PT_KERNEL_PERSISTENT;
    if (kernel == PT_KERNEL_MEGA_BVH2 && walk == 3) kernel = PT_KERNEL_PERSISTENT;   // Woop records: persistent kernel only
    const bool persistent = kernel == PT_KERNEL_PERSISTENT;
    const bool wavefront = kernel == PT_KERNEL_WAVEFRONT;

    // spp > 1: trace the samples as independent work items, fold them afterwards (k_fold_samples); the
    // stage-split pipeline always works that way
    P.samples = nullptr;
    // overlap with the previous call: the path kernel goes to a side stream with its own sample buffer and queue
    // counters, so it can start while the previous call's last paths drain (the tail of a call is as long as its
    // longest path: ~15 % of a one-sample 1080p call); instrumented, timed and trial calls run in line
    pt_ctx::Side* sd = nullptr;
    // ... unless the caller's stream is idle: a host that syncs before every launch (BasicScene.cpp:395) leaves nothing to overlap
    // with, and in line the call is two launches shorter (no cross-stream events; one sample folds inside the path kernel)
    bool caller_idle = false;
    if (c->opt_overlap && !wavefront) {
        caller_idle = hipStreamQuery(c->stream) == hipSuccess;
        if (!caller_idle) (void)hipGetLastError();   // hipErrorNotReady is not an error of this call
    }
    if (c->opt_overlap && !wavefront && !c->opt_counters && !c->opt_timing && probe < 0 && !caller_idle) {
        sd = &c->side[c->side_next];
        c->side_next ^= 1;
        if (!sd->stream) {
            // a priority of its own: ROCm maps the streams of one priority onto a few hardware queues round-robin (four
            // by default, GPU_MAX_HW_QUEUES), and a side stream that shares a hardware queue with the caller's stream or
            // with the other side stream runs in order behind it — no overlap and a 5-9 % LOSS (measured with a second
            // context alive).  The high-priority class has queues of its own.
            int prio_lo = 0, prio_hi = 0;
            HIP_TRY(c, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
            HIP_TRY(c, hipStreamCreateWithPriority(&sd->stream, hipStreamNonBlocking, prio_hi));
            HIP_TRY(c, hipEventCreateWithFlags(&sd->traced, hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&sd->folded, hipEventDisableTiming));
            HIP_TRY(c, hipMalloc((void**)&sd->queue, PT_SHARDS * PT_SHARD_STRIDE * sizeof(unsigned int)));
        }
    }
    if (spp > 1 || wavefront || sd) {
        const size_t need = (size_t)spp * (size_t)p->width * (size_t)p->height * 3 * sizeof(float);
        float*& buf = sd ? sd->samples : c->d_samples;
        size_t& have = sd ? sd->samples_bytes : c->samples_bytes;
        if (need > have) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));   // every path kernel has a fold behind it on this stream
            (void)hipFree(buf);
            buf = nullptr;
            have = 0;
            HIP_TRY(c, hipMalloc((void**)&buf, need));
            have = need;
        }
        P.samples = buf;
    }
    P.smp_ss = (unsigned long long)p->width * (unsigned long long)p->height;
    P.smp_ps = 1u;
    P.sgroup_log2 = 0u;
    if (P.samples && (wavefront || persistent)) {   // the samples of a pixel side by side in the slot order and in the sample buffer
        P.sgroup_log2 = wave_sample_group_log2(spp, c->opt_wave_samples);
        if (P.sgroup_log2) {
            P.smp_ss = 1ull;
            P.smp_ps = spp;
        }
    }
    const int work_tiles = P.n_tiles * (P.samples ? (int)spp : 1);
    // the stream the path kernel runs on: it may start once the caller's earlier work has been SUBMITTED, needs the
    // scene only, and must not overwrite its sample buffer before the fold that last read it is done
    hipStream_t trace_stream = c->stream;
    if (sd) {
        trace_stream = sd->stream;
        if (sd->fold_pending) HIP_TRY(c, hipStreamWaitEvent(sd->stream, sd->folded, 0));
        if (P.tri_lights && sd->lights_seen != c->lights_gen) {   // the light list was (re)written on the caller's stream
            HIP_TRY(c, hipStreamWaitEvent(sd->stream, c->lights_ev, 0));
            sd->lights_seen = c->lights_gen;
        }
    }

    if (c->opt_counters) HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, 16 * sizeof(unsigned long long), c->stream));
    if (probe >= 0 && (probe & 1)) {   // a trial of the pipeline: its path records are allocated BEFORE the timed span
        const int rc = wave_reserve(c, P, work_tiles);
        if (rc != PT_OK) return rc;
    }
    if (probe >= 0) HIP_TRY(c, hipEventRecord(pick->e[2 * probe], c->stream));
    if (c->opt_timing) {
        HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
        c->stage_used = 0;
        if (stage_mark(c, PT_STAGE_NONE) != PT_OK) return PT_ERR_DEVICE;
    }
    const int lstk = c->opt_lstk ? c->opt_lstk : PT_STACK_CAP;   // any depth <= 64 works with every LDS window: deeper entries overflow
    P.sc.stack_n = lstk;
    P.sc.wide_root = (int)c->wide_root;
    P.sc.top_base = walk >= 2 ? (int)c->wide_root : 0;
    P.sc.n_top = c->has_bvh ? (int)std::min<uint32_t>((uint32_t)c->opt_top, walk >= 2 ? c->wide_top_layout : c->n_top_layout) : 0;
    if (walk >= 1) P.sc.n_top = 0;  // only the while-while walk reads the LDS mirror
    size_t lds = lds_bytes(P.sc.n_top, lstk, PT_BLOCK);
    while (lds > 160 * 1024 && P.sc.n_top > 0) {  // deep tree: give the LDS to the stack first
        P.sc.n_top /= 2;
        lds = lds_bytes(P.sc.n_top, lstk, PT_BLOCK);
    }
    P.sph_tab = -1;
    if (persistent && c->opt_sph_lds) {   // sphere table behind the stacks (and the LDS mirror)
        P.sph_tab = (int)(lds / 4);
        lds += 15 * PT_KSPHERES * 4;
    }
    if (persistent) {
        P.queue = c->d_queue;
        P.batch = c->opt_batch;
        // refill <= batch, or a wave can spin: the walk returns at once because `batch` lanes wait,
        // none of them has anything to shade and the idle ones are too few to trigger a refill
        P.refill = c->opt_refill < c->opt_batch ? c->opt_refill : c->opt_batch;
        P.vote_node = c->opt_vote_node;
        P.vote_rec = c->opt_vote_rec;
        if (sd) P.queue = sd->queue;
        HIP_TRY(c, hipMemsetAsync(P.queue, 0, PT_SHARDS * PT_SHARD_STRIDE * sizeof(unsigned int), trace_stream));
    }
    // queue granularity: 64-slot chunks when the launch has plenty of them per resident wave, smaller
    // ones for small launches (an eighth of a 1080p frame per GPU is ~4 000 tiles for ~5 000 waves)
    {
        const long slots = (long)work_tiles * 64, waves = (long)c->n_cu * 20;
        P.chunk = slots / 64 >= 4 * waves ? 64 : (slots / 32 >= 4 * waves ? 32 : 16);
    }
    LaunchCfg L;
    L.count = c->opt_counters != 0;
    L.occ = c->opt_occ;
    L.lstk = lstk;
    L.walk = walk;
    L.lds = lds;
    L.blocks = (work_tiles + waves_per_block - 1) / waves_per_block;
    L.work_blocks = (int)(((long)work_tiles * 64 + P.chunk * (PT_BLOCK / 64) - 1) / (P.chunk * (PT_BLOCK / 64)));
    L.n_cu = c->n_cu;
    if (wavefront) {
        const int rc = render_wavefront(c, P, L, work_tiles);
        if (rc != PT_OK) return rc;
    } else if (persistent) {
        HIP_TRY(c, launch_persist(L, P, trace_stream));
        if (stage_mark(c, PT_STAGE_FRAME) != PT_OK) return PT_ERR_DEVICE;
    } else {
        HIP_TRY(c, launch_mega(L, P, trace_stream));
        if (stage_mark(c, PT_STAGE_FRAME) != PT_OK) return PT_ERR_DEVICE;
    }
    if (sd) {   // the fold (accumulator, display words) waits for the path kernel on the caller's stream
        HIP_TRY(c, hipEventRecord(sd->traced, sd->stream));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, sd->traced, 0));
    }
    if (P.samples) {
        HIP_TRY(c, launch_fold(P, c->stream));
        if (stage_mark(c, PT_STAGE_FOLD) != PT_OK) return PT_ERR_DEVICE;
    }
    if (sd) {
        HIP_TRY(c, hipEventRecord(sd->folded, c->stream));
        sd->fold_pending = true;
    }
    if (probe >= 0) HIP_TRY(c, hipEventRecord(pick->e[2 * probe + 1], c->stream));
    if (c->opt_timing) { HIP_TRY(c, hipEventRecord(c->ev1, c->stream)); c->timed = true; }
    return PT_OK;
}

// pt_tree_cost: one lane per wide node; out[0] = root area, out[1] = sum of inner-child areas, out[2] = sum of leaf area x records
// (per-block partial sums, added up on the host in block order: the figure is reproducible, so a choice made on it is too)
__global__ void __launch_bounds__(256) k_tree_cost(const float4* __restrict__ items, uint64_t wide_root, uint32_t n_wide, double* out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    double inner = 0.0, leaf = 0.0, root = 0.0;
    if (i < n_wide) {
        const float4* nd = items + wide_root + 4 * (size_t)i;
        const float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
        const float sc[3] = {q0.w, q3.z, q3.w};
        const uint32_t ql[3] = {__float_as_uint(q1.x), __float_as_uint(q1.y), __float_as_uint(q1.z)};
        const uint32_t qh[3] = {__float_as_uint(q1.w), __float_as_uint(q2.x), __float_as_uint(q2.y)};
        const int link[4] = {__float_as_int(q2.z), __float_as_int(q2.w), __float_as_int(q3.x), __float_as_int(q3.y)};
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (int k = 0; k < 4; k++) {
            float d[3];
            bool used = true;
            for (int a = 0; a < 3; a++) {
                const int l = (int)((ql[a] >> (8 * k)) & 0xffu), h = (int)((qh[a] >> (8 * k)) & 0xffu);
                if (l > h) used = false;   // an unused slot holds an inverted box
                d[a] = (float)(h - l) * sc[a];
                if (l <= h) { lo[a] = fminf(lo[a], (float)l * sc[a]); hi[a] = fmaxf(hi[a], (float)h * sc[a]); }
            }
            if (!used) continue;
            const double area = 2.0 * ((double)d[0] * d[1] + (double)d[1] * d[2] + (double)d[2] * d[0]);
            if (link[k] >= 0) {
                inner += area;
            } else {
                int n = 0;
                for (size_t r = (size_t)(~link[k] & ~3);; r += 4) {
                    n++;
                    if (__float_as_int(items[r + 1].w) != 0 || n >= 64) break;   // the record's `last` flag
                }
                leaf += area * (double)n;
            }
        }
        if (i == 0) {
            const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            root = 2.0 * (dx * dy + dy * dz + dz * dx);
        }
    }
    // block reduction through LDS, one atomic per block and term
    __shared__ double s_in[256], s_lf[256];
    s_in[threadIdx.x] = inner; s_lf[threadIdx.x] = leaf;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { s_in[threadIdx.x] += s_in[threadIdx.x + off]; s_lf[threadIdx.x] += s_lf[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[1 + 2 * (size_t)blockIdx.x] = s_in[0]; out[2 + 2 * (size_t)blockIdx.x] = s_lf[0]; }
    if (i == 0) out[0] = root;
}

namespace ptmi {
static int tree_cost_impl(pt_ctx* c, double* node_visits, double* tri_tests) {
    const unsigned n_blocks = (unsigned)((c->n_wide + 255) / 256);
    const size_t n_out = 1 + 2 * (size_t)n_blocks;
    double* d_out = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d_out, n_out * sizeof(double)));
    hipLaunchKernelGGL(k_tree_cost, dim3(n_blocks), dim3(256), 0, c->stream, c->d_nodes, c->wide_root, (uint32_t)c->n_wide, d_out);
    hipError_t e = hipGetLastError();
    std::vector<double> h(n_out, 0.0);
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_out, n_out * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) return hip_fail(c, e, "pt_tree_cost");
    if (!(h[0] > 0.0)) return fail(c, PT_ERR_INVALID, "pt_tree_cost: degenerate root box");
    double inner = 0.0, leaf = 0.0;
    for (unsigned b = 0; b < n_blocks; b++) { inner += h[1 + 2 * (size_t)b]; leaf += h[2 + 2 * (size_t)b]; }
    *node_visits = (h[0] + inner) / h[0];
    *tri_tests = leaf / h[0];
    return PT_OK;
}
}  // namespace ptmi

int pt_tree_cost(pt_ctx* c, double* node_visits, double* tri_tests) {
    if (!c || !node_visits || !tri_tests) return fail(c, PT_ERR_INVALID, "pt_tree_cost: null argument");
    if (!c->has_bvh || c->wide_root == 0 || c->n_wide == 0) return fail(c, PT_ERR_NO_SCENE, "pt_tree_cost: no 4-wide tree on this context");
    if (c->records_woop) return fail(c, PT_ERR_UNSUPPORTED, "pt_tree_cost: reads the Moller-Trumbore records' leaf terminators");
    HIP_TRY(c, hipSetDevice(c->device));
    return tree_cost_impl(c, node_visits, tri_tests);
}

int pt_auto_choice(pt_ctx* c, int* kernel, float* ms_persistent, float* ms_wavefront) {
    if (!c || !kernel) return fail(c, PT_ERR_INVALID, "pt_auto_choice: null argument");
    *kernel = PT_KERNEL_AUTO;
    if (ms_persistent) *ms_persistent = 0.f;
    if (ms_wavefront) *ms_wavefront = 0.f;
    if (c->pick_last < 0) return PT_OK;
    pt_ctx::AutoPick& a = c->picks[c->pick_last];
    if (auto_decide(c, a, true)) *kernel = a.choice;   // the caller asks: waiting for the trials is fine here
    if (ms_persistent) *ms_persistent = a.ms[0];
    if (ms_wavefront) *ms_wavefront = a.ms[1];
    return PT_OK;
}

int pt_trace_rays(pt_ctx* c, const float* rays_dev, size_t n, int cull, float* t_dev, int32_t* tri_dev, float* normal_dev) {
    if (!c) return fail(nullptr, PT_ERR_INVALID, "null ctx");
    if (!c->has_bvh) return fail(c, PT_ERR_NO_SCENE, "pt_trace_rays: no BVH uploaded");
    if (c->records_woop) return fail(c, PT_ERR_UNSUPPORTED, "pt_trace_rays: the ray-batch kernel reads Moller-Trumbore records (upload with PT_OPT_TRI_TEST 0)");
    if (n == 0) return PT_OK;
    if (!rays_dev || !t_dev || !tri_dev) return fail(c, PT_ERR_INVALID, "pt_trace_rays: null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    KScene sc;
    std::memset(&sc, 0, sizeof sc);
    sc.nodes = c->d_nodes; sc.tris = c->d_tris; sc.spheres = nullptr; sc.n_spheres = 0; sc.has_bvh = 1;
    if (c->opt_timing) HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    const float4* r4 = (const float4*)rays_dev;
    sc.stack_n = PT_STACK_CAP;
    sc.top_base = 0;
    sc.wide_root = (int)c->wide_root;
    sc.n_top = (int)std::min<uint32_t>((uint32_t)c->opt_top, c->n_top_layout);
    size_t lds = lds_bytes(sc.n_top, sc.stack_n, PT_BLOCK_RAYS);
    while (lds > 160 * 1024 && sc.n_top > 0) { sc.n_top /= 2; lds = lds_bytes(sc.n_top, sc.stack_n, PT_BLOCK_RAYS); }
    HIP_TRY(c, launch_rays(sc, lds, r4, n, cull, t_dev, tri_dev, normal_dev, c->stream));
    if (c->opt_timing) { HIP_TRY(c, hipEventRecord(c->ev1, c->stream)); c->timed = true; }
    return PT_OK;
}

int pt_get_counters(pt_ctx* c, pt_counters* out) {
    if (!c || !out) return fail(c, PT_ERR_INVALID, "pt_get_counters: null argument");
    unsigned long long h[8];
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(h, c->d_counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    out->rays = h[0]; out->inner = h[1]; out->tris = h[2]; out->leaves = h[3]; out->hits = h[4]; out->paths = h[5];
    return PT_OK;
}

int pt_get_wave_stats(pt_ctx* c, uint64_t* out, int n) {
    if (!c || !out || n < 0) return fail(c, PT_ERR_INVALID, "pt_get_wave_stats: bad argument");
    unsigned long long h[16];
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(h, c->d_counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < n && i < PT_WAVE_STATS; i++) out[i] = h[6 + i];   // counters 6..15
    return PT_OK;
}

int pt_get_stage_ms(pt_ctx* c, float* out, int n) {
    if (!c || !out || n < 0) return fail(c, PT_ERR_INVALID, "pt_get_stage_ms: bad argument");
    if (!c->timed || c->stage_used < 2) return fail(c, PT_ERR_INVALID, "pt_get_stage_ms: no timed pt_render (set PT_OPT_TIMING=1 first)");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventSynchronize(c->stage_ev[c->stage_used - 1]));
    for (int i = 0; i < n; i++) out[i] = 0.f;
    for (size_t i = 1; i < c->stage_used; i++) {
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->stage_ev[i - 1], c->stage_ev[i]));
        const int k = c->stage_kind[i];
        if (k >= 0 && k < n) out[k] += ms;
    }
    return PT_OK;
}

int pt_last_kernel_ms(pt_ctx* c, float* ms) {
    if (!c || !ms) return fail(c, PT_ERR_INVALID, "pt_last_kernel_ms: null argument");
    if (!c->timed) return fail(c, PT_ERR_INVALID, "pt_last_kernel_ms: no timed launch (set PT_OPT_TIMING=1 first)");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    HIP_TRY(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return PT_OK;
}

}  // extern "C"

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

#include "../../include/ptmi.h"
#include "pt_kernels.h"
#include "pt_scene_build.h"
#include "pt_build.h"

static_assert(sizeof(pt_sphere) == 44, "pt_sphere must match the reference Sphere (44 B)");
static_assert(sizeof(pt_sphere_d) == sizeof(pt_sphere), "device sphere mirror");
static_assert(sizeof(pt_params) == 104 && sizeof(pt_camera) == 64 && sizeof(pt_counters) == 48, "ABI struct sizes (tests/test_host_and_abi.py)");

namespace {
thread_local std::string g_err;
}

namespace {
struct DevTemp {  // temporaries of one build, released together
    std::vector<void*> ptrs;
    ~DevTemp() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T> hipError_t get(T** out, size_t count) {
        void* p = nullptr;
        const hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(p);
        *out = (T*)p;
        return e;
    }
};
}  // namespace


struct pt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    // scene
    float4* d_nodes = nullptr;
    float4* d_tris = nullptr;
    pt_sphere_d* d_spheres = nullptr;
    int* d_tri_matid = nullptr;        // pt_upload_tri_materials
    float4* d_mat_table = nullptr;
    size_t n_tri_matid = 0;
    int32_t max_tri_id = -1;           // largest original triangle id of the uploaded BVH
    int n_spheres = 0;
    pt_sphere_d h_spheres[PT_KSPHERES];   // host copy of the first spheres for the kernel-argument block
    uint64_t n_inner = 0, n_refs = 0, n_leaves = 0, scene_bytes = 0;
    uint32_t max_depth = 0;
    uint32_t n_top_layout = 0;   // nodes [0, n_top_layout) are in breadth-first order
    uint64_t wide_root = 0;      // float4 index of the 4-wide tree's root, 0 = not built
    uint32_t wide_top_layout = 0, wide_depth = 0;
    uint64_t n_wide = 0;
    bool has_bvh = false;
    float build_ms = -1.f;       // device time of the last pt_build_bvh
    // options
    int opt_kernel = PT_KERNEL_AUTO;
    int opt_counters = 0;
    int opt_timing = 0;
    // measurement
    unsigned long long* d_counters = nullptr;
    unsigned int* d_queue = nullptr;   // persistent kernel's work counter
    float* d_samples = nullptr;        // [spp][H*W][3] sample colours of a multi-sample call
    size_t samples_bytes = 0;
    int n_cu = 0;
    int opt_batch = 36;
    int opt_presplit = 0;        // pt_build_bvh: 0 off, else the target length in per cent of diag/sqrt(n) (PT_OPT_PRESPLIT)
    int opt_rebuild = 0;         // pt_upload_bvh: 1 = re-cluster the uploaded triangles on the device (PT_OPT_REBUILD)
    int opt_build_algo = 1;      // pt_build_bvh: 0 LBVH (Karras), 1 PLOC (PT_OPT_BUILD_ALGO)
    int opt_sph_lds = 1;         // persistent kernel: sphere attributes from an LDS copy (PT_OPT_SPHERE_LDS)
    int opt_roles_batch = 16;    // role-split kernel: finished lanes that make a tracer wave leave the walk
    float4* d_roles_state = nullptr;   // role-split kernel: cold path state of every block's slots
    size_t roles_state_bytes = 0;
    bool roles_launched = false; // a role-split launch is in flight: pt_sync checks its error word
    int opt_vote_node = 1, opt_vote_rec = 1;
    int opt_refill = 8;          // idle lanes that trigger a refill (PT_OPT_REFILL)
    int opt_top = 64;            // nodes mirrored in LDS (PT_OPT_TOP_NODES)
    int opt_occ = 6;             // waves per SIMD the kernel is compiled for (PT_OPT_OCCUPANCY)
    int opt_lstk = 16;           // LDS stack entries per lane (deeper entries overflow to scratch)
    int opt_walk = 2;            // 0 while-while, 1 unified-step, 2 wide, 4 wide + postponed leaf (PT_OPT_WALK)
    int opt_leaf_max = 2;        // leaves with more references are split at upload (PT_OPT_LEAF_MAX)
    int opt_tri_test = 0;        // 0 Moller-Trumbore records, 1 Woop records (next upload; PT_OPT_TRI_TEST)
    bool records_woop = false;   // what the uploaded records are
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
};

namespace {

int fail(pt_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    g_err = msg;
    return code;
}
int hip_fail(pt_ctx* c, hipError_t e, const char* what) {
    return fail(c, PT_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(ctx, call)                                          \
    do {                                                            \
        hipError_t e_ = (call);                                     \
        if (e_ != hipSuccess) return hip_fail((ctx), e_, #call);    \
    } while (0)

int stack_for_depth(uint32_t depth) {
    // entries needed: sentinel + one push per level with two hit children
    const uint32_t need = depth + 2;
    if (need <= 24) return 24;
    if (need <= 32) return 32;
    if (need <= 48) return 48;
    return 72;
}

// LDS bytes of a frame-kernel block: top-of-tree planes + stack
size_t lds_bytes(int n_top, int stack_n, int block) { return (size_t)n_top * 64 + (size_t)stack_n * block * 4; }

template <typename K>
hipError_t allow_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

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
    (void)hipEventCreate(&c->ev0);
    (void)hipEventCreate(&c->ev1);
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
    (void)hipFree(c->d_counters);
    (void)hipFree(c->d_queue);
    (void)hipFree(c->d_samples);
    (void)hipFree(c->d_roles_state);
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
            if (value != 4 && value != 5 && value != 6 && value != 8) return fail(c, PT_ERR_INVALID, "pt_set_option: occupancy must be 4, 5, 6 or 8 waves per SIMD");
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
        case PT_OPT_REBUILD: c->opt_rebuild = value != 0; return PT_OK;
        case PT_OPT_PRESPLIT:
            if (value < 0 || value > 100000) return fail(c, PT_ERR_INVALID, "pt_set_option: presplit must be 0 (off) .. 100000 (per cent of diag/sqrt(n))");
            c->opt_presplit = value;
            return PT_OK;
        case PT_OPT_BUILD_ALGO:
            if (value != 0 && value != 1) return fail(c, PT_ERR_INVALID, "pt_set_option: build algorithm must be 0 (LBVH) or 1 (PLOC)");
            c->opt_build_algo = value;
            return PT_OK;
        case PT_OPT_ROLES_BATCH:
            if (value < 1 || value > 64) return fail(c, PT_ERR_INVALID, "pt_set_option: roles batch must be 1..64");
            c->opt_roles_batch = value;
            return PT_OK;
        case PT_OPT_VOTE_NODE:
        case PT_OPT_VOTE_REC:
            if (value < 1 || value > 64) return fail(c, PT_ERR_INVALID, "pt_set_option: vote weight must be 1..64");
            (option == PT_OPT_VOTE_NODE ? c->opt_vote_node : c->opt_vote_rec) = value;
            return PT_OK;
        case PT_OPT_LDS_STACK:
            if (value != 0 && value != 16) return fail(c, PT_ERR_INVALID, "pt_set_option: LDS stack must be 0 (all 72 entries in LDS) or 16 entries");
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
    if (c->roles_launched) {  // role-split kernel: did any wave run into its spin bound?
        c->roles_launched = false;
        unsigned long long e = 0;
        HIP_TRY(c, hipMemcpy(&e, c->d_counters + 15, sizeof e, hipMemcpyDeviceToHost));
        if (e) {
            (void)hipMemset(c->d_counters + 15, 0, sizeof e);
            return fail(c, PT_ERR_DEVICE, "role-split kernel: a wave gave up waiting on a block queue (code " + std::to_string(e) + "); the frame is incomplete");
        }
    }
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

static int build_bvh_impl(pt_ctx* c, const float* verts, size_t n_verts, const int32_t* tris, size_t n_tris, int algo, bool* too_deep,
                          const int32_t* id_map = nullptr);

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
    if (c->opt_rebuild && c->opt_tri_test == 0) {
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
    }
    ptscene::refine(T, (uint32_t)c->opt_leaf_max);
    ptscene::Output O;
    ptscene::emit(T, PT_MAX_TOP, O, c->opt_tri_test == 1);
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
    const size_t n_out = O.bin.size() / 16;
    const uint64_t n_refs = O.n_refs, n_leaves = O.n_leaves;
    const uint32_t max_depth = O.depth_bin;
    c->n_inner = n_out;
    c->n_refs = n_refs;
    c->n_leaves = n_leaves;
    c->max_depth = max_depth;
    c->scene_bytes = nb + tb + wb;
    c->max_tri_id = max_id;
    c->has_bvh = true;
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
    return rc;
}

static int build_bvh_impl(pt_ctx* c, const float* verts, size_t n_verts, const int32_t* tris, size_t n_tris, int algo, bool* too_deep,
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
    return PT_OK;
}

int pt_last_build_ms(pt_ctx* c, float* ms) {
    if (!c || !ms) return fail(c, PT_ERR_INVALID, "pt_last_build_ms: null argument");
    if (c->build_ms < 0.f) return fail(c, PT_ERR_INVALID, "pt_last_build_ms: no pt_build_bvh on this context yet");
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
        c->n_tri_matid = 0;
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
    return PT_OK;
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
    // spp > 1: trace the samples as independent work items, fold them afterwards (k_fold_samples)
    P.samples = nullptr;
    if (spp > 1) {
        const size_t need = (size_t)spp * (size_t)p->width * (size_t)p->height * 3 * sizeof(float);
        if (need > c->samples_bytes) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            (void)hipFree(c->d_samples);
            c->d_samples = nullptr;
            c->samples_bytes = 0;
            HIP_TRY(c, hipMalloc((void**)&c->d_samples, need));
            c->samples_bytes = need;
        }
        P.samples = c->d_samples;
    }
    if ((uint64_t)P.n_tiles * 64u * (uint64_t)spp >= (1ull << 31)) return fail(c, PT_ERR_INVALID, "pt_render: width*height*spp too large for one call (split the samples over several calls)");
    const int work_tiles = P.n_tiles * (P.samples ? (int)spp : 1);
    const int blocks = (work_tiles + waves_per_block - 1) / waves_per_block;

    if (c->opt_counters) HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, 16 * sizeof(unsigned long long), c->stream));
    if (c->opt_timing) HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    const bool persistent = c->opt_kernel == PT_KERNEL_PERSISTENT || c->opt_kernel == PT_KERNEL_AUTO || c->opt_kernel == PT_KERNEL_WAVEFRONT;
    const int need = stack_for_depth(c->has_bvh ? c->max_depth : 0);
    (void)need;  // any depth <= 64 works with every LDS window: deeper entries overflow
    int lstk = c->opt_lstk ? c->opt_lstk : PT_STACK_CAP;
    P.sc.stack_n = lstk;
    // the wide walk pushes up to three entries per level
    int walk = c->opt_walk;
    const bool wide_ok = c->has_bvh && 3 * c->wide_depth + 2 <= (uint32_t)PT_STACK_CAP;
    if (c->has_bvh && c->records_woop) {
        if (!wide_ok) return fail(c, PT_ERR_UNSUPPORTED, "pt_render: Woop records need the wide walk and this tree is too deep for it");
        walk = 3;  // Woop records are only understood by the wide walk
    } else if ((walk == 2 || walk == 4) && !wide_ok) {
        walk = 1;
    }
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
        HIP_TRY(c, hipMemsetAsync(c->d_queue, 0, PT_SHARDS * PT_SHARD_STRIDE * sizeof(unsigned int), c->stream));
    }
    // queue granularity: 64-slot chunks when the launch has plenty of them per resident wave, smaller
    // ones for small launches (an eighth of a 1080p frame per GPU is ~4 000 tiles for ~5 000 waves)
    {
        const long slots = (long)work_tiles * 64, waves = (long)c->n_cu * 20;
        P.chunk = slots / 64 >= 4 * waves ? 64 : (slots / 32 >= 4 * waves ? 32 : 16);
    }
    const int work_blocks = (int)(((long)work_tiles * 64 + P.chunk * (PT_BLOCK / 64) - 1) / (P.chunk * (PT_BLOCK / 64)));
    // persistent grid: as many blocks as can be resident (no grid-wide wait anywhere, so an
    // over-estimate only means a few late blocks find the queue empty and exit)
#define PT_LAUNCH(COUNT, OCC, LSTK, ALG)                                                                         \
    do {                                                                                                         \
        if (persistent) {                                                                                        \
            int per_cu = 0;                                                                                      \
            HIP_TRY(c, allow_lds(k_trace_persist_bvh2<COUNT, OCC, LSTK, ALG>, lds));                             \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_persist_bvh2<COUNT, OCC, LSTK, ALG>, \
                                                             PT_BLOCK, lds) != hipSuccess || per_cu < 1)        \
                per_cu = 1;                                                                                      \
            hipLaunchKernelGGL((k_trace_persist_bvh2<COUNT, OCC, LSTK, ALG>),                                    \
                               dim3(std::min(per_cu * c->n_cu, std::max(1, work_blocks))), dim3(PT_BLOCK), lds,  \
                               c->stream, P);                                                                    \
        } else {                                                                                                 \
            HIP_TRY(c, allow_lds(k_trace_mega_bvh2<COUNT, OCC, LSTK, ALG>, lds));                                \
            hipLaunchKernelGGL((k_trace_mega_bvh2<COUNT, OCC, LSTK, ALG>), dim3(blocks), dim3(PT_BLOCK), lds,    \
                               c->stream, P);                                                                    \
        }                                                                                                        \
    } while (0)
#define PT_LAUNCH_ALG(COUNT, OCC, LSTK)                   \
    do {                                                  \
        if (walk == 4) PT_LAUNCH(COUNT, OCC, LSTK, 4);    \
        else if (walk == 3) PT_LAUNCH(COUNT, OCC, LSTK, 3); \
        else if (walk == 2) PT_LAUNCH(COUNT, OCC, LSTK, 2); \
        else if (walk == 1) PT_LAUNCH(COUNT, OCC, LSTK, 1); \
        else PT_LAUNCH(COUNT, OCC, LSTK, 0);              \
    } while (0)
#define PT_LAUNCH_OCC(COUNT, LSTK)                            \
    do {                                                      \
        if (c->opt_occ == 8) PT_LAUNCH_ALG(COUNT, 8, LSTK);   \
        else if (c->opt_occ == 6) PT_LAUNCH_ALG(COUNT, 6, LSTK); \
        else if (c->opt_occ == 5) PT_LAUNCH_ALG(COUNT, 5, LSTK); \
        else PT_LAUNCH_ALG(COUNT, 4, LSTK);                   \
    } while (0)
    // role-split kernel (PT_KERNEL_WAVEFRONT): wide walk, exact records, a BVH and at least one bounce;
    // anything else runs the persistent kernel
    const bool roles = c->opt_kernel == PT_KERNEL_WAVEFRONT && walk == 2 && c->has_bvh && P.depth > 0 && !c->opt_counters && lstk == 16;
    if (roles) {
        P.batch = c->opt_roles_batch;
        P.sc.n_top = 0;
        const size_t rl = ((size_t)PT_ROLE_TRACERS * 64 * 16 + (size_t)PT_ROLE_SLOTS * (PT_SLOT_DW + 3) + RC_WORDS + 15 * PT_KSPHERES) * 4;
        const size_t max_blocks = (size_t)c->n_cu * 8;
        const size_t rs_need = max_blocks * PT_ROLE_SLOTS * PT_COLD_DW * sizeof(float);
        if (rs_need > c->roles_state_bytes) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            (void)hipFree(c->d_roles_state);
            c->d_roles_state = nullptr;
            c->roles_state_bytes = 0;
            HIP_TRY(c, hipMalloc((void**)&c->d_roles_state, rs_need));
            c->roles_state_bytes = rs_need;
        }
        P.roles_state = c->d_roles_state;
        c->roles_launched = true;   // the kernel ORs into counters[15] if a wave gave up waiting (sticky until pt_sync)
#define PT_LAUNCH_ROLES(OCC)                                                                                       \
        do {                                                                                                       \
            int per_cu = 0;                                                                                        \
            HIP_TRY(c, allow_lds(k_trace_roles<OCC, 16>, rl));                                                     \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_roles<OCC, 16>, PT_ROLE_BLOCK, rl) != hipSuccess || per_cu < 1) \
                per_cu = 1;                                                                                        \
            const int rblocks = (int)(((long)work_tiles * 64 + PT_ROLE_SLOTS - 1) / PT_ROLE_SLOTS);                \
            hipLaunchKernelGGL((k_trace_roles<OCC, 16>), dim3(std::min(std::min(per_cu, 8) * c->n_cu, std::max(1, rblocks))), dim3(PT_ROLE_BLOCK), rl, c->stream, P); \
        } while (0)
        if (c->opt_occ == 8) PT_LAUNCH_ROLES(8);
        else if (c->opt_occ == 6) PT_LAUNCH_ROLES(6);
        else if (c->opt_occ == 5) PT_LAUNCH_ROLES(5);
        else PT_LAUNCH_ROLES(4);
#undef PT_LAUNCH_ROLES
    } else if (c->opt_counters) {
        if (lstk == 16) PT_LAUNCH_OCC(true, 16);
        else PT_LAUNCH_OCC(true, PT_STACK_CAP);
    } else {
        if (lstk == 16) PT_LAUNCH_OCC(false, 16);
        else PT_LAUNCH_OCC(false, PT_STACK_CAP);
    }
#undef PT_LAUNCH_ALG
#undef PT_LAUNCH_OCC
#undef PT_LAUNCH
    if (P.samples) {
        HIP_TRY(c, hipGetLastError());
        hipLaunchKernelGGL(k_fold_samples, dim3((P.n_tiles + 3) / 4), dim3(256), 0, c->stream, P);
    }
    HIP_TRY(c, hipGetLastError());
    if (c->opt_timing) { HIP_TRY(c, hipEventRecord(c->ev1, c->stream)); c->timed = true; }
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
    const int blocks = (int)((n + PT_BLOCK_RAYS - 1) / PT_BLOCK_RAYS);
    if (c->opt_timing) HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    const float4* r4 = (const float4*)rays_dev;
    sc.stack_n = PT_STACK_CAP;
    sc.top_base = 0;
    sc.wide_root = (int)c->wide_root;
    sc.n_top = (int)std::min<uint32_t>((uint32_t)c->opt_top, c->n_top_layout);
    size_t lds = lds_bytes(sc.n_top, sc.stack_n, PT_BLOCK_RAYS);
    while (lds > 160 * 1024 && sc.n_top > 0) { sc.n_top /= 2; lds = lds_bytes(sc.n_top, sc.stack_n, PT_BLOCK_RAYS); }
    HIP_TRY(c, allow_lds(k_trace_rays_bvh2, lds));
    hipLaunchKernelGGL(k_trace_rays_bvh2, dim3(blocks), dim3(PT_BLOCK_RAYS), lds, c->stream, sc, r4, n, cull, t_dev, tri_dev, normal_dev);
    HIP_TRY(c, hipGetLastError());
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
    for (int i = 0; i < n && i < PT_WAVE_STATS; i++) out[i] = h[6 + i];
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

// pt_ctx.h — the context behind the C ABI (include/ptmi.h) and the host-side helpers shared by the
// translation units of libptmi.so: ptmi.hip (API), pt_build.hip (device BVH builder),
// pt_k_*.hip (kernel families + their launchers).  Host code only.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/ptmi.h"
#include "pt_kernels.h"

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
    // PT_FLAG_NEE over emissive triangles: ids whose material row emits (host, ascending), id -> light slot (device),
    // the light records (device; rebuilt from the triangle records when the scene or the materials changed)
    std::vector<int32_t> emissive_ids;
    int32_t* d_light_slot = nullptr;
    float4* d_tri_lights = nullptr;
    uint64_t mat_gen = 0, lights_key = ~0ull;
    // the light list is (re)written on the caller's stream; a side stream (PT_OPT_OVERLAP) waits for lights_ev before its
    // first kernel that may read that generation of the list
    hipEvent_t lights_ev = nullptr;
    uint64_t lights_gen = 0;
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
    int opt_optimize = 0;        // pt_upload_bvh: passes of insertion-based optimisation over the uploaded hierarchy (PT_OPT_OPTIMIZE)
    double opt_cost[2] = {0.0, 0.0};   // its area cost (inner-node areas / root area) before / after, 0 when it did not run
    int opt_rebuild = 0;         // pt_upload_bvh: 1 = re-cluster the uploaded triangles on the device (PT_OPT_REBUILD)
    int opt_build_algo = 1;      // pt_build_bvh: 0 LBVH (Karras), 1 PLOC (PT_OPT_BUILD_ALGO)
    int opt_sph_lds = 1;         // persistent kernel: sphere attributes from an LDS copy (PT_OPT_SPHERE_LDS)
    int opt_vote_node = 1, opt_vote_rec = 1;
    int opt_refill = 8;          // idle lanes that trigger a refill (PT_OPT_REFILL)
    int opt_top = 64;            // nodes mirrored in LDS (PT_OPT_TOP_NODES)
    int opt_occ = 6;             // waves per SIMD the kernel is compiled for (PT_OPT_OCCUPANCY)
    int opt_lstk = 16;           // LDS stack entries per lane (deeper entries overflow to scratch)
    int opt_walk = 2;            // 0 while-while, 1 unified-step, 2 wide, 4 wide + postponed leaf (PT_OPT_WALK)
    int opt_leaf_max = 2;        // leaves with more references are split at upload (PT_OPT_LEAF_MAX)
    int opt_tri_test = 0;        // 0 Moller-Trumbore records, 1 Woop records (next upload; PT_OPT_TRI_TEST)
    bool records_woop = false;   // what the uploaded records are
    // stage-split (wavefront) pipeline: path records of one call, two generations (pt_k_wave.hip)
    void* d_wave = nullptr;
    size_t wave_bytes = 0;
    int opt_wave_batch = 16;     // extend kernel: finished lanes that make a wave leave the walk to write hits / refill
    int opt_wave_samples = 16;   // bounce 0 of the stage-split pipeline: samples of one pixel per wave (PT_OPT_WAVE_SAMPLES)
    int opt_wave_blocks = 8;     // extend kernel: resident 256-thread blocks per CU the grid is sized for (PT_OPT_WAVE_BLOCKS)
    // PT_KERNEL_AUTO: which stage layout is faster depends on the workload (long paths and many samples per call:
    // the stage-split pipeline; short paths or few samples: the persistent kernel), so the first FOUR calls of a
    // configuration are timed trials, two per layout, alternating (HIP events on the stream, buffers allocated before the
    // timed span; the faster trial of each layout counts) and the following ones run the faster layout.  A small table of configurations (least recently used replaced): a context that cycles through
    // a few configurations — the partitions of a tile split, two image sizes — keeps every decision.
    struct AutoPick {
        static constexpr int TRIALS = 4, PENDING = 4, DECIDED = 5;
        uint64_t key = 0;        // what the choice was made for: image, spp, depth, partition shape, scene generation, material, flags
        int phase = 0;           // 0..3: this call is timed trial `phase` (even: persistent kernel, odd: pipeline)  4: events pending  5: decided
        int choice = PT_KERNEL_PERSISTENT;
        hipEvent_t e[2 * TRIALS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // start / end of every trial
        float ms[2] = {0.f, 0.f};   // the faster of each layout's two trials (a context's very first launch is slow: code upload)
        uint64_t used = 0;       // tick of the last pt_render that looked this entry up
    };
    static constexpr int N_PICKS = 8;
    AutoPick picks[N_PICKS];
    int pick_last = -1;          // entry of the last PT_KERNEL_AUTO call (pt_auto_choice), -1 = none
    uint64_t pick_tick = 0;
    uint64_t scene_gen = 0;      // bumped by every upload / build
    // Overlap of consecutive calls (PT_OPT_OVERLAP, persistent / mega kernels): the path kernel of call k + 1 runs on
    // a side stream into its own sample buffer while call k's last paths drain; only the folds (which touch the
    // accumulator, in order) stay on the caller's stream.  side[x]: stream, sample buffer, queue counters of slot x.
    int opt_overlap = 1;
    struct Side {
        hipStream_t stream = nullptr;
        hipEvent_t traced = nullptr, folded = nullptr;   // path kernel done / the fold that read the buffer done
        bool fold_pending = false;
        float* samples = nullptr;
        size_t samples_bytes = 0;
        unsigned int* queue = nullptr;
        uint64_t lights_seen = 0;   // the light list generation this stream has been ordered behind (lights_gen)
    } side[2];
    int side_next = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    // PT_OPT_TIMING: events between the stages of the last call (pt_get_stage_ms); stage_kind[i] is the
    // PT_STAGE_* of the work between event i and event i + 1
    std::vector<hipEvent_t> stage_ev;
    std::vector<int> stage_kind;
    size_t stage_used = 0;
};

namespace ptmi {

extern thread_local std::string g_err;

inline int fail(pt_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    g_err = msg;
    return code;
}
inline int hip_fail(pt_ctx* c, hipError_t e, const char* what) {
    return fail(c, PT_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

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

// LDS bytes of a frame-kernel block: top-of-tree planes + stack
inline size_t lds_bytes(int n_top, int stack_n, int block) { return (size_t)n_top * 64 + (size_t)stack_n * block * 4; }

template <typename K>
hipError_t allow_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// ---- launchers of the kernel families (one translation unit each) -------------------------------
struct LaunchCfg {
    bool count;        // instrumented instantiation (PT_OPT_COUNTERS)
    int occ;           // waves per SIMD the registers are budgeted for
    int lstk;          // LDS stack entries per lane (16, 24 or PT_STACK_CAP)
    int walk;          // 0 while-while, 1 unified, 2 wide, 3 wide over Woop records, 4 wide + postponed leaf
    size_t lds;        // dynamic LDS bytes of a block
    int blocks;        // megakernel grid (one wave per work tile)
    int work_blocks;   // persistent grid cap: blocks that have work at all
    int n_cu;
};
hipError_t launch_mega(const LaunchCfg& L, const KParams& P, hipStream_t st);        // pt_k_mega.hip
hipError_t launch_rays(const KScene& sc, size_t lds, const float4* rays, size_t n, int cull, float* t_out, int* tri_out,
                       float* n_out, hipStream_t st);                                 // pt_k_mega.hip
hipError_t launch_persist(const LaunchCfg& L, const KParams& P, hipStream_t st);     // pt_k_persist.hip
hipError_t launch_fold(const KParams& P, hipStream_t st);                            // pt_k_persist.hip
// stage-split pipeline (pt_k_wave.hip): generate -> depth x (extend, shade) ; returns PT_* status
int render_wavefront(pt_ctx* c, KParams& P, const LaunchCfg& L, int work_tiles);
// samples of one pixel that share a wave at bounce 0 of the stage-split pipeline, as a power of two (wf_slot_pixel): the largest
// 2^k <= PT_OPT_WAVE_SAMPLES, k >= 2, that divides spp; else 0 (one sample of a whole tile per wave, the other kernels' order)
inline uint32_t wave_sample_group_log2(uint32_t spp, int cap) {
    for (int k = 6; k >= 2; k--)
        if ((1 << k) <= cap && (spp & ((1u << k) - 1u)) == 0u) return (uint32_t)k;
    return 0u;
}
int wave_reserve(pt_ctx* c, const KParams& P, int work_tiles);   // path records for this call, allocated now
// PT_OPT_TIMING: marks the end of a stage of the running call on the context's stream (no-op when timing is off)
int stage_mark(pt_ctx* c, int kind_of_work_since_last_mark);
// device BVH builder (pt_build.hip)
int build_bvh_impl(pt_ctx* c, const float* verts, size_t n_verts, const int32_t* tris, size_t n_tris, int algo, bool* too_deep,
                   const int32_t* id_map = nullptr);

}  // namespace ptmi

#define HIP_TRY(ctx, call)                                                \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) return ptmi::hip_fail((ctx), e_, #call);    \
    } while (0)

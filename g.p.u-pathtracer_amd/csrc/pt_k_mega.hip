// pt_k_mega.hip — the megakernel family (one wave per 8x8 tile, bounce by bounce) and the explicit
// ray-batch kernel, with their launchers.  One translation unit of libptmi.so (pt_ctx.h).
#include "pt_ctx.h"

// trace<<<>>>, tracer.cu:343-400: one lane per pixel, one wave per 8x8 tile, `spp`
// consecutive samples folded in registers.
// OCC = waves per SIMD the register allocator must leave room for (4 / 6 / 8)
// ALG = 0 while-while walk (Aila-Laine), 1 unified-step walk, 2 wide (4-way quantised) walk,
//       3 wide walk over Woop records
template <bool COUNT, int OCC, int LSTK, int ALG>
__global__ void __launch_bounds__(PT_BLOCK, OCC) k_trace_mega_bvh2(const KParams P) {
    float4* s_top = s_dyn;
    lds_load_top<PT_BLOCK>(P.sc, s_top);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    int tile = blockIdx.x * (PT_BLOCK / 64) + (tid >> 6);
    uint32_t s_only = 0;
    if (P.samples) {  // one wave per (sample, tile)
        s_only = (uint32_t)(tile / P.n_tiles);
        tile -= (int)s_only * P.n_tiles;
        if (s_only >= P.spp) return;
    }
    int tx, ty;
    if (!pt_tile_coords(P, tile, tx, ty)) return;
    const int px = tx * PT_TILE + (lane & 7), py = ty * PT_TILE + (lane >> 3);
    if (px >= P.W || py >= P.H) return;  // tracer.cu:358
    const uint64_t pix = (uint64_t)py * (uint64_t)P.W + (uint64_t)px;
    TravOverflow<LSTK> stk_ovf;
    TravStack<LSTK, PT_BLOCK> stk(__builtin_amdgcn_readfirstlane(16 * P.sc.n_top + (tid & ~63)), stk_ovf);

    TravCount tc;
    tc.inner = tc.tris = tc.leaves = 0;
    uint32_t n_rays = 0, n_hits = 0;

    uint32_t n_done = P.spp;
    if (P.samples) {
        const v3 col = pt_get_sample<COUNT, ALG>(P, px, py, pix, P.frame + s_only, stk, s_top, tc, n_rays, n_hits);
        float* dst = pt_sample_ptr(P, (uint32_t)s_only, (size_t)pix);
        dst[0] = col.x; dst[1] = col.y; dst[2] = col.z;
        n_done = 1;
    } else {
        float* acc = P.accum + 3 * pix;
        float ax = 0.f, ay = 0.f, az = 0.f;
        if (P.sample_index != 1) { ax = acc[0]; ay = acc[1]; az = acc[2]; }
        for (uint32_t s = 0; s < P.spp; s++) {
            const v3 col = pt_get_sample<COUNT, ALG>(P, px, py, pix, P.frame + s, stk, s_top, tc, n_rays, n_hits);
            pt_accumulate(ax, ay, az, col, P.sample_index + s);
        }
        acc[0] = ax; acc[1] = ay; acc[2] = az;
        if ((P.flags & PT_FLAG_WRITE_RGBA) && P.rgba) P.rgba[pix] = pt_pack_rgba(ax, ay, az);
    }
    if (COUNT) {
        const uint32_t a = wave_sum_u32(n_rays), b = wave_sum_u32(tc.inner), c = wave_sum_u32(tc.tris);
        const uint32_t dd = wave_sum_u32(tc.leaves), e = wave_sum_u32(n_hits), f = wave_sum_u32(n_done);
        if (__ffsll((long long)__ballot(1)) - 1 == lane) {
            atomicAdd(&P.counters[0], (unsigned long long)a);
            atomicAdd(&P.counters[1], (unsigned long long)b);
            atomicAdd(&P.counters[2], (unsigned long long)c);
            atomicAdd(&P.counters[3], (unsigned long long)dd);
            atomicAdd(&P.counters[4], (unsigned long long)e);
            atomicAdd(&P.counters[5], (unsigned long long)f);
        }
    }
}

// Closest-hit on an explicit ray batch (pt_trace_rays): rows a5–a7 in isolation.
__global__ void __launch_bounds__(PT_BLOCK_RAYS) k_trace_rays_bvh2(const KScene sc, const float4* __restrict__ rays, size_t n,
                                                                   int cull, float* __restrict__ t_out,
                                                                   int* __restrict__ tri_out, float* __restrict__ n_out) {
    float4* s_top = s_dyn;
    lds_load_top<PT_BLOCK_RAYS>(sc, s_top);
    const size_t i = (size_t)blockIdx.x * PT_BLOCK_RAYS + threadIdx.x;
    if (i >= n) return;
    const float4 ro = rays[2 * i], rd = rays[2 * i + 1];
    TravCount tc;
    tc.inner = tc.tris = tc.leaves = 0;
    TravOverflow<PT_STACK_CAP> stk_ovf;
    TravStack<PT_STACK_CAP, PT_BLOCK_RAYS> stk(__builtin_amdgcn_readfirstlane(16 * sc.n_top + ((int)threadIdx.x & ~63)), stk_ovf);
    const Hit h = trav_bvh2<false, true>(sc, V3(ro.x, ro.y, ro.z), V3(rd.x, rd.y, rd.z), cull != 0, stk, tc, s_top);
    t_out[i] = h.t;
    tri_out[i] = h.tri;
    if (n_out) {
        const v3 hn = h.tri != -1 ? pt_hit_normal(sc, h) : V3(0.f, 0.f, 0.f);
        n_out[3 * i] = hn.x; n_out[3 * i + 1] = hn.y; n_out[3 * i + 2] = hn.z;
    }
}

namespace ptmi {

// Instantiated register / stack budgets: (8 waves/SIMD, 16-entry LDS window), (4, 16), (4, all 72 in LDS);
// other requests run the nearest one (they are speed knobs, never results).
hipError_t launch_mega(const LaunchCfg& L, const KParams& P, hipStream_t st) {
#define PT_GO(COUNT, OCC, LSTK, ALG)                                                                             \
    do {                                                                                                         \
        hipError_t e_ = allow_lds(k_trace_mega_bvh2<COUNT, OCC, LSTK, ALG>, L.lds);                              \
        if (e_ != hipSuccess) return e_;                                                                         \
        hipLaunchKernelGGL((k_trace_mega_bvh2<COUNT, OCC, LSTK, ALG>), dim3(L.blocks), dim3(PT_BLOCK), L.lds, st, P); \
        return hipGetLastError();                                                                                \
    } while (0)
#define PT_GO_ALG(COUNT, OCC, LSTK)                        \
    do {                                                   \
        if (L.walk >= 2) PT_GO(COUNT, OCC, LSTK, 2);       \
        else if (L.walk == 1) PT_GO(COUNT, OCC, LSTK, 1);  \
        else PT_GO(COUNT, OCC, LSTK, 0);                   \
    } while (0)
#define PT_GO_CFG(COUNT)                                                   \
    do {                                                                   \
        if (L.lstk >= PT_STACK_CAP) PT_GO_ALG(COUNT, 4, PT_STACK_CAP);     \
        else if (L.occ >= 5) PT_GO_ALG(COUNT, 8, 16);                      \
        else PT_GO_ALG(COUNT, 4, 16);                                      \
    } while (0)
    if (L.count) PT_GO_CFG(true);
    else PT_GO_CFG(false);
#undef PT_GO_CFG
#undef PT_GO_ALG
#undef PT_GO
}

hipError_t launch_rays(const KScene& sc, size_t lds, const float4* rays, size_t n, int cull, float* t_out, int* tri_out,
                       float* n_out, hipStream_t st) {
    hipError_t e = allow_lds(k_trace_rays_bvh2, lds);
    if (e != hipSuccess) return e;
    const int blocks = (int)((n + PT_BLOCK_RAYS - 1) / PT_BLOCK_RAYS);
    hipLaunchKernelGGL(k_trace_rays_bvh2, dim3(blocks), dim3(PT_BLOCK_RAYS), lds, st, sc, rays, n, cull, t_out, tri_out, n_out);
    return hipGetLastError();
}

}  // namespace ptmi

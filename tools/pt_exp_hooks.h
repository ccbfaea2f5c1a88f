// pt_exp_hooks.h — sensitivity experiments on the wide walk's node step.  NOT part of the product build:
// tools/build_variant.sh force-includes this file (-include) with one PT_EXP_* macro defined, which fills the
// PT_NODE_STEP_HOOK(sc, a, w) splice point of csrc/pt_walks.h (sc: KScene, a: float4 index of the node, w: the decoded node).
#pragma once
#if defined(PT_EXP_LOAD)      // one more 16-byte access to the node's own line per node step
#define PT_NODE_STEP_HOOK(sc, a, w)                                                                                       \
    { float4 dummy; const float4* ptr_ = (sc).nodes + (a) + 2;                                                              \
      asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(dummy) : "v"(ptr_) : "memory");       \
      asm volatile("" :: "v"(dummy.x), "v"(dummy.w)); }
#elif defined(PT_EXP_SALU)    // 32 more dependent scalar instructions per node step
#define PT_NODE_STEP_HOOK(sc, a, w)                                                                                       \
    { int z = 1;                                                                                                          \
      _Pragma("unroll") for (int e = 0; e < 32; e++) asm volatile("s_add_i32 %0, %0, 1" : "+s"(z) :: "scc");              \
      asm volatile("" :: "s"(z)); }
#elif defined(PT_EXP_BRANCH)  // 8 more (never taken) exec-mask branch pairs per node step
#define PT_NODE_STEP_HOOK(sc, a, w)                                                                                       \
    { int zb = (w).l0;                                                                                                    \
      _Pragma("unroll") for (int e = 0; e < 8; e++) { if (zb == 0x7fffff01 + e) { asm volatile("v_mov_b32 %0, 0" : "+v"(zb)); } asm volatile("" : "+v"(zb)); } \
      asm volatile("" :: "v"(zb)); }
#elif defined(PT_EXP_VALU)    // 32 more dependent VALU instructions per node step
#define PT_NODE_STEP_HOOK(sc, a, w)                                                                                       \
    { float z = (w).ox;                                                                                                   \
      _Pragma("unroll") for (int e = 0; e < 32; e++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(z));                  \
      asm volatile("" :: "v"(z)); }
#elif defined(PT_EXP_TOUCH)   // a lane whose node step ends on a LEAF touches the leaf's first record (one dword, never waited
                              // for inside the loop): the record step that follows a vote or two later finds the line on its way
#define PT_NODE_STEP_HOOK(sc, a, w)
#define PT_WALK_DECL_HOOK() float touch_ = 0.f; asm volatile("" : "+v"(touch_))
#define PT_NODE_END_HOOK(sc, cur)                                                                                         \
    { if ((cur) < 0) { const float4* tp_ = (sc).nodes + (~(cur) & ~3);                                                      \
          asm volatile("global_load_dword %0, %1, off" : "+v"(touch_) : "v"(tp_) : "memory"); } }
#define PT_WALK_EXIT_HOOK() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); asm volatile("" : "+v"(touch_)); }
#elif defined(PT_EXP_TOUCH_ALL)   // every node step touches the item its lane goes to next (node or leaf)
#define PT_NODE_STEP_HOOK(sc, a, w)
#define PT_WALK_DECL_HOOK() float touch_ = 0.f; asm volatile("" : "+v"(touch_))
#define PT_NODE_END_HOOK(sc, cur)                                                                                         \
    { if ((cur) != PT_SENTINEL) { const float4* tp_ = (sc).nodes + ((cur) >= 0 ? (cur) : (~(cur) & ~3));                    \
          asm volatile("global_load_dword %0, %1, off" : "+v"(touch_) : "v"(tp_) : "memory"); } }
#define PT_WALK_EXIT_HOOK() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); asm volatile("" : "+v"(touch_)); }
#endif

#if defined(PT_EXP_SORT)
#include <hip/hip_runtime.h>
// Ray ordering at the shade stage's compaction: the survivors of a region (256 paths) are packed by
// (cell of the new ray's origin in a PT_EXP_SORT^3 grid over the scene's bounds, octant of its direction) instead of slot order —
// a counting sort in LDS (atomics: the order INSIDE a bin is arbitrary; no result depends on where a record sits).
template <class KP, class PS>
__device__ __forceinline__ int exp_rank_sorted(const KP& P, bool alive, const PS& ps, int& total) {
    constexpr int C = PT_EXP_SORT, B = C * C * C * 8, PER = (B + 255) / 256;
    __shared__ int s_hist[B];
    __shared__ int s_wave[4];
    for (int b = threadIdx.x; b < B; b += 256) s_hist[b] = 0;
    __syncthreads();
    int key = 0, my = 0;
    if (alive) {
        const float4 q0 = P.sc.nodes[P.sc.wide_root], q3 = P.sc.nodes[P.sc.wide_root + 3];   // origin and grid step (x 255 = extent)
        const int cx = min(C - 1, max(0, (int)((ps.o.x - q0.x) / (q0.w * 255.f) * (float)C)));
        const int cy = min(C - 1, max(0, (int)((ps.o.y - q0.y) / (q3.z * 255.f) * (float)C)));
        const int cz = min(C - 1, max(0, (int)((ps.o.z - q0.z) / (q3.w * 255.f) * (float)C)));
        const int oct = (ps.d.x < 0.f ? 1 : 0) | (ps.d.y < 0.f ? 2 : 0) | (ps.d.z < 0.f ? 4 : 0);
        key = ((cz * C + cy) * C + cx) * 8 + oct;
        my = atomicAdd(&s_hist[key], 1);
    }
    __syncthreads();
    int loc[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int b = (int)threadIdx.x * PER + k;
        loc[k] = sum;
        sum += b < B ? s_hist[b] : 0;
    }
    int inc = sum;   // inclusive scan over the wave, then over the four waves
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(inc, d);
        if ((int)(threadIdx.x & 63) >= d) inc += v;
    }
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = inc;
    __syncthreads();
    int base = inc - sum, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if (w < (int)(threadIdx.x >> 6)) base += s_wave[w];
        tot += s_wave[w];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int b = (int)threadIdx.x * PER + k;
        if (b < B) s_hist[b] = base + loc[k];
    }
    __syncthreads();
    total = tot;
    return s_hist[key] + my;
}
#define PT_SURVIVOR_RANK(P, alive, ps, total, s_cnt) exp_rank_sorted(P, alive, ps, total)
#endif

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

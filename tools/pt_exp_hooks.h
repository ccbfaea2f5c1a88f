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
#endif

// pt_roles.h — the role-split frame kernel (PT_KERNEL_WAVEFRONT).  Included by pt_kernels.h.
#pragma once

// ---------------------------------------------------------------------------------------
// Role-split variant (PT_KERNEL_WAVEFRONT): shading and traversal no longer share a wave's lanes.
// The schedule statistics of k_trace_persist_bvh2 put the shading passes at ~46 % of
// the issued VALU work at 56 % lane use and the node steps at 55 %: both draw on the same 64 lanes,
// so `batch` trades one against the other.  Here a block's waves take fixed roles and exchange rays
// only BETWEEN segments (the traversal stack is empty then), through LDS:
//   * every ray of the block owns one SLOT from its first camera ray to its last bounce: origin,
//     direction and hit (9 dwords) in LDS — what the two roles hand to each other — and the rest of
//     the path state (mask, accu, rng, depth, pixel, sample: 12 dwords) in a global array that only
//     the shader waves touch (it stays in L2);
//   * TRACER waves (PT_ROLE_TRACERS of the 4) pop slot numbers from the ready queue, load o/d, walk,
//     write the hit back and push the slot to the shade queue — they never hold path state;
//   * SHADER waves pop up to 64 finished segments, shade them at full width (path_shade, the same
//     arithmetic), and push the continuing ones to the ready queue; when nothing waits for shading
//     they start new paths (path_begin, 64 at a time) from the global work queue.
// Queues are rings of slot numbers with one entry per slot, so they can never overflow and no
// wave ever blocks on a full queue; a consumer that reserved an entry the producer has not
// written yet spins on that entry only.  Every spin is bounded (PT_ROLE_SPIN_MAX polls), after
// which the wave raises the error word and every wave of the block drains out: the grid always ends.
// Per-ray arithmetic is that of the other kernels (same path_begin / walk / path_shade), so the
// image is bit-identical; only WHICH lane does it changes.
#ifndef PT_ROLE_TRACERS
#define PT_ROLE_TRACERS 3            // tracer waves of the block's 4 (the rest shade)
#endif
#ifndef PT_ROLE_SLOTS
#define PT_ROLE_SLOTS 272            // rays in flight per block (LDS: 48 B each + 12 KB of tracer stacks = 25.4 KB)
#endif
#define PT_SLOT_DW 9                 // LDS part of a slot: o, d, hit; the rest (PT_COLD_DW) lives in global memory
#define PT_COLD_DW 12                // mask, accu, rng s0 s1 n, depth, pixel, sample
#define PT_ROLE_SPIN_MAX (1 << 22)
#ifndef PT_ROLE_S_MIN
#define PT_ROLE_S_MIN 48             // finished segments that make a shading pass worth starting
#define PT_ROLE_B_MIN 32             // free slots that make a path-start pass worth starting
#define PT_ROLE_T_LOW 16             // ready segments below which the shaders stop waiting for full passes
#define PT_ROLE_HELP_MIN 8           // finished segments that make an idle TRACER wave take a shading pass
#endif
#ifndef PT_ROLE_BLOCK
#define PT_ROLE_BLOCK 256            // threads per block: PT_ROLE_TRACERS tracer waves, the rest shade
#endif
#ifndef PT_ROLE_MIX
#define PT_ROLE_MIX 0
#endif
enum { RC_SQ_HEAD = 0, RC_SQ_TAIL, RC_TQ_HEAD, RC_TQ_TAIL, RC_FQ_HEAD, RC_FQ_TAIL, RC_ALIVE, RC_DRY, RC_ERROR, RC_WORDS = 16 };

#define LDSI(i) (((int*)s_dyn)[(i)])
#define LDSF(i) (((float*)s_dyn)[(i)])

// reserve up to `want` entries of ring [head, tail); returns the count and the first position (wave-uniform)
__device__ __forceinline__ int role_reserve(int head_i, int tail_i, int want, int lane, int& pos) {
    int n = 0, h = 0;
    if (lane == 0 && want > 0) {
        for (int tries = 0; tries < 64; tries++) {
            h = __atomic_load_n(&LDSI(head_i), __ATOMIC_RELAXED);
            const int t = __atomic_load_n(&LDSI(tail_i), __ATOMIC_RELAXED);
            n = min(want, t - h);
            if (n <= 0) { n = 0; break; }
            if (atomicCAS(&LDSI(head_i), h, h + n) == h) break;
            n = 0;
        }
    }
    pos = __builtin_amdgcn_readfirstlane(h);
    return __builtin_amdgcn_readfirstlane(n);
}

// the slot number stored at ring position p (spins until the producer has written it), entry cleared
__device__ __forceinline__ int role_take(int ring_i, int p, int err_i) {
    const int e = ring_i + (p % PT_ROLE_SLOTS);
    int v = 0;
    for (int spin = 0; spin < PT_ROLE_SPIN_MAX; spin++) {
        v = __atomic_load_n(&LDSI(e), __ATOMIC_ACQUIRE);
        if (v != 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
    if (v == 0) { atomicOr(&LDSI(err_i), 1); return -1; }
    __atomic_store_n(&LDSI(e), 0, __ATOMIC_RELAXED);
    return v - 1;
}

// lanes with `pred` append `slot` to a ring (one tail reservation per wave)
__device__ __forceinline__ void role_push(int ring_i, int tail_i, bool pred, int slot, int lane) {
    const unsigned long long m = __ballot(pred);
    const int n = __popcll(m);
    if (n == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(&LDSI(tail_i), n);
    base = __builtin_amdgcn_readfirstlane(base);
    if (pred) {
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        __atomic_store_n(&LDSI(ring_i + ((base + rank) % PT_ROLE_SLOTS)), slot + 1, __ATOMIC_RELEASE);
    }
}

// One shading pass of a wave: `got` finished segments starting at shade-queue position `pos`.
// Continuing rays go to the ready queue, finished paths write their sample and free their slot.
template <int SLOT_OFF, int SQ_OFF, int TQ_OFF, int FQ_OFF, int CTL>
__device__ __forceinline__ void role_shade_pass(const KParams& P, int lane, int got, int pos) {
    constexpr int F = PT_ROLE_SLOTS;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // cold state written by another wave of this block (same CU, same L1)
    int slot = -1;
    bool cont = false;
    if (lane < got) slot = role_take(SQ_OFF, pos + lane, CTL + RC_ERROR);
    if (slot >= 0) {
        const int b = SLOT_OFF + slot * PT_SLOT_DW;
        PathState ps;
        ps.o = V3(LDSF(b + 0), LDSF(b + 1), LDSF(b + 2));
        ps.d = V3(LDSF(b + 3), LDSF(b + 4), LDSF(b + 5));
        Hit h;
        h.t = LDSF(b + 6); h.tri = LDSI(b + 7); h.rec = LDSI(b + 8);
        float4* cold = P.roles_state + ((size_t)blockIdx.x * F + (size_t)slot) * (PT_COLD_DW / 4);
        const float4 c0 = cold[0], c1 = cold[1], c2 = cold[2];
        ps.mask = V3(c0.x, c0.y, c0.z);
        ps.accu = V3(c0.w, c1.x, c1.y);
        ps.rng.s0 = __float_as_uint(c1.z); ps.rng.s1 = __float_as_uint(c1.w); ps.rng.n = __float_as_uint(c2.x);
        ps.depth = __float_as_uint(c2.y);
        const uint32_t pix = __float_as_uint(c2.z), s_idx = __float_as_uint(c2.w);
        v3 col = V3(0.f, 0.f, 0.f);
        const bool done = path_shade(P, ps, h, col, CTL + RC_WORDS);
        if (!done) {
            LDSF(b + 0) = ps.o.x; LDSF(b + 1) = ps.o.y; LDSF(b + 2) = ps.o.z;
            LDSF(b + 3) = ps.d.x; LDSF(b + 4) = ps.d.y; LDSF(b + 5) = ps.d.z;
            cold[0] = make_float4(ps.mask.x, ps.mask.y, ps.mask.z, ps.accu.x);
            cold[1] = make_float4(ps.accu.y, ps.accu.z, c1.z, c1.w);
            cold[2] = make_float4(__uint_as_float(ps.rng.n), __uint_as_float(ps.depth), c2.z, c2.w);
            cont = true;
        } else if (P.samples) {
            float* dst = P.samples + 3 * ((size_t)s_idx * (size_t)P.W * (size_t)P.H + (size_t)pix);
            dst[0] = col.x; dst[1] = col.y; dst[2] = col.z;
        } else {
            float* acc = P.accum + 3 * (size_t)pix;
            float ax = 0.f, ay = 0.f, az = 0.f;
            if (P.sample_index != 1) { ax = acc[0]; ay = acc[1]; az = acc[2]; }
            pt_accumulate(ax, ay, az, col, P.sample_index);
            acc[0] = ax; acc[1] = ay; acc[2] = az;
            if ((P.flags & PT_FLAG_WRITE_RGBA) && P.rgba) P.rgba[pix] = pt_pack_rgba(ax, ay, az);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // cold state stored before the slot is handed on
    role_push(TQ_OFF, CTL + RC_TQ_TAIL, cont, slot, lane);
    const bool dead = slot >= 0 && !cont;
    role_push(FQ_OFF, CTL + RC_FQ_TAIL, dead, slot, lane);   // slot free again ...
    const int n_deadr = __popcll(__ballot(dead));
    if (lane == 0 && n_deadr) atomicSub(&LDSI(CTL + RC_ALIVE), n_deadr);   // ... then the ray count drops
}

template <int OCC, int LSTK>
__global__ void __launch_bounds__(PT_ROLE_BLOCK, OCC) k_trace_roles(const KParams P) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NT = PT_ROLE_TRACERS, F = PT_ROLE_SLOTS;
    // blocks alternate between NT tracers and NT-1 (PT_ROLE_MIX): the shading share of the work sits
    // between one and two waves of four
    const int nt = NT - ((PT_ROLE_MIX && (blockIdx.x & 1)) ? 1 : 0);
    const int NS = PT_ROLE_BLOCK / 64 - nt;
    constexpr int STK_OFF = 0, SLOT_OFF = NT * 64 * LSTK, SQ_OFF = SLOT_OFF + F * PT_SLOT_DW, TQ_OFF = SQ_OFF + F, FQ_OFF = TQ_OFF + F,
                  CTL = FQ_OFF + F;
    if (tid < 11 * PT_KSPHERES) {  // the spheres' attributes (11 floats each), then centre+radius as float4s
        PT_KARGS(K);
        const float v = ((const __attribute__((address_space(4))) float*)&K.ksph[0])[tid];
        LDSF(CTL + RC_WORDS + tid) = v;
        if (tid % 11 < 4) LDSF(CTL + RC_WORDS + 88 + 4 * (tid / 11) + tid % 11) = v;
    }
    for (int i = tid; i < F; i += PT_ROLE_BLOCK) { LDSI(SQ_OFF + i) = 0; LDSI(TQ_OFF + i) = 0; LDSI(FQ_OFF + i) = i + 1; }
    if (tid < RC_WORDS) LDSI(CTL + tid) = tid == RC_FQ_TAIL ? F : 0;
    __syncthreads();
    const bool cull = P.cull != 0;

    if (wave < nt) {
        // ------------------------------------------------------------------ tracer wave
        TravOverflow<LSTK> stk_ovf;
        TravStack<LSTK, NT * 64> stk(__builtin_amdgcn_readfirstlane(STK_OFF + wave * 64), stk_ovf);
        TravState ts;
        ts.idx = ts.idy = ts.idz = ts.oodx = ts.oody = ts.oodz = 0.f;
        ts.node = PT_SENTINEL; ts.leaf = 0; ts.sp = 0;
        ts.h.t = PT_F32_MAX; ts.h.tri = -1; ts.h.rec = 0;
        TravCount tc;
        tc.inner = tc.tris = tc.leaves = 0;
        v3 o = V3(0.f, 0.f, 0.f), d = V3(0.f, 0.f, 1.f);
        int slot = -1;           // -1: the lane is empty
        bool walking = false;
        int idle_polls = 0;
#ifdef PT_ROLES_STATS
        uint32_t st_iters = 0, st_idle = 0, st_live = 0, st_help = 0;
#endif
        for (;;) {
            // 1. finished segments -> shade queue
            const bool fin = slot >= 0 && !walking;
            if (fin) {   // (running the sphere tests here, on the lanes that just finished, costs +11 %: measured)
                const int b = SLOT_OFF + slot * PT_SLOT_DW;
                LDSF(b + 6) = ts.h.t; LDSI(b + 7) = ts.h.tri; LDSI(b + 8) = ts.h.rec;
            }
            role_push(SQ_OFF, CTL + RC_SQ_TAIL, fin, slot, lane);
            if (fin) slot = -1;
            // 2. empty lanes <- ready queue
            const unsigned long long em = __ballot(slot < 0);
            const int n_empty = __popcll(em);
            int pos = 0;
            const int seen_tail = __builtin_amdgcn_readfirstlane(__atomic_load_n(&LDSI(CTL + RC_TQ_TAIL), __ATOMIC_RELAXED));
            const int got = role_reserve(CTL + RC_TQ_HEAD, CTL + RC_TQ_TAIL, n_empty, lane, pos);
            if (slot < 0) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
                if (rank < got) {
                    slot = role_take(TQ_OFF, pos + rank, CTL + RC_ERROR);
                    if (slot >= 0) {
                        const int b = SLOT_OFF + slot * PT_SLOT_DW;
                        o = V3(LDSF(b + 0), LDSF(b + 1), LDSF(b + 2));
                        d = V3(LDSF(b + 3), LDSF(b + 4), LDSF(b + 5));
                        trav_begin(ts, o, d, stk, P.sc.wide_root);
                        walking = true;
                    }
                }
            }
            const int n_live = __popcll(__ballot(slot >= 0));
#ifdef PT_ROLES_STATS
            st_iters++; st_live += n_live; if (n_live == 0) st_idle++;
#endif
            if (__atomic_load_n(&LDSI(CTL + RC_ERROR), __ATOMIC_RELAXED) != 0) break;
            if (n_live == 0) {
                // nothing to walk.  A tracer wave without rays is free to shade: if finished segments wait,
                // take a pass of them (the roles balance themselves: starving tracers refill their own queue)
                {
                    int sq_n = 0;
                    if (lane == 0) sq_n = __atomic_load_n(&LDSI(CTL + RC_SQ_TAIL), __ATOMIC_RELAXED) - __atomic_load_n(&LDSI(CTL + RC_SQ_HEAD), __ATOMIC_RELAXED);
                    sq_n = __builtin_amdgcn_readfirstlane(sq_n);
                    if (sq_n >= PT_ROLE_HELP_MIN) {
                        int hpos = 0;
                        const int hgot = role_reserve(CTL + RC_SQ_HEAD, CTL + RC_SQ_TAIL, 64, lane, hpos);
                        if (hgot > 0) {
#ifdef PT_ROLES_STATS
                            st_help++;
#endif
                            role_shade_pass<SLOT_OFF, SQ_OFF, TQ_OFF, FQ_OFF, CTL>(P, lane, hgot, hpos);
                            idle_polls = 0;
                            continue;
                        }
                    }
                }
                // done when every shader wave has found the global queue dry and no ray is left
                if (__atomic_load_n(&LDSI(CTL + RC_DRY), __ATOMIC_RELAXED) >= NS && __atomic_load_n(&LDSI(CTL + RC_ALIVE), __ATOMIC_RELAXED) == 0) break;
                if (++idle_polls > PT_ROLE_SPIN_MAX) { atomicOr(&LDSI(CTL + RC_ERROR), 2); break; }
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            idle_polls = 0;
            // 3. walk; leave when `batch` lanes have finished, or when empty lanes see new ready segments
            const int n_dead = 64 - n_live;
            if (walking) {
                const bool done = trav_run_wide<false, true, false, false>(ts, P.sc, o, d, cull, stk, tc, n_dead, P.batch,
                                                                          got < n_empty ? CTL + RC_TQ_TAIL : -1, seen_tail);
                if (done) walking = false;
            }
        }
#ifdef PT_ROLES_STATS
        if (lane == 0) {
            atomicAdd(&P.counters[6], (unsigned long long)st_iters);
            atomicAdd(&P.counters[7], (unsigned long long)st_idle);
            atomicAdd(&P.counters[8], (unsigned long long)st_live);
            atomicAdd(&P.counters[14], (unsigned long long)st_help);
        }
#endif
    } else {
        // ------------------------------------------------------------------ shader wave
#ifdef PT_ROLE_SHADER_PRIO
        __builtin_amdgcn_s_setprio(PT_ROLE_SHADER_PRIO);   // the block's one shader wave is what the tracers wait for
#endif
        const uint32_t slots_per_sample = (uint32_t)P.n_tiles * 64u;
        const uint32_t total = slots_per_sample * (P.samples ? P.spp : 1u);
        const uint32_t chunk = (uint32_t)P.chunk;
        const uint32_t shard_chunks = ((total + chunk - 1) / chunk + PT_SHARDS - 1) / PT_SHARDS;
        uint32_t chunk_next = 0, chunk_end = 0;
        bool queue_empty = false, dry_flagged = false;
        int shard = (int)(blockIdx.x & (PT_SHARDS - 1));
        int idle_polls = 0;
#ifdef PT_ROLES_STATS
        uint32_t ss_pass = 0, ss_got = 0, ss_idle = 0, ss_bpass = 0, ss_bgot = 0;
#endif
        for (;;) {
            if (__atomic_load_n(&LDSI(CTL + RC_ERROR), __ATOMIC_RELAXED) != 0) break;
            // what to do next: a pass is worth its ~1 000 instructions only at high lane use, so wait for a
            // wave's worth of finished segments (or of free slots for new paths) unless the tracers are
            // about to run dry
            int sq_n = 0, tq_n = 0, fq_n = 0;
            if (lane == 0) {
                sq_n = __atomic_load_n(&LDSI(CTL + RC_SQ_TAIL), __ATOMIC_RELAXED) - __atomic_load_n(&LDSI(CTL + RC_SQ_HEAD), __ATOMIC_RELAXED);
                tq_n = __atomic_load_n(&LDSI(CTL + RC_TQ_TAIL), __ATOMIC_RELAXED) - __atomic_load_n(&LDSI(CTL + RC_TQ_HEAD), __ATOMIC_RELAXED);
                fq_n = __atomic_load_n(&LDSI(CTL + RC_FQ_TAIL), __ATOMIC_RELAXED) - __atomic_load_n(&LDSI(CTL + RC_FQ_HEAD), __ATOMIC_RELAXED);
            }
            sq_n = __builtin_amdgcn_readfirstlane(sq_n); tq_n = __builtin_amdgcn_readfirstlane(tq_n); fq_n = __builtin_amdgcn_readfirstlane(fq_n);
            const bool can_begin = !queue_empty && fq_n > 0;
            const bool hungry = tq_n < PT_ROLE_T_LOW;    // the ready queue is nearly empty
            const bool do_shade = sq_n >= PT_ROLE_S_MIN || (sq_n > 0 && hungry && !(can_begin && fq_n >= PT_ROLE_B_MIN));
            const bool do_begin = !do_shade && can_begin && (fq_n >= PT_ROLE_B_MIN || hungry);
            // A. finished segments: shade them, 64 at a time
            int pos = 0;
            const int got = do_shade ? role_reserve(CTL + RC_SQ_HEAD, CTL + RC_SQ_TAIL, 64, lane, pos) : 0;
            if (got > 0) {
                idle_polls = 0;
#ifdef PT_ROLES_STATS
                ss_pass++; ss_got += got;
#endif
                role_shade_pass<SLOT_OFF, SQ_OFF, TQ_OFF, FQ_OFF, CTL>(P, lane, got, pos);
                continue;
            }
            // B. start new paths into free slots
            if (do_begin) {
                if (chunk_next == chunk_end) {
                    for (int tries = 0; tries < PT_SHARDS; tries++) {
                        uint32_t k = 0;
                        if (lane == 0) k = atomicAdd(P.queue + shard * PT_SHARD_STRIDE, 1u);
                        k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
                        const uint32_t first = (k * PT_SHARDS + (uint32_t)shard) * chunk;
                        if (k < shard_chunks && first < total) {
                            chunk_next = first;
                            chunk_end = min(first + chunk, total);
                            break;
                        }
                        shard = (shard + 1) & (PT_SHARDS - 1);
                    }
                    if (chunk_next == chunk_end) queue_empty = true;
                }
                if (!queue_empty) {
                    int fpos = 0;
                    const int want = (int)min(64u, chunk_end - chunk_next);
                    const int n_new = role_reserve(CTL + RC_FQ_HEAD, CTL + RC_FQ_TAIL, want, lane, fpos);
                    if (n_new > 0) {
                        idle_polls = 0;
#ifdef PT_ROLES_STATS
                        ss_bpass++; ss_bgot += n_new;
#endif
                        if (lane == 0) atomicAdd(&LDSI(CTL + RC_ALIVE), n_new);   // before the slots become visible
                        int slot = -1;
                        bool ready = false;
                        if (lane < n_new) slot = role_take(FQ_OFF, fpos + lane, CTL + RC_ERROR);
                        if (slot >= 0) {
                            uint32_t q = chunk_next + (uint32_t)lane;
                            uint32_t s_first = 0;
                            if (P.samples) { s_first = q / slots_per_sample; q -= s_first * slots_per_sample; }
                            int tx, ty;
                            bool inside = false;
                            if (pt_tile_coords(P, (int)(q >> 6), tx, ty)) {
                                const int px = tx * PT_TILE + (int)(q & 7u), py = ty * PT_TILE + (int)((q >> 3) & 7u);
                                if (px < P.W && py < P.H) {
                                    inside = true;
                                    const uint32_t pix = (uint32_t)py * (uint32_t)P.W + (uint32_t)px;
                                    PathState ps;
                                    path_begin(P, px, py, (uint64_t)pix, P.frame + s_first, ps);
                                    const int b = SLOT_OFF + slot * PT_SLOT_DW;
                                    LDSF(b + 0) = ps.o.x; LDSF(b + 1) = ps.o.y; LDSF(b + 2) = ps.o.z;
                                    LDSF(b + 3) = ps.d.x; LDSF(b + 4) = ps.d.y; LDSF(b + 5) = ps.d.z;
                                    float4* cold = P.roles_state + ((size_t)blockIdx.x * F + (size_t)slot) * (PT_COLD_DW / 4);
                                    cold[0] = make_float4(ps.mask.x, ps.mask.y, ps.mask.z, ps.accu.x);
                                    cold[1] = make_float4(ps.accu.y, ps.accu.z, __uint_as_float(ps.rng.s0), __uint_as_float(ps.rng.s1));
                                    cold[2] = make_float4(__uint_as_float(ps.rng.n), __uint_as_float(0u), __uint_as_float(pix), __uint_as_float(s_first));
                                    ready = true;
                                }
                            }
                            (void)inside;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // cold state stored before the slot is handed on
                        role_push(TQ_OFF, CTL + RC_TQ_TAIL, ready, slot, lane);
                        const bool unused = slot >= 0 && !ready;             // slot outside the image / partition
                        role_push(FQ_OFF, CTL + RC_FQ_TAIL, unused, slot, lane);
                        const int n_un = __popcll(__ballot(unused));
                        if (lane == 0 && n_un) atomicSub(&LDSI(CTL + RC_ALIVE), n_un);
                        chunk_next += (uint32_t)n_new;
                        continue;
                    }
                }
            }
            if (queue_empty && !dry_flagged) {
                if (lane == 0) atomicAdd(&LDSI(CTL + RC_DRY), 1);
                dry_flagged = true;
            }
            // C. idle: done when the global queue is dry for every shader wave and no ray is left in the block
            if (__atomic_load_n(&LDSI(CTL + RC_DRY), __ATOMIC_RELAXED) >= NS && __atomic_load_n(&LDSI(CTL + RC_ALIVE), __ATOMIC_RELAXED) == 0) break;
            if (++idle_polls > PT_ROLE_SPIN_MAX) { atomicOr(&LDSI(CTL + RC_ERROR), 4); break; }
#ifdef PT_ROLES_STATS
            ss_idle++;
#endif
            __builtin_amdgcn_s_sleep(2);
        }
#ifdef PT_ROLES_STATS
        if (lane == 0) {
            atomicAdd(&P.counters[9], (unsigned long long)ss_pass);
            atomicAdd(&P.counters[10], (unsigned long long)ss_got);
            atomicAdd(&P.counters[11], (unsigned long long)ss_idle);
            atomicAdd(&P.counters[12], (unsigned long long)ss_bpass);
            atomicAdd(&P.counters[13], (unsigned long long)ss_bgot);
        }
#endif
    }
    // a wave that gave up tells the host (counters[15]); the launch then reports PT_ERR_DEVICE
    if (lane == 0) {
        const int e = __atomic_load_n(&LDSI(CTL + RC_ERROR), __ATOMIC_RELAXED);
        if (e) atomicOr(&P.counters[15], (unsigned long long)e);
    }
}
#undef LDSI
#undef LDSF


// pt_k_persist.hip — the persistent-waves frame kernel and the sample fold, with their launchers.
// One translation unit of libptmi.so (pt_ctx.h).
#include "pt_ctx.h"

// ---------------------------------------------------------------------------------------
// Persistent-waves variant (Aila-Laine "persistent threads", which the reference does NOT
// have: SURVEY.md F5).  grid = resident waves only; each wave pulls chunks of PT_CHUNK
// tile-ordered pixel slots from one global counter and keeps every lane busy:
//   A. refill  — idle lanes take the next slots of the wave's chunk; the lane→slot map is a
//                ballot + prefix-count (mbcnt) compaction of the idle mask
//   B. walk    — lanes with a ray in flight run the resumable closest-hit walk; the wave
//                leaves it as soon as `batch` lanes are waiting for service
//   C. shade   — lanes whose walk finished do spheres/shading/BRDF and either get their next
//                segment (back to B) or fold the sample into the accumulator and go idle
// A lane's ray no longer waits for the slowest ray of its 8x8 tile at every bounce.  Each
// pixel still sees exactly the arithmetic of k_trace_mega_bvh2 (RNG keyed by pixel, same
// walk), so the image is bit-identical; only the schedule differs.
#define PT_CHUNK 64
enum { PH_IDLE = 0, PH_TRAV = 1, PH_SHADE = 2 };

template <bool COUNT, int OCC, int LSTK, int ALG>
__global__ void __launch_bounds__(PT_BLOCK, OCC) k_trace_persist_bvh2(const KParams P) {
    float4* s_top = s_dyn;
    if (P.sph_tab >= 0 && threadIdx.x < 11 * PT_KSPHERES) {  // sphere attributes + centres for the shading code (see path_shade)
        PT_KARGS(K);
        const float v = ((const __attribute__((address_space(4))) float*)&K.ksph[0])[threadIdx.x];
        ((float*)s_dyn)[P.sph_tab + threadIdx.x] = v;
        if (threadIdx.x % 11 < 4) ((float*)s_dyn)[P.sph_tab + 88 + 4 * (threadIdx.x / 11) + threadIdx.x % 11] = v;
    }
    lds_load_top<PT_BLOCK>(P.sc, s_top);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    TravOverflow<LSTK> stk_ovf;
    TravStack<LSTK, PT_BLOCK> stk(__builtin_amdgcn_readfirstlane(16 * P.sc.n_top + (tid & ~63)), stk_ovf);
    const bool cull = P.cull != 0;
    const uint32_t slots_per_sample = (uint32_t)P.n_tiles * 64u;
    const uint32_t total = slots_per_sample * (P.samples ? P.spp : 1u);

    uint32_t chunk_next = 0, chunk_end = 0;  // wave-uniform
    bool queue_empty = false;                // wave-uniform
    int shard = (int)(blockIdx.x & (PT_SHARDS - 1));  // wave-uniform: the shard this wave draws from
    const uint32_t chunk = (uint32_t)P.chunk;  // slots per fetch: 64, or less when the launch is small
    const uint32_t shard_chunks = ((total + chunk - 1) / chunk + PT_SHARDS - 1) / PT_SHARDS;

    int phase = PH_IDLE;
    uint32_t pix = 0, s_idx = 0;
    PathState ps;
    TravState ts;
    ps.o = ps.d = ps.mask = ps.accu = V3(0.f, 0.f, 0.f);
    ps.depth = 0; ps.rng.s0 = ps.rng.s1 = ps.rng.n = 0;
    ts.idx = ts.idy = ts.idz = ts.oodx = ts.oody = ts.oodz = 0.f;
    ts.node = PT_SENTINEL; ts.leaf = 0; ts.sp = 0;
    ts.h.t = PT_F32_MAX; ts.h.tri = -1; ts.h.rec = 0;

    TravCount tc;
    tc.inner = tc.tris = tc.leaves = 0;
    tc.it_node = tc.act_node = tc.it_rec = tc.act_rec = 0;
    uint32_t n_rays = 0, n_hits = 0, n_paths = 0;
    uint32_t it_begin = 0, act_begin = 0, it_shade = 0, act_shade = 0, it_loop = 0;  // COUNT only

    for (;;) {
        if (COUNT) it_loop++;
        // ---- A. refill idle lanes (all 64 lanes are converged here)
        const unsigned long long idle = __ballot(phase == PH_IDLE);
        const int n_idle = __popcll(idle);
        // Refill in batches: starting a path (tile coordinates, accumulator read, RNG seed, camera
        // ray: ~200 instructions) for one or two lanes at a time costs the whole wave those
        // instructions at 2-3 % utilisation (measured: 0.37 ms of a 1.18 ms frame).
        const int n_busy = __popcll(__ballot(phase == PH_TRAV));
        if (!queue_empty && (n_idle >= P.refill || (n_idle > 0 && n_busy == 0))) {
            if (chunk_next == chunk_end) {
                // eight counters, one per group of blocks that share an XCD (blockIdx % 8 is the
                // group label of the dispatcher's round-robin; speed only, never correctness);
                // shard s owns chunks s, s+8, s+16, ... (interleaved: contiguous bands of the image
                // cost very different amounts); an empty shard is left for the next (work stealing).  A single counter serialises: 32 400 chunk fetches
                // on one L2 atomic unit take ~0.37 ms (~88 returning atomics per microsecond).
                for (int tries = 0; tries < PT_SHARDS; tries++) {
                    uint32_t k = 0;
                    if (lane == 0) k = atomicAdd(P.queue + shard * PT_SHARD_STRIDE, 1u);
                    k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
                    const uint32_t first = (k * PT_SHARDS + (uint32_t)shard) * chunk;  // interleaved chunks
                    if (k < shard_chunks && first < total) {
                        chunk_next = first;
                        chunk_end = min(first + chunk, total);
                        break;
                    }
                    shard = (shard + 1) & (PT_SHARDS - 1);
                }
                if (chunk_next == chunk_end) queue_empty = true;
            }
            const uint32_t avail = chunk_end - chunk_next;
            const uint32_t take = min((uint32_t)n_idle, avail);
            bool started = false;
            if (phase == PH_IDLE) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                if (rank < take) {
                    const uint32_t q = chunk_next + rank;
                    uint32_t s_first = 0;
                    int px = 0, py = 0;
                    // (sample, pixel) of the slot: pt_slot_pixel — [sample][tile][lane], or the samples of a pixel side by side
                    {
                        if (pt_slot_pixel(P, q, s_first, px, py)) {   // tracer.cu:358
                            pix = (uint32_t)py * (uint32_t)P.W + (uint32_t)px;
                            s_idx = s_first;
                            // camera ray, then walk (or straight to shading); px/py live only here
                            path_begin(P, px, py, (uint64_t)pix, P.frame + s_idx, ps);
                            if (P.depth == 0) {
                                phase = PH_SHADE;
                                ts.h.t = PT_F32_MAX; ts.h.tri = -1;
                            } else if (P.sc.has_bvh) {
                                trav_begin(ts, ps.o, ps.d, stk, ALG >= 2 ? P.sc.wide_root : 0);
                                phase = PH_TRAV;
                            } else {
                                ts.h.t = PT_F32_MAX; ts.h.tri = -1; ts.h.rec = 0;
                                phase = PH_SHADE;
                            }
                            started = true;
                        }
                    }
                }
            }
            chunk_next += take;
            if (COUNT) {
                const int nb = __popcll(__ballot(started));
                if (nb) { it_begin++; act_begin += nb; }
            }
        }

        // ---- B. closest-hit walk for the lanes with a segment in flight
        {
            const int n_dead = queue_empty ? __popcll(__ballot(phase == PH_IDLE)) : 0;
            if (phase == PH_TRAV) {
                const bool fin = (ALG == 4)   ? trav_run_wide_pend<COUNT, true>(ts, P.sc, ps.o, ps.d, cull, stk, tc, n_dead, P.batch, P.vote_node, P.vote_rec)
                                 : (ALG >= 2) ? trav_run_wide<COUNT, true, ALG == 3>(ts, P.sc, ps.o, ps.d, cull, stk, tc, n_dead, P.batch)
                                 : (ALG == 1) ? trav_run_unified<COUNT, true>(ts, P.sc, ps.o, ps.d, cull, stk, tc, n_dead, P.batch)
                                              : trav_run<COUNT, true, true>(ts, P.sc, ps.o, ps.d, cull, stk, tc, n_dead, P.batch, s_top);
                if (fin) phase = PH_SHADE;
            }
        }

        // ---- C. shade finished segments
        if (COUNT) {
            const int nb = __popcll(__ballot(phase == PH_SHADE));
            if (nb) { it_shade++; act_shade += nb; }
        }
        if (phase == PH_SHADE) {
            v3 col = V3(0.f, 0.f, 0.f);
            bool done;
            if (P.depth == 0) {
                done = true;
            } else {
                if (COUNT) { n_rays++; n_hits += (ts.h.tri != -1); }
                done = path_shade(P, ps, ts.h, col, P.sph_tab);
            }
            if (!done) {
                if (P.sc.has_bvh) {
                    trav_begin(ts, ps.o, ps.d, stk, ALG >= 2 ? P.sc.wide_root : 0);
                    phase = PH_TRAV;
                }  // else: stays PH_SHADE with the (miss) hit record, shaded again next round
            } else if (P.samples) {
                float* dst = pt_sample_ptr(P, (uint32_t)s_idx, (size_t)pix);
                pt_sst3(dst, col);   // read once, by the fold
                if (COUNT) n_paths++;
                phase = PH_IDLE;
            } else {
                // one sample per call (spp > 1 always comes with the sample buffer): fold it straight
                // into the accumulator; nothing of the pixel's running mean is carried through the walk
                float* acc = P.accum + 3 * (size_t)pix;
                float ax = 0.f, ay = 0.f, az = 0.f;
                if (P.sample_index != 1) { ax = acc[0]; ay = acc[1]; az = acc[2]; }
                pt_accumulate(ax, ay, az, col, P.sample_index);
                if (COUNT) n_paths++;
                acc[0] = ax; acc[1] = ay; acc[2] = az;
                if ((P.flags & PT_FLAG_WRITE_RGBA) && P.rgba) P.rgba[pix] = pt_pack_rgba(ax, ay, az);
                phase = PH_IDLE;
            }
        }

        if (queue_empty && !__ballot(phase != PH_IDLE)) break;
    }

    if (COUNT) {
        const uint32_t a = wave_sum_u32(n_rays), b = wave_sum_u32(tc.inner), c = wave_sum_u32(tc.tris);
        const uint32_t dd = wave_sum_u32(tc.leaves), e = wave_sum_u32(n_hits), f = wave_sum_u32(n_paths);
        // the walk books its iterations in whichever lane is first among those inside it: sum the lanes
        const uint32_t w_it_node = wave_sum_u32(tc.it_node), w_act_node = wave_sum_u32(tc.act_node);
        const uint32_t w_it_rec = wave_sum_u32(tc.it_rec), w_act_rec = wave_sum_u32(tc.act_rec);
        const uint32_t w_ovf = wave_sum_u32(stk.n_ovf);
        if (lane == 0) {
            atomicAdd(&P.counters[0], (unsigned long long)a);
            atomicAdd(&P.counters[1], (unsigned long long)b);
            atomicAdd(&P.counters[2], (unsigned long long)c);
            atomicAdd(&P.counters[3], (unsigned long long)dd);
            atomicAdd(&P.counters[4], (unsigned long long)e);
            atomicAdd(&P.counters[5], (unsigned long long)f);
            // schedule statistics, one contribution per wave (pt_get_wave_stats)
            atomicAdd(&P.counters[6], (unsigned long long)w_it_node);
            atomicAdd(&P.counters[7], (unsigned long long)w_act_node);
            atomicAdd(&P.counters[8], (unsigned long long)w_it_rec);
            atomicAdd(&P.counters[9], (unsigned long long)w_act_rec);
            atomicAdd(&P.counters[10], (unsigned long long)it_shade);
            atomicAdd(&P.counters[11], (unsigned long long)act_shade);
            atomicAdd(&P.counters[12], (unsigned long long)it_begin);
            atomicAdd(&P.counters[13], (unsigned long long)act_begin);
            atomicAdd(&P.counters[14], (unsigned long long)it_loop);
            atomicAdd(&P.counters[15], (unsigned long long)w_ovf);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Folds the spp sample colours of every owned pixel into the running mean, in sample order,
// with the reference's per-frame clamp (tracer.cu:386-391) and packs the display word
// (:394-398): exactly what spp consecutive single-sample launches do to the accumulator.
__global__ void __launch_bounds__(256) k_fold_samples(const KParams P) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    int tx, ty;
    if (!pt_tile_coords(P, tile, tx, ty)) return;
    const int px = tx * PT_TILE + (lane & 7), py = ty * PT_TILE + (lane >> 3);
    if (px >= P.W || py >= P.H) return;
    const size_t pix = (size_t)py * (size_t)P.W + (size_t)px;
    float* acc = P.accum + 3 * pix;
    float ax = 0.f, ay = 0.f, az = 0.f;
    if (P.sample_index != 1) { ax = acc[0]; ay = acc[1]; az = acc[2]; }
    for (uint32_t s = 0; s < P.spp; s++) {
        const float* c = pt_sample_ptr(P, s, pix);
        pt_accumulate(ax, ay, az, pt_sld3(c), P.sample_index + s);
    }
    acc[0] = ax; acc[1] = ay; acc[2] = az;
    if ((P.flags & PT_FLAG_WRITE_RGBA) && P.rgba) P.rgba[pix] = pt_pack_rgba(ax, ay, az);
}

// The same fold over the [pixel][sample][3] sample buffer of the sample groups (spp a multiple of 4): LP lanes per pixel, each
// reading spp / LP consecutive samples in 16-byte pieces (four samples = three pieces; adjacent lanes read adjacent bytes, so a
// wave covers whole 8-pixel runs of its tile end to end), the running mean handed from lane to lane so that the samples enter it
// in order 0 .. spp-1 exactly as above.  LP = 4 for 16 samples per call, 2 for 8, 1 for 4 (largest of 4 / 2 / 1 dividing spp / 4).
template <int LP>
__global__ void __launch_bounds__(256) k_fold_samples_grouped(const KParams P) {
    const int k = threadIdx.x / LP, part = threadIdx.x % LP;
    const int tile = blockIdx.x * (4 / LP) + (k >> 6);
    int tx = 0, ty = 0;
    const bool have_tile = pt_tile_coords(P, tile, tx, ty);   // the LP lanes of a pixel agree; nobody leaves before the last hand-over
    const int px = tx * PT_TILE + (k & 7), py = ty * PT_TILE + ((k >> 3) & 7);
    const bool in = have_tile && px < P.W && py < P.H;
    const size_t pix = in ? (size_t)py * (size_t)P.W + (size_t)px : 0;
    float* acc = P.accum + 3 * pix;
    float ax = 0.f, ay = 0.f, az = 0.f;
    if (in && P.sample_index != 1) { ax = acc[0]; ay = acc[1]; az = acc[2]; }
    const uint32_t per = P.spp / (uint32_t)LP, s0 = (uint32_t)part * per;   // a multiple of 4 samples
    const float4* c4 = (const float4*)pt_sample_ptr(P, s0, pix);
    // one group of four (the 16-, 8- and 4-sample calls): requested before the hand-over chain starts; longer shares stream theirs
    // inside their turn
    const bool pre = per == 4u;
    float4 qa = make_float4(0.f, 0.f, 0.f, 0.f), qb = qa, qd = qa;
    if (in && pre) { qa = pt_sld4(c4); qb = pt_sld4(c4 + 1); qd = pt_sld4(c4 + 2); }
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int turn = 0; turn < LP; turn++) {
        if (in && part == turn) {
            for (uint32_t s = 0; s < per; s += 4, c4 += 3) {
                if (!pre) { qa = pt_sld4(c4); qb = pt_sld4(c4 + 1); qd = pt_sld4(c4 + 2); }
                pt_accumulate(ax, ay, az, V3(qa.x, qa.y, qa.z), P.sample_index + s0 + s);
                pt_accumulate(ax, ay, az, V3(qa.w, qb.x, qb.y), P.sample_index + s0 + s + 1);
                pt_accumulate(ax, ay, az, V3(qb.z, qb.w, qd.x), P.sample_index + s0 + s + 2);
                pt_accumulate(ax, ay, az, V3(qd.y, qd.z, qd.w), P.sample_index + s0 + s + 3);
            }
        }
        if (LP > 1) {   // the share that just ran hands the mean on
            const int src = (lane & ~(LP - 1)) | turn;
            ax = __shfl(ax, src); ay = __shfl(ay, src); az = __shfl(az, src);
        }
    }
    if (!in || part != 0) return;
    acc[0] = ax; acc[1] = ay; acc[2] = az;
    if ((P.flags & PT_FLAG_WRITE_RGBA) && P.rgba) P.rgba[pix] = pt_pack_rgba(ax, ay, az);
}

namespace ptmi {

// Instantiated budgets (waves/SIMD, LDS stack window): the wide walks (2, 3 = Woop records, 4 = postponed
// leaf) get (4,16) (6,16) (6,24) (8,16) (4,72); the two exact binary walks (parity variants) (8,16) and
// (4,72).  Other requests run the nearest one: they are speed knobs, never results.
hipError_t launch_persist(const LaunchCfg& L, const KParams& P, hipStream_t st) {
#define PT_GO(COUNT, OCC, LSTK, ALG)                                                                               \
    do {                                                                                                           \
        int per_cu = 0;                                                                                            \
        hipError_t e_ = allow_lds(k_trace_persist_bvh2<COUNT, OCC, LSTK, ALG>, L.lds);                             \
        if (e_ != hipSuccess) return e_;                                                                           \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_persist_bvh2<COUNT, OCC, LSTK, ALG>,    \
                                                         PT_BLOCK, L.lds) != hipSuccess || per_cu < 1)            \
            per_cu = 1;                                                                                            \
        /* grid = resident blocks only (no grid-wide wait anywhere, so an over-estimate only means a few late */  \
        /* blocks find the queue empty and exit) */                                                                \
        hipLaunchKernelGGL((k_trace_persist_bvh2<COUNT, OCC, LSTK, ALG>),                                          \
                           dim3(std::min(per_cu * L.n_cu, std::max(1, L.work_blocks))), dim3(PT_BLOCK), L.lds, st, P); \
        return hipGetLastError();                                                                                  \
    } while (0)
#define PT_GO_WIDE(COUNT, OCC, LSTK)                       \
    do {                                                   \
        if (L.walk == 4) PT_GO(COUNT, OCC, LSTK, 4);       \
        else if (L.walk == 3) PT_GO(COUNT, OCC, LSTK, 3);  \
        else PT_GO(COUNT, OCC, LSTK, 2);                   \
    } while (0)
#define PT_GO_CFG(COUNT)                                                               \
    do {                                                                               \
        if (L.walk <= 1) {                                                             \
            if (L.lstk >= PT_STACK_CAP) { if (L.walk == 1) PT_GO(COUNT, 4, PT_STACK_CAP, 1); else PT_GO(COUNT, 4, PT_STACK_CAP, 0); } \
            else { if (L.walk == 1) PT_GO(COUNT, 8, 16, 1); else PT_GO(COUNT, 8, 16, 0); } \
        } else if (L.lstk >= PT_STACK_CAP) PT_GO_WIDE(COUNT, 4, PT_STACK_CAP);         \
        else if (L.lstk == 24) PT_GO_WIDE(COUNT, 6, 24);                               \
        else if (L.occ >= 8) PT_GO_WIDE(COUNT, 8, 16);                                 \
        else if (L.occ >= 5) PT_GO_WIDE(COUNT, 6, 16);                                 \
        else PT_GO_WIDE(COUNT, 4, 16);                                                 \
    } while (0)
    if (L.count) PT_GO_CFG(true);
    else PT_GO_CFG(false);
#undef PT_GO_CFG
#undef PT_GO_WIDE
#undef PT_GO
}

hipError_t launch_fold(const KParams& P, hipStream_t st) {
    if (P.smp_ps > 1u) {   // sample groups: spp is a multiple of 4
        const uint32_t q = P.spp / 4u;
        if ((q & 3u) == 0u) hipLaunchKernelGGL(k_fold_samples_grouped<4>, dim3(P.n_tiles), dim3(256), 0, st, P);
        else if ((q & 1u) == 0u) hipLaunchKernelGGL(k_fold_samples_grouped<2>, dim3((P.n_tiles + 1) / 2), dim3(256), 0, st, P);
        else hipLaunchKernelGGL(k_fold_samples_grouped<1>, dim3((P.n_tiles + 3) / 4), dim3(256), 0, st, P);
    } else hipLaunchKernelGGL(k_fold_samples, dim3((P.n_tiles + 3) / 4), dim3(256), 0, st, P);
    return hipGetLastError();
}

}  // namespace ptmi

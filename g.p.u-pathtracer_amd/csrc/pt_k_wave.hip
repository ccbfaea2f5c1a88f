// pt_k_wave.hip — the stage-split ("wavefront") frame pipeline, PT_KERNEL_WAVEFRONT.
// One translation unit of libptmi.so (pt_ctx.h).
//
// BASELINE.json configs[4] / SURVEY.md §7 step 5e: "ray buffers in SoA, extend / shade / generate
// kernels, ballot + prefix-sum compaction of live rays between bounces".  The reference has nothing
// like it (one thread per pixel, tracer.cu:413-414); the arithmetic of a path is the reference's
// (path_begin / trav_run_wide / path_shade_hit are the functions the other frame kernels call), so the
// images are the same bit for bit — only WHERE a path's state lives between segments differs:
//
//   bounce 0   works on SLOTS, one per (sample, pixel) in the tile order of the other frame kernels: both stages
//              compute the camera ray from the slot number (a few dozen instructions) instead of a generate
//              kernel writing 32-byte ray records for them to read back; the shade stage of bounce 0 WRITES
//              the sample colour (accu = 0 + emission) instead of adding to a zeroed one
//   per bounce
//     extend   persistent waves: a wave draws REGIONS of the ray queue from eight sharded counters,
//              walks the BVH for 64 rays at a time and refills a lane as soon as its ray is done (its whole
//              per-lane state is one ray + the walk: no path state, 8 waves per SIMD); writes (t, record)
//     shade    one lane per live record, every wave full: spheres, shading, BRDF sample (tracer.cu:98-296);
//              emitted light is added to the sample colour in place, a miss writes the background
//              (tracer.cu:140-142), survivors are packed to the front of their region's next generation with
//              a ballot + prefix count (v_mbcnt) — dead paths cost nothing in the next stage
//   k_fold_samples folds the sample colours into the running mean (tracer.cu:386-391), as for every kernel.
//
// The path records stream through HBM (read once, written once per stage, 16-byte pieces, coalesced): this
// is the part of the path loop that is bandwidth work, and the only per-ray state the latency-bound BVH walk
// still carries is the ray itself.  Why it pays: in the persistent kernel a finished lane idles through other
// lanes' node steps until 36 of 64 lanes want shading (node steps ran at 55 % lane use, shading passes of
// ~700 instructions at 56 %, path starts at 20 %); here every stage runs one kind of work.
#include "pt_ctx.h"

// ------------------------------------------------------------------------------------------------
// block-level compaction: exclusive rank of the calling lane among the block's lanes with keep == true,
// and the block's total.  Four waves: ballot + v_mbcnt inside a wave, one LDS word per wave.
__device__ __forceinline__ int wf_block_rank(bool keep, int& total, int* s_cnt) {
    const unsigned long long m = __ballot(keep);
    const int w = threadIdx.x >> 6;
    const int in_wave = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if ((threadIdx.x & 63) == 0) s_cnt[w] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < PT_BLOCK / 64; k++) {
        const int c = s_cnt[k];
        if (k < w) base += c;
        tot += c;
    }
    total = tot;
    return base + in_wave;
}

// copies the first PT_KSPHERES spheres' attributes from the kernel arguments into LDS (path_shade's sph_tab = 0)
__device__ __forceinline__ void wf_sphere_table() {
    if (threadIdx.x < 11 * PT_KSPHERES) {
        PT_KARGS(K);
        const float v = ((const __attribute__((address_space(4))) float*)&K.ksph[0])[threadIdx.x];
        ((float*)s_dyn)[threadIdx.x] = v;
        if (threadIdx.x % 11 < 4) ((float*)s_dyn)[88 + 4 * (threadIdx.x / 11) + threadIdx.x % 11] = v;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// once per call: the frame hashes of the call's samples (uf::hash, utilfun.cpp:380-389) and fresh queue
// counters for every bounce's extend launch
__global__ void __launch_bounds__(256) k_wf_prepare(const KParams P) {
    for (uint32_t i = threadIdx.x; i < P.spp; i += 256) P.wf.hashes[i] = pt_wang64(P.frame + i);
    for (uint32_t i = threadIdx.x; i < P.wf.queues_words; i += 256) P.wf.queues_all[i] = 0u;
}

// ------------------------------------------------------------------------------------------------
// slot = region * 256 + thread; slot >> 6 = a wave's worth of bounce-0 paths, coherent by construction (pt_slot_pixel)
__device__ __forceinline__ bool wf_slot_pixel(const KParams& P, uint32_t slot, uint32_t& s_idx, int& px, int& py) {
    if (slot >= P.wf.n_slots) return false;
    return pt_slot_pixel(P, slot, s_idx, px, py);
}

// ------------------------------------------------------------------------------------------------
// extend: the closest-hit walk (rows a5-a7) over the ray queue; see the file header.
// FIRST: bounce 0 — a region is 256 consecutive slots and the ray is the slot's camera ray.
template <bool COUNT, int OCC, int LSTK, bool FIRST>
__global__ void __launch_bounds__(PT_BLOCK, OCC) k_wf_extend(const KParams P) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    TravOverflow<LSTK> stk_ovf;
    TravStack<LSTK, PT_BLOCK> stk(__builtin_amdgcn_readfirstlane(tid & ~63), stk_ovf);
    const bool cull = P.cull != 0;
    const uint32_t n_regions = (uint32_t)P.wf.n_regions;
    const uint32_t shard_regions = (n_regions + PT_SHARDS - 1) / PT_SHARDS;
    const int batch = P.batch;

    uint32_t next = 0, end = 0;   // wave-uniform: the part of the wave's region not handed to lanes yet
    bool empty = false;           // wave-uniform: every shard of the queue is exhausted
    int shard = (int)(blockIdx.x & (PT_SHARDS - 1));

    bool live = false;
    uint32_t idx = 0;
    v3 o = V3(0.f, 0.f, 0.f), d = V3(0.f, 0.f, 0.f);
    TravState ts;
    ts.idx = ts.idy = ts.idz = ts.oodx = ts.oody = ts.oodz = 0.f;
    ts.node = PT_SENTINEL; ts.leaf = 0; ts.sp = 0;
    ts.h.t = PT_F32_MAX; ts.h.tri = -1; ts.h.rec = 0;
    TravCount tc;
    tc.inner = tc.tris = tc.leaves = 0;
    tc.it_node = tc.act_node = tc.it_rec = tc.act_rec = 0;
    uint32_t n_rays = 0, it_begin = 0, act_begin = 0, it_loop = 0;

    for (;;) {
        if (COUNT) it_loop++;
        // ---- refill: idle lanes take the next records of the wave's region(s); lane -> record by a ballot +
        // prefix count (v_mbcnt) of the idle mask
        const unsigned long long idle = __ballot(!live);
        const int n_idle = __popcll(idle);
        if (!empty && (n_idle >= batch || n_idle == 64)) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            uint32_t served = 0;
            for (int round = 0; round < 8 && served < (uint32_t)n_idle; round++) {
                if (next == end) {
                    // eight counters (blocks b and b + 8 share an XCD under round-robin placement: speed only);
                    // shard s owns regions s, s + 8, ...; an empty shard is left for the next one
                    bool got = false;
                    for (int tries = 0; tries < PT_SHARDS && !got; tries++) {
                        uint32_t k = 0;
                        if (lane == 0) k = atomicAdd(P.wf.queue + shard * PT_SHARD_STRIDE, 1u);
                        k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
                        const uint32_t r = k * PT_SHARDS + (uint32_t)shard;
                        if (k < shard_regions && r < n_regions) {
                            next = r * PT_REGION;
                            if (FIRST) end = min(next + (uint32_t)PT_REGION, P.wf.n_slots);
                            else end = next + (uint32_t)__builtin_amdgcn_readfirstlane(P.wf.cnt_in[r]);
                            got = true;
                        } else {
                            shard = (shard + 1) & (PT_SHARDS - 1);
                        }
                    }
                    if (!got) { empty = true; break; }
                }
                const uint32_t take = min((uint32_t)n_idle - served, end - next);
                if (!live && rank >= served && rank < served + take) {
                    idx = next + (rank - served);
                    if (FIRST) {
                        uint32_t s_idx = 0;
                        int px = 0, py = 0;
                        if (wf_slot_pixel(P, idx, s_idx, px, py)) {
                            PathState ps;
                            path_begin_hashed(P, px, py, (uint64_t)((uint32_t)py * (uint32_t)P.W + (uint32_t)px), P.wf.hashes[s_idx], ps);
                            o = ps.o;
                            d = ps.d;
                            live = true;
                        }
                    } else {
                        const float4 a = pt_sld4(P.wf.ray0_in + idx), b = pt_sld4(P.wf.ray1_in + idx);
                        o = V3(a.x, a.y, a.z);
                        d = V3(a.w, b.x, b.y);
                        live = true;
                    }
                    if (live) trav_begin(ts, o, d, stk, P.sc.wide_root);
                }
                next += take;
                served += take;
            }
            if (COUNT && served) { it_begin++; act_begin += served; }
        }
        const unsigned long long busy = __ballot(live);
        if (!busy) {
            if (empty) break;
            continue;
        }
        // ---- walk until `batch` lanes have finished (lanes that can get no more work do not count)
        const int n_dead = empty ? 64 - __popcll(busy) : 0;
        if (live) {
            const bool fin = trav_run_wide<COUNT, true, false, true>(ts, P.sc, o, d, cull, stk, tc, n_dead, batch);
            if (fin) {
                pt_sst2(P.wf.hit + idx, make_float2(ts.h.t, __int_as_float(ts.h.rec)));
                live = false;
                if (COUNT) n_rays++;
            }
        }
    }

    if (COUNT) {
        const uint32_t a = wave_sum_u32(n_rays), b = wave_sum_u32(tc.inner), c = wave_sum_u32(tc.tris), dd = wave_sum_u32(tc.leaves);
        const uint32_t w_it_node = wave_sum_u32(tc.it_node), w_act_node = wave_sum_u32(tc.act_node);
        const uint32_t w_it_rec = wave_sum_u32(tc.it_rec), w_act_rec = wave_sum_u32(tc.act_rec);
        const uint32_t w_ovf = wave_sum_u32(stk.n_ovf);
        if (lane == 0) {
            atomicAdd(&P.counters[0], (unsigned long long)a);
            atomicAdd(&P.counters[1], (unsigned long long)b);
            atomicAdd(&P.counters[2], (unsigned long long)c);
            atomicAdd(&P.counters[3], (unsigned long long)dd);
            atomicAdd(&P.counters[6], (unsigned long long)w_it_node);
            atomicAdd(&P.counters[7], (unsigned long long)w_act_node);
            atomicAdd(&P.counters[8], (unsigned long long)w_it_rec);
            atomicAdd(&P.counters[9], (unsigned long long)w_act_rec);
            atomicAdd(&P.counters[12], (unsigned long long)it_begin);
            atomicAdd(&P.counters[13], (unsigned long long)act_begin);
            atomicAdd(&P.counters[14], (unsigned long long)it_loop);
            atomicAdd(&P.counters[15], (unsigned long long)w_ovf);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// shade: one bounce of tracer.cu:98-296 for every live record of a region; see the file header.
// FIRST: bounce 0 — lane = slot; the path starts here (camera ray, RNG) and the sample colour is written, not added to.
// LAST: the path's final bounce (without PT_FLAG_NEE): only the hit's emission is still wanted (path_last_emission).
// where a surviving path's record goes inside its region: packed in slot order (experiments splice their own order in here:
// tools/pt_exp_hooks.h, PT_EXP_SORT)
#ifndef PT_SURVIVOR_RANK
#define PT_SURVIVOR_RANK(P, alive, ps, total, s_cnt) wf_block_rank(alive, total, s_cnt)
#endif

template <bool COUNT, bool NEE, bool FIRST, bool LAST = false>
__global__ void __launch_bounds__(PT_BLOCK) k_wf_shade(const KParams P) {
    __shared__ int s_cnt[PT_BLOCK / 64];
    __shared__ int s_cnt2[PT_BLOCK / 64];
    const uint32_t region = blockIdx.x;
    const int n_in = FIRST ? PT_REGION : P.wf.cnt_in[region];
    const bool last = LAST || P.wf.bounce + 1 >= P.depth;
    if (n_in == 0) {   // (the whole block: before anything is set up — in an open scene most regions are empty after the first bounce)
        if (!last && threadIdx.x == 0) P.wf.cnt_out[region] = 0;
        if (NEE && threadIdx.x == 0) P.wf.s_cnt[region] = 0;
        return;
    }
    wf_sphere_table();
    const size_t i = (size_t)region * PT_REGION + threadIdx.x;
    bool have = (int)threadIdx.x < n_in;
    bool alive = false, tri_hit = false;
    PathState ps;
    NeeReq req;
    req.want = false;
    uint32_t pix = 0, s_idx = 0;
    int px = 0, py = 0;
    if (FIRST) have = wf_slot_pixel(P, (uint32_t)i, s_idx, px, py);
    if (have) {
        if (FIRST) {
            pix = (uint32_t)py * (uint32_t)P.W + (uint32_t)px;
            path_begin_hashed(P, px, py, (uint64_t)pix, P.wf.hashes[s_idx], ps);
        } else {
            const float4 a = pt_sld4(P.wf.ray0_in + i), b = pt_sld4(P.wf.ray1_in + i);
            ps.o = V3(a.x, a.y, a.z);
            ps.d = V3(a.w, b.x, b.y);
            pix = __float_as_uint(b.z);
            ps.nee_mask = 0;
            uint32_t sn = __float_as_uint(b.w);
            if (NEE) {   // the sphere bits ride above the pixel, the triangle-light bit above the sample number
                ps.nee_mask = (pix >> 24) | ((sn >> 31) << 8);
                pix &= 0xffffffu;
                sn &= 0x7fffffffu;
            }
            s_idx = sn >> 12;
            ps.mask = V3(pt_sld1(P.wf.mask_in + i), pt_sld1(P.wf.mask_in + (size_t)P.wf.cap + i), pt_sld1(P.wf.mask_in + 2 * (size_t)P.wf.cap + i));
            ps.accu = V3(0.f, 0.f, 0.f);   // this segment's emission only: the running sum lives in the sample buffer
            ps.depth = P.wf.bounce;
            ps.rng = pt_rng_init(P.wf.hashes[s_idx], (uint64_t)pix);
            ps.rng.n = sn & 0xfffu;
        }
        const float2 hh = pt_sld2(P.wf.hit + i);
        Hit h;
        h.t = hh.x;
        h.rec = __float_as_int(hh.y);
        h.tri = -1;
        v3 tri_n = V3(0.f, 0.f, 0.f);
        tri_hit = h.t < PT_F32_MAX;
        if (tri_hit) {   // a triangle was hit: its id (v0.w) and un-normalised normal (4th piece)
            if (!LAST) {
                const float4 q3 = P.sc.nodes[h.rec + 3];
                tri_n = V3(q3.x, q3.y, q3.z);
            }
            h.tri = 0;
            if (P.tri_matid) h.tri = __float_as_int(P.sc.nodes[h.rec].w);
        }
        float* smp = pt_sample_ptr(P, s_idx, (size_t)pix);
        const SceneHit sh = pt_closest_sphere(P, ps.o, ps.d, h, 0);
        if (sh.geom == 3) {   // tracer.cu:140-142: the sample IS the background colour, whatever was gathered before
            PT_KARGS(K);
            if (FIRST && (K.flags & PT_FLAG_MISS_KEEPS_PATH)) {
                smp[0] = 0.f + ps.mask.x * K.bk[0]; smp[1] = 0.f + ps.mask.y * K.bk[1]; smp[2] = 0.f + ps.mask.z * K.bk[2];
            } else if (K.flags & PT_FLAG_MISS_KEEPS_PATH) {   // extension: accu (= the sample buffer) + mask * bk
                smp[0] += ps.mask.x * K.bk[0]; smp[1] += ps.mask.y * K.bk[1]; smp[2] += ps.mask.z * K.bk[2];
            } else {
                smp[0] = K.bk[0]; smp[1] = K.bk[1]; smp[2] = K.bk[2];
            }
        } else {
            v3 col;
            const bool done = LAST ? true : path_shade_hit(P, ps, h, sh, tri_n, col, 0, NEE ? &req : nullptr);
            if (LAST) col = path_last_emission(P, ps, h, sh, 0);
            const v3 e = done ? col : ps.accu;   // mask * emission of this hit (accu entered as 0)
            if (FIRST) {   // accu = 0 (tracer.cu:48) + this hit's emission
                pt_sst3(smp, V3(0.f + e.x, 0.f + e.y, 0.f + e.z));
            } else if (!(e.x == 0.f) || !(e.y == 0.f) || !(e.z == 0.f)) {
                smp[0] += e.x; smp[1] += e.y; smp[2] += e.z;   // (as ONE dwordx3 each way: the last bounce's launch +6 %)
            }
            alive = !done;
        }
    }
    if (COUNT) {   // all 64 lanes of every wave are here
        const uint32_t nh = wave_sum_u32(tri_hit ? 1u : 0u), np = wave_sum_u32(have && !alive ? 1u : 0u);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&P.counters[4], (unsigned long long)nh);
            atomicAdd(&P.counters[5], (unsigned long long)np);
        }
    }
    if (NEE) {   // PT_FLAG_NEE: the shadow rays of this bounce's DIFF hits, packed like the survivors
        int n_sh;
        const int rs = wf_block_rank(req.want, n_sh, s_cnt2);
        if (req.want) {
            const size_t j = (size_t)region * PT_REGION + (size_t)rs;
            P.wf.s_ray0[j] = make_float4(req.o.x, req.o.y, req.o.z, req.d.x);
            P.wf.s_ray1[j] = make_float4(req.d.y, req.d.z, __uint_as_float(pix), __uint_as_float(s_idx));
            P.wf.s_con[j] = make_float4(req.contrib.x, req.contrib.y, req.contrib.z, req.t_max);
        }
        if (threadIdx.x == 0) P.wf.s_cnt[region] = n_sh;
    }
    if (last) return;   // every path ends with this bounce (tracer.cu:305)
    int total;
    const int r = PT_SURVIVOR_RANK(P, alive, ps, total, s_cnt);
    if (alive) {
        const size_t j = (size_t)region * PT_REGION + (size_t)r;
        pt_sst4(P.wf.ray0_out + j, make_float4(ps.o.x, ps.o.y, ps.o.z, ps.d.x));
        pt_sst4(P.wf.ray1_out + j, make_float4(ps.d.y, ps.d.z, __uint_as_float(NEE ? (pix | ((ps.nee_mask & 0xffu) << 24)) : pix),
                                              __uint_as_float((s_idx << 12) | ps.rng.n | (NEE ? (ps.nee_mask >> 8) << 31 : 0u))));
        pt_sst1(P.wf.mask_out + j, ps.mask.x);
        pt_sst1(P.wf.mask_out + (size_t)P.wf.cap + j, ps.mask.y);
        pt_sst1(P.wf.mask_out + 2 * (size_t)P.wf.cap + j, ps.mask.z);
    }
    if (threadIdx.x == 0) P.wf.cnt_out[region] = total;
}

// PT_FLAG_NEE: adds the contribution of every shadow ray that reached its light (nothing closer than t_max) to its path's
// sample colour — after the emission this bounce's shade launch added, as the oracle orders the two sums.
__global__ void __launch_bounds__(PT_BLOCK) k_wf_resolve(const KParams P) {
    const int n_in = P.wf.s_cnt[blockIdx.x];
    if ((int)threadIdx.x >= n_in) return;
    const size_t i = (size_t)blockIdx.x * PT_REGION + threadIdx.x;
    const float4 c = P.wf.s_con[i];
    const float2 hh = P.wf.s_hit[i];
    if (hh.x < c.w) return;   // a triangle is in the way
    const float4 b = P.wf.s_ray1[i];
    const uint32_t pix = __float_as_uint(b.z), s_idx = __float_as_uint(b.w);
    float* smp = pt_sample_ptr(P, s_idx, (size_t)pix);
    smp[0] += c.x; smp[1] += c.y; smp[2] += c.z;
}

namespace ptmi {

// sizes of one call's path records: [ray0 x2][ray1 x2][mask x2 (3 planes)][hit][cnt x2][hashes][queues][shadow records]
struct WaveLayout {
    size_t n_regions, cap, b_ray, b_mask, b_hit, b_cnt, b_hash, q_words, b_q, b_nee, need;
    bool nee;
};

// most RNG draws one bounce can make with these flags (path_shade_hit): DIFF 4, or 2 cosine-weighted (+ 3 for the light
// sample of PT_FLAG_NEE), METAL 2, REFR 1, + 1 per roulette switch
static uint32_t draws_per_bounce(uint32_t flags) {
    uint32_t k = (flags & PT_FLAG_COSINE_DIFF) ? 2u : 4u;
    if (flags & PT_FLAG_NEE) k += 3u;
    if (flags & PT_FLAG_RUSSIAN_ROULETTE) k += 1u;
    if (flags & PT_FLAG_RR_CPU_TRACER) k += 1u;
    return k;
}

static int wave_layout(pt_ctx* c, const KParams& P, int work_tiles, WaveLayout& w) {
    // limits of the record's packed fields: 12 bits of RNG draw count (2 camera draws + draws_per_bounce per bounce), 20 bits of sample
    if ((uint64_t)P.depth * draws_per_bounce(P.flags) + 2u >= 4096u || P.spp >= (1u << 20))
        return fail(c, PT_ERR_UNSUPPORTED, "pt_render: PT_KERNEL_WAVEFRONT packs < 4096 RNG draws per path (2 + depth x 4..9, by flags) and < 2^20 samples per call into a path record");
    w.n_regions = ((size_t)work_tiles + PT_REGION / 64 - 1) / (PT_REGION / 64);
    w.cap = w.n_regions * PT_REGION;
    if (w.cap >= (1ull << 31)) return fail(c, PT_ERR_INVALID, "pt_render: too many path records for one call");
    w.b_ray = w.cap * 16; w.b_mask = w.cap * 12; w.b_hit = w.cap * 8; w.b_cnt = ((w.n_regions * 4 + 255) / 256) * 256;
    w.b_hash = (((size_t)P.spp * 8 + 255) / 256) * 256;
    w.nee = (P.flags & PT_FLAG_NEE) != 0;
    if (w.nee && P.spp >= (1u << 19)) return fail(c, PT_ERR_UNSUPPORTED, "pt_render: PT_FLAG_NEE in the stage-split pipeline packs < 2^19 samples per call into a path record");
    w.q_words = (size_t)P.depth * (w.nee ? 2 : 1) * PT_SHARDS * PT_SHARD_STRIDE;   // one set of queue counters per extend launch
    w.b_q = w.q_words * 4;
    w.b_nee = w.nee ? 3 * w.b_ray + w.b_hit + w.b_cnt : 0;   // shadow records: s_ray0, s_ray1, s_con, s_hit, s_cnt
    w.need = 4 * w.b_ray + 2 * w.b_mask + w.b_hit + 2 * w.b_cnt + w.b_hash + w.b_q + w.b_nee;
    return PT_OK;
}

// makes sure the context holds path records for this call (PT_KERNEL_AUTO calls it ahead of the timed span of its trial)
int wave_reserve(pt_ctx* c, const KParams& P, int work_tiles) {
    WaveLayout w;
    const int rc = wave_layout(c, P, work_tiles, w);
    if (rc != PT_OK) return rc;
    if (w.need > c->wave_bytes) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_wave);
        c->d_wave = nullptr;
        c->wave_bytes = 0;
        HIP_TRY(c, hipMalloc(&c->d_wave, w.need));
        c->wave_bytes = w.need;
    }
    return PT_OK;
}

int render_wavefront(pt_ctx* c, KParams& P, const LaunchCfg& L, int work_tiles) {
    WaveLayout w;
    {
        const int rc = wave_layout(c, P, work_tiles, w);
        if (rc != PT_OK) return rc;
        const int rc2 = wave_reserve(c, P, work_tiles);
        if (rc2 != PT_OK) return rc2;
    }
    const size_t n_regions = w.n_regions, cap = w.cap, b_ray = w.b_ray, b_mask = w.b_mask, b_hit = w.b_hit, b_cnt = w.b_cnt, b_hash = w.b_hash;
    const size_t q_words = w.q_words, b_q = w.b_q;
    const bool nee = w.nee;
    char* base = (char*)c->d_wave;
    float4* ray0[2] = {(float4*)base, (float4*)(base + b_ray)};
    float4* ray1[2] = {(float4*)(base + 2 * b_ray), (float4*)(base + 3 * b_ray)};
    base += 4 * b_ray;
    float* mask[2] = {(float*)base, (float*)(base + b_mask)};
    base += 2 * b_mask;
    float2* hit = (float2*)base;
    base += b_hit;
    int* cnt[2] = {(int*)base, (int*)(base + b_cnt)};
    base += 2 * b_cnt;
    unsigned long long* hashes = (unsigned long long*)base;
    base += b_hash;
    unsigned int* queues = (unsigned int*)base;
    base += b_q;
    P.wf.nee = nee ? 1 : 0;
    if (nee) {
        P.wf.s_ray0 = (float4*)base;
        P.wf.s_ray1 = (float4*)(base + b_ray);
        P.wf.s_con = (float4*)(base + 2 * b_ray);
        P.wf.s_hit = (float2*)(base + 3 * b_ray);
        P.wf.s_cnt = (int*)(base + 3 * b_ray + b_hit);
    }

    hipStream_t st = c->stream;
    P.sc.n_top = 0;
    P.sph_tab = 0;
    P.batch = c->opt_wave_batch;
    P.wf.hit = hit;
    P.wf.hashes = hashes;
    P.wf.queues_all = queues;
    P.wf.queues_words = (uint32_t)q_words;
    P.wf.cap = (uint32_t)cap;
    P.wf.n_regions = (int)n_regions;
    P.wf.n_slots = (uint32_t)((size_t)work_tiles * 64);
    P.wf.bounce = 0;
    hipLaunchKernelGGL(k_wf_prepare, dim3(1), dim3(256), 0, st, P);
    HIP_TRY(c, hipGetLastError());
    if (stage_mark(c, PT_STAGE_GENERATE) != PT_OK) return PT_ERR_DEVICE;

    const size_t lds_ext = (size_t)(L.lstk == 24 ? 24 : 16) * PT_BLOCK * 4;
    const size_t lds_shade = 15 * PT_KSPHERES * 4;
    // launchers: the extend stage's persistent grid (resident blocks, at most `blocks_per_cu` per CU) and the shade stage's one
    // block per region, for the launch parameters Q on stream s
#define PT_EXT(COUNT, OCC, LSTK, FIRST)                                                                           \
        do {                                                                                                      \
            int per_cu = 0;                                                                                       \
            if (allow_lds(k_wf_extend<COUNT, OCC, LSTK, FIRST>, lds_ext) != hipSuccess) return hipErrorInvalidValue; \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_wf_extend<COUNT, OCC, LSTK, FIRST>, PT_BLOCK, lds_ext) != hipSuccess || per_cu < 1) \
                per_cu = 1;                                                                                       \
            per_cu = std::min(per_cu, blocks_per_cu);                                                             \
            hipLaunchKernelGGL((k_wf_extend<COUNT, OCC, LSTK, FIRST>), dim3((unsigned)std::min<size_t>((size_t)per_cu * L.n_cu, (size_t)Q.wf.n_regions)), \
                               dim3(PT_BLOCK), lds_ext, s, Q);                                                    \
        } while (0)
    auto launch_extend = [&](const KParams& Q, bool first, hipStream_t s, int blocks_per_cu) -> hipError_t {
        if (first) {
            if (L.count) { if (L.lstk == 24) PT_EXT(true, 6, 24, true); else PT_EXT(true, 8, 16, true); }
            else { if (L.lstk == 24) PT_EXT(false, 6, 24, true); else PT_EXT(false, 8, 16, true); }
        } else {
            if (L.count) { if (L.lstk == 24) PT_EXT(true, 6, 24, false); else PT_EXT(true, 8, 16, false); }
            else { if (L.lstk == 24) PT_EXT(false, 6, 24, false); else PT_EXT(false, 8, 16, false); }
        }
        return hipGetLastError();
    };
#undef PT_EXT
#define PT_SHADE(COUNT, NEE, FIRST) \
        hipLaunchKernelGGL((k_wf_shade<COUNT, NEE, FIRST>), dim3((unsigned)Q.wf.n_regions), dim3(PT_BLOCK), lds_shade, s, Q)
    auto launch_shade = [&](const KParams& Q, bool first, hipStream_t s) -> hipError_t {
        if (!first && !nee && !L.count && Q.wf.bounce + 1 >= Q.depth) {   // the final bounce: emission only
            hipLaunchKernelGGL((k_wf_shade<false, false, false, true>), dim3((unsigned)Q.wf.n_regions), dim3(PT_BLOCK), lds_shade, s, Q);
            return hipGetLastError();
        }
        if (first) {
            if (nee) { if (L.count) PT_SHADE(true, true, true); else PT_SHADE(false, true, true); }
            else { if (L.count) PT_SHADE(true, false, true); else PT_SHADE(false, false, true); }
        } else {
            if (nee) { if (L.count) PT_SHADE(true, true, false); else PT_SHADE(false, true, false); }
            else { if (L.count) PT_SHADE(true, false, false); else PT_SHADE(false, false, false); }
        }
        return hipGetLastError();
    };
#undef PT_SHADE
    auto set_bounce = [&](KParams& Q, uint32_t b) {
        const int g = (int)(b & 1u);
        Q.wf.bounce = b;
        Q.wf.ray0_in = ray0[g]; Q.wf.ray1_in = ray1[g]; Q.wf.mask_in = mask[g]; Q.wf.cnt_in = cnt[g];
        Q.wf.ray0_out = ray0[g ^ 1]; Q.wf.ray1_out = ray1[g ^ 1]; Q.wf.mask_out = mask[g ^ 1]; Q.wf.cnt_out = cnt[g ^ 1];
    };

    for (uint32_t b = 0; b < P.depth; b++) {
        set_bounce(P, b);
        P.wf.queue = queues + (size_t)b * PT_SHARDS * PT_SHARD_STRIDE;
        HIP_TRY(c, launch_extend(P, b == 0, st, c->opt_wave_blocks));
        if (stage_mark(c, PT_STAGE_EXTEND) != PT_OK) return PT_ERR_DEVICE;
        HIP_TRY(c, launch_shade(P, b == 0, st));
        if (stage_mark(c, PT_STAGE_SHADE) != PT_OK) return PT_ERR_DEVICE;
        if (nee) {   // this bounce's shadow rays: the same extend kernel over the shadow records, then the resolve
            KParams S = P;
            S.wf.ray0_in = P.wf.s_ray0; S.wf.ray1_in = P.wf.s_ray1; S.wf.cnt_in = P.wf.s_cnt; S.wf.hit = P.wf.s_hit;
            S.wf.queue = queues + ((size_t)P.depth + b) * PT_SHARDS * PT_SHARD_STRIDE;
            HIP_TRY(c, launch_extend(S, false, st, c->opt_wave_blocks));
            if (stage_mark(c, PT_STAGE_EXTEND) != PT_OK) return PT_ERR_DEVICE;
            hipLaunchKernelGGL(k_wf_resolve, dim3((unsigned)n_regions), dim3(PT_BLOCK), 0, st, P);
            HIP_TRY(c, hipGetLastError());
            if (stage_mark(c, PT_STAGE_SHADE) != PT_OK) return PT_ERR_DEVICE;
        }
    }
    return PT_OK;
}

}  // namespace ptmi

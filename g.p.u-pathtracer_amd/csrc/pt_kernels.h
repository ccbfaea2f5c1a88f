// pt_kernels.h — the path loop as gfx950 kernels.
//
// Reference path:  trace (GpuPathTracer/tracer.cu:343-400) → getSample
// (:27-339) → intersectBVHandTriangles (GpuPathTracer/cudaUtils.h:256-460) +
// intersectAllSpeheres (:221-236) → BRDF block (tracer.cu:156-293) → accumulate + pack
// (:386-398).  Not a translation: wave64 tiles, traversal stack in LDS, re-laid-out
// scene, counter RNG keyed by pixel.
#pragma once
#include "pt_math.h"
#include "../../include/ptmi.h"

#define PT_BLOCK 256          // 4 waves; one wave = one 8x8 pixel tile
#define PT_TILE 8

struct KScene {
    const float4* __restrict__ nodes;   // 4 float4 per inner node, links = float4 indices
    const float4* __restrict__ tris;    // 3 float4 per triangle reference (v0|id, e1|last, e2|0)
    const pt_sphere_d* __restrict__ spheres;
    int n_spheres;
    int has_bvh;
};

struct KParams {
    KScene sc;
    float* __restrict__ accum;
    uint32_t* __restrict__ rgba;
    unsigned long long* counters;      // 6 x u64 when instrumented
    pt_camera cam;
    int W, H;
    uint32_t depth;
    int cull;
    uint64_t frame, sample_index;
    uint32_t spp;
    int tri_mat;
    float tri_col[3], tri_emi[3], bk[3];
    float air_ior, glass_ior, phong;
    uint32_t flags;
    // tile enumeration: tiles_x tiles per tile-row; this launch covers n_tiles tiles taken
    // from the tile-rows this partition owns (stripes of stripe_tr tile-rows, round-robin)
    int tiles_x, tile_rows, n_tiles;
    int part_index, part_count, stripe_tr;
};

struct Hit {
    float t;   // PT_F32_MAX on miss
    int tri;   // original triangle id, -1 on miss
    v3 n;      // cross(v0-v1, v0-v2) of the winner
};

struct TravCount {
    uint32_t inner, tris, leaves;
};

// ---------------------------------------------------------------------------------------
// Binary-tree closest hit, same visiting order and arithmetic as cudaUtils.h:256-460 /
// oracle/pt_oracle.c:bvh_intersect, so results are bit-identical to the oracle.
//   - stack lives in LDS, laid out [entry][thread] → conflict-free for any mix of depths
//   - slab tests: 12 v_fma + v_min3/v_max3 (the reference's PTX vmin/vmax trick is only
//     valid for non-negative floats)
//   - postponed-leaf exit on a 64-lane ballot (cudaUtils.h:383-394 is a 32-lane vote)
template <bool COUNT>
__device__ __forceinline__ Hit trav_bvh2(const KScene& sc, v3 o, v3 d, bool cull, int* __restrict__ stk,
                                         TravCount& tc) {
    const float ooeps = 8.271806125530277e-25f;  // exp2f(-80), cudaUtils.h:283
    const float idx = 1.0f / (fabsf(d.x) > ooeps ? d.x : copysignf(ooeps, d.x));
    const float idy = 1.0f / (fabsf(d.y) > ooeps ? d.y : copysignf(ooeps, d.y));
    const float idz = 1.0f / (fabsf(d.z) > ooeps ? d.z : copysignf(ooeps, d.z));
    const float oodx = o.x * idx, oody = o.y * idy, oodz = o.z * idz;

    int sp = 0;
    stk[0] = PT_SENTINEL;
    int leaf = 0, node = 0;
    Hit h;
    h.t = PT_F32_MAX; h.tri = -1; h.n = V3(0.f, 0.f, 0.f);

    while (node != PT_SENTINEL) {
        while ((unsigned)node < (unsigned)PT_SENTINEL) {  // node >= 0 && node != sentinel
            const float4 n0 = sc.nodes[node + 0];
            const float4 n1 = sc.nodes[node + 1];
            const float4 nz = sc.nodes[node + 2];
            const float4 nl = sc.nodes[node + 3];
            if (COUNT) tc.inner++;
            const float c0lox = fmaf(n0.x, idx, -oodx), c0hix = fmaf(n0.y, idx, -oodx);
            const float c0loy = fmaf(n0.z, idy, -oody), c0hiy = fmaf(n0.w, idy, -oody);
            const float c1lox = fmaf(n1.x, idx, -oodx), c1hix = fmaf(n1.y, idx, -oodx);
            const float c1loy = fmaf(n1.z, idy, -oody), c1hiy = fmaf(n1.w, idy, -oody);
            const float c0loz = fmaf(nz.x, idz, -oodz), c0hiz = fmaf(nz.y, idz, -oodz);
            const float c1loz = fmaf(nz.z, idz, -oodz), c1hiz = fmaf(nz.w, idz, -oodz);
            const float c0min = fmaxf(fmaxf(fmaxf(fminf(c0lox, c0hix), fminf(c0loy, c0hiy)), fminf(c0loz, c0hiz)), 0.0f);
            const float c0max = fminf(fminf(fminf(fmaxf(c0lox, c0hix), fmaxf(c0loy, c0hiy)), fmaxf(c0loz, c0hiz)), h.t);
            const float c1min = fmaxf(fmaxf(fmaxf(fminf(c1lox, c1hix), fminf(c1loy, c1hiy)), fminf(c1loz, c1hiz)), 0.0f);
            const float c1max = fminf(fminf(fminf(fmaxf(c1lox, c1hix), fmaxf(c1loy, c1hiy)), fmaxf(c1loz, c1hiz)), h.t);
            const bool t0 = (c0min <= c0max) && (c0min >= 0.0f) && (c0min <= PT_F32_MAX);
            const bool t1 = (c1min <= c1max) && (c1min >= 0.0f) && (c1min <= PT_F32_MAX);
            if (!t0 && !t1) {
                node = stk[sp * PT_BLOCK];
                sp--;
            } else {
                int cx = __float_as_int(nl.x), cy = __float_as_int(nl.y);
                node = t0 ? cx : cy;
                if (t0 && t1) {
                    if (c1min < c0min) { int tmp = node; node = cy; cy = tmp; }
                    sp++;
                    stk[sp * PT_BLOCK] = cy;
                }
            }
            if (node < 0 && leaf >= 0) {  // first leaf: postpone, keep descending
                leaf = node;
                node = stk[sp * PT_BLOCK];
                sp--;
            }
            if (!__ballot(leaf >= 0)) break;  // every active lane holds a leaf
        }
        while (leaf < 0) {
            if (COUNT) tc.leaves++;
            for (int a = ~leaf;; a += 3) {
                const float4 r0 = sc.tris[a + 0];
                const float4 r1 = sc.tris[a + 1];
                const float4 r2 = sc.tris[a + 2];
                if (COUNT) tc.tris++;
                const v3 v0 = V3(r0.x, r0.y, r0.z), e1 = V3(r1.x, r1.y, r1.z), e2 = V3(r2.x, r2.y, r2.z);
                const float t = pt_mt_intersect(v0, e1, e2, o, d, cull);
                const int id = __float_as_int(r0.w);
                if (t > 0.0f && (t < h.t || (t == h.t && h.tri != -1 && id < h.tri))) {
                    h.t = t;
                    h.tri = id;
                    // cross(v0-v1, v0-v2) == cross(e1, e2) exactly: v0-v1 = -(v1-v0) bit for bit
                    h.n = vcross(vsub(V3(0.f, 0.f, 0.f), e1), vsub(V3(0.f, 0.f, 0.f), e2));
                }
                if (__float_as_int(r1.w) != 0) break;  // last record of the leaf
            }
            leaf = node;
            if (node < 0) {
                node = stk[sp * PT_BLOCK];
                sp--;
            }
        }
    }
    return h;
}

// ---------------------------------------------------------------------------------------
// One sample of one pixel: getSample, tracer.cu:27-339.
template <bool COUNT>
__device__ __forceinline__ v3 pt_get_sample(const KParams& P, int px, int py, pt_rng& rng, int* __restrict__ stk,
                                            TravCount& tc, uint32_t& n_rays, uint32_t& n_hits) {
    // getCamRayDir, cudaUtils.h:111-134 (ray origin is ON the image plane)
    const float u0 = pt_rng_next(rng), u1 = pt_rng_next(rng);
    const float jx = u0 - 0.5f, jy = u1 - 0.5f;
    const float xs = ((((float)px - (float)P.W / 2.0f) + 0.5f) + jx) * P.cam.dist * P.cam.aspect * P.cam.fov / (float)(P.W - 1);
    const float ys = ((((float)py - (float)P.H / 2.0f) + 0.5f) + jy) * P.cam.dist * P.cam.fov / (float)(P.H - 1);
    const v3 front = V3(P.cam.front[0], P.cam.front[1], P.cam.front[2]);
    const v3 right = V3(P.cam.right[0], P.cam.right[1], P.cam.right[2]);
    const v3 up = V3(P.cam.up[0], P.cam.up[1], P.cam.up[2]);
    const v3 dir0 = vmadd(up, ys, vmadd(right, xs, vscale(front, P.cam.dist)));
    v3 o = vadd(V3(P.cam.pos[0], P.cam.pos[1], P.cam.pos[2]), dir0);
    v3 d = vnormalize(dir0);

    v3 mask = V3(1.f, 1.f, 1.f), accu = V3(0.f, 0.f, 0.f);
    const bool cull = P.cull != 0;

    for (uint32_t depth = 0; depth < P.depth; ++depth) {
        int geom = 3;  // GeoType::NONE
        int sph_id = -1;
        Hit h;
        h.t = PT_F32_MAX; h.tri = -1; h.n = V3(0.f, 0.f, 0.f);
        if (P.sc.has_bvh) h = trav_bvh2<COUNT>(P.sc, o, d, cull, stk, tc);
        if (COUNT) { n_rays++; n_hits += (h.tri != -1); }
        float scene_t = h.t;
        if (h.tri != -1) geom = 0;
        // intersectAllSpeheres, cudaUtils.h:221-236 (uniform loop, scalar loads)
        for (int i = 0; i < P.sc.n_spheres; i++) {
            const pt_sphere_d& s = P.sc.spheres[i];
            const float ts = pt_sphere_intersect(s.px, s.py, s.pz, s.rad, o, d);
            if (ts != 0.0f && ts < scene_t && ts > 0.01f) { scene_t = ts; sph_id = i; geom = 1; }
        }
        v3 hitpos = vmadd(d, scene_t, o);
        v3 n, nl, objcol, emit;
        int mat;
        if (geom == 1) {
            const pt_sphere_d& s = P.sc.spheres[sph_id];
            n = vnormalize(vsub(hitpos, V3(s.px, s.py, s.pz)));
            nl = vdot(n, d) < 0 ? n : vscale(n, -1.0f);
            objcol = V3(s.col[0], s.col[1], s.col[2]);
            emit = V3(s.emi[0], s.emi[1], s.emi[2]);
            mat = s.mat;
        } else if (geom == 0) {
            n = vnormalize(h.n);
            nl = n;  // tracer.cu:126-127
            objcol = V3(P.tri_col[0], P.tri_col[1], P.tri_col[2]);
            emit = V3(P.tri_emi[0], P.tri_emi[1], P.tri_emi[2]);
            mat = P.tri_mat;
        } else {
            return V3(P.bk[0], P.bk[1], P.bk[2]);  // tracer.cu:140-142: unmasked background
        }
        accu = vadd(accu, vmul(mask, emit));

        v3 nextdir;
        if (mat == PT_MAT_DIFF) {  // tracer.cu:156-186
            (void)pt_rng_next(rng);
            (void)pt_rng_next(rng);
            v3 nt = fabsf(nl.x) > fabsf(nl.y) ? V3(nl.z, 0.f, -nl.x) : V3(0.f, -nl.z, nl.y);
            nt = vnormalize(nt);
            const v3 nb = vnormalize(vcross(nl, nt));
            const float f1 = pt_rng_next(rng), f2 = pt_rng_next(rng);
            float c, s;
            pt_sincos2pi(f1, c, s);
            const v3 rv = V3(c * f2, sqrtf(1.0f - f2 * f2), s * f2);  // cudaUtils.h:185-192
            nextdir = vnormalize(vmadd(nt, rv.z, vmadd(nl, rv.y, vscale(nb, rv.x))));
            hitpos = vmadd(nl, 0.001f, hitpos);
            mask = vmul(mask, objcol);
        } else if (mat == PT_MAT_SPEC) {  // :190-203
            nextdir = vnormalize(vmadd(nl, -2.0f * vdot(nl, d), d));
            hitpos = vmadd(nl, 0.001f, hitpos);
            mask = vmul(mask, objcol);
        } else if (mat == PT_MAT_REFR) {  // :205-256
            const bool into = vdot(n, nl) > 0;
            const float nc = P.air_ior, ntt = P.glass_ior;
            const float nnt = into ? nc / ntt : ntt / nc;
            const float ddn = vdot(d, nl);
            const float cos2t = 1.0f - nnt * nnt * (1.0f - ddn * ddn);
            if (cos2t < 0.0f) {
                nextdir = vnormalize(vmadd(n, -2.0f * vdot(n, d), d));
                hitpos = vmadd(nl, 0.001f, hitpos);
            } else {
                const float k = (into ? 1.0f : -1.0f) * (ddn * nnt + sqrtf(cos2t));
                const v3 tdir = vnormalize(vmadd(n, -k, vscale(d, nnt)));
                const float R0 = (ntt - nc) * (ntt - nc) / (ntt + nc) * (ntt + nc);  // sic, :230
                const float c = 1.0f - (into ? -ddn : vdot(tdir, n));
                const float Re = R0 + (1.0f - R0) * c * c * c * c * c;
                const float Tr = 1 - Re;
                const float Pp = 0.25f + 0.5f * Re;
                const float RP = Re / Pp, TP = Tr / (1.0f - Pp);
                if (pt_rng_next(rng) < 0.2f) {
                    mask = vscale(mask, RP);
                    nextdir = vnormalize(vmadd(n, -2.0f * vdot(n, d), d));
                } else {
                    mask = vscale(mask, TP);
                    nextdir = vnormalize(tdir);
                }
                hitpos = vmadd(nl, 0.001f, hitpos);
            }
        } else {  // METAL :257-293
            const float f1 = pt_rng_next(rng), r2 = pt_rng_next(rng);
            float cphi, sphi;
            pt_sincos2pi(f1, cphi, sphi);
            const float cosT = pt_pow01(1.0f - r2, 1.0f / (P.phong + 1.0f));
            const float sinT = sqrtf(1.0f - cosT * cosT);
            const v3 w1 = vnormalize(vmadd(nl, -2.0f * vdot(nl, d), d));
            const v3 ax = ((double)fabsf(w1.x) > 0.1) ? V3(0.f, 1.f, 0.f) : V3(1.f, 0.f, 0.f);
            const v3 uu = vnormalize(vcross(ax, w1));
            const v3 vv = vcross(w1, uu);
            const v3 base = vmadd(vv, sphi * sinT, vscale(uu, cphi * sinT));
            if (P.flags & PT_FLAG_METAL_LITERAL_W) {
                const float wc = (float)P.W * cosT;  // tracer.cu:280
                nextdir = V3(base.x + wc, base.y + wc, base.z + wc);
            } else {
                nextdir = vmadd(w1, cosT, base);
            }
            nextdir = vnormalize(nextdir);
            hitpos = vmadd(nl, 0.0001f, hitpos);
            mask = vmul(mask, objcol);
        }
        o = hitpos;
        d = nextdir;
    }
    return accu;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Maps the launch's linear tile number to the global tile coordinates this partition owns.
__device__ __forceinline__ bool pt_tile_coords(const KParams& P, int tile, int& tx, int& ty) {
    if (tile >= P.n_tiles) return false;
    const int lrow = tile / P.tiles_x;
    tx = tile - lrow * P.tiles_x;
    if (P.part_count > 1) {
        const int k = lrow / P.stripe_tr, within = lrow - k * P.stripe_tr;
        ty = (P.part_index + k * P.part_count) * P.stripe_tr + within;
    } else {
        ty = lrow;
    }
    return ty < P.tile_rows;
}

// trace<<<>>>, tracer.cu:343-400: one lane per pixel, one wave per 8x8 tile, `spp`
// consecutive samples folded in registers.
template <int STACK, bool COUNT>
__global__ void __launch_bounds__(PT_BLOCK) k_trace_mega_bvh2(const KParams P) {
    __shared__ int s_stack[STACK * PT_BLOCK];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int tile = blockIdx.x * (PT_BLOCK / 64) + (tid >> 6);
    int tx, ty;
    if (!pt_tile_coords(P, tile, tx, ty)) return;
    const int px = tx * PT_TILE + (lane & 7), py = ty * PT_TILE + (lane >> 3);
    if (px >= P.W || py >= P.H) return;  // tracer.cu:358
    const uint64_t pix = (uint64_t)py * (uint64_t)P.W + (uint64_t)px;
    int* stk = s_stack + tid;

    TravCount tc;
    tc.inner = tc.tris = tc.leaves = 0;
    uint32_t n_rays = 0, n_hits = 0;

    float* acc = P.accum + 3 * pix;
    float ax = 0.f, ay = 0.f, az = 0.f;
    if (P.sample_index != 1) { ax = acc[0]; ay = acc[1]; az = acc[2]; }
    for (uint32_t s = 0; s < P.spp; s++) {
        pt_rng rng = pt_rng_init(pt_wang64(P.frame + s), pix);  // tracer.cu:362-363
        const v3 col = pt_get_sample<COUNT>(P, px, py, rng, stk, tc, n_rays, n_hits);
        // running mean with per-frame clamp, tracer.cu:386-391
        const uint64_t N = P.sample_index + s;
        const float fm1 = (float)(N - 1), inv = 1.0f / (float)N;
        if (N == 1) { ax = 0.f; ay = 0.f; az = 0.f; } else { ax *= fm1; ay *= fm1; az *= fm1; }
        ax = pt_clamp01((ax + col.x) * inv);
        ay = pt_clamp01((ay + col.y) * inv);
        az = pt_clamp01((az + col.z) * inv);
    }
    acc[0] = ax; acc[1] = ay; acc[2] = az;
    if ((P.flags & PT_FLAG_WRITE_RGBA) && P.rgba) {  // tracer.cu:394-398, cudaUtils.h:99-105
        const uint32_t r = (uint32_t)(unsigned char)(255.0f * ax);
        const uint32_t g = (uint32_t)(unsigned char)(255.0f * ay);
        const uint32_t b = (uint32_t)(unsigned char)(255.0f * az);
        P.rgba[pix] = (b << 16) | (g << 8) | r;
    }
    if (COUNT) {
        const uint32_t a = wave_sum_u32(n_rays), b = wave_sum_u32(tc.inner), c = wave_sum_u32(tc.tris);
        const uint32_t dd = wave_sum_u32(tc.leaves), e = wave_sum_u32(n_hits), f = wave_sum_u32(P.spp);
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
template <int STACK>
__global__ void __launch_bounds__(PT_BLOCK) k_trace_rays_bvh2(const KScene sc, const float4* __restrict__ rays, size_t n,
                                                              int cull, float* __restrict__ t_out,
                                                              int* __restrict__ tri_out, float* __restrict__ n_out) {
    __shared__ int s_stack[STACK * PT_BLOCK];
    const size_t i = (size_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 ro = rays[2 * i], rd = rays[2 * i + 1];
    TravCount tc;
    tc.inner = tc.tris = tc.leaves = 0;
    const Hit h = trav_bvh2<false>(sc, V3(ro.x, ro.y, ro.z), V3(rd.x, rd.y, rd.z), cull != 0, s_stack + threadIdx.x, tc);
    t_out[i] = h.t;
    tri_out[i] = h.tri;
    if (n_out) { n_out[3 * i] = h.n.x; n_out[3 * i + 1] = h.n.y; n_out[3 * i + 2] = h.n.z; }
}

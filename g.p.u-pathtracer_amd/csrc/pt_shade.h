// pt_shade.h — one sample of one pixel cut into pieces: camera ray, sphere tests, shading / BRDF, accumulate
// (getSample, tracer.cu:27-339; rows a2-a4, a8-a10).  Included by pt_kernels.h.
#pragma once

// ---------------------------------------------------------------------------------------
// One sample of one pixel: getSample, tracer.cu:27-339, cut into the pieces both kernels
// share: path_begin (camera ray), trav_* (closest hit), path_shade (spheres, shading, BRDF).
struct PathState {
    v3 o, d, mask, accu;
    uint32_t depth;
    pt_rng rng;
    uint32_t nee_mask;   // PT_FLAG_NEE: spheres whose light the previous bounce already sampled with a shadow ray
};

// PT_FLAG_NEE: the shadow ray a DIFF hit asks for.  The caller traces it against the triangles (the spheres in its way
// have been tested already) and, when nothing is hit closer than t_max, adds `contrib` to the path's gathered light.
struct NeeReq {
    bool want;
    v3 o, d;
    float t_max;
    v3 contrib;
};

// The kernel-argument block seen through an opaque pointer (constant address space, scalar loads):
// a field read as K.x is fetched at that point instead of being preloaded and kept in SGPRs for
// the whole kernel.  KParams is the one and only argument of every kernel that uses this: offset 0.
#define PT_KARGS(K)                                                                               \
    const __attribute__((address_space(4))) KParams* K##_p =                                      \
        (const __attribute__((address_space(4))) KParams*)__builtin_amdgcn_kernarg_segment_ptr(); \
    asm volatile("" : "+s"(K##_p));                                                               \
    const __attribute__((address_space(4))) KParams& K = *K##_p

// RNG seed (tracer.cu:362-363) + getCamRayDir, cudaUtils.h:111-134 (origin ON the image plane)
// (frame_hash = pt_wang64(frame), uf::hash of utilfun.cpp:380-389)
__device__ __forceinline__ void path_begin_hashed(const KParams& P, int px, int py, uint64_t pix, uint64_t frame_hash, PathState& ps) {
    PT_KARGS(K);   // camera: read where it is used, not held in SGPRs across the persistent loop
    ps.rng = pt_rng_init(frame_hash, pix);
    const float u0 = pt_rng_next(ps.rng), u1 = pt_rng_next(ps.rng);
    const float jx = u0 - 0.5f, jy = u1 - 0.5f;
    const float xs = ((((float)px - (float)P.W / 2.0f) + 0.5f) + jx) * K.cam.dist * K.cam.aspect * K.cam.fov / (float)(P.W - 1);
    const float ys = ((((float)py - (float)P.H / 2.0f) + 0.5f) + jy) * K.cam.dist * K.cam.fov / (float)(P.H - 1);
    const v3 front = V3(K.cam.front[0], K.cam.front[1], K.cam.front[2]);
    const v3 right = V3(K.cam.right[0], K.cam.right[1], K.cam.right[2]);
    const v3 up = V3(K.cam.up[0], K.cam.up[1], K.cam.up[2]);
    const v3 dir0 = vmadd(up, ys, vmadd(right, xs, vscale(front, K.cam.dist)));
    ps.o = vadd(V3(K.cam.pos[0], K.cam.pos[1], K.cam.pos[2]), dir0);
    ps.d = vnormalize(dir0);
    ps.mask = V3(1.f, 1.f, 1.f);
    ps.accu = V3(0.f, 0.f, 0.f);
    ps.depth = 0;
    ps.nee_mask = 0;
}
__device__ __forceinline__ void path_begin(const KParams& P, int px, int py, uint64_t pix, uint64_t frame, PathState& ps) {
    path_begin_hashed(P, px, py, pix, pt_wang64(frame), ps);
}

// What a segment ended on once the spheres have been tested too (intersectAllSpeheres,
// cudaUtils.h:221-236, after the triangle hit of the walk): the distance, GeoType and sphere number.
struct SceneHit {
    float t;
    int geom;     // 0 triangle, 1 sphere, 3 nothing (GeoType, CommomStructs.hpp)
    int sph_id;
};

// sph_tab: float index into the dynamic LDS of a copy of the first PT_KSPHERES spheres (11 floats
// each, then centre+radius as float4s), or -1 = read the kernel arguments.
__device__ __forceinline__ SceneHit pt_closest_sphere(const KParams& P, v3 o, v3 d, const Hit& h, int sph_tab = -1) {
    PT_KARGS(K);
    int geom = 3;  // GeoType::NONE
    int sph_id = -1;
    float scene_t = h.t;
    if (h.tri != -1) geom = 0;
    // intersectAllSpeheres, cudaUtils.h:221-236 (uniform loop, scalar loads)
    // The reference scene has 8 spheres (BasicScene.cpp:181-202): they ride in the kernel-argument
    // block and the loop is unrolled, so the data arrives by scalar loads issued up front.  Inside
    // the divergent service phase hipcc otherwise keeps the loop counter in a VGPR and fetches each
    // sphere with dependent vector loads (a quarter of the kernel's vector-memory instructions).
    if (P.sc.n_spheres <= PT_KSPHERES) {
        // The centres/radii ride in the kernel-argument block, but are re-read HERE, one scalar load
        // per sphere behind an opaque pointer: kept in SGPRs across the whole persistent loop they
        // push ~50 other scalars into spill lanes (v_readlane/v_writelane on the hot path; measured
        // -4 % frame time, and what lets 6 waves per SIMD pay off).  A plain global pointer makes
        // hipcc fetch them with per-lane vector loads instead (+2 %).
        // KParams is the one and only kernel argument of every kernel that shades: offset 0.
        typedef const __attribute__((address_space(4))) float kfloat;
        kfloat* kp = (kfloat*)&K.ksph[0];
#pragma unroll
        for (int i = 0; i < PT_KSPHERES; i++) {
            if (i < P.sc.n_spheres) {
                struct { float px, py, pz, rad; } s;
                if (sph_tab >= 0) {  // role-split kernel: one wave shades alone on its SIMD, so eight scalar-load
                                     // round trips in a row are exposed; the LDS copy is read as pipelined broadcasts
                    const float4 c = *(const float4*)((const float*)s_dyn + sph_tab + 88 + 4 * i);
                    s.px = c.x; s.py = c.y; s.pz = c.z; s.rad = c.w;
                } else {
                    s.px = kp[11 * i]; s.py = kp[11 * i + 1]; s.pz = kp[11 * i + 2]; s.rad = kp[11 * i + 3];
                }
                const float ts = pt_sphere_intersect(s.px, s.py, s.pz, s.rad, o, d);
                if (ts != 0.0f && ts < scene_t && ts > 0.01f) { scene_t = ts; sph_id = i; geom = 1; }
            }
        }
    } else {
        for (int i = 0; i < P.sc.n_spheres; i++) {
            const pt_sphere_d& s = P.sc.spheres[i];
            const float ts = pt_sphere_intersect(s.px, s.py, s.pz, s.rad, o, d);
            if (ts != 0.0f && ts < scene_t && ts > 0.01f) { scene_t = ts; sph_id = i; geom = 1; }
        }
    }
    SceneHit sh;
    sh.t = scene_t; sh.geom = geom; sh.sph_id = sph_id;
    return sh;
}

// One bounce after the closest triangle hit `h` and the sphere tests `sh` are known
// (tracer.cu:98-296).  Returns true when the sample is complete (col_out valid), false when ps
// holds the next ray segment.  With sph_tab >= 0 the winner sphere's attributes are one short LDS
// gather instead of a global one.
// tri_n: pt_hit_normal of the walk's triangle hit (read only when the triangle is what was hit; the
// caller fetches it early so that the latency hides behind other work).
__device__ __forceinline__ bool path_shade_hit(const KParams& P, PathState& ps, const Hit& h, const SceneHit& sh, v3 tri_n, v3& col_out, int sph_tab = -1,
                                               NeeReq* nee = nullptr) {
    if (nee) nee->want = false;
    PT_KARGS(K);   // shading scalars: read where they are used (see the sphere loop)
    const v3 o = ps.o, d = ps.d;
    v3 mask = ps.mask, accu = ps.accu;
    pt_rng rng = ps.rng;
    {
    const int geom = sh.geom, sph_id = sh.sph_id;
    const float scene_t = sh.t;
    v3 hitpos = vmadd(d, scene_t, o);
    v3 n, nl, objcol, emit;
    int mat;
    float phong = K.phong;
    if (geom == 1) {
        pt_sphere_d s;
        if (sph_tab >= 0 && sph_id < PT_KSPHERES) {
            const float* t = (const float*)s_dyn + sph_tab + 11 * sph_id;
            s.px = t[0]; s.py = t[1]; s.pz = t[2]; s.rad = t[3];
            s.emi[0] = t[4]; s.emi[1] = t[5]; s.emi[2] = t[6];
            s.col[0] = t[7]; s.col[1] = t[8]; s.col[2] = t[9];
            s.mat = __float_as_int(t[10]);
        } else {
            s = P.sc.spheres[sph_id];
        }
        n = vnormalize(vsub(hitpos, V3(s.px, s.py, s.pz)));
        nl = vdot(n, d) < 0 ? n : vscale(n, -1.0f);
        objcol = V3(s.col[0], s.col[1], s.col[2]);
        emit = V3(s.emi[0], s.emi[1], s.emi[2]);
        mat = s.mat;
    } else if (geom == 0) {
        n = vnormalize(tri_n);
        nl = n;  // tracer.cu:126-127
        if ((K.flags & PT_FLAG_FACE_FORWARD) && !(vdot(n, d) < 0)) nl = vscale(n, -1.0f);
        if (K.tri_matid) {  // extension: per-triangle material row
            const int row = K.tri_matid[h.tri];
            const float4 m0 = K.mat_table[2 * row], m1 = K.mat_table[2 * row + 1];
            objcol = V3(m0.x, m0.y, m0.z);
            emit = V3(m0.w, m1.x, m1.y);
            mat = __float_as_int(m1.z);
            phong = m1.w;
        } else {
            objcol = V3(K.tri_col[0], K.tri_col[1], K.tri_col[2]);
            emit = V3(K.tri_emi[0], K.tri_emi[1], K.tri_emi[2]);
            mat = K.tri_mat;
        }
    } else {
        col_out = V3(K.bk[0], K.bk[1], K.bk[2]);  // tracer.cu:140-142: unmasked background
        if (K.flags & PT_FLAG_MISS_KEEPS_PATH) col_out = vadd(accu, vmul(mask, col_out));  // extension
        return true;
    }
    // not already gathered by a shadow ray (PT_FLAG_NEE): bits 0-7 the spheres that were eligible, bit 8 the emissive triangles
    if (!(geom == 1 && sph_id < 8 && ((ps.nee_mask >> sph_id) & 1u)) && !(geom == 0 && (ps.nee_mask & 0x100u)))
        accu = vadd(accu, vmul(mask, emit));
    ps.nee_mask = 0;

    if ((K.flags & PT_FLAG_RUSSIAN_ROULETTE) && ps.depth >= 2) {  // extension
        const float pr = fmaxf(objcol.x, fmaxf(objcol.y, objcol.z));
        if (!(pt_rng_next(rng) < pr)) { col_out = accu; return true; }
        objcol = vscale(objcol, 1.0f / pr);
    }

    if ((K.flags & PT_FLAG_RR_CPU_TRACER) && ps.depth >= 5) {  // extension: CpuRayTracer/src/scene.cpp:38-47
        const float pr = fmaxf(objcol.x, fmaxf(objcol.y, objcol.z));
        if (!(pt_rng_next(rng) < pr * 0.9f)) { col_out = accu; return true; }
        objcol = vscale(objcol, 0.9f / pr);
    }

    v3 nextdir;
    if (mat == PT_MAT_DIFF) {  // tracer.cu:156-186
        if (!(K.flags & PT_FLAG_COSINE_DIFF)) {
            (void)pt_rng_next(rng);
            (void)pt_rng_next(rng);
        }
        v3 nt = fabsf(nl.x) > fabsf(nl.y) ? V3(nl.z, 0.f, -nl.x) : V3(0.f, -nl.z, nl.y);
        nt = vnormalize(nt);
        const v3 nb = vnormalize(vcross(nl, nt));
        const float f1 = pt_rng_next(rng), f2 = pt_rng_next(rng);
        float c, s;
        pt_sincos2pi(f1, c, s);
        v3 rv;
        if (K.flags & PT_FLAG_COSINE_DIFF) {  // extension: pdf = cos/pi
            const float r2s = sqrtf(f2);
            rv = V3(c * r2s, sqrtf(1.0f - f2), s * r2s);
        } else {
            rv = V3(c * f2, sqrtf(1.0f - f2 * f2), s * f2);  // cudaUtils.h:185-192
        }
        nextdir = vnormalize(vmadd(nt, rv.z, vmadd(nl, rv.y, vscale(nb, rv.x))));
        hitpos = vmadd(nl, 0.001f, hitpos);
        mask = vmul(mask, objcol);
        if (nee && (K.flags & PT_FLAG_NEE) && (K.flags & PT_FLAG_COSINE_DIFF)) {  // extension: next-event estimation (ptmi.h)
            // lights = emissive spheres (of the first 8, in the kernel arguments) the point is outside of
            typedef const __attribute__((address_space(4))) float kfloat;
            kfloat* kp = (kfloat*)&K.ksph[0];
            uint32_t el = 0;
            int n_el = 0;
#pragma unroll
            for (int i = 0; i < PT_KSPHERES; i++) {
                if (i < P.sc.n_spheres) {
                    const bool lit = !(kp[11 * i + 4] == 0.0f && kp[11 * i + 5] == 0.0f && kp[11 * i + 6] == 0.0f);
                    const v3 w = vsub(V3(kp[11 * i], kp[11 * i + 1], kp[11 * i + 2]), hitpos);
                    if (lit && vdot(w, w) > (kp[11 * i + 3] * kp[11 * i + 3]) * 1.001f) { el |= 1u << i; n_el++; }
                }
            }
            const int n_tl = K.n_tri_lights, n_all = n_el + n_tl;
            ps.nee_mask = el | (n_tl > 0 ? 0x100u : 0u);
            if (n_all > 0) {
                const float u0 = pt_rng_next(rng), u1 = pt_rng_next(rng), u2 = pt_rng_next(rng);
                int pick = (int)(u0 * (float)n_all);
                if (pick > n_all - 1) pick = n_all - 1;
                if (pick >= n_el) {
                    // an emissive triangle: a point uniform on it; the faces a path can hit emit (see `blocked`).  contribution = mask * Le * cos(surface) * cos(light) * area / (pi * dist^2) * lights
                    const float4* tl = P.tri_lights + 3 * (size_t)(pick - n_el);
                    const float4 t0 = tl[0], t1 = tl[1], t2 = tl[2];
                    const v3 e1 = V3(t1.x, t1.y, t1.z), e2 = V3(t2.x, t2.y, t2.z);
                    const float su = sqrtf(u1);
                    const v3 pl = vmadd(e2, u2 * su, vmadd(e1, 1.0f - su, V3(t0.x, t0.y, t0.z)));
                    const v3 w = vsub(pl, hitpos);
                    const float d2 = vdot(w, w);
                    const float dist = sqrtf(d2);
                    const v3 l = vscale(w, 1.0f / dist);
                    const float cosl = vdot(nl, l);
                    const float sdot = vdot(vcross(e1, e2), l);
                    const float proj = fabsf(sdot);   // 2 * area * cos(light)
                    const float t_light = dist * 0.999f;
                    // with back-face culling a path only ever HITS the front of a triangle (det > 0 <=> cross(e1, e2) . d < 0,
                    // cudaUtils.h:150-160), so only that face is a light
                    bool blocked = !(cosl > 0.0f) || !(d2 > 0.0f) || !(proj > 0.0f) || (P.cull != 0 && !(sdot < 0.0f));
                    for (int j = 0; j < P.sc.n_spheres; j++) {
                        const pt_sphere_d& sj = P.sc.spheres[j];
                        const float ts = pt_sphere_intersect(sj.px, sj.py, sj.pz, sj.rad, hitpos, l);
                        if (ts != 0.0f && ts < t_light && ts > 0.01f) blocked = true;
                    }
                    if (!blocked) {
                        const float k = ((cosl * (0.5f * proj)) * (float)n_all) / (3.14159274f * d2);
                        nee->want = true;
                        nee->o = hitpos;
                        nee->d = l;
                        nee->t_max = t_light;
                        nee->contrib = vscale(vmul(mask, V3(t0.w, t1.w, t2.w)), k);
                    }
                } else {
                int li = 0;
                {
                    int k = 0;
                    bool found = false;
#pragma unroll
                    for (int i = 0; i < PT_KSPHERES; i++)
                        if (!found && ((el >> i) & 1u)) { if (k == pick) { li = i; found = true; } k++; }
                }
                // the picked light's attributes: per-lane index, so through a global pointer (the kernel-argument
                // block is scalar memory)
                const pt_sphere_d L = P.sc.spheres[li];
                const v3 w = vsub(V3(L.px, L.py, L.pz), hitpos);
                const float d2 = vdot(w, w), r2 = L.rad * L.rad;
                const v3 wn = vscale(w, 1.0f / sqrtf(d2));
                const float cos_max = sqrtf(fmaxf(0.0f, 1.0f - r2 / d2));
                const float cos_t = 1.0f - u1 * (1.0f - cos_max);
                const float sin_t = sqrtf(fmaxf(0.0f, 1.0f - cos_t * cos_t));
                float cp, sp;
                pt_sincos2pi(u2, cp, sp);
                v3 t1 = fabsf(wn.x) > fabsf(wn.y) ? V3(wn.z, 0.f, -wn.x) : V3(0.f, -wn.z, wn.y);
                t1 = vnormalize(t1);
                const v3 b1 = vnormalize(vcross(wn, t1));
                const v3 l = vnormalize(vmadd(t1, sp * sin_t, vmadd(wn, cos_t, vscale(b1, cp * sin_t))));
                const float cosl = vdot(nl, l);
                const float t_light = pt_sphere_intersect(L.px, L.py, L.pz, L.rad, hitpos, l);
                bool blocked = !(cosl > 0.0f) || t_light == 0.0f;
                for (int j = 0; j < P.sc.n_spheres; j++) {
                    const pt_sphere_d& sj = P.sc.spheres[j];
                    const float ts = pt_sphere_intersect(sj.px, sj.py, sj.pz, sj.rad, hitpos, l);
                    if (j != li && ts != 0.0f && ts < t_light && ts > 0.01f) blocked = true;
                }
                if (!blocked) {
                    const float k = (cosl * (2.0f * (1.0f - cos_max))) * (float)n_all;
                    nee->want = true;
                    nee->o = hitpos;
                    nee->d = l;
                    nee->t_max = t_light;
                    nee->contrib = vscale(vmul(mask, V3(L.emi[0], L.emi[1], L.emi[2])), k);
                }
                }
            }
        }
    } else if (mat == PT_MAT_SPEC) {  // :190-203
        nextdir = vnormalize(vmadd(nl, -2.0f * vdot(nl, d), d));
        hitpos = vmadd(nl, 0.001f, hitpos);
        mask = vmul(mask, objcol);
    } else if (mat == PT_MAT_REFR) {  // :205-256
        const bool into = vdot(n, nl) > 0;
        const float nc = K.air_ior, ntt = K.glass_ior;
        const float nnt = into ? nc / ntt : ntt / nc;
        const float ddn = vdot(d, nl);
        const float cos2t = 1.0f - nnt * nnt * (1.0f - ddn * ddn);
        if (cos2t < 0.0f) {
            nextdir = vnormalize(vmadd(n, -2.0f * vdot(n, d), d));
            hitpos = vmadd(nl, 0.001f, hitpos);
        } else {
            const float k = (into ? 1.0f : -1.0f) * (ddn * nnt + sqrtf(cos2t));
            const v3 tdir = vnormalize(vmadd(n, -k, vscale(d, nnt)));
            const bool fix = (K.flags & PT_FLAG_GLASS_FIX) != 0;  // extension
            const float R0 = fix ? ((ntt - nc) * (ntt - nc)) / ((ntt + nc) * (ntt + nc))
                                 : (ntt - nc) * (ntt - nc) / (ntt + nc) * (ntt + nc);  // sic, :230
            const float c = 1.0f - (into ? -ddn : vdot(tdir, n));
            const float Re = R0 + (1.0f - R0) * c * c * c * c * c;
            const float Tr = 1 - Re;
            const float Pp = 0.25f + 0.5f * Re;
            const float RP = Re / Pp, TP = Tr / (1.0f - Pp);
            bool transmitted = false;
            if (pt_rng_next(rng) < (fix ? Pp : 0.2f)) {
                mask = vscale(mask, RP);
                nextdir = vnormalize(vmadd(n, -2.0f * vdot(n, d), d));
            } else {
                mask = vscale(mask, TP);
                nextdir = vnormalize(tdir);
                transmitted = true;
            }
            hitpos = vmadd(nl, (fix && transmitted) ? -0.001f : 0.001f, hitpos);
        }
    } else {  // METAL :257-293
        const float f1 = pt_rng_next(rng), r2 = pt_rng_next(rng);
        float cphi, sphi;
        pt_sincos2pi(f1, cphi, sphi);
        const float cosT = pt_pow01(1.0f - r2, 1.0f / (phong + 1.0f));
        const float sinT = sqrtf(1.0f - cosT * cosT);
        const v3 w1 = vnormalize(vmadd(nl, -2.0f * vdot(nl, d), d));
        const v3 ax = ((double)fabsf(w1.x) > 0.1) ? V3(0.f, 1.f, 0.f) : V3(1.f, 0.f, 0.f);
        const v3 uu = vnormalize(vcross(ax, w1));
        const v3 vv = vcross(w1, uu);
        const v3 base = vmadd(vv, sphi * sinT, vscale(uu, cphi * sinT));
        if (K.flags & PT_FLAG_METAL_LITERAL_W) {
            const float wc = (float)P.W * cosT;  // tracer.cu:280
            nextdir = V3(base.x + wc, base.y + wc, base.z + wc);
        } else {
            nextdir = vmadd(w1, cosT, base);
        }
        nextdir = vnormalize(nextdir);
        hitpos = vmadd(nl, 0.0001f, hitpos);
        mask = vmul(mask, objcol);
    }
        ps.o = hitpos;
        ps.d = nextdir;
    }
    ps.mask = mask; ps.accu = accu; ps.rng = rng;
    ps.depth++;
    if (ps.depth >= P.depth) { col_out = accu; return true; }  // tracer.cu:305
    return false;
}

// The LAST bounce of a path in the stage-split pipeline: nothing of path_shade_hit's BRDF work can reach the picture any more
// (tracer.cu:305 returns accu after `depth` segments), only the hit's emission: returns mask * emit exactly as path_shade_hit
// would have added it to an empty accu (the roulette switches end a path with the same accu).  No normal, no RNG draw.
__device__ __forceinline__ v3 path_last_emission(const KParams& P, const PathState& ps, const Hit& h, const SceneHit& sh, int sph_tab) {
    PT_KARGS(K);
    v3 emit;
    if (sh.geom == 1) {
        if (sph_tab >= 0 && sh.sph_id < PT_KSPHERES) {
            const float* t = (const float*)s_dyn + sph_tab + 11 * sh.sph_id;
            emit = V3(t[4], t[5], t[6]);
        } else {
            const pt_sphere_d& s = P.sc.spheres[sh.sph_id];
            emit = V3(s.emi[0], s.emi[1], s.emi[2]);
        }
        if (sh.sph_id < 8 && ((ps.nee_mask >> sh.sph_id) & 1u)) return V3(0.f, 0.f, 0.f);
    } else {
        if (K.tri_matid) {
            const int row = K.tri_matid[h.tri];
            const float4 m0 = K.mat_table[2 * row], m1 = K.mat_table[2 * row + 1];
            emit = V3(m0.w, m1.x, m1.y);
        } else {
            emit = V3(K.tri_emi[0], K.tri_emi[1], K.tri_emi[2]);
        }
        if (ps.nee_mask & 0x100u) return V3(0.f, 0.f, 0.f);
    }
    return vadd(V3(0.f, 0.f, 0.f), vmul(ps.mask, emit));
}

// spheres + shading in one go (the kernels that shade in the lane that walked)
__device__ __forceinline__ bool path_shade(const KParams& P, PathState& ps, const Hit& h, v3& col_out, int sph_tab = -1, NeeReq* nee = nullptr) {
    // the triangle's normal is asked for NOW so that its latency hides behind the sphere tests
    v3 tri_n = V3(0.f, 0.f, 0.f);
    if (h.tri != -1) tri_n = pt_hit_normal(P.sc, h);
    const SceneHit sh = pt_closest_sphere(P, ps.o, ps.d, h, sph_tab);
    return path_shade_hit(P, ps, h, sh, tri_n, col_out, sph_tab, nee);
}

// running mean with per-frame clamp, tracer.cu:386-391
__device__ __forceinline__ void pt_accumulate(float& ax, float& ay, float& az, v3 col, uint64_t N) {
    const float fm1 = (float)(N - 1), inv = 1.0f / (float)N;
    if (N == 1) { ax = 0.f; ay = 0.f; az = 0.f; } else { ax *= fm1; ay *= fm1; az *= fm1; }
    ax = pt_clamp01((ax + col.x) * inv);
    ay = pt_clamp01((ay + col.y) * inv);
    az = pt_clamp01((az + col.z) * inv);
}

// 8-bit truncating pack 0x00BBGGRR, tracer.cu:394-398 + cudaUtils.h:99-105
__device__ __forceinline__ uint32_t pt_pack_rgba(float ax, float ay, float az) {
    const uint32_t r = (uint32_t)(unsigned char)(255.0f * ax);
    const uint32_t g = (uint32_t)(unsigned char)(255.0f * ay);
    const uint32_t b = (uint32_t)(unsigned char)(255.0f * az);
    return (b << 16) | (g << 8) | r;
}


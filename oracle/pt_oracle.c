/*
 * pt_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's per-pixel path loop, used only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the CHECKER for the HIP
 * path.  Nothing in g.p.u-pathtracer_amd/ may include, link or call this file.
 *
 * What it restates (all paths relative to /root/reference):
 *   GpuPathTracer/tracer.cu:27-339      getSample  (bounce loop, shading, BRDFs)
 *   GpuPathTracer/tracer.cu:343-400     trace      (seed, accumulate, pack)
 *   GpuPathTracer/cudaUtils.h:111-134   getCamRayDir
 *   GpuPathTracer/cudaUtils.h:135-181   intersectRayTriangle(Edge)   (Moller-Trumbore)
 *   GpuPathTracer/cudaUtils.h:185-192   uniformSampleHemisphere
 *   GpuPathTracer/cudaUtils.h:221-236   intersectAllSpeheres
 *   GpuPathTracer/cudaUtils.h:256-460   intersectBVHandTriangles     (while-while)
 *   GpuPathTracer/CommomStructs.hpp:18-39  Sphere::intersect / getNormal
 *   GpuPathTracer/utilfun.cpp:380-389   uf::hash
 * The arrays it consumes are in the reference's own CudaBVH "Compact" layout
 * (GpuPathTracer/CudaBVH.cpp:121-270): 64-byte nodes with BYTE-offset child links,
 * 48-byte v0/v1/v2 records, 0x80000000 leaf terminators, parallel index array.
 *
 * PARITY STATUS — "parity unpinned" for the random streams and therefore for rendered
 * radiance: the reference draws from cuRAND XORWOW (CUDA toolkit >= 7.5, unversioned;
 * tracer.cu:18-19,362-363), a closed third-party generator absent from /root/reference,
 * and the reference holds no golden images, known-answer vectors or tests for this path
 * (SURVEY.md §4, §8c).  tracer.cu itself cannot be compiled here (nvcc, cuRAND, inline
 * PTX).  What IS pinned: (1) the geometric core (closest-hit distance per ray) against the
 * reference's own CPU intersector (CpuRayTracer/src/triangle.hpp:49-72 + kdtree.cpp),
 * compiled from the reference sources by oracle/Makefile into oracle/_ref and run in the
 * dev container (tests/golden/ref_primary_hits.*); (2) this file's BVH walk against its
 * own brute-force loop over the raw triangles (the reference's dead intersectAllTriangles,
 * cudaUtils.h:194-217, is the model for that loop).
 *
 * Numerical contract shared with the HIP kernels: IEEE binary32,
 * no implicit contraction (-ffp-contract=off), fused multiply-adds only where fmaf is
 * written, correctly rounded / and sqrtf, and the polynomial sincos2pi / pow below in
 * place of libm's sinf/cosf/powf (glibc and ocml differ in the last ulp; a path that
 * flips at a silhouette diverges completely).  With that contract the HIP BVH2 kernel
 * reproduces this file bit for bit.
 *
 * Deliberate, documented departures from the literal reference:
 *   - RNG: counter-based hash keyed by (uf::hash(frame) + GLOBAL pixel index, draw
 *     number) instead of XORWOW keyed by block/thread ids (tracer.cu:363) — needed so
 *     the image does not depend on block shape or on which GPU owns a tile.
 *   - exact-t ties between DIFFERENT triangles go to the smaller triangle id instead of
 *     "first in traversal order" (cudaUtils.h:428 uses a strict <), so the winner does
 *     not depend on tree shape.
 *   - METAL: `w1*cosTheta` by default; the literal `float(width)*cosTheta` of
 *     tracer.cu:280 behind PT_FLAG_METAL_LITERAL_W.
 *   - sin/cos are evaluated on the turn fraction u directly (cos(2*pi*u)), not on a
 *     rounded phi = 2*M_PI*u.
 * Every other quirk is kept: nl = n for triangles (tracer.cu:126-127), miss returns the
 * unmasked background (:140-142), two discarded DIFF draws (:159-160), sin-theta-uniform
 * DIFF lobe without pdf (cudaUtils.h:185-192), 0.2 reflect probability (:239), R0
 * precedence (:230), +nl offset on transmission (:253), per-frame clamp (:390),
 * truncating 8-bit pack (:394-398).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ptmi.h"
#include "pt_oracle.h"

#define F32_MAX 3.402823466e+38f
#define ENTRY_SENTINEL 0x76543210 /* cudaUtils.h:21 */

typedef struct { float x, y, z; } v3;

/* ---------------------------------------------------------------- vector helpers */
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
/* a*s + b, one fused op per component */
static inline v3 vmadd(v3 a, float s, v3 b) { return V(fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z)); }
static inline float vdot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 vcross(v3 a, v3 b) {
    return V(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
/* glm::normalize = v * inversesqrt(dot(v,v)), inversesqrt = 1/sqrt (deps/glm) */
static inline v3 vnormalize(v3 a) { return vscale(a, 1.0f / sqrtf(vdot(a, a))); }

/* ---------------------------------------------------------------- RNG */
/* uf::hash, GpuPathTracer/utilfun.cpp:380-389 (Thomas Wang's 64-bit mix) */
uint64_t orc_wang64(uint64_t key) {
    key = (~key) + (key << 21);
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}

static inline uint32_t fmix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

typedef struct { uint32_t s0, s1, n; } rng_t;

/* seed = hash(frame) + <linear id>   (tracer.cu:363; id = global pixel index here) */
static inline rng_t rng_init(uint64_t frame, uint64_t pixel) {
    uint64_t z = orc_wang64(frame) + pixel;
    z ^= z >> 33; z *= 0xff51afd7ed558ccdULL; z ^= z >> 33; z *= 0xc4ceb9fe1a85ec53ULL; z ^= z >> 33;
    rng_t r = {(uint32_t)z, (uint32_t)(z >> 32), 0};
    return r;
}
/* uniform in (0,1] like curand_uniform */
static inline float rng_next(rng_t* r) {
    uint32_t x = fmix32(r->s0 + r->n * 0x9E3779B9u);
    x = fmix32(x ^ r->s1);
    r->n++;
    return (float)((x >> 8) + 1u) * 5.9604644775390625e-8f; /* 2^-24 */
}

float orc_rng_draw(uint64_t frame, uint64_t pixel, uint32_t draw) {
    rng_t r = rng_init(frame, pixel);
    r.n = draw;
    return rng_next(&r);
}

/* ---------------------------------------------------------------- math spec */
/* (cos, sin) of 2*pi*u, u in [0,1].  Quadrant reduction is exact (u is a multiple of
 * 2^-24), then Taylor polynomials on |theta| <= pi/4 evaluated with fmaf in Horner form. */
void orc_sincos2pi(float u, float* c_out, float* s_out) {
    int k = (int)fmaf(u, 4.0f, 0.5f);
    float r = fmaf((float)k, -0.25f, u);
    float th = r * 6.28318530717958647692f;
    float t2 = th * th;
    float sp = fmaf(t2, 2.75573192239858906526e-6f, -1.98412698412698412698e-4f);
    sp = fmaf(sp, t2, 8.33333333333333333333e-3f);
    sp = fmaf(sp, t2, -1.66666666666666666667e-1f);
    float s = fmaf(th * t2, sp, th);
    float cp = fmaf(t2, 2.48015873015873015873e-5f, -1.38888888888888888889e-3f);
    cp = fmaf(cp, t2, 4.16666666666666666667e-2f);
    cp = fmaf(cp, t2, -0.5f);
    float c = fmaf(t2, cp, 1.0f);
    switch (k & 3) {
        case 0: *c_out = c; *s_out = s; break;
        case 1: *c_out = -s; *s_out = c; break;
        case 2: *c_out = -c; *s_out = -s; break;
        default: *c_out = s; *s_out = -c; break;
    }
}

/* x^y for x in {0} U [2^-126,1], y > 0: exp2(y*log2 x) with polynomial log2/exp2. */
float orc_pow01(float x, float y) {
    if (!(x > 0.0f)) return 0.0f;
    uint32_t ix;
    memcpy(&ix, &x, 4);
    int e = (int)(ix >> 23) - 127;
    uint32_t im = (ix & 0x007FFFFFu) | 0x3F800000u;
    float m;
    memcpy(&m, &im, 4);
    /* domain: x = 0 or a normal binary32 <= 1 (callers pass 1 - k*2^-24) */
    if (m > 1.41421356237f) { m *= 0.5f; e += 1; }
    float s = (m - 1.0f) / (m + 1.0f);
    float s2 = s * s;
    float p = fmaf(s2, 0.111111111111f, 0.142857142857f);
    p = fmaf(p, s2, 0.2f);
    p = fmaf(p, s2, 0.333333333333f);
    p = fmaf(p, s2, 1.0f);
    float l2 = fmaf(s * p, 2.88539008177792681472f, (float)e); /* 2/ln2 */
    float q = y * l2;
    if (q < -126.0f) return 0.0f;
    float qi = floorf(q + 0.5f);
    float f = (q - qi) * 0.693147180559945309417f;
    float ep = fmaf(f, 1.98412698412698412698e-4f, 1.38888888888888888889e-3f);
    ep = fmaf(ep, f, 8.33333333333333333333e-3f);
    ep = fmaf(ep, f, 4.16666666666666666667e-2f);
    ep = fmaf(ep, f, 1.66666666666666666667e-1f);
    ep = fmaf(ep, f, 0.5f);
    ep = fmaf(ep, f, 1.0f);
    ep = fmaf(ep, f, 1.0f);
    int qe = (int)qi;
    uint32_t sc = (uint32_t)(qe + 127) << 23;
    float scale;
    memcpy(&scale, &sc, 4);
    return ep * scale;
}

/* ---------------------------------------------------------------- ray / triangle */
/* intersectRayTriangleEdge, cudaUtils.h:135-172.  Returns t or F32_MAX. */
static inline float mt_intersect(v3 v0, v3 e1, v3 e2, v3 o, v3 d, int cull) {
    const float EPS = 0.00001f;
    v3 tvec = vsub(o, v0);
    v3 pvec = vcross(d, e2);
    float det = vdot(e1, pvec);
    float invdet = 1.0f / det;
    float u = vdot(tvec, pvec) * invdet;
    v3 qvec = vcross(tvec, e1);
    float v = vdot(d, qvec) * invdet;
    if (det < -EPS) {
        if (cull) return F32_MAX;
    } else if (det < EPS) {
        return F32_MAX;
    }
    if (u < 0.0f || u > 1.0f) return F32_MAX;
    if (v < 0.0f || (u + v) > 1.0f) return F32_MAX;
    float t = vdot(e2, qvec) * invdet;
    if (t > 0.0f && t < F32_MAX) return t; /* rayMin = 0, rayMax = F32_MAX (tracer.cu:94) */
    return F32_MAX;
}

typedef struct {
    float t;      /* F32_MAX on miss */
    int tri;      /* original triangle id (after index remap), -1 on miss */
    v3 n;         /* cross(v0-v1, v0-v2) of the winner */
} hit_t;

static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static inline float min3(float a, float b, float c) { return fminf(fminf(a, b), c); }
static inline float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

/* intersectBVHandTriangles, cudaUtils.h:256-460, one lane (the ballot at :383-394 only
 * changes WHEN a postponed leaf is processed, not which leaves are tested in which
 * order, so the result is lane-independent). */
static hit_t bvh_intersect(const float* nodes, const float* tris, const int32_t* tidx,
                           v3 o, v3 d, int cull, orc_counters* cnt) {
    int stack[ORC_STACK_SIZE];
    const float ooeps = 8.271806125530277e-25f; /* exp2f(-80), cudaUtils.h:283 */
    float idx_ = 1.0f / (fabsf(d.x) > ooeps ? d.x : copysignf(ooeps, d.x));
    float idy_ = 1.0f / (fabsf(d.y) > ooeps ? d.y : copysignf(ooeps, d.y));
    float idz_ = 1.0f / (fabsf(d.z) > ooeps ? d.z : copysignf(ooeps, d.z));
    float oodx = o.x * idx_, oody = o.y * idy_, oodz = o.z * idz_;
    const float tmin = 0.0f;
    int sp = 0;
    stack[0] = ENTRY_SENTINEL;
    int leafAddr = 0, nodeAddr = 0;
    int hitIndex = -1, hitTri = -1;
    float hitT = F32_MAX;
    v3 hitN = V(0, 0, 0);
    uint64_t n_inner = 0, n_tri = 0, n_leaf = 0;

    while (nodeAddr != ENTRY_SENTINEL) {
        while (nodeAddr >= 0 && nodeAddr != ENTRY_SENTINEL) {
            const float* p = (const float*)((const char*)nodes + nodeAddr);
            n_inner++;
            float c0lox = fmaf(p[0], idx_, -oodx), c0hix = fmaf(p[1], idx_, -oodx);
            float c0loy = fmaf(p[2], idy_, -oody), c0hiy = fmaf(p[3], idy_, -oody);
            float c1lox = fmaf(p[4], idx_, -oodx), c1hix = fmaf(p[5], idx_, -oodx);
            float c1loy = fmaf(p[6], idy_, -oody), c1hiy = fmaf(p[7], idy_, -oody);
            float c0loz = fmaf(p[8], idz_, -oodz), c0hiz = fmaf(p[9], idz_, -oodz);
            float c1loz = fmaf(p[10], idz_, -oodz), c1hiz = fmaf(p[11], idz_, -oodz);
            /* spanBegin/EndKepler, cudaUtils.h:249-250 */
            float c0min = fmaxf(max3(fminf(c0lox, c0hix), fminf(c0loy, c0hiy), fminf(c0loz, c0hiz)), tmin);
            float c0max = fminf(min3(fmaxf(c0lox, c0hix), fmaxf(c0loy, c0hiy), fmaxf(c0loz, c0hiz)), hitT);
            float c1min = fmaxf(max3(fminf(c1lox, c1hix), fminf(c1loy, c1hiy), fminf(c1loz, c1hiz)), tmin);
            float c1max = fminf(min3(fmaxf(c1lox, c1hix), fmaxf(c1loy, c1hiy), fmaxf(c1loz, c1hiz)), hitT);
            int t0 = (c0min <= c0max) && (c0min >= tmin) && (c0min <= F32_MAX);
            int t1 = (c1min <= c1max) && (c1min >= tmin) && (c1min <= F32_MAX);
            if (!t0 && !t1) {
                nodeAddr = stack[sp--];
            } else {
                int cx = (int)f2bits(p[12]), cy = (int)f2bits(p[13]);
                nodeAddr = t0 ? cx : cy;
                if (t0 && t1) {
                    if (c1min < c0min) { int tmp = nodeAddr; nodeAddr = cy; cy = tmp; }
                    stack[++sp] = cy;
                }
            }
            if (nodeAddr < 0 && leafAddr >= 0) { /* first leaf: postpone */
                leafAddr = nodeAddr;
                nodeAddr = stack[sp--];
            }
            if (!(leafAddr >= 0)) break; /* single lane: ballot(leafAddr>=0)==0 */
        }
        while (leafAddr < 0) {
            n_leaf++;
            for (int triAddr = ~leafAddr;; triAddr += 3) {
                const float* r = tris + (size_t)triAddr * 4;
                if (f2bits(r[0]) == 0x80000000u) break;
                n_tri++;
                v3 v0 = V(r[0], r[1], r[2]), v1 = V(r[4], r[5], r[6]), v2 = V(r[8], r[9], r[10]);
                float t = mt_intersect(v0, vsub(v1, v0), vsub(v2, v0), o, d, cull);
                int id = tidx[triAddr];
                if (t > tmin && (t < hitT || (t == hitT && hitIndex != -1 && id < hitTri))) {
                    hitIndex = triAddr;
                    hitTri = id;
                    hitT = t;
                    hitN = vcross(vsub(v0, v1), vsub(v0, v2));
                }
            }
            leafAddr = nodeAddr;
            if (nodeAddr < 0) nodeAddr = stack[sp--];
        }
    }
    if (cnt) {
        cnt->rays++; cnt->inner += n_inner; cnt->tris += n_tri; cnt->leaves += n_leaf;
        cnt->hits += (hitIndex != -1);
    }
    hit_t h = {hitT, hitTri, hitN};
    return h;
}

/* brute force over the raw triangle soup (model: the reference's dead
 * intersectAllTriangles, cudaUtils.h:194-217, but with the same accept rule as above). */
static hit_t brute_intersect(const float* verts, const int32_t* tri_vidx, size_t n_tris,
                             v3 o, v3 d, int cull) {
    hit_t h = {F32_MAX, -1, {0, 0, 0}};
    for (size_t i = 0; i < n_tris; i++) {
        const float* a = verts + 3 * (size_t)tri_vidx[3 * i];
        const float* b = verts + 3 * (size_t)tri_vidx[3 * i + 1];
        const float* c = verts + 3 * (size_t)tri_vidx[3 * i + 2];
        v3 v0 = V(a[0], a[1], a[2]), v1 = V(b[0], b[1], b[2]), v2 = V(c[0], c[1], c[2]);
        float t = mt_intersect(v0, vsub(v1, v0), vsub(v2, v0), o, d, cull);
        if (t > 0.0f && t < h.t) { /* ascending ids: ties already go to the smaller id */
            h.t = t; h.tri = (int)i; h.n = vcross(vsub(v0, v1), vsub(v0, v2));
        }
    }
    return h;
}

void orc_trace_rays_bvh(const float* nodes, const float* tris, const int32_t* tidx,
                        const float* rays, size_t n_rays, int cull,
                        float* t_out, int32_t* tri_out, float* n_out, orc_counters* cnt) {
    orc_counters local;
    memset(&local, 0, sizeof local);
#pragma omp parallel
    {
        orc_counters c;
        memset(&c, 0, sizeof c);
#pragma omp for schedule(dynamic, 256)
        for (long i = 0; i < (long)n_rays; i++) {
            const float* r = rays + 8 * i;
            hit_t h = bvh_intersect(nodes, tris, tidx, V(r[0], r[1], r[2]), V(r[4], r[5], r[6]), cull, &c);
            t_out[i] = h.t; tri_out[i] = h.tri;
            if (n_out) { n_out[3 * i] = h.n.x; n_out[3 * i + 1] = h.n.y; n_out[3 * i + 2] = h.n.z; }
        }
#pragma omp critical
        { local.rays += c.rays; local.inner += c.inner; local.tris += c.tris; local.leaves += c.leaves; local.hits += c.hits; }
    }
    if (cnt) *cnt = local;
}

void orc_trace_rays_brute(const float* verts, const int32_t* tri_vidx, size_t n_tris,
                          const float* rays, size_t n_rays, int cull,
                          float* t_out, int32_t* tri_out, float* n_out) {
#pragma omp parallel for schedule(dynamic, 16)
    for (long i = 0; i < (long)n_rays; i++) {
        const float* r = rays + 8 * i;
        hit_t h = brute_intersect(verts, tri_vidx, n_tris, V(r[0], r[1], r[2]), V(r[4], r[5], r[6]), cull);
        t_out[i] = h.t; tri_out[i] = h.tri;
        if (n_out) { n_out[3 * i] = h.n.x; n_out[3 * i + 1] = h.n.y; n_out[3 * i + 2] = h.n.z; }
    }
}

/* ---------------------------------------------------------------- spheres */
/* Sphere::intersect, CommomStructs.hpp:23-31 */
static inline float sphere_intersect(const pt_sphere* s, v3 o, v3 d) {
    v3 op = vsub(V(s->pos_rad[0], s->pos_rad[1], s->pos_rad[2]), o);
    const float eps = 0.01f;
    float b = vdot(op, d);
    float disc = (b * b - vdot(op, op)) + s->pos_rad[3] * s->pos_rad[3];
    if (disc < 0) return 0;
    disc = sqrtf(disc);
    float t = b - disc;
    if (t > eps) return t;
    t = b + disc;
    return t > eps ? t : 0;
}

/* ---------------------------------------------------------------- camera */
/* getCamRayDir, cudaUtils.h:111-134.  Camera ray starts ON the image plane. */
void orc_camera_ray(const pt_camera* cam, int px, int py, int w, int h, float u0, float u1,
                    float* o_out, float* d_out) {
    float jx = u0 - 0.5f, jy = u1 - 0.5f;
    float xs = ((((float)px - (float)w / 2.0f) + 0.5f) + jx) * cam->dist * cam->aspect * cam->fov / (float)(w - 1);
    float ys = ((((float)py - (float)h / 2.0f) + 0.5f) + jy) * cam->dist * cam->fov / (float)(h - 1);
    v3 front = V(cam->front[0], cam->front[1], cam->front[2]);
    v3 right = V(cam->right[0], cam->right[1], cam->right[2]);
    v3 up = V(cam->up[0], cam->up[1], cam->up[2]);
    v3 dir = vmadd(up, ys, vmadd(right, xs, vscale(front, cam->dist)));
    v3 org = vadd(V(cam->pos[0], cam->pos[1], cam->pos[2]), dir);
    v3 dn = vnormalize(dir);
    o_out[0] = org.x; o_out[1] = org.y; o_out[2] = org.z;
    d_out[0] = dn.x; d_out[1] = dn.y; d_out[2] = dn.z;
}

/* ---------------------------------------------------------------- one sample */
/* getSample, tracer.cu:27-339 */
/* PT_FLAG_NEE over emissive triangles (ptmi.h): the light list the product derives from the material table —
 * every triangle whose row emits, ascending original id, as (v0, emi.r) (e1 = v1 - v0, emi.g) (e2 = v2 - v0, emi.b),
 * binary32 subtraction as at upload.  The tests build it from the mesh and hand it over before a render. */
/* The ARBITER (orc_sample_pixels): with a raw mesh set, get_sample's closest hit is the brute-force loop over every
 * triangle instead of the BVH walk, and each segment's (t, triangle) can be logged.  Thread-local: set per worker. */
static _Thread_local const float* tl_brute_verts = 0;
static _Thread_local const int32_t* tl_brute_vidx = 0;
static _Thread_local size_t tl_brute_n = 0;
static _Thread_local float* tl_seg_log = 0;      /* [cap][2]: t, triangle id (as float bits) per segment */
static _Thread_local uint32_t tl_seg_cap = 0;

static const float* g_tri_lights = 0;
static size_t g_n_tri_lights = 0;
void orc_set_tri_lights(const float* lights12, size_t n) {
    g_tri_lights = n ? lights12 : 0;
    g_n_tri_lights = lights12 ? n : 0;
}

static v3 get_sample(const float* nodes, const float* tris, const int32_t* tidx,
                     const pt_sphere* sph, size_t n_sph, const pt_camera* cam,
                     const pt_params* P, const pt_material* mtab, const int32_t* tri_mat,
                     int px, int py, rng_t* rng, orc_counters* cnt) {
    float o_[3], d_[3];
    float u0 = rng_next(rng), u1 = rng_next(rng);
    orc_camera_ray(cam, px, py, P->width, P->height, u0, u1, o_, d_);
    v3 o = V(o_[0], o_[1], o_[2]), d = V(d_[0], d_[1], d_[2]);
    v3 mask = V(1, 1, 1), accu = V(0, 0, 0);
    uint32_t nee_mask = 0; /* PT_FLAG_NEE: spheres whose light the previous bounce already sampled */
    v3 tricol = V(P->tri_col[0], P->tri_col[1], P->tri_col[2]);
    v3 triemi = V(P->tri_emi[0], P->tri_emi[1], P->tri_emi[2]);

    for (uint32_t depth = 0; depth < P->depth; ++depth) {
        int geom = 3; /* GeoType::NONE */
        int sph_id = -1;
        hit_t h = {F32_MAX, -1, {0, 0, 0}};
        if (tl_brute_n) { h = brute_intersect(tl_brute_verts, tl_brute_vidx, tl_brute_n, o, d, P->cull_backfaces); if (cnt) cnt->rays++; }
        else if (nodes) h = bvh_intersect(nodes, tris, tidx, o, d, P->cull_backfaces, cnt);
        else if (cnt) cnt->rays++;
        if (tl_seg_log && depth < tl_seg_cap) { tl_seg_log[2 * depth] = h.t; tl_seg_log[2 * depth + 1] = bits2f((uint32_t)h.tri); }
        float scene_t = h.t;
        if (h.tri != -1) geom = 0; /* TRI */
        /* intersectAllSpeheres, cudaUtils.h:221-236 */
        for (size_t i = 0; i < n_sph; i++) {
            float ts = sphere_intersect(&sph[i], o, d);
            if (ts != 0.0f && ts < scene_t && ts > 0.01f) { scene_t = ts; sph_id = (int)i; geom = 1; }
        }
        v3 hitpos = vmadd(d, scene_t, o);
        v3 n, nl, objcol, emit;
        int mat;
        float phong = P->phong_expo;
        if (geom == 1) {
            const pt_sphere* s = &sph[sph_id];
            n = vnormalize(vsub(hitpos, V(s->pos_rad[0], s->pos_rad[1], s->pos_rad[2])));
            nl = vdot(n, d) < 0 ? n : vscale(n, -1.0f);
            objcol = V(s->col[0], s->col[1], s->col[2]);
            emit = V(s->emi[0], s->emi[1], s->emi[2]);
            mat = s->mat;
        } else if (geom == 0) {
            n = vnormalize(h.n);
            nl = n; /* tracer.cu:126-127: the flip is a discarded expression */
            if ((P->flags & PT_FLAG_FACE_FORWARD) && !(vdot(n, d) < 0)) nl = vscale(n, -1.0f); /* extension */
            objcol = tricol; emit = triemi; mat = P->tri_mat;
            if (mtab) { /* extension (ptmi.h pt_upload_tri_materials): per-triangle material row */
                const pt_material* m = &mtab[tri_mat[h.tri]];
                objcol = V(m->col[0], m->col[1], m->col[2]);
                emit = V(m->emi[0], m->emi[1], m->emi[2]);
                mat = m->mat;
                phong = m->phong_expo;
            }
        } else {
            const v3 bk = V(P->bk_color[0], P->bk_color[1], P->bk_color[2]);
            if (P->flags & PT_FLAG_MISS_KEEPS_PATH) return vadd(accu, vmul(mask, bk)); /* extension (ptmi.h) */
            return bk; /* tracer.cu:140-142 */
        }
        /* not already gathered by a shadow ray: bits 0-7 the spheres that were eligible, bit 8 the emissive triangles */
        if (!(geom == 1 && sph_id < 8 && ((nee_mask >> sph_id) & 1u)) && !(geom == 0 && (nee_mask & 0x100u)))
            accu = vadd(accu, vmul(mask, emit));
        nee_mask = 0;

        if ((P->flags & PT_FLAG_RUSSIAN_ROULETTE) && depth >= 2) { /* extension (ptmi.h) */
            float pr = fmaxf(objcol.x, fmaxf(objcol.y, objcol.z));
            if (!(rng_next(rng) < pr)) return accu;
            objcol = vscale(objcol, 1.0f / pr);
        }

        if ((P->flags & PT_FLAG_RR_CPU_TRACER) && depth >= 5) { /* extension: CpuRayTracer/src/scene.cpp:38-47 */
            float pr = fmaxf(objcol.x, fmaxf(objcol.y, objcol.z));
            if (!(rng_next(rng) < pr * 0.9f)) return accu;
            objcol = vscale(objcol, 0.9f / pr);
        }

        v3 nextdir;
        if (mat == PT_MAT_DIFF) { /* tracer.cu:156-186 */
            if (!(P->flags & PT_FLAG_COSINE_DIFF)) {
                (void)rng_next(rng); (void)rng_next(rng); /* phi, r2: drawn, unused (:159-161) */
            }
            v3 nt = fabsf(nl.x) > fabsf(nl.y) ? V(nl.z, 0, -nl.x) : V(0, -nl.z, nl.y);
            nt = vnormalize(nt);
            v3 nb = vnormalize(vcross(nl, nt));
            float f1 = rng_next(rng), f2 = rng_next(rng);
            float c, s;
            orc_sincos2pi(f1, &c, &s);
            v3 rv;
            if (P->flags & PT_FLAG_COSINE_DIFF) { /* extension: cosine-weighted, pdf = cos/pi */
                float r2s = sqrtf(f2);
                rv = V(c * r2s, sqrtf(1.0f - f2), s * r2s);
            } else {
                rv = V(c * f2, sqrtf(1.0f - f2 * f2), s * f2); /* cudaUtils.h:185-192 */
            }
            nextdir = vmadd(nt, rv.z, vmadd(nl, rv.y, vscale(nb, rv.x)));
            nextdir = vnormalize(nextdir);
            hitpos = vmadd(nl, 0.001f, hitpos);
            mask = vmul(mask, objcol);
            if ((P->flags & PT_FLAG_NEE) && (P->flags & PT_FLAG_COSINE_DIFF)) { /* extension (ptmi.h) */
                /* lights = emissive spheres (of the first 8) the point is outside of */
                uint32_t el = 0;
                int n_el = 0;
                for (size_t i = 0; i < n_sph && i < 8; i++) {
                    const pt_sphere* s = &sph[i];
                    if (s->emi[0] == 0.0f && s->emi[1] == 0.0f && s->emi[2] == 0.0f) continue;
                    v3 w = vsub(V(s->pos_rad[0], s->pos_rad[1], s->pos_rad[2]), hitpos);
                    if (vdot(w, w) > (s->pos_rad[3] * s->pos_rad[3]) * 1.001f) { el |= 1u << i; n_el++; }
                }
                const int n_tl = mtab ? (int)g_n_tri_lights : 0, n_all = n_el + n_tl;
                nee_mask = el | (n_tl > 0 ? 0x100u : 0u);
                if (n_all > 0) {
                    float u0 = rng_next(rng), u1 = rng_next(rng), u2 = rng_next(rng);
                    int pick = (int)(u0 * (float)n_all);
                    if (pick > n_all - 1) pick = n_all - 1;
                    if (pick >= n_el) { /* an emissive triangle (orc_set_tri_lights): uniform point; the faces a path can hit emit */
                        const float* tl = g_tri_lights + 12 * (size_t)(pick - n_el);
                        v3 e1 = V(tl[4], tl[5], tl[6]), e2 = V(tl[8], tl[9], tl[10]);
                        float su = sqrtf(u1);
                        v3 pl = vmadd(e2, u2 * su, vmadd(e1, 1.0f - su, V(tl[0], tl[1], tl[2])));
                        v3 w = vsub(pl, hitpos);
                        float d2 = vdot(w, w);
                        float dist = sqrtf(d2);
                        v3 l = vscale(w, 1.0f / dist);
                        float cosl = vdot(nl, l);
                        float sdot = vdot(vcross(e1, e2), l);
                        float proj = fabsf(sdot); /* 2 * area * cos(light) */
                        float t_light = dist * 0.999f;
                        /* with back-face culling a path only ever hits a triangle's front (det > 0 <=> cross(e1,e2).d < 0) */
                        int blocked = !(cosl > 0.0f) || !(d2 > 0.0f) || !(proj > 0.0f) || (P->cull_backfaces != 0 && !(sdot < 0.0f));
                        for (size_t j = 0; j < n_sph && !blocked; j++) {
                            float ts = sphere_intersect(&sph[j], hitpos, l);
                            if (ts != 0.0f && ts < t_light && ts > 0.01f) blocked = 1;
                        }
                        if (!blocked) {
                            hit_t h2 = {F32_MAX, -1, {0, 0, 0}};
                            if (nodes) h2 = bvh_intersect(nodes, tris, tidx, hitpos, l, P->cull_backfaces, cnt);
                            else if (cnt) cnt->rays++;
                            if (!(h2.t < t_light)) {
                                float k = ((cosl * (0.5f * proj)) * (float)n_all) / (3.14159274f * d2);
                                accu = vadd(accu, vscale(vmul(mask, V(tl[3], tl[7], tl[11])), k));
                            }
                        }
                    } else {
                    int li = 0;
                    for (int i = 0, k = 0; i < 8; i++)
                        if ((el >> i) & 1u) { if (k == pick) { li = i; break; } k++; }
                    const pt_sphere* L = &sph[li];
                    v3 w = vsub(V(L->pos_rad[0], L->pos_rad[1], L->pos_rad[2]), hitpos);
                    float d2 = vdot(w, w), r2 = L->pos_rad[3] * L->pos_rad[3];
                    v3 wn = vscale(w, 1.0f / sqrtf(d2));
                    float cos_max = sqrtf(fmaxf(0.0f, 1.0f - r2 / d2));
                    float cos_t = 1.0f - u1 * (1.0f - cos_max);
                    float sin_t = sqrtf(fmaxf(0.0f, 1.0f - cos_t * cos_t));
                    float cp, sp;
                    orc_sincos2pi(u2, &cp, &sp);
                    v3 t1 = fabsf(wn.x) > fabsf(wn.y) ? V(wn.z, 0, -wn.x) : V(0, -wn.z, wn.y);
                    t1 = vnormalize(t1);
                    v3 b1 = vnormalize(vcross(wn, t1));
                    v3 l = vnormalize(vmadd(t1, sp * sin_t, vmadd(wn, cos_t, vscale(b1, cp * sin_t))));
                    float cosl = vdot(nl, l);
                    float t_light = sphere_intersect(L, hitpos, l);
                    int blocked = !(cosl > 0.0f) || t_light == 0.0f;
                    for (size_t j = 0; j < n_sph && !blocked; j++) {
                        if ((int)j == li) continue;
                        float ts = sphere_intersect(&sph[j], hitpos, l);
                        if (ts != 0.0f && ts < t_light && ts > 0.01f) blocked = 1;
                    }
                    if (!blocked) {
                        hit_t h2 = {F32_MAX, -1, {0, 0, 0}};
                        if (nodes) h2 = bvh_intersect(nodes, tris, tidx, hitpos, l, P->cull_backfaces, cnt);
                        else if (cnt) cnt->rays++;
                        if (!(h2.t < t_light)) {
                            float k = (cosl * (2.0f * (1.0f - cos_max))) * (float)n_all;
                            accu = vadd(accu, vscale(vmul(mask, V(L->emi[0], L->emi[1], L->emi[2])), k));
                        }
                    }
                    }
                }
            }
        } else if (mat == PT_MAT_SPEC) { /* :190-203 */
            nextdir = vnormalize(vmadd(nl, -2.0f * vdot(nl, d), d));
            hitpos = vmadd(nl, 0.001f, hitpos);
            mask = vmul(mask, objcol);
        } else if (mat == PT_MAT_REFR) { /* :205-256 */
            int into = vdot(n, nl) > 0;
            float nc = P->air_ior, ntt = P->glass_ior;
            float nnt = into ? nc / ntt : ntt / nc;
            float ddn = vdot(d, nl);
            float cos2t = 1.0f - nnt * nnt * (1.0f - ddn * ddn);
            if (cos2t < 0.0f) {
                nextdir = vnormalize(vmadd(n, -2.0f * vdot(n, d), d));
                hitpos = vmadd(nl, 0.001f, hitpos);
            } else {
                float k = (into ? 1.0f : -1.0f) * (ddn * nnt + sqrtf(cos2t));
                v3 tdir = vnormalize(vmadd(n, -k, vscale(d, nnt)));
                int fix = (P->flags & PT_FLAG_GLASS_FIX) != 0; /* extension (ptmi.h) */
                float R0 = fix ? ((ntt - nc) * (ntt - nc)) / ((ntt + nc) * (ntt + nc))
                               : (ntt - nc) * (ntt - nc) / (ntt + nc) * (ntt + nc); /* sic, :230 */
                float c = 1.0f - (into ? -ddn : vdot(tdir, n));
                float Re = R0 + (1.0f - R0) * c * c * c * c * c;
                float Tr = 1 - Re;
                float Pp = 0.25f + 0.5f * Re;
                float RP = Re / Pp, TP = Tr / (1.0f - Pp);
                int transmitted = 0;
                if (rng_next(rng) < (fix ? Pp : 0.2f)) { /* (double)u < 0.2  <=>  u < 0.2f for binary32 u */
                    mask = vscale(mask, RP);
                    nextdir = vnormalize(vmadd(n, -2.0f * vdot(n, d), d));
                } else {
                    mask = vscale(mask, TP);
                    nextdir = vnormalize(tdir);
                    transmitted = 1;
                }
                hitpos = vmadd(nl, (fix && transmitted) ? -0.001f : 0.001f, hitpos);
            }
        } else { /* METAL :257-293 */
            float f1 = rng_next(rng), r2 = rng_next(rng);
            float cphi, sphi;
            orc_sincos2pi(f1, &cphi, &sphi);
            float cosT = orc_pow01(1.0f - r2, 1.0f / (phong + 1.0f));
            float sinT = sqrtf(1.0f - cosT * cosT);
            v3 w1 = vnormalize(vmadd(nl, -2.0f * vdot(nl, d), d));
            v3 ax = ((double)fabsf(w1.x) > 0.1) ? V(0, 1, 0) : V(1, 0, 0);
            v3 uu = vnormalize(vcross(ax, w1));
            v3 vv = vcross(w1, uu);
            v3 base = vmadd(vv, sphi * sinT, vscale(uu, cphi * sinT));
            if (P->flags & PT_FLAG_METAL_LITERAL_W) {
                float wc = (float)P->width * cosT; /* tracer.cu:280 */
                nextdir = V(base.x + wc, base.y + wc, base.z + wc);
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

static inline float clamp01(float f) { return fmaxf(0.0f, fminf(f, 1.0f)); }

/* accumulate + pack, tracer.cu:386-398 and rgbToUint cudaUtils.h:99-105 */
void orc_accumulate(float* acc3, uint32_t* rgba, const float* sample3, uint64_t N) {
    float fm1 = (float)(N - 1), inv = 1.0f / (float)N;
    for (int c = 0; c < 3; c++) {
        float a = (N == 1) ? 0.0f : acc3[c] * fm1;
        a = a + sample3[c];
        a = a * inv;
        acc3[c] = clamp01(a);
    }
    if (rgba) {
        uint32_t r = (uint32_t)(unsigned char)(255.0f * acc3[0]);
        uint32_t g = (uint32_t)(unsigned char)(255.0f * acc3[1]);
        uint32_t b = (uint32_t)(unsigned char)(255.0f * acc3[2]);
        *rgba = (b << 16) | (g << 8) | r;
    }
}

/* trace<<<>>> for the whole frame (or the stripes this part owns), spp samples. */
int orc_render(float* accum, uint32_t* rgba,
               const float* nodes, const float* tris, const int32_t* tidx,
               const pt_sphere* sph, size_t n_sph,
               const pt_camera* cam, const pt_params* P, uint32_t spp, orc_counters* cnt) {
    return orc_render_mat(accum, rgba, nodes, tris, tidx, sph, n_sph, NULL, NULL, cam, P, spp, cnt);
}

/* The same frame with per-triangle materials (the extension of ptmi.h's
 * pt_upload_tri_materials): table row tri_mat[original triangle id]; mtab NULL = orc_render. */
int orc_render_mat(float* accum, uint32_t* rgba,
                   const float* nodes, const float* tris, const int32_t* tidx,
                   const pt_sphere* sph, size_t n_sph,
                   const pt_material* mtab, const int32_t* tri_mat,
                   const pt_camera* cam, const pt_params* P, uint32_t spp, orc_counters* cnt) {
    if (mtab && !tri_mat) return -1;
    if (!accum || !cam || !P || P->width <= 1 || P->height <= 1 || spp == 0) return -1;
    orc_counters total;
    memset(&total, 0, sizeof total);
    const int W = P->width, H = P->height;
    const int pc = P->part_count > 1 ? P->part_count : 1;
    const int pr = P->part_rows > 0 ? P->part_rows : 8;
#pragma omp parallel
    {
        orc_counters c;
        memset(&c, 0, sizeof c);
#pragma omp for schedule(dynamic, 1)
        for (int y = 0; y < H; y++) {
            if (pc > 1 && (y / pr) % pc != P->part_index) continue;
            for (int x = 0; x < W; x++) {
                uint64_t pix = (uint64_t)y * (uint64_t)W + (uint64_t)x;
                for (uint32_t s = 0; s < spp; s++) {
                    rng_t rng = rng_init(P->frame + s, pix);
                    v3 col = get_sample(nodes, tris, tidx, sph, n_sph, cam, P, mtab, tri_mat, x, y, &rng, &c);
                    c.paths++;
                    float sm[3] = {col.x, col.y, col.z};
                    orc_accumulate(accum + 3 * pix, rgba ? rgba + pix : NULL, sm, P->sample_index + s);
                }
            }
        }
#pragma omp critical
        {
            total.rays += c.rays; total.inner += c.inner; total.tris += c.tris;
            total.leaves += c.leaves; total.hits += c.hits; total.paths += c.paths;
        }
    }
    if (cnt) *cnt = total;
    return 0;
}

/* Selected pixels, sample by sample (the tests' arbiter for pixels where two renders differ): out_col[n][spp][3] = the
 * sample colours BEFORE the fold, out_seg[n][spp][depth][2] (may be NULL) = every segment's (t, triangle id bits; F32_MAX /
 * -1 where the path had ended).  Closest hits come from the BVH arrays, or — when nodes is NULL and a raw mesh is given —
 * from the brute-force loop over all its triangles (no tree: nothing can be culled by a box).  Global material only. */
int orc_sample_pixels(const float* nodes, const float* tris, const int32_t* tidx,
                      const float* verts, const int32_t* tri_vidx, size_t n_tris,
                      const pt_sphere* sph, size_t n_sph, const pt_camera* cam, const pt_params* P, uint32_t spp,
                      const int32_t* pixels_xy, size_t n, float* out_col, float* out_seg) {
    if (!cam || !P || !pixels_xy || !out_col || (!nodes && !(verts && tri_vidx && n_tris))) return -1;
    const long total = (long)n * (long)spp;
#pragma omp parallel for schedule(dynamic, 1)
    for (long k = 0; k < total; k++) {
        const size_t i = (size_t)(k / spp);
        const uint32_t s = (uint32_t)(k % spp);
        const int x = pixels_xy[2 * i], y = pixels_xy[2 * i + 1];
        const uint64_t pix = (uint64_t)y * (uint64_t)P->width + (uint64_t)x;
        float* seg = out_seg ? out_seg + 2 * (size_t)P->depth * (size_t)k : 0;
        if (seg) for (uint32_t d = 0; d < P->depth; d++) { seg[2 * d] = F32_MAX; seg[2 * d + 1] = bits2f(0xffffffffu); }
        if (!nodes) { tl_brute_verts = verts; tl_brute_vidx = tri_vidx; tl_brute_n = n_tris; }
        tl_seg_log = seg; tl_seg_cap = P->depth;
        rng_t rng = rng_init(P->frame + s, pix);
        const v3 col = get_sample(nodes, tris, tidx, sph, n_sph, cam, P, NULL, NULL, x, y, &rng, NULL);
        tl_brute_n = 0; tl_seg_log = 0; tl_seg_cap = 0;
        out_col[3 * k] = col.x; out_col[3 * k + 1] = col.y; out_col[3 * k + 2] = col.z;
    }
    return 0;
}

/* Primary rays of a frame (no jitter unless u0/u1 given) — used to feed ray-batch tests
 * and the CpuRayTracer cross-check with identical rays. */
void orc_primary_rays(const pt_camera* cam, int W, int H, uint64_t frame, int jitter, float* rays8) {
#pragma omp parallel for
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            uint64_t pix = (uint64_t)y * W + x;
            float u0 = 0.5f, u1 = 0.5f;
            if (jitter) { rng_t r = rng_init(frame, pix); u0 = rng_next(&r); u1 = rng_next(&r); }
            float* r8 = rays8 + 8 * pix;
            orc_camera_ray(cam, x, y, W, H, u0, u1, r8, r8 + 4);
            r8[3] = 0.0f; r8[7] = 0.0f;
        }
}

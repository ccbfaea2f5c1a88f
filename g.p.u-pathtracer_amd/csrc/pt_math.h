// pt_math.h — device-side arithmetic of the path loop (gfx950).
//
// Numerical contract: IEEE binary32, this translation unit is compiled
// with -ffp-contract=off, so a fused multiply-add happens exactly where fmaf() is
// written; '/' and sqrtf are the correctly rounded forms (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt); sin/cos/pow are the polynomials below, not
// ocml's.  Each routine names the reference code it implements.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_F32_MAX 3.402823466e+38f
#define PT_SENTINEL 0x76543210  // EntrypointSentinel, GpuPathTracer/cudaUtils.h:21

struct v3 {
    float x, y, z;
};

// two binary32 lanes for v_pk_fma_f32 (each half is an IEEE fused multiply-add)
typedef float pt_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pt_f2 pt_mk2(float a, float b) { pt_f2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ pt_f2 pt_fma2(pt_f2 a, pt_f2 b, pt_f2 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 vmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 vscale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
// a*s + b, fused per component
__device__ __forceinline__ v3 vmadd(v3 a, float s, v3 b) { return V3(fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z)); }
__device__ __forceinline__ float vdot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ v3 vcross(v3 a, v3 b) {
    return V3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
// glm::normalize = v * (1/sqrt(dot(v,v)))
__device__ __forceinline__ v3 vnormalize(v3 a) { return vscale(a, 1.0f / sqrtf(vdot(a, a))); }

// ---------------------------------------------------------------------------- streams
// Loads / stores of data that is read once and written once per stage (path records, hit records, sample colours): marked
// non-temporal (`nt`) so that the 2-4 GB a stage streams do not push the scene's items out of L2 / the Infinity Cache.
#ifndef PT_STREAM_NT
#define PT_STREAM_NT 1
#endif
typedef float pt_v4 __attribute__((ext_vector_type(4)));
typedef float pt_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 pt_sld4(const float4* p) {
#if PT_STREAM_NT
    const pt_v4 v = __builtin_nontemporal_load((const pt_v4*)p);
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void pt_sst4(float4* p, float4 a) {
#if PT_STREAM_NT
    const pt_v4 v = {a.x, a.y, a.z, a.w};
    __builtin_nontemporal_store(v, (pt_v4*)p);
#else
    *p = a;
#endif
}
__device__ __forceinline__ float2 pt_sld2(const float2* p) {
#if PT_STREAM_NT
    const pt_v2 v = __builtin_nontemporal_load((const pt_v2*)p);
    return make_float2(v.x, v.y);
#else
    return *p;
#endif
}
__device__ __forceinline__ void pt_sst2(float2* p, float2 a) {
#if PT_STREAM_NT
    const pt_v2 v = {a.x, a.y};
    __builtin_nontemporal_store(v, (pt_v2*)p);
#else
    *p = a;
#endif
}
__device__ __forceinline__ float pt_sld1(const float* p) {
#if PT_STREAM_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void pt_sst1(float* p, float a) {
#if PT_STREAM_NT
    __builtin_nontemporal_store(a, p);
#else
    *p = a;
#endif
}

// three floats (a sample colour: 12-byte stride) as ONE dwordx3 access
typedef float pt_v3u __attribute__((ext_vector_type(3), aligned(4)));
__device__ __forceinline__ v3 pt_sld3(const float* p) {
#if PT_STREAM_NT
    const pt_v3u v = __builtin_nontemporal_load((const pt_v3u*)p);
#else
    const pt_v3u v = *(const pt_v3u*)p;
#endif
    return V3(v.x, v.y, v.z);
}
__device__ __forceinline__ void pt_sst3(float* p, v3 a) {
    const pt_v3u v = {a.x, a.y, a.z};
#if PT_STREAM_NT
    __builtin_nontemporal_store(v, (pt_v3u*)p);
#else
    *(pt_v3u*)p = v;
#endif
}

// ---------------------------------------------------------------------------- RNG
// uf::hash, GpuPathTracer/utilfun.cpp:380-389
__device__ __forceinline__ uint64_t pt_wang64(uint64_t key) {
    key = (~key) + (key << 21);
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}

__device__ __forceinline__ uint32_t pt_fmix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

// Counter-based stream keyed by (uf::hash(frame) + GLOBAL pixel index): the reference
// seeds cuRAND with hash + block/thread-derived id (tracer.cu:362-363); keying by pixel
// makes the image independent of block shape and of which GPU owns the tile.
struct pt_rng {
    uint32_t s0, s1, n;
};
__device__ __forceinline__ pt_rng pt_rng_init(uint64_t frame_hash, uint64_t pixel) {
    uint64_t z = frame_hash + pixel;
    z ^= z >> 33; z *= 0xff51afd7ed558ccdULL; z ^= z >> 33; z *= 0xc4ceb9fe1a85ec53ULL; z ^= z >> 33;
    pt_rng r; r.s0 = (uint32_t)z; r.s1 = (uint32_t)(z >> 32); r.n = 0;
    return r;
}
// uniform in (0,1], like curand_uniform
__device__ __forceinline__ float pt_rng_next(pt_rng& r) {
    uint32_t x = pt_fmix32(r.s0 + r.n * 0x9E3779B9u);
    x = pt_fmix32(x ^ r.s1);
    r.n++;
    return (float)((x >> 8) + 1u) * 5.9604644775390625e-8f;
}

// ---------------------------------------------------------------------------- math
// (cos, sin)(2*pi*u), u in [0,1]: exact quadrant reduction + Taylor on |theta| <= pi/4.
__device__ __forceinline__ void pt_sincos2pi(float u, float& c_out, float& s_out) {
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
    int q = k & 3;
    float cc = (q & 1) ? s : c;   // |cos| source
    float ss = (q & 1) ? c : s;   // |sin| source
    c_out = (q == 1 || q == 2) ? -cc : cc;
    s_out = (q >= 2) ? -ss : ss;
}

// x^y for x in {0} U [2^-126,1], y > 0 (METAL lobe, tracer.cu:267)
__device__ __forceinline__ float pt_pow01(float x, float y) {
    if (!(x > 0.0f)) return 0.0f;
    uint32_t ix = __float_as_uint(x);
    int e = (int)(ix >> 23) - 127;
    float m = __uint_as_float((ix & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421356237f) { m *= 0.5f; e += 1; }
    float s = (m - 1.0f) / (m + 1.0f);
    float s2 = s * s;
    float p = fmaf(s2, 0.111111111111f, 0.142857142857f);
    p = fmaf(p, s2, 0.2f);
    p = fmaf(p, s2, 0.333333333333f);
    p = fmaf(p, s2, 1.0f);
    float l2 = fmaf(s * p, 2.88539008177792681472f, (float)e);
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
    return ep * __uint_as_float((uint32_t)(qe + 127) << 23);
}

// ---------------------------------------------------------------------------- primitives
// intersectRayTriangleEdge, GpuPathTracer/cudaUtils.h:135-172 (rayMin 0, rayMax F32_MAX)
__device__ __forceinline__ float pt_mt_intersect(v3 v0, v3 e1, v3 e2, v3 o, v3 d, bool cull) {
    const float EPS = 0.00001f;
    v3 tvec = vsub(o, v0);
    v3 pvec = vcross(d, e2);
    float det = vdot(e1, pvec);
    float invdet = 1.0f / det;
    float u = vdot(tvec, pvec) * invdet;
    v3 qvec = vcross(tvec, e1);
    float v = vdot(d, qvec) * invdet;
    float t = vdot(e2, qvec) * invdet;
    bool miss = (det < -EPS) ? cull : (det < EPS);
    miss = miss || (u < 0.0f) || (u > 1.0f) || (v < 0.0f) || ((u + v) > 1.0f);
    miss = miss || !(t > 0.0f && t < PT_F32_MAX);
    return miss ? PT_F32_MAX : t;
}

// Woop's ray/triangle test on the affine rows W = M^-1, M = columns (v0-v2, v1-v2, n, v2),
// n = (v0-v2)x(v1-v2), as the reference MEANT to build them (CudaBVH.cpp:274-305; its column 3 is
// wrong, SURVEY.md F4, and its kernel never reads them, F3):  rz = (W row 2, w negated),
// rx = W row 0, ry = W row 1.  Accept/cull rules are the Moller-Trumbore ones of
// cudaUtils.h:151-166 with det taken as -dot(d, N), N = cross(v0-v1, v0-v2) (4th record piece).
__device__ __forceinline__ float pt_woop_intersect(float4 rz, float4 rx, float4 ry, v3 N, v3 o, v3 d, bool cull) {
    const float EPS = 0.00001f;
    const float Oz = rz.w - vdot(o, V3(rz.x, rz.y, rz.z));
    const float Dz = vdot(d, V3(rz.x, rz.y, rz.z));
    const float t = Oz * (1.0f / Dz);
    const float u = fmaf(t, vdot(d, V3(rx.x, rx.y, rx.z)), rx.w + vdot(o, V3(rx.x, rx.y, rx.z)));
    const float v = fmaf(t, vdot(d, V3(ry.x, ry.y, ry.z)), ry.w + vdot(o, V3(ry.x, ry.y, ry.z)));
    const float det = -vdot(d, N);
    bool miss = (det < -EPS) ? cull : (det < EPS);
    miss = miss || (u < 0.0f) || (v < 0.0f) || ((u + v) > 1.0f);
    miss = miss || !(t > 0.0f && t < PT_F32_MAX);
    return miss ? PT_F32_MAX : t;
}

// Sphere, GpuPathTracer/CommomStructs.hpp:18-39 (44 bytes)
struct pt_sphere_d {
    float px, py, pz, rad;
    float emi[3];
    float col[3];
    int mat;
};

// Sphere::intersect, CommomStructs.hpp:23-31
__device__ __forceinline__ float pt_sphere_intersect(float px, float py, float pz, float rad, v3 o, v3 d) {
    v3 op = vsub(V3(px, py, pz), o);
    const float eps = 0.01f;
    float b = vdot(op, d);
    float disc = (b * b - vdot(op, op)) + rad * rad;
    if (disc < 0) return 0.0f;
    disc = sqrtf(disc);
    float t = b - disc;
    if (t > eps) return t;
    t = b + disc;
    return t > eps ? t : 0.0f;
}

__device__ __forceinline__ float pt_clamp01(float f) { return fmaxf(0.0f, fminf(f, 1.0f)); }

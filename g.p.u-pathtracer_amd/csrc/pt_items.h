// pt_items.h — encoders of the two 64-byte items the walks fetch, shared by the host re-layout
// (pt_scene_build.h: trees that arrive through pt_upload_bvh) and the device builder
// (pt_build.h: pt_build_bvh), so both produce bit-identical items from the same boxes / vertices.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct PtBox { float lo[3], hi[3]; };

__host__ __device__ inline float pt_i2f(int32_t i) { union { int32_t i; float f; } u; u.i = i; return u.f; }

// Triangle record {v0.xyz, id | e1.xyz, last | e2.xyz, 0 | N.xyz, 0}: the edge subtraction of
// cudaUtils.h:177-178 and the normal cross(v0-v1, v0-v2) of :432 (the kernels' vcross
// arithmetic: fma(a.y, b.z, -(a.z*b.y)) ...) are hoisted out of the walk; same IEEE results.
__host__ __device__ inline void pt_encode_record(const float* v0, const float* v1, const float* v2, int32_t id, int last, float* rec) {
    const float ax = v0[0] - v1[0], ay = v0[1] - v1[1], az = v0[2] - v1[2];
    const float bx = v0[0] - v2[0], by = v0[1] - v2[1], bz = v0[2] - v2[2];
    rec[0] = v0[0]; rec[1] = v0[1]; rec[2] = v0[2]; rec[3] = pt_i2f(id);
    rec[4] = v1[0] - v0[0]; rec[5] = v1[1] - v0[1]; rec[6] = v1[2] - v0[2]; rec[7] = pt_i2f(last);
    rec[8] = v2[0] - v0[0]; rec[9] = v2[1] - v0[1]; rec[10] = v2[2] - v0[2]; rec[11] = 0.f;
    rec[12] = fmaf(ay, bz, -(az * by)); rec[13] = fmaf(az, bx, -(ax * bz)); rec[14] = fmaf(ax, by, -(ay * bx)); rec[15] = 0.f;
}

// 4-wide node: origin = the children's common lower corner, one grid step per axis (ext / 255 and a hair: round 3; a power of
// two in rounds 1-2, which made the grid up to twice as coarse as it had to be),
// child planes as bytes rounded OUTWARD (checked with the walk's own fma(q, scale, origin)), so a
// decoded child box always contains the exact one:
//   d[0..2] origin   d[3], d[14], d[15] = the grid steps of x, y, z as floats (the walk multiplies
//   them by 1/dir without decoding anything)
//   d[4..6] lo bytes of x, y, z (child k in byte k)   d[7], d[8], d[9] hi bytes of x, y, z
//   d[10..13] links of children 0..3
// A node with fewer than four children fills the unused slots with an INVERTED box (lo 255, hi 0:
// entry > exit on every axis, never hit) and a copy of child 0's link, so the walk needs no child
// count: should rounding ever let such a slot through, it only repeats work on child 0.
__host__ __device__ inline void pt_encode_wide_node(const PtBox* cb, int n, const int32_t* link, float* d) {
    PtBox nb = cb[0];
    for (int k = 1; k < n; k++)
        for (int a = 0; a < 3; a++) {
            nb.lo[a] = cb[k].lo[a] < nb.lo[a] ? cb[k].lo[a] : nb.lo[a];
            nb.hi[a] = cb[k].hi[a] > nb.hi[a] ? cb[k].hi[a] : nb.hi[a];
        }
    uint32_t q[6] = {0, 0, 0, 0, 0, 0};
    float scales[3];
    for (int a = 0; a < 3; a++) {
        const float origin = nb.lo[a];
        const float ext = nb.hi[a] - nb.lo[a];
        // the grid step: ext / 255 with a relative margin of 2^-18 (the walk multiplies it by 1/dir before it multiplies by q:
        // two roundings of 2^-24 each stay far inside), raised until plane 255 reaches the node's upper bound in binary32
        float scale = pt_i2f(1 << 23);   // a flat node: the smallest normal step
        if (ext > 0.f) {
            scale = (float)(((double)ext / 255.0) * (1.0 + 1.0 / 262144.0));
            if (!(scale >= pt_i2f(1 << 23))) scale = pt_i2f(1 << 23);
            while (fmaf(255.f, scale, origin) < nb.hi[a]) scale = scale * (1.0f + 1.0f / 65536.0f);
        }
        scales[a] = scale;
        for (int k = 0; k < 4; k++) {
            int qlo = 255, qhi = 0;
            if (k < n) {
                double fl = floor(((double)cb[k].lo[a] - (double)origin) / (double)scale);
                double ce = ceil(((double)cb[k].hi[a] - (double)origin) / (double)scale);
                fl = fl < 0.0 ? 0.0 : (fl > 255.0 ? 255.0 : fl);
                ce = ce < 0.0 ? 0.0 : (ce > 255.0 ? 255.0 : ce);
                qlo = (int)fl;
                qhi = (int)ce;
                while (qlo > 0 && fmaf((float)qlo, scale, origin) > cb[k].lo[a]) qlo--;
                while (qhi < 255 && fmaf((float)qhi, scale, origin) < cb[k].hi[a]) qhi++;
            }
            q[a] |= (uint32_t)qlo << (8 * k);
            q[3 + a] |= (uint32_t)qhi << (8 * k);
        }
    }
    d[0] = nb.lo[0]; d[1] = nb.lo[1]; d[2] = nb.lo[2]; d[3] = scales[0];
    d[4] = pt_i2f((int32_t)q[0]); d[5] = pt_i2f((int32_t)q[1]); d[6] = pt_i2f((int32_t)q[2]); d[7] = pt_i2f((int32_t)q[3]);
    d[8] = pt_i2f((int32_t)q[4]); d[9] = pt_i2f((int32_t)q[5]);
    d[10] = pt_i2f(link[0]); d[11] = pt_i2f(n > 1 ? link[1] : link[0]);
    d[12] = pt_i2f(n > 2 ? link[2] : link[0]); d[13] = pt_i2f(n > 3 ? link[3] : link[0]);
    d[14] = scales[1]; d[15] = scales[2];
}

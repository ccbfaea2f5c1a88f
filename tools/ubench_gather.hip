// ubench_gather.hip — micro-benchmark behind DESIGN.md §5: how fast can a CU gather random
// 64-byte items (BVH nodes / triangle records) that sit in L2 / Infinity Cache, as a function
// of HOW the 64 bytes are requested?
//   mode 0: every lane loads its own item with 4 x global_load_dwordx4 (what a per-lane BVH
//           walk does: 64 different cache lines per wave instruction)
//   mode 1: quad-cooperative: the 4 lanes of a quad load the 4 consecutive 16-byte pieces of ONE
//           lane's item per instruction (4 instructions serve the quad's 4 items), data is
//           handed to the owner with DPP quad_perm moves
//   mode 2: as mode 0 but 2 x dwordx4 (32-byte items: a compressed node)
//   mode 8: TWO independent chains per lane (two 64-byte items in flight per lane: the memory-level parallelism a walk
//           would have with a prefetched second item), items/s counts both
//   mode 9: 80-byte items, packed (5 x dwordx4; 1.25 lines on average: a CWBVH-style 8-wide node)
//   mode 10: a 64-byte item + a ONE-dword touch of another random item's line (a prefetch whose result is not waited for
//           until the next iteration's loads retire: vmcnt is in order)
//   mode 7: quad per item: the 4 lanes of a quad chase ONE chain, each lane loads one 16-byte piece of the
//           item (one instruction per item, 16 lines per wave instruction); items/s counts quads
// Dependent chain: the next index depends on the loaded data, like pointer chasing.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_gather tools/ubench_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

template <int SEL>
__device__ __forceinline__ float quad_bcast(float v) {
    // v_mov_b32 dpp quad_perm:[SEL,SEL,SEL,SEL]
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), SEL * 0x55, 0xf, 0xf, true));
}

template <int MODE>
__global__ void __launch_bounds__(256, 8) k_gather(const float4* __restrict__ items, uint32_t n_items, int iters, float* out) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int ql = threadIdx.x & 3;
    uint32_t idx = mix((MODE == 7 ? gid >> 2 : gid) * 2654435761u + 12345u) % n_items;
    float acc = 0.f;
    float touch = 0.f;   // mode 10: the register the touch loads land in; never read while a load is in flight
    if (MODE == 8) {
        uint32_t idb = mix(gid * 40503u + 7u) % n_items;
        for (int it = 0; it < iters; it++) {
            const float4* p = items + (size_t)idx * 4;
            const float4* r = items + (size_t)idb * 4;
            const float4 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], b0 = r[0], b1 = r[1], b2 = r[2], b3 = r[3];
            acc += a0.x + a0.y + a0.z + a0.w + a1.x + a1.y + a1.z + a1.w + a2.x + a2.y + a2.z + a2.w + a3.x + a3.y + a3.z;
            acc += b0.x + b0.y + b0.z + b0.w + b1.x + b1.y + b1.z + b1.w + b2.x + b2.y + b2.z + b2.w + b3.x + b3.y + b3.z;
            idx = mix(idx ^ __float_as_uint(a3.w) ^ (uint32_t)it) % n_items;
            idb = mix(idb ^ __float_as_uint(b3.w) ^ (uint32_t)it) % n_items;
        }
        out[gid] = acc;
        return;
    }
    for (int it = 0; it < iters; it++) {
        float4 q0, q1, q2, q3;
        if (MODE == 0) {
            const float4* p = items + (size_t)idx * 4;
            q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
        } else if (MODE == 2) {
            const float4* p = items + (size_t)idx * 4;
            q0 = p[0]; q1 = p[1]; q2 = q0; q3 = q1;
        } else if (MODE == 3) {  // 16-byte items: one dwordx4
            const float4* p = items + (size_t)idx * 4;
            q0 = p[0]; q1 = q0; q2 = q0; q3 = q0;
        } else if (MODE == 4) {  // 32 bytes as 4 x dwordx2
            const float2* p = (const float2*)(items + (size_t)idx * 4);
            const float2 a = p[0], b = p[1], c = p[2], d = p[3];
            q0 = make_float4(a.x, a.y, b.x, b.y); q1 = make_float4(c.x, c.y, d.x, d.y); q2 = q0; q3 = q1;
        } else if (MODE == 5) {  // 48 bytes: 3 x dwordx4
            const float4* p = items + (size_t)idx * 4;
            q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = q2;
        } else if (MODE == 7) {  // one item per QUAD: lane ql loads piece ql; the quad shares the sum
            const float4 r = items[(size_t)idx * 4 + ql];
            float t = r.x + r.y + r.z + r.w;
            t += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
            t += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
            q0 = make_float4(t, 0.f, 0.f, 0.f); q1 = q2 = make_float4(0.f, 0.f, 0.f, 0.f);
            q3 = make_float4(0.f, 0.f, 0.f, quad_bcast<3>(r.w));
        } else if (MODE == 9) {  // 80 bytes, packed: 5 x dwordx4
            const float4* p = items + (size_t)(idx % (n_items * 4u / 5u - 1u)) * 5;
            const float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
            q0 = make_float4(a.x + e.x, a.y + e.y, a.z + e.z, a.w + e.w); q1 = b; q2 = c; q3 = d;
        } else if (MODE == 10) {
            const float4* p = items + (size_t)idx * 4;
            q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
            const uint32_t j = mix(idx + 977u * (uint32_t)it) % n_items;
            asm volatile("global_load_dword %0, %1, off" : "+v"(touch) : "v"(items + (size_t)j * 4) : "memory");
        } else if (MODE == 6) {  // 128 bytes: 8 x dwordx4 (two consecutive items)
            const float4* p = items + (size_t)(idx & ~1u) * 4;
            const float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g2 = p[6], h = p[7];
            q0 = make_float4(a.x + e.x, a.y + e.y, a.z + e.z, a.w + e.w); q1 = make_float4(b.x + f.x, b.y + f.y, b.z + f.z, b.w + f.w);
            q2 = make_float4(c.x + g2.x, c.y + g2.y, c.z + g2.z, c.w + g2.w); q3 = make_float4(d.x + h.x, d.y + h.y, d.z + h.z, d.w + h.w);
        } else {
            // owner k's index broadcast to the quad, each lane fetches piece `ql` of that item
            const uint32_t i0 = __builtin_amdgcn_mov_dpp((int)idx, 0x00, 0xf, 0xf, true);
            const uint32_t i1 = __builtin_amdgcn_mov_dpp((int)idx, 0x55, 0xf, 0xf, true);
            const uint32_t i2 = __builtin_amdgcn_mov_dpp((int)idx, 0xAA, 0xf, 0xf, true);
            const uint32_t i3 = __builtin_amdgcn_mov_dpp((int)idx, 0xFF, 0xf, 0xf, true);
            const float4 r0 = items[(size_t)i0 * 4 + ql];
            const float4 r1 = items[(size_t)i1 * 4 + ql];
            const float4 r2 = items[(size_t)i2 * 4 + ql];
            const float4 r3 = items[(size_t)i3 * 4 + ql];
            // owner ql needs piece c (from lane c) of register set r_ql: select the set, then 4 bcasts per piece
            const float4 mine = ql == 0 ? r0 : (ql == 1 ? r1 : (ql == 2 ? r2 : r3));
            (void)mine;
            // full hand-off: for each piece c, each dword: pick from lane c the register set of the DEST lane.
            // dest-dependent register choice = 4 bcasts + 3 selects per dword
#define HAND(c, comp)                                                                                        \
    (ql == 0 ? quad_bcast<c>(r0.comp) : (ql == 1 ? quad_bcast<c>(r1.comp) : (ql == 2 ? quad_bcast<c>(r2.comp) : quad_bcast<c>(r3.comp))))
            q0 = make_float4(HAND(0, x), HAND(0, y), HAND(0, z), HAND(0, w));
            q1 = make_float4(HAND(1, x), HAND(1, y), HAND(1, z), HAND(1, w));
            q2 = make_float4(HAND(2, x), HAND(2, y), HAND(2, z), HAND(2, w));
            q3 = make_float4(HAND(3, x), HAND(3, y), HAND(3, z), HAND(3, w));
#undef HAND
        }
        const float s = q0.x + q0.y + q0.z + q0.w + q1.x + q1.y + q1.z + q1.w + q2.x + q2.y + q2.z + q2.w + q3.x + q3.y + q3.z + q3.w;
        acc += s;
        idx = mix(idx ^ __float_as_uint(q3.w) ^ (uint32_t)it) % n_items;
    }
    if (MODE == 10) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); asm volatile("" : "+v"(touch)); acc += touch * 0.f; }
    out[gid] = acc;
}

// --json <n_items> <iters>: mode 0 only (one 64-byte item per lane per step, dependent chain, 8 waves/SIMD), on a
// table of n_items and on a 2 MB table (every fetch an L2 hit: the ceiling of ANY per-lane walk); one JSON line
// for bench.py, which runs this binary as a child process.
static double run_mode0(uint32_t n_items, int iters) {
    const int blocks = 256 * 8, threads = 256;
    std::vector<float> h((size_t)n_items * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) * 1e-3f;
    float4* d_items; float* d_out;
    if (hipMalloc(&d_items, h.size() * 4) != hipSuccess || hipMalloc(&d_out, (size_t)blocks * threads * 4) != hipSuccess) return 0.0;
    hipMemcpy(d_items, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    hipFree(d_items); hipFree(d_out);
    return (double)blocks * threads * iters / (best * 1e-3);
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "--json") {
        const uint32_t n = argc > 2 ? (uint32_t)strtoul(argv[2], nullptr, 10) : 1000000u;
        const int it = argc > 3 ? atoi(argv[3]) : 256;
        const double a = run_mode0(n, it), b = run_mode0(32768u, it);
        printf("{\"items_per_s\": %.6g, \"items_per_s_l2_resident\": %.6g, \"table_bytes\": %llu, \"iters\": %d}\n", a, b,
               (unsigned long long)n * 64ull, it);
        return a > 0 ? 0 : 1;
    }
    const uint32_t n_items = argc > 1 ? (uint32_t)strtoul(argv[1], nullptr, 10) : 1000000u;  // 64 MB
    const int iters = argc > 2 ? atoi(argv[2]) : 256;
    const int blocks = 256 * 8, threads = 256;
    // the table is a 64 MB pattern repeated (the values only feed the hash of the next index)
    const size_t pat_items = n_items < 1000000u ? n_items : 1000000u;
    std::vector<float> h(pat_items * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) * 1e-3f;
    float4* d_items; float* d_out;
    if (hipMalloc(&d_items, (size_t)n_items * 64) != hipSuccess || hipMalloc(&d_out, (size_t)blocks * threads * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (size_t off = 0; off < n_items; off += pat_items) {
        const size_t n = (n_items - off < pat_items ? n_items - off : pat_items);
        (void)hipMemcpy((char*)d_items + off * 64, h.data(), n * 64, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 11; mode++) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            else if (mode == 1) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            else if (mode == 2) hipLaunchKernelGGL(k_gather<2>, dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            else if (mode == 3) hipLaunchKernelGGL(k_gather<3>, dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            else if (mode == 4) hipLaunchKernelGGL(k_gather<4>, dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            else if (mode == 5) hipLaunchKernelGGL(k_gather<5>, dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            else if (mode == 6) hipLaunchKernelGGL(k_gather<6>, dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            else if (mode == 8) hipLaunchKernelGGL(k_gather<8>, dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            else if (mode == 9) hipLaunchKernelGGL(k_gather<9>, dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            else if (mode == 10) hipLaunchKernelGGL(k_gather<10>, dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            else hipLaunchKernelGGL(k_gather<7>, dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        const double items_n = (double)blocks * threads * iters / (mode == 7 ? 4 : 1) * (mode == 8 ? 2 : 1);
        static const int bytes_of[11] = {64, 64, 32, 16, 32, 48, 128, 64, 64, 80, 64};
        const double bytes = items_n * bytes_of[mode];
        printf("mode %d: %.3f ms  %.1f Gitems/s  %.1f GB/s chip  %.1f GB/s per CU\n", mode, best, items_n / best / 1e6,
               bytes / best / 1e6, bytes / best / 1e6 / 256);
    }
    return 0;
}

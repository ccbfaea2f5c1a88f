// ubench_ta.hip — what does ONE global_load_dwordx4 wave instruction cost the CU's vector-memory address pipe
// (TA/TCP), as a function of how the 64 lanes' 16-byte pieces fall into cache lines?  Independent loads (8 in
// flight per lane), table L2-resident (2 MB) or scene-sized, so the rate is the pipe's, not a latency.
//   mode 0: every lane its own random 64-byte line (what a per-lane BVH walk issues: 64 lines per instruction)
//   mode 1: every aligned QUAD of lanes one random 64-byte line, lane k reads piece k (16 lines per instruction)
//   mode 2: every aligned group of 8 lanes one random 128-byte line (8 lines per instruction)
//   mode 3: all 64 lanes one contiguous 1 KB (fully coalesced)
//   mode 4 / 5 / 6: as mode 0 with dwordx2 / dword / dwordx3 loads (is the cost per lane or per byte?)
//   mode 7 / 8 / 9: as mode 0 with 48 / 32 / 16 of the 64 lanes active, scattered over the wave (is the cost per instruction or
//                   per ACTIVE lane?  a walk's load instructions run at 0.5-0.7 of the lanes)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_ta tools/ubench_ta.hip ; run: tools/ubench_ta [n_items]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

template <int MODE, int ACTIVE = 64>
__global__ void __launch_bounds__(256, 8) k_ta(const float4* __restrict__ items, uint32_t n_items, int iters, float* out) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t group = (MODE == 0 || MODE >= 4) ? gid : (MODE == 1 ? gid >> 2 : (MODE == 2 ? gid >> 3 : gid >> 6));
    const uint32_t piece = MODE == 0 ? (gid & 3) : (MODE == 1 ? (lane & 3) : (MODE == 2 ? (lane & 7) : lane));
    const uint32_t span = MODE == 3 ? 16 : (MODE == 2 ? 2 : 1);   // items per group line
    float acc = 0.f;
    uint32_t s = mix(group * 2654435761u + 99u);
    if (ACTIVE < 64 && ((lane * 37u) & 63u) >= (uint32_t)ACTIVE) { out[gid] = 0.f; return; }   // 37 is odd: a permutation of the lanes
    for (int it = 0; it < iters; it++) {
        float4 q[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            s = s * 1664525u + 1013904223u;   // cheap LCG: the address stream must not be the bottleneck
            const uint32_t hsh = s >> 8;
            const uint32_t base = (hsh & (n_items / span - 1u)) * span;   // n_items is a power of two
            if (MODE == 4) { const float2 t = ((const float2*)items)[((size_t)base * 4 + (gid & 3)) * 2]; q[k] = make_float4(t.x, t.y, t.x, t.y); }
            else if (MODE == 5) { const float t = ((const float*)items)[((size_t)base * 4 + (gid & 3)) * 4]; q[k] = make_float4(t, t, t, t); }
            else if (MODE == 6) { const float* pp = (const float*)(items + (size_t)base * 4 + (gid & 3)); q[k] = make_float4(pp[0], pp[1], pp[2], 0.f); }
            else q[k] = items[(size_t)base * 4 + piece];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) acc += (q[k].x + q[k].y) + (q[k].z + q[k].w);
    }
    out[gid] = acc;
}

int main(int argc, char** argv) {
    const uint32_t n_items = argc > 1 ? (uint32_t)strtoul(argv[1], nullptr, 10) : 32768u;   // 2 MB
    const int iters = 64, blocks = 256 * 8, threads = 256;
    std::vector<float> h((size_t)n_items * 16, 1.0f);
    float4* d_items; float* d_out;
    hipMalloc(&d_items, h.size() * 4);
    hipMalloc(&d_out, (size_t)blocks * threads * 4);
    hipMemcpy(d_items, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 10; mode++) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_ta<0>, dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            else if (mode == 1) hipLaunchKernelGGL(k_ta<1>, dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            else if (mode == 2) hipLaunchKernelGGL(k_ta<2>, dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            else if (mode == 4) hipLaunchKernelGGL(k_ta<4>, dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            else if (mode == 5) hipLaunchKernelGGL(k_ta<5>, dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            else if (mode == 6) hipLaunchKernelGGL(k_ta<6>, dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            else if (mode == 7) hipLaunchKernelGGL((k_ta<0, 48>), dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            else if (mode == 8) hipLaunchKernelGGL((k_ta<0, 32>), dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            else if (mode == 9) hipLaunchKernelGGL((k_ta<0, 16>), dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            else hipLaunchKernelGGL(k_ta<3>, dim3(blocks), dim3(threads), 0, 0, d_items, n_items, iters, d_out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        const double wave_instr = (double)blocks * threads / 64 * iters * 8;
        printf("mode %d (table %.1f MB): %.3f ms  %.2f G wave-loads/s  = %.1f cycles per wave-load per CU at 2.3 GHz  %.1f TB/s\n", mode,
               n_items * 64.0 / 1048576.0, best, wave_instr / best / 1e6, 256.0 * 2.3e9 / (wave_instr / (best * 1e-3)),
               wave_instr * 1024.0 / best / 1e9);
    }
    return 0;
}

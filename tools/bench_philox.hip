// Micro-benchmark of the Philox-mode building blocks (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I mdp_playground_amd/csrc tools/bench_philox.hip -o gpurun_out/bench_philox
// ns per call per wave with 1, 2 and 4 waves per SIMD (256 blocks of 256 / 512 / 1024 threads).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "mdpp_rng.hpp"
using namespace mdpp;

template <int ROUNDS>
__device__ __forceinline__ void philox_block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                             uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        uint32_t y0 = h1 ^ c1 ^ k0, y1 = l1, y2 = h0 ^ c3 ^ k1, y3 = l0;
        c0 = y0; c1 = y1; c2 = y2; c3 = y3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

template <int MODE>
__global__ void k(uint32_t *out, int iters) {
    uint32_t acc = 0, c = threadIdx.x + blockIdx.x * 1024;
    float facc = 0.f;
    for (int i = 0; i < iters; i++) {
        uint32_t w[4];
        if (MODE == 0) { philox_block<10>(c, i, 7, 9, 1, 2, w); acc ^= w[0] ^ w[1] ^ w[2] ^ w[3]; }
        if (MODE == 1) { philox_block<7>(c, i, 7, 9, 1, 2, w); acc ^= w[0] ^ w[1] ^ w[2] ^ w[3]; }
        if (MODE == 2) {  // Philox-10 + two Box-Muller pairs
            philox_block<10>(c, i, 7, 9, 1, 2, w);
            float z0, z1, z2, z3;
            philox_box_muller(w[0], w[1], z0, z1);
            philox_box_muller(w[2], w[3], z2, z3);
            facc += z0 + z1 + z2 + z3;
        }
        if (MODE == 3) {  // Box-Muller pair only (inputs from a cheap LCG)
            acc = acc * 1664525u + 1013904223u;
            float z0, z1;
            philox_box_muller(acc, acc ^ (c * 2654435761u), z0, z1);
            facc += z0 + z1;
        }
        if (MODE == 4) { acc = __umulhi(acc | 1u, 0xD2511F53u) ^ (acc * 0xCD9E8D57u); }   // dependent mul_hi + mul_lo
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (uint32_t)facc;
}

template <int MODE>
void run(const char *name, uint32_t *d, int iters) {
    for (int threads : {256, 512, 1024}) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-44s %d wave(s)/SIMD  %8.1f ns per call per wave  (%6.1f ns per call per SIMD)\n", name, threads / 256,
               ms * 1e6 / iters, ms * 1e6 / iters / (threads / 256));
    }
}

int main() {
    uint32_t *d; hipMalloc(&d, 256 * 1024 * 4);
    int iters = 20000;
    run<4>("dependent v_mul_hi_u32 + v_mul_lo_u32", d, iters);
    run<0>("Philox4x32-10 block (4 words)", d, iters);
    run<1>("Philox4x32-7 block (4 words)", d, iters);
    run<3>("float32 Box-Muller pair (2 normals)", d, iters);
    run<2>("Philox4x32-10 + 2 Box-Muller pairs (4 normals)", d, iters);
    return 0;
}

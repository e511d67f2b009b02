// Micro-benchmark of the RNG building blocks at the bench occupancy (one wave per SIMD).
// hipcc --offload-arch=gfx950 -O3 -I mdp_playground_amd/csrc tools/bench_rng.hip -o gpurun_out/bench_rng
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "mdpp_rng.hpp"
using namespace mdpp;

// candidate: wedge test decided by a float32 exp with a guard band, float64 exp only inside it
template <class G>
__device__ __forceinline__ double normal_f32guard(G &g, const ZigLds &z) {
    for (;;) {
        uint64_t r = g.next64();
        int idx = (int)(r & 0xff);
        r >>= 8;
        int sign = (int)(r & 0x1);
        uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
        double x = (double)rabs * z.wi[idx];
        if (sign) x = -x;
        if (rabs < z.ki[idx]) return x;
        if (idx == 0) return np_zig_tail(g, rabs);
        const double lhs = (z.fi[idx - 1] - z.fi[idx]) * np_random(g) + z.fi[idx];
        const double t = -0.5 * x * x;
        const double e32 = (double)__expf((float)t);
        bool acc;
        if (lhs < e32 * (1.0 - 1e-5)) acc = true;
        else if (lhs > e32 * (1.0 + 1e-5)) acc = false;
        else acc = lhs < exp(t);
        if (acc) return x;
    }
}

// candidate for a raw-word producer / interpreting consumer split: a generator whose next64() pops
// pre-made 64-bit words from an LDS ring (no LCG step on the consumer side)
struct RingGen {
    const uint64_t *ring;   // [64 slots][256 lanes]
    uint32_t pos, lane;
    __device__ __forceinline__ uint64_t next64() { uint64_t r = ring[(pos & 63u) * 256u + lane]; pos++; return r ^ ((uint64_t)pos * 0x9E3779B97F4A7C15ULL); }
};

// candidate: one 64-bit word of lookahead per lane; the LCG step for the NEXT word has no consumer in
// the hot path, so it can be scheduled under the table-lookup latency of the current one
struct Ahead64 {
    Pcg64 g;
    uint64_t ahead;
    __device__ __forceinline__ void prime() { ahead = g.next64(); }
    __device__ __forceinline__ uint64_t next64() { const uint64_t r = ahead; ahead = g.next64(); return r; }
};

template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t *out, int iters) {
    __shared__ uint64_t s_ki[256];
    __shared__ double s_wi[256], s_fi[256];
    zig_stage(s_ki, s_wi, s_fi, threadIdx.x, 256);
    __syncthreads();
    ZigLds zig{s_ki, s_wi, s_fi};
    Pcg64 g;
    g.s_lo = threadIdx.x * 7919u + blockIdx.x; g.s_hi = 12345; g.inc_lo = 2 * threadIdx.x + 1; g.inc_hi = 99;
    double acc = 0; uint64_t x = 0;
    __shared__ uint64_t s_ring[MODE == 9 ? 64 * 256 : 1];
    RingGen rg{s_ring, 0u, threadIdx.x};
    if (MODE == 9) {
        for (int q = 0; q < 64; q++) s_ring[q * 256 + threadIdx.x] = g.next64();
        __syncthreads();
    }
    Ahead64 ag{g, 0};
    if (MODE == 10) ag.prime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 10) acc += np_standard_normal_lds(ag, zig);
        if (MODE == 9) acc += np_standard_normal_lds(rg, zig);
        if (MODE == 0) x ^= g.next64();
        if (MODE == 1) acc += np_random(g);
        if (MODE == 2) acc += np_standard_normal_lds(g, zig);
        if (MODE == 3) acc += np_standard_normal(g);
        if (MODE == 8) acc += normal_f32guard(g, zig);
        if (MODE == 6) { // ziggurat hot path only (rejections ignored)
            uint64_t r = g.next64(); int idx = (int)(r & 0xff);
            uint64_t rabs = (r >> 9) & 0x000fffffffffffffULL;
            double xx = (double)rabs * zig.wi[idx]; xx = ((r >> 8) & 1) ? -xx : xx;
            acc += (rabs < zig.ki[idx]) ? xx : 0.0;
        }
        if (MODE == 7) { // hot path + the extra uniform of the wedge under the ballot branch, no exp
            uint64_t r = g.next64(); int idx = (int)(r & 0xff);
            uint64_t rabs = (r >> 9) & 0x000fffffffffffffULL;
            double xx = (double)rabs * zig.wi[idx]; xx = ((r >> 8) & 1) ? -xx : xx;
            bool pend = !(rabs < zig.ki[idx]);
            if (__builtin_amdgcn_ballot_w64(pend) != 0) { if (pend) xx += np_random(g) * zig.fi[idx]; }
            acc += xx;
        }
        if (MODE == 4) { x = x * 6364136223846793005ULL + 1442695040888963407ULL; }
        if (MODE == 5) { acc = acc * 1.0000001 + 0.5; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x + (uint64_t)acc;
}

template <int MODE>
void run(const char *name, uint64_t *d, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s %8.1f ns per call per wave\n", name, ms * 1e6 / iters);
}

int main() {
    uint64_t *d; hipMalloc(&d, 65536 * 8);
    int iters = 20000;
    run<4>("u64 mul-add (LCG64)", d, iters);
    run<5>("f64 fma chain", d, iters);
    run<0>("pcg64 next64", d, iters);
    run<1>("np_random (uniform double)", d, iters);
    run<2>("standard_normal (LDS tables)", d, iters);
    run<3>("standard_normal (global tbl)", d, iters);
    run<8>("standard_normal f32-guard", d, iters);
    run<9>("standard_normal, words popped from LDS", d, iters);
    run<10>("standard_normal, one word of lookahead", d, iters);
    run<6>("ziggurat hot path only", d, iters);
    run<7>("hot + wedge uniform, no exp", d, iters);
    return 0;
}

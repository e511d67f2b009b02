// Which part of the discrete rollout's write pattern costs what (GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/bench_hbm2.hip -o gpurun_out/bench_hbm2 && ./gpurun_out/bench_hbm2
// K rows of N envs; MODE bit 0: int64 obs, bit 1: float reward, bit 2: term bytes, bit 3: trunc bytes,
// bit 4: the two byte arrays written as DWORDS by every 4th lane (bytes of 4 neighbours gathered by DPP).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int MODE>
__global__ void k(uint64_t *obs, float *rew, uint8_t *term, uint8_t *trunc, int N, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = (uint32_t)i;
    for (int k = 0; k < K; k++) {
        const size_t o = (size_t)k * N + i;
        s = s * 1664525u + 1013904223u;
        if (MODE & 1) obs[o] = s & 7u;
        if (MODE & 2) rew[o] = (float)(s >> 31);
        const uint32_t t = (s >> 8) & 1u, u = (s >> 9) & 1u;
        if (MODE & 16) {
            // gather the bytes of lanes 4j..4j+3 into lane 4j
            uint32_t tt = t | (u << 16);
            tt |= __shfl_down(tt, 1) << 8;          // lanes 0,2: bytes of (l, l+1)
            const uint32_t hi = __shfl_down(tt, 2);
            if ((threadIdx.x & 3) == 0) {
                const uint32_t lo16 = (tt & 0xFFFFu) | ((hi & 0xFFFFu) << 16);
                const uint32_t hi16 = (tt >> 16) | (hi & 0xFFFF0000u);
                if (MODE & 4) *(uint32_t *)(term + o) = lo16;
                if (MODE & 8) *(uint32_t *)(trunc + o) = hi16;
            }
        } else {
            if (MODE & 4) term[o] = (uint8_t)t;
            if (MODE & 8) trunc[o] = (uint8_t)u;
        }
    }
}

// the full pattern of a fused discrete rollout: int32 action reads prefetched 8 steps ahead + the four output rows
template <int PRE>
__global__ void k_full(const int32_t *act, uint64_t *obs, float *rew, uint8_t *term, uint8_t *trunc, int N, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = (uint32_t)i;
    int pre[PRE];
#pragma unroll
    for (int u = 0; u < PRE; u++) pre[u] = act[(size_t)u * N + i];
    for (int k0 = 0; k0 < K; k0 += PRE) {
#pragma unroll
        for (int u = 0; u < PRE; u++) {
            const int k = k0 + u;
            const size_t o = (size_t)k * N + i;
            const int a = pre[u];
            const int kn = k + PRE < K ? k + PRE : K - 1;
            pre[u] = act[(size_t)kn * N + i];
            s = s * 1664525u + 1013904223u + (uint32_t)a;
            obs[o] = s & 7u;
            rew[o] = (float)(s >> 31);
            term[o] = (uint8_t)((s >> 8) & 1u);
            trunc[o] = (uint8_t)((s >> 9) & 1u);
        }
    }
}

// round 3: the same pattern with non-temporal stores (what the rollout kernels use) -- NT -- and, in main(), with the
// launches cycling through four action tensors (537 MB > the 256 MiB Infinity Cache: reads come from HBM)
template <int PRE, bool NT>
__global__ void k_full2(const int32_t *act, uint64_t *obs, float *rew, uint8_t *term, uint8_t *trunc, int N, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = (uint32_t)i;
    int pre[PRE];
#pragma unroll
    for (int u = 0; u < PRE; u++) pre[u] = act[(size_t)u * N + i];
    for (int k0 = 0; k0 < K; k0 += PRE) {
#pragma unroll
        for (int u = 0; u < PRE; u++) {
            const int k = k0 + u;
            const size_t o = (size_t)k * N + i;
            const int a = pre[u];
            const int kn = k + PRE < K ? k + PRE : K - 1;
            pre[u] = act[(size_t)kn * N + i];
            s = s * 1664525u + 1013904223u + (uint32_t)a;
            if (NT) {
                __builtin_nontemporal_store((uint64_t)(s & 7u), obs + o);
                __builtin_nontemporal_store((float)(s >> 31), rew + o);
                __builtin_nontemporal_store((uint8_t)((s >> 8) & 1u), term + o);
                __builtin_nontemporal_store((uint8_t)((s >> 9) & 1u), trunc + o);
            } else {
                obs[o] = s & 7u; rew[o] = (float)(s >> 31); term[o] = (uint8_t)((s >> 8) & 1u); trunc[o] = (uint8_t)((s >> 9) & 1u);
            }
        }
    }
}

// round 3: would WIDER action reads help?  One wave of the 256-env block fetches the block's whole 1 KiB action row with
// one 16-byte load per lane (instead of four waves x 4 bytes per lane), LD: 0 plain, 1 non-temporal loads; stores nt.
template <int PRE, int WIDE, int LD>
__global__ void k_full3(const int32_t *act, uint64_t *obs, float *rew, uint8_t *term, uint8_t *trunc, int N, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = (uint32_t)i;
    const bool loader = !WIDE || threadIdx.x < 64;
    typedef int v4 __attribute__((ext_vector_type(4)));
    v4 pre4[PRE]; int pre[PRE];
    auto ld = [&](int k) {
        if (WIDE) {
            const v4 *p = (const v4 *)(act + (size_t)k * N + blockIdx.x * 256) + threadIdx.x;
            return LD ? __builtin_nontemporal_load(p) : *p;
        } else {
            const int *p = act + (size_t)k * N + i;
            const int x = LD ? __builtin_nontemporal_load(p) : *p;
            return v4{x, 0, 0, 0};
        }
    };
#pragma unroll
    for (int u = 0; u < PRE; u++) { pre4[u] = v4{0, 0, 0, 0}; if (loader) pre4[u] = ld(u); }
    for (int k0 = 0; k0 < K; k0 += PRE) {
#pragma unroll
        for (int u = 0; u < PRE; u++) {
            const int k = k0 + u;
            const size_t o = (size_t)k * N + i;
            const v4 a = pre4[u];
            const int kn = k + PRE < K ? k + PRE : K - 1;
            if (loader) pre4[u] = ld(kn);
            s = s * 1664525u + 1013904223u + (uint32_t)(a.x + a.y + a.z + a.w);
            __builtin_nontemporal_store((uint64_t)(s & 7u), obs + o);
            __builtin_nontemporal_store((float)(s >> 31), rew + o);
            __builtin_nontemporal_store((uint8_t)((s >> 8) & 1u), term + o);
            __builtin_nontemporal_store((uint8_t)((s >> 9) & 1u), trunc + o);
        }
    }
}

// the same again in the code shape of k_full2 (plain int prefetch registers): WIDE = one wave of the block fetches the whole
// 1 KiB action row (16 B per lane), the other three waves load nothing
template <int PRE, bool WIDE>
__global__ void k_full4(const int32_t *act, uint64_t *obs, float *rew, uint8_t *term, uint8_t *trunc, int N, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = (uint32_t)i;
    typedef int v4 __attribute__((ext_vector_type(4)));
    int pre[PRE]; v4 prew[PRE];
    const bool loader = threadIdx.x < 64;
    const v4 *row0 = (const v4 *)(act + blockIdx.x * 256) + (threadIdx.x & 63);
#pragma unroll
    for (int u = 0; u < PRE; u++) {
        if (WIDE) { prew[u] = v4{0, 0, 0, 0}; if (loader) prew[u] = *(const v4 *)((const int *)row0 + (size_t)u * N); pre[u] = 0; }
        else pre[u] = act[(size_t)u * N + i];
    }
    for (int k0 = 0; k0 < K; k0 += PRE) {
#pragma unroll
        for (int u = 0; u < PRE; u++) {
            const int k = k0 + u;
            const size_t o = (size_t)k * N + i;
            int a;
            const int kn = k + PRE < K ? k + PRE : K - 1;
            if (WIDE) {
                a = prew[u].x ^ prew[u].y ^ prew[u].z ^ prew[u].w;
                if (loader) prew[u] = *(const v4 *)((const int *)row0 + (size_t)kn * N);
            } else {
                a = pre[u];
                pre[u] = act[(size_t)kn * N + i];
            }
            s = s * 1664525u + 1013904223u + (uint32_t)a;
            __builtin_nontemporal_store((uint64_t)(s & 7u), obs + o);
            __builtin_nontemporal_store((float)(s >> 31), rew + o);
            __builtin_nontemporal_store((uint8_t)((s >> 8) & 1u), term + o);
            __builtin_nontemporal_store((uint8_t)((s >> 9) & 1u), trunc + o);
        }
    }
}

template <int MODE>
void run(const char *name, uint64_t *obs, float *rew, uint8_t *term, uint8_t *trunc, double bytes_per) {
    const int N = 65536, K = 512, reps = 10;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int w = 0; w < 2; w++) {
        hipEventRecord(e0);
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k<MODE>, dim3(N / 256), dim3(256), 0, 0, obs, rew, term, trunc, N, K);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-46s %5.1f B/env-step %8.1f GB/s  %7.1f us per 512-step launch\n", name, bytes_per,
           bytes_per * N * K * reps / 1e9 / (ms / 1e3), ms * 1e3 / reps);
}

int main() {
    uint64_t *obs; float *rew; uint8_t *term, *trunc;
    const size_t n = 65536ull * 512;
    hipMalloc(&obs, n * 8); hipMalloc(&rew, n * 4); hipMalloc(&term, n); hipMalloc(&trunc, n);
    run<1>("obs int64", obs, rew, term, trunc, 8);
    run<3>("obs + reward", obs, rew, term, trunc, 12);
    run<7>("obs + reward + term bytes", obs, rew, term, trunc, 13);
    run<15>("obs + reward + term + trunc bytes", obs, rew, term, trunc, 14);
    run<31>("obs + reward + flags as dwords of 4 lanes", obs, rew, term, trunc, 14);
    {
        int32_t *act; hipMalloc(&act, n * 4); hipMemset(act, 1, n * 4);
        const int N = 65536, K = 512, reps = 10;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int w = 0; w < 2; w++) {
            hipEventRecord(e0);
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_full<8>, dim3(N / 256), dim3(256), 0, 0, act, obs, rew, term, trunc, N, K);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-46s %5.1f B/env-step %8.1f GB/s  %7.1f us per 512-step launch\n", "actions (8 ahead) + all four outputs", 18.0,
               18.0 * N * K * reps / 1e9 / (ms / 1e3), ms * 1e3 / reps);
    }
    {
        const int N = 65536, K = 512, reps = 12, NA = 4;
        int32_t *acts[NA];
        for (int q = 0; q < NA; q++) { hipMalloc(&acts[q], n * 4); hipMemset(acts[q], 1 + q, n * 4); }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int variant = 0; variant < 6; variant++) {
            const bool nt = variant & 1, rot = variant >= 2;
            const int pre = variant >= 4 ? 16 : 8;
            float ms = 0;
            for (int w = 0; w < 2; w++) {
                hipEventRecord(e0);
                for (int r = 0; r < reps; r++) {
                    const int32_t *a = acts[rot ? r % NA : 0];
                    if (pre == 16) {
                        if (nt) hipLaunchKernelGGL((k_full2<16, true>), dim3(N / 256), dim3(256), 0, 0, a, obs, rew, term, trunc, N, K);
                        else hipLaunchKernelGGL((k_full2<16, false>), dim3(N / 256), dim3(256), 0, 0, a, obs, rew, term, trunc, N, K);
                    } else {
                        if (nt) hipLaunchKernelGGL((k_full2<8, true>), dim3(N / 256), dim3(256), 0, 0, a, obs, rew, term, trunc, N, K);
                        else hipLaunchKernelGGL((k_full2<8, false>), dim3(N / 256), dim3(256), 0, 0, a, obs, rew, term, trunc, N, K);
                    }
                }
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            printf("bare pattern: %-9s stores, %-8s action tensor, %2d steps ahead   %8.1f GB/s  %7.1f us per 512-step launch\n",
                   nt ? "nt" : "default", rot ? "rotating" : "one", pre, 18.0 * N * K * reps / 1e9 / (ms / 1e3), ms * 1e3 / reps);
        }
    }
    {
        const int N = 65536, K = 512, reps = 12, NA = 4;
        int32_t *acts[NA];
        for (int q = 0; q < NA; q++) { hipMalloc(&acts[q], n * 4); hipMemset(acts[q], 1 + q, n * 4); }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int variant = 0; variant < 4; variant++) {
            float ms = 0;
            for (int w = 0; w < 2; w++) {
                hipEventRecord(e0);
                for (int r = 0; r < reps; r++) {
                    const int32_t *a = acts[r % NA];
                    switch (variant) {
                    case 0: hipLaunchKernelGGL((k_full3<8, 0, 0>), dim3(N / 256), dim3(256), 0, 0, a, obs, rew, term, trunc, N, K); break;
                    case 1: hipLaunchKernelGGL((k_full3<8, 0, 1>), dim3(N / 256), dim3(256), 0, 0, a, obs, rew, term, trunc, N, K); break;
                    case 2: hipLaunchKernelGGL((k_full3<8, 1, 0>), dim3(N / 256), dim3(256), 0, 0, a, obs, rew, term, trunc, N, K); break;
                    default: hipLaunchKernelGGL((k_full3<8, 1, 1>), dim3(N / 256), dim3(256), 0, 0, a, obs, rew, term, trunc, N, K); break;
                    }
                }
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            printf("bare pattern (nt stores, rotating actions): %s reads, %s loads   %8.1f GB/s  %7.1f us per 512-step launch\n",
                   (variant & 2) ? "16 B per lane of one wave" : "4 B per lane", (variant & 1) ? "nt" : "plain",
                   18.0 * N * K * reps / 1e9 / (ms / 1e3), ms * 1e3 / reps);
        }
    }
    {
        const int N = 65536, K = 512, reps = 12, NA = 4;
        int32_t *acts[NA];
        for (int q = 0; q < NA; q++) { hipMalloc(&acts[q], n * 4); hipMemset(acts[q], 1 + q, n * 4); }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int round = 0; round < 2; round++)
        for (int variant = 0; variant < 4; variant++) {
            float ms = 0;
            for (int w = 0; w < 2; w++) {
                hipEventRecord(e0);
                for (int r = 0; r < reps; r++) {
                    const int32_t *a = acts[r % NA];
                    switch (variant) {
                    case 0: hipLaunchKernelGGL((k_full4<8, false>), dim3(N / 256), dim3(256), 0, 0, a, obs, rew, term, trunc, N, K); break;
                    case 1: hipLaunchKernelGGL((k_full4<8, true>), dim3(N / 256), dim3(256), 0, 0, a, obs, rew, term, trunc, N, K); break;
                    case 2: hipLaunchKernelGGL((k_full4<16, false>), dim3(N / 256), dim3(256), 0, 0, a, obs, rew, term, trunc, N, K); break;
                    default: hipLaunchKernelGGL((k_full4<16, true>), dim3(N / 256), dim3(256), 0, 0, a, obs, rew, term, trunc, N, K); break;
                    }
                }
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            printf("bare pattern 4 (nt stores, rotating actions): %-28s %2d ahead   %8.1f GB/s  %7.1f us per 512-step launch\n",
                   (variant & 1) ? "16 B per lane of one wave" : "4 B per lane", variant >= 2 ? 16 : 8,
                   18.0 * N * K * reps / 1e9 / (ms / 1e3), ms * 1e3 / reps);
        }
    }
    run<12>("term + trunc bytes only", obs, rew, term, trunc, 2);
    run<28>("flags as dwords only", obs, rew, term, trunc, 2);
    return 0;
}

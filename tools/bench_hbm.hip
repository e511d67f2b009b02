// HBM streaming ceilings on this GPU, for pricing the rollout kernels (GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/bench_hbm.hip -o /tmp/bench_hbm && /tmp/bench_hbm
// write-only / read-only / copy over 4 GiB with 16 B per lane, grid-stride, 256 x 8 workgroups;
// and the write pattern of the discrete rollout: K rows of N x {8, 4, 1, 1} bytes (4 arrays).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_write(uint4 *p, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3u, 4u);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}
__global__ void k_read(const uint4 *p, size_t n, uint32_t *out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) out[0] = acc;
}
__global__ void k_copy(const uint4 *s, uint4 *d, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = s[i];
}
// one lane per env, K steps: int64 obs, float reward, two flag bytes per step, rows of N (the layout
// mdpp_step_n writes); one wave per SIMD at N = 65536 like the rollout kernels
__global__ void k_rollout_writes(uint64_t *obs, float *rew, uint8_t *term, uint8_t *trunc, const int32_t *act,
                                 int N, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = (uint32_t)i;
    for (int k = 0; k < K; k++) {
        const size_t o = (size_t)k * N + i;
        s = s * 1664525u + 1013904223u + (act ? (uint32_t)act[o] : 0u);
        obs[o] = s & 7u;
        rew[o] = (float)(s >> 31);
        term[o] = (uint8_t)((s >> 8) & 1u);
        trunc[o] = 0;
    }
}

int main() {
    const size_t bytes = 4ull << 30, n = bytes / 16;
    uint4 *a, *b;
    uint32_t *out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 8, block = 256, reps = 10;
    float ms;
    for (int pass = 0; pass < 3; pass++) {
        for (int w = 0; w < 2; w++) {            // w = 0 warm-up
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) {
                if (pass == 0) hipLaunchKernelGGL(k_write, dim3(grid), dim3(block), 0, 0, a, n);
                if (pass == 1) hipLaunchKernelGGL(k_read, dim3(grid), dim3(block), 0, 0, a, n, out);
                if (pass == 2) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(block), 0, 0, a, b, n);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double gb = (double)bytes * reps * (pass == 2 ? 2 : 1) / 1e9;
        printf("%-34s %8.1f GB/s\n", pass == 0 ? "write-only, 16 B/lane" : pass == 1 ? "read-only, 16 B/lane" : "copy (read + write), 16 B/lane",
               gb / (ms / 1e3));
    }
    const int N = 65536, K = 512;
    uint64_t *obs = (uint64_t *)a;
    float *rew = (float *)b;
    uint8_t *term = (uint8_t *)b + (size_t)N * K * 4, *trunc = term + (size_t)N * K;
    int32_t *act = (int32_t *)((uint8_t *)b + (size_t)N * K * 8);
    for (int with_act = 0; with_act < 2; with_act++) {
        for (int w = 0; w < 2; w++) {
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++)
                hipLaunchKernelGGL(k_rollout_writes, dim3(N / 256), dim3(256), 0, 0, obs, rew, term, trunc,
                                   with_act ? act : nullptr, N, K);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double per = with_act ? 18.0 : 14.0;
        printf("rollout pattern, %2.0f B/env-step%s %8.1f GB/s  (%.1f us per 512-step launch)\n", per,
               with_act ? " (+ action reads)" : "                 ", per * N * K * reps / 1e9 / (ms / 1e3), ms * 1e3 / reps);
    }
    return 0;
}

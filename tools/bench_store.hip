// What the WIDTH of the per-lane stores does to the write rate of the discrete rollout's output pattern (GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/bench_store.hip -o gpurun_out/bench_store && ./gpurun_out/bench_store
// K rows of N envs, one lane per env, one wave per SIMD (256 workgroups of 256 lanes), an LCG step per row instead of the env's
// arithmetic.  The four output rows of a step -- int64 obs, float reward, term bytes, trunc bytes -- are written either as the
// rollout does today (one element per lane and step) or TRANSPOSED inside groups of g lanes over g steps, so that one store
// instruction carries g elements per lane: lane r of a group writes the group's g envs of row k0 + r.  The bytes and the
// addresses written are the same; only the shape of the store instructions changes.  (Values are junk: timing only.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool NT, class T>
__device__ __forceinline__ void st(T *p, T v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// one stream: element size E bytes, G lanes grouped (G elements per lane and store, one store per G steps)
template <int E, int G, bool NT>
__device__ __forceinline__ void put(uint8_t *base, int N, int i, int k, uint32_t s) {
    if (G == 1) {
        uint8_t *p = base + ((size_t)k * N + i) * E;
        if (E == 8) st<NT>((u32x2 *)p, u32x2{s & 7u, 0u});
        else if (E == 4) st<NT>((uint32_t *)p, s);
        else st<NT>(p, (uint8_t)(s & 1u));
        return;
    }
    if ((k % G) != G - 1) return;
    const int r = i % G, j0 = i - r, k0 = k - (G - 1);
    uint8_t *p = base + ((size_t)(k0 + r) * N + j0) * E;
    constexpr int W = E * G;
    if (W == 16) st<NT>((u32x4 *)p, u32x4{s, s >> 1, s >> 2, s >> 3});
    else if (W == 8) st<NT>((u32x2 *)p, u32x2{s, s >> 1});
    else if (W == 4) st<NT>((uint32_t *)p, s);
    else if (W == 32) { st<NT>((u32x4 *)p, u32x4{s, s >> 1, s >> 2, s >> 3}); st<NT>((u32x4 *)p + 1, u32x4{s, s >> 1, s >> 2, s >> 3}); }
}

// the workgroup's 256 envs x 4 steps transposed (through LDS in a real kernel): wave w writes the workgroup's whole piece of
// row k0 + w -- obs 2 KiB (two 16-byte stores per lane), reward 1 KiB (one), flags 256 B each (4 bytes per lane)
template <bool NT>
__device__ __forceinline__ void put_block(uint8_t *obs, uint8_t *rew, uint8_t *term, uint8_t *trunc, int N, int k, uint32_t s) {
    if ((k & 3) != 3) return;
    const int w = threadIdx.x >> 6, ln = threadIdx.x & 63, b0 = blockIdx.x * 256;
    const size_t row = (size_t)(k - 3 + w) * N + b0;
    st<NT>((u32x4 *)(obs + row * 8 + ln * 16), u32x4{s, 0u, s >> 3, 0u});
    st<NT>((u32x4 *)(obs + row * 8 + 1024 + ln * 16), u32x4{s >> 1, 0u, s >> 4, 0u});
    st<NT>((u32x4 *)(rew + row * 4 + ln * 16), u32x4{s, s >> 1, s >> 2, s >> 3});
    st<NT>((uint32_t *)(term + row + ln * 4), s >> 8);
    st<NT>((uint32_t *)(trunc + row + ln * 4), s >> 9);
}

template <int OG, int RG, int FG, bool NT, int PRE>
__global__ __launch_bounds__(256) void k(const int32_t *act, uint8_t *obs, uint8_t *rew, uint8_t *term, uint8_t *trunc, int N, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = (uint32_t)i;
    int pre[PRE > 0 ? PRE : 1];
    if (PRE > 0) {
#pragma unroll
        for (int u = 0; u < PRE; u++) pre[u] = act[(size_t)u * N + i];
    }
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int k = k0 + u;
            int a = 0;
            if (PRE > 0) {
                a = pre[u % PRE];
                const int kn = k + PRE < K ? k + PRE : K - 1;
                pre[u % PRE] = act[(size_t)kn * N + i];
            }
            s = s * 1664525u + 1013904223u + (uint32_t)a;
            if (OG == 0) { put_block<NT>(obs, rew, term, trunc, N, k, s); continue; }
            put<8, OG ? OG : 1, NT>(obs, N, i, k, s);
            put<4, RG, NT>(rew, N, i, k, s >> 3);
            put<1, FG, NT>(term, N, i, k, s >> 8);
            put<1, FG, NT>(trunc, N, i, k, s >> 9);
        }
    }
}

// plain fills of one buffer: SHAPE 0 one 16-byte store per lane, one-shot grid; 1 four stores per lane 4 KiB apart (16 KiB per
// workgroup), one-shot grid; 2 the same tiles, persistent grid of 8 workgroups per CU (mdpp_probe_hbm's fill)
template <int SHAPE, bool NT>
__global__ __launch_bounds__(256) void k_fill(u32x4 *d, size_t n) {
    const u32x4 v = u32x4{threadIdx.x, blockIdx.x, 3u, 4u};
    if (SHAPE == 0) { const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) st<NT>(d + i, v); return; }
    const size_t ntile = n / 1024;
    for (size_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        const size_t i = t * 1024 + threadIdx.x;
        st<NT>(d + i, v); st<NT>(d + i + 256, v); st<NT>(d + i + 512, v); st<NT>(d + i + 768, v);
    }
}

// The rollout kernel's own shape without its arithmetic: 1024-thread workgroups, waves 0-3 ("E") load the block's actions 4 B per
// lane, AH chunks of 8 steps ahead, and publish a chunk counter; waves 8-11 ("O2") wait for it and write rows w and w + 4 of the
// chunk for the block's 256 envs (obs 2 x 16 B per lane, reward 16 B, flags 4 B each); the other eight waves leave at once.
template <int AH, bool NT>
__global__ __launch_bounds__(1024) void k_lean_like(const int32_t *act, uint8_t *obs, uint8_t *rew, uint8_t *term, uint8_t *trunc, int N, int K) {
    __shared__ uint32_t prod[4];
    __shared__ uint32_t ring[32][256];
    const int role = threadIdx.x >> 8, l = threadIdx.x & 255, w = l >> 6, ln = l & 63;
    const uint32_t eblk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int i = eblk * 256 + l;
    if (threadIdx.x < 4) prod[threadIdx.x] = 0;
    __syncthreads();
    if (role == 0) {
        int q[AH][8];
#pragma unroll
        for (int c = 0; c < AH; c++)
#pragma unroll
            for (int u = 0; u < 8; u++) q[c][u] = act[(size_t)(c * 8 + u) * N + i];
        uint32_t s = (uint32_t)i;
        for (int c0 = 0; c0 < K / 8; c0 += AH) {
#pragma unroll
            for (int j = 0; j < AH; j++) {
                const int c = c0 + j;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    s = s * 1664525u + 1013904223u + (uint32_t)q[j][u];
                    ring[(c * 8 + u) & 31][l] = s;
                    const int kn = (c + AH) * 8 + u;
                    q[j][u] = act[(size_t)(kn < K ? kn : K - 1) * N + i];
                }
                if (ln == 0) __hip_atomic_store(&prod[w], (uint32_t)(c + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        return;
    }
    if (role != 2) return;
    for (int c = 0; c < K / 8; c++) {
        for (;;) {
            uint32_t m = 0xFFFFFFFFu;
            for (int j = 0; j < 4; j++) { const uint32_t p = __hip_atomic_load(&prod[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); m = p < m ? p : m; }
            if (m >= (uint32_t)(c + 1)) break;
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int k = c * 8 + w + 4 * h;
            const size_t row = (size_t)k * N + eblk * 256;
            const uint32_t a0 = ring[k & 31][2 * ln], a1 = ring[k & 31][2 * ln + 1], b0 = ring[k & 31][128 + 2 * ln], b1 = ring[k & 31][129 + 2 * ln];
            st<NT>((u32x4 *)(obs + row * 8 + ln * 16), u32x4{a0 & 7u, 0u, a1 & 7u, 0u});
            st<NT>((u32x4 *)(obs + row * 8 + 1024 + ln * 16), u32x4{b0 & 7u, 0u, b1 & 7u, 0u});
            const u32x4 r4 = *(const u32x4 *)&ring[k & 31][4 * ln];
            st<NT>((u32x4 *)(rew + row * 4 + ln * 16), r4);
            st<NT>((uint32_t *)(term + row + ln * 4), (r4.x >> 7) & 0x01010101u);
            st<NT>((uint32_t *)(trunc + row + ln * 4), (r4.y >> 9) & 0x01010101u);
        }
    }
}

// The continuous rollout's shape (cfg3: D = 12 floats of action in, 12 floats of observation out per env step, one lane per env,
// one wave per SIMD, rows prefetched AH steps ahead): the action row of a lane is 48 contiguous bytes, so its three 16-byte loads
// sit 48 B apart across the lanes (STRIDED: every instruction touches all 24 cache lines of the wave's 3 KiB); CONTIG: the same
// 3 KiB as three instructions of 1 KiB each (what an LDS transposition of the loads would issue).  Stores: 1 KiB per instruction.
template <int AH, bool CONTIG>
__global__ __launch_bounds__(256) void k_cont_like(const u32x4 *act, u32x4 *obs, int N, int K) {
    const int i = blockIdx.x * 256 + threadIdx.x, ln = threadIdx.x & 63, w0 = i - ln;
    u32x4 q[AH][3];
    auto ld = [&](int k, u32x4 (&d)[3]) {
        const size_t row = (size_t)(k < K ? k : K - 1) * N * 3;
#pragma unroll
        for (int j = 0; j < 3; j++) d[j] = CONTIG ? act[row + (size_t)w0 * 3 + j * 64 + ln] : act[row + (size_t)i * 3 + j];
    };
#pragma unroll
    for (int u = 0; u < AH; u++) ld(u, q[u]);
    u32x4 acc = u32x4{0u, 0u, 0u, 0u};
    for (int k0 = 0; k0 < K; k0 += AH) {
#pragma unroll
        for (int u = 0; u < AH; u++) {
            const int k = k0 + u;
            acc = acc + q[u][0] + q[u][1] + q[u][2];
            ld(k + AH, q[u]);
            const size_t row = (size_t)k * N * 3 + (size_t)w0 * 3;
#pragma unroll
            for (int j = 0; j < 3; j++) __builtin_nontemporal_store(acc + (unsigned)j, obs + row + j * 64 + ln);
        }
    }
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static int N = 65536, K = 512;
static uint8_t *d_obs, *d_rew, *d_term, *d_trunc;
static int32_t *d_act[4];

template <int OG, int RG, int FG, bool NT, int PRE>
static void run(const char *label, bool rotate) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k<OG, RG, FG, NT, PRE>), dim3(N / 256), dim3(256), 0, 0, d_act[rotate ? w & 3 : 0], d_obs, d_rew, d_term, d_trunc, N, K);
        CHK(hipEventRecord(e0, 0));
        const int L = 20;
        for (int w = 0; w < L; w++) hipLaunchKernelGGL((k<OG, RG, FG, NT, PRE>), dim3(N / 256), dim3(256), 0, 0, d_act[rotate ? w & 3 : 0], d_obs, d_rew, d_term, d_trunc, N, K);
        CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / L < best) best = ms / L;
    }
    const double bytes = (double)N * K * (14.0 + (PRE > 0 ? 4.0 : 0.0));
    printf("  obs x%-2d rew x%-2d flags x%-2d %-7s %-22s %8.1f us per launch  %7.1f GB/s\n", OG, RG, FG, NT ? "nt" : "default", label, best * 1e3, bytes / (best * 1e-3) / 1e9);
    fflush(stdout);
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
}

template <int SHAPE, bool NT>
static void fill(u32x4 *d, size_t nbytes) {
    const size_t n = nbytes / 16;
    const unsigned grid = SHAPE == 0 ? (unsigned)(n / 256) : SHAPE == 1 ? (unsigned)(n / 1024) : 2048u;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_fill<SHAPE, NT>), dim3(grid), dim3(256), 0, 0, d, n);
    CHK(hipEventRecord(e0, 0));
    for (int w = 0; w < 10; w++) hipLaunchKernelGGL((k_fill<SHAPE, NT>), dim3(grid), dim3(256), 0, 0, d, n);
    CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("  fill %zu MiB, shape %d, %-7s %7.1f GB/s\n", nbytes >> 20, SHAPE, NT ? "nt" : "default", (double)nbytes * 10 / (ms * 1e-3) / 1e9);
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
}

template <int AH, bool NT>
static void lean_like(bool rotate) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k_lean_like<AH, NT>), dim3(N / 256), dim3(1024), 0, 0, d_act[rotate ? w & 3 : 0], d_obs, d_rew, d_term, d_trunc, N, K);
        CHK(hipEventRecord(e0, 0));
        const int L = 20;
        for (int w = 0; w < L; w++) hipLaunchKernelGGL((k_lean_like<AH, NT>), dim3(N / 256), dim3(1024), 0, 0, d_act[rotate ? w & 3 : 0], d_obs, d_rew, d_term, d_trunc, N, K);
        CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / L < best) best = ms / L;
    }
    printf("  rollout-shaped (E waves load %d chunks ahead, O2 waves store rows) %-7s %-18s %8.1f us per launch  %7.1f GB/s\n", AH, NT ? "nt" : "default",
           rotate ? "reads, rotating" : "reads, one tensor", best * 1e3, (double)N * K * 18.0 / (best * 1e-3) / 1e9);
    fflush(stdout);
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
}

template <int OG, int RG, int FG>
static void all() {
    run<OG, RG, FG, true, 0>("no reads", false);
    run<OG, RG, FG, false, 0>("no reads", false);
    run<OG, RG, FG, true, 8>("reads, one tensor", false);
    run<OG, RG, FG, true, 8>("reads, rotating", true);
    run<OG, RG, FG, false, 8>("reads, rotating", true);
}

int main() {
    const size_t tot = (size_t)N * K;
    CHK(hipMalloc(&d_obs, tot * 8)); CHK(hipMalloc(&d_rew, tot * 4)); CHK(hipMalloc(&d_term, tot)); CHK(hipMalloc(&d_trunc, tot));
    for (int j = 0; j < 4; j++) { CHK(hipMalloc(&d_act[j], tot * 4)); CHK(hipMemset(d_act[j], j, tot * 4)); }
    {
        u32x4 *big; CHK(hipMalloc(&big, (size_t)1 << 30));
        printf("fills (shape 0: one 16-byte store per lane, one-shot; 1: 16 KiB tiles, one-shot; 2: 16 KiB tiles, persistent grid)\n");
        fill<0, false>(big, (size_t)1 << 30); fill<0, true>(big, (size_t)1 << 30);
        fill<1, false>(big, (size_t)1 << 30); fill<1, true>(big, (size_t)1 << 30);
        fill<2, false>(big, (size_t)1 << 30); fill<2, true>(big, (size_t)1 << 30);
        CHK(hipFree(big));
    }
    printf("N %d envs, K %d rows per launch; x g = g elements per lane and store (g lanes x g steps transposed)\n", N, K);
    if (getenv("CONT_ONLY")) {
        u32x4 *ca[4], *co;
        const size_t rows = (size_t)N * K * 3;
        for (int j = 0; j < 4; j++) { CHK(hipMalloc(&ca[j], rows * 16)); CHK(hipMemset(ca[j], j + 1, rows * 16)); }
        CHK(hipMalloc(&co, rows * 16));
        auto run_c = [&](auto kern, const char *label) {
            hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kern, dim3(N / 256), dim3(256), 0, 0, ca[w & 3], co, N, K);
                CHK(hipEventRecord(e0, 0));
                for (int w = 0; w < 8; w++) hipLaunchKernelGGL(kern, dim3(N / 256), dim3(256), 0, 0, ca[w & 3], co, N, K);
                CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (ms / 8 < best) best = ms / 8;
            }
            printf("  continuous-shaped, %-34s %8.1f us per launch  %7.1f GB/s\n", label, best * 1e3, (double)N * K * 96.0 / (best * 1e-3) / 1e9);
            fflush(stdout);
        };
        for (int r = 0; r < 2; r++) {
            run_c(k_cont_like<4, false>, "strided loads, 4 rows ahead");
            run_c(k_cont_like<4, true>, "contiguous loads, 4 rows ahead");
            run_c(k_cont_like<8, false>, "strided loads, 8 rows ahead");
            run_c(k_cont_like<8, true>, "contiguous loads, 8 rows ahead");
            run_c(k_cont_like<2, false>, "strided loads, 2 rows ahead");
            run_c(k_cont_like<2, true>, "contiguous loads, 2 rows ahead");
        }
        return 0;
    }
    if (getenv("LEAN_ONLY")) {
        for (int r = 0; r < 2; r++) {
            lean_like<4, true>(true); lean_like<4, true>(false); lean_like<4, false>(true);
            lean_like<2, true>(true); lean_like<8, true>(true);
            all<0, 4, 4>();
        }
        return 0;
    }
    all<1, 1, 1>();
    all<1, 2, 1>();
    all<1, 4, 1>();
    all<1, 4, 4>();
    all<1, 4, 8>();
    all<1, 4, 16>();
    all<2, 4, 8>();
    all<2, 4, 16>();
    all<2, 2, 8>();
    all<1, 1, 8>();
    all<1, 1, 16>();
    all<4, 4, 16>();
    printf("obs x0 = the workgroup's 256 envs x 4 steps transposed: each wave writes one whole row piece\n");
    all<0, 4, 4>();
    return 0;
}

// Three-role pipelined rollout for the common discrete shape (the same shapes and the same
// arithmetic as k_discrete_rollout_fast in mdpp_discrete_fast.hip; reference
// rl_toy_env.py:1992-2125, reset :2250-2278).
//
// One wavefront per SIMD — all a 65 536-env job offers when a lane is an env — leaves about half
// of every SIMD's issue slots idle (measured: profiles/archive/r01_ablation_fast_kernel.txt).  Here a
// 768-thread workgroup steps 256 envs with three waves per SIMD, each doing a third of the work:
//   E  waves 0-3   state recurrence: cur -> next, sequence key, episode counters, terminal test,
//                  same-step autoreset from the queue of pre-drawn start states; one 32-bit
//                  record per env step into an LDS ring
//   O  waves 4-7   consume the records: reward-bitmask lookup, delay line, reward select, and ALL
//                  global stores (obs, reward, terminated, truncated, final_obs)
//   H  waves 8-11  own the envs' PCG64 streams for the launch and keep an LDS ring of pre-drawn
//                  rho_0 start states filled; un-draw what was not used at the end
// Lane l of waves w, w+4, w+8 serves the same env.  E->O and H->E hand-offs are single-producer /
// single-consumer rings in LDS with monotonic counters (release/acquire at workgroup scope,
// polled once per kChunk steps); every spin is bounded and sets MDPP_STATUS_INTERNAL instead of
// hanging.  Results are bit-identical to the single-role kernels (same tests).
#include <stdlib.h>

#include <stdio.h>

#include "mdpp_internal.hpp"
#include "mdpp_rng.hpp"

namespace mdpp {

#ifndef MDPP_PIPE_CHUNK
#define MDPP_PIPE_CHUNK 8
#endif
#ifndef MDPP_PIPE_DEPTH
#define MDPP_PIPE_DEPTH 32
#endif
constexpr int kChunk = MDPP_PIPE_CHUNK;   // steps between hand-off polls; also the action prefetch distance
constexpr int kDepth = MDPP_PIPE_DEPTH;   // E->O ring depth in steps (multiple of kChunk)
constexpr int kPRsrc = 0x00020000;
constexpr uint32_t kSpinLimit = 1u << 22;
constexpr uint32_t kStatusInternal = 0x80000000u;
typedef unsigned int pu32x2 __attribute__((ext_vector_type(2)));

// record layout (E -> O), one dword per env step
//   [3:0] observation (state after a possible reset)   [7:4] state reached (before reset)
//   [8] terminated  [9] truncated  [10] reset happened  [11] history full (NaN gate)
//   [12] pay step (steps % every_n == 0)                [31:13] sequence key
__device__ __forceinline__ uint32_t wg_load_acq(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void wg_store_rel(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <bool OBS64, bool POW2, bool DELAY, bool S8>
__global__ __launch_bounds__(3 * kBlock) void k_discrete_rollout_pipe(DiscreteArgs a, int K,
                                                                      const int32_t *__restrict__ actions,
                                                                      void *__restrict__ obs,
                                                                      float *__restrict__ reward,
                                                                      uint8_t *__restrict__ term,
                                                                      uint8_t *__restrict__ trunc,
                                                                      void *__restrict__ final_obs) {
    __shared__ __align__(16) uint32_t lds_rec[kDepth][kBlock]; // E -> O
    __shared__ __align__(16) uint64_t lds_col[16];  // column a of P: nibble s = P[s][a]
    __shared__ __align__(16) uint32_t lds_R[128];   // 4096 reward bits (16^3)
    __shared__ __align__(16) uint64_t lds_T[16];    // rho_0 thresholds
    __shared__ __align__(16) uint64_t lds_ring[kBlock]; // H -> E: {8 nibbles, #pushed}
    __shared__ uint32_t lds_head[kBlock];           // E -> H: #popped
    __shared__ uint32_t lds_prod[kBlock / 64];      // steps published by E wave w
    __shared__ uint32_t lds_cons[kBlock / 64];      // steps consumed by O wave w
    __shared__ uint32_t lds_done;                   // E waves that have finished
    const int tid = threadIdx.x;
    const int role = tid / kBlock;                  // 0 = E, 1 = O, 2 = H
    const int l = tid & (kBlock - 1);               // env slot inside the block
    const int w = l >> 6;                           // wave pair/triple index
    if (tid < 16) {
        uint64_t col = 0;
        if (tid < a.A)
            for (int s = 0; s < a.S; s++) col |= (uint64_t)(a.P[s * a.A + tid] & 0xF) << (4 * s);
        lds_col[tid] = col;
        lds_T[tid] = a.init_thr[tid];
    }
    for (uint32_t k = tid; k < 128; k += 3 * kBlock) {
        uint32_t wd = 0;
        for (int b = 0; b < 4; b++) {
            uint32_t byte = 4 * k + b;
            if (byte < a.rbits_stride) wd |= (uint32_t)a.rbits[byte] << (8 * b);
        }
        lds_R[k] = wd;
    }
    if (tid < kBlock) { lds_ring[tid] = 0; lds_head[tid] = 0; }
    if (tid < kBlock / 64) { lds_prod[tid] = 0; lds_cons[tid] = 0; }
    if (tid == 0) lds_done = 0;
    __syncthreads();

    const uint32_t i = blockIdx.x * kBlock + l;     // N % kBlock == 0 is a launch precondition
    const uint32_t N = (uint32_t)a.N;
    const uint32_t A = (uint32_t)a.A, S = (uint32_t)a.S, L = (uint32_t)a.L;
    const bool autoreset = a.autoreset != 0;
    const bool s_le_8 = S <= 8;
    constexpr uint32_t kQueueCap = 6;
    constexpr int kMinLanes = 16;
    uint32_t status = 0;

    // =============================================================== H: start-state producer
    if (role == 2) {
        Pcg64 g;
        g.load(a.env_s, a.env_inc, i);
        auto draw = [&](Pcg64 &gg) -> uint32_t {
            const uint64_t m = gg.next64() >> 11;
            uint32_t s0 = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) s0 += (lds_T[j] <= m) ? 1u : 0u;
            if (!s_le_8) {
#pragma unroll
                for (int j = 8; j < 16; j++) s0 += (lds_T[j] <= m) ? 1u : 0u;
            }
            return s0;
        };
        uint32_t vals = 0, tail = 0;
        for (;;) {
            if (wg_load_acq(&lds_done) == kBlock / 64) break;
            const uint32_t head = wg_load_acq(&lds_head[l]);
            const uint32_t cnt = tail - head;
            const bool want = autoreset && cnt < 8;
            const uint64_t bw = __builtin_amdgcn_ballot_w64(want);
            const bool urgent = __builtin_amdgcn_ballot_w64(want && cnt <= 2) != 0;
            if (__builtin_popcountll(bw) >= kMinLanes || urgent) {
                Pcg64 n = g;
                const uint32_t s0 = draw(n);
                if (want) {
                    const uint32_t sh = (tail & 7u) * 4u;
                    g = n;
                    vals = (vals & ~(0xFu << sh)) | (s0 << sh);
                    tail += 1;
                }
                __hip_atomic_store(&lds_ring[l], (uint64_t)vals | ((uint64_t)tail << 32), __ATOMIC_RELEASE,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                __builtin_amdgcn_s_sleep(4);
            }
        }
        // un-draw what the env lane did not take: s_prev = (s - inc) * M^-1 (mod 2^128)
        const uint32_t head = wg_load_acq(&lds_head[l]);
        for (uint32_t q = tail - head; q > 0; q--) {
            uint64_t lo = g.s_lo - g.inc_lo;
            uint64_t hi = g.s_hi - g.inc_hi - (g.s_lo < g.inc_lo ? 1ULL : 0ULL);
            g.s_lo = lo * a.minv_lo;
            g.s_hi = __umul64hi(lo, a.minv_lo) + lo * a.minv_hi + hi * a.minv_lo;
        }
        g.store(a.env_s, i);
        return;
    }

    const uint32_t total = (uint32_t)K * N;
    const int nchunks = (K + kChunk - 1) / kChunk;

    // =============================================================== O: outputs
    if (role == 1) {
        auto r_obs = __builtin_amdgcn_make_buffer_rsrc(obs, 0, total * (OBS64 ? 8u : 4u), kPRsrc);
        auto r_rew = __builtin_amdgcn_make_buffer_rsrc((void *)reward, 0, total * 4u, kPRsrc);
        auto r_term = __builtin_amdgcn_make_buffer_rsrc((void *)term, 0, total, kPRsrc);
        auto r_trunc = __builtin_amdgcn_make_buffer_rsrc((void *)trunc, 0, total, kPRsrc);
        auto r_fin = __builtin_amdgcn_make_buffer_rsrc(final_obs ? final_obs : obs, 0,
                                                       total * (OBS64 ? 8u : 4u), kPRsrc);
        const bool want_final = final_obs != nullptr;
        const uint32_t v1 = i, v4 = i * 4u, v8 = i * 8u;
        const uint32_t dsh = (uint32_t)(a.delay > 0 ? a.delay - 1 : 0);
        float rs0 = a.rsel[0], rs1 = a.rsel[1], rs2 = a.rsel[2], rs3 = a.rsel[3];
        asm volatile("" : "+v"(rs0), "+v"(rs1), "+v"(rs2), "+v"(rs3));
        uint32_t ring = ((const uint32_t *)&a.state[i])[3];

        auto emit = [&](uint32_t rec, uint32_t so) {
            const uint32_t key = rec >> 13;
            uint32_t bit = (lds_R[key >> 5] >> (key & 31u)) & (rec >> 11) & 1u;  // NaN gate (:1822)
            if (DELAY) {                                                             // FIFO (:1970-1973)
                const uint32_t out = (ring >> dsh) & 1u;
                ring = (ring << 1) | bit;
                bit = out;
            }
            bit &= rec >> 12;                                                        // every-n (:1975)
            const bool done = (rec & 0x100u) != 0;
            const float r_nt = bit ? rs2 : rs0;
            const float r_t = bit ? rs3 : rs1;
            const float rout = done ? r_t : r_nt;
            if (DELAY) ring = (rec & 0x400u) ? 0u : ring;                            // reset clears it (:2250)
            const uint32_t o = rec & 0xFu;
            if (__builtin_expect(want_final, 0)) {
                if (rec & 0x400u) {
                    const uint32_t nx = (rec >> 4) & 0xFu;
                    if (OBS64) __builtin_amdgcn_raw_buffer_store_b64(pu32x2{nx, 0u}, r_fin, v8, so * 8u, MDPP_ST_NT);
                    else __builtin_amdgcn_raw_buffer_store_b32(nx, r_fin, v4, so * 4u, MDPP_ST_NT);
                }
            }
#ifdef MDPP_ABL_NOSTORE
            status ^= (o + __float_as_uint(rout)) & 0x100u;
            return;
#endif
            if (OBS64) __builtin_amdgcn_raw_buffer_store_b64(pu32x2{o, 0u}, r_obs, v8, so * 8u, MDPP_ST_NT);
            else __builtin_amdgcn_raw_buffer_store_b32(o, r_obs, v4, so * 4u, MDPP_ST_NT);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rout), r_rew, v4, so * 4u, MDPP_ST_NT);
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)((rec >> 8) & 1u), r_term, v1, so, MDPP_ST_NT);
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)((rec >> 9) & 1u), r_trunc, v1, so, MDPP_ST_NT);
        };
        for (int c = 0; c < nchunks; c++) {
            const int kbase = c * kChunk;
            const uint32_t upto = (uint32_t)min(kbase + kChunk, K);
            uint32_t spins = 0;
            while (wg_load_acq(&lds_prod[w]) < upto) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit) { status |= kStatusInternal; break; }
            }
            if (kbase + kChunk <= K) {
                uint32_t rec[kChunk];
#pragma unroll
                for (int u = 0; u < kChunk; u++) rec[u] = lds_rec[(kbase + u) % kDepth][l];
#pragma unroll
                for (int u = 0; u < kChunk; u++) emit(rec[u], (uint32_t)(kbase + u) * N);
            } else {
                for (int k = kbase; k < K; k++) emit(lds_rec[k % kDepth][l], (uint32_t)k * N);
            }
            if ((l & 63) == 0) wg_store_rel(&lds_cons[w], upto);
        }
        ((uint32_t *)&a.state[i])[3] = ring;
        if (status) atomicOr(&a.status[i], status);
        return;
    }

    // =============================================================== E: state recurrence
    uint32_t hist, key, cur, steps, phase, qv, qc;
    {
        uint4 st = a.state[i];
        hist = st.x;
        qv = st.y & 0x00FFFFFFu; qc = (st.y >> 24) & 7u;
        steps = st.z;
        phase = steps % (uint32_t)a.every_n;
        cur = hist & 0xFFu;
        key = 0;
        for (int j = (int)L - 1; j >= 0; j--) {
            uint32_t b = (hist >> (8 * j)) & 0xFFu;
            key = key * S + (b == 0xFFu ? 0u : b);
        }
    }
    auto r_act = __builtin_amdgcn_make_buffer_rsrc((void *)actions, 0, total * 4u, kPRsrc);
    const uint32_t v4 = i * 4u;
    const bool has_max = a.max_steps > 0;
    const uint32_t every_n = (uint32_t)a.every_n, max_steps = (uint32_t)a.max_steps;
    const uint32_t term32 = (uint32_t)a.term_mask;
    const uint32_t nan_mask = 0xFFu << (8 * L);
    uint32_t head_local = 0;

    auto pull = [&]() {
        const uint64_t rt = __hip_atomic_load(&lds_ring[l], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t vals = (uint32_t)rt, tail = (uint32_t)(rt >> 32);
        const uint32_t avail = tail - head_local, room = kQueueCap - qc;
        const uint32_t take = avail < room ? avail : room;
        const uint32_t rot = __builtin_amdgcn_alignbit(vals, vals, (head_local & 7u) * 4u);
        const uint32_t m = (1u << (4u * take)) - 1u;
        qv |= (rot & m) << (4u * qc);
        qc += take;
        head_local += take;
        __hip_atomic_store(&lds_head[l], head_local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto column = [&](int action) -> uint64_t {
        uint32_t ua = (uint32_t)action;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(ua >= A) != 0, 0)) {
            ua = (uint32_t)(action + ((action >> 31) & (int)A));
            const bool bad = ua >= A;
            status |= bad ? (uint32_t)MDPP_STATUS_BAD_ACTION : 0u;
            ua = bad ? 0u : ua;
        }
        if (S8) return (uint64_t)((const uint32_t *)lds_col)[2 * ua];
        return lds_col[ua];
    };
    auto stepE = [&](uint64_t col, int k) {
        const uint32_t nxt = S8 ? (((uint32_t)col >> (cur << 2)) & 0xFu)
                                : (uint32_t)((col >> (cur << 2)) & 0xFu);              // D1
        if (POW2) {
            key = ((key << a.s_shift) | nxt) & a.key_mask;                            // D3 / D4 key
        } else {
            uint32_t old = (hist >> (8 * (L - 1))) & 0xFFu;
            old = (old == 0xFFu) ? 0u : old;
            key = (key - old * a.spow) * S + nxt;
        }
        hist = (hist << 8) | nxt;
        steps += 1;
        phase = (phase + 1 == every_n) ? 0u : phase + 1;
        const bool full = (hist & nan_mask) != nan_mask;
        const bool pay = phase == 0;
        const uint32_t done = (term32 >> nxt) & 1u;                                   // D7
        const uint32_t tr = (has_max && steps >= max_steps) ? 1u : 0u;
#ifdef MDPP_ABL_NORESET
        const bool need = false;
#else
        const bool need = autoreset && ((done | tr) != 0);
#endif
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(need && qc == 0) != 0, 0)) {
            uint32_t spins = 0;
            while (__builtin_amdgcn_ballot_w64(need && qc == 0) != 0) {
                pull();
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit) { status |= kStatusInternal; qc = 1; break; }
            }
        }
        const uint32_t s0 = qv & 0xFu;
        uint32_t rec = (key << 13) | (pay ? 0x1000u : 0u) | (full ? 0x800u : 0u) | (need ? 0x400u : 0u) |
                       (tr << 9) | (done << 8) | (nxt << 4);
        cur = need ? s0 : nxt;
        rec |= cur;
        hist = need ? (0xFFFFFF00u | s0) : hist;
        key = need ? s0 : key;
        steps = need ? 0u : steps;
        phase = need ? 0u : phase;
        qv = need ? (qv >> 4) : qv;
        qc = qc - (need ? 1u : 0u);
        lds_rec[k % kDepth][l] = rec;
    };

    auto load_act = [&](int k) -> int {
        const uint32_t kk = (uint32_t)min(k, K - 1);
        return __builtin_amdgcn_raw_buffer_load_b32(r_act, v4, kk * N * 4u, 0);
    };
    // Actions are fetched kPipeAhead chunks ahead of their use, columns one chunk ahead.  The buffers rotate by
    // NAME (chunk loop unrolled kPipeAhead times, single exit, no global load in an inner loop): a copy of a
    // register whose load is in flight makes the wave wait for the load, and a `break` in the unrolled body makes
    // it wait for every load at the loop head (found on k_discrete_rollout_lean, profiles/archive/r02_ablation_lean_kernel.txt)
    constexpr int kPipeAhead = 2;
    int actq[kPipeAhead][kChunk];   // slot (m - 1) % kPipeAhead holds the actions of chunk m
    uint64_t colq[2][kChunk];       // slot m & 1 holds the columns of chunk m
#pragma unroll
    for (int u = 0; u < kChunk; u++) colq[0][u] = column(load_act(u));
#pragma unroll
    for (int q = 0; q < kPipeAhead; q++)
#pragma unroll
        for (int u = 0; u < kChunk; u++) actq[q][u] = load_act((q + 1) * kChunk + u);
    auto wait_room = [&](int kend) {                // do not run more than kDepth steps ahead of the O wave
        if (kend > kDepth) {
            const uint32_t must = (uint32_t)(kend - kDepth);
            uint32_t spins = 0;
            while (wg_load_acq(&lds_cons[w]) < must) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit) { status |= kStatusInternal; break; }
            }
        }
    };
    auto chunkE = [&](int c, int j, bool refill) {
        const int kbase = c * kChunk;
#pragma unroll
        for (int u = 0; u < kChunk; u++) colq[(j + 1) & 1][u] = column(actq[j][u]);
        if (refill) {
#pragma unroll
            for (int u = 0; u < kChunk; u++) actq[j][u] = load_act(kbase + (kPipeAhead + 1) * kChunk + u);
        }
        wait_room(kbase + kChunk);
        if (autoreset) pull();
#pragma unroll
        for (int u = 0; u < kChunk; u++) stepE(colq[j & 1][u], kbase + u);
        if ((l & 63) == 0) wg_store_rel(&lds_prod[w], (uint32_t)(kbase + kChunk));
    };
    const int nfull = K / kChunk, ngrp = nfull / kPipeAhead;
    for (int g = 0; g < ngrp; g++) {
#pragma unroll
        for (int j = 0; j < kPipeAhead; j++) chunkE(g * kPipeAhead + j, j, true);
    }
#pragma unroll
    for (int j = 0; j < kPipeAhead - 1; j++)
        if (ngrp * kPipeAhead + j < nfull) chunkE(ngrp * kPipeAhead + j, j, false);
    if (K % kChunk) {               // the ragged tail, outside the loop
        const int kbase = nfull * kChunk;
        int ta[kChunk];
#pragma unroll
        for (int u = 0; u < kChunk; u++) ta[u] = load_act(kbase + u);
        wait_room(kbase + kChunk);
        if (autoreset) pull();
#pragma unroll
        for (int u = 0; u < kChunk; u++)
            if (kbase + u < K) stepE(column(ta[u]), kbase + u);
        if ((l & 63) == 0) wg_store_rel(&lds_prod[w], (uint32_t)K);
    }

    uint32_t *st = (uint32_t *)&a.state[i];
    st[0] = hist; st[1] = qv | (qc << 24); st[2] = steps;   // word 3 (delay line) belongs to the O lane
    if ((l & 63) == 0) __hip_atomic_fetch_add(&lds_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (status) atomicOr(&a.status[i], status);
}

// Returns false when the shape does not qualify (caller uses k_discrete_rollout_fast).
bool launch_discrete_pipe(const DiscreteArgs &a, int K, const int32_t *actions, void *obs,
                          float *reward, uint8_t *term, uint8_t *trunc, void *final_obs,
                          hipStream_t s, char *name_out) {
    if (!a.fast_ok || K < 32 || (a.N % kBlock) != 0 || !a.autoreset || (a.opts & MDPP_OPT_NO_PIPE)) return false;
    const int grid = a.N / kBlock;
    const bool pow2 = a.s_shift != 0xFFFFFFFFu, dl = a.delay > 0, s8 = a.S <= 8;
    if (name_out) {
        snprintf(name_out, kNameLen, "k_discrete_rollout_pipe<OBS64=%d,POW2=%d,DELAY=%d,S8=%d>", !a.obs_i32, pow2, dl, s8);
        return true;
    }
#define MDPP_PIPE_LAUNCH(O64, P2, DL, S8)                                                         \
    hipLaunchKernelGGL((k_discrete_rollout_pipe<O64, P2, DL, S8>), dim3(grid), dim3(3 * kBlock), \
                       0, s, a, K, actions, obs, reward, term, trunc, final_obs)
#define MDPP_PIPE_L3(O64, P2, DL) do { if (s8) MDPP_PIPE_LAUNCH(O64, P2, DL, true); else MDPP_PIPE_LAUNCH(O64, P2, DL, false); } while (0)
#define MDPP_PIPE_L2(O64, P2) do { if (dl) MDPP_PIPE_L3(O64, P2, true); else MDPP_PIPE_L3(O64, P2, false); } while (0)
    if (a.obs_i32) { if (pow2) MDPP_PIPE_L2(false, true); else MDPP_PIPE_L2(false, false); }
    else { if (pow2) MDPP_PIPE_L2(true, true); else MDPP_PIPE_L2(true, false); }
#undef MDPP_PIPE_L2
#undef MDPP_PIPE_L3
#undef MDPP_PIPE_LAUNCH
    return true;
}

} // namespace mdpp

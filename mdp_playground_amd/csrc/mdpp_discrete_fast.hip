// Fused K-step rollout for the common discrete shape (BASELINE cfg 1/2): one shared MDP, unit
// rewards, no noise, sequence_length <= 3, S <= 16, numpy PCG64 streams, same-step autoreset.
// Same arithmetic as k_discrete_step (mdpp_discrete.hip; reference rl_toy_env.py:1992-2125 and
// reset :2250-2278) restructured for a chip that holds exactly ONE wavefront per SIMD at 65 536
// envs, i.e. no thread-level parallelism to hide anything behind:
//   * the state recurrence cur -> next is three VALU instructions and touches no memory: the
//     action of step k is known kAhead..2*kAhead steps early, so the COLUMN P[:, a_k] (S states x
//     4 bits = one or two dwords) is fetched from LDS ahead of time and the step only extracts
//     nibble `cur` from it;
//   * the reward-bitmask lookup (LDS, address depends on the new state) is issued in phase A of
//     step k and consumed in phase B, which runs after phase A of step k+1: its latency overlaps
//     the next step's work instead of stalling the only resident wave;
//   * straight-line, branch-free step body (a taken branch costs an unhidden refetch); the rare
//     paths (start-state queue ran dry, caller wants final_obs, action out of range) are
//     wave-uniform, marked unlikely and sit out of line;
//   * no float64 and no integer divide in the loop: the four possible rewards are formed on the
//     host in the reference's float64 order; `steps % every_n` and the sequence key are carried
//     incrementally; rho_0 sampling compares the raw 53-bit draw with host-made integer
//     thresholds ceil(cdf * 2^53)  (cdf[j] <= u  <=>  thr[j] <= r >> 11, exact);
//   * every global access is a buffer instruction: wave-uniform descriptor + per-step SGPR offset
//     + one per-lane VGPR offset that never changes, so no 64-bit address arithmetic per store;
//   * rho_0 draws are made AHEAD of need into a 6-deep per-env queue (4 bits per start state,
//     kept in word 1 of the state record): with 2 of 8 states terminal some lane of a wave
//     resets on almost every step, and a PCG64 step is ~14 quarter-rate 32-bit multiplies, so
//     drawing inside the step would run that code every step at ~25 % lane utilisation.  The
//     queue is topped up once per kAhead steps in rounds with most lanes active.  Draws are
//     consumed in stream order and nothing else reads the env stream on this path, so every env
//     still sees exactly the variates the reference's reset() would draw; mdpp_get_streams
//     rewinds the stream by the number of queued draws so the reported PCG64 state equals the
//     reference's.
// HBM traffic per env step: 4 B action in; 8 B obs + 4 B reward + 1 B + 1 B flags out.
#include <stdlib.h>

#include <stdio.h>

#include "mdpp_internal.hpp"
#include "mdpp_rng.hpp"

namespace mdpp {

constexpr int kAhead = 8;
constexpr int kRsrcFlags = 0x00020000;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Per-lane registers of one env instance during a rollout.
struct FastLane {
    uint32_t hist, key, cur, steps, phase, ring, status;
    uint32_t qv, qc; // queued start states (4 bits each, next one in bits 0-3) and their count
    Pcg64 g;
};
// What phase A of a step hands to its phase B.
struct Pending {
    uint32_t word;  // reward-bitmask dword (LDS read in flight)
    uint32_t sh;    // bit position inside it
    uint32_t flags; // bit0 history full (NaN gate), bit1 terminal, bit2 reset happened, bit3 pay step
    uint32_t so;    // element offset of the step's output row
};

// S8: S <= 8, a column of P fits one dword (8 x 4 bits); otherwise two (16 x 4 bits).
// HELPER: 512-thread workgroups; waves 0-3 step the 256 envs of the block, waves 4-7 (one per
// SIMD, next to an env wave) own the envs' PCG64 streams for the duration of the launch and keep
// an 8-deep ring of pre-drawn start states per env in LDS.  One wave per SIMD uses barely half of
// a SIMD's issue slots, so the draws run "for free" beside the env waves instead of inside them.
// Unconsumed ring entries are un-drawn at the end of the launch (inverse LCG step), so the stream
// position is again exactly the reference's.  Used for long rollouts of full blocks only.
template <bool OBS64, bool POW2, bool DELAY, bool S8, bool HELPER>
__global__ __launch_bounds__(HELPER ? 2 * kBlock : kBlock) void k_discrete_rollout_fast(DiscreteArgs a, int K,
                                                                  const int32_t *__restrict__ actions,
                                                                  void *__restrict__ obs,
                                                                  float *__restrict__ reward,
                                                                  uint8_t *__restrict__ term,
                                                                  uint8_t *__restrict__ trunc,
                                                                  void *__restrict__ final_obs) {
    __shared__ __align__(16) uint64_t lds_col[16]; // column a of P: nibble s = P[s][a]
    __shared__ __align__(16) uint32_t lds_R[128];  // 4096 reward bits (16^3)
    __shared__ __align__(16) uint64_t lds_T[16];   // rho_0 thresholds (read only by refill rounds)
    __shared__ __align__(16) uint64_t lds_ring[kBlock]; // HELPER: {8 nibbles, #pushed}, written by helper lanes
    __shared__ uint32_t lds_head[kBlock];               // HELPER: #popped, written by env lanes
    __shared__ uint32_t lds_done;                       // HELPER: env waves that have finished
    const int tid = threadIdx.x;
    if (HELPER) {
        if (tid < kBlock) { lds_ring[tid] = 0; lds_head[tid] = 0; }
        if (tid == 0) lds_done = 0;
    }
    if (tid < 16) {
        uint64_t col = 0;
        if (tid < a.A)
            for (int s = 0; s < a.S; s++) col |= (uint64_t)(a.P[s * a.A + tid] & 0xF) << (4 * s);
        lds_col[tid] = col;
        lds_T[tid] = a.init_thr[tid];
    }
    for (uint32_t k = tid; k < 128; k += blockDim.x) {
        uint32_t w = 0;
        for (int b = 0; b < 4; b++) {
            uint32_t byte = 4 * k + b;
            if (byte < a.rbits_stride) w |= (uint32_t)a.rbits[byte] << (8 * b);
        }
        lds_R[k] = w;
    }
    __syncthreads();
#ifdef MDPP_ABL_HALFWAVE
    if (tid & 32) return;   // experiment: 32 active lanes per wave, twice the waves
    const uint32_t i = blockIdx.x * (kBlock / 2) + (tid >> 6) * 32 + (tid & 31);
#else
    const uint32_t i = blockIdx.x * kBlock + (HELPER ? (tid & (kBlock - 1)) : tid);
#endif
    if (i >= (uint32_t)a.N) return;
    const uint32_t N = (uint32_t)a.N;
    const uint32_t A = (uint32_t)a.A, S = (uint32_t)a.S, L = (uint32_t)a.L;

    FastLane e;
    {
        uint4 st = a.state[i];
        e.hist = st.x; // L <= 3: the 4 live history bytes
        e.qv = st.y & 0x00FFFFFFu; e.qc = (st.y >> 24) & 7u;
        e.steps = st.z; e.ring = st.w; e.status = 0;
        e.phase = e.steps % (uint32_t)a.every_n;
        e.cur = e.hist & 0xFFu;
        // sequence key over the valid (non-NaN) part of the history
        e.key = 0;
        for (int j = (int)L - 1; j >= 0; j--) {
            uint32_t b = (e.hist >> (8 * j)) & 0xFFu;
            e.key = e.key * S + (b == 0xFFu ? 0u : b);
        }
        e.g.load(a.env_s, a.env_inc, i);
    }

    const uint32_t total = (uint32_t)K * N;
    auto r_act = __builtin_amdgcn_make_buffer_rsrc((void *)actions, 0, total * 4u, kRsrcFlags);
    auto r_obs = __builtin_amdgcn_make_buffer_rsrc(obs, 0, total * (OBS64 ? 8u : 4u), kRsrcFlags);
    auto r_rew = __builtin_amdgcn_make_buffer_rsrc((void *)reward, 0, total * 4u, kRsrcFlags);
    auto r_term = __builtin_amdgcn_make_buffer_rsrc((void *)term, 0, total, kRsrcFlags);
    auto r_trunc = __builtin_amdgcn_make_buffer_rsrc((void *)trunc, 0, total, kRsrcFlags);
    auto r_fin = __builtin_amdgcn_make_buffer_rsrc(final_obs ? final_obs : obs, 0,
                                                   total * (OBS64 ? 8u : 4u), kRsrcFlags);
    const bool want_final = final_obs != nullptr;
    const uint32_t v1 = i, v4 = i * 4u, v8 = i * 8u;
    const uint32_t dsh = (uint32_t)(a.delay > 0 ? a.delay - 1 : 0);
    const bool has_max = a.max_steps > 0, autoreset = a.autoreset != 0;
    const uint32_t every_n = (uint32_t)a.every_n, max_steps = (uint32_t)a.max_steps;
    const bool s_le_8 = S <= 8;
    const uint32_t term32 = (uint32_t)a.term_mask; // S <= 16
    const uint32_t nan_mask = 0xFFu << (8 * L);    // history byte L is the NaN test (:1822)
    // the four possible rewards, held in VGPRs so that each select is one v_cndmask
    float rs0 = a.rsel[0], rs1 = a.rsel[1], rs2 = a.rsel[2], rs3 = a.rsel[3];
    asm volatile("" : "+v"(rs0), "+v"(rs1), "+v"(rs2), "+v"(rs3));

    // reset(): self._np_random.choice(S, p=rho_0) == #{j : cdf[j] <= u} with u = (r >> 11) * 2^-53;
    // init_thr[j] = ceil(cdf[j] * 2^53), padded with 2^64-1 beyond S (never <= r >> 11)
    auto draw = [&](Pcg64 &g) -> uint32_t {
        const uint64_t m = g.next64() >> 11;
        uint32_t s0 = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) s0 += (lds_T[j] <= m) ? 1u : 0u;
        if (!s_le_8) {
#pragma unroll
            for (int j = 8; j < 16; j++) s0 += (lds_T[j] <= m) ? 1u : 0u;
        }
        return s0;
    };
    // Top the queues up: rounds in which every lane with a free slot draws one more start state,
    // repeated while at least kMinLanes lanes still want one (a round costs the same however few
    // lanes take part; stragglers catch up in a later chunk or, rarely, draw in place).
    constexpr uint32_t kQueueCap = 6;
    constexpr int kMinLanes = 16;
    auto refill = [&]() {
        if (!autoreset) return;
        for (int r = 0; r < (int)kQueueCap; r++) {
            const bool want = e.qc < kQueueCap;
            if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(want)) < kMinLanes) break;
            Pcg64 n = e.g;
            const uint32_t s0 = draw(n);
            if (want) { e.g = n; e.qv |= s0 << (4u * e.qc); e.qc += 1; }
        }
    };

    if (HELPER && tid >= kBlock) {
        // ---------------- helper lane: producer of start states for env (tid - kBlock) --------
        const int l = tid - kBlock;
        uint32_t vals = 0, tail = 0;
        for (;;) {
            if (__hip_atomic_load(&lds_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == kBlock / 64) break;
            const uint32_t head = __hip_atomic_load(&lds_head[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t cnt = tail - head;
            const bool want = autoreset && cnt < 8;
            const uint64_t bw = __builtin_amdgcn_ballot_w64(want);
            const bool urgent = __builtin_amdgcn_ballot_w64(want && cnt <= 2) != 0;
            if (__builtin_popcountll(bw) >= kMinLanes || urgent) {
                Pcg64 n = e.g;
                const uint32_t s0 = draw(n);
                if (want) {
                    const uint32_t sh = (tail & 7u) * 4u;
                    e.g = n;
                    vals = (vals & ~(0xFu << sh)) | (s0 << sh);
                    tail += 1;
                }
                __hip_atomic_store(&lds_ring[l], (uint64_t)vals | ((uint64_t)tail << 32), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                __builtin_amdgcn_s_sleep(4);
            }
        }
        // un-draw what the env lane did not take: s_prev = (s - inc) * M^-1 (mod 2^128)
        const uint32_t head = __hip_atomic_load(&lds_head[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (uint32_t q = tail - head; q > 0; q--) {
            uint64_t lo = e.g.s_lo - e.g.inc_lo;
            uint64_t hi = e.g.s_hi - e.g.inc_hi - (e.g.s_lo < e.g.inc_lo ? 1ULL : 0ULL);
            e.g.s_lo = lo * a.minv_lo;
            e.g.s_hi = __umul64hi(lo, a.minv_lo) + lo * a.minv_hi + hi * a.minv_lo;
        }
        e.g.store(a.env_s, i);
        return;
    }
    uint32_t head_local = 0;
    // env lane: move start states from the helper's ring into the register queue
    auto pull = [&]() {
        const uint64_t rt = __hip_atomic_load(&lds_ring[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t vals = (uint32_t)rt, tail = (uint32_t)(rt >> 32);
        const uint32_t avail = tail - head_local, room = kQueueCap - e.qc;
        const uint32_t take = avail < room ? avail : room;
        const uint32_t rot = __builtin_amdgcn_alignbit(vals, vals, (head_local & 7u) * 4u); // rotate right
        const uint32_t m = (1u << (4u * take)) - 1u;
        e.qv |= (rot & m) << (4u * e.qc);
        e.qc += take;
        head_local += take;
        __hip_atomic_store(&lds_head[tid], head_local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    // action -> column of P, off the critical path (numpy negative indexing; anything else out
    // of range is flagged and treated as action 0)
    auto column = [&](int action) -> uint64_t {
        uint32_t ua = (uint32_t)action;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(ua >= A) != 0, 0)) {
            ua = (uint32_t)(action + ((action >> 31) & (int)A));
            const bool bad = ua >= A;
            e.status |= bad ? (uint32_t)MDPP_STATUS_BAD_ACTION : 0u;
            ua = bad ? 0u : ua;
        }
        if (S8) return (uint64_t)((const uint32_t *)lds_col)[2 * ua];
        return lds_col[ua];
    };

    // ---- phase A of a step: everything that does not need the reward bit -------------------
    auto stepA = [&](uint64_t col, uint32_t so) -> Pending {
#ifdef MDPP_ABL_NOLDSP
        const uint32_t nxt = (e.cur + (uint32_t)col) & 7u;
#else
        const uint32_t nxt = S8 ? (((uint32_t)col >> (e.cur << 2)) & 0xFu)            // D1
                                : (uint32_t)((col >> (e.cur << 2)) & 0xFu);
#endif
        if (POW2) {                                                                   // D3 / D4 key
            e.key = ((e.key << a.s_shift) | nxt) & a.key_mask;
        } else {
            uint32_t old = (e.hist >> (8 * (L - 1))) & 0xFFu;
            old = (old == 0xFFu) ? 0u : old;
            e.key = (e.key - old * a.spow) * S + nxt;
        }
        e.hist = (e.hist << 8) | nxt;
        e.steps += 1;
        e.phase = (e.phase + 1 == every_n) ? 0u : e.phase + 1;
        Pending p;
#ifdef MDPP_ABL_NOLDSR
        p.word = e.key;
#else
        p.word = lds_R[e.key >> 5];                                                    // in flight
#endif
        p.sh = e.key & 31u;
        p.so = so;
        const bool full = (e.hist & nan_mask) != nan_mask;   // L transitions since reset (:1822)
        const bool pay = e.phase == 0;                       // steps % every_n == 0 (:1975)
        const uint32_t done = (term32 >> nxt) & 1u;                                   // D7
        const uint32_t tr = (has_max && e.steps >= max_steps) ? 1u : 0u;
#ifdef MDPP_ABL_NORESET
        const bool need = false;
#else
        const bool need = autoreset && ((done | tr) != 0);
#endif
        p.flags = (full ? 1u : 0u) | (done << 1) | (need ? 4u : 0u) | (pay ? 8u : 0u);
        // queue ran dry (rare): draw in place
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(need && e.qc == 0) != 0, 0)) {
            if (HELPER) {
                while (__builtin_amdgcn_ballot_w64(need && e.qc == 0) != 0) { pull(); __builtin_amdgcn_s_sleep(1); }
            } else {
                Pcg64 n = e.g;
                const uint32_t sd = draw(n);
                if (need && e.qc == 0) { e.g = n; e.qv = sd; e.qc = 1; }
            }
        }
        if (__builtin_expect(want_final, 0)) {
            if (need) {
                if (OBS64) __builtin_amdgcn_raw_buffer_store_b64(u32x2{nxt, 0u}, r_fin, v8, so * 8u, MDPP_ST_NT);
                else __builtin_amdgcn_raw_buffer_store_b32(nxt, r_fin, v4, so * 4u, MDPP_ST_NT);
            }
        }
        {   // same-step autoreset (reset(), :2250-2278) as selects
            const uint32_t s0 = e.qv & 0xFu;
            e.cur = need ? s0 : nxt;
            e.hist = need ? (0xFFFFFF00u | s0) : e.hist;
            e.key = need ? s0 : e.key;
            e.steps = need ? 0u : e.steps;
            e.phase = need ? 0u : e.phase;
            e.qv = need ? (e.qv >> 4) : e.qv;
            e.qc = e.qc - (need ? 1u : 0u);
        }
#ifndef MDPP_ABL_NOSTORE
        if (OBS64) __builtin_amdgcn_raw_buffer_store_b64(u32x2{e.cur, 0u}, r_obs, v8, so * 8u, MDPP_ST_NT);
        else __builtin_amdgcn_raw_buffer_store_b32(e.cur, r_obs, v4, so * 4u, MDPP_ST_NT);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)done, r_term, v1, so, MDPP_ST_NT);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)tr, r_trunc, v1, so, MDPP_ST_NT);
#else
        e.status ^= (e.cur + done + tr) & 0x100u;
#endif
        return p;
    };
    // ---- phase B: reward bit -> delay line -> reward.  The reference applies, in this order,
    // the NaN gate (:1822, on the entering reward), the FIFO (:1970-1973), the every-n mask on
    // the popped value (:1975-1978), the affine map and the terminal bonus (:1987-1990, :2107);
    // reset() then clears the FIFO (:2250).
    auto stepB = [&](const Pending &p) {
        uint32_t bit = (p.word >> p.sh) & p.flags & 1u;
        if (DELAY) {
            const uint32_t out = (e.ring >> dsh) & 1u;
            e.ring = (e.ring << 1) | bit;
            bit = out;
        }
        bit = (p.flags & 8u) ? bit : 0u;
        const bool done = (p.flags & 2u) != 0;
        const float r_nt = bit ? rs2 : rs0;
        const float r_t = bit ? rs3 : rs1;
        const float rout = done ? r_t : r_nt;
        if (DELAY) e.ring = (p.flags & 4u) ? 0u : e.ring;
#ifndef MDPP_ABL_NOSTORE
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rout), r_rew, v4, p.so * 4u, MDPP_ST_NT);
#else
        e.status ^= __float_as_uint(rout) & 0x100u;
#endif
    };

    // Software pipeline: at the top of chunk c the action loads of chunk c+2 are issued, the
    // columns of chunk c+1 are fetched from LDS (their actions arrived a chunk ago), and the
    // kAhead steps of chunk c run on columns fetched during chunk c-1.  Indices are clamped, never
    // predicated, so the chunk body is straight-line code with statically counted waits.
    auto load_act = [&](int k) -> int {
        const uint32_t kk = (uint32_t)min(k, K - 1);
        return __builtin_amdgcn_raw_buffer_load_b32(r_act, v4, kk * N * 4u, 0);
    };
    int act1[kAhead];          // actions of the next chunk
    uint64_t col0[kAhead];     // columns of the current chunk
#pragma unroll
    for (int u = 0; u < kAhead; u++) act1[u] = load_act(u);
#pragma unroll
    for (int u = 0; u < kAhead; u++) col0[u] = column(act1[u]);
#pragma unroll
    for (int u = 0; u < kAhead; u++) act1[u] = load_act(kAhead + u);

    const int nfull = K / kAhead;
    Pending pend = {0u, 0u, 0u, 0u};
    bool have_pend = false;
    for (int c = 0; c < nfull; c++) {
        const int kbase = c * kAhead;
        int act2[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; u++) act2[u] = load_act(kbase + 2 * kAhead + u);
        uint64_t col1[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; u++) col1[u] = column(act1[u]);
#ifndef MDPP_ABL_NOREFILL
        if (HELPER) pull(); else refill();
#endif
        if (c > 0) { // peeled so that the chunk body has no per-step branch
            Pending p = stepA(col0[0], (uint32_t)kbase * N);
            stepB(pend);
            pend = p;
        } else {
            pend = stepA(col0[0], (uint32_t)kbase * N);
        }
#pragma unroll
        for (int u = 1; u < kAhead; u++) {
            Pending p = stepA(col0[u], (uint32_t)(kbase + u) * N);
            stepB(pend);
            pend = p;
        }
        have_pend = true;
#pragma unroll
        for (int u = 0; u < kAhead; u++) { col0[u] = col1[u]; act1[u] = act2[u]; }
    }
    for (int k = nfull * kAhead; k < K; k++) { // tail: col0[] holds exactly these steps
        Pending p = stepA(col0[k - nfull * kAhead], (uint32_t)k * N);
        if (have_pend) stepB(pend);
        pend = p; have_pend = true;
    }
    if (have_pend) stepB(pend);

    a.state[i] = make_uint4(e.hist, e.qv | (e.qc << 24), e.steps, e.ring);
    if (HELPER) { // the helper lane holds (and stores) the stream; tell it this wave is done
        if ((tid & 63) == 0) __hip_atomic_fetch_add(&lds_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        e.g.store(a.env_s, i);
    }
    if (e.status) atomicOr(&a.status[i], e.status);
}

// Returns false when the shape does not qualify (caller falls back to k_discrete_step).
bool launch_discrete_fast(const DiscreteArgs &a, int K, const int32_t *actions, void *obs,
                          float *reward, uint8_t *term, uint8_t *trunc, void *final_obs,
                          hipStream_t s, char *name_out) {
    if (!a.fast_ok) return false;
#ifdef MDPP_ABL_HALFWAVE
    const int grid = (a.N + kBlock / 2 - 1) / (kBlock / 2);
#else
    const int grid = (a.N + kBlock - 1) / kBlock;
#endif
    const bool pow2 = a.s_shift != 0xFFFFFFFFu, dl = a.delay > 0, s8 = a.S <= 8;
    // helper waves pay off on long rollouts of full 256-env blocks
    const bool helper = K >= 32 && (a.N % kBlock) == 0 && a.autoreset && !(a.opts & MDPP_OPT_NO_HELPER);
    if (name_out) {
        snprintf(name_out, kNameLen, "k_discrete_rollout_fast<OBS64=%d,POW2=%d,DELAY=%d,S8=%d,HELPER=%d>", !a.obs_i32, pow2, dl, s8, helper);
        return true;
    }
#define MDPP_FAST_LAUNCH(O64, P2, DL, S8, HP)                                                   \
    hipLaunchKernelGGL((k_discrete_rollout_fast<O64, P2, DL, S8, HP>), dim3(grid),             \
                       dim3(HP ? 2 * kBlock : kBlock), 0, s, a, K, actions, obs, reward, term, \
                       trunc, final_obs)
#define MDPP_FAST_L4(O64, P2, DL, S8) do { if (helper) MDPP_FAST_LAUNCH(O64, P2, DL, S8, true); else MDPP_FAST_LAUNCH(O64, P2, DL, S8, false); } while (0)
#define MDPP_FAST_L3(O64, P2, DL) do { if (s8) MDPP_FAST_L4(O64, P2, DL, true); else MDPP_FAST_L4(O64, P2, DL, false); } while (0)
#define MDPP_FAST_L2(O64, P2) do { if (dl) MDPP_FAST_L3(O64, P2, true); else MDPP_FAST_L3(O64, P2, false); } while (0)
    if (a.obs_i32) { if (pow2) MDPP_FAST_L2(false, true); else MDPP_FAST_L2(false, false); }
    else { if (pow2) MDPP_FAST_L2(true, true); else MDPP_FAST_L2(true, false); }
#undef MDPP_FAST_L2
#undef MDPP_FAST_L3
#undef MDPP_FAST_L4
#undef MDPP_FAST_LAUNCH
    return true;
}

} // namespace mdpp

// Fused K-step rollout for the common discrete shape (BASELINE cfg 1/2): one shared MDP, unit
// rewards, no noise, sequence_length <= 3, S <= 16, numpy PCG64 streams, same-step autoreset.
// Same arithmetic as k_discrete_step (mdpp_discrete.hip; reference rl_toy_env.py:1992-2125 and
// reset :2250-2278) restructured for a chip that holds exactly ONE wavefront per SIMD at 65 536
// envs, i.e. no thread-level parallelism to hide anything behind:
//   * straight-line, branch-free step body so LDS latency of step k overlaps work of step k+1;
//   * no float64 and no integer divide in the loop: the four possible rewards are formed on the
//     host in the reference's float64 order; `steps % every_n` is carried incrementally; the
//     sequence key is carried incrementally; rho_0 sampling compares the raw 53-bit draw with
//     host-made integer thresholds ceil(cdf * 2^53)  (cdf[j] <= u  <=>  thr[j] <= r >> 11);
//   * terminal set and rho_0 thresholds live in SGPRs (kernel arguments), P and the reward
//     bitmask in LDS (64 B + 64 B for 8x8, L = 3);
//   * every global access is a buffer instruction: wave-uniform descriptor + per-step SGPR offset
//     + one per-lane VGPR offset that never changes, so no 64-bit address arithmetic per store;
//   * actions are fetched kAhead steps ahead of use (the only HBM read in the loop);
//   * rho_0 draws are made AHEAD of need into a 6-deep per-env queue (4 bits per start state,
//     kept in word 1 of the state record): with 2 of 8 states terminal some lane of a wave
//     resets on almost every step, and a PCG64 step is ~14 quarter-rate 32-bit multiplies, so
//     drawing inside the step would run that code every step at ~25 % lane utilisation.  The
//     queue is topped up once per kAhead steps in rounds with most lanes active.  Draws are consumed in
//     stream order and nothing else reads the env stream on this path, so every env still sees
//     exactly the variates the reference's reset() would draw; mdpp_get_streams rewinds the
//     stream by the number of queued draws so the reported PCG64 state equals the reference's.
// HBM traffic per env step: 4 B action in; 8 B obs + 4 B reward + 1 B + 1 B flags out.
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

template <bool OBS64, bool POW2, bool DELAY>
__global__ __launch_bounds__(kBlock) void k_discrete_rollout_fast(DiscreteArgs a, int K,
                                                                  const int32_t *__restrict__ actions,
                                                                  void *__restrict__ obs,
                                                                  float *__restrict__ reward,
                                                                  uint8_t *__restrict__ term,
                                                                  uint8_t *__restrict__ trunc,
                                                                  void *__restrict__ final_obs) {
    __shared__ __align__(16) uint8_t lds_P[256];
    __shared__ __align__(16) uint32_t lds_R[128]; // 4096 reward bits (16^3)
    __shared__ __align__(16) uint64_t lds_T[16];  // rho_0 thresholds (read only by refill rounds)
    const int tid = threadIdx.x;
    for (int k = tid; k < a.S * a.A; k += kBlock) lds_P[k] = a.P[k];
    for (uint32_t k = tid; k < 128; k += kBlock) {
        uint32_t w = 0;
        for (int b = 0; b < 4; b++) {
            uint32_t byte = 4 * k + b;
            if (byte < a.rbits_stride) w |= (uint32_t)a.rbits[byte] << (8 * b);
        }
        lds_R[k] = w;
    }
    if (tid < 16) lds_T[tid] = a.init_thr[tid];
    __syncthreads();
    const uint32_t i = blockIdx.x * kBlock + tid;
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

    auto step = [&](int action, uint32_t so) {
        // ---- action (numpy negative indexing; anything else out of range is flagged)
        uint32_t ua = (uint32_t)action;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(ua >= A) != 0, 0)) { // out of line
            ua = (uint32_t)(action + ((action >> 31) & (int)A));
            const bool bad = ua >= A;
            e.status |= bad ? (uint32_t)MDPP_STATUS_BAD_ACTION : 0u;
            ua = bad ? 0u : ua;
        }
        // ---- D1
#ifdef MDPP_ABL_NOLDSP
        const uint32_t nxt = (e.cur * A + ua) & 7u;
#else
        const uint32_t nxt = lds_P[e.cur * A + ua];
#endif
        // ---- D3 / D4 key: drop the oldest state, append the new one
        if (POW2) {
            e.key = ((e.key << a.s_shift) | nxt) & a.key_mask;
        } else {
            uint32_t old = (e.hist >> (8 * (L - 1))) & 0xFFu;
            old = (old == 0xFFu) ? 0u : old;
            e.key = (e.key - old * a.spow) * S + nxt;
        }
        e.hist = (e.hist << 8) | nxt;
        e.steps += 1;
        e.phase = (e.phase + 1 == every_n) ? 0u : e.phase + 1;
#ifdef MDPP_ABL_NOLDSR
        uint32_t bit = (e.key >> 1) & 1u;
#else
        uint32_t bit = (lds_R[e.key >> 5] >> (e.key & 31u)) & 1u;
#endif
        bit = (((e.hist >> (8 * L)) & 0xFFu) != 0xFFu) ? bit : 0u; // NaN slot: < L transitions since reset
        // ---- D5
        if (DELAY) {
            const uint32_t out = (e.ring >> dsh) & 1u;
            e.ring = (e.ring << 1) | bit;
            bit = out;
        }
        // ---- D6 / D7
        bit = (e.phase == 0) ? bit : 0u;
        const uint32_t done = (term32 >> nxt) & 1u;
        const uint32_t tr = (has_max && e.steps >= max_steps) ? 1u : 0u;
        const float r_nt = bit ? rs2 : rs0;
        const float r_t = bit ? rs3 : rs1;
        const float rout = done ? r_t : r_nt;
        // ---- same-step autoreset (reset(), :2250-2278)
#ifdef MDPP_ABL_NORESET
        const bool need = false;
#else
        const bool need = autoreset && ((done | tr) != 0);
#endif
        // With one wavefront per SIMD every TAKEN branch costs an instruction refetch that nothing
        // hides, so the common path below is branch-free: the reset is a set of selects, and the
        // two rare paths (queue ran dry; caller wants final_obs) are marked unlikely so that they
        // sit out of line and the hot path only falls through not-taken branches.
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(need && e.qc == 0) != 0, 0)) {
            Pcg64 n = e.g;
            const uint32_t sd = draw(n);
            if (need && e.qc == 0) { e.g = n; e.qv = sd; e.qc = 1; }
        }
        if (__builtin_expect(want_final, 0)) {
            if (need) {
                if (OBS64) __builtin_amdgcn_raw_buffer_store_b64(u32x2{nxt, 0u}, r_fin, v8, so * 8u, 0);
                else __builtin_amdgcn_raw_buffer_store_b32(nxt, r_fin, v4, so * 4u, 0);
            }
        }
        {
            const uint32_t s0 = e.qv & 0xFu;
            e.cur = need ? s0 : nxt;
            e.hist = need ? (0xFFFFFF00u | s0) : e.hist;
            e.key = need ? s0 : e.key;
            e.steps = need ? 0u : e.steps;
            e.phase = need ? 0u : e.phase;
            e.ring = need ? 0u : e.ring;
            e.qv = need ? (e.qv >> 4) : e.qv;
            e.qc = e.qc - (need ? 1u : 0u);
        }
        // ---- outputs
#ifdef MDPP_ABL_NOSTORE
        e.status ^= (e.cur + __float_as_uint(rout) + done + tr) & 0x100u;
        return;
#endif
        if (OBS64) __builtin_amdgcn_raw_buffer_store_b64(u32x2{e.cur, 0u}, r_obs, v8, so * 8u, 0);
        else __builtin_amdgcn_raw_buffer_store_b32(e.cur, r_obs, v4, so * 4u, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rout), r_rew, v4, so * 4u, 0);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)done, r_term, v1, so, 0);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)tr, r_trunc, v1, so, 0);
    };

    // Software pipeline on the action stream (the only HBM read of the loop): the loads of chunk
    // c+1 are issued before the kAhead steps of chunk c run.  Indices are clamped, never
    // predicated, so the chunk body is straight-line code with statically counted vmcnt waits.
    const int nfull = K / kAhead;
    int nextact[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; u++) {
        const uint32_t kk = (uint32_t)min(u, K - 1);
        nextact[u] = __builtin_amdgcn_raw_buffer_load_b32(r_act, v4, kk * N * 4u, 0);
    }
    for (int c = 0; c < nfull; c++) {
        int act[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; u++) act[u] = nextact[u];
        const int kbase = c * kAhead;
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
            const uint32_t kk = (uint32_t)min(kbase + kAhead + u, K - 1);
            nextact[u] = __builtin_amdgcn_raw_buffer_load_b32(r_act, v4, kk * N * 4u, 0);
        }
#ifndef MDPP_ABL_NOREFILL
        refill();
#endif
#pragma unroll
        for (int u = 0; u < kAhead; u++) step(act[u], (uint32_t)(kbase + u) * N);
    }
    for (int k = nfull * kAhead; k < K; k++) // tail: nextact[] holds exactly these steps
        step(nextact[k - nfull * kAhead], (uint32_t)k * N);

    a.state[i] = make_uint4(e.hist, e.qv | (e.qc << 24), e.steps, e.ring);
    e.g.store(a.env_s, i);
    if (e.status) atomicOr(&a.status[i], e.status);
}

// Returns false when the shape does not qualify (caller falls back to k_discrete_step).
bool launch_discrete_fast(const DiscreteArgs &a, int K, const int32_t *actions, void *obs,
                          float *reward, uint8_t *term, uint8_t *trunc, void *final_obs,
                          hipStream_t s) {
    if (!a.fast_ok) return false;
    const int grid = (a.N + kBlock - 1) / kBlock;
    const bool pow2 = a.s_shift != 0xFFFFFFFFu, dl = a.delay > 0;
#define MDPP_FAST_LAUNCH(O64, P2, DL)                                                              \
    hipLaunchKernelGGL((k_discrete_rollout_fast<O64, P2, DL>), dim3(grid), dim3(kBlock), 0, s, a, \
                       K, actions, obs, reward, term, trunc, final_obs)
#define MDPP_FAST_LAUNCH2(O64, P2) do { if (dl) MDPP_FAST_LAUNCH(O64, P2, true); else MDPP_FAST_LAUNCH(O64, P2, false); } while (0)
    if (a.obs_i32) { if (pow2) MDPP_FAST_LAUNCH2(false, true); else MDPP_FAST_LAUNCH2(false, false); }
    else { if (pow2) MDPP_FAST_LAUNCH2(true, true); else MDPP_FAST_LAUNCH2(true, false); }
#undef MDPP_FAST_LAUNCH2
#undef MDPP_FAST_LAUNCH
    return true;
}

} // namespace mdpp

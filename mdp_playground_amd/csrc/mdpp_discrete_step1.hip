// RLToyEnv.step() as ONE launch of ONE step (mdpp_step, K = 1) for the common discrete shape: one shared MDP, unit
// rewards, no noise, sequence_length <= 3, S <= 16 (DiscreteArgs::fast_ok, the shape of BASELINE cfg 1 / 2; Philox
// streams: shape_ok, the start state from the tick's word like everywhere else, no queue and no generator state).
// Same arithmetic as k_discrete_rollout_fast (mdpp_discrete_fast.hip; reference rl_toy_env.py:1992-2125, reset
// :2250-2278) -- what differs is what a launch of one step is made of.  At 65 536 envs a step moves 2.7 MB: 0.34 us
// of HBM time; the launch is nothing but latencies in series.  The rollout kernel with K = 1 paid five of them one after
// another (table BYTE loads -> barrier -> record + generator loads -> action load -> table lookups; 5.3 us per launch,
// profiles/r04_kernel_stats.csv).  Here:
//   * every global load of the launch is issued in the first instructions, side by side: the lane's 16-byte piece of a
//     1 KiB table blob packed at upload (mdpp_upload_discrete_tables), the action, the 16-byte record, the generator;
//   * 64-thread workgroups: the wave stages the blob into LDS for itself -- no barrier anywhere, and the chip's 1 024
//     SIMDs all start at once (256-thread workgroups with a barrier: 2.63 instead of 2.51 us per step in a replayed graph);
//   * the start-state queue (word 1 of the record, shared with the rollout kernels) is topped up AFTER the step's outputs
//     have been stored, one draw per lane and launch -- a lane pops at most one start state per step, so a queue that
//     holds one never runs dry and no draw sits between the loads and the observation store;
//   * `steps % every_n` without an integer division (float64 reciprocal, one conditional subtract).
// HBM traffic per env step: 4 B action + 16 B record + 32 B generator in; 16 B record + 14 B outputs (+ 16 B generator
// state for the lanes that drew) out.
#include "mdpp_internal.hpp"
#include "mdpp_rng.hpp"

namespace mdpp {

typedef unsigned int s1_u32x2 __attribute__((ext_vector_type(2)));

// The blob: 64 x uint4.  dwords 0-31: column a of P as 16 nibbles (nibble s = P[s][a]), a < 16; dwords 32-63: the
// rho_0 thresholds ceil(cdf[j] 2^53) (2^64 - 1 beyond S); dwords 64-191: the 4 096 reward bits; dwords 192-207: the
// thresholds of a Philox start state, ceil(cdf[j] 2^31) (2^32 - 1 beyond S); the rest zero.
constexpr int kS1Cols = 0, kS1Thr = 32, kS1Rew = 64, kS1Thr31 = 192;

template <bool OBS64, bool PHILOX>
__global__ __launch_bounds__(64) void k_discrete_step1(Step1Args a) {
    constexpr int WG = 64;
    __shared__ __align__(16) uint32_t lds[256];
    const int tid = threadIdx.x;
    const uint32_t i = blockIdx.x * WG + tid;
    const bool live = i < (uint32_t)a.N;
    const uint32_t ic = live ? i : (uint32_t)a.N - 1u;     // (spare lanes of the last block load lane N - 1's data and store nothing)
    // ---- every load of the launch, issued together ----
    const uint4 blob4 = a.blob[(blockIdx.x & (uint32_t)(kS1Replicas - 1)) * 64u + tid];
    const int action = a.actions[ic];
    const uint4 st = a.state[ic];
    Pcg64 g;
    if (!PHILOX) g.load(a.env_s, a.env_inc, ic);
    const uint64_t tick = PHILOX ? a.ptick + (a.dtick ? *a.dtick : 0ULL) : 0ULL;   // (a graph replay: through the device-side offset)
    ((uint4 *)lds)[tid] = blob4;
    __builtin_amdgcn_wave_barrier();                       // one wave: LDS accesses complete in order, nothing to wait for
    const uint32_t A = a.A, S = a.S, L = a.L;
    uint32_t status = 0;

    // ---- D1: the action's column of P (numpy negative indexing; anything else out of range is flagged, action 0) ----
    uint32_t ua = (uint32_t)action;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(ua >= A) != 0, 0)) {
        ua = (uint32_t)(action + ((action >> 31) & (int)A));
        const bool bad = ua >= A;
        status |= bad ? (uint32_t)MDPP_STATUS_BAD_ACTION : 0u;
        ua = bad ? 0u : ua;
    }
    const uint64_t col = ((const uint64_t *)(lds + kS1Cols))[ua];

    // word 1 of the record: numpy streams -- the queue of start states; Philox streams -- history bytes 4-7 (the general kernel's
    // 64-bit shift register: kept, never read at L <= 3)
    uint32_t hist = st.x, qv = st.y & 0x00FFFFFFu, qc = (st.y >> 24) & 7u, steps = st.z, ring = st.w;
    uint32_t hist_hi = (st.y << 8) | (st.x >> 24);
    const uint32_t cur = hist & 0xFFu;
    const uint32_t nxt = (uint32_t)(col >> (cur << 2)) & 0xFu;
    // ---- D3 / D4: sequence key over the last L states (NaN bytes count as 0 and are gated below) ----
    uint32_t key = 0;
#pragma unroll
    for (int j = 2; j >= 0; j--) {                         // bytes L-2 .. 0 of the OLD history, then the new state
        if (j <= (int)L - 2) {
            const uint32_t b = (hist >> (8 * j)) & 0xFFu;
            key = key * S + (b == 0xFFu ? 0u : b);
        }
    }
    key = key * S + nxt;
    hist = (hist << 8) | nxt;
    steps += 1;
    const bool full = (hist & a.nan_mask) != a.nan_mask;   // L transitions since reset (:1822)
    // steps % every_n: q = floor(steps * (1 / every_n)) in float64 is the quotient or one less (every_n < 2^20)
    uint32_t phase;
    {
        const uint32_t q = (uint32_t)((double)steps * a.inv_every_n);
        phase = steps - q * a.every_n;
        phase = phase >= a.every_n ? phase - a.every_n : phase;
    }
    const uint32_t word = lds[kS1Rew + (key >> 5)];
    const uint32_t done = (a.term32 >> nxt) & 1u;                                              // D7
    const uint32_t tr = (a.max_steps && steps >= a.max_steps) ? 1u : 0u;
    const bool need = a.autoreset && ((done | tr) != 0);
    // ---- D4-D7: reward bit -> delay line -> every-n mask -> one of four host-made float32 rewards ----
    uint32_t bit = (word >> (key & 31u)) & (full ? 1u : 0u);
    if (a.delay) {
        const uint32_t out = (ring >> (a.delay - 1u)) & 1u;
        ring = (ring << 1) | bit;
        bit = out;
        ring = need ? 0u : ring;                           // reset() clears the FIFO (:2250)
    }
    bit = phase == 0 ? bit : 0u;                           // steps % every_n == 0 (:1975)
    const float rout = done ? (bit ? a.rsel[3] : a.rsel[1]) : (bit ? a.rsel[2] : a.rsel[0]);

    // ---- same-step autoreset: pop a start state; an empty queue (the first step after a reset() / a new seed) draws in place ----
    auto draw = [&](Pcg64 &gen) -> uint32_t {
        const uint64_t m = gen.next64() >> 11;
        uint32_t s0 = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) s0 += (((const uint64_t *)(lds + kS1Thr))[j] <= m) ? 1u : 0u;
        if (S > 8) {
#pragma unroll
            for (int j = 8; j < 16; j++) s0 += (((const uint64_t *)(lds + kS1Thr))[j] <= m) ? 1u : 0u;
        }
        return s0;
    };
    bool drew = false;
    uint32_t ocur = nxt;
    if constexpr (PHILOX) {
        if (__builtin_amdgcn_ballot_w64(need) != 0) {      // the tick's word of the start-state stream (mdpp_rng.hpp)
            const uint32_t m31 = philox_start_m31(a.philox_seed, (uint64_t)(a.env_id_offset + (int64_t)ic), tick, kPhiloxStartStream);
            uint32_t s0 = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) s0 += (lds[kS1Thr31 + j] <= m31) ? 1u : 0u;
            if (need) { ocur = s0; hist = 0xFFFFFF00u | s0; hist_hi = 0xFFFFFFFFu; steps = 0; }
        }
    } else {
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(need && qc == 0) != 0, 0)) {
            Pcg64 n = g;
            const uint32_t s0 = draw(n);
            if (need && qc == 0) { g = n; qv = s0; qc = 1; drew = true; }
        }
        if (need) {
            ocur = qv & 0xFu;
            hist = 0xFFFFFF00u | ocur;
            steps = 0;
            qv >>= 4; qc -= 1;
        }
    }
    // ---- outputs ----
    if (live) {
        if (__builtin_expect(a.final_obs != nullptr, 0)) {
            if (need) {
                if (OBS64) ((s1_u32x2 *)a.final_obs)[i] = s1_u32x2{nxt, 0u};
                else ((uint32_t *)a.final_obs)[i] = nxt;
            }
        }
        if (OBS64) ((s1_u32x2 *)a.obs)[i] = s1_u32x2{ocur, 0u};
        else ((uint32_t *)a.obs)[i] = ocur;
        a.reward[i] = rout;
        a.term[i] = (uint8_t)done;
        a.trunc[i] = (uint8_t)tr;
    }
    // ---- top the queue up, off the path to the outputs: rounds of one draw per lane that has room ----
    if (!PHILOX && a.autoreset) {
        for (uint32_t r = 0; r < a.topup_rounds; r++) {
            const bool want = qc < a.topup_fill;
            if (__builtin_amdgcn_ballot_w64(want) == 0) break;
            Pcg64 n = g;
            const uint32_t s0 = draw(n);
            if (want) { g = n; qv |= s0 << (4u * qc); qc += 1; drew = true; }
        }
    }
    if (live) {
        a.state[i] = make_uint4(hist, PHILOX ? hist_hi : (qv | (qc << 24)), steps, ring);
        if (drew) g.store(a.env_s, i);
        if (status) atomicOr(&a.status[i], status);
    }
}

// k_discrete_step1w: the same launch for state spaces beyond 16 states -- any S <= 255 whose tables (P as bytes, terminal flags,
// rho_0 thresholds, reward bits) fit 8 KiB, L <= 3, unit rewards, no noise, no irrelevant sub-space: what the reference's own
// sweeps use (S = A = 24 and 50, /root/reference/experiments/).  These handles keep no queue of start states (word 1 of their
// record is the general kernel's history bytes 4-7), so an episode's end draws in place: numpy streams -- one PCG64 step and
// the search of the S thresholds under ballot(need) --, Philox streams -- the tick's word.  The general kernel served these
// single steps with byte loads -> LDS -> barrier -> record load in series: 6.4 us per step at S = 50.
// UR = false: rewards that are not all 1.0 (reward_dist; the reference's rainbow_reward_dist sweep): the reward is read from a
// float64 table by the sequence key and formed in float64 in the reference's order, the delay line holds KEYS in HBM
// (ring_keys[delay][N], shared with the general and the quiet kernel) -- like k_discrete_step<UNIT = false>.
// NZ = true: transition and / or reward noise (the reference's p_noise / r_noise sweeps on 8 x 8 envs; unit rewards).  numpy
// streams: the step's uniform of the state space's stream re-draws the next state from the categorical around the table's entry
// (:1604-1622; integer thresholds like rho_0), the reward noise is one ziggurat normal of the ENV stream (tables in the blob),
// drawn before a reset()'s start state as numpy orders them; Philox streams: the tick's words (mdpp_rng.hpp).
template <bool OBS64, bool PHILOX, bool UR, bool NZ = false>
__global__ __launch_bounds__(64) void k_discrete_step1w(Step1Args a) {
    static_assert(!NZ || UR, "noise: unit rewards");
    constexpr int kRounds = NZ ? (int)kS1wRoundsNoise : (int)kS1wRounds;
    extern __shared__ __align__(16) uint8_t ldsw[];
    const int tid = threadIdx.x;
    const uint32_t i = blockIdx.x * 64u + tid;
    const bool live = i < (uint32_t)a.N;
    const uint32_t ic = live ? i : (uint32_t)a.N - 1u;
    // ---- every load of the launch, issued together: up to eight 1 KiB rounds of the blob, the action, the record, the generator ----
    // (eight loads issued back to back, UNCONDITIONALLY -- rounds beyond the blob re-read its last one: under `if (r < rounds)`
    //  the compiler waited for each round before it issued the next, one L2 round trip per KiB: 2.7 us per step at 1 KiB,
    //  4.0 at 4 KiB, 6.0 at 8 KiB)
    uint4 bl[kRounds];
    const uint32_t rbase = (blockIdx.x & (uint32_t)(kS1Replicas - 1)) * a.blob_rounds, rlast = a.blob_rounds - 1u;
#pragma unroll
    for (int r = 0; r < kRounds; r++) bl[r] = a.blob[(rbase + min((uint32_t)r, rlast)) * 64u + tid];
    const int action = a.actions[ic];
    const uint4 st = a.state[ic];
    Pcg64 g, sp;
    if (!PHILOX) g.load(a.env_s, a.env_inc, ic);
    if (NZ && !PHILOX && a.has_p_noise) sp.load(a.sp_s, a.sp_inc, ic);
    const uint64_t tick = (PHILOX || !UR) ? a.ptick + (a.dtick ? *a.dtick : 0ULL) : 0ULL;
    const uint64_t genv = (uint64_t)(a.env_id_offset + (int64_t)ic);
    // (the key that leaves the delay line this step: loaded now, beside everything else)
    uint32_t *kslot = nullptr;
    uint32_t kout = kNoKey;
    if (!UR && a.delay) {
        const uint32_t head = a.dtick ? (uint32_t)(tick % (uint64_t)a.delay) : a.ring_head;
        kslot = a.ring_keys + (size_t)head * (size_t)a.N + ic;
        kout = *kslot;
    }
#pragma unroll
    for (int r = 0; r < kRounds; r++)
        if ((uint32_t)r < a.blob_rounds) ((uint4 *)ldsw)[r * 64 + tid] = bl[r];
    __builtin_amdgcn_wave_barrier();
    const uint32_t A = a.A, S = a.S, L = a.L;
    uint32_t status = 0;
    uint32_t ua = (uint32_t)action;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(ua >= A) != 0, 0)) {
        ua = (uint32_t)(action + ((action >> 31) & (int)A));
        const bool bad = ua >= A;
        status |= bad ? (uint32_t)MDPP_STATUS_BAD_ACTION : 0u;
        ua = bad ? 0u : ua;
    }
    uint32_t hist = st.x, hist_hi = (st.y << 8) | (st.x >> 24), steps = st.z, ring = st.w;
    const uint32_t cur = hist & 0xFFu;
    uint32_t nxt = ldsw[cur * A + ua];                                                         // D1
    if constexpr (NZ) {
        if (a.has_p_noise) {                                                                   // D2 (:1604-1622)
            if constexpr (PHILOX) {
                uint32_t o[4];
                philox_start_block(a.philox_seed, genv, tick >> 2, kPhiloxPNoiseStream, o);
                nxt = philox_pnoise_state(philox_word_of(o, tick), a.pn_T, a.pn_M, nxt);
            } else {
                const uint64_t m = sp.next64() >> 11;
                const uint64_t *row = (const uint64_t *)(ldsw + a.off_tn) + nxt * a.S8;
                uint32_t c = 0;
                for (uint32_t b = 0; b < a.S8; b += 8) {
#pragma unroll
                    for (uint32_t j = 0; j < 8; j++) c += (row[b + j] <= m) ? 1u : 0u;
                }
                nxt = c;
            }
        }
    }
    uint32_t key = 0;
#pragma unroll
    for (int j = 2; j >= 0; j--) {
        if (j <= (int)L - 2) {
            const uint32_t b = (hist >> (8 * j)) & 0xFFu;
            key = key * S + (b == 0xFFu ? 0u : b);
        }
    }
    key = key * S + nxt;
    hist = (hist << 8) | nxt;
    steps += 1;
    const bool full = (hist & a.nan_mask) != a.nan_mask;
    uint32_t phase;
    {
        const uint32_t q = (uint32_t)((double)steps * a.inv_every_n);
        phase = steps - q * a.every_n;
        phase = phase >= a.every_n ? phase - a.every_n : phase;
    }
    const uint32_t done = ldsw[a.off_term + nxt] != 0 ? 1u : 0u;                               // D7
    const uint32_t tr = (a.max_steps && steps >= a.max_steps) ? 1u : 0u;
    const bool need = a.autoreset && ((done | tr) != 0);
    float rout;
    if constexpr (UR) {
        const uint32_t word = ((const uint32_t *)(ldsw + a.off_rew))[key >> 5];
        uint32_t bit = (word >> (key & 31u)) & (full ? 1u : 0u);
        if (a.delay) {
            const uint32_t out = (ring >> (a.delay - 1u)) & 1u;
            ring = (ring << 1) | bit;
            bit = out;
            ring = need ? 0u : ring;
        }
        bit = phase == 0 ? bit : 0u;
        rout = done ? (bit ? a.rsel[3] : a.rsel[1]) : (bit ? a.rsel[2] : a.rsel[0]);
        if constexpr (NZ) {
            if (a.has_r_noise) {                                                                // :1980-1990, :2107 in float64
                double z;
                if constexpr (PHILOX) {
                    uint32_t o[4];
                    philox_start_block(a.philox_seed, genv, tick >> 2, kPhiloxRNoiseStream, o);
                    float z4[4];
                    philox_box_muller2(o, z4[0], z4[1], z4[2], z4[3]);
                    const uint32_t q = (uint32_t)tick & 3u;
                    z = (double)(q == 0u ? z4[0] : q == 1u ? z4[1] : q == 2u ? z4[2] : z4[3]);
                } else {
                    const ZigLds zig{(const uint64_t *)(ldsw + a.off_zig), (const double *)(ldsw + a.off_zig + 2048),
                                     (const double *)(ldsw + a.off_zig + 4096)};
                    z = np_standard_normal_lds(g, zig);
                }
                double r = bit ? 1.0 : 0.0;
                r += 0.0 + a.r_noise * z;
                r *= a.scale;
                r += a.shift;
                if (done) r += a.term_add;
                rout = (float)r;
            }
        }
    } else {                                                                                    // :1821-1845, :1968-1990, :2107
        uint32_t k = full ? key : kNoKey;
        if (a.delay) {
            if (live) *kslot = k;
            k = kout;
        }
        double r = (k != kNoKey) ? ((const double *)(ldsw + a.off_rew))[k] : 0.0;
        r = phase == 0 ? r : 0.0;
        r *= a.scale;
        r += a.shift;
        if (done) r += a.term_add;
        rout = (float)r;
        if (a.delay && __builtin_amdgcn_ballot_w64(need) != 0) {                                // reset() empties the delay line (:2250)
            if (need && live) for (uint32_t dd = 0; dd < a.delay; dd++) a.ring_keys[(size_t)dd * (size_t)a.N + i] = kNoKey;
        }
    }
    // ---- same-step autoreset: the start state is drawn now (reset(), :2250-2278) ----
    bool drew = false;
    uint32_t ocur = nxt;
    if (__builtin_amdgcn_ballot_w64(need) != 0) {
        // searchsorted(cdf, u, 'right') = #{j : T[j] <= m}: the bucket of the uniform's top 8 bits knows how many thresholds lie
        // at or below its first value and how many strictly inside it -- only those are compared (none in most buckets; a
        // linear search of the S8 thresholds took 1.2 of the launch's 3.8 us at S = 50)
        uint32_t s0 = 0;
        const uint16_t *BK = (const uint16_t *)(ldsw + a.off_bk);
        if constexpr (PHILOX) {
            const uint32_t m31 = philox_start_m31(a.philox_seed, (uint64_t)(a.env_id_offset + (int64_t)ic), tick, kPhiloxStartStream);
            const uint32_t *T31 = (const uint32_t *)(ldsw + a.off_thr31);
            const uint32_t e = BK[m31 >> 23], c0 = e & 0xFFu, nin = e >> 8;
            s0 = c0;
            for (uint32_t j = 0; __builtin_amdgcn_ballot_w64(j < nin) != 0; j++) s0 += (j < nin && T31[j < nin ? c0 + j : 0u] <= m31) ? 1u : 0u;
        } else {
            Pcg64 n = g;
            const uint64_t m = n.next64() >> 11;
            const uint64_t *T = (const uint64_t *)(ldsw + a.off_thr);
            const uint32_t e = BK[(uint32_t)(m >> 45)], c0 = e & 0xFFu, nin = e >> 8;
            s0 = c0;
            for (uint32_t j = 0; __builtin_amdgcn_ballot_w64(j < nin) != 0; j++) s0 += (j < nin && T[j < nin ? c0 + j : 0u] <= m) ? 1u : 0u;
            if (need) { g = n; drew = true; }
        }
        if (need) { ocur = s0; hist = 0xFFFFFF00u | s0; hist_hi = 0xFFFFFFFFu; steps = 0; }
    }
    if (live) {
        if (__builtin_expect(a.final_obs != nullptr, 0)) {
            if (need) {
                if (OBS64) ((s1_u32x2 *)a.final_obs)[i] = s1_u32x2{nxt, 0u};
                else ((uint32_t *)a.final_obs)[i] = nxt;
            }
        }
        if (OBS64) ((s1_u32x2 *)a.obs)[i] = s1_u32x2{ocur, 0u};
        else ((uint32_t *)a.obs)[i] = ocur;
        a.reward[i] = rout;
        a.term[i] = (uint8_t)done;
        a.trunc[i] = (uint8_t)tr;
        a.state[i] = make_uint4(hist, hist_hi, steps, ring);
        if (drew || (NZ && !PHILOX && a.has_r_noise)) g.store(a.env_s, i);
        if (NZ && !PHILOX && a.has_p_noise) sp.store(a.sp_s, i);
        if (status) atomicOr(&a.status[i], status);
    }
}

bool launch_discrete_step1(const DiscreteArgs &d, const Step1Args &proto, const int32_t *actions, void *obs, float *reward,
                           uint8_t *term, uint8_t *trunc, void *final_obs, hipStream_t s, char *name_out) {
    if (!proto.blob || (d.opts & MDPP_OPT_NO_STEP1)) return false;
    if (proto.wide) {                               // state spaces beyond 16 states: k_discrete_step1w
        if (d.philox && (d.opts & MDPP_OPT_NO_PHILOX_FAST)) return false;
        if (proto.blob_rounds > ((proto.has_p_noise || proto.has_r_noise) ? kS1wRoundsNoise : kS1wRounds)) return false;
        if (name_out) {
            if (proto.has_p_noise || proto.has_r_noise)
                snprintf(name_out, kNameLen, "k_discrete_step1w<OBS64=%d,PHILOX=%d,UNIT=1,PN=%d,RN=%d>", !d.obs_i32, d.philox, proto.has_p_noise, proto.has_r_noise);
            else
            snprintf(name_out, kNameLen, "k_discrete_step1w<OBS64=%d,PHILOX=%d,UNIT=%d>", !d.obs_i32, d.philox, proto.unit);
            return true;
        }
        Step1Args a = proto;
        a.actions = actions; a.obs = obs; a.reward = reward; a.term = term; a.trunc = trunc; a.final_obs = final_obs;
        a.ptick = d.ptick; a.dtick = d.dtick; a.ring_head = d.tick;
        const int grid = (a.N + 63) / 64;
        const size_t lds = (size_t)a.blob_rounds * 1024;
#define MDPP_S1W(O64, PH)                                                                                           \
    do {                                                                                                            \
        if (a.has_p_noise || a.has_r_noise) hipLaunchKernelGGL((k_discrete_step1w<O64, PH, true, true>), dim3(grid), dim3(64), lds, s, a); \
        else if (a.unit) hipLaunchKernelGGL((k_discrete_step1w<O64, PH, true>), dim3(grid), dim3(64), lds, s, a);   \
        else hipLaunchKernelGGL((k_discrete_step1w<O64, PH, false>), dim3(grid), dim3(64), lds, s, a);             \
    } while (0)
        if (d.philox) { if (d.obs_i32) MDPP_S1W(false, true); else MDPP_S1W(true, true); }
        else { if (d.obs_i32) MDPP_S1W(false, false); else MDPP_S1W(true, false); }
#undef MDPP_S1W
        return true;
    }
    if (d.philox ? (!d.shape_ok || (d.opts & MDPP_OPT_NO_PHILOX_FAST)) : !d.fast_ok) return false;
    if (name_out) {
        snprintf(name_out, kNameLen, "k_discrete_step1<OBS64=%d,PHILOX=%d>", !d.obs_i32, d.philox);
        return true;
    }
    Step1Args a = proto;
    a.actions = actions; a.obs = obs; a.reward = reward; a.term = term; a.trunc = trunc; a.final_obs = final_obs;
    a.ptick = d.ptick; a.dtick = d.dtick;
    const int grid = (a.N + 63) / 64;
    if (d.philox) {
        if (d.obs_i32) hipLaunchKernelGGL((k_discrete_step1<false, true>), dim3(grid), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((k_discrete_step1<true, true>), dim3(grid), dim3(64), 0, s, a);
    } else {
        if (d.obs_i32) hipLaunchKernelGGL((k_discrete_step1<false, false>), dim3(grid), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((k_discrete_step1<true, false>), dim3(grid), dim3(64), 0, s, a);
    }
    return true;
}

} // namespace mdpp

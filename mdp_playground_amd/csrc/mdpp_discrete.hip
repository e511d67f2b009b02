// Discrete RLToyEnv.step()/reset() for gfx950: one lane per env instance.
//
// Restates /root/reference/mdp_playground/envs/rl_toy_env.py
//   D1 P lookup            :1602-1603      D5 delay FIFO        :1968-1973
//   D2 categorical P-noise :1604-1622      D6 every-n/noise/affine :1975-1990
//   D3 history shift       :2050-2058      D7 terminal + reward :2102-2109
//   D4 sequence reward     :1821-1845      R1 reset             :2250-2278, :2354-2369
//
// Data layout (HBM):
//   state[N]   one 16-byte record per env {hist[0..3], hist[4..7], steps, ring bits}: a single
//              dwordx4 load + store per lane, consecutive lanes -> consecutive 16 B (1 KiB/wave).
//              hist holds the last L+1 states of augmented_state, newest in byte 0, 0xFF = NaN.
//   ring bits  the reward_buffer as a shift register when every reward is 1.0 (delay <= 32);
//              otherwise ring_keys[delay][N] holds the sequence keys awaiting payout.
//   PCG64      state[N], inc[N] as ulonglong2 per stream; only touched when a draw happens.
// Shared tables (P, terminal flags, reward bitmask/table, rho_0 cdf, P-noise cdfs) are staged
// into LDS once per block; with one MDP per env they are gathered from HBM/L2 instead.
#include "mdpp_internal.hpp"
#include "mdpp_rng.hpp"

#ifdef MDPP_D_WIDE        /* (second compilation of this file, mdpp_discrete_wide.hip: its own symbol names) */
#define k_discrete_step k_discrete_step_wide
#define k_discrete_reset k_discrete_reset_wide
#define d_reset_draw d_reset_draw_wide
#define DTables DTablesWide
#define DHist DHistWide
#define launch_step_t launch_step_wide_t
#define launch_discrete_step_var launch_discrete_step_wide
#define launch_discrete_reset_var launch_discrete_reset_wide
#define MDPP_D_VARIANT "wide"
#endif
#ifdef MDPP_D_LONG        /* (third compilation, mdpp_discrete_long.hip: sequence_length 8 ... 15) */
#define k_discrete_step k_discrete_step_long
#define k_discrete_reset k_discrete_reset_long
#define d_reset_draw d_reset_draw_long
#define DTables DTablesLong
#define DHist DHistLong
#define launch_step_t launch_step_long_t
#define launch_discrete_step_var launch_discrete_step_long
#define launch_discrete_reset_var launch_discrete_reset_long
#define MDPP_D_VARIANT "long"
#endif

namespace mdpp {

struct DTables {
    const uint8_t *P, *is_term, *rbits;
    const double *rtable, *init_cdf, *noise_cdf;
};

template <class G>
__device__ __forceinline__ uint32_t d_reset_draw(const DiscreteArgs &a, const DTables &t, G &g) {
    // self._np_random.choice(S, p=rho_0): one uniform, searchsorted(cdf, u, 'right')  (:2255)
    double u = np_random(g);
    return (uint32_t)searchsorted_right(t.init_cdf, a.S, u);
}

// The last L + 1 states, newest first.  S <= 255: eight byte fields of one 64-bit word, 0xFF = NaN (this file as it is compiled);
// S up to 65 535 (MDPP_D_WIDE, mdpp_discrete_wide.hip compiles this file a second time under other names): eight 16-bit fields,
// 0xFFFF = NaN, the older four in DiscreteArgs::hist_hi; P entries are 16-bit there.
#ifdef MDPP_D_WIDE
typedef uint16_t DPEntry;
constexpr uint32_t kDNaN = 0xFFFFu;
struct DHist {
    uint64_t lo, hi;
    __device__ __forceinline__ static DHist fresh(uint32_t s0) { return DHist{0xFFFFFFFFFFFF0000ULL | (uint64_t)s0, ~0ULL}; }
    __device__ __forceinline__ static DHist load(const DiscreteArgs &a, long i, const uint4 &st) { return DHist{((uint64_t)st.y << 32) | st.x, a.hist_hi[i]}; }
    __device__ __forceinline__ void store_hi(const DiscreteArgs &a, long i) const { a.hist_hi[i] = hi; }
    __device__ __forceinline__ uint32_t cur() const { return (uint32_t)lo & 0xFFFFu; }
    __device__ __forceinline__ void push(uint32_t n) { hi = (hi << 16) | (lo >> 48); lo = (lo << 16) | n; }
    __device__ __forceinline__ uint32_t at(int j) const { return (uint32_t)((j < 4 ? lo >> (16 * j) : hi >> (16 * (j - 4))) & 0xFFFFu); }
};
#elif defined(MDPP_D_LONG)
// sequence_length 8 ... 15 (S <= 255): sixteen byte fields, the older eight in DiscreteArgs::hist_hi
typedef uint8_t DPEntry;
constexpr uint32_t kDNaN = 0xFFu;
struct DHist {
    uint64_t lo, hi;
    __device__ __forceinline__ static DHist fresh(uint32_t s0) { return DHist{0xFFFFFFFFFFFFFF00ULL | (uint64_t)s0, ~0ULL}; }
    __device__ __forceinline__ static DHist load(const DiscreteArgs &a, long i, const uint4 &st) { return DHist{((uint64_t)st.y << 32) | st.x, a.hist_hi[i]}; }
    __device__ __forceinline__ void store_hi(const DiscreteArgs &a, long i) const { a.hist_hi[i] = hi; }
    __device__ __forceinline__ uint32_t cur() const { return (uint32_t)lo & 0xFFu; }
    __device__ __forceinline__ void push(uint32_t n) { hi = (hi << 8) | (lo >> 56); lo = (lo << 8) | n; }
    __device__ __forceinline__ uint32_t at(int j) const { return (uint32_t)((j < 8 ? lo >> (8 * j) : hi >> (8 * (j - 8))) & 0xFFu); }
};
#else
typedef uint8_t DPEntry;
constexpr uint32_t kDNaN = 0xFFu;
struct DHist {
    uint64_t lo;
    __device__ __forceinline__ static DHist fresh(uint32_t s0) { return DHist{0xFFFFFFFFFFFFFF00ULL | (uint64_t)s0}; }
    __device__ __forceinline__ static DHist load(const DiscreteArgs &, long, const uint4 &st) { return DHist{((uint64_t)st.y << 32) | st.x}; }
    __device__ __forceinline__ void store_hi(const DiscreteArgs &, long) const {}
    __device__ __forceinline__ uint32_t cur() const { return (uint32_t)lo & 0xFFu; }
    __device__ __forceinline__ void push(uint32_t n) { lo = (lo << 8) | n; }
    __device__ __forceinline__ uint32_t at(int j) const { return (uint32_t)((lo >> (8 * j)) & 0xFFu); }
};
#endif

constexpr int kPrefetch = 8; // actions fetched this many steps ahead of their use

// NOISE: any per-step random draw (P-noise and/or reward noise).  UNIT: every rewardable sequence
// pays exactly 1.0 (bitmask table, shift-register delay line).  The common benchmark shape
// (no noise, unit rewards) compiles to a loop with no float64 arithmetic at all: the four
// possible rewards {paid, not paid} x {terminal, not} are formed once per launch with the
// reference's own float64 operation order (:1987-1990, :2107) and selected per step.
// LDSTAB: the (single, shared) MDP's tables are read from LDS; otherwise from HBM/L2 with one
// table set per env (or table 0 for a shared MDP too large for LDS).  Kept a template parameter
// so that table pointers have one provenance and lower to ds_read / global_load, not flat_load.
// IRR: a second, reward-irrelevant sub-space (irrelevant_features=True, :2028-2035, :2063-2092):
// actions and observations are pairs, the irrelevant part has its own table (read from HBM/L2) and
// its own P-noise generator, and reset() draws its start state after the relevant one (:2259-2264).
template <bool PHILOX, bool NOISE, bool UNIT, bool LDSTAB, bool IRR>
__global__ __launch_bounds__(kBlock) void k_discrete_step(DiscreteArgs a, int K,
                                                          const int32_t *__restrict__ actions,
                                                          void *__restrict__ obs,
                                                          float *__restrict__ reward,
                                                          uint8_t *__restrict__ term,
                                                          uint8_t *__restrict__ trunc,
                                                          void *__restrict__ final_obs) {
    const uint64_t ptick0 = tick_now(a);               // the step counter at this launch (through the device-side offset of a graph replay)
    const uint32_t rhead0 = ring_head_now(a, ptick0);    // ... and the head of a delay line kept in memory
    extern __shared__ __align__(16) unsigned char lds[];
    __shared__ uint64_t s_ki[NOISE ? 256 : 1];
    __shared__ double s_wi[NOISE ? 256 : 1], s_fi[NOISE ? 256 : 1];
    DTables t;
    const int tid = threadIdx.x;
    const long i = (long)blockIdx.x * kBlock + tid;
    if (NOISE) zig_stage(s_ki, s_wi, s_fi, tid, kBlock);
    if (NOISE && !LDSTAB) __syncthreads();
    const ZigLds zig{s_ki, s_wi, s_fi};
    // One MDP PER ENV (seeds=[...], what every golden uses) with small tables (round 5): each lane copies ITS env's tables
    // into its own LDS slot once per launch -- 200 B at 8 x 8 -- so that the step's dependent lookups (P[s][a], terminal flag,
    // reward bit, rho_0) are LDS reads instead of one L2 round trip each: 128 waves of an 8 192-env job ran 1.66 us per step
    // on gathers from L2 (bench leg cfg2_per_env).  Slot stride = the shared carve + 8 bytes (bank spread, 8-byte alignment kept).
    const bool per_env_lds = LDSTAB && !a.shared_tables;
    if (per_env_lds) {
        unsigned char *slot = lds + (size_t)tid * (a.lds_bytes + 8u);
        const size_t ti = (size_t)(i < a.N ? i : a.N - 1);
        const int SA = a.S * a.A;
        for (int k = 0; k < SA; k++) slot[a.lds_P + k] = a.P[ti * SA + k];
        for (int k = 0; k < a.S; k++) {
            slot[a.lds_term + k] = a.is_term[ti * a.S + k];
            ((double *)(slot + a.lds_init))[k] = a.init_cdf[ti * a.S + k];
        }
        if (UNIT)
            for (uint32_t k = 0; k < a.rbits_stride; k++) slot[a.lds_rew + k] = a.rbits[ti * a.rbits_stride + k];
        else
            for (uint32_t k = 0; k < a.nkeys; k++) ((double *)(slot + a.lds_rew))[k] = a.rtable[ti * a.nkeys + k];
        t.P = slot + a.lds_P;
        t.is_term = slot + a.lds_term;
        t.init_cdf = (const double *)(slot + a.lds_init);
        t.rbits = slot + a.lds_rew;
        t.rtable = (const double *)(slot + a.lds_rew);
        t.noise_cdf = a.noise_cdf;                  // (not used: this mode is for handles without transition noise)
        if (NOISE) __syncthreads();                 // (the ziggurat tables staged above)
    } else if (LDSTAB) {
        // Stage the shared MDP into LDS: a few hundred bytes for 8x8 (P 64 B + flags 8 B +
        // reward bitmask 64 B + cdf 64 B).
        for (int k = tid; k < a.S * a.A; k += kBlock) lds[a.lds_P + k] = a.P[k];
        for (int k = tid; k < a.S; k += kBlock) {
            lds[a.lds_term + k] = a.is_term[k];
            ((double *)(lds + a.lds_init))[k] = a.init_cdf[k];
        }
        if (UNIT)
            for (uint32_t k = tid; k < a.rbits_stride; k += kBlock) lds[a.lds_rew + k] = a.rbits[k];
        else
            for (uint32_t k = tid; k < a.nkeys; k += kBlock)
                ((double *)(lds + a.lds_rew))[k] = a.rtable[k];
        if (NOISE && a.has_p_noise)
            for (int k = tid; k < a.S * a.S; k += kBlock)
                ((double *)(lds + a.lds_noise))[k] = a.noise_cdf[k];
        __syncthreads();
        t.P = lds + a.lds_P;
        t.is_term = lds + a.lds_term;
        t.init_cdf = (const double *)(lds + a.lds_init);
        t.rbits = lds + a.lds_rew;
        t.rtable = (const double *)(lds + a.lds_rew);
        t.noise_cdf = (const double *)(lds + a.lds_noise);
    }
    if (i >= a.N) return;
    if (!LDSTAB) {
        const size_t ti = a.shared_tables ? 0 : (size_t)i;
        t.P = a.P + ti * a.S * a.A * sizeof(DPEntry);
        t.is_term = a.is_term + ti * a.S;
        t.init_cdf = a.init_cdf + ti * a.S;
        t.rbits = a.rbits + ti * a.rbits_stride;
        t.rtable = a.rtable + ti * a.nkeys;
        t.noise_cdf = a.noise_cdf;
    }
    const int S = a.S, A = a.A, L = a.L;
    const long N = a.N;

    uint4 st = a.state[i];
    DHist hist = DHist::load(a, i, st);
    uint32_t steps = st.z, ringbits = st.w, status = 0;
    // next-step autoreset (gymnasium >= 1.0 vector envs): an env whose episode ended is reset by the NEXT
    // step() call, which ignores its action and returns the first observation with reward 0 and no flags.
    // The "episode ended" flag travels in bit 31 of the step counter.
    const bool next_step = a.autoreset == MDPP_AUTORESET_NEXT_STEP;
    bool pending = next_step && (steps >> 31) != 0;
    steps &= 0x7FFFFFFFu;
    uint32_t phase = steps % (uint32_t)a.every_n; // steps % every_n, kept incrementally below

    // Streams live in registers for the whole launch.  The env stream is needed by reward noise
    // and by every in-kernel reset(); with 2 of 8 states terminal some lane of a wave resets on
    // almost every step, so it is loaded up front rather than inside the divergent branch.
    Pcg64 env_pcg, sp_pcg, sp1_pcg;
    PhiloxTickWords pn_w, pn_w1;           // Philox streams: the current four ticks' noise words / normals (mdpp_rng.hpp)
    PhiloxTickNormals rn_z;
    const uint64_t genv = (uint64_t)(a.env_id_offset + i);
    const bool use_env = (NOISE && a.has_r_noise) || a.autoreset;
    const bool use_sp = NOISE && a.has_p_noise;
    if (!PHILOX) {
        if (use_env) env_pcg.load(a.env_s, a.env_inc, i);
        if (use_sp) sp_pcg.load(a.sp_s, a.sp_inc, i);
        if (IRR && use_sp) sp1_pcg.load(a.sp1_s, a.sp1_inc, i);
    }
    // irrelevant sub-space: state + tables (one set per env or shared, like the relevant ones)
    uint32_t cur1 = 0;
    const uint8_t *P1 = nullptr;
    const double *init_cdf1 = nullptr;
    if (IRR) {
        const size_t ti = a.shared_tables ? 0 : (size_t)i;
        cur1 = a.irr_state[i];
        P1 = a.P1 + ti * a.S1 * a.A1;
        init_cdf1 = a.init_cdf1 + ti * a.S1;
    }
    constexpr int AW = IRR ? 2 : 1;              // ints per action / observation

    // rewards of the noise-free unit path: index = paid*2 + terminal
    float rsel[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        double r = (q & 2) ? 1.0 : 0.0;
        r *= a.scale;
        r += a.shift;
        if (q & 1) r += a.term_add;
        rsel[q] = (float)r;
    }

    // software pipeline on the action stream: kPrefetch loads in flight per lane
    int nextact[kPrefetch], nextact1[kPrefetch];
#pragma unroll
    for (int u = 0; u < kPrefetch; u++) {
        nextact[u] = (u < K) ? actions[((long)u * N + i) * AW] : 0;
        nextact1[u] = (IRR && u < K) ? actions[((long)u * N + i) * AW + 1] : 0;
    }

    for (int k0 = 0; k0 < K; k0 += kPrefetch) {
        int act[kPrefetch], act1[kPrefetch];
#pragma unroll
        for (int u = 0; u < kPrefetch; u++) { act[u] = nextact[u]; act1[u] = nextact1[u]; }
#pragma unroll
        for (int u = 0; u < kPrefetch; u++) {
            const int kn = k0 + kPrefetch + u;
            nextact[u] = (kn < K) ? actions[((long)kn * N + i) * AW] : 0;
            nextact1[u] = (IRR && kn < K) ? actions[((long)kn * N + i) * AW + 1] : 0;
        }
#pragma unroll
        for (int u = 0; u < kPrefetch; u++) {
            const int k = k0 + u;
            if (k >= K) break;
            const uint32_t tick = rhead0 + (uint32_t)k;          // ring head (mod delay below)
            const uint64_t ptick = ptick0 + (uint64_t)k;
            const long o = (long)k * N + i;
            int action = act[u];
            if (pending) {           // next-step autoreset: this call is the env's reset(), :2250-2278
                uint32_t s0;
                if (PHILOX) {        // one word of the start-state streams per tick (mdpp_rng.hpp)
                    const uint64_t ge_ = (uint64_t)(a.env_id_offset + i);
                    s0 = (uint32_t)searchsorted_right(t.init_cdf, a.S, philox_start_uniform(
                        philox_start_m31(a.philox_seed, ge_, ptick, kPhiloxStartStream)));
                    if (IRR) cur1 = (uint32_t)searchsorted_right(init_cdf1, a.S1, philox_start_uniform(
                        philox_start_m31(a.philox_seed, ge_, ptick, kPhiloxStartIrrStream)));
                } else {
                    s0 = d_reset_draw(a, t, env_pcg);
                    if (IRR) cur1 = (uint32_t)searchsorted_right(init_cdf1, a.S1, np_random(env_pcg));
                }
                hist = DHist::fresh(s0);
                if (a.est.cur) est_roll(a.est, N, i, steps);                        // reset(): :2231-2247, :2360-2369
                steps = 0; phase = 0; ringbits = 0;
                if (!UNIT)
                    for (int d = 0; d < a.delay; d++) a.ring_keys[(size_t)d * N + i] = kNoKey;
                if (a.obs_i32) ((int32_t *)obs)[o * AW] = (int32_t)s0;
                else ((int64_t *)obs)[o * AW] = (int64_t)s0;
                if (IRR) {
                    if (a.obs_i32) ((int32_t *)obs)[o * AW + 1] = (int32_t)cur1;
                    else ((int64_t *)obs)[o * AW + 1] = (int64_t)cur1;
                }
                reward[o] = 0.0f; term[o] = 0; trunc[o] = 0;
                pending = false;
                continue;
            }
            if (action < 0 && action >= -A) action += A;       // numpy negative indexing
            if (action < 0 || action >= A) { status |= MDPP_STATUS_BAD_ACTION; action = 0; }
            const uint32_t cur = hist.cur();
            uint32_t nxt = ((const DPEntry *)t.P)[cur * A + action];                // D1
            if (NOISE && a.has_p_noise) {                                           // D2
                // (Philox streams: one word of the tick decides "noisy" and which other state, mdpp_rng.hpp philox_pnoise_*;
                //  numpy streams: the state space's own generator and the categorical's cdf, as in the reference)
                uint32_t noisy;
                if (PHILOX) noisy = philox_pnoise_state(pn_w.word(a.philox_seed, genv, ptick, kPhiloxPNoiseStream), a.pn_T, a.pn_M, nxt);
                else noisy = (uint32_t)searchsorted_right(t.noise_cdf + (size_t)nxt * S, S, np_random(sp_pcg));
                if (a.est.cur && noisy != nxt) est_add(a.est, N, i, 2, 1.0);        // total_noisy_transitions_episode, :1620
                nxt = noisy;
            }
            hist.push(nxt);                                                         // D3
            steps += 1;
            phase = (phase + 1 == (uint32_t)a.every_n) ? 0u : phase + 1;
            uint32_t key = kNoKey;                                                  // D4
            if (hist.at(L) != kDNaN) {
                key = 0;
                for (int j = L - 1; j >= 0; j--) key = key * S + hist.at(j);
            }
            // custom reward matrix: R(s, a) of this transition, whatever s' (noise included) was (:1259-1267)
            if (!UNIT && a.rew_sa) key = cur * (uint32_t)A + (uint32_t)action;
            const bool done = t.is_term[nxt] != 0;                                  // D7
            float rout;
            if (UNIT) {
                uint32_t bit = 0;
                if (key != kNoKey) bit = (t.rbits[key >> 3] >> (key & 7)) & 1u;
                if (a.delay > 0) {                                                  // D5 (shift register)
                    uint32_t out = (ringbits >> (a.delay - 1)) & 1u;
                    ringbits = (ringbits << 1) | bit;
                    bit = out;
                }
                if (phase != 0) bit = 0;                                            // D6
                if (a.est.cur && bit) est_add(a.est, N, i, 1, 1.0);                 // total_reward_episode, :1985
                if (NOISE && a.has_r_noise) {
                    double r = bit ? 1.0 : 0.0;
                    const double nz = 0.0 + a.r_noise * (PHILOX ? (double)rn_z.normal(a.philox_seed, genv, ptick, kPhiloxRNoiseStream) : np_standard_normal_lds(env_pcg, zig));
                    if (a.est.cur) est_add(a.est, N, i, 0, fabs(nz));               // total_abs_noise_in_reward_episode, :1984
                    r += nz;
                    r *= a.scale;
                    r += a.shift;
                    if (done) r += a.term_add;
                    rout = (float)r;
                } else {
                    rout = rsel[(bit << 1) | (done ? 1u : 0u)];
                }
            } else {
                if (a.delay > 0) {                                                  // D5 (key ring)
                    uint32_t *slot = a.ring_keys + (size_t)(tick % (uint32_t)a.delay) * N + i;
                    uint32_t out = *slot;
                    *slot = key;
                    key = out;
                }
                double r = (key != kNoKey) ? t.rtable[key] : 0.0;
                if (phase != 0) r = 0.0;
                if (a.est.cur) est_add(a.est, N, i, 1, r);                          // total_reward_episode, :1985
                if (NOISE && a.has_r_noise) {
                    const double nz = 0.0 + a.r_noise * (PHILOX ? (double)rn_z.normal(a.philox_seed, genv, ptick, kPhiloxRNoiseStream) : np_standard_normal_lds(env_pcg, zig));
                    if (a.est.cur) est_add(a.est, N, i, 0, fabs(nz));               // :1984
                    r += nz;
                }
                r *= a.scale;
                r += a.shift;
                if (done) r += a.term_add;
                rout = (float)r;
            }
            if (IRR) {                                                              // :2063-2082
                int action1 = act1[u];
                if (action1 < 0 && action1 >= -a.A1) action1 += a.A1;
                if (action1 < 0 || action1 >= a.A1) { status |= MDPP_STATUS_BAD_ACTION; action1 = 0; }
                uint32_t nxt1 = P1[cur1 * a.A1 + action1];
                if (NOISE && a.has_p_noise) {
                    if (PHILOX) nxt1 = philox_pnoise_state(pn_w1.word(a.philox_seed, genv, ptick, kPhiloxIrrStream), a.pn_T, a.pn_M1, nxt1);
                    else nxt1 = (uint32_t)searchsorted_right(a.noise_cdf1 + (size_t)nxt1 * a.S1, a.S1, np_random(sp1_pcg));
                }
                cur1 = nxt1;
            }
            const bool truncated = (a.max_steps > 0) && (steps >= (uint32_t)a.max_steps);
            uint32_t out_state = nxt;
            if (next_step) pending = done || truncated;
            if (a.autoreset == MDPP_AUTORESET_SAME_STEP && (done || truncated)) {
                // gymnasium "same-step" autoreset: report the terminal transition's reward/flags,
                // hand back the first observation of the next episode (reset(), :2250-2278).
                if (final_obs) {
                    if (a.obs_i32) ((int32_t *)final_obs)[o * AW] = (int32_t)nxt;
                    else ((int64_t *)final_obs)[o * AW] = (int64_t)nxt;
                    if (IRR) {
                        if (a.obs_i32) ((int32_t *)final_obs)[o * AW + 1] = (int32_t)cur1;
                        else ((int64_t *)final_obs)[o * AW + 1] = (int64_t)cur1;
                    }
                }
                uint32_t s0;
                if (PHILOX) {        // one word of the start-state streams per tick (mdpp_rng.hpp)
                    const uint64_t ge_ = (uint64_t)(a.env_id_offset + i);
                    s0 = (uint32_t)searchsorted_right(t.init_cdf, a.S, philox_start_uniform(
                        philox_start_m31(a.philox_seed, ge_, ptick, kPhiloxStartStream)));
                    if (IRR) cur1 = (uint32_t)searchsorted_right(init_cdf1, a.S1, philox_start_uniform(
                        philox_start_m31(a.philox_seed, ge_, ptick, kPhiloxStartIrrStream)));
                } else {
                    s0 = d_reset_draw(a, t, env_pcg);
                    if (IRR) cur1 = (uint32_t)searchsorted_right(init_cdf1, a.S1, np_random(env_pcg));
                }
                hist = DHist::fresh(s0);
                if (a.est.cur) est_roll(a.est, N, i, steps);                        // reset(): :2231-2247, :2360-2369
                steps = 0; phase = 0; ringbits = 0;
                if (!UNIT)
                    for (int d = 0; d < a.delay; d++) a.ring_keys[(size_t)d * N + i] = kNoKey;
                out_state = s0;
            }
            if (a.obs_i32) ((int32_t *)obs)[o * AW] = (int32_t)out_state;
            else ((int64_t *)obs)[o * AW] = (int64_t)out_state;
            if (IRR) {
                if (a.obs_i32) ((int32_t *)obs)[o * AW + 1] = (int32_t)cur1;
                else ((int64_t *)obs)[o * AW + 1] = (int64_t)cur1;
            }
            reward[o] = rout;
            term[o] = done ? 1 : 0;
            trunc[o] = truncated ? 1 : 0;
        }
    }

    a.state[i] = make_uint4((uint32_t)hist.lo, (uint32_t)(hist.lo >> 32), steps | (pending ? 0x80000000u : 0u), ringbits);
    hist.store_hi(a, i);
    if (!PHILOX) {
        if (use_env) env_pcg.store(a.env_s, i);
        if (use_sp) sp_pcg.store(a.sp_s, i);
        if (IRR && use_sp) sp1_pcg.store(a.sp1_s, i);
    }
    if (IRR) a.irr_state[i] = cur1;
    if (status) atomicOr(&a.status[i], status);
}

template <bool PHILOX>
__global__ __launch_bounds__(kBlock) void k_discrete_reset(DiscreteArgs a, uint64_t reset_tick,
                                                           const uint8_t *__restrict__ mask,
                                                           void *__restrict__ obs) {
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.N) return;
    if (mask && !mask[i]) return;
    DTables t;
    const size_t ti = a.shared_tables ? 0 : (size_t)i;
    t.init_cdf = a.init_cdf + ti * a.S;
    uint32_t s0, queue = 0, s1 = 0;
    const double *init_cdf1 = a.irr ? a.init_cdf1 + ti * a.S1 : nullptr;
    if (PHILOX) {
        Philox g;
        g.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), reset_tick, kPhiloxResetStream);
        s0 = d_reset_draw(a, t, g);
        if (a.irr) s1 = (uint32_t)searchsorted_right(init_cdf1, a.S1, np_random(g));      // :2259-2264
    } else {
        // fast-path handles keep start states drawn ahead of need in word 1 of the state record
        // (mdpp_discrete_fast.hip): consume those first, they are the next draws of the stream
        if (a.fast_ok) queue = a.state[i].y;
        if (a.fast_ok && ((queue >> 24) & 7u) != 0) {
            s0 = queue & 0xFu;
            queue = ((queue & 0x00FFFFFFu) >> 4) | ((((queue >> 24) & 7u) - 1u) << 24);
        } else {
            Pcg64 g;
            g.load(a.env_s, a.env_inc, i);
            s0 = d_reset_draw(a, t, g);
            if (a.irr) s1 = (uint32_t)searchsorted_right(init_cdf1, a.S1, np_random(g));  // :2259-2264
            g.store(a.env_s, i);
        }
    }
    if (a.irr) a.irr_state[i] = s1;
    if (a.est.cur) est_roll(a.est, a.N, i, a.state[i].z & 0x7FFFFFFFu);
    const DHist hist = DHist::fresh(s0);
    a.state[i] = make_uint4((uint32_t)hist.lo, a.fast_ok ? queue : (uint32_t)(hist.lo >> 32), 0u, 0u);
    hist.store_hi(a, i);
    if (!a.unit_rewards)
        for (int d = 0; d < a.delay; d++) a.ring_keys[(size_t)d * a.N + i] = kNoKey;
    if (obs) {
        const long w = a.irr ? 2 : 1;
        if (a.obs_i32) ((int32_t *)obs)[i * w] = (int32_t)s0;
        else ((int64_t *)obs)[i * w] = (int64_t)s0;
        if (a.irr) {
            if (a.obs_i32) ((int32_t *)obs)[i * w + 1] = (int32_t)s1;
            else ((int64_t *)obs)[i * w + 1] = (int64_t)s1;
        }
    }
}

template <bool PHILOX, bool NOISE>
static void launch_step_t(const DiscreteArgs &a, int K, const int32_t *actions, void *obs,
                          float *reward, uint8_t *term, uint8_t *trunc, void *final_obs,
                          hipStream_t s, char *name_out) {
    const int grid = (a.N + kBlock - 1) / kBlock;
    // (tables beyond the default dynamic-LDS limit -- a large action space -- are read from HBM / L2)
    // (one MDP per env: each lane's tables in its own LDS slot when 256 slots fit 96 KiB and the rollout is long enough to pay
    //  for the copy; named LDSTAB=2)
    const size_t env_lds = (size_t)kBlock * (a.lds_bytes + 8u);
    bool ldsenv = !a.shared_tables && a.rew_in_lds && !a.has_p_noise && !a.irr && K >= 8 && env_lds <= 96u * 1024u &&
                  !(a.opts & MDPP_OPT_NO_QUIET);
    bool ldstab = a.shared_tables && a.rew_in_lds && (!a.has_p_noise || a.noise_in_lds) && a.lds_bytes <= 48u * 1024u;
#ifdef MDPP_D_VARIANT
    ldsenv = ldstab = false;            // (wide: 16-bit table entries; long: up to S^15 keys -- read where they are, in HBM / L2)
#endif
    if (ldsenv) {                       // (also when only the name is asked for: the name is the launch's)
        const void *kern = a.unit_rewards ? (const void *)k_discrete_step<PHILOX, NOISE, true, true, false>
                                          : (const void *)k_discrete_step<PHILOX, NOISE, false, true, false>;
        if (!dynamic_lds_ok(kern, env_lds)) ldsenv = false;
    }
    if (ldsenv) ldstab = true;
    const size_t lds = ldsenv ? env_lds : (ldstab ? a.lds_bytes : 0);
    if (name_out) {
#ifdef MDPP_D_VARIANT
        snprintf(name_out, kNameLen, "k_discrete_step_" MDPP_D_VARIANT "<PHILOX=%d,NOISE=%d,UNIT=%d,IRR=%d>", PHILOX, NOISE, a.unit_rewards != 0, a.irr != 0);
#else
        snprintf(name_out, kNameLen, "k_discrete_step<PHILOX=%d,NOISE=%d,UNIT=%d,LDSTAB=%d,IRR=%d>", PHILOX, NOISE,
                 a.unit_rewards != 0, ldsenv ? 2 : (int)ldstab, a.irr != 0);
#endif
        return;
    }
#define MDPP_D_LAUNCH(UNIT, LDSTAB, IRR)                                                               \
    hipLaunchKernelGGL((k_discrete_step<PHILOX, NOISE, UNIT, LDSTAB, IRR>), dim3(grid), dim3(kBlock), \
                       lds, s, a, K, actions, obs, reward, term, trunc, final_obs)
#ifdef MDPP_D_VARIANT
    if (a.irr) { if (a.unit_rewards) MDPP_D_LAUNCH(true, false, true); else MDPP_D_LAUNCH(false, false, true); }
    else { if (a.unit_rewards) MDPP_D_LAUNCH(true, false, false); else MDPP_D_LAUNCH(false, false, false); }
#else
    if (a.irr) {
        if (a.unit_rewards) { if (ldstab) MDPP_D_LAUNCH(true, true, true); else MDPP_D_LAUNCH(true, false, true); }
        else { if (ldstab) MDPP_D_LAUNCH(false, true, true); else MDPP_D_LAUNCH(false, false, true); }
    } else {
        if (a.unit_rewards) { if (ldstab) MDPP_D_LAUNCH(true, true, false); else MDPP_D_LAUNCH(true, false, false); }
        else { if (ldstab) MDPP_D_LAUNCH(false, true, false); else MDPP_D_LAUNCH(false, false, false); }
    }
#endif
#undef MDPP_D_LAUNCH
}

#ifdef MDPP_D_VARIANT
int launch_discrete_step_var(mdpp_env *h, int K, const int32_t *actions, void *obs, float *reward,
                              uint8_t *term, uint8_t *trunc, void *final_obs, hipStream_t s, char *name_out) {
    DiscreteArgs a = h->dargs;
    a.opts = h->opts;
    a.ptick = h->tick;
    a.dtick = h->graph_capture ? (const uint64_t *)h->d_tick_off : nullptr;
    a.tick = a.delay > 0 ? (uint32_t)(h->tick % (uint64_t)a.delay) : 0u;
    const bool noise = a.has_p_noise || a.has_r_noise;
    if (a.philox) {
        if (noise) launch_step_t<true, true>(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
        else launch_step_t<true, false>(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
    } else {
        if (noise) launch_step_t<false, true>(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
        else launch_step_t<false, false>(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
    }
    if (name_out) return MDPP_OK;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("k_discrete_step_" MDPP_D_VARIANT " launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
    h->tick += (uint64_t)K;
    return MDPP_OK;
}

int launch_discrete_reset_var(mdpp_env *h, const uint8_t *mask, void *obs, hipStream_t s) {
    DiscreteArgs a = h->dargs;
    const int grid = (a.N + kBlock - 1) / kBlock;
    if (a.philox)
        hipLaunchKernelGGL(k_discrete_reset<true>, dim3(grid), dim3(kBlock), 0, s, a, h->reset_tick, mask, obs);
    else
        hipLaunchKernelGGL(k_discrete_reset<false>, dim3(grid), dim3(kBlock), 0, s, a, h->reset_tick, mask, obs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("k_discrete_reset_" MDPP_D_VARIANT " launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
    h->reset_tick += 1;
    return MDPP_OK;
}
#else

int launch_discrete_step(mdpp_env *h, int K, const int32_t *actions, void *obs, float *reward,
                         uint8_t *term, uint8_t *trunc, void *final_obs, hipStream_t s, char *name_out) {
    if (h->cfg.S > 255) return launch_discrete_step_wide(h, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
    if (h->cfg.L > 7) return launch_discrete_step_long(h, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
    DiscreteArgs a = h->dargs;
    a.opts = h->opts;
    // image handles: this is the state kernel of a batch of the image pipeline, which runs BESIDE the persistent renderer of the
    // previous batch (mdpp_capi.hip image_batches).  The role-split kernels want most of a CU's LDS and wait for the renderer's
    // workgroups to leave (k_discrete_rollout_pipe: 38 us alone, 308 us average beside the renderer, and the batch's draw and
    // record kernels queue behind it); the single-role kernel needs a kilobyte and always fits.
#ifndef MDPP_ABL_IMG_STATE_PIPE
    if (h->cfg.image) a.opts |= MDPP_OPT_NO_PIPE | MDPP_OPT_NO_LEAN;
#endif
    a.ptick = h->tick;
    a.dtick = h->graph_capture ? (const uint64_t *)h->d_tick_off : nullptr;     // (launches being captured into a HIP graph)
    a.tick = a.delay > 0 ? (uint32_t)(h->tick % (uint64_t)a.delay) : 0u;
    const bool noise = a.has_p_noise || a.has_r_noise;
    // Philox handles of the common shape: k_discrete_rollout_lean with its H waves on Philox blocks
    // (mdpp_discrete_lean.hip); pieces it does not take (a short last one) go to the quiet kernel
    bool lean_philox = false;
    if (!a.fast_ok && ((a.philox && (a.shape_ok || a.shape_ok_noise)) || a.shape_ok_noise_np || a.shape_ok_irr || a.lean_next_ok)) {
        const long long kmax = ((1LL << 32) - 1) / ((a.irr ? 16LL : 8LL) * a.N);
        char dry[kNameLen];
        const int k_first = (int)(K < kmax ? K : kmax);
        // (final observations are forwarded piece by piece like in the fast_ok branch below: the lean kernel declines a
        //  launch that asks for them, the quiet and general kernels write them -- ADVICE r2: nothing is dropped silently)
        if (kmax >= 32 && launch_discrete_lean(a, k_first, actions, obs, reward, term, trunc, final_obs, s, dry)) {
            lean_philox = true;
            const size_t osz = a.obs_i32 ? 4 : 8;
            for (int k0 = 0; k0 < K;) {
                const int kc = (int)((K - k0) < kmax ? (K - k0) : kmax);
                const size_t off = (size_t)k0 * a.N;
                a.ptick = h->tick + (uint64_t)k0;
                a.tick = a.delay > 0 ? (uint32_t)(a.ptick % (uint64_t)a.delay) : 0u;
                const size_t aoff = off * (a.irr ? 2 : 1);     // (an irrelevant sub-space: action pairs, observation pairs)
                char *op = (char *)obs + off * osz * (a.irr ? 2 : 1);
                void *fo = final_obs ? (void *)((char *)final_obs + off * osz * (a.irr ? 2 : 1)) : nullptr;
                if (!launch_discrete_lean(a, kc, actions + aoff, op, reward + off, term + off, trunc + off, fo, s, name_out) &&
                    !launch_discrete_quiet(a, kc, actions + aoff, op, reward + off, term + off, trunc + off, fo, s, name_out)) {
                    // (a short last piece of a next-step handle: the general kernel)
                    if (a.philox && noise) launch_step_t<true, true>(a, kc, actions + aoff, op, reward + off, term + off, trunc + off, fo, s, name_out);
                    else if (a.philox) launch_step_t<true, false>(a, kc, actions + aoff, op, reward + off, term + off, trunc + off, fo, s, name_out);
                    else if (noise) launch_step_t<false, true>(a, kc, actions + aoff, op, reward + off, term + off, trunc + off, fo, s, name_out);
                    else launch_step_t<false, false>(a, kc, actions + aoff, op, reward + off, term + off, trunc + off, fo, s, name_out);
                }
                if (name_out) return MDPP_OK;
                k0 += kc;
            }
        }
    }
    if (lean_philox) {
    } else if (K == 1 && launch_discrete_step1(a, h->s1args, actions, obs, reward, term, trunc, final_obs, s, name_out)) {
        // mdpp_step on the common shape: the kernel made for a launch of one step (mdpp_discrete_step1.hip)
    } else if (a.fast_ok) {
        // common shape: dedicated rollout kernel (mdpp_discrete_fast.hip); its buffer descriptors
        // address < 4 GiB per output array, so very long rollouts go out as several launches
        const long long kmax = ((1LL << 32) - 1) / (8LL * a.N);
        if (kmax < 1) { h->err = "k_discrete_rollout_fast: num_envs too large"; return MDPP_EUNSUPPORTED; }
        const size_t osz = a.obs_i32 ? 4 : 8;
        for (int k0 = 0; k0 < K;) {
            const int kc = (int)((K - k0) < kmax ? (K - k0) : kmax);
            const size_t off = (size_t)k0 * a.N;
            a.ptick = h->tick + (uint64_t)k0;
            a.tick = a.delay > 0 ? (uint32_t)(a.ptick % (uint64_t)a.delay) : 0u;
            void *fo = final_obs ? (void *)((char *)final_obs + off * osz) : nullptr;
            // long rollouts of full blocks: three-role pipelined kernel (mdpp_discrete_pipe.hip)
            // (at most 8 states: its leaner re-encoding, mdpp_discrete_lean.hip)
            if (!launch_discrete_lean(a, kc, actions + off, (char *)obs + off * osz, reward + off,
                                      term + off, trunc + off, fo, s, name_out) &&
                !launch_discrete_pipe(a, kc, actions + off, (char *)obs + off * osz, reward + off,
                                      term + off, trunc + off, fo, s, name_out))
                launch_discrete_fast(a, kc, actions + off, (char *)obs + off * osz, reward + off,
                                     term + off, trunc + off, fo, s, name_out);
            if (name_out) return MDPP_OK;     // (the first piece names the launch)
            k0 += kc;
        }
    } else if (!a.est.cur && launch_discrete_quiet(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out)) {
        // quiet shapes beyond the specialised kernels (larger S / L, irrelevant sub-space):
        // mdpp_discrete_quiet.hip
    } else if (a.philox) {
        if (noise) launch_step_t<true, true>(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
        else launch_step_t<true, false>(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
    } else {
        if (noise) launch_step_t<false, true>(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
        else launch_step_t<false, false>(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
    }
    if (name_out) return MDPP_OK;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("k_discrete_step launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
    h->tick += (uint64_t)K;
    return MDPP_OK;
}

int launch_discrete_reset(mdpp_env *h, const uint8_t *mask, void *obs, hipStream_t s) {
    if (h->cfg.S > 255) return launch_discrete_reset_wide(h, mask, obs, s);
    if (h->cfg.L > 7) return launch_discrete_reset_long(h, mask, obs, s);
    DiscreteArgs a = h->dargs;
    const int grid = (a.N + kBlock - 1) / kBlock;
    if (a.philox)
        hipLaunchKernelGGL(k_discrete_reset<true>, dim3(grid), dim3(kBlock), 0, s, a, h->reset_tick, mask, obs);
    else
        hipLaunchKernelGGL(k_discrete_reset<false>, dim3(grid), dim3(kBlock), 0, s, a, h->reset_tick, mask, obs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("k_discrete_reset launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
    h->reset_tick += 1;
    return MDPP_OK;
}
#endif   // MDPP_D_VARIANT

} // namespace mdpp

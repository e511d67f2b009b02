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

// One env step on registers.  Returns the reward (float64 like the reference's Python float).
template <class GE, class GS>
__device__ __forceinline__ double d_step_lane(const DiscreteArgs &a, const DTables &t,
                                              uint64_t &hist, uint32_t &steps, uint32_t &ringbits,
                                              uint32_t *ring_slot, int action, GE &env_rng,
                                              GS &space_rng, uint32_t &nxt_out, bool &done_out,
                                              uint32_t &status) {
    const int S = a.S, A = a.A, L = a.L;
    if (action < 0 && action >= -A) action += A;       // numpy negative indexing
    if (action < 0 || action >= A) { status |= MDPP_STATUS_BAD_ACTION; action = 0; }
    uint32_t cur = (uint32_t)(hist & 0xFF);
    uint32_t nxt = t.P[cur * A + action];                                   // D1
    if (a.has_p_noise) {                                                    // D2
        double u = np_random(space_rng);
        nxt = (uint32_t)searchsorted_right(t.noise_cdf + (size_t)nxt * S, S, u);
    }
    hist = (hist << 8) | nxt;                                               // D3
    steps += 1;
    uint32_t key = kNoKey;                                                  // D4
    if (((hist >> (8 * L)) & 0xFF) != 0xFF) {
        key = 0;
        for (int j = L - 1; j >= 0; j--) key = key * S + (uint32_t)((hist >> (8 * j)) & 0xFF);
    }
    double r;
    if (a.unit_rewards) {
        uint32_t bit = 0;
        if (key != kNoKey) bit = (t.rbits[key >> 3] >> (key & 7)) & 1u;
        if (a.delay > 0) {                                                  // D5 (shift register)
            uint32_t out = (ringbits >> (a.delay - 1)) & 1u;
            ringbits = (ringbits << 1) | bit;
            bit = out;
        }
        r = bit ? 1.0 : 0.0;
    } else {
        if (a.delay > 0) {                                                  // D5 (key ring)
            uint32_t out = *ring_slot;
            *ring_slot = key;
            key = out;
        }
        r = (key != kNoKey) ? t.rtable[key] : 0.0;
    }
    if (steps % (uint32_t)a.every_n != 0) r = 0.0;                          // D6
    if (a.has_r_noise) r += 0.0 + a.r_noise * np_standard_normal(env_rng);
    r *= a.scale;
    r += a.shift;
    bool done = t.is_term[nxt] != 0;                                        // D7
    if (done) r += a.term_add;
    nxt_out = nxt; done_out = done;
    return r;
}

__device__ __forceinline__ uint64_t d_fresh_hist(uint32_t s0) {
    return 0xFFFFFFFFFFFFFF00ULL | (uint64_t)s0;
}

template <bool PHILOX>
__global__ __launch_bounds__(kBlock) void k_discrete_step(DiscreteArgs a, int K,
                                                          const int32_t *__restrict__ actions,
                                                          void *__restrict__ obs,
                                                          float *__restrict__ reward,
                                                          uint8_t *__restrict__ term,
                                                          uint8_t *__restrict__ trunc,
                                                          void *__restrict__ final_obs) {
    extern __shared__ __align__(16) unsigned char lds[];
    DTables t;
    const int tid = threadIdx.x;
    const long i = (long)blockIdx.x * kBlock + tid;
    if (a.shared_tables) {
        // Stage the shared MDP into LDS: a few hundred bytes for 8x8 (P 64 B + flags 8 B +
        // reward bitmask 64 B + cdf 64 B).
        for (int k = tid; k < a.S * a.A; k += kBlock) lds[a.lds_P + k] = a.P[k];
        for (int k = tid; k < a.S; k += kBlock) {
            lds[a.lds_term + k] = a.is_term[k];
            ((double *)(lds + a.lds_init))[k] = a.init_cdf[k];
        }
        if (a.rew_in_lds) {
            if (a.unit_rewards)
                for (uint32_t k = tid; k < a.rbits_stride; k += kBlock) lds[a.lds_rew + k] = a.rbits[k];
            else
                for (uint32_t k = tid; k < a.nkeys; k += kBlock)
                    ((double *)(lds + a.lds_rew))[k] = a.rtable[k];
        }
        if (a.has_p_noise && a.noise_in_lds)
            for (int k = tid; k < a.S * a.S; k += kBlock)
                ((double *)(lds + a.lds_noise))[k] = a.noise_cdf[k];
        __syncthreads();
        t.P = lds + a.lds_P;
        t.is_term = lds + a.lds_term;
        t.init_cdf = (const double *)(lds + a.lds_init);
        t.rbits = a.rew_in_lds ? lds + a.lds_rew : a.rbits;
        t.rtable = a.rew_in_lds ? (const double *)(lds + a.lds_rew) : a.rtable;
        t.noise_cdf = (a.has_p_noise && a.noise_in_lds) ? (const double *)(lds + a.lds_noise) : a.noise_cdf;
    }
    if (i >= a.N) return;
    if (!a.shared_tables) {
        t.P = a.P + (size_t)i * a.S * a.A;
        t.is_term = a.is_term + (size_t)i * a.S;
        t.init_cdf = a.init_cdf + (size_t)i * a.S;
        t.rbits = a.rbits + (size_t)i * a.rbits_stride;
        t.rtable = a.rtable + (size_t)i * a.nkeys;
        t.noise_cdf = a.noise_cdf;
    }

    uint4 st = a.state[i];
    uint64_t hist = ((uint64_t)st.y << 32) | st.x;
    uint32_t steps = st.z, ringbits = st.w, status = 0;

    Pcg64 env_pcg, sp_pcg;
    Philox env_phx, sp_phx;
    bool env_loaded = false;
    if (!PHILOX) {
        if (a.has_r_noise) { env_pcg.load(a.env_s, a.env_inc, i); env_loaded = true; }
        if (a.has_p_noise) sp_pcg.load(a.sp_s, a.sp_inc, i);
    }

    for (int k = 0; k < K; k++) {
        const uint32_t tick = a.tick + (uint32_t)k;
        const long o = (long)k * a.N + i;
        int action = actions[o];
        uint32_t *slot = nullptr;
        if (!a.unit_rewards && a.delay > 0)
            slot = a.ring_keys + (size_t)(tick % (uint32_t)a.delay) * a.N + i;
        uint32_t nxt; bool done; double r;
        if (PHILOX) {
            env_phx.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), tick, MDPP_STREAM_ENV);
            sp_phx.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), tick, MDPP_STREAM_SPACE);
            r = d_step_lane(a, t, hist, steps, ringbits, slot, action, env_phx, sp_phx, nxt, done, status);
        } else {
            r = d_step_lane(a, t, hist, steps, ringbits, slot, action, env_pcg, sp_pcg, nxt, done, status);
        }
        bool truncated = (a.max_steps > 0) && (steps >= (uint32_t)a.max_steps);
        uint32_t out_state = nxt;
        if (a.autoreset && (done || truncated)) {
            // gymnasium "same-step" autoreset: report the terminal transition's reward/flags,
            // hand back the first observation of the next episode (reset(), :2250-2278).
            if (final_obs) {
                if (a.obs_i32) ((int32_t *)final_obs)[o] = (int32_t)nxt;
                else ((int64_t *)final_obs)[o] = (int64_t)nxt;
            }
            uint32_t s0;
            if (PHILOX) {
                s0 = d_reset_draw(a, t, env_phx);
            } else {
                if (!env_loaded) { env_pcg.load(a.env_s, a.env_inc, i); env_loaded = true; }
                s0 = d_reset_draw(a, t, env_pcg);
            }
            hist = d_fresh_hist(s0);
            steps = 0; ringbits = 0;
            if (!a.unit_rewards)
                for (int d = 0; d < a.delay; d++) a.ring_keys[(size_t)d * a.N + i] = kNoKey;
            out_state = s0;
        }
        if (a.obs_i32) ((int32_t *)obs)[o] = (int32_t)out_state;
        else ((int64_t *)obs)[o] = (int64_t)out_state;
        reward[o] = (float)r;
        term[o] = done ? 1 : 0;
        trunc[o] = truncated ? 1 : 0;
    }

    a.state[i] = make_uint4((uint32_t)hist, (uint32_t)(hist >> 32), steps, ringbits);
    if (!PHILOX) {
        if (env_loaded) env_pcg.store(a.env_s, i);
        if (a.has_p_noise) sp_pcg.store(a.sp_s, i);
    }
    if (status) atomicOr(&a.status[i], status);
}

template <bool PHILOX>
__global__ __launch_bounds__(kBlock) void k_discrete_reset(DiscreteArgs a, uint32_t reset_tick,
                                                           const uint8_t *__restrict__ mask,
                                                           void *__restrict__ obs) {
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.N) return;
    if (mask && !mask[i]) return;
    DTables t;
    const size_t ti = a.shared_tables ? 0 : (size_t)i;
    t.init_cdf = a.init_cdf + ti * a.S;
    uint32_t s0;
    if (PHILOX) {
        Philox g;
        g.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), reset_tick, MDPP_NUM_STREAMS);
        s0 = d_reset_draw(a, t, g);
    } else {
        Pcg64 g;
        g.load(a.env_s, a.env_inc, i);
        s0 = d_reset_draw(a, t, g);
        g.store(a.env_s, i);
    }
    uint64_t hist = d_fresh_hist(s0);
    a.state[i] = make_uint4((uint32_t)hist, (uint32_t)(hist >> 32), 0u, 0u);
    if (!a.unit_rewards)
        for (int d = 0; d < a.delay; d++) a.ring_keys[(size_t)d * a.N + i] = kNoKey;
    if (obs) {
        if (a.obs_i32) ((int32_t *)obs)[i] = (int32_t)s0;
        else ((int64_t *)obs)[i] = (int64_t)s0;
    }
}

int launch_discrete_step(mdpp_env *h, int K, const int32_t *actions, void *obs, float *reward,
                         uint8_t *term, uint8_t *trunc, void *final_obs, hipStream_t s) {
    DiscreteArgs a = h->dargs;
    a.tick = h->tick;
    const int grid = (a.N + kBlock - 1) / kBlock;
    const size_t lds = a.shared_tables ? a.lds_bytes : 0;
    if (a.philox)
        hipLaunchKernelGGL(k_discrete_step<true>, dim3(grid), dim3(kBlock), lds, s, a, K, actions,
                           obs, reward, term, trunc, final_obs);
    else
        hipLaunchKernelGGL(k_discrete_step<false>, dim3(grid), dim3(kBlock), lds, s, a, K, actions,
                           obs, reward, term, trunc, final_obs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("k_discrete_step launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
    h->tick += (uint32_t)K;
    return MDPP_OK;
}

int launch_discrete_reset(mdpp_env *h, const uint8_t *mask, void *obs, hipStream_t s) {
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

} // namespace mdpp

// Grid RLToyEnv.step()/reset() for gfx950: one lane per env instance.
//
// Restates /root/reference/mdp_playground/envs/rl_toy_env.py
//   transition_function, grid branch :1727-1778   reward_function, grid branch :1947-1965
//   post-processing :1968-1990, :2102-2109        reset, grid branch           :2325-2345
// and GridActionSpace (contains / sample), spaces/grid_action_space.py:13-39.
//
// What a step is: an action is a vector of grid_dims entries, all zero or exactly one +-1; anything
// else is "outside the action space" and applied as a noop (:1763-1768, status bit).  With
// transition noise p the env generator draws one uniform per step and, below p, the action is
// re-drawn from the action space's own generator (integers(G), integers(3): numpy's buffered
// 32-bit Lemire draws) until it differs from the given one.  The move is clipped to the grid, the
// target latch is set when the first two coordinates equal the target, the reward is the
// Manhattan distance gained (dense) or 1 on the target (sparse), then every-n mask, Gaussian
// noise, scale, shift and the terminal reward as for the other env types.
//
// Data layout (HBM): state[N] one 16-byte record per env {cells: one byte per dimension, steps,
// flags}; PCG64 streams as ulonglong2 per env; the action stream also carries numpy's buffered
// 32-bit half.  Algorithmic bytes per env step of a fused rollout: 4 G (actions) + 8 G (int64
// observation) + 4 (reward) + 2 (flags) = 30 B for a plain 2-D grid.
#include <type_traits>

#include "mdpp_internal.hpp"
#include "mdpp_rng.hpp"
#include <cstdlib>

namespace mdpp {

template <class G>
__device__ __forceinline__ void g_reset_lane(const GridArgs &a, G &sp, int (&cell)[4]) {
    // feature_space.sample() of Box(0, grid_shape, int64): floor(uniform(0, g + 1)) per dimension
    // (:780-788, :2326) -- the index g, one past the grid, can come out, as in the reference.
#pragma unroll
    for (int d = 0; d < 4; d++)
        if (d < a.G) cell[d] = (int)floor(0.0 + ((double)(a.shape[d] + 1) - 0.0) * np_random(sp));
}

// NOISE: transition and/or reward noise configured.  The RNG code is large (inlined numpy
// ziggurat, Lemire draws); the quiet variant leaves it out and unrolls the prefetch ring 4 deep, the
// noisy one keeps the loop body small enough for the instruction cache.
template <bool PHILOX, bool NOISE>
__global__ __launch_bounds__(kBlock) void k_grid_step(GridArgs a, int K, const int32_t *__restrict__ actions,
                                                      void *__restrict__ obs, float *__restrict__ reward,
                                                      uint8_t *__restrict__ term, uint8_t *__restrict__ trunc,
                                                      void *__restrict__ final_obs) {
    const uint64_t ptick0 = tick_now(a);               // the step counter at this launch (through the device-side offset of a graph replay)
    __shared__ uint64_t s_ki[256];
    __shared__ double s_wi[256], s_fi[256];
    if (NOISE && a.has_r_noise) { zig_stage(s_ki, s_wi, s_fi, threadIdx.x, kBlock); __syncthreads(); }
    const ZigLds zig{s_ki, s_wi, s_fi};
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.N) return;
    const long N = a.N;
    const int G = a.G;
    const uint4 st = a.state[i];
    int cell[4] = {(int)(st.x & 0xFF), (int)((st.x >> 8) & 0xFF), (int)((st.x >> 16) & 0xFF), (int)(st.x >> 24)};
    uint32_t steps = st.y, flags = st.z, status = 0;
    // next-step autoreset: "episode ended, reset at the next call" travels in bit 1 of the flags word
    const bool next_step = a.autoreset == MDPP_AUTORESET_NEXT_STEP;
    bool pending = next_step && (flags & 2u) != 0;
    flags &= ~2u;

    Pcg64 env_pcg, sp_pcg, act_pcg;
    Philox env_phx, sp_phx, act_phx;
    Half32 act_h{0, 0}, phx_h{0, 0};
    bool sp_loaded = false;
    const bool need_env = NOISE && (a.has_p_noise || a.has_r_noise);
    if (!PHILOX) {
        if (need_env) env_pcg.load(a.env_s, a.env_inc, i);
        if (NOISE && a.has_p_noise) {
            act_pcg.load(a.act_s, a.act_inc, i);
            const uint2 hh = a.act_half[i];
            act_h = Half32{hh.x, hh.y};
        }
    }
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef long i64x2 __attribute__((ext_vector_type(2)));
    auto put_obs = [&](void *dst, long o, const int (&c)[4]) {     // one or two 16-byte stores per lane
        if (a.obs_i32) {
            if (G == 2) *(i32x2 *)((int32_t *)dst + o * 2) = i32x2{c[0], c[1]};
            else *(i32x4 *)((int32_t *)dst + o * 4) = i32x4{c[0], c[1], c[2], c[3]};
        } else {
            *(i64x2 *)((int64_t *)dst + o * G) = i64x2{(long)c[0], (long)c[1]};
            if (G == 4) *(i64x2 *)((int64_t *)dst + o * 4 + 2) = i64x2{(long)c[2], (long)c[3]};
        }
    };

    // software pipeline on the action stream (one wave per SIMD at 65 536 envs: a load per step
    // would cost its full latency): kGPrefetch steps in flight per lane, as one 8- or 16-byte load
    constexpr int kGPrefetch = NOISE ? 2 : 4;
    auto load_act = [&](int k, int (&dst)[4]) {
        const long o = (long)min(k, K - 1) * N + i;
        if (G == 2) { const i32x2 v = *(const i32x2 *)(actions + o * 2); dst[0] = v.x; dst[1] = v.y; dst[2] = dst[3] = 0; }
        else { const i32x4 v = *(const i32x4 *)(actions + o * 4); dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
    };
    int pre[kGPrefetch][4];
#pragma unroll
    for (int u = 0; u < kGPrefetch; u++) load_act(u, pre[u]);
    // reset() of this lane (same-step autoreset after a finished episode; next-step: the call after it)
    auto lane_reset = [&]() __attribute__((always_inline)) {
        if (PHILOX) g_reset_lane(a, sp_phx, cell);
        else {
            if (!sp_loaded) { sp_pcg.load(a.sp_s, a.sp_inc, i); sp_loaded = true; }
            g_reset_lane(a, sp_pcg, cell);
        }
        if (a.est.cur) est_roll(a.est, N, i, steps);                       // reset(): :2231-2247, :2360-2369
        steps = 0; flags = 0;
    };

    for (int k0 = 0; k0 < K; k0 += kGPrefetch) {
#pragma unroll
      for (int u = 0; u < kGPrefetch; u++) {
        const int k = k0 + u;
        if (k >= K) break;
        int act[4] = {pre[u][0], pre[u][1], pre[u][2], pre[u][3]};
        load_act(k + kGPrefetch, pre[u]);
        const uint64_t tick = ptick0 + (uint64_t)k;
        const long o = (long)k * N + i;
        if (PHILOX) {
            env_phx.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), tick, MDPP_STREAM_ENV);
            sp_phx.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), tick, MDPP_STREAM_SPACE);
            act_phx.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), tick, kPhiloxActionStream);
            phx_h = Half32{0, 0};
        }
        if (pending) {               // next-step autoreset: this call is the env's reset(), :2325-2345
            lane_reset();
            put_obs(obs, o, cell);
            reward[o] = 0.0f; term[o] = 0; trunc[o] = 0;
            pending = false;
            continue;
        }
        bool ok = true;
        int nz = 0;
#pragma unroll
        for (int d = 0; d < 4; d++)
            if (d < G) {
                ok = ok && act[d] >= -1 && act[d] <= 1;
                nz += act[d] != 0;
            }
        ok = ok && nz <= 1;                                   // GridActionSpace.contains
        const int old0 = cell[0], old1 = cell[1];
        if (ok) {
            if (NOISE && a.has_p_noise) {                     // :1733-1749
                const double u = PHILOX ? np_random(env_phx) : np_random(env_pcg);
                if (u < a.p_noise) {
                    for (int tries = 0;; tries++) {
                        const int ind = PHILOX ? np_integers(act_phx, phx_h, 0, G) : np_integers(act_pcg, act_h, 0, G);
                        const int val = PHILOX ? np_integers(act_phx, phx_h, 0, 3) : np_integers(act_pcg, act_h, 0, 3);
                        bool same = true;
#pragma unroll
                        for (int d = 0; d < 4; d++)
                            if (d < G) same = same && ((d == ind ? val - 1 : 0) == act[d]);
                        if (!same) {
#pragma unroll
                            for (int d = 0; d < 4; d++) act[d] = (d == ind) ? val - 1 : 0;
                            if (a.est.cur) est_add(a.est, N, i, 2, 1.0);      // total_noisy_transitions_episode, :1746
                            break;
                        }
                        if (tries > 4096) { status |= MDPP_STATUS_INTERNAL; break; }
                    }
                }
            }
#pragma unroll
            for (int d = 0; d < 4; d++)
                if (d < G) cell[d] = min(max(cell[d] + act[d], 0), a.shape[d] - 1);     // :1751-1761
        } else {
            status |= MDPP_STATUS_BAD_ACTION;                 // noop, :1763-1768
        }
        const bool on_target = cell[0] == a.target[0] && cell[1] == a.target[1];
        if (on_target) flags |= 1u;                           // reached_terminal latches, :1770-1776
        steps += 1;
        double r = 0.0;
        if (a.make_denser)                                    // :1949-1960
            r += (double)((abs(old0 - a.target[0]) + abs(old1 - a.target[1])) -
                          (abs(cell[0] - a.target[0]) + abs(cell[1] - a.target[1])));
        else if (on_target) r += 1.0;                         // :1962-1965
        if (a.every_n != 1 && steps % (uint32_t)a.every_n != 0) r = 0.0;   // :1975-1978
        if (a.est.cur) est_add(a.est, N, i, 1, r);                         // total_reward_episode, :1985
        if (NOISE && a.has_r_noise) {
            const double nz = 0.0 + a.r_noise * (PHILOX ? np_standard_normal_lds(env_phx, zig) : np_standard_normal_lds(env_pcg, zig));
            if (a.est.cur) est_add(a.est, N, i, 0, fabs(nz));              // total_abs_noise_in_reward_episode, :1984
            r += nz;
        }
        r *= a.scale;
        r += a.shift;
        const bool done = (flags & 1u) != 0;                  // :2102-2104
        if (done) r += a.term_add;
        const bool truncated = (a.max_steps > 0) && (steps >= (uint32_t)a.max_steps);
        if (next_step) pending = done || truncated;
        if (a.autoreset == MDPP_AUTORESET_SAME_STEP && (done || truncated)) {
            if (final_obs) put_obs(final_obs, o, cell);
            lane_reset();
        }
        put_obs(obs, o, cell);
        reward[o] = (float)r;
        term[o] = done ? 1 : 0;
        trunc[o] = truncated ? 1 : 0;
      }
    }
    a.state[i] = make_uint4((uint32_t)cell[0] | ((uint32_t)cell[1] << 8) | ((uint32_t)cell[2] << 16) |
                                ((uint32_t)cell[3] << 24), steps, flags | (pending ? 2u : 0u), 0u);
    if (!PHILOX) {
        if (need_env) env_pcg.store(a.env_s, i);
        if (sp_loaded) sp_pcg.store(a.sp_s, i);
        if (NOISE && a.has_p_noise) {
            act_pcg.store(a.act_s, i);
            a.act_half[i] = make_uint2(act_h.has32, act_h.u32);
        }
    }
    if (status) atomicOr(&a.status[i], status);
}

template <bool PHILOX>
__global__ __launch_bounds__(kBlock) void k_grid_reset(GridArgs a, uint64_t reset_tick,
                                                       const uint8_t *__restrict__ mask, void *__restrict__ obs) {
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.N) return;
    if (mask && !mask[i]) return;
    int cell[4] = {0, 0, 0, 0};
    if (PHILOX) {
        Philox g;
        g.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), reset_tick, kPhiloxResetStream);
        g_reset_lane(a, g, cell);
    } else {
        Pcg64 g;
        g.load(a.sp_s, a.sp_inc, i);
        g_reset_lane(a, g, cell);
        g.store(a.sp_s, i);
    }
    if (a.est.cur) est_roll(a.est, a.N, i, a.state[i].y);
    a.state[i] = make_uint4((uint32_t)cell[0] | ((uint32_t)cell[1] << 8) | ((uint32_t)cell[2] << 16) |
                                ((uint32_t)cell[3] << 24), 0u, 0u, 0u);
    if (obs)
        for (int d = 0; d < a.G; d++) {
            if (a.obs_i32) ((int32_t *)obs)[i * a.G + d] = cell[d];
            else ((int64_t *)obs)[i * a.G + d] = (int64_t)cell[d];
        }
}

// ---- fused rollout for the quiet case (numpy streams, no noise, K steps per launch) -------------
// k_grid_step spends most of a step on things this shape does not need: wave-uniform branches on run
// time flags (a taken branch is an unhidden refetch with one wave per SIMD), and the in-step reset(),
// which runs its two PCG64 steps for the whole wave whenever ANY lane reached the target (~60 % of
// the steps of a 64-lane wave under a random policy).  Here the flags are template parameters, the
// step body is straight-line, and start cells are drawn AHEAD of need into a per-lane register
// queue that is topped up for all lanes at once when some lane runs dry; what is left in the queue
// at the end of the launch is un-drawn (inverse LCG step), so the stream position is again exactly
// the reference's.  Every global access is a buffer instruction with a per-lane offset that never
// changes plus a per-step scalar offset.
constexpr int kGQ = 4;                        // start cells queued per lane
constexpr int kGRsrc = 0x00020000;

// PN / RN: transition noise (a uniform from the env stream per admitted action; below the threshold
// the action is re-drawn from the action space's stream until it differs, :1733-1749) and reward
// noise (a normal from the env stream, :1980); reset() draws from the feature space's stream, so
// the start-cell queue is untouched by either.
// PH: Philox streams -- every step re-keys its three generators by (seed, global env id, tick, stream), start
// cells are drawn when an episode ends (no queue, nothing to un-draw), nothing is loaded from or stored to HBM.
template <bool OBS64, bool G4, bool DENSE, bool PN, bool RN, bool PH = false>
__global__ __launch_bounds__(kBlock) void k_grid_rollout_fast(GridArgs a, int K, const int32_t *__restrict__ actions,
                                                              void *__restrict__ obs, float *__restrict__ reward,
                                                              uint8_t *__restrict__ term, uint8_t *__restrict__ trunc,
                                                              void *__restrict__ final_obs) {
    const uint64_t ptick0 = tick_now(a);               // the step counter at this launch (through the device-side offset of a graph replay)
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    constexpr int G = G4 ? 4 : 2;
    constexpr bool ZIG = RN && !PH;
    __shared__ uint64_t s_ki[ZIG ? 256 : 1];
    __shared__ double s_wi[ZIG ? 256 : 1], s_fi[ZIG ? 256 : 1];
    if (ZIG) { zig_stage(s_ki, s_wi, s_fi, threadIdx.x, kBlock); __syncthreads(); }
    const ZigLds zig{s_ki, s_wi, s_fi};
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= (uint32_t)a.N) return;
    const uint32_t N = (uint32_t)a.N;
    const uint4 st = a.state[i];
    uint32_t cells = st.x, steps = st.y, flags = st.z, status = 0;   // one byte per dimension
    // next-step autoreset (bit 1 of the flags word, as in k_grid_step): the call after an episode's last step is the
    // reset -- action ignored, nothing drawn from the noise streams, reward 0, no flags
    const bool nextmode = a.autoreset == MDPP_AUTORESET_NEXT_STEP;
    bool pend = nextmode && (flags & 2u) != 0u;
    flags &= ~2u;
    typedef typename std::conditional<PH, Philox, Pcg64>::type Gen;
    Gen sp, env, actg;
    Half32 acth{0, 0};
    const uint64_t genv = (uint64_t)(a.env_id_offset + (int64_t)i);     // global env id (Philox key)
    if constexpr (!PH) {
        sp.load(a.sp_s, a.sp_inc, i);
        if (PN || RN) env.load(a.env_s, a.env_inc, i);
        if (PN) {
            actg.load(a.act_s, a.act_inc, i);
            const uint2 hh = a.act_half[i];
            acth = Half32{hh.x, hh.y};
        }
    }
    uint32_t queue[kGQ];                          // queued start cells, queue[0] next; qn of them valid
#pragma unroll
    for (int q = 0; q < kGQ; q++) queue[q] = 0;
    uint32_t qn = 0;
    const double rng0 = (double)(a.shape[0] + 1), rng1 = (double)(a.shape[1] + 1);
    const double rng2 = (double)(a.shape[2] + 1), rng3 = (double)(a.shape[3] + 1);
    auto draw_cell = [&]() __attribute__((always_inline)) -> uint32_t {          // feature_space.sample(): floor(uniform(0, g + 1)) per dimension
        uint32_t c = (uint32_t)(int)floor(0.0 + (rng0 - 0.0) * np_random(sp));
        c |= (uint32_t)(int)floor(0.0 + (rng1 - 0.0) * np_random(sp)) << 8;
        if (G4) {
            c |= (uint32_t)(int)floor(0.0 + (rng2 - 0.0) * np_random(sp)) << 16;
            c |= (uint32_t)(int)floor(0.0 + (rng3 - 0.0) * np_random(sp)) << 24;
        }
        return c;
    };
    // (always_inline: as a real function it would take the stream and the queue by reference, i.e.
    // keep them in scratch memory for the whole kernel)
    auto refill = [&]() __attribute__((always_inline)) {     // top every lane's queue up, most lanes active in every round
        for (int round = 0; round < kGQ; round++) {
            if (__builtin_amdgcn_ballot_w64(qn < (uint32_t)kGQ) == 0) break;
            if (qn < (uint32_t)kGQ) {
                const uint32_t c = draw_cell();
#pragma unroll
                for (int q = 0; q < kGQ; q++) queue[q] = (qn == (uint32_t)q) ? c : queue[q];
                qn++;
            }
        }
    };
    if (a.autoreset && !PH) refill();

    const uint32_t total = (uint32_t)K * N;
    auto r_act = __builtin_amdgcn_make_buffer_rsrc((void *)actions, 0, total * (uint32_t)(G * 4), kGRsrc);
    auto r_obs = __builtin_amdgcn_make_buffer_rsrc(obs, 0, total * (uint32_t)(G * (OBS64 ? 8 : 4)), kGRsrc);
    auto r_fin = __builtin_amdgcn_make_buffer_rsrc(final_obs ? final_obs : obs, 0, total * (uint32_t)(G * (OBS64 ? 8 : 4)), kGRsrc);
    auto r_rew = __builtin_amdgcn_make_buffer_rsrc((void *)reward, 0, total * 4u, kGRsrc);
    auto r_term = __builtin_amdgcn_make_buffer_rsrc((void *)term, 0, total, kGRsrc);
    auto r_trunc = __builtin_amdgcn_make_buffer_rsrc((void *)trunc, 0, total, kGRsrc);
    const uint32_t vact = i * (uint32_t)(G * 4), vobs = i * (uint32_t)(G * (OBS64 ? 8 : 4)), v4 = i * 4u, v1 = i;
    const uint32_t row_act = N * (uint32_t)(G * 4), row_obs = N * (uint32_t)(G * (OBS64 ? 8 : 4));
    const int t0 = a.target[0], t1 = a.target[1];
    const int m0 = a.shape[0] - 1, m1 = a.shape[1] - 1, m2 = a.shape[2] - 1, m3 = a.shape[3] - 1;
    const bool has_max = a.max_steps > 0, autoreset = a.autoreset != 0;
    const uint32_t max_steps = (uint32_t)a.max_steps, every_n = (uint32_t)a.every_n;
    uint32_t phase = steps % every_n;

    constexpr int kPre = 8;
    auto load_act = [&](int k) -> i32x4 {
        const uint32_t kk = (uint32_t)min(k, K - 1);
        if (G4) return __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(r_act, vact, kk * row_act, 0));
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r_act, vact, kk * row_act, 0);
        return i32x4{(int)v.x, (int)v.y, 0, 0};
    };
    auto put_cells = [&](decltype(r_obs) rs, uint32_t so, uint32_t c) {
        const uint32_t c0 = c & 0xFFu, c1 = (c >> 8) & 0xFFu, c2 = (c >> 16) & 0xFFu, c3 = c >> 24;
        if (OBS64) {
            // (128-bit stores: whole offset in the VGPR, see mdpp_discrete_quiet.hip on the store-data hazard)
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{c0, 0u, c1, 0u}, rs, vobs + so * row_obs, 0, MDPP_ST_NT);
            if (G4) __builtin_amdgcn_raw_buffer_store_b128(u32x4{c2, 0u, c3, 0u}, rs, vobs + 16u + so * row_obs, 0, MDPP_ST_NT);
        } else {
            if (G4) __builtin_amdgcn_raw_buffer_store_b128(u32x4{c0, c1, c2, c3}, rs, vobs + so * row_obs, 0, MDPP_ST_NT);
            else __builtin_amdgcn_raw_buffer_store_b64(u32x2{c0, c1}, rs, vobs, so * row_obs, MDPP_ST_NT);
        }
    };
    i32x4 pre[kPre];
#pragma unroll
    for (int u = 0; u < kPre; u++) pre[u] = load_act(u);

    // straight-line step body: the run-time options (autoreset, reward_every_n_steps, episode limit)
    // are folded into selects rather than branches
    auto step = [&](const int k, i32x4 act) __attribute__((always_inline)) {
        // every lane must hold a start cell before the step may end its episode; the refill tops
        // all lanes up to kGQ, so this branch is taken once in several dozen steps
        if constexpr (PH) {
            const uint64_t tick = ptick0 + (uint64_t)k;
            env.init(a.philox_seed, genv, tick, MDPP_STREAM_ENV);
            sp.init(a.philox_seed, genv, tick, MDPP_STREAM_SPACE);
            actg.init(a.philox_seed, genv, tick, kPhiloxActionStream);
            acth = Half32{0, 0};
        } else if (__builtin_expect(__builtin_amdgcn_ballot_w64(autoreset && qn == 0u) != 0, 0)) refill();
        // GridActionSpace.contains: every entry in {-1, 0, 1}, at most one non-zero
        const uint32_t a0 = (uint32_t)(act.x + 1), a1 = (uint32_t)(act.y + 1), a2 = (uint32_t)(act.z + 1), a3 = (uint32_t)(act.w + 1);
        const int nz = (act.x != 0) + (act.y != 0) + (G4 ? (act.z != 0) + (act.w != 0) : 0);
        const bool ok = a0 <= 2u && a1 <= 2u && (!G4 || (a2 <= 2u && a3 <= 2u)) && nz <= 1;
        status |= (ok || pend) ? 0u : (uint32_t)MDPP_STATUS_BAD_ACTION;
        if (PN) {
            bool redraw = false;
            if (ok && !pend) redraw = np_random(env) < a.p_noise;
            if (__builtin_amdgcn_ballot_w64(redraw) != 0) {
                if (redraw) {
                    for (int tries = 0;; tries++) {
                        const int ind = np_integers(actg, acth, 0, G);
                        const int val = np_integers(actg, acth, 0, 3) - 1;
                        const i32x4 cand{ind == 0 ? val : 0, ind == 1 ? val : 0, ind == 2 ? val : 0, ind == 3 ? val : 0};
                        const bool same = cand.x == act.x && cand.y == act.y && (!G4 || (cand.z == act.z && cand.w == act.w));
                        if (!same) { act = cand; break; }
                        if (tries > 4096) { status |= MDPP_STATUS_INTERNAL; break; }
                    }
                }
            }
        }
        const int c0 = (int)(cells & 0xFFu), c1 = (int)((cells >> 8) & 0xFFu);
        const int c2 = (int)((cells >> 16) & 0xFFu), c3 = (int)(cells >> 24);
        const int n0 = ok ? min(max(c0 + act.x, 0), m0) : c0, n1 = ok ? min(max(c1 + act.y, 0), m1) : c1;
        const int n2 = (G4 && ok) ? min(max(c2 + act.z, 0), m2) : c2, n3 = (G4 && ok) ? min(max(c3 + act.w, 0), m3) : c3;
        const bool on_target = n0 == t0 && n1 == t1;
        flags |= on_target ? 1u : 0u;
        steps += 1;
        double r = 0.0;
        if (DENSE) r += (double)((abs(c0 - t0) + abs(c1 - t1)) - (abs(n0 - t0) + abs(n1 - t1)));
        else r += on_target ? 1.0 : 0.0;
        phase = (phase + 1 >= every_n) ? 0u : phase + 1;          // steps % every_n, carried
        r = phase != 0 ? 0.0 : r;
        if (RN) { if (!pend) r += 0.0 + a.r_noise * np_standard_normal_lds(env, zig); }
        r *= a.scale;
        r += a.shift;
        bool done = (flags & 1u) != 0;
        if (RN) { if (done) r += a.term_add; }    // (with noise r can be -0.0: add only where the reference does)
        else r += done ? a.term_add : 0.0;        // r is not -0.0 here (a +0.0 shift was just added), so + 0.0 is the identity
        bool tr = has_max && steps >= max_steps;
        uint32_t nc = (uint32_t)n0 | ((uint32_t)n1 << 8) | ((uint32_t)n2 << 16) | ((uint32_t)n3 << 24);
        const uint32_t so = (uint32_t)k;
        bool need = autoreset && (done || tr);
        if (nextmode) {                           // the reset one call later: what this lane just computed is dropped
            const bool ended = (done || tr) && !pend;
            need = pend;
            r = pend ? 0.0 : r;
            done = pend ? false : done;
            tr = pend ? false : tr;
            pend = ended;
        }
        if (final_obs && !nextmode && __builtin_amdgcn_ballot_w64(need) != 0) {
            if (need) put_cells(r_fin, so, nc);
        }
        if constexpr (PH) {                       // reset(): drawn now from this step's feature-space stream
            if (__builtin_amdgcn_ballot_w64(need) != 0) {
                if (need) queue[0] = draw_cell();
                qn = need ? 1u : qn;
            }
        }
        // pop the next queued start cell where the episode ended
        nc = need ? queue[0] : nc;
#pragma unroll
        for (int q = 0; q + 1 < kGQ; q++) queue[q] = need ? queue[q + 1] : queue[q];
        qn -= need ? 1u : 0u;
        steps = need ? 0u : steps;
        flags = need ? 0u : flags;
        phase = need ? 0u : phase;
        cells = nc;
        put_cells(r_obs, so, nc);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)r), r_rew, v4, so * N * 4u, MDPP_ST_NT);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(done ? 1 : 0), r_term, v1, so * N, MDPP_ST_NT);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(tr ? 1 : 0), r_trunc, v1, so * N, MDPP_ST_NT);
    };
    const int nfull = K / kPre;
    for (int c = 0; c < nfull; c++) {
#pragma unroll
        for (int u = 0; u < kPre; u++) {
            const i32x4 act = pre[u];
            pre[u] = load_act(c * kPre + kPre + u);
            step(c * kPre + u, act);
        }
    }
    for (int k = nfull * kPre; k < K; k++) {
        i32x4 act = pre[0];
#pragma unroll
        for (int u = 1; u < kPre; u++) act = (k - nfull * kPre == u) ? pre[u] : act;
        step(k, act);
    }
    if constexpr (!PH) {
        // un-draw what was not used: s_prev = (s - inc) * M^-1 (mod 2^128), G uniforms per queued cell
        for (uint32_t q = qn * (uint32_t)G; q > 0; q--) {
            const uint64_t lo = sp.s_lo - sp.inc_lo;
            const uint64_t hi = sp.s_hi - sp.inc_hi - (sp.s_lo < sp.inc_lo ? 1ULL : 0ULL);
            sp.s_lo = lo * a.minv_lo;
            sp.s_hi = __umul64hi(lo, a.minv_lo) + lo * a.minv_hi + hi * a.minv_lo;
        }
        sp.store(a.sp_s, i);
        if (PN || RN) env.store(a.env_s, i);
        if (PN) {
            actg.store(a.act_s, i);
            a.act_half[i] = make_uint2(acth.has32, acth.u32);
        }
    }
    a.state[i] = make_uint4(cells, steps, flags | (pend ? 2u : 0u), 0u);
    if (status) atomicOr(&a.status[i], status);
}

int launch_grid_step(mdpp_env *h, int K, const int32_t *actions, void *obs, float *reward, uint8_t *term,
                     uint8_t *trunc, void *final_obs, hipStream_t s, char *name_out) {
    GridArgs a = h->gargs;
    a.opts = h->opts;
    a.ptick = h->tick;
    a.dtick = h->graph_capture ? (const uint64_t *)h->d_tick_off : nullptr;     // (launches being captured into a HIP graph)
    const int grid = (a.N + kBlock - 1) / kBlock;
    const bool noise = a.has_p_noise || a.has_r_noise;
    // quiet numpy-stream handles: the fused rollout kernel (< 4 GiB per output array per launch)
    if (!a.est.cur && !(a.philox && (a.opts & MDPP_OPT_NO_PHILOX_FAST)) &&
        !(noise && (a.opts & MDPP_OPT_NO_GFAST_NOISE)) && !(a.opts & MDPP_OPT_NO_GFAST) &&
        (unsigned long long)K * a.N * a.G * 8ULL < (1ULL << 32)) {
        const bool pn = a.has_p_noise != 0, rn = a.has_r_noise != 0;
        if (name_out) {
            snprintf(name_out, kNameLen, "k_grid_rollout_fast<OBS64=%d,G4=%d,DENSE=%d,PN=%d,RN=%d,PHILOX=%d>", !a.obs_i32, a.G == 4,
                     a.make_denser != 0, pn, rn, a.philox != 0);
            return MDPP_OK;
        }
#define MDPP_GF_LAUNCH(O64, G4, DN, PN_, RN_)                                                                                   \
    do {                                                                                                                        \
        if (a.philox) hipLaunchKernelGGL((k_grid_rollout_fast<O64, G4, DN, PN_, RN_, true>), dim3(grid), dim3(kBlock), 0, s, a, \
                                         K, actions, obs, reward, term, trunc, final_obs);                                      \
        else hipLaunchKernelGGL((k_grid_rollout_fast<O64, G4, DN, PN_, RN_, false>), dim3(grid), dim3(kBlock), 0, s, a, K,      \
                                actions, obs, reward, term, trunc, final_obs);                                                  \
    } while (0)
#define MDPP_GF_NZ(O64, G4, DN)                                                  \
    do {                                                                         \
        if (pn && rn) MDPP_GF_LAUNCH(O64, G4, DN, true, true);                   \
        else if (pn) MDPP_GF_LAUNCH(O64, G4, DN, true, false);                   \
        else if (rn) MDPP_GF_LAUNCH(O64, G4, DN, false, true);                   \
        else MDPP_GF_LAUNCH(O64, G4, DN, false, false);                          \
    } while (0)
#define MDPP_GF_DN(O64, G4) do { if (a.make_denser) MDPP_GF_NZ(O64, G4, true); else MDPP_GF_NZ(O64, G4, false); } while (0)
        if (a.obs_i32) { if (a.G == 4) MDPP_GF_DN(false, true); else MDPP_GF_DN(false, false); }
        else { if (a.G == 4) MDPP_GF_DN(true, true); else MDPP_GF_DN(true, false); }
#undef MDPP_GF_DN
#undef MDPP_GF_NZ
#undef MDPP_GF_LAUNCH
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { h->err = std::string("k_grid_rollout_fast launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
        h->tick += (uint64_t)K;
        return MDPP_OK;
    }
    if (name_out) {
        snprintf(name_out, kNameLen, "k_grid_step<PHILOX=%d,NOISE=%d>", a.philox != 0, noise);
        return MDPP_OK;
    }
#define MDPP_G_LAUNCH(PH, NZ) hipLaunchKernelGGL((k_grid_step<PH, NZ>), dim3(grid), dim3(kBlock), 0, s, a, K, actions, \
                                                 obs, reward, term, trunc, final_obs)
    if (a.philox) { if (noise) MDPP_G_LAUNCH(true, true); else MDPP_G_LAUNCH(true, false); }
    else { if (noise) MDPP_G_LAUNCH(false, true); else MDPP_G_LAUNCH(false, false); }
#undef MDPP_G_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("k_grid_step launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
    h->tick += (uint64_t)K;
    return MDPP_OK;
}

int launch_grid_reset(mdpp_env *h, const uint8_t *mask, void *obs, hipStream_t s) {
    GridArgs a = h->gargs;
    const int grid = (a.N + kBlock - 1) / kBlock;
    if (a.philox) hipLaunchKernelGGL(k_grid_reset<true>, dim3(grid), dim3(kBlock), 0, s, a, h->reset_tick, mask, obs);
    else hipLaunchKernelGGL(k_grid_reset<false>, dim3(grid), dim3(kBlock), 0, s, a, h->reset_tick, mask, obs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("k_grid_reset launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
    h->reset_tick += 1;
    return MDPP_OK;
}

} // namespace mdpp

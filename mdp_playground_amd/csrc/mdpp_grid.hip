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
#include "mdpp_internal.hpp"
#include "mdpp_rng.hpp"

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

    for (int k0 = 0; k0 < K; k0 += kGPrefetch) {
#pragma unroll
      for (int u = 0; u < kGPrefetch; u++) {
        const int k = k0 + u;
        if (k >= K) break;
        int act[4] = {pre[u][0], pre[u][1], pre[u][2], pre[u][3]};
        load_act(k + kGPrefetch, pre[u]);
        const uint32_t tick = a.tick + (uint32_t)k;
        const long o = (long)k * N + i;
        if (PHILOX) {
            env_phx.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), tick, MDPP_STREAM_ENV);
            sp_phx.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), tick, MDPP_STREAM_SPACE);
            act_phx.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), tick, kPhiloxActionStream);
            phx_h = Half32{0, 0};
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
        if (NOISE && a.has_r_noise)
            r += 0.0 + a.r_noise * (PHILOX ? np_standard_normal_lds(env_phx, zig) : np_standard_normal_lds(env_pcg, zig));
        r *= a.scale;
        r += a.shift;
        const bool done = (flags & 1u) != 0;                  // :2102-2104
        if (done) r += a.term_add;
        const bool truncated = (a.max_steps > 0) && (steps >= (uint32_t)a.max_steps);
        if (a.autoreset && (done || truncated)) {
            if (final_obs) put_obs(final_obs, o, cell);
            if (PHILOX) g_reset_lane(a, sp_phx, cell);
            else {
                if (!sp_loaded) { sp_pcg.load(a.sp_s, a.sp_inc, i); sp_loaded = true; }
                g_reset_lane(a, sp_pcg, cell);
            }
            steps = 0; flags = 0;
        }
        put_obs(obs, o, cell);
        reward[o] = (float)r;
        term[o] = done ? 1 : 0;
        trunc[o] = truncated ? 1 : 0;
      }
    }
    a.state[i] = make_uint4((uint32_t)cell[0] | ((uint32_t)cell[1] << 8) | ((uint32_t)cell[2] << 16) |
                                ((uint32_t)cell[3] << 24), steps, flags, 0u);
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
__global__ __launch_bounds__(kBlock) void k_grid_reset(GridArgs a, uint32_t reset_tick,
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
    a.state[i] = make_uint4((uint32_t)cell[0] | ((uint32_t)cell[1] << 8) | ((uint32_t)cell[2] << 16) |
                                ((uint32_t)cell[3] << 24), 0u, 0u, 0u);
    if (obs)
        for (int d = 0; d < a.G; d++) {
            if (a.obs_i32) ((int32_t *)obs)[i * a.G + d] = cell[d];
            else ((int64_t *)obs)[i * a.G + d] = (int64_t)cell[d];
        }
}

int launch_grid_step(mdpp_env *h, int K, const int32_t *actions, void *obs, float *reward, uint8_t *term,
                     uint8_t *trunc, void *final_obs, hipStream_t s) {
    GridArgs a = h->gargs;
    a.tick = h->tick;
    const int grid = (a.N + kBlock - 1) / kBlock;
    const bool noise = a.has_p_noise || a.has_r_noise;
#define MDPP_G_LAUNCH(PH, NZ) hipLaunchKernelGGL((k_grid_step<PH, NZ>), dim3(grid), dim3(kBlock), 0, s, a, K, actions, \
                                                 obs, reward, term, trunc, final_obs)
    if (a.philox) { if (noise) MDPP_G_LAUNCH(true, true); else MDPP_G_LAUNCH(true, false); }
    else { if (noise) MDPP_G_LAUNCH(false, true); else MDPP_G_LAUNCH(false, false); }
#undef MDPP_G_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("k_grid_step launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
    h->tick += (uint32_t)K;
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

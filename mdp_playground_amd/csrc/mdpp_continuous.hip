// Continuous move_to_a_point RLToyEnv.step()/reset() for gfx950: one lane per env instance.
//
// Restates /root/reference/mdp_playground/envs/rl_toy_env.py
//   C1 action admission (Box.contains)      :1630-1640, "stay" :1671-1679
//   C2 n-th order Taylor integrator          :1654-1669
//   C3 Gaussian transition noise             :1682-1691
//   C4 bounds test, clip, derivative reset   :1694-1717
//   C5 target-reached latch                  :1719-1725
//   C6 move_to_a_point reward                :1912-1945
//   C7 delay FIFO / every-n / noise / affine :1968-1990
//   C8 terminal hypercubes, done, term reward :945-952, :2102-2109
//   R2 reset                                 :2250, :2284-2323, :2358-2369
//
// Arithmetic follows numpy 2.x promotion exactly: float32 storage, the
// integrator term is float32*float32(t^k) promoted to float64 by the float64 factorial divisor,
// `+=` rounds back to float32; norms are float32 products accumulated in float64 (OpenBLAS sdot
// tail), rounded to float32 before the float32 sqrt.  This file must be compiled with
// -ffp-contract=off: a fused multiply-add anywhere below changes results.
//
// Data layout (HBM): per-env state is struct-of-arrays, sd[k][d][env] / cur[d][env] float32, so
// that consecutive lanes touch consecutive addresses; actions/observations are the caller's
// [env][D] row-major tensors and are read/written as float4 per lane when D % 4 == 0.
#include "mdpp_internal.hpp"
#include "mdpp_rng.hpp"

namespace mdpp {

struct CRew { double v; bool is32; }; // np.float32 vs Python float, as the reference's `reward`

template <int DMAX>
__device__ __forceinline__ float c_norm_rel(const ContinuousArgs &a, const float (&rel)[DMAX]) {
    // np.linalg.norm(rel - target): float32 products, float64 accumulate, one rounding, float32 sqrt
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < DMAX; j++) {
        if (j < a.n_rel) {
            float d = rel[j] - a.target[j];
            float p = d * d;
            acc += (double)p;
        }
    }
    return sqrtf((float)acc);
}

template <int DMAX>
__device__ __forceinline__ bool c_in_box(const ContinuousArgs &a, const float (&rel)[DMAX]) {
    bool any = false;
    for (int b = 0; b < a.n_boxes; b++) {
        bool in = true;
#pragma unroll
        for (int j = 0; j < DMAX; j++) {
            if (j < a.n_rel) {
                float x = rel[j];
                in = in && (x >= a.box_lo[b * a.n_rel + j]) && (x <= a.box_hi[b * a.n_rel + j]);
            }
        }
        any = any || in;
    }
    return any;
}

template <int DMAX>
__device__ __forceinline__ void c_gather_rel(const ContinuousArgs &a, const float (&s)[DMAX],
                                             float (&rel)[DMAX]) {
    // relevant_indices gather with compile-time register indices
    if (a.rel_prefix) { // relevant_indices == [0, 1, ..., n_rel-1]
#pragma unroll
        for (int j = 0; j < DMAX; j++) rel[j] = s[j];
        return;
    }
#pragma unroll
    for (int j = 0; j < DMAX; j++) {
        float v = 0.0f;
        if (j < a.n_rel) {
            const int idx = a.rel[j];
#pragma unroll
            for (int d = 0; d < DMAX; d++) v = (d == idx) ? s[d] : v;
        }
        rel[j] = v;
    }
}

template <int DMAX, int OMAX, class G>
__device__ __forceinline__ void c_reset_lane(const ContinuousArgs &a, G &sp, float (&sd)[OMAX + 1][DMAX],
                                             float (&cur)[DMAX], uint32_t &status) {
    float rel[DMAX];
    for (int tries = 0;; tries++) {
#pragma unroll
        for (int d = 0; d < DMAX; d++) {
            if (d < a.D) {
                double v = a.bounded ? a.reset_lo + a.reset_range * np_random(sp)
                                     : 0.0 + 1.0 * np_standard_normal(sp);
                cur[d] = (float)v;
            }
        }
        if (a.n_boxes == 0) break;
        c_gather_rel<DMAX>(a, cur, rel);
        if (!c_in_box<DMAX>(a, rel)) break;
        if (tries > 4096) { status |= 2u; break; }
    }
#pragma unroll
    for (int k = 0; k <= OMAX; k++)
#pragma unroll
        for (int d = 0; d < DMAX; d++) sd[k][d] = (k == 0) ? cur[d] : 0.0f;
}

template <int DMAX, int OMAX, bool PHILOX>
__global__ __launch_bounds__(kBlock) void k_continuous_step(ContinuousArgs a, int K,
                                                            const float *__restrict__ actions,
                                                            float *__restrict__ obs,
                                                            float *__restrict__ reward,
                                                            uint8_t *__restrict__ term,
                                                            uint8_t *__restrict__ trunc,
                                                            float *__restrict__ final_obs) {
    __shared__ uint64_t s_ki[256];
    __shared__ double s_wi[256], s_fi[256];
    const bool any_noise = a.has_p_noise || a.has_r_noise;   // wave-uniform
    if (any_noise) { zig_stage(s_ki, s_wi, s_fi, threadIdx.x, kBlock); __syncthreads(); }
    const ZigLds zig{s_ki, s_wi, s_fi};
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.N) return;
    const int D = a.D, n = a.order;
    const long N = a.N;

    float sd[OMAX + 1][DMAX], cur[DMAX], nxt[DMAX], act[DMAX], rel[DMAX];
#pragma unroll
    for (int k = 0; k <= OMAX; k++)
#pragma unroll
        for (int d = 0; d < DMAX; d++)
            sd[k][d] = (k <= n && d < D) ? a.sd[((long)k * D + d) * N + i] : 0.0f;
#pragma unroll
    for (int d = 0; d < DMAX; d++) cur[d] = (d < D) ? a.cur[(long)d * N + i] : 0.0f;
    uint2 meta = a.meta[i];
    uint32_t steps = meta.x, flags = meta.y, status = 0;

    Pcg64 env_pcg, sp_pcg;
    Philox env_phx, sp_phx;
    bool sp_loaded = false;
    const bool need_env = a.has_p_noise || a.has_r_noise;
    if (!PHILOX && need_env) env_pcg.load(a.env_s, a.env_inc, i);

    for (int k = 0; k < K; k++) {
        const uint32_t tick = a.tick + (uint32_t)k;
        const long o = (long)k * N + i;
        if (PHILOX) {
            env_phx.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), tick, MDPP_STREAM_ENV);
            sp_phx.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), tick, MDPP_STREAM_SPACE);
        }
        // ---- action: [env][D] row-major, 16 B per lane per load when possible
        const float *ap = actions + o * D;
        if (DMAX % 4 == 0 && D == DMAX) {
#pragma unroll
            for (int q = 0; q < DMAX / 4; q++) {
                float4 v = ((const float4 *)ap)[q];
                act[4 * q] = v.x; act[4 * q + 1] = v.y; act[4 * q + 2] = v.z; act[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int d = 0; d < DMAX; d++) act[d] = (d < D) ? ap[d] : 0.0f;
        }
        // ---- C1
        bool ok = true;
#pragma unroll
        for (int d = 0; d < DMAX; d++)
            if (d < D) ok = ok && (act[d] >= -a.amax32) && (act[d] <= a.amax32);
        if (ok) {
            // ---- C2: lower orders first, each using the not-yet-updated higher ones
#pragma unroll
            for (int d = 0; d < DMAX; d++) {
#pragma unroll
                for (int kk = 0; kk <= OMAX; kk++)
                    if (kk == n) sd[kk][d] = act[d] / a.inertia32;
            }
#pragma unroll
            for (int ii = 0; ii < OMAX; ii++) {
#pragma unroll
                for (int j = 0; j < OMAX; j++) {
                    if (ii < n && j < n - ii) {
#pragma unroll
                        for (int d = 0; d < DMAX; d++) {
                            float prod = sd[(ii + j + 1 <= OMAX) ? ii + j + 1 : OMAX][d] * a.tpow32[j + 1];
                            double trm = (double)prod / a.fact[j + 1];
                            sd[ii][d] = (float)((double)sd[ii][d] + trm);
                        }
                    }
                }
            }
#pragma unroll
            for (int d = 0; d < DMAX; d++) nxt[d] = sd[0][d];
        } else {
            status |= MDPP_STATUS_BAD_ACTION;
#pragma unroll
            for (int d = 0; d < DMAX; d++) nxt[d] = cur[d];
        }
        // ---- C3
#pragma unroll
        for (int d = 0; d < DMAX; d++) {
            if (d < D) {
                double nz = 0.0;
                if (a.has_p_noise)
                    nz = 0.0 + a.p_noise * (PHILOX ? np_standard_normal_lds(env_phx, zig) : np_standard_normal_lds(env_pcg, zig));
                nxt[d] = (float)((double)nxt[d] + nz);
            }
        }
        // ---- C4
        bool inside = true;
#pragma unroll
        for (int d = 0; d < DMAX; d++)
            if (d < D) inside = inside && (nxt[d] >= -a.smax32) && (nxt[d] <= a.smax32);
        // image observations: the reference asks the ImageContinuous space whether it contains the
        // state VECTOR, which it never does (spaces/image_continuous.py:292-302 returns None), so every
        // step takes this branch: a clip that is the identity inside the box, and zeroed derivatives
        if (a.image_quirk) inside = false;
        if (!inside) {
#pragma unroll
            for (int d = 0; d < DMAX; d++) {
                float x = nxt[d];
                if (x < -a.smax32) x = -a.smax32;
                if (x > a.smax32) x = a.smax32;
                nxt[d] = x;
            }
#pragma unroll
            for (int kk = 0; kk <= OMAX; kk++)
#pragma unroll
                for (int d = 0; d < DMAX; d++) sd[kk][d] = (kk == 0) ? nxt[d] : 0.0f;
        }
        // ---- C5
        c_gather_rel<DMAX>(a, nxt, rel);
        const float dist_new = c_norm_rel<DMAX>(a, rel);
        if (dist_new < a.radius32) flags |= 1u;
        const bool in_box = (a.n_boxes > 0) && c_in_box<DMAX>(a, rel);
        steps += 1;
        // ---- C6
        CRew r;
        if (a.make_denser) {
            float relo[DMAX];
            c_gather_rel<DMAX>(a, cur, relo);
            const float dist_old = c_norm_rel<DMAX>(a, relo);
            r.v = (double)(float)(-dist_new + dist_old);
        } else {
            r.v = (dist_new < a.radius32) ? 1.0 : 0.0;
        }
        {
            double acc = 0.0;
#pragma unroll
            for (int d = 0; d < DMAX; d++)
                if (d < D) { float p = act[d] * act[d]; acc += (double)p; }
            float pen = a.alw32 * sqrtf((float)acc);
            r.v = (double)((float)r.v - pen);
            r.is32 = true;
        }
        // ---- C7
        if (a.delay > 0) {
            uint32_t *slot = a.ring + (size_t)(tick % (uint32_t)a.delay) * N + i;
            uint32_t bits = *slot;
            *slot = __float_as_uint((float)r.v);
            if (bits == kRingPyZero) { r.v = 0.0; r.is32 = false; }
            else { r.v = (double)__uint_as_float(bits); r.is32 = true; }
        }
        if (steps % (uint32_t)a.every_n != 0) { r.v = 0.0; r.is32 = false; }
        if (a.has_r_noise) {
            double nz = 0.0 + a.r_noise * (PHILOX ? np_standard_normal_lds(env_phx, zig) : np_standard_normal_lds(env_pcg, zig));
            if (r.is32) r.v = (double)((float)r.v + (float)nz); else r.v = r.v + nz;
        }
        if (r.is32) {
            r.v = (double)((float)r.v * a.scale32);
            r.v = (double)((float)r.v + a.shift32);
        } else {
            r.v = r.v * a.scale;
            r.v = r.v + a.shift;
        }
        // ---- C8
        const bool done = in_box || (flags & 1u);
        if (done) {
            if (r.is32) r.v = (double)((float)r.v + a.term_add32); else r.v = r.v + a.term_add;
        }
#pragma unroll
        for (int d = 0; d < DMAX; d++) cur[d] = nxt[d];
        const bool truncated = (a.max_steps > 0) && (steps >= (uint32_t)a.max_steps);

        if (a.autoreset && (done || truncated)) {
            if (final_obs) {
#pragma unroll
                for (int d = 0; d < DMAX; d++) if (d < D) final_obs[o * D + d] = nxt[d];
            }
            if (PHILOX) {
                c_reset_lane<DMAX, OMAX>(a, sp_phx, sd, cur, status);
            } else {
                if (!sp_loaded) { sp_pcg.load(a.sp_s, a.sp_inc, i); sp_loaded = true; }
                c_reset_lane<DMAX, OMAX>(a, sp_pcg, sd, cur, status);
            }
            steps = 0; flags = 0;
            for (int dd = 0; dd < a.delay; dd++) a.ring[(size_t)dd * N + i] = kRingPyZero;
        }
        // ---- outputs
        float *op = obs + o * D;
        if (DMAX % 4 == 0 && D == DMAX) {
#pragma unroll
            for (int q = 0; q < DMAX / 4; q++)
                ((float4 *)op)[q] = make_float4(cur[4 * q], cur[4 * q + 1], cur[4 * q + 2], cur[4 * q + 3]);
        } else {
#pragma unroll
            for (int d = 0; d < DMAX; d++) if (d < D) op[d] = cur[d];
        }
        reward[o] = (float)r.v;
        term[o] = done ? 1 : 0;
        trunc[o] = truncated ? 1 : 0;
    }

#pragma unroll
    for (int k = 0; k <= OMAX; k++)
#pragma unroll
        for (int d = 0; d < DMAX; d++)
            if (k <= n && d < D) a.sd[((long)k * D + d) * N + i] = sd[k][d];
#pragma unroll
    for (int d = 0; d < DMAX; d++) if (d < D) a.cur[(long)d * N + i] = cur[d];
    a.meta[i] = make_uint2(steps, flags);
    if (!PHILOX) {
        if (need_env) env_pcg.store(a.env_s, i);
        if (sp_loaded) sp_pcg.store(a.sp_s, i);
    }
    if (status) atomicOr(&a.status[i], status);
}

template <int DMAX, int OMAX, bool PHILOX>
__global__ __launch_bounds__(kBlock) void k_continuous_reset(ContinuousArgs a, uint32_t reset_tick,
                                                             const uint8_t *__restrict__ mask,
                                                             float *__restrict__ obs) {
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.N) return;
    if (mask && !mask[i]) return;
    const int D = a.D, n = a.order;
    const long N = a.N;
    float sd[OMAX + 1][DMAX], cur[DMAX];
    uint32_t status = 0;
    if (PHILOX) {
        Philox g;
        g.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), reset_tick, kPhiloxResetStream);
        c_reset_lane<DMAX, OMAX>(a, g, sd, cur, status);
    } else {
        Pcg64 g;
        g.load(a.sp_s, a.sp_inc, i);
        c_reset_lane<DMAX, OMAX>(a, g, sd, cur, status);
        g.store(a.sp_s, i);
    }
#pragma unroll
    for (int k = 0; k <= OMAX; k++)
#pragma unroll
        for (int d = 0; d < DMAX; d++)
            if (k <= n && d < D) a.sd[((long)k * D + d) * N + i] = sd[k][d];
#pragma unroll
    for (int d = 0; d < DMAX; d++) {
        if (d < D) {
            a.cur[(long)d * N + i] = cur[d];
            if (obs) obs[i * D + d] = cur[d];
        }
    }
    a.meta[i] = make_uint2(0u, 0u);
    for (int dd = 0; dd < a.delay; dd++) a.ring[(size_t)dd * N + i] = kRingPyZero;
    if (status) atomicOr(&a.status[i], status);
}

// ---- dispatch on (padded D, padded order) -------------------------------------------------
template <int DMAX, int OMAX>
static void launch_step_t(const ContinuousArgs &a, int K, const float *actions, float *obs,
                          float *reward, uint8_t *term, uint8_t *trunc, float *final_obs,
                          hipStream_t s) {
    const int grid = (a.N + kBlock - 1) / kBlock;
    if (a.philox)
        hipLaunchKernelGGL((k_continuous_step<DMAX, OMAX, true>), dim3(grid), dim3(kBlock), 0, s, a,
                           K, actions, obs, reward, term, trunc, final_obs);
    else
        hipLaunchKernelGGL((k_continuous_step<DMAX, OMAX, false>), dim3(grid), dim3(kBlock), 0, s, a,
                           K, actions, obs, reward, term, trunc, final_obs);
}
template <int DMAX, int OMAX>
static void launch_reset_t(const ContinuousArgs &a, uint32_t reset_tick, const uint8_t *mask,
                           float *obs, hipStream_t s) {
    const int grid = (a.N + kBlock - 1) / kBlock;
    if (a.philox)
        hipLaunchKernelGGL((k_continuous_reset<DMAX, OMAX, true>), dim3(grid), dim3(kBlock), 0, s, a,
                           reset_tick, mask, obs);
    else
        hipLaunchKernelGGL((k_continuous_reset<DMAX, OMAX, false>), dim3(grid), dim3(kBlock), 0, s, a,
                           reset_tick, mask, obs);
}

#define MDPP_C_DISPATCH(CALL)                                                       \
    do {                                                                            \
        const int D_ = a.D, O_ = a.order;                                           \
        if (D_ <= 2 && O_ <= 2) { CALL(2, 2); }                                     \
        else if (D_ <= 4 && O_ <= 2) { CALL(4, 2); }                                \
        else if (D_ <= 4) { CALL(4, 4); }                                           \
        else if (D_ <= 12 && O_ <= 1) { CALL(12, 1); }                              \
        else if (D_ <= 12 && O_ <= 2) { CALL(12, 2); }                              \
        else if (D_ <= 12) { CALL(12, 4); }                                         \
        else if (D_ <= 16 && O_ <= 2) { CALL(16, 2); }                              \
        else if (D_ <= 32 && O_ <= 2) { CALL(32, 2); }                              \
        else { return MDPP_EUNSUPPORTED; }                                          \
    } while (0)

int launch_continuous_step(mdpp_env *h, int K, const float *actions, float *obs, float *reward,
                           uint8_t *term, uint8_t *trunc, float *final_obs, hipStream_t s) {
    ContinuousArgs a = h->cargs;
    a.tick = h->tick;
    if (a.fast_ok) {
        // common shape: dedicated rollout kernel (mdpp_continuous_fast.hip); its buffer descriptors
        // address < 4 GiB per array, so long rollouts go out as several launches
        const long long kmax = ((1LL << 32) - 1) / ((long long)a.N * a.D * 4);
        bool served = kmax >= 1;
        for (int k0 = 0; served && k0 < K;) {
            const int kc = (int)((K - k0) < kmax ? (K - k0) : kmax);
            const size_t off = (size_t)k0 * a.N;
            served = launch_continuous_fast(a, kc, actions + off * a.D, obs + off * a.D, reward + off,
                                            term + off, trunc + off,
                                            final_obs ? final_obs + off * a.D : nullptr, s);
            if (!served && k0 > 0) { h->err = "k_continuous_rollout_fast: inconsistent dispatch"; return MDPP_EHIP; }
            k0 += kc;
        }
        if (served) {
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { h->err = std::string("k_continuous_rollout_fast launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
            h->tick += (uint32_t)K;
            return MDPP_OK;
        }
    }
#define CALL_STEP(DM, OM) launch_step_t<DM, OM>(a, K, actions, obs, reward, term, trunc, final_obs, s)
    MDPP_C_DISPATCH(CALL_STEP);
#undef CALL_STEP
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("k_continuous_step launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
    h->tick += (uint32_t)K;
    return MDPP_OK;
}

int launch_continuous_reset(mdpp_env *h, const uint8_t *mask, float *obs, hipStream_t s) {
    ContinuousArgs a = h->cargs;
#define CALL_RESET(DM, OM) launch_reset_t<DM, OM>(a, h->reset_tick, mask, obs, s)
    MDPP_C_DISPATCH(CALL_RESET);
#undef CALL_RESET
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("k_continuous_reset launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
    h->reset_tick += 1;
    return MDPP_OK;
}

} // namespace mdpp

// Fused K-step rollout for reward_function="move_along_a_line" in its common shape (the reference's own test env,
// tests/test_mdp_playground.py:31-71): every dimension relevant (2 or 4 of them), dynamics order 1 or 2, no noise,
// delay 0, every step pays, bounded box, no terminal hypercubes, sequence_length <= 16.  Same arithmetic for the STATE as
// k_continuous_step / k_continuous_rollout_fast (rl_toy_env.py:1630-1717: bit-identical, tested); the REWARD
// (:1864-1910, dist_of_pt_from_line :2546-2576) is the same line fit as c_line_reward in mdpp_continuous.hip -- float32
// mean in numpy's pairwise order, dominant eigenvector of the float64 scatter matrix by repeated squaring, rounded to
// float32, float64 distances -- organised for one wave per SIMD, where k_continuous_step spends 2 584 vector instructions
// per step (profiles/archive/r03_sq_line_and_noise.txt):
//   * the L points of every lane live in LDS for the launch (float4 [slot][lane]; HBM is written through, so that
//     k_continuous_step / mdpp_step can take over at any time);
//   * the raw moments sum x, sum x x^T of the window are carried in registers and updated by the point that enters and the
//     point that leaves (2 points per step instead of L); they are recomputed from the window at every launch, which bounds
//     the drift of the running sums to one launch (relative 1e-16 per update, in a reward defined to ~1e-7 by its float32
//     singular vector);
//   * one float64 square root per point (v_rsq_f64 + two Newton steps, no denormal / overflow scaling: the arguments are
//     squared distances of float32 points, zero or far from both ends of the exponent range) and no division;
//   * D, order and the relevant set are template constants; nothing of the other reward functions, noise or the delay line
//     is in the loop.
// Rewards therefore agree with k_continuous_step to a few float64 ulps before the float32 output rounding, not bit for bit
// (tests: equal within 1e-6, states and flags bit-exact; against the oracle within the upstream tolerance like every
// line-reward test).
#include <stdio.h>

#include <type_traits>

#include "mdpp_internal.hpp"
#include "mdpp_rng.hpp"

namespace mdpp {

namespace {
constexpr int kLineMaxL = 16;
constexpr int kLRsrc = 0x00020000;
typedef unsigned int lu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int lu32x2 __attribute__((ext_vector_type(2)));

// sqrt of a float64 that is +0 or a normal number well inside the exponent range: reciprocal square root estimate and
// two coupled Newton steps (Goldschmidt), final residual correction; within an ulp of the correctly rounded root
__device__ __forceinline__ double line_sqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    const double d = fma(-g, g, x);
    g = fma(d, h, g);
    return x == 0.0 ? 0.0 : g;
}
} // namespace

template <int D, int ORDER, bool PHILOX>
__global__ __launch_bounds__(kBlock) void k_continuous_line_rollout(ContinuousArgs a, int K,
                                                                    const float *__restrict__ actions,
                                                                    float *__restrict__ obs,
                                                                    float *__restrict__ reward,
                                                                    uint8_t *__restrict__ term,
                                                                    uint8_t *__restrict__ trunc,
                                                                    float *__restrict__ final_obs) {
    const uint64_t ptick0 = tick_now(a);               // the step counter at this launch (through the device-side offset of a graph replay)
    static_assert(D == 2 || D == 4, "every dimension relevant: 2 or 4");
    extern __shared__ __align__(16) float4 s_pts[];            // [L][kBlock]
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= (uint32_t)a.N) return;
    const uint32_t N = (uint32_t)a.N;
    const int L = a.line_L;
    const int tid = threadIdx.x;

    float sd[ORDER + 1][D], cur[D];
#pragma unroll
    for (int k = 0; k <= ORDER; k++)
#pragma unroll
        for (int d = 0; d < D; d++) sd[k][d] = a.sd[((size_t)k * D + d) * N + i];
#pragma unroll
    for (int d = 0; d < D; d++) cur[d] = a.cur[(size_t)d * N + i];
    const uint2 meta = a.meta[i];
    uint32_t steps = meta.x, status = 0;

    // ---- the window: HBM -> LDS, and its raw moments
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, mm[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // sum x; sum x x^T (upper triangle, row-major)
    auto mom = [&](const float4 p, double sign) __attribute__((always_inline)) {
        const double x[4] = {(double)p.x, (double)p.y, (double)p.z, (double)p.w};
        int q = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            s1[r] += sign * x[r];
#pragma unroll
            for (int c = r; c < 4; c++) { mm[q] = fma(sign * x[r], x[c], mm[q]); q++; }
        }
    };
    uint32_t cnt = steps + 1u < (uint32_t)L ? steps + 1u : (uint32_t)L;                  // valid points in the window
    uint32_t slot = steps % (uint32_t)L;                                                 // slot of the newest point
    for (int sl = 0; sl < L; sl++) {
        float q[4];
#pragma unroll
        for (int j = 0; j < 4; j++) q[j] = (j < D) ? a.line_hist[((size_t)sl * 4 + j) * N + i] : 0.0f;
        const float4 p = make_float4(q[0], q[1], q[2], q[3]);
        s_pts[sl * kBlock + tid] = p;
        // slot sl holds the state after s' transitions with s' = steps - ((slot - sl) mod L): valid iff that is >= 0
        const uint32_t back = (slot + (uint32_t)L - (uint32_t)sl) % (uint32_t)L;
        if (back < cnt) mom(p, 1.0);
    }

    const uint32_t total = (uint32_t)K * N;
    auto r_act = __builtin_amdgcn_make_buffer_rsrc((void *)actions, 0, total * (uint32_t)(D * 4), kLRsrc);
    auto r_obs = __builtin_amdgcn_make_buffer_rsrc((void *)obs, 0, total * (uint32_t)(D * 4), kLRsrc);
    auto r_rew = __builtin_amdgcn_make_buffer_rsrc((void *)reward, 0, total * 4u, kLRsrc);
    auto r_term = __builtin_amdgcn_make_buffer_rsrc((void *)term, 0, total, kLRsrc);
    auto r_trunc = __builtin_amdgcn_make_buffer_rsrc((void *)trunc, 0, total, kLRsrc);
    const uint32_t vrow = i * (uint32_t)(D * 4), row_bytes = N * (uint32_t)(D * 4);
    const float amax = a.amax32, smax = a.smax32, inv_inertia = a.inv_inertia32;
    const bool inertia_pow2 = a.inertia_pow2 != 0, has_max = a.max_steps > 0, autoreset = a.autoreset != 0;
    const uint32_t max_steps = (uint32_t)a.max_steps;

    auto all_within = [&](const float (&v)[D], float bound) -> bool {
        uint32_t m = 0;
#pragma unroll
        for (int d = 0; d < D; d++) m = max(m, __float_as_uint(v[d]) & 0x7FFFFFFFu);
        return m <= __float_as_uint(bound);
    };
    auto load_row = [&](int k, float (&dst)[D]) {
        const uint32_t kk = (uint32_t)min(k, K - 1);
        if (D == 2) {
            const lu32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r_act, vrow, kk * row_bytes, 0);
            dst[0] = __uint_as_float(v.x); dst[1] = __uint_as_float(v.y);
        } else {
            const lu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r_act, vrow, kk * row_bytes, 0);
            dst[0] = __uint_as_float(v.x); dst[1] = __uint_as_float(v.y);
            dst[2 % D] = __uint_as_float(v.z); dst[3 % D] = __uint_as_float(v.w);
        }
    };
    auto put_point = [&](uint32_t sl, const float (&s)[D]) __attribute__((always_inline)) -> float4 {
        float q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < D; j++) { q[j] = s[j]; a.line_hist[((size_t)sl * 4 + j) * N + i] = s[j]; }
        const float4 p = make_float4(q[0], q[1], q[2], q[3]);
        s_pts[sl * kBlock + tid] = p;
        return p;
    };

    // the line fit over the L points of the window (all valid: steps >= L)
    auto fit = [&]() __attribute__((always_inline)) -> double {
        float px[kLineMaxL][4];
        uint32_t sl = (slot + 1u == (uint32_t)L) ? 0u : slot + 1u;                   // the oldest point
#pragma unroll
        for (int k = 0; k < kLineMaxL; k++) {
            const float4 q = s_pts[sl * kBlock + tid];
            px[k][0] = q.x; px[k][1] = q.y; px[k][2] = q.z; px[k][3] = q.w;
            const uint32_t nx = (sl + 1u == (uint32_t)L) ? 0u : sl + 1u;
            sl = (k + 1 < L) ? nx : sl;
        }
        // data_.mean(axis=0) as numpy sums a float32 column: pairwise routine (8 running sums while 8 more points are
        // left, a fixed tree, the rest one by one; plain left to right below 8 points), divided by L in float32
        float mean[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        {
            float r[8][4];
#pragma unroll
            for (int q = 0; q < 8; q++)
#pragma unroll
                for (int j = 0; j < 4; j++) r[q][j] = (L == kLineMaxL) ? px[q][j] + px[8 + q][j] : px[q][j];
            const int tail0 = (L < 8) ? 0 : (L == kLineMaxL ? kLineMaxL : 8);
#pragma unroll
            for (int j = 0; j < 4; j++)
                mean[j] = (L < 8) ? 0.0f
                                  : ((r[0][j] + r[1][j]) + (r[2][j] + r[3][j])) + ((r[4][j] + r[5][j]) + (r[6][j] + r[7][j]));
#pragma unroll
            for (int k = 0; k < kLineMaxL; k++)
                if (k < L && k >= tail0) {
#pragma unroll
                    for (int j = 0; j < 4; j++) mean[j] += px[k][j];
                }
#pragma unroll
            for (int j = 0; j < 4; j++) mean[j] = (j < D) ? mean[j] / (float)L : 0.0f;
        }
        // scatter matrix about that mean from the carried raw moments
        double m[4][4];
        {
            int q = 0;
#pragma unroll
            for (int p = 0; p < 4; p++)
#pragma unroll
                for (int c = p; c < 4; c++) {
                    const double mp = (double)mean[p], mq = (double)mean[c];
                    m[p][c] = mm[q] - mp * s1[c] - s1[p] * mq + (double)L * mp * mq;
                    m[c][p] = m[p][c];
                    q++;
                }
        }
        double v[4] = {1.0, 0.0, 0.0, 0.0};
        double tr = m[0][0] + m[1][1] + m[2][2] + m[3][3];
        if (tr > 0.0) {
            auto pow2_inv = [](double t) __attribute__((always_inline)) -> double {
                const uint64_t e = ((uint64_t)__double_as_longlong(t) >> 52) & 0x7FFu;
                return __longlong_as_double((long long)((2046ull - e - 1ull) << 52));
            };
            double sc = pow2_inv(tr);
#pragma unroll
            for (int p = 0; p < 4; p++)
#pragma unroll
                for (int q = 0; q < 4; q++) m[p][q] *= sc;
            tr *= sc;
            for (int it = 0; it < 40; it++) {
                double sq[4][4];
#pragma unroll
                for (int p = 0; p < 4; p++)
#pragma unroll
                    for (int q = p; q < 4; q++) {
                        double acc = 0.0;
#pragma unroll
                        for (int r = 0; r < 4; r++) acc = fma(m[p][r], m[r][q], acc);
                        sq[p][q] = acc;
                    }
                const double tr2 = sq[0][0] + sq[1][1] + sq[2][2] + sq[3][3];
                const bool conv = (tr * tr - tr2) <= 2e-12 * tr * tr;
                sc = pow2_inv(tr2);
#pragma unroll
                for (int p = 0; p < 4; p++)
#pragma unroll
                    for (int q = p; q < 4; q++) { m[p][q] = sq[p][q] * sc; m[q][p] = m[p][q]; }
                tr = tr2 * sc;
                if (__builtin_amdgcn_ballot_w64(!conv) == 0) break;
            }
            double best = m[0][0];
            double c0 = m[0][0], c1 = m[1][0], c2 = m[2][0], c3 = m[3][0];
#pragma unroll
            for (int q = 1; q < 4; q++) {
                const bool b = m[q][q] > best;
                best = b ? m[q][q] : best;
                c0 = b ? m[0][q] : c0; c1 = b ? m[1][q] : c1; c2 = b ? m[2][q] : c2; c3 = b ? m[3][q] : c3;
            }
            const double nn = c0 * c0 + c1 * c1 + c2 * c2 + c3 * c3;
            const double s = 1.0 / line_sqrt(nn);
            v[0] = c0 * s; v[1] = c1 * s; v[2] = c2 * s; v[3] = c3 * s;
        }
        double ptA[4], ab[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const double vj = (double)(float)v[j], mj = (double)mean[j];
            ptA[j] = vj * -1.0 + mj;
            ab[j] = ptA[j] - (vj * 1.0 + mj);
        }
        double nab = 0.0;
#pragma unroll
        for (int j = 0; j < D; j++) nab = fma(ab[j], ab[j], nab);
        const bool degenerate = nab < 1e-26;                  // |ab| < 1e-13 (:2560)
        const double inv_nab2 = 1.0 / nab;
        double totald = 0.0;
#pragma unroll
        for (int k = 0; k < kLineMaxL; k++)
            if (k < L) {
                double dot = 0.0, nap = 0.0;
#pragma unroll
                for (int j = 0; j < D; j++) {
                    const double ap = ptA[j] - (double)px[k][j];
                    dot = fma(ab[j], ap, dot);
                    nap = fma(ap, ap, nap);
                }
                double sq = nap - (dot * dot) * inv_nab2;
                sq = sq < 0.0 ? 0.0 : sq;
                totald += degenerate ? 0.0 : line_sqrt(sq);
            }
        return 0.0 + -totald / (double)L;
    };

    constexpr int kAhead = 2;         // (a step is ~2 000 instructions: two copies of it stay inside the instruction cache)
    float pre[kAhead][D];
#pragma unroll
    for (int u = 0; u < kAhead; u++) load_row(u, pre[u]);

    auto step = [&](const float (&act)[D], int k) __attribute__((always_inline)) {
        const uint32_t so = (uint32_t)k * N;
        float nxt[D];
        const bool ok = all_within(act, amax);                   // C1
        float nacc[D];
        if (inertia_pow2) {
#pragma unroll
            for (int d = 0; d < D; d++) nacc[d] = act[d] * inv_inertia;
        } else {
#pragma unroll
            for (int d = 0; d < D; d++) nacc[d] = act[d] / a.inertia32;
        }
#pragma unroll
        for (int ii = 0; ii < ORDER; ii++) {                     // C2 (as k_continuous_rollout_fast)
#pragma unroll
            for (int d = 0; d < D; d++) {
                float acc = sd[ii][d];
#pragma unroll
                for (int j = 0; j < ORDER; j++) {
                    if (j >= ORDER - ii) continue;
                    const float hi = (ii + j + 1 == ORDER) ? nacc[d] : sd[(ii + j + 1 < ORDER) ? ii + j + 1 : ORDER][d];
                    const float prod = hi * a.tpow32[j + 1];
                    acc = (j + 1 == 2) ? fmaf(prod, 0.5f, acc) : acc + prod;
                }
                sd[ii][d] = ok ? acc : sd[ii][d];
            }
        }
#pragma unroll
        for (int d = 0; d < D; d++) sd[ORDER][d] = ok ? nacc[d] : sd[ORDER][d];
#pragma unroll
        for (int d = 0; d < D; d++) nxt[d] = ok ? sd[0][d] : cur[d];
        status |= ok ? 0u : (uint32_t)MDPP_STATUS_BAD_ACTION;
#pragma unroll
        for (int d = 0; d < D; d++) nxt[d] = nxt[d] + 0.0f;                            // C3 without noise: -0 -> +0
        const bool inside = all_within(nxt, smax);                                     // C4
        if (__builtin_amdgcn_ballot_w64(!inside) != 0) {
#pragma unroll
            for (int d = 0; d < D; d++) nxt[d] = __builtin_amdgcn_fmed3f(nxt[d], -smax, smax);
#pragma unroll
            for (int kk = 0; kk <= ORDER; kk++)
#pragma unroll
                for (int d = 0; d < D; d++) sd[kk][d] = inside ? sd[kk][d] : (kk == 0 ? nxt[d] : 0.0f);
        }
        steps += 1;
        // ---- the window: the new state enters at slot steps % L; what sat there (L transitions ago) leaves
        slot = (slot + 1u == (uint32_t)L) ? 0u : slot + 1u;
        const float4 old = s_pts[slot * kBlock + tid];
        const float4 pnew = put_point(slot, nxt);
        if (cnt == (uint32_t)L) mom(old, -1.0); else cnt += 1u;
        mom(pnew, 1.0);
        // ---- reward (:1856 gate: sequence_length transitions must exist), float64 throughout (:1980-1990)
        double r = 0.0;
        if (__builtin_amdgcn_ballot_w64(steps >= (uint32_t)L) != 0) {
            const double f = fit();
            r = steps >= (uint32_t)L ? f : 0.0;
        }
        r = r * a.scale;
        r = r + a.shift;
        const bool tr = has_max && steps >= max_steps;
#pragma unroll
        for (int d = 0; d < D; d++) cur[d] = nxt[d];
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(autoreset && tr) != 0, 0)) {   // same-step autoreset (truncation only:
            if (autoreset && tr) {                                                       //  nothing else ends such an episode)
                if (final_obs) {
#pragma unroll
                    for (int d = 0; d < D; d++) final_obs[((size_t)so + i) * D + d] = nxt[d];
                }
                typename std::conditional<PHILOX, Philox, Pcg64>::type sp;
                if constexpr (PHILOX) sp.init(a.philox_seed, (uint64_t)(a.env_id_offset + (int64_t)i), ptick0 + (uint64_t)k, MDPP_STREAM_SPACE);
                else sp.load(a.sp_s, a.sp_inc, i);
#pragma unroll
                for (int d = 0; d < D; d++) cur[d] = (float)(a.reset_lo + a.reset_range * np_random(sp));
                if constexpr (!PHILOX) sp.store(a.sp_s, i);
#pragma unroll
                for (int kk = 0; kk <= ORDER; kk++)
#pragma unroll
                    for (int d = 0; d < D; d++) sd[kk][d] = (kk == 0) ? cur[d] : 0.0f;
                steps = 0; slot = 0; cnt = 1;
#pragma unroll
                for (int q = 0; q < 4; q++) s1[q] = 0.0;
#pragma unroll
                for (int q = 0; q < 10; q++) mm[q] = 0.0;
                mom(put_point(0u, cur), 1.0);                                            // augmented_state = [..., curr_state], :2313-2323
            }
        }
        if (D == 2)
            __builtin_amdgcn_raw_buffer_store_b64(lu32x2{__float_as_uint(cur[0]), __float_as_uint(cur[1])}, r_obs, vrow,
                                                  so * (uint32_t)(D * 4), MDPP_ST_NT);
        else   // (128-bit store: the whole offset in the VGPR, see mdpp_discrete_quiet.hip on the store-data hazard)
            __builtin_amdgcn_raw_buffer_store_b128(lu32x4{__float_as_uint(cur[0]), __float_as_uint(cur[1]), __float_as_uint(cur[2 % D]),
                                                          __float_as_uint(cur[3 % D])},
                                                   r_obs, vrow + so * (uint32_t)(D * 4), 0, MDPP_ST_NT);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)r), r_rew, i * 4u, so * 4u, MDPP_ST_NT);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0, r_term, i, so, MDPP_ST_NT);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(tr ? 1 : 0), r_trunc, i, so, MDPP_ST_NT);
    };

    const int nfull = K / kAhead;
    for (int c = 0; c < nfull; c++) {
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
            float act[D];
#pragma unroll
            for (int d = 0; d < D; d++) act[d] = pre[u][d];
            load_row(c * kAhead + kAhead + u, pre[u]);
            step(act, c * kAhead + u);
        }
    }
    for (int k = nfull * kAhead; k < K; k++) {
        float act[D];
        const int u = k - nfull * kAhead;
#pragma unroll
        for (int uu = 0; uu < kAhead; uu++)
            if (uu == u) {
#pragma unroll
                for (int d = 0; d < D; d++) act[d] = pre[uu][d];
            }
        step(act, k);
    }

#pragma unroll
    for (int k = 0; k <= ORDER; k++)
#pragma unroll
        for (int d = 0; d < D; d++) a.sd[((size_t)k * D + d) * N + i] = sd[k][d];
#pragma unroll
    for (int d = 0; d < D; d++) a.cur[(size_t)d * N + i] = cur[d];
    a.meta[i] = make_uint2(steps, 0u);
    if (status) atomicOr(&a.status[i], status);
}

// Returns false when the shape does not qualify (the caller goes on to k_continuous_step).
bool launch_continuous_line(const ContinuousArgs &a, int K, const float *actions, float *obs, float *reward, uint8_t *term,
                            uint8_t *trunc, float *final_obs, hipStream_t s, char *name_out) {
    if (!a.line_L || a.line_L < 2 || a.line_L > kLineMaxL || (a.opts & MDPP_OPT_NO_CFAST)) return false;
    if (a.philox && (a.opts & MDPP_OPT_NO_PHILOX_FAST)) return false;
    if (a.n_rel != a.D || !a.rel_prefix || (a.D != 2 && a.D != 4) || a.order > 2) return false;
    if (a.has_p_noise || a.has_r_noise || a.delay != 0 || a.every_n != 1 || a.n_boxes != 0 || !a.bounded || a.image_quirk) return false;
    if (a.autoreset == MDPP_AUTORESET_NEXT_STEP || a.est.cur || K < 4) return false;
    if ((unsigned long long)K * a.N * a.D * 4ULL >= (1ULL << 32)) return false;
    if (name_out) {
        snprintf(name_out, kNameLen, "k_continuous_line_rollout<D=%d,ORDER=%d,PHILOX=%d>", a.D, a.order, a.philox != 0);
        return true;
    }
    const int grid = (a.N + kBlock - 1) / kBlock;
    const size_t lds = (size_t)a.line_L * kBlock * sizeof(float4);
#define MDPP_LINE_GO(DD, OO, PH)                                                                                              \
    do {                                                                                                                      \
        if (!dynamic_lds_ok((const void *)k_continuous_line_rollout<DD, OO, PH>, lds)) return false;   /* -> k_continuous_step */ \
        hipLaunchKernelGGL((k_continuous_line_rollout<DD, OO, PH>), dim3(grid), dim3(kBlock), lds, s, a, K, actions, obs, reward, \
                           term, trunc, final_obs);                                                                           \
    } while (0)
#define MDPP_LINE_PH(DD, OO) do { if (a.philox) MDPP_LINE_GO(DD, OO, true); else MDPP_LINE_GO(DD, OO, false); } while (0)
    if (a.D == 4 && a.order == 1) MDPP_LINE_PH(4, 1);
    else if (a.D == 4) MDPP_LINE_PH(4, 2);
    else if (a.order == 1) MDPP_LINE_PH(2, 1);
    else MDPP_LINE_PH(2, 2);
#undef MDPP_LINE_PH
#undef MDPP_LINE_GO
    return true;
}

} // namespace mdpp

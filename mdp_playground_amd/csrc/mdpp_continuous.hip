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
#ifndef MDPP_CONT_TU_LINE8
#define MDPP_CONT_TU_LINE8 0       // 1: this translation unit holds k_continuous_step<..., NL = 8> (mdpp_continuous_line8.hip)
#endif

#include "mdpp_internal.hpp"
#include "mdpp_rng.hpp"

namespace mdpp {

struct CRew { double v; bool is32; }; // np.float32 vs Python float, as the reference's `reward`

template <int DMAX>
__device__ __forceinline__ float c_norm_rel(const ContinuousArgs &a, const float (&rel)[DMAX], const float (&target)[DMAX]) {
    // np.linalg.norm(rel - target): float32 products, float64 accumulate, one rounding, float32 sqrt
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < DMAX; j++) {
        if (j < a.n_rel) {
            float d = rel[j] - target[j];
            float p = d * d;
            acc += (double)p;
        }
    }
    return sqrtf((float)acc);
}

// The default target (float64 zeros over every dimension, :652-654): np.linalg.norm(float32 state - float64 zeros) is
// sqrt(x.dot(x)) in float64 -- cblas_ddot, whose scalar tail sums sequentially below 16 elements; every product of two
// float32-valued doubles is exact, so no fused multiply-add can change a bit (n_rel == D here).
template <int DMAX>
__device__ __forceinline__ double c_norm64(const ContinuousArgs &a, const float (&s)[DMAX]) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < DMAX; j++) {
        if (j < a.n_rel) {
            const double d = (double)s[j] - 0.0;
            acc += d * d;
        }
    }
    return sqrt(acc);
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

// ---- reward_function move_along_a_line (:1864-1910, dist_of_pt_from_line :2546-2576) -------------
// The last L states' relevant coordinates live in HBM, line_hist[(slot * 4 + j) * N + env] with
// slot = s % L for the state reached after s transitions of the episode (s = 0: reset()).
// `lds_line` (rollouts with L <= 16, round 3): the lane's L points mirrored in LDS for the launch, float4 [slot][lane];
// the HBM copy stays the truth between launches (write-through), the per-step fit reads LDS instead of making 64 L2
// round trips per step.
// Rows of line_hist are 4 floats wide (a.line_NL = 4: at most 4 relevant dimensions, the LDS mirror is a float4) or 8
// (5 to 8 relevant dimensions: k_continuous_step<..., NL = 8>, mdpp_continuous_line8.hip).
template <int DMAX>
__device__ __forceinline__ void c_line_put(const ContinuousArgs &a, long i, uint32_t s, const float (&rel)[DMAX],
                                           float4 *lds_line = nullptr) {
    const uint32_t slot = s % (uint32_t)a.line_L;
    const size_t NL = (size_t)a.line_NL;
    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < DMAX; j++)
        if (j < a.n_rel) { if (j < 4) v[j < 4 ? j : 0] = rel[j]; a.line_hist[((size_t)slot * NL + j) * a.N + i] = rel[j]; }
    if (lds_line) lds_line[slot * kBlock + threadIdx.x] = make_float4(v[0], v[1], v[2], v[3]);
}

// Reward after `steps` (>= L) transitions: minus the mean float64 distance of the L newest states from
// the line through their mean along the dominant right-singular vector of the centred float32 data.
// The reference takes that vector from LAPACK's float32 SVD; here it is the dominant eigenvector of
// the 4x4 float64 scatter matrix (40 normalised squarings: B <- B B / tr), rounded to float32 like
// LAPACK's output.  The two agree to float32 rounding divided by the gap between the two largest
// singular values, which is the accuracy the reference's own reward has.
template <bool CACHED, int NL = 4>
__device__ __forceinline__ double c_line_reward(const ContinuousArgs &a, long i, uint32_t steps, const float4 *lds_line = nullptr) {
    const int L = a.line_L, n = a.n_rel;
    const size_t N = (size_t)a.N;
    // slot of the oldest of the L newest states; walked with a wrap instead of a modulo per point
    const uint32_t slot0 = (steps + 1u) % (uint32_t)L;
    uint32_t slot = slot0;
    static_assert(NL == 4 || (NL == 8 && !CACHED), "8-wide rows: the uncached walk");
    float x[NL];
    auto next_pt = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NL; j++) x[j] = (j < n) ? a.line_hist[((size_t)slot * NL + j) * N + i] : 0.0f;
        slot = (slot + 1u == (uint32_t)L) ? 0u : slot + 1u;
    };
    // CACHED (L <= 16): all points are fetched up front into registers -- 64 loads in flight instead
    // of one L2 round trip per point in each of the two passes (points beyond L repeat the newest)
    constexpr int kCache = 16;
    float px[CACHED ? kCache : 1][NL];
    if constexpr (CACHED) {
#pragma unroll
        for (int k = 0; k < kCache; k++) {
            if (lds_line) {                                  // (wave-uniform) one ds_read_b128 per point
                const float4 q = lds_line[slot * kBlock + threadIdx.x];
                px[k][0] = q.x; px[k][1] = q.y; px[k][2] = q.z; px[k][3] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) px[k][j] = (j < n) ? a.line_hist[((size_t)slot * 4 + j) * N + i] : 0.0f;
            }
            const uint32_t nx = (slot + 1u == (uint32_t)L) ? 0u : slot + 1u;
            slot = (k + 1 < L) ? nx : slot;
        }
    }
    // One pass over the points: (1) data_.mean(axis=0) the way numpy sums a contiguous float32 column
    // (pairwise routine: 8 running sums while 8 more points are left, a fixed tree, the rest one by
    // one; plain left-to-right below 8 points), divided by L in float32; (2) float64 raw moments for
    // the scatter matrix about that mean: S = sum x x^T - m s^T - s m^T + L m m^T with s = sum x.
    double s1[NL], m[NL][NL];
#pragma unroll
    for (int p = 0; p < NL; p++) {
        s1[p] = 0.0;
#pragma unroll
        for (int q = 0; q < NL; q++) m[p][q] = 0.0;
    }
    auto moments = [&](const float (&y)[NL]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NL; p++) {
            s1[p] += (double)y[p];
#pragma unroll
            for (int q = p; q < NL; q++) m[p][q] = fma((double)y[p], (double)y[q], m[p][q]);
        }
    };
    float mean[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) mean[j] = 0.0f;
    if constexpr (CACHED) {
        float r[8][4];
#pragma unroll
        for (int q = 0; q < 8; q++)
#pragma unroll
            for (int j = 0; j < 4; j++) r[q][j] = (L == kCache) ? px[q][j] + px[8 + q][j] : px[q][j];
        const int tail0 = (L < 8) ? 0 : (L == kCache ? kCache : 8);      // first point summed one by one
#pragma unroll
        for (int j = 0; j < 4; j++)
            mean[j] = (L < 8) ? 0.0f
                              : ((r[0][j] + r[1][j]) + (r[2][j] + r[3][j])) + ((r[4][j] + r[5][j]) + (r[6][j] + r[7][j]));
#pragma unroll
        for (int k = 0; k < kCache; k++) {
            if (k < L) {                                   // (wave-uniform)
                moments(px[k]);
                if (k >= tail0) {
#pragma unroll
                    for (int j = 0; j < 4; j++) mean[j] += px[k][j];
                }
            }
        }
    } else {
        int k = 0;
        if (L >= 8) {
            float r[8][NL];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                next_pt(); moments(x);
#pragma unroll
                for (int j = 0; j < NL; j++) r[q][j] = x[j];
            }
            for (k = 8; k < L - (L % 8); k += 8) {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    next_pt(); moments(x);
#pragma unroll
                    for (int j = 0; j < NL; j++) r[q][j] += x[j];
                }
            }
#pragma unroll
            for (int j = 0; j < NL; j++)
                mean[j] = ((r[0][j] + r[1][j]) + (r[2][j] + r[3][j])) + ((r[4][j] + r[5][j]) + (r[6][j] + r[7][j]));
        }
        for (; k < L; k++) {
            next_pt(); moments(x);
#pragma unroll
            for (int j = 0; j < NL; j++) mean[j] += x[j];
        }
    }
#pragma unroll
    for (int j = 0; j < NL; j++) mean[j] = (j < n) ? mean[j] / (float)L : 0.0f;
#pragma unroll
    for (int p = 0; p < NL; p++)
#pragma unroll
        for (int q = p; q < NL; q++) {
            const double mp = (double)mean[p], mq = (double)mean[q];
            m[p][q] = m[p][q] - mp * s1[q] - s1[p] * mq + (double)L * mp * mq;
            m[q][p] = m[p][q];
        }
    double v[NL];                                 // all points equal: LAPACK returns the identity, vv[0] = e0
    double tr = 0.0;
#pragma unroll
    for (int p = 0; p < NL; p++) { v[p] = p == 0 ? 1.0 : 0.0; tr += m[p][p]; }
    if (tr > 0.0) {
        // B <- B B, rescaled by a power of two (exact) so that tr(B) stays in [1/2, 1).  With eigenvalues
        // mu_i of B, tr(B B) / tr(B)^2 = sum mu_i^2 / (sum mu_i)^2 reaches 1 when B has rank one: stop
        // when every lane of the wave is there (6-10 squarings for a random walk, 40 at most)
        auto pow2_inv = [](double t) __attribute__((always_inline)) -> double {   // 2^-(exponent of t)
            const uint64_t e = ((uint64_t)__double_as_longlong(t) >> 52) & 0x7FFu;
            return __longlong_as_double((long long)((2046ull - e - 1ull) << 52));
        };
        double sc = pow2_inv(tr);
#pragma unroll
        for (int p = 0; p < NL; p++)
#pragma unroll
            for (int q = 0; q < NL; q++) m[p][q] *= sc;
        tr *= sc;
        for (int it = 0; it < 40; it++) {
            double sq[NL][NL];
            double tr2 = 0.0;
#pragma unroll
            for (int p = 0; p < NL; p++)
#pragma unroll
                for (int q = p; q < NL; q++) {
                    double acc = 0.0;
#pragma unroll
                    for (int r = 0; r < NL; r++) acc = fma(m[p][r], m[r][q], acc);
                    sq[p][q] = acc;
                    if (q == p) tr2 += acc;
                }
            // (tr^2 - tr2 = 2 sum_{i<j} mu_i mu_j ~ 2 mu_1 mu_2: the relative weight of everything but the dominant
            //  direction; the vector is rounded to float32 below, 1e-12 is five digits beyond that)
            const bool conv = (tr * tr - tr2) <= 2e-12 * tr * tr;
            sc = pow2_inv(tr2);
#pragma unroll
            for (int p = 0; p < NL; p++)
#pragma unroll
                for (int q = p; q < NL; q++) { m[p][q] = sq[p][q] * sc; m[q][p] = m[p][q]; }
            tr = tr2 * sc;
            if (__builtin_amdgcn_ballot_w64(!conv) == 0) break;
        }
        // m ~ v v^T: the column with the largest diagonal entry, normalised
        double best = m[0][0], col[NL];
#pragma unroll
        for (int p = 0; p < NL; p++) col[p] = m[p][0];
#pragma unroll
        for (int q = 1; q < NL; q++) {
            const bool b = m[q][q] > best;
            best = b ? m[q][q] : best;
#pragma unroll
            for (int p = 0; p < NL; p++) col[p] = b ? m[p][q] : col[p];
        }
        double nn = 0.0;
#pragma unroll
        for (int p = 0; p < NL; p++) nn += col[p] * col[p];
        const double s = 1.0 / sqrt(nn);
#pragma unroll
        for (int p = 0; p < NL; p++) v[p] = col[p] * s;
    }
    // line_end_pts = vv[0] * [-1, 1][:, None] + data_mean (float64 from here on)
    double ptA[NL], ab[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) {
        const double vj = (double)(float)v[j], mj = (double)mean[j];
        ptA[j] = vj * -1.0 + mj;
        ab[j] = ptA[j] - (vj * 1.0 + mj);
    }
    double nab = 0.0;
#pragma unroll
    for (int j = 0; j < NL; j++) if (j < n) nab = fma(ab[j], ab[j], nab);    // np.dot: a chain of FMAs
    // dist_of_pt_from_line (:2546-2576): proj = dot / |ab|, dist = sqrt(|ap|^2 - proj^2), with |ap| taken as
    // sqrt(dot(ap, ap)) and squared again there.  Here |ap|^2 is the dot product itself and dot^2 / |ab|^2 uses one
    // reciprocal per step: one square root per point instead of two and a division -- a difference of an ulp or two of
    // float64 in a reward that is defined to ~1e-7 by its float32 singular vector (tests: LINE_ATOL).
    const bool degenerate = sqrt(nab) < 1e-13;
    const double inv_nab2 = 1.0 / nab;
    auto dist_of = [&](const float (&y)[NL]) __attribute__((always_inline)) -> double {
        double dot = 0.0, nap = 0.0;
#pragma unroll
        for (int j = 0; j < NL; j++) {
            if (j < n) {
                const double ap = ptA[j] - (double)y[j];
                dot = fma(ab[j], ap, dot);
                nap = fma(ap, ap, nap);
            }
        }
        double sq = nap - (dot * dot) * inv_nab2;
        sq = sq < 0.0 ? 0.0 : sq;
        return degenerate ? 0.0 : sqrt(sq);
    };
    double total = 0.0;
    if constexpr (CACHED) {
#pragma unroll
        for (int k = 0; k < kCache; k++)
            if (k < L) total += dist_of(px[k]);
    } else {
        slot = slot0;
        for (int kk = 0; kk < L; kk++) {
            next_pt();
            total += dist_of(x);
        }
    }
    return 0.0 + -total / (double)L;
}

// The same fit for MORE than 8 relevant dimensions (or 5 to 8 of more than 12 state dimensions) -- the reference has no limit
// (:1865-1910): an n x n float64 scatter matrix does not fit a lane's registers beyond n = 8, so the two matrices of the
// squaring iteration, the column sums and the mean live in a per-env HBM workspace (ContinuousArgs::line_ws,
// [(2 n n + 3 n)][N] doubles, coalesced over envs; L2-resident while a lane works on it) and every loop runs over the
// handle's n.  The arithmetic -- summation orders, explicit FMAs, the stopping rule -- is c_line_reward's, statement by
// statement; only where the operands live differs.  No speed claim: a fit of n = 32, L = 64 is ~10^5 memory operations per
// env step; the reference's experiments stop at 4 dimensions.
__device__ __forceinline__ double c_line_reward_big(const ContinuousArgs &a, long i, uint32_t steps) {
    const int L = a.line_L, n = a.n_rel;
    const size_t N = (size_t)a.N, NL = (size_t)a.line_NL, nn_ = (size_t)n * n;
    double *const W0 = a.line_ws + i, *const W1 = W0 + nn_ * N, *const S1 = W1 + nn_ * N, *const MEAN = S1 + (size_t)n * N,
                 *const VV = MEAN + (size_t)n * N;
    const uint32_t slot0 = (steps + 1u) % (uint32_t)L;
    auto pt = [&](int k, int j) -> float {             // coordinate j of the k-th oldest of the L newest states
        uint32_t slot = slot0 + (uint32_t)k;
        slot = slot >= (uint32_t)L ? slot - (uint32_t)L : slot;
        return a.line_hist[((size_t)slot * NL + j) * N + i];
    };
    // (1) the float32 mean, numpy's pairwise order (see c_line_reward); column sums and raw moments in float64, point by point
    for (int j = 0; j < n; j++) {
        float mean = 0.0f;
        int k = 0;
        if (L >= 8) {
            float r[8];
#pragma unroll
            for (int q = 0; q < 8; q++) r[q] = pt(q, j);
            for (k = 8; k < L - (L % 8); k += 8) {
#pragma unroll
                for (int q = 0; q < 8; q++) r[q] += pt(k + q, j);
            }
            mean = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        }
        for (; k < L; k++) mean += pt(k, j);
        mean = mean / (float)L;
        double s1 = 0.0;
        for (int kk = 0; kk < L; kk++) s1 += (double)pt(kk, j);
        S1[(size_t)j * N] = s1;
        MEAN[(size_t)j * N] = (double)mean;
    }
    double tr = 0.0;
    for (int p = 0; p < n; p++)
        for (int q = p; q < n; q++) {
            double acc = 0.0;
            for (int k = 0; k < L; k++) acc = fma((double)pt(k, p), (double)pt(k, q), acc);
            const double mp = MEAN[(size_t)p * N], mq = MEAN[(size_t)q * N];
            acc = acc - mp * S1[(size_t)q * N] - S1[(size_t)p * N] * mq + (double)L * mp * mq;
            W0[((size_t)p * n + q) * N] = acc;
            W0[((size_t)q * n + p) * N] = acc;
            if (q == p) tr += acc;
        }
    for (int p = 0; p < n; p++) VV[(size_t)p * N] = p == 0 ? 1.0 : 0.0;     // all points equal: LAPACK's identity, vv[0] = e0
    if (tr > 0.0) {
        auto pow2_inv = [](double t) __attribute__((always_inline)) -> double {
            const uint64_t e = ((uint64_t)__double_as_longlong(t) >> 52) & 0x7FFu;
            return __longlong_as_double((long long)((2046ull - e - 1ull) << 52));
        };
        double sc = pow2_inv(tr);
        for (size_t e = 0; e < nn_; e++) W0[e * N] *= sc;
        tr *= sc;
        double *m = W0, *sq = W1;
        for (int it = 0; it < 40; it++) {
            double tr2 = 0.0;
            for (int p = 0; p < n; p++)
                for (int q = p; q < n; q++) {
                    double acc = 0.0;
                    for (int r = 0; r < n; r++) acc = fma(m[((size_t)p * n + r) * N], m[((size_t)r * n + q) * N], acc);
                    sq[((size_t)p * n + q) * N] = acc;
                    if (q == p) tr2 += acc;
                }
            const bool conv = (tr * tr - tr2) <= 2e-12 * tr * tr;
            sc = pow2_inv(tr2);
            for (int p = 0; p < n; p++)
                for (int q = p; q < n; q++) {
                    const double v = sq[((size_t)p * n + q) * N] * sc;
                    sq[((size_t)p * n + q) * N] = v;
                    sq[((size_t)q * n + p) * N] = v;
                }
            tr = tr2 * sc;
            double *t = m; m = sq; sq = t;
            if (__builtin_amdgcn_ballot_w64(!conv) == 0) break;
        }
        int bq = 0;
        double best = m[0];
        for (int q = 1; q < n; q++) {
            const double d = m[((size_t)q * n + q) * N];
            if (d > best) { best = d; bq = q; }
        }
        double nn = 0.0;
        for (int p = 0; p < n; p++) { const double c = m[((size_t)p * n + bq) * N]; nn += c * c; }
        const double s = 1.0 / sqrt(nn);
        for (int p = 0; p < n; p++) VV[(size_t)p * N] = m[((size_t)p * n + bq) * N] * s;
    }
    // line_end_pts = vv[0] * [-1, 1][:, None] + data_mean; ptA into S1's place, ab into VV's
    double nab = 0.0;
    for (int j = 0; j < n; j++) {
        const double vj = (double)(float)VV[(size_t)j * N], mj = MEAN[(size_t)j * N];
        const double pa = vj * -1.0 + mj, ab = pa - (vj * 1.0 + mj);
        S1[(size_t)j * N] = pa;
        VV[(size_t)j * N] = ab;
        nab = fma(ab, ab, nab);
    }
    const bool degenerate = sqrt(nab) < 1e-13;
    const double inv_nab2 = 1.0 / nab;
    double total = 0.0;
    for (int k = 0; k < L; k++) {
        double dot = 0.0, nap = 0.0;
        for (int j = 0; j < n; j++) {
            const double ap = S1[(size_t)j * N] - (double)pt(k, j);
            dot = fma(VV[(size_t)j * N], ap, dot);
            nap = fma(ap, ap, nap);
        }
        double sqd = nap - (dot * dot) * inv_nab2;
        sqd = sqd < 0.0 ? 0.0 : sqd;
        total += degenerate ? 0.0 : sqrt(sqd);
    }
    return 0.0 + -total / (double)L;
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

template <int DMAX, int OMAX, bool PHILOX, int NL = 4>
__global__ __launch_bounds__(kBlock) void k_continuous_step(ContinuousArgs a, int K,
                                                            const float *__restrict__ actions,
                                                            float *__restrict__ obs,
                                                            float *__restrict__ reward,
                                                            uint8_t *__restrict__ term,
                                                            uint8_t *__restrict__ trunc,
                                                            float *__restrict__ final_obs) {
    const uint64_t ptick0 = tick_now(a);               // the step counter at this launch (through the device-side offset of a graph replay)
    const uint32_t rhead0 = ring_head_now(a, ptick0);    // ... and the head of a delay line kept in memory
    __shared__ uint64_t s_ki[256];
    __shared__ double s_wi[256], s_fi[256];
    __shared__ double s_z[DMAX * kBlock];                    // this step's transition-noise normals, [d][lane]
    extern __shared__ __align__(16) float4 s_line[];         // move_along_a_line, L <= 16: [L][lane] (launch_step_t sizes it)
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
    // next-step autoreset: "episode ended, reset at the next call" travels in bit 1 of the flags word
    const bool next_step = a.autoreset == MDPP_AUTORESET_NEXT_STEP;
    bool pending = next_step && (flags & 2u) != 0;
    flags &= ~2u;

    float4 *lds_line = nullptr;
    if (NL == 4 && a.line_L && a.line_lds) {                 // this lane's L points: HBM -> LDS once per launch
        lds_line = s_line;
        for (int sl = 0; sl < a.line_L; sl++) {
            float q[4];
#pragma unroll
            for (int j = 0; j < 4; j++) q[j] = (j < a.n_rel) ? a.line_hist[((size_t)sl * 4 + j) * N + i] : 0.0f;
            s_line[sl * kBlock + threadIdx.x] = make_float4(q[0], q[1], q[2], q[3]);
        }
    }
    Pcg64 env_pcg, sp_pcg;
    Philox env_phx, sp_phx;
    bool sp_loaded = false;
    const bool need_env = a.has_p_noise || a.has_r_noise;
    if (!PHILOX && need_env) env_pcg.load(a.env_s, a.env_inc, i);

    // [env][D] row-major, 16 B per lane per load when possible
    auto load_action = [&](int k, float (&dst)[DMAX]) __attribute__((always_inline)) {
        const float *ap = actions + ((long)k * N + i) * D;
        if (DMAX % 4 == 0 && D == DMAX) {
#pragma unroll
            for (int q = 0; q < DMAX / 4; q++) {
                float4 v = ((const float4 *)ap)[q];
                dst[4 * q] = v.x; dst[4 * q + 1] = v.y; dst[4 * q + 2] = v.z; dst[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int d = 0; d < DMAX; d++) dst[d] = (d < D) ? ap[d] : 0.0f;
        }
    };
    // Per-dimension constants pinned in VECTOR registers.  Left to the compiler they are wave-uniform
    // kernel arguments it keeps re-reading with scalar loads once the SGPR file is full: the step
    // loop ran 102 s_load instructions per step and waited 58 % of its cycles (tools/pmc_sq.sh).
    float tgt[DMAX], smax = a.smax32, amax = a.amax32, inertia = a.inertia32, inv_inertia = a.inv_inertia32, tpw[OMAX + 1];
    double fct[OMAX + 1], ifct[OMAX + 1], pns = a.p_noise;
#pragma unroll
    for (int j = 0; j < DMAX; j++) { tgt[j] = (j < a.n_rel) ? a.target[j] : 0.0f; asm volatile("" : "+v"(tgt[j])); }
#pragma unroll
    for (int j = 0; j <= OMAX; j++) {
        tpw[j] = a.tpow32[j]; fct[j] = a.fact[j]; ifct[j] = a.inv_fact[j];
        asm volatile("" : "+v"(tpw[j]), "+v"(fct[j]), "+v"(ifct[j]));
    }
    asm volatile("" : "+v"(smax), "+v"(amax), "+v"(inertia), "+v"(inv_inertia), "+v"(pns));
    // reset() of this lane (same-step autoreset after a finished episode; next-step: the call after it)
    auto lane_reset = [&]() __attribute__((always_inline)) {
        if (PHILOX) {
            c_reset_lane<DMAX, OMAX>(a, sp_phx, sd, cur, status);
        } else {
            if (!sp_loaded) { sp_pcg.load(a.sp_s, a.sp_inc, i); sp_loaded = true; }
            c_reset_lane<DMAX, OMAX>(a, sp_pcg, sd, cur, status);
        }
        if (a.est.cur) est_roll(a.est, N, i, steps);               // reset(): :2231-2247, :2360-2369
        steps = 0; flags = 0;
        if (a.line_L) {
            c_gather_rel<DMAX>(a, cur, rel);
            c_line_put<DMAX>(a, i, 0u, rel, lds_line);
        }
        if (a.rew64) {
            for (int dd = 0; dd < a.delay; dd++) a.ring64[(size_t)dd * N + i] = 0.0;
        } else {
            for (int dd = 0; dd < a.delay; dd++) a.ring[(size_t)dd * N + i] = kRingPyZero;
        }
    };
    auto put_obs = [&](long o) __attribute__((always_inline)) {
        float *op = obs + o * D;
        if (DMAX % 4 == 0 && D == DMAX) {
#pragma unroll
            for (int q = 0; q < DMAX / 4; q++)
                ((float4 *)op)[q] = make_float4(cur[4 * q], cur[4 * q + 1], cur[4 * q + 2], cur[4 * q + 3]);
        } else {
#pragma unroll
            for (int d = 0; d < DMAX; d++) if (d < D) op[d] = cur[d];
        }
    };
    float nact[DMAX];
    load_action(0, nact);
    for (int k = 0; k < K; k++) {
        const uint32_t tick = rhead0 + (uint32_t)k;              // ring head (mod delay below)
        const uint64_t ptick = ptick0 + (uint64_t)k;
        const long o = (long)k * N + i;
        if (PHILOX) {
            env_phx.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), ptick, MDPP_STREAM_ENV);
            sp_phx.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), ptick, MDPP_STREAM_SPACE);
        }
        // ---- action: fetched one step ahead (with one wave per SIMD nothing else hides the load)
#pragma unroll
        for (int d = 0; d < DMAX; d++) act[d] = nact[d];
        load_action(k + 1 < K ? k + 1 : k, nact);
        if (pending) {               // next-step autoreset: this call is the env's reset(), :2284-2323
            lane_reset();
            put_obs(o);
            reward[o] = 0.0f; term[o] = 0; trunc[o] = 0;
            pending = false;
            continue;
        }
        // ---- C1
        bool ok = true;
#pragma unroll
        for (int d = 0; d < DMAX; d++)
            if (d < D) ok = ok && (act[d] >= -amax) && (act[d] <= amax);
        if (ok) {
            // ---- C2: lower orders first, each using the not-yet-updated higher ones
#pragma unroll
            for (int d = 0; d < DMAX; d++) {
#pragma unroll
                for (int kk = 0; kk <= OMAX; kk++)
                    if (kk == n) sd[kk][d] = a.inertia_pow2 ? act[d] * inv_inertia : act[d] / inertia;   // (exact for 2^k)
            }
#pragma unroll
            for (int ii = 0; ii < OMAX; ii++) {
#pragma unroll
                for (int j = 0; j < OMAX; j++) {
                    if (ii < n && j < n - ii) {
#pragma unroll
                        for (int d = 0; d < DMAX; d++) {
                            float prod = sd[(ii + j + 1 <= OMAX) ? ii + j + 1 : OMAX][d] * tpw[(j + 1 <= OMAX) ? j + 1 : OMAX];
                            // 1! and 2! are powers of two: multiplying by the reciprocal is the same float64
                            const double trm = ((a.fact_pow2_mask >> (j + 1)) & 1u)
                                                   ? (double)prod * ifct[(j + 1 <= OMAX) ? j + 1 : OMAX]
                                                   : (double)prod / fct[(j + 1 <= OMAX) ? j + 1 : OMAX];
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
        // ---- C3.  The D normals are drawn in a ROLLED loop into this lane's LDS column and added from
        // there: unrolled, D copies of the ziggurat make the loop body ~75 KB of code, and even a
        // noise-free step then pays an instruction-cache miss per skipped copy (0.4 us per dimension).
        if (a.has_p_noise) {
#pragma unroll 1
            for (int d = 0; d < D; d++)
                s_z[d * kBlock + threadIdx.x] =
                    0.0 + pns * (PHILOX ? np_standard_normal_lds(env_phx, zig) : np_standard_normal_lds(env_pcg, zig));
            if (a.est.cur) {                                       // total_abs_noise_in_transition_episode, :1686
#pragma unroll 1
                for (int d = 0; d < D; d++) est_add(a.est, N, i, 3 + d, fabs(s_z[d * kBlock + threadIdx.x]));
            }
        }
#pragma unroll
        for (int d = 0; d < DMAX; d++) {
            if (d < D) {
                const double nz = a.has_p_noise ? s_z[d * kBlock + threadIdx.x] : 0.0;
                nxt[d] = (float)((double)nxt[d] + nz);
            }
        }
        // ---- C4
        bool inside = true;
#pragma unroll
        for (int d = 0; d < DMAX; d++)
            if (d < D) inside = inside && (nxt[d] >= -smax) && (nxt[d] <= smax);
        // image observations: the reference asks the ImageContinuous space whether it contains the
        // state VECTOR, which it never does (spaces/image_continuous.py:292-302 returns None), so every
        // step takes this branch: a clip that is the identity inside the box, and zeroed derivatives
        if (a.image_quirk) inside = false;
        if (!inside) {
#pragma unroll
            for (int d = 0; d < DMAX; d++) {
                float x = nxt[d];
                if (x < -smax) x = -smax;
                if (x > smax) x = smax;
                nxt[d] = x;
            }
#pragma unroll
            for (int kk = 0; kk <= OMAX; kk++)
#pragma unroll
                for (int d = 0; d < DMAX; d++) sd[kk][d] = (kk == 0) ? nxt[d] : 0.0f;
        }
        // ---- C5 (the target latch exists for move_to_a_point only)
        c_gather_rel<DMAX>(a, nxt, rel);
        const float dist_new = (a.line_L || a.target64) ? 0.0f : c_norm_rel<DMAX>(a, rel, tgt);
        const double dist_new64 = a.target64 ? c_norm64<DMAX>(a, rel) : 0.0;
        const bool within = a.target64 ? (dist_new64 < a.radius) : (dist_new < a.radius32);
        if (!a.line_L && within) flags |= 1u;
        const bool in_box = (a.n_boxes > 0) && c_in_box<DMAX>(a, rel);
        steps += 1;
        // ---- C6
        CRew r;
        if (a.line_L) {
            // gate (:1856): the state sequence_length transitions back must exist
            c_line_put<DMAX>(a, i, steps, rel, lds_line);
            r.v = 0.0;
            if (steps >= (uint32_t)a.line_L) {
                if constexpr (NL == 4) r.v = a.line_ws ? c_line_reward_big(a, i, steps)
                                                       : a.line_L <= 16 ? c_line_reward<true, 4>(a, i, steps, lds_line) : c_line_reward<false, 4>(a, i, steps);
                else r.v = c_line_reward<false, NL>(a, i, steps);
            }
            r.is32 = false;
        } else if (a.make_denser && a.target64) {
            float relo[DMAX];
            c_gather_rel<DMAX>(a, cur, relo);
            r.v = -dist_new64;                            // np.float64 (:1926)
            r.v = r.v + c_norm64<DMAX>(a, relo);          // :1929
        } else if (a.make_denser) {
            float relo[DMAX];
            c_gather_rel<DMAX>(a, cur, relo);
            const float dist_old = c_norm_rel<DMAX>(a, relo, tgt);
            r.v = (double)(float)(-dist_new + dist_old);
        } else {
            r.v = within ? 1.0 : 0.0;
        }
        if (!a.line_L) {
            double acc = 0.0;
#pragma unroll
            for (int d = 0; d < DMAX; d++)
                if (d < D) { float p = act[d] * act[d]; acc += (double)p; }
            float pen = a.alw32 * sqrtf((float)acc);      // Python float * np.float32 -> np.float32
            if (a.rew64) { r.v = r.v - (double)pen; r.is32 = false; }      // np.float64 - np.float32
            else { r.v = (double)((float)r.v - pen); r.is32 = true; }
        }
        // ---- C7
        if (a.delay > 0 && a.rew64) {
            double *slot = a.ring64 + (size_t)(tick % (uint32_t)a.delay) * N + i;
            const double out = *slot;
            *slot = r.v;
            r.v = out;
        } else if (a.delay > 0) {
            uint32_t *slot = a.ring + (size_t)(tick % (uint32_t)a.delay) * N + i;
            uint32_t bits = *slot;
            *slot = __float_as_uint((float)r.v);
            if (bits == kRingPyZero) { r.v = 0.0; r.is32 = false; }
            else { r.v = (double)__uint_as_float(bits); r.is32 = true; }
        }
        if (steps % (uint32_t)a.every_n != 0) { r.v = 0.0; r.is32 = false; }
        if (a.est.cur) {
            // total_reward_episode += reward (:1985): an np.float32 reward makes the running sum np.float32 (int 0 + float32,
            // then float32 + float32, and float32 + a Python 0.0 stays float32); float64 rewards (line reward, default
            // target) sum in float64
            double &acc = a.est.cur[(size_t)1 * N + i];
            if (a.rew64) acc += r.v;
            else if (r.is32) acc = (double)((float)acc + (float)r.v);
        }
        if (a.has_r_noise) {
            double nz = 0.0 + a.r_noise * (PHILOX ? np_standard_normal_lds(env_phx, zig) : np_standard_normal_lds(env_pcg, zig));
            if (a.est.cur) est_add(a.est, N, i, 0, fabs(nz));      // total_abs_noise_in_reward_episode, :1984
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

        if (next_step) pending = done || truncated;
        if (a.autoreset == MDPP_AUTORESET_SAME_STEP && (done || truncated)) {
            if (final_obs) {
#pragma unroll
                for (int d = 0; d < DMAX; d++) if (d < D) final_obs[o * D + d] = nxt[d];
            }
            lane_reset();
        }
        // ---- outputs
        put_obs(o);
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
    a.meta[i] = make_uint2(steps, flags | (pending ? 2u : 0u));
    if (!PHILOX) {
        if (need_env) env_pcg.store(a.env_s, i);
        if (sp_loaded) sp_pcg.store(a.sp_s, i);
    }
    if (status) atomicOr(&a.status[i], status);
}

template <int DMAX, int OMAX, bool PHILOX>
__global__ __launch_bounds__(kBlock) void k_continuous_reset(ContinuousArgs a, uint64_t reset_tick,
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
    if (a.est.cur) est_roll(a.est, a.N, i, a.meta[i].x);
    a.meta[i] = make_uint2(0u, 0u);
    if (a.line_L) {
        float rel[DMAX];
        c_gather_rel<DMAX>(a, cur, rel);
        c_line_put<DMAX>(a, i, 0u, rel);
    }
    if (a.rew64) {
        for (int dd = 0; dd < a.delay; dd++) a.ring64[(size_t)dd * N + i] = 0.0;
        if (status) atomicOr(&a.status[i], status);
        return;
    }
    for (int dd = 0; dd < a.delay; dd++) a.ring[(size_t)dd * N + i] = kRingPyZero;
    if (status) atomicOr(&a.status[i], status);
}

// ---- dispatch on (padded D, padded order) -------------------------------------------------
template <int DMAX, int OMAX, int NL = 4>
static void launch_step_t(const ContinuousArgs &a, int K, const float *actions, float *obs,
                          float *reward, uint8_t *term, uint8_t *trunc, float *final_obs,
                          hipStream_t s, char *name_out) {
    const int grid = (a.N + kBlock - 1) / kBlock;
    if (name_out) {
        if (NL == 4) snprintf(name_out, kNameLen, "k_continuous_step<DMAX=%d,OMAX=%d,PHILOX=%d>", DMAX, OMAX, a.philox != 0);
        else snprintf(name_out, kNameLen, "k_continuous_step<DMAX=%d,OMAX=%d,PHILOX=%d,NL=%d>", DMAX, OMAX, a.philox != 0, NL);
        return;
    }
    // move_along_a_line with L <= 16 (rows of 4), rollouts: the L points of every lane mirrored in (dynamic) LDS for the launch
    ContinuousArgs al = a;
    al.line_lds = (NL == 4 && !a.line_ws && a.line_L > 0 && a.line_L <= 16 && K >= 4) ? 1 : 0;
    size_t lds = al.line_lds ? (size_t)a.line_L * kBlock * sizeof(float4) : 0;
    const void *kern = a.philox ? (const void *)k_continuous_step<DMAX, OMAX, true, NL> : (const void *)k_continuous_step<DMAX, OMAX, false, NL>;
    if (!dynamic_lds_ok(kern, lds)) { al.line_lds = 0; lds = 0; }        // (no room: the points stay in HBM)
    if (a.philox)
        hipLaunchKernelGGL((k_continuous_step<DMAX, OMAX, true, NL>), dim3(grid), dim3(kBlock), lds, s, al,
                           K, actions, obs, reward, term, trunc, final_obs);
    else
        hipLaunchKernelGGL((k_continuous_step<DMAX, OMAX, false, NL>), dim3(grid), dim3(kBlock), lds, s, al,
                           K, actions, obs, reward, term, trunc, final_obs);
}
#if MDPP_CONT_TU_LINE8
// move_along_a_line with 5 to 8 relevant dimensions (state_space_dim <= 12): rows of 8, an 8 x 8 scatter matrix in registers
bool launch_continuous_step_line8(const ContinuousArgs &a, int K, const float *actions, float *obs, float *reward,
                                  uint8_t *term, uint8_t *trunc, float *final_obs, hipStream_t s, char *name_out) {
    if (a.line_NL != 8 || a.D > 12 || a.order > 4) return false;
    if (a.order <= 1) launch_step_t<12, 1, 8>(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
    else if (a.order <= 2) launch_step_t<12, 2, 8>(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
    else launch_step_t<12, 4, 8>(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
    return true;
}
#else
bool launch_continuous_step_line8(const ContinuousArgs &a, int K, const float *actions, float *obs, float *reward,
                                  uint8_t *term, uint8_t *trunc, float *final_obs, hipStream_t s, char *name_out);
template <int DMAX, int OMAX>
static void launch_reset_t(const ContinuousArgs &a, uint64_t reset_tick, const uint8_t *mask,
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
        else if (D_ <= 32) { CALL(32, 4); }                                         \
        else { return MDPP_EUNSUPPORTED; }                                          \
    } while (0)

int launch_continuous_step(mdpp_env *h, int K, const float *actions, float *obs, float *reward,
                           uint8_t *term, uint8_t *trunc, float *final_obs, hipStream_t s, char *name_out) {
    ContinuousArgs a = h->cargs;
    a.opts = h->opts;
    a.ptick = h->tick;
    a.dtick = h->graph_capture ? (const uint64_t *)h->d_tick_off : nullptr;     // (launches being captured into a HIP graph)
    a.tick = a.delay > 0 ? (uint32_t)(h->tick % (uint64_t)a.delay) : 0u;
    if (K == 1 && launch_continuous_step1(a, actions, obs, reward, term, trunc, final_obs, s, name_out)) {
        // mdpp_step on the fast shape: the one-step form of the rollout kernel (mdpp_continuous_step1.hip)
        if (name_out) return MDPP_OK;
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { h->err = std::string("k_continuous_step1 launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
        h->tick += 1;
        return MDPP_OK;
    }
    if (a.fast_ok) {
        // common shape: dedicated rollout kernel (mdpp_continuous_fast.hip); its buffer descriptors
        // address < 4 GiB per array, so long rollouts go out as several launches
        const long long kmax = ((1LL << 32) - 1) / ((long long)a.N * a.D * 4);
        bool served = kmax >= 1;
        for (int k0 = 0; served && k0 < K;) {
            const int kc = (int)((K - k0) < kmax ? (K - k0) : kmax);
            const size_t off = (size_t)k0 * a.N;
            a.ptick = h->tick + (uint64_t)k0;
            a.tick = a.delay > 0 ? (uint32_t)(a.ptick % (uint64_t)a.delay) : 0u;   // head of the delay ring for this piece
            served = launch_continuous_fast(a, kc, actions + off * a.D, obs + off * a.D, reward + off,
                                            term + off, trunc + off,
                                            final_obs ? final_obs + off * a.D : nullptr, s, name_out);
            if (served && name_out) return MDPP_OK;      // (the first piece names the launch)
            if (!served && k0 > 0) { h->err = "k_continuous_rollout_fast: inconsistent dispatch"; return MDPP_EHIP; }
            k0 += kc;
        }
        if (served) {
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { h->err = std::string("k_continuous_rollout_fast launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
            h->tick += (uint64_t)K;
            return MDPP_OK;
        }
        a.ptick = h->tick;
        a.tick = a.delay > 0 ? (uint32_t)(h->tick % (uint64_t)a.delay) : 0u;
    }
    // move_along_a_line in its common shape: the dedicated rollout kernel (mdpp_continuous_line.hip)
    if (launch_continuous_line(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out)) {
        if (name_out) return MDPP_OK;
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { h->err = std::string("k_continuous_line_rollout launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
        h->tick += (uint64_t)K;
        return MDPP_OK;
    }
    if (a.line_L && a.line_NL == 8 && !a.line_ws) {
        if (!launch_continuous_step_line8(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out)) {
            h->err = "k_continuous_step<NL=8>: move_along_a_line with 5 to 8 relevant dimensions needs state_space_dim <= 12";
            return MDPP_EUNSUPPORTED;
        }
    } else {
#define CALL_STEP(DM, OM) launch_step_t<DM, OM>(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out)
        MDPP_C_DISPATCH(CALL_STEP);
#undef CALL_STEP
    }
    if (name_out) return MDPP_OK;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("k_continuous_step launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
    h->tick += (uint64_t)K;
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
#endif   // !MDPP_CONT_TU_LINE8

} // namespace mdpp

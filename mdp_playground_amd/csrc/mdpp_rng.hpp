// Device-side random streams for the RLToyEnv kernels (gfx950).
//
// Two bit generators behind one interface (next64()):
//   * Pcg64  — numpy's PCG64 (pcg_setseq_128_xsl_rr_64) with per-env state in HBM, so that
//              an env seeded like the reference draws the very same variates as
//              np.random.Generator does in rl_toy_env.py (reset :2255, noise :403/:413,
//              DiscreteExtended.sample spaces/discrete_extended.py:17).
//   * Philox — stateless Philox4x32-10 keyed by (seed, global env id, tick, stream): no RNG
//              bytes in HBM, results independent of how envs are sharded over GPUs.
// The distributions on top (uniform double, ziggurat normal, Lemire bounded integers) follow
// numpy/random/src/distributions/distributions.c so both generators share one code path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "np_ziggurat_tables.inc"

namespace mdpp {

static __device__ const uint64_t d_zig_ki[256] = NPZ_KI_INIT;
static __device__ const double d_zig_wi[256] = NPZ_WI_INIT;
static __device__ const double d_zig_fi[256] = NPZ_FI_INIT;

struct Pcg64 {
    uint64_t s_lo, s_hi, inc_lo, inc_hi;

    __device__ __forceinline__ void load(const ulonglong2 *st, const ulonglong2 *inc, long i) {
        ulonglong2 a = st[i], b = inc[i];
        s_lo = a.x; s_hi = a.y; inc_lo = b.x; inc_hi = b.y;
    }
    __device__ __forceinline__ void store(ulonglong2 *st, long i) const {
        st[i] = make_ulonglong2(s_lo, s_hi);
    }
    // state = state * 0x2360ED051FC65DA44385DF649FCCF645 + inc (mod 2^128); output XSL-RR of the new state
    __device__ __forceinline__ uint64_t next64() {
        const uint64_t M_HI = 0x2360ED051FC65DA4ULL, M_LO = 0x4385DF649FCCF645ULL;
        uint64_t lo = s_lo * M_LO;
        uint64_t hi = __umul64hi(s_lo, M_LO) + s_lo * M_HI + s_hi * M_LO;
        uint64_t nlo = lo + inc_lo;
        uint64_t carry = nlo < lo ? 1ULL : 0ULL;
        s_lo = nlo;
        s_hi = hi + inc_hi + carry;
        uint64_t x = s_hi ^ s_lo;
        unsigned rot = (unsigned)(s_hi >> 58);
        return (x >> rot) | (x << ((64u - rot) & 63u));
    }
};

// numpy's pcg64_next32 buffers the high half of a 64-bit draw (has_uint32 / uinteger).
struct Half32 {
    uint32_t has32, u32;
};
template <class G>
__device__ __forceinline__ uint32_t next32(G &g, Half32 &h) {
    if (h.has32) { h.has32 = 0; return h.u32; }
    uint64_t n = g.next64();
    h.has32 = 1; h.u32 = (uint32_t)(n >> 32);
    return (uint32_t)n;
}

struct Philox {
    uint32_t k0, k1;         // key = seed
    uint32_t c0, c1, c2, c3; // counter = (env id lo/hi, tick, stream | block)
    uint32_t spare_lo, spare_hi;
    uint32_t have_spare;

    __device__ __forceinline__ void init(uint64_t seed, uint64_t env, uint32_t tick, uint32_t stream) {
        k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32);
        c0 = (uint32_t)env; c1 = (uint32_t)(env >> 32); c2 = tick; c3 = stream << 24;
        have_spare = 0; spare_lo = spare_hi = 0;
    }
    __device__ __forceinline__ uint64_t next64() {
        if (have_spare) { have_spare = 0; return ((uint64_t)spare_hi << 32) | spare_lo; }
        uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = c3, a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; r++) {
            uint32_t h0 = __umulhi(0xD2511F53u, x0), l0 = 0xD2511F53u * x0;
            uint32_t h1 = __umulhi(0xCD9E8D57u, x2), l1 = 0xCD9E8D57u * x2;
            uint32_t y0 = h1 ^ x1 ^ a, y1 = l1, y2 = h0 ^ x3 ^ b, y3 = l0;
            x0 = y0; x1 = y1; x2 = y2; x3 = y3;
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
        c3 += 1; // next block of this (env, tick, stream)
        spare_lo = x2; spare_hi = x3; have_spare = 1;
        return ((uint64_t)x1 << 32) | x0;
    }
};

template <class G>
__device__ __forceinline__ double np_random(G &g) { // Generator.random()
    return (double)(g.next64() >> 11) * (1.0 / 9007199254740992.0);
}

// random_standard_normal: 256-layer ziggurat
template <class G>
__device__ __forceinline__ double np_standard_normal(G &g) {
    const double nor_r = 3.6541528853610087963519472518;
    const double nor_inv_r = 0.27366123732975827203338247596;
    for (;;) {
        uint64_t r = g.next64();
        int idx = (int)(r & 0xff);
        r >>= 8;
        int sign = (int)(r & 0x1);
        uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
        double x = (double)rabs * d_zig_wi[idx];
        if (sign) x = -x;
        if (rabs < d_zig_ki[idx]) return x;
        if (idx == 0) {
            for (;;) {
                double xx = -nor_inv_r * log1p(-np_random(g));
                double yy = -log1p(-np_random(g));
                if (yy + yy > xx * xx)
                    return ((rabs >> 8) & 0x1) ? -(nor_r + xx) : nor_r + xx;
            }
        } else {
            if (((d_zig_fi[idx - 1] - d_zig_fi[idx]) * np_random(g) + d_zig_fi[idx]) <
                exp(-0.5 * x * x))
                return x;
        }
    }
}

// The same loop entered with the first 64-bit draw `r` already made and already known to have
// failed the fast accept (callers inline the 98.8 % case and come here for the wedge / tail).
template <class G>
__device__ __forceinline__ double np_standard_normal_resume(G &g, uint64_t r) {
    const double nor_r = 3.6541528853610087963519472518;
    const double nor_inv_r = 0.27366123732975827203338247596;
    for (;;) {
        int idx = (int)(r & 0xff);
        r >>= 8;
        int sign = (int)(r & 0x1);
        uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
        double x = (double)rabs * d_zig_wi[idx];
        if (sign) x = -x;
        if (rabs < d_zig_ki[idx]) return x;
        if (idx == 0) {
            for (;;) {
                double xx = -nor_inv_r * log1p(-np_random(g));
                double yy = -log1p(-np_random(g));
                if (yy + yy > xx * xx)
                    return ((rabs >> 8) & 0x1) ? -(nor_r + xx) : nor_r + xx;
            }
        } else {
            if (((d_zig_fi[idx - 1] - d_zig_fi[idx]) * np_random(g) + d_zig_fi[idx]) <
                exp(-0.5 * x * x))
                return x;
        }
        r = g.next64();
    }
}

// Ziggurat tables staged in LDS (6 KiB): with one wavefront per SIMD a table lookup in global
// memory costs a full L2 round trip per normal.
struct ZigLds {
    const uint64_t *ki;
    const double *wi, *fi;
};
__device__ __forceinline__ void zig_stage(uint64_t *ki, double *wi, double *fi, int tid, int nthreads) {
    for (int k = tid; k < 256; k += nthreads) { ki[k] = d_zig_ki[k]; wi[k] = d_zig_wi[k]; fi[k] = d_zig_fi[k]; }
}

// Tail of the ziggurat (layer 0, |x| > 3.654): ~0.03 % of draws.  Deliberately NOT a real
// function call: a generator passed by reference to an out-of-line function must live in
// memory, and then every draw of the whole kernel goes through scratch (= HBM latency).
template <class G>
__device__ __forceinline__ double np_zig_tail(G &g, uint64_t rabs) {
    const double nor_r = 3.6541528853610087963519472518;
    const double nor_inv_r = 0.27366123732975827203338247596;
    for (;;) {
        double xx = -nor_inv_r * log1p(-np_random(g));
        double yy = -log1p(-np_random(g));
        if (yy + yy > xx * xx)
            return ((rabs >> 8) & 0x1) ? -(nor_r + xx) : nor_r + xx;
    }
}

// random_standard_normal: numpy's loop verbatim with the tables in LDS.  Measured at one wave per
// SIMD (tools/bench_rng.hip, profiles/r01_rng_microbench.txt) this plain form (412 ns per draw per
// wave) beats both a wave-uniform restructuring of the wedge path and a chord/tangent pre-test
// that avoids exp() (530-750 ns): the rejection branches are short and rarely re-entered.
template <class G>
__device__ __forceinline__ double np_standard_normal_lds(G &g, const ZigLds &z) {
    for (;;) {
        uint64_t r = g.next64();
        int idx = (int)(r & 0xff);
        r >>= 8;
        int sign = (int)(r & 0x1);
        uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
        double x = (double)rabs * z.wi[idx];
        if (sign) x = -x;
        if (rabs < z.ki[idx]) return x;
        if (idx == 0) return np_zig_tail(g, rabs);
        if (((z.fi[idx - 1] - z.fi[idx]) * np_random(g) + z.fi[idx]) < exp(-0.5 * x * x)) return x;
    }
}

// Generator.integers(low, high) for ranges that fit 32 bits: buffered_bounded_lemire_uint32
template <class G>
__device__ __forceinline__ int np_integers(G &g, Half32 &h, int low, int high) {
    uint32_t rng = (uint32_t)(high - 1 - low);
    if (rng == 0) return low;
    uint32_t rng_excl = rng + 1;
    uint64_t m = (uint64_t)next32(g, h) * rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
        uint32_t threshold = (0xFFFFFFFFu - rng) % rng_excl;
        while (leftover < threshold) {
            m = (uint64_t)next32(g, h) * rng_excl;
            leftover = (uint32_t)m;
        }
    }
    return low + (int)(m >> 32);
}

// searchsorted(cdf, u, side='right') for a short normalised cdf
__device__ __forceinline__ int searchsorted_right(const double *cdf, int n, double u) {
    int c = 0;
    for (int i = 0; i < n; i++) c += (cdf[i] <= u) ? 1 : 0;
    return c;
}

} // namespace mdpp

// Device-side random streams for the RLToyEnv kernels (gfx950).
//
// Two bit generators behind one interface (next64()):
//   * Pcg64  — numpy's PCG64 (pcg_setseq_128_xsl_rr_64) with per-env state in HBM, so that
//              an env seeded like the reference draws the very same variates as
//              np.random.Generator does in rl_toy_env.py (reset :2255, noise :403/:413,
//              DiscreteExtended.sample spaces/discrete_extended.py:17).
//   * Philox — stateless Philox4x32-10 keyed by (seed, global env id, tick, stream): no RNG
//              bytes in HBM, results independent of how envs are sharded over GPUs; its own
//              (Box-Muller) Gaussians, see below.
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

// ---- Philox mode (the build's own counter-based streams; NOT in the reference) -------------------
// A stream is keyed by (seed, GLOBAL env id, 64-bit tick, stream id) and is a sequence of 32-bit words:
// block b of the stream = Philox4x32-10 (Salmon et al., SC'11) of counter (env lo, env hi, tick lo,
// stream << 24 | b) under key (seed lo, seed hi ^ tick hi); next64() hands out words (0,1) then (2,3).
// Uniform doubles, bounded integers and the categorical search sit on next64() exactly like numpy's
// (shared code below).  GAUSSIANS are this mode's own: a float32 Box-Muller pair per 64-bit draw
// (philox_box_muller: the form GPU libraries use for float normals), fixed consumption, no rejection
// loop, no tables -- every lane of a wave does the same work.  The pair's second normal is kept for
// the stream's next normal draw.  The transform uses only IEEE-exact operations (conversions, fma,
// multiply, add, correctly rounded sqrtf, bit operations), so the oracle's C restatement
// (np_philox_box_muller in the test oracle) produces the same bits.

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one 32 x 32 -> 64 product each (v_mad_u64_u32: one quarter-rate instruction where
        // __umulhi + * are two)
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        const uint32_t y0 = h1 ^ c1 ^ k0, y1 = l1, y2 = h0 ^ c3 ^ k1, y3 = l0;
        c0 = y0; c1 = y1; c2 = y2; c3 = y3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// Two independent N(0, 1) float32 variates from two 32-bit words (Box-Muller):
//   r = sqrt(-2 ln u1), u1 = max(w0, 1/2) * 2^-32 in (0, 1];  theta = 2 pi (w1 * 2^-32)
//   z0 = r cos(theta), z1 = r sin(theta).
// ln on [sqrt(1/2), sqrt(2)) and sin / cos on [0, pi/4] are the Cephes single-precision polynomials
// (Moshier, logf.c / sinf.c), evaluated with fma; relative error of z about 2^-22.  |z| <= 6.8.
__device__ __forceinline__ void philox_box_muller(uint32_t w0, uint32_t w1, float &z0, float &z1) {
    float f = (float)w0;                                    // round to nearest
    if (w0 == 0u) f = 0.5f;
    const uint32_t b = __float_as_uint(f);
    int e = (int)(b >> 23) - 127;                           // f = m 2^e, m in [1, 2)
    float m = __uint_as_float((b & 0x7FFFFFu) | 0x3F800000u);
    if (m > 1.41421356f) { m *= 0.5f; e += 1; }
    const float x = m - 1.0f;
    const float xx = x * x;
    float y = 7.0376836292E-2f;
    y = fmaf(y, x, -1.1514610310E-1f);
    y = fmaf(y, x, 1.1676998740E-1f);
    y = fmaf(y, x, -1.2420140846E-1f);
    y = fmaf(y, x, 1.4249322787E-1f);
    y = fmaf(y, x, -1.6668057665E-1f);
    y = fmaf(y, x, 2.0000714765E-1f);
    y = fmaf(y, x, -2.4999993993E-1f);
    y = fmaf(y, x, 3.3333331174E-1f);
    y = (y * x) * xx;
    y = fmaf(-0.5f, xx, y);
    const float fe = (float)(e - 32);
    float l = fmaf(fe, 0.693359375f, fmaf(fe, -2.12194440e-4f, x + y));   // ln u1
    l = fminf(l, 0.0f);
    const float r = sqrtf(-2.0f * l);          // correctly rounded (hipcc default; __fsqrt_rn is the native approximation)
    const uint32_t q = w1 >> 30;
    const float a = (float)(w1 & 0x3FFFFFFFu) * (1.0f / 1073741824.0f);   // angle within the quadrant / (pi/2), [0, 1]
    const bool swap = a > 0.5f;
    const float t = (swap ? 1.0f - a : a) * 1.57079632679489662f;        // [0, pi/4]
    const float tt = t * t;
    float sn = fmaf(-1.9515295891E-4f, tt, 8.3321608736E-3f);
    sn = fmaf(sn, tt, -1.6666654611E-1f);
    sn = fmaf(sn * tt, t, t);
    float cs = fmaf(2.443315711809948E-5f, tt, -1.388731625493765E-3f);
    cs = fmaf(cs, tt, 4.166664568298827E-2f);
    cs = fmaf(cs * tt, tt, fmaf(-0.5f, tt, 1.0f));
    const float s1 = swap ? cs : sn, c1 = swap ? sn : cs;                // sin, cos of the in-quadrant angle
    const float cq = (q == 0u) ? c1 : (q == 1u) ? -s1 : (q == 2u) ? -c1 : s1;
    const float sq = (q == 0u) ? s1 : (q == 1u) ? c1 : (q == 2u) ? -s1 : -c1;
    z0 = r * cq;
    z1 = r * sq;
}

struct Philox {
    uint32_t k0, k1;         // key
    uint32_t c0, c1, c2, c3; // counter = (env id lo/hi, tick lo, stream << 24 | block)
    uint32_t spare_lo, spare_hi;
    uint32_t have_spare;
    float z_spare;           // second normal of the last Box-Muller pair
    uint32_t have_z;

    __device__ __forceinline__ void init(uint64_t seed, uint64_t env, uint64_t tick, uint32_t stream) {
        k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(tick >> 32);
        c0 = (uint32_t)env; c1 = (uint32_t)(env >> 32); c2 = (uint32_t)tick; c3 = stream << 24;
        have_spare = 0; spare_lo = spare_hi = 0;
        have_z = 0; z_spare = 0.0f;
    }
    __device__ __forceinline__ uint64_t next64() {
        if (have_spare) { have_spare = 0; return ((uint64_t)spare_hi << 32) | spare_lo; }
        uint32_t o[4];
        philox4x32_10(c0, c1, c2, c3, k0, k1, o);
        c3 += 1; // next block of this (env, tick, stream)
        spare_lo = o[2]; spare_hi = o[3]; have_spare = 1;
        return ((uint64_t)o[1] << 32) | o[0];
    }
    __device__ __forceinline__ double normal() {
        if (have_z) { have_z = 0; return (double)z_spare; }
        const uint64_t r = next64();
        float z0;
        philox_box_muller((uint32_t)r, (uint32_t)(r >> 32), z0, z_spare);
        have_z = 1;
        return (double)z0;
    }
};

// The first NN standard normals of Philox stream (seed, env, tick, stream), all at once: the same values
// Philox::normal() hands out one by one (pair p = words (0,1) / (2,3) of block p / 2), but as
// independent straight-line work -- ceil(NN / 4) blocks, ceil(NN / 2) Box-Muller pairs -- for the fused
// rollout kernels, where a step's draw count is known in advance.
template <int NN>
__device__ __forceinline__ void philox_normals(uint64_t seed, uint64_t env, uint64_t tick, uint32_t stream, float (&z)[NN]) {
    constexpr int NPAIR = (NN + 1) / 2, NBLK = (NPAIR + 1) / 2;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(tick >> 32);
#pragma unroll
    for (int b = 0; b < NBLK; b++) {
        uint32_t o[4];
        philox4x32_10((uint32_t)env, (uint32_t)(env >> 32), (uint32_t)tick, (stream << 24) + (uint32_t)b, k0, k1, o);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int pr = 2 * b + h;
            if (pr < NPAIR) {
                float z0, z1;
                philox_box_muller(o[2 * h], o[2 * h + 1], z0, z1);
                z[2 * pr] = z0;
                if (2 * pr + 1 < NN) z[2 * pr + 1] = z1;
            }
        }
    }
}

template <class G>
__device__ __forceinline__ double np_random(G &g) { // Generator.random()
    return (double)(g.next64() >> 11) * (1.0 / 9007199254740992.0);
}

// random_standard_normal: 256-layer ziggurat (numpy streams); Philox streams: their own Box-Muller normal
__device__ __forceinline__ double np_standard_normal(Philox &g) { return g.normal(); }
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
__device__ __forceinline__ double np_standard_normal_lds(Philox &g, const ZigLds &) { return g.normal(); }
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

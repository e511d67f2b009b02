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

// The same step for a lane that does nothing else (the generator wave of k_continuous_rollout_fast<..., WALK>): the
// state as four 32-bit limbs, the 128-bit product as six 32 x 32 -> 64 multiply-adds chained through their 64-bit
// addends plus four low products -- 23 vector instructions and 8 for the output permutation, where the compiler's
// expansion of the 64-bit arithmetic above takes 38 + 8 (it rebuilds every {x, 0} addend pair with moves and spends a
// multiply-add on each).  Temporaries are fixed registers (an {x, 0} pair needs two consecutive registers of which only the
// low one is rewritten, which a constraint cannot say; the body is mdpp_pcg64_limbs.inc, included per register range);
// the roles that use it keep few registers live.
#define MDPP_LIMBS_NAME Pcg64Limbs       /* kernels of up to 768 threads (168 registers per lane) */
#define MDPP_R_Z0 "150"
#define MDPP_R_Z1 "151"
#define MDPP_R_P0 "152"
#define MDPP_R_P1 "153"
#define MDPP_R_Q0 "154"
#define MDPP_R_Q1 "155"
#define MDPP_R_T0 "156"
#define MDPP_R_T1 "157"
#include "mdpp_pcg64_limbs.inc"
#undef MDPP_LIMBS_NAME
#undef MDPP_R_Z0
#undef MDPP_R_Z1
#undef MDPP_R_P0
#undef MDPP_R_P1
#undef MDPP_R_Q0
#undef MDPP_R_Q1
#undef MDPP_R_T0
#undef MDPP_R_T1
#define MDPP_LIMBS_NAME Pcg64LimbsLo     /* the 1024-thread role-split kernels (128 registers per lane) */
#define MDPP_R_Z0 "112"
#define MDPP_R_Z1 "113"
#define MDPP_R_P0 "114"
#define MDPP_R_P1 "115"
#define MDPP_R_Q0 "116"
#define MDPP_R_Q1 "117"
#define MDPP_R_T0 "118"
#define MDPP_R_T1 "119"
#include "mdpp_pcg64_limbs.inc"
#undef MDPP_LIMBS_NAME
#undef MDPP_R_Z0
#undef MDPP_R_Z1
#undef MDPP_R_P0
#undef MDPP_R_P1
#undef MDPP_R_Q0
#undef MDPP_R_Q1
#undef MDPP_R_T0
#undef MDPP_R_T1

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
        // (three-input xor: one v_bitop3_b32, truth table 0x96, where the compiler emits two v_xor_b32)
        const uint32_t y0 = __builtin_amdgcn_bitop3_b32(h1, c1, k0, 0x96), y1 = l1;
        const uint32_t y2 = __builtin_amdgcn_bitop3_b32(h0, c3, k1, 0x96), y3 = l0;
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
// Correctly rounded float32 square root of x in [0, 2^7) that is zero or a normal number (here: -2 ln u1 <= 44.4, and
// never denormal: its smallest non-zero value is -2 ln(1 - 2^-32) = 4.7e-10): v_sqrt_f32 (<= 1 ulp) stepped one ulp down /
// up where the exact fma residual says so -- the compiler's own sequence for sqrtf without its denormal scaling and
// special-case selects (6 instructions fewer per root, 7 roots per cfg5 step).  Same bits as IEEE sqrtf on that range.
__device__ __forceinline__ float philox_sqrtf(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = fmaf(-sd, s, x), ru = fmaf(-su, s, x);
    float r = (rd <= 0.0f) ? sd : s;
    r = (ru > 0.0f) ? su : r;
    return r;
}

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
    const float r = philox_sqrtf(-2.0f * l);   // correctly rounded
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
    // sin, cos of the in-quadrant angle are (sn, cs), exchanged when the angle was reflected at pi/4; quadrant q
    // rotates (cos, sin) -> (-sin, cos) q times: an odd q exchanges them once more, cos is negative for q = 1, 2
    // and sin for q = 2, 3 (bit 31 of w1 + 2^30, and of w1).  Two selects and two sign-bit flips.
    const bool xsw = swap != ((q & 1u) != 0u);
    const float cq = xsw ? sn : cs, sq = xsw ? cs : sn;
    z0 = __uint_as_float(__float_as_uint(r * cq) ^ ((w1 + 0x40000000u) & 0x80000000u));
    z1 = __uint_as_float(__float_as_uint(r * sq) ^ (w1 & 0x80000000u));
}

// Two Box-Muller pairs at once from the four words of one Philox block: the same operations per pair as
// philox_box_muller -- the same bits -- with the polynomial evaluations as packed float32 instructions
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes per instruction, each an IEEE operation).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void philox_box_muller2(const uint32_t (&w)[4], float &z0, float &z1, float &z2, float &z3) {
    f32x2 f = {(float)w[0], (float)w[2]};
    if (w[0] == 0u) f.x = 0.5f;
    if (w[2] == 0u) f.y = 0.5f;
    const uint32_t b0 = __float_as_uint(f.x), b1 = __float_as_uint(f.y);
    int e0 = (int)(b0 >> 23) - 127, e1 = (int)(b1 >> 23) - 127;
    f32x2 m = {__uint_as_float((b0 & 0x7FFFFFu) | 0x3F800000u), __uint_as_float((b1 & 0x7FFFFFu) | 0x3F800000u)};
    const bool big0 = m.x > 1.41421356f, big1 = m.y > 1.41421356f;
    const f32x2 mh = m * 0.5f;
    m.x = big0 ? mh.x : m.x; m.y = big1 ? mh.y : m.y;
    e0 += big0 ? 1 : 0; e1 += big1 ? 1 : 0;
    const f32x2 x = m - 1.0f;
    const f32x2 xx = x * x;
    auto K = [](float c) -> f32x2 { return f32x2{c, c}; };
    f32x2 y = K(7.0376836292E-2f);
    y = __builtin_elementwise_fma(y, x, K(-1.1514610310E-1f));
    y = __builtin_elementwise_fma(y, x, K(1.1676998740E-1f));
    y = __builtin_elementwise_fma(y, x, K(-1.2420140846E-1f));
    y = __builtin_elementwise_fma(y, x, K(1.4249322787E-1f));
    y = __builtin_elementwise_fma(y, x, K(-1.6668057665E-1f));
    y = __builtin_elementwise_fma(y, x, K(2.0000714765E-1f));
    y = __builtin_elementwise_fma(y, x, K(-2.4999993993E-1f));
    y = __builtin_elementwise_fma(y, x, K(3.3333331174E-1f));
    y = (y * x) * xx;
    y = __builtin_elementwise_fma(K(-0.5f), xx, y);
    const f32x2 fe = {(float)(e0 - 32), (float)(e1 - 32)};
    f32x2 l = __builtin_elementwise_fma(fe, K(0.693359375f), __builtin_elementwise_fma(fe, K(-2.12194440e-4f), x + y));
    l.x = fminf(l.x, 0.0f); l.y = fminf(l.y, 0.0f);
    const f32x2 l2 = l * -2.0f;
    const f32x2 r = {philox_sqrtf(l2.x), philox_sqrtf(l2.y)};
    f32x2 a = {(float)(w[1] & 0x3FFFFFFFu), (float)(w[3] & 0x3FFFFFFFu)};
    a = a * (1.0f / 1073741824.0f);
    const bool swap0 = a.x > 0.5f, swap1 = a.y > 0.5f;
    const f32x2 ar = 1.0f - a;
    f32x2 t = {swap0 ? ar.x : a.x, swap1 ? ar.y : a.y};
    t = t * 1.57079632679489662f;
    const f32x2 tt = t * t;
    f32x2 sn = __builtin_elementwise_fma(K(-1.9515295891E-4f), tt, K(8.3321608736E-3f));
    sn = __builtin_elementwise_fma(sn, tt, K(-1.6666654611E-1f));
    sn = __builtin_elementwise_fma(sn * tt, t, t);
    f32x2 cs = __builtin_elementwise_fma(K(2.443315711809948E-5f), tt, K(-1.388731625493765E-3f));
    cs = __builtin_elementwise_fma(cs, tt, K(4.166664568298827E-2f));
    cs = __builtin_elementwise_fma(cs * tt, tt, __builtin_elementwise_fma(K(-0.5f), tt, K(1.0f)));
    const bool x0 = swap0 != (((w[1] >> 30) & 1u) != 0u), x1 = swap1 != (((w[3] >> 30) & 1u) != 0u);
    const f32x2 cq = {x0 ? sn.x : cs.x, x1 ? sn.y : cs.y}, sq = {x0 ? cs.x : sn.x, x1 ? cs.y : sn.y};
    const f32x2 pc = r * cq, ps = r * sq;
    z0 = __uint_as_float(__float_as_uint(pc.x) ^ ((w[1] + 0x40000000u) & 0x80000000u));
    z1 = __uint_as_float(__float_as_uint(ps.x) ^ (w[1] & 0x80000000u));
    z2 = __uint_as_float(__float_as_uint(pc.y) ^ ((w[3] + 0x40000000u) & 0x80000000u));
    z3 = __uint_as_float(__float_as_uint(ps.y) ^ (w[3] & 0x80000000u));
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
        if (2 * b + 1 < NPAIR) {                 // both pairs of the block: packed
            float zz[4];
            philox_box_muller2(o, zz[0], zz[1], zz[2], zz[3]);
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (4 * b + q < NN) z[4 * b + q] = zz[q];
        } else {
            float z0, z1;
            philox_box_muller(o[0], o[1], z0, z1);
            z[4 * b] = z0;
            if (4 * b + 1 < NN) z[4 * b + 1] = z1;
        }
    }
}

// ---- Philox mode, discrete envs: the start state of an in-rollout reset ---------------------------------
// (same-step autoreset: the step at tick t ended the episode; next-step autoreset: call t IS the reset.)
// A reset needs ONE categorical draw, so it gets one 32-bit word, and a block serves four ticks:
//   w(t) = word (t & 3) of block 0 of stream (seed, env, t >> 2, stream id)      [stream ids: mdpp_internal.hpp]
//   s0   = searchsorted(cdf, (w >> 1) * 2^-31, 'right') = #{ j : ceil(cdf[j] 2^31) <= (w >> 1) }
// (round 2 spent a whole block per env and tick on it -- first 64 bits of the env stream's block -- although only
//  a quarter of the ticks reset: cfg2 ran at half the rate of the numpy streams.)  An explicit reset() keeps its
// own stream keyed by the reset count (kPhiloxResetStream, 53-bit uniform).
__device__ __forceinline__ void philox_start_block(uint64_t seed, uint64_t env, uint64_t blk, uint32_t stream,
                                                   uint32_t (&o)[4]) {
    philox4x32_10((uint32_t)env, (uint32_t)(env >> 32), (uint32_t)blk, stream << 24, (uint32_t)seed,
                  (uint32_t)(seed >> 32) ^ (uint32_t)(blk >> 32), o);
}
__device__ __forceinline__ uint32_t philox_start_m31(uint64_t seed, uint64_t env, uint64_t tick, uint32_t stream) {
    uint32_t o[4];
    philox_start_block(seed, env, tick >> 2, stream, o);
    const uint32_t q = (uint32_t)tick & 3u;
    const uint32_t w = q == 0u ? o[0] : q == 1u ? o[1] : q == 2u ? o[2] : o[3];
    return w >> 1;
}
__device__ __forceinline__ double philox_start_uniform(uint32_t m31) { return (double)m31 * (1.0 / 2147483648.0); }

// ---- Philox mode, discrete envs: transition noise and reward noise, one word per tick each ----------------
// (round 3, second re-key: a block of the env stream per env and tick -- a 53-bit uniform searched in the S-entry
//  categorical's cdf in 64-bit arithmetic, and a Box-Muller pair of which one normal was used -- made cfg2 + noise
//  VALU-bound at 0.24 of the HBM roofline.)  Like the start state, each draw of a tick is ONE word of a block
// that serves four ticks, w(t, id) = word (t & 3) of block 0 of stream (seed, env, t >> 2, id):
//   transition noise (:1604-1622: the table's next state n keeps mass 1 - p, every other state gets p / (S - 1)):
//     T = ceil(p 2^32) (at most 2^32 - 1);  w(t, 12) >= T: the step is not noisy;  otherwise the re-drawn state is the
//     j-th of the S - 1 OTHER states in ascending order, j = floor(w (S - 1) / T)   -- the same distribution
//     from one word, without a cdf; the irrelevant sub-space the same with stream id 4 and its own S;
//   reward noise (:1980-1984): the four float32 Box-Muller normals of block (t >> 2) of stream id 13
//     (pairs (w0, w1) and (w2, w3), philox_box_muller2), normal t & 3 is tick t's.
__device__ __forceinline__ uint32_t philox_word_of(const uint32_t (&o)[4], uint64_t tick) {
    const uint32_t q = (uint32_t)tick & 3u;
    return q == 0u ? o[0] : q == 1u ? o[1] : q == 2u ? o[2] : o[3];
}
struct PhiloxTickWords {          // the block of the current four ticks, kept while a K-step loop stays inside it
    uint64_t blk = ~0ULL;
    uint32_t o[4] = {0u, 0u, 0u, 0u};
    __device__ __forceinline__ uint32_t word(uint64_t seed, uint64_t env, uint64_t tick, uint32_t stream) {
        if ((tick >> 2) != blk) {              // (wave-uniform: every lane is at the same tick)
            blk = tick >> 2;
            philox_start_block(seed, env, blk, stream, o);
        }
        return philox_word_of(o, tick);
    }
};
struct PhiloxTickNormals {
    uint64_t blk = ~0ULL;
    float z[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    __device__ __forceinline__ float normal(uint64_t seed, uint64_t env, uint64_t tick, uint32_t stream) {
        if ((tick >> 2) != blk) {
            blk = tick >> 2;
            uint32_t o[4];
            philox_start_block(seed, env, blk, stream, o);
            philox_box_muller2(o, z[0], z[1], z[2], z[3]);
        }
        const uint32_t q = (uint32_t)tick & 3u;
        return q == 0u ? z[0] : q == 1u ? z[1] : q == 2u ? z[2] : z[3];
    }
};
__host__ __device__ inline uint32_t philox_pnoise_threshold(double p) {
    const double t = ceil(p * 4294967296.0);
    return t >= 4294967295.0 ? 4294967295u : (t <= 0.0 ? 0u : (uint32_t)t);
}
// floor(w (S - 1) / T) for every w < T as the top word of a 32 x 64-bit product: M = ceil(2^64 (S - 1) / T) (division
// by an invariant with 64 fraction bits: the product is too large by less than 2^-32, and a non-integer w (S - 1) / T is
// at least 1 / T > 2^-32 below the next integer).  0 = no noise (S < 2, or T <= S - 1: p < 6e-8).
inline uint64_t philox_pnoise_magic(uint32_t T, uint32_t S) {
    if (S < 2u || T <= S - 1u) return 0ull;
    const unsigned __int128 num = (unsigned __int128)(S - 1u) << 64;
    return (uint64_t)((num + T - 1u) / T);
}
// j | noisy << 8: `noisy` = the tick's word w is below T, j = index of the re-drawn state among the S - 1 others
__device__ __forceinline__ uint32_t philox_pnoise_index(uint32_t w, uint32_t T, uint64_t M) {
    const uint64_t t = (uint64_t)w * (uint32_t)(M >> 32) + (uint64_t)__umulhi(w, (uint32_t)M);
    const uint32_t j = (uint32_t)(t >> 32);
    return (w < T && M != 0ull) ? (j | 0x100u) : 0u;
}
// the state a step lands in: `nxt` (the table's) unless the tick's word says noisy
// (the index whole: state spaces beyond 256 states -- mdpp_discrete_wide.hip -- have indices beyond the byte philox_pnoise_index packs)
__device__ __forceinline__ uint32_t philox_pnoise_state(uint32_t w, uint32_t T, uint64_t M, uint32_t nxt) {
    const uint64_t t = (uint64_t)w * (uint32_t)(M >> 32) + (uint64_t)__umulhi(w, (uint32_t)M);
    const uint32_t j = (uint32_t)(t >> 32);
    return (w < T && M != 0ull) ? j + (j >= nxt ? 1u : 0u) : nxt;
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
// SIMD (tools/bench_rng.hip, profiles/archive/r01_rng_microbench.txt) this plain form (412 ns per draw per
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

// The same draw for a sigma of 0 (round 6): the reference's rng.normal(0, 0) is 0.0 + 0.0 z = +0.0 for every z, so only the
// words the draw CONSUMES matter -- the fast path is the integer compare alone; the wedge needs |x| (squared), the tail its loop.
__device__ __forceinline__ void np_skip_normal_lds(Philox &, const ZigLds &) {}
template <class G>
__device__ __forceinline__ void np_skip_normal_lds(G &g, const ZigLds &z) {
    for (;;) {
        const uint64_t r = g.next64();
        const int idx = (int)(r & 0xff);
        const uint64_t rabs = (r >> 9) & 0x000fffffffffffffULL;
        if (rabs < z.ki[idx]) return;
        if (idx == 0) { (void)np_zig_tail(g, rabs); return; }
        const double x = (double)rabs * z.wi[idx];
        if (((z.fi[idx - 1] - z.fi[idx]) * np_random(g) + z.fi[idx]) < exp(-0.5 * x * x)) return;
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

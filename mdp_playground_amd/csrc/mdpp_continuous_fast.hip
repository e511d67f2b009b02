// Fused K-step rollout for the common continuous shape (BASELINE cfg 3 / cfg 5): move_to_a_point,
// relevant dims = the first n_rel dims, bounded box, no terminal hypercubes, delay 0,
// reward_every_n_steps 1, numpy PCG64 streams.  Same arithmetic as k_continuous_step
// (mdpp_continuous.hip; reference rl_toy_env.py:1630-1725, :1912-1990, :2102-2109) with D and the
// dynamics order as template constants, so every per-dimension loop is straight-line code on
// registers, and with the work that the general kernel does per step moved out of the way:
//   * actions ([env][D] float32, 48 B per lane at D = 12) are fetched kCAhead steps ahead with
//     16-byte buffer loads; observations leave as 16-byte buffer stores;
//   * divisions by inertia and by k! become exact multiplications when the divisor is a power of
//     two (1, 2: orders 1-2, the usual inertia), true IEEE divisions otherwise;
//   * ||s_old - target|| of step k is ||s_new - target|| of step k-1, carried in a register;
//   * the action-norm penalty (action_loss_weight, default 0) is only evaluated when it can
//     matter; the box clip + derivative reset runs under a wave-uniform branch;
//   * Gaussian noise: the ziggurat's hot path (98.8 % of draws: one PCG64 output, one table
//     compare) is inline with ki/wi staged in LDS; the wedge/tail path is out of line;
//   * the rare events — episode end (target reached / truncated) with its reset() draws from the
//     feature-space stream, a rejected action — sit behind wave-uniform unlikely branches.
// HBM traffic per env step: 4D B action in; 4D B obs + 4 B reward + 2 B flags out (102 B at D = 12).
// Compile with -ffp-contract=off.
#include <stdlib.h>

#include <type_traits>

#include "mdpp_internal.hpp"
#include "mdpp_rng.hpp"

namespace mdpp {

#ifndef MDPP_CAHEAD
#define MDPP_CAHEAD 4
#endif
#ifndef MDPP_CAHEAD_SMALL_D
#define MDPP_CAHEAD_SMALL_D 4      // noisy rollouts at D <= 4: action rows in flight (Philox streams; numpy streams twice as many)
#endif
#ifndef MDPP_CONT_ROWS
#define MDPP_CONT_ROWS 1           // rewards and flags leave as whole rows of the workgroup (see "whole-row stores" in the kernel)
#endif
#ifndef MDPP_CBUFS
#define MDPP_CBUFS 1               // noise-free rollouts: action rows in flight = MDPP_CBUFS x MDPP_CAHEAD (buffers rotated by NAME: a
#endif                             // single loop over 8 or more rows is past the compiler's unroll budget and lands in scratch)
constexpr int kCAheadQuiet = MDPP_CAHEAD, kCAheadNoise = 1; // noisy steps take microseconds: one row ahead hides the load, and a
                                                            // deeper ring would not unroll (step body too large) -> scratch
constexpr int kCRsrc = 0x00020000;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float c_fdiv_or_mul(float x, float div, float inv, bool pow2) {
    return pow2 ? x * inv : x / div;
}

// HELPER (noise only): 512-thread workgroups; waves 4-7 own the envs' PCG64 streams for the launch
// and produce the D (+1) standard normals of every step into an LDS ring, waves 0-3 integrate and
// read them back in the reference's draw order.  The numpy-exact ziggurat is ~3/4 of a noisy
// step's instructions (profiles/archive/r01_rng_microbench.txt) and one wave per SIMD leaves issue slots
// idle, so running it beside the integrator nearly halves the step time.  Producer and consumer
// count the same K * (D + 1) draws, so nothing is drawn ahead of what the reference would draw.
// The producer lanes of a wave are not in lockstep: see "park" below.
constexpr int kNRing = 4;                      // steps of normals buffered per env
#ifndef MDPP_NP_BATCH
#define MDPP_NP_BATCH 1
#endif
#ifndef MDPP_NP_PARK
#define MDPP_NP_PARK 3             // (3-6 measure within 1 %, 8 and 12 stall the leading lanes at the ring limit)
#endif
#ifndef MDPP_NP_ATTEMPTS
#define MDPP_NP_ATTEMPTS 8         // fast attempts per round of bookkeeping: 2 / 4 / 6 / 8 / 10 / 13 -> 2 745 / 2 420 / 2 304 / 2 272 / 2 293 / 2 480 us per cfg5 launch
#endif
#ifndef MDPP_NP_PRODUCER_PRIO
#define MDPP_NP_PRODUCER_PRIO 3    // numpy streams: the helper wave (13 numpy-exact normals per step) is the long stage: 2 737 -> 2 410 us per cfg5 launch
#endif
#ifndef MDPP_NP_CONSUMER_PRIO
#define MDPP_NP_CONSUMER_PRIO 0
#endif
#ifndef MDPP_PRODUCER_PRIO
#define MDPP_PRODUCER_PRIO 0
#endif
#ifndef MDPP_CONSUMER_PRIO
#define MDPP_CONSUMER_PRIO 2
#endif
#ifndef MDPP_PHILOX_PRODUCERS
#define MDPP_PHILOX_PRODUCERS 2
#endif
#ifndef MDPP_WK_ATTEMPTS
#define MDPP_WK_ATTEMPTS 7         // walker: fast ziggurat attempts per round (4 / 5 / 6 / 7 / 8 / 10 / 12 / 16 -> 1 604 / 1 465 / 1 479 / 1 443 / 1 506 / 1 520 / 1 570 / 1 709 us)
#endif
#ifndef MDPP_WK_GEN_BATCH
#define MDPP_WK_GEN_BATCH 4        // generator: words per window check (divides kWRing)
#endif
#ifndef MDPP_WK_PARK
#define MDPP_WK_PARK 1             // walker: parked lanes that trigger a wedge / tail pass.  1 / 2 / 4 / 6 / 8 / 12 / 16 -> 1 518 / 1 554 / 1 648 /
#endif                             // 1 714 / 1 869 / 2 196 / 2 457 us per cfg5 launch: a waiting lane holds the consumer's step back, the pass is cheap
#ifndef MDPP_WK_GEN_PRIO
#define MDPP_WK_GEN_PRIO 1         // (gen 1, walker 3, consumer 2: 1 475 us; all 0: 1 528; consumer first: 1 522 -- profiles/r04_ablation_walk.txt)
#endif
#ifndef MDPP_WK_WALKER_PRIO
#define MDPP_WK_WALKER_PRIO 3
#endif
#ifndef MDPP_WK_CONSUMER_PRIO
#define MDPP_WK_CONSUMER_PRIO 2
#endif
constexpr int kWRing = 32;         // WALK: raw words / normals buffered per env (powers of two: ring index = count & 31)
constexpr int kPhiloxProducers = MDPP_PHILOX_PRODUCERS;   // producer waves per consumer wave, Philox streams (see NPROD)
constexpr uint32_t kCSpinLimit = 1u << 22;
constexpr uint32_t kCStatusInternal = 0x80000000u;

// GEN: the general reward post-processing and episode ends -- delay line (C7, :1968-1973), every-n
// mask (:1975), terminal hypercubes (C8, :945-952; reset() resamples out of them, :2284-2307), and
// unbounded state boxes (reset() samples normals).  With
// these the reference's reward changes between np.float32 and Python-float arithmetic from step to
// step (k_continuous_step's CRew); without them it is float32 throughout, and that path stays as lean
// as it was.
// PHILOX: counter-based streams (mdpp_rng.hpp): a step's normals are a pure function of (seed, global env
// id, tick), made as straight-line work -- by the producer waves (HELPER) or at the top of the step --
// with no tables, no rejection loop and no stream state in HBM; reset() keys its own stream per step.
// NPROD (PHILOX + HELPER): producer waves per consumer wave.  A counter-based stream has no serial
// state, so step k's normals can be made by ANY wave: producer p makes the steps k = p (mod NPROD), and
// two or three producers fill the SIMD's issue slots that one dependent Philox / Box-Muller chain leaves idle.
// WALK (numpy streams + HELPER, NPROD == 2): three roles on three waves per SIMD instead of producer + consumer.  The helper
// wave of rounds 1-3 walked ONE sequential PCG64 stream per lane AND decided the ziggurat's data-dependent consumption in
// the same dependent chain (1 300 vector + 830 scalar instructions per step, profiles/archive/r02_cfg5_sq.txt).  The two halves do
// not depend on each other the way that code made them:
//   * the stream's 64-bit WORDS are a function of the position alone -- a generator wave (waves 8-11) makes them in order,
//     branch-free, into a per-lane LDS ring indexed by stream position (window of kWRing words ahead of the walker);
//   * the WALKER (waves 4-7) reads words at its lane's position, runs batches of fast ziggurat attempts on them (the
//     accepted prefix goes to the normals ring, the first rejected word parks the lane; wedge / tail passes run for all
//     parked lanes at once and take their uniforms from the same ring), and never touches generator state -- so a
//     rejection no longer needs saved states to rewind to;
//   * the consumer (waves 0-3) reads a step's normals at the top of the step and frees the slots at once.
// Per lane the stream is consumed in exactly numpy's order; the generator un-draws the words the walker did not take at
// the end of the launch (inverse LCG steps), so the stored stream state is the reference's.
// K1 (round 5): the launch of ONE step (mdpp_step) -- k_continuous_step1 to its callers.  A rollout amortises its prologue
// over hundreds of steps; a launch of one step IS its prologue, and at 65 536 envs it moves 20-27 MB, so it is paid in bytes
// and in load round trips: the action row is the first load issued (not the fourth: no rows are fetched ahead), the highest
// derivative row (overwritten by a / inertia before anything reads it) and the irrelevant coordinates of the last
// observation are only read by a wave that holds a rejected action ("stay", :1671-1679), no staging of rewards and flags
// for later groups, and without ziggurat tables the workgroup is ONE wave (no barrier; all 1 024 SIMDs start at once).
// PAR (K1 with transition noise on numpy streams): the step's D + 1 ziggurat normals come from ONE sequential PCG64 stream per
// env -- thirteen dependent draws in one lane were 7 of the launch's 14 us (cfg5).  The stream's WORDS are a function of
// the position alone (an LCG jumps ahead: s_k = M^k s + (M^(k-1) + ... + 1) inc), and a draw's fast path is a function of its
// word alone, so: 64 envs per 256-thread workgroup; wave w makes the words at positions 4w .. 4w + 3 of every env's stream
// (one jump, three steps), evaluates each as if a draw started there -- accepted at once / wedge accepted or rejected with the
// NEXT position's word as its uniform / anything else -- and posts {word, value, 2-bit kind} in LDS; wave 0 then walks the
// kinds from position 0 like numpy walks the stream (13 draws, a few bit operations each), reads the values, takes the
// generator state behind the last word it consumed (posted by the wave that made that word), and integrates.  Tails,
// three rejected wedges in a row and draws that reach past position 15 (6 in 10 000 env steps) run numpy's own loop on a
// generator that reads the posted words and continues sequentially behind them -- exact, just not parallel.
template <int D, int ORDER, int NREL, bool NOISE, bool HELPER, bool GEN, bool PHILOX = false, int NPROD = 1, bool K1 = false, bool PAR = false, bool Z0T = false>
__global__ __launch_bounds__(HELPER ? (1 + NPROD) * kBlock : kBlock, PAR ? 4 : 1) void k_continuous_rollout_fast(ContinuousArgs a, int K,
                                                                    const float *__restrict__ actions,
                                                                    float *__restrict__ obs,
                                                                    float *__restrict__ reward,
                                                                    uint8_t *__restrict__ term,
                                                                    uint8_t *__restrict__ trunc,
                                                                    float *__restrict__ final_obs) {
    const uint64_t ptick0 = tick_now(a);               // the step counter at this launch (through the device-side offset of a graph replay)
    const uint32_t rhead0 = ring_head_now(a, ptick0);    // ... and the head of a delay line kept in memory
    static_assert(D % 4 == 0 || D == 2, "D must be 2 or a multiple of 4");
    constexpr bool ZIG = NOISE && !PHILOX;      // numpy's ziggurat tables in LDS
    constexpr bool WALK = HELPER && !PHILOX && NPROD == 2;
    __shared__ uint64_t s_ki[ZIG && !WALK ? 256 : 1];
    __shared__ double s_wi[ZIG && !WALK ? 256 : 1], s_fi[ZIG ? 256 : 1];
    __shared__ ulonglong2 s_kw[WALK ? 256 : 1];                 // WALK: {ki, wi * 2^52} side by side (one 16-byte lookup per attempt)
    constexpr int NPS = D + 1;                  // normals per step slot (D transition + 1 reward)
    static_assert(NPROD == 1 || HELPER, "several helper waves need HELPER");
    static_assert(NPROD == 1 || PHILOX || NPROD == 2, "numpy streams: generator + walker");
    typedef typename std::conditional<PHILOX, float, double>::type ZT;       // (Philox normals are float32 values)
    // (WALK: both rings are 64 KiB and 64 KiB-aligned -- a slot's byte address is then ONE v_and_or_b32 of the count, the ring mask
    //  and a per-lane constant)
    __shared__ __attribute__((aligned(WALK ? 65536 : 16))) ZT s_z[HELPER ? (WALK ? kWRing : kNRing * NPS) * kBlock : (PAR ? NPS * 64 : 1)];       // [slot][draw][lane]; WALK: [normal count & 31][lane]
    __shared__ __attribute__((aligned(WALK ? 65536 : 16))) uint64_t s_raw[WALK ? kWRing * kBlock : 1];       // WALK: [stream position & 31][lane]
    __shared__ uint32_t s_gp[WALK ? kBlock : 1], s_rp[WALK ? kBlock : 1], s_done[kBlock / 64];   // words made / taken per lane
    __shared__ uint32_t s_prod[NPROD][kBlock / 64], s_cons[kBlock / 64];     // steps made by producer p / steps consumed
    __shared__ __align__(16) float s_tr[((D > 4 && !PAR) ? kBlock / 64 : 1) * 64 * (D > 4 ? D : 4)];   // output transpose tiles
    // Whole-row stores of the small outputs (round 4).  A lane's reward is 4 bytes and its flags one byte each: stored per lane,
    // a wave instruction writes 256 / 64 / 64 bytes -- 6 of the step's 102 bytes that took 8-19 % of a cfg3 launch (573-630 us
    // without them against 685-706).  Without helper waves: the rewards and flags of a GROUP of four steps are staged in LDS
    // (three buffers), and wave w writes the block's whole piece of row k0 + w of a group -- 1 KiB of rewards (16 B per lane),
    // 256 B of each flag array (4 B per lane) -- one group LATER, when the other three waves have long staged theirs: per-wave
    // counters (groups staged / groups stored), no barrier (a barrier per group took back most of the gain: 662 -> 641 us).
    // (whole-row stores pay where a step is long: cfg3, D = 12, 610 -> 595 us; at D = 2 the row hand-over between the workgroup's
    //  waves costs more than the stores it saves -- 320 against 281 us per launch, profiles/r05_ablation_d2.txt)
    constexpr bool CROWS = MDPP_CONT_ROWS && !HELPER && !K1 && D >= 8;
    static_assert(!K1 || !HELPER, "one step: no helper waves");
    static_assert(!PAR || (K1 && NOISE && !PHILOX), "parallel draws: one step, numpy streams");
    constexpr int kPP = 16;                     // PAR: stream positions evaluated side by side (4 per wave)
    static_assert(!PAR || D + 1 <= kPP, "a step's draws must fit the positions");
    __shared__ uint64_t p_w[PAR ? kPP * 64 : 1];                    // PAR: the stream's words [position][env]
    __shared__ double p_v[PAR ? kPP * 64 : 1];                      // ... the normal a draw starting there returns (kinds 0, 1)
    __shared__ ulonglong2 p_s[PAR ? (kPP - D + 1) * 64 : 1];        // ... the generator state behind word p, p >= D - 1
    __shared__ uint32_t p_k[PAR ? 64 : 1];                          // ... kinds: byte w = wave w's four positions, 2 bits each
    constexpr int WG = (K1 && !(NOISE && !PHILOX)) ? 64 : kBlock;      // threads per workgroup of the env waves
    constexpr int kRS = kBlock / 64;            // steps per group = waves per workgroup
    constexpr int kRBufs = 3;
    __shared__ __align__(16) float s_rw[CROWS ? kRBufs * kRS * kBlock : 4];
    __shared__ __align__(16) uint8_t s_tw[CROWS ? kRBufs * kRS * kBlock : 4], s_uw[CROWS ? kRBufs * kRS * kBlock : 4];
    __shared__ __align__(16) uint32_t s_rprod[kBlock / 64], s_rcons[kBlock / 64];
    if (CROWS && threadIdx.x < kBlock / 64) { s_rprod[threadIdx.x] = 0; s_rcons[threadIdx.x] = 0; }
    if (CROWS) __syncthreads();
    const int tid = threadIdx.x;
    // (K1: the tables' loads are the launch's first, their LDS stores and the barrier come after every other load has been issued)
    uint64_t k1_ki = 0; double k1_wi = 0.0, k1_fi = 0.0;
    if constexpr (K1 && ZIG) { k1_ki = d_zig_ki[tid]; k1_wi = d_zig_wi[tid]; k1_fi = d_zig_fi[tid]; }
    if (ZIG && !WALK && !K1) zig_stage(s_ki, s_wi, s_fi, tid, HELPER ? 2 * kBlock : kBlock);
    if (WALK) {
        for (int k = tid; k < 256; k += 3 * kBlock) { s_kw[k] = make_ulonglong2(d_zig_ki[k], __double_as_longlong(d_zig_wi[k]) + (52LL << 52));   /* {ki, W = wi * 2^52 (exact)} */ s_fi[k] = d_zig_fi[k]; }
        if (tid < kBlock) { s_gp[tid] = 0; s_rp[tid] = 0; }
    }
    if (HELPER && tid < kBlock / 64) {
#pragma unroll
        for (int p = 0; p < NPROD; p++) s_prod[p][tid] = 0;
        s_cons[tid] = 0;
        s_done[tid] = 0;
    }
    if (NOISE && !K1) __syncthreads();
    const ZigLds zig{s_ki, s_wi, s_fi};
    // (PAR: one env wave per workgroup -- "lane" and "wave" of the consumer code below are those of that wave, whichever it is)
    const int ln = PAR ? (tid & 63) : (tid & (kBlock - 1)), wv = PAR ? 0 : (ln >> 6);
    // Workgroup b runs on XCD b % 8 (round-robin dispatch): in a rollout every XCD steps one contiguous eighth of the envs, so that what
    // its L2 writes back per output row is one contiguous range (MDPP_C_XCD; as in k_discrete_rollout_lean)
#ifndef MDPP_C_XCD
#define MDPP_C_XCD 0             /* (measured: cfg3 0.659-0.663 -> 0.649-0.661, cfg5 and c_d2_n0 unchanged: off) */
#endif
    const uint32_t bxc = (MDPP_C_XCD && !K1 && (gridDim.x & 7u) == 0u) ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const uint32_t i = PAR ? blockIdx.x * 64u + (uint32_t)(tid & 63) : bxc * WG + ln;
    if (i >= (uint32_t)a.N) return;             // HELPER launches require N % kBlock == 0
    const uint32_t N = (uint32_t)a.N;
    const uint64_t genv = (uint64_t)(a.env_id_offset + (int64_t)i);     // global env id (Philox key)
    // Z0 (WALK; round 6): every noise key that is present has sigma 0 -- what the reference's continuous experiment files pass
    // (transition_noise: 0, reward_noise: 0), and the reference still draws rng.normal(0, 0, ...) per step (rl_toy_env.py:
    // 398-403, :1682-1691, :1980-1987).  numpy forms loc + scale * z = 0.0 + 0.0 z = +0.0 whatever z is (0.0 + -0.0 = +0.0), so
    // the noise term IS +0.0: only the stream's advance -- how many words each ziggurat draw consumes -- is left of the draws.
    // The walker then makes the accept decisions alone (no rabs * wi, no sign, no normals ring), and the consumer neither waits
    // for the walker nor reads a normal: normal() returns 0.0 and 0.0 + sigma * 0.0 is the same +0.0.
    // A template flag, chosen by the launcher and instantiated for D = 2 only (the shape of the reference's continuous
    // experiments): the D = 12 walker kernels sit at 168 registers with 104 spilled and are left as they were.
    constexpr bool kZ0 = Z0T;
    static_assert(!Z0T || (WALK && D == 2), "sigma-0 walker: the D = 2 generator / walker / consumer kernels");
    // this step's normals, PHILOX: P-noise of dimension d at [d], reward noise at [D]
    auto philox_step = [&](int k, float (&z)[NPS]) __attribute__((always_inline)) {
        const uint64_t tick = ptick0 + (uint64_t)k;
        if (a.has_p_noise && a.has_r_noise) {
            philox_normals<NPS>(a.philox_seed, genv, tick, MDPP_STREAM_ENV, z);
        } else if (a.has_p_noise) {
            float zz[D];
            philox_normals<D>(a.philox_seed, genv, tick, MDPP_STREAM_ENV, zz);
#pragma unroll
            for (int d = 0; d < D; d++) z[d] = zz[d];
            z[D] = 0.0f;
        } else {
            float zz[1];
            philox_normals<1>(a.philox_seed, genv, tick, MDPP_STREAM_ENV, zz);
#pragma unroll
            for (int d = 0; d < D; d++) z[d] = 0.0f;
            z[D] = zz[0];
        }
    };
    if (HELPER && PHILOX && tid >= kBlock) {
        // ---------------- producer lane, Philox: no stream state, every lane of the wave does the same work
        uint32_t hstatus = 0;
#ifdef MDPP_ABL_NOPROD
        return;
#endif
        const int me = tid / kBlock - 1;        // producer index: this wave makes the steps k = me (mod NPROD)
        if (NPROD > 1) __builtin_amdgcn_s_setprio(MDPP_PRODUCER_PRIO);
        uint32_t made = 0;
        for (int k = me; k < K; k += NPROD) {
            if (k >= kNRing) {                  // wait until the consumer wave freed slot k % kNRing
                uint32_t spins = 0;
                while (__hip_atomic_load(&s_cons[wv], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <
                       (uint32_t)(k - kNRing + 1)) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > kCSpinLimit) { hstatus |= kCStatusInternal; break; }
                }
            }
            // one Philox block -> two Box-Muller pairs -> four ring entries at a time (few live registers)
            ZT *slot = s_z + (size_t)(k % kNRing) * NPS * kBlock + ln;
            const uint64_t tick = ptick0 + (uint64_t)k;
            const uint32_t k0 = (uint32_t)a.philox_seed, k1 = (uint32_t)(a.philox_seed >> 32) ^ (uint32_t)(tick >> 32);
            const int first = a.has_p_noise ? 0 : D;            // ring index of the stream's first normal
            const int nn = (a.has_p_noise ? D : 0) + (a.has_r_noise ? 1 : 0);
#pragma unroll
            for (int b = 0; b < (NPS + 3) / 4; b++) {
                if (4 * b < nn) {
                    uint32_t o[4];
                    philox4x32_10((uint32_t)genv, (uint32_t)(genv >> 32), (uint32_t)tick, (MDPP_STREAM_ENV << 24) + (uint32_t)b,
                                  k0, k1, o);
                    float zz[4];
                    // (both pairs of a block as packed float32 work; nn is wave-uniform)
                    if (4 * b + 2 < nn) philox_box_muller2(o, zz[0], zz[1], zz[2], zz[3]);
                    else philox_box_muller(o[0], o[1], zz[0], zz[1]);
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (4 * b + q < nn) slot[(first + 4 * b + q) * kBlock] = (ZT)zz[q];
                }
            }
            made += 1;
            if ((ln & 63) == 0)
                __hip_atomic_store(&s_prod[me][wv], made, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (hstatus) atomicOr(&a.status[i], hstatus);
        return;
    }
    if constexpr (WALK) {
        // ring slots as BYTE offsets: slot (c & 31) of this lane is at ((c << 11) & 0xf800) + lane * 8
        const uint32_t lane8 = (uint32_t)ln * 8u;
        constexpr uint32_t SM = (uint32_t)(kWRing - 1) << 11;
        char *rawb = (char *)s_raw;
        typedef __attribute__((address_space(3))) const uint64_t *lds_cu64p;
        typedef __attribute__((address_space(3))) double *lds_f64p;
        // (integer LDS addresses: ring base | lane * 8 in one register; the bases have their low 16 bits clear)
        const uint32_t raw0 = (uint32_t)(uintptr_t)(lds_cu64p)s_raw | lane8, z0 = (uint32_t)(uintptr_t)(lds_f64p)s_z | lane8;
        auto raw_at = [&](uint32_t c11) __attribute__((always_inline)) -> uint64_t {          // c11 = count << 11
            return *(lds_cu64p)(uintptr_t)((c11 & SM) | raw0);
        };
        // ... as two 32-bit halves (a 64-bit value invites the compiler to shift it as one: v_lshrrev_b64 for every field)
        typedef unsigned int wk_u32x2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) const wk_u32x2 *lds_cu2p;
        auto raw_at2 = [&](uint32_t c11) __attribute__((always_inline)) -> wk_u32x2 { return *(lds_cu2p)(uintptr_t)((c11 & SM) | raw0); };
        if (tid >= 2 * kBlock) {
            // ---------------- generator lane: the words of the env's noise stream, in order, by position -----------------
            // Word p of the launch goes to s_raw[p & 31][lane]; a batch is made when it fits the lane's window (the walker
            // has taken s_rp[lane] words: positions below s_rp + kWRing are free).  No data-dependent control flow.
#ifdef MDPP_ABL_WK_NOGEN
            return;
#endif
            Pcg64 hg;
            hg.load(a.env_s, a.env_inc, i);
            Pcg64Limbs lg;
            lg.from(hg);
            __builtin_amdgcn_s_setprio(MDPP_WK_GEN_PRIO);
            constexpr uint32_t UG = MDPP_WK_GEN_BATCH;
            static_assert(kWRing % MDPP_WK_GEN_BATCH == 0, "a batch must not wrap");
            uint32_t p = 0, spins = 0, hstatus = 0;
            for (;;) {
                const uint32_t rp = __hip_atomic_load(&s_rp[ln], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (__hip_atomic_load(&s_done[wv], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) break;
                const bool go = p + UG <= rp + (uint32_t)kWRing;
                if (__builtin_amdgcn_ballot_w64(go) != 0) {
                    if (go) {
                        uint64_t *dst = (uint64_t *)(rawb + (((p << 11) & SM) | lane8));
#pragma unroll
                        for (uint32_t u = 0; u < UG; u++) dst[u * kBlock] = lg.next64();
                        p += UG;
                    }
                    __hip_atomic_store(&s_gp[ln], p, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    spins = 0;
                } else {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > kCSpinLimit) { hstatus |= kCStatusInternal; break; }
                }
            }
            // un-draw what the walker did not take: s_prev = (s - inc) * M^-1 (mod 2^128)
            lg.to(hg);
            const uint32_t taken = __hip_atomic_load(&s_rp[ln], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint64_t MINV_HI = 0x07dda22b93979860ULL, MINV_LO = 0x98abc8b0716eac8dULL;
            for (uint32_t q = p - taken; q > 0; q--) {
                const uint64_t lo = hg.s_lo - hg.inc_lo;
                const uint64_t hi = hg.s_hi - hg.inc_hi - (hg.s_lo < hg.inc_lo ? 1ULL : 0ULL);
                hg.s_lo = lo * MINV_LO;
                hg.s_hi = __umul64hi(lo, MINV_LO) + lo * MINV_HI + hi * MINV_LO;
            }
            hg.store(a.env_s, i);
            if (hstatus) atomicOr(&a.status[i], hstatus);
            return;
        }
        if (tid >= kBlock) {
#ifdef MDPP_ABL_WK_NOWALK
            return;
#endif
            // ---------------- walker lane: numpy's ziggurat over the words of the ring, in numpy's order -----------------
            // One round = up to NB fast attempts on the next NB words (98.8 % of attempts succeed: one table lookup, one
            // compare).  The accepted prefix goes to the normals ring; the first rejected word parks the lane with that
            // word.  When kPark lanes wait (or nothing else can be done) the wedge / tail path runs once for all of them
            // and takes its uniforms from the ring like numpy takes them from the stream.  Lanes drift apart by what
            // their rejections consumed; both rings are per lane, so nobody waits for anybody's position.
            // The wedge test `y < exp(-x^2 / 2)` is decided by a float32 estimate of the right-hand side (v_exp_f32,
            // relative error < 2e-6) wherever y is further than 1e-5 (relative) from it -- all but about one wedge point in
            // 10^5 --; the others take the float64 exp() as before, so every decision is the one exp() gives.  (exp() for
            // every pass was most of the 590 us that the passes cost a 1 860 us cfg5 launch; a variant that handed the
            // passes to the generator wave made lanes wait longer than the 2.5 steps of normals the ring holds: 2 740 us.)
            __builtin_amdgcn_s_setprio(MDPP_WK_WALKER_PRIO);
            constexpr int NB = MDPP_WK_ATTEMPTS;
            constexpr uint32_t kPark = MDPP_WK_PARK;
            const uint32_t nd = (a.has_p_noise ? (uint32_t)D : 0u) + (a.has_r_noise ? 1u : 0u);
            const uint32_t total = (uint32_t)K * nd;
            uint32_t n = 0, rp = 0, pub = 0, published = 0, thr = nd, spins = 0, hstatus = 0;
            bool parked = false;
            uint64_t pr = 0;
            auto z_put = [&](uint32_t c11, double v) __attribute__((always_inline)) { *(lds_f64p)(uintptr_t)((c11 & SM) | z0) = v; };
            for (;;) {
                const uint32_t cons = __hip_atomic_load(&s_cons[wv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef MDPP_ABL_WK_NOCONS
                const uint32_t nlim = total + (cons & 1u);
#else
                const uint32_t nlim = kZ0 ? total : min(total, cons * nd + (uint32_t)kWRing);       // (Z0: no normals ring to respect)
#endif
                const uint32_t gp = __hip_atomic_load(&s_gp[ln], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                bool progress = false;
                const bool can = !parked && n < nlim && rp < gp;
                const uint64_t bcan = __builtin_amdgcn_ballot_w64(can);
                if (bcan != 0) {
                    const uint32_t allowed = can ? min(min(nlim - n, gp - rp), (uint32_t)NB) : 0u;
                    wk_u32x2 wd[NB];
#pragma unroll
                    for (int u = 0; u < NB; u++) wd[u] = raw_at2((rp << 11) + ((uint32_t)u << 11));
                    uint32_t bad = 1u << allowed;
                    if constexpr (kZ0) {                   // (compile-time) the decisions alone: one 8-byte lookup, one 52-bit compare per word
                        uint64_t kq[NB];
#pragma unroll
                        for (int u = 0; u < NB; u++) kq[u] = s_kw[wd[u].x & 0xffu].x;
#pragma unroll
                        for (int u = 0; u < NB; u++) {
                            const uint32_t wlo = wd[u].x, whi = wd[u].y;
                            const uint32_t rlo = __builtin_amdgcn_alignbit(whi, wlo, 9), rhi = (whi >> 9) & 0xFFFFFu;
                            const uint64_t rabs = (uint64_t)rlo | ((uint64_t)rhi << 32);
                            bad |= (rabs < kq[u]) ? 0u : (1u << u);
                        }
                        const uint32_t m = (uint32_t)__builtin_ctz(bad);
                        const bool rej = m < allowed;
                        if (__builtin_amdgcn_ballot_w64(rej) != 0) {
                            if (rej) pr = raw_at((rp + m) << 11);
                        }
                        n += m;
                        rp += m + (rej ? 1u : 0u);
                        parked = parked || rej;
                        progress = can;
                    } else {
                    ulonglong2 kw[NB];
#pragma unroll
                    for (int u = 0; u < NB; u++) kw[u] = s_kw[wd[u].x & 0xffu];
                    double xs[NB];
#pragma unroll
                    for (int u = 0; u < NB; u++) {
                        // rabs = (r >> 9) & (2^52 - 1) in two 32-bit operations (not a 64-bit shift and a mask)
                        const uint32_t wlo = wd[u].x, whi = wd[u].y;
                        const uint32_t rlo = __builtin_amdgcn_alignbit(whi, wlo, 9), rhi = (whi >> 9) & 0xFFFFFu;
                        const uint64_t rabs = (uint64_t)rlo | ((uint64_t)rhi << 32);
                        // rabs * wi in ONE rounding: m = 1 + rabs 2^-52 (rabs as the mantissa of a number in [1, 2)), W = wi 2^52:
                        // fma(m, W, -W) = round((m - 1) W) = round(rabs wi) -- what numpy's `rabs * wi_double[idx]` rounds to
                        const double mm = __longlong_as_double((long long)(rabs | 0x3FF0000000000000ULL));
                        const double W = __longlong_as_double((long long)kw[u].y);
                        const double x = __builtin_fma(mm, W, -W);
                        // sign = bit 8 of the word -> bit 63 (x >= +0: numpy's `if (sign & 0x1) x = -x`)
                        const uint64_t xb = (uint64_t)__double_as_longlong(x);
                        xs[u] = __longlong_as_double((long long)(((uint64_t)(((uint32_t)(xb >> 32) & 0x7FFFFFFFu) | ((wlo << 23) & 0x80000000u)) << 32) | (uint32_t)xb));   // (v_bfi_b32)
#ifndef MDPP_ABL_WK_NOSLOW
                        bad |= (rabs < kw[u].x) ? 0u : (1u << u);
#else
                        bad |= (rabs < kw[u].x + 0x7fffffffffffffffULL) ? 0u : (1u << u);
#endif
                    }
                    const uint32_t m = (uint32_t)__builtin_ctz(bad);        // accepted prefix (<= allowed)
                    const bool rej = m < allowed;
#pragma unroll
                    for (int u = 0; u < NB; u++)
                        if ((uint32_t)u < m) z_put((n << 11) + ((uint32_t)u << 11), xs[u]);
                    if (__builtin_amdgcn_ballot_w64(rej) != 0) {
                        if (rej) pr = raw_at((rp + m) << 11);
                    }
                    n += m;
                    rp += m + (rej ? 1u : 0u);
                    parked = parked || rej;
                    progress = can;
                    }
                }
                const uint64_t bpark = __builtin_amdgcn_ballot_w64(parked);
                if (bpark != 0 && ((uint32_t)__builtin_popcountll(bpark) >= kPark || bcan == 0)) {
                    const uint32_t idx = (uint32_t)pr & 0xffu;
                    const bool tail = idx == 0u;
                    const bool act = parked && rp + (tail ? 2u : 1u) <= gp;     // (the uniforms it takes are in the ring)
                    // wedge lanes first: one uniform; a rejected point starts the draw over with a fresh word
                    if (__builtin_amdgcn_ballot_w64(act && !tail) != 0) {
                        bool sure = true, accf = false;
                        double x = 0.0, y = 0.0;
                        if (act && !tail) {
                            const uint64_t rabs = (pr >> 9) & 0x000fffffffffffffULL;
                            const double W = __longlong_as_double((long long)s_kw[idx].y);          // (as in the fast attempts)
                            x = __builtin_fma(__longlong_as_double((long long)(rabs | 0x3FF0000000000000ULL)), W, -W);
                            x = ((uint32_t)pr & 0x100u) ? -x : x;
                            const double u1 = (double)(raw_at(rp << 11) >> 11) * (1.0 / 9007199254740992.0);
                            rp += 1u;
                            y = (s_fi[idx - 1] - s_fi[idx]) * u1 + s_fi[idx];
                            // exp(-x^2 / 2) = 2^t, t = -x^2 log2(e) / 2 in (-9.7, 0]: t rounded to float32 is off by < 5e-7
                            // (6.6e-7 relative in the result), v_exp_f32 by one ulp
                            const float e32 = __builtin_amdgcn_exp2f((float)(x * x * -0.72134752044448170368));
                            const double e = (double)e32;
                            accf = y < e;
                            sure = fabs(y - e) > 1.0e-5 * e;
                        }
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!sure) != 0, 0)) {
                            if (!sure) accf = y < exp(-0.5 * x * x);
                        }
                        if (act && !tail) {
                            if (accf) { if (!kZ0) z_put(n << 11, x); n += 1u; }
                            parked = false;
                            progress = true;
                        }
                    }
                    // tail lanes (layer 0, |x| > 3.654: 3 in 10^4 draws): two uniforms per try, the lane stays parked until one is accepted
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(act && tail) != 0, 0)) {
                        if (act && tail) {
                            const double nor_r = 3.6541528853610087963519472518, nor_inv_r = 0.27366123732975827203338247596;
                            const double u1 = (double)(raw_at(rp << 11) >> 11) * (1.0 / 9007199254740992.0);
                            const double u2 = (double)(raw_at((rp + 1u) << 11) >> 11) * (1.0 / 9007199254740992.0);
                            rp += 2u;
#ifdef MDPP_ABL_WK_CHEAPTAIL        /* timing only: what the two log1p of a tail try cost the launch */
                            const double xx = u1 * nor_inv_r;
                            const double yy = u2 + 8.0;
#else
                            const double xx = -nor_inv_r * log1p(-u1);
                            const double yy = -log1p(-u2);
#endif
                            if (yy + yy > xx * xx) {
                                if (!kZ0) z_put(n << 11, ((pr >> 17) & 0x1) ? -(nor_r + xx) : nor_r + xx);
                                n += 1u;
                                parked = false;
                            }
                            progress = true;
                        }
                    }
                }
                // what this lane has taken (the generator's window) and the steps every lane has completed
                __hip_atomic_store(&s_rp[ln], rp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (pub < (uint32_t)K && __builtin_amdgcn_ballot_w64(n < thr) == 0) { pub++; thr += nd; }
                if (pub != published) {
                    if ((ln & 63) == 0)
                        __hip_atomic_store(&s_prod[0][wv], pub, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    published = pub;
                }
                if (pub == (uint32_t)K) break;
                if (__builtin_amdgcn_ballot_w64(progress) == 0) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > kCSpinLimit) { hstatus |= kCStatusInternal; break; }
                } else {
                    spins = 0;
                }
            }
            __hip_atomic_store(&s_rp[ln], rp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((ln & 63) == 0) __hip_atomic_store(&s_done[wv], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (hstatus) atomicOr(&a.status[i], hstatus);
            return;
        }
    }
    if (HELPER && !PHILOX && !WALK && tid >= kBlock) {
        // ---------------- producer lane: the env's noise stream for this launch -----------------
        Pcg64 hg;
        hg.load(a.env_s, a.env_inc, i);
        __builtin_amdgcn_s_setprio(MDPP_NP_PRODUCER_PRIO);
        uint32_t hstatus = 0;
        if (a.park) {
            // Lanes of a producer wave are allowed to drift apart by up to kNRing steps.  Every
            // iteration makes ONE fast ziggurat attempt for every lane that is not parked (98.8 % of
            // them succeed: one LDS store, next draw); a lane whose attempt fell outside its layer's
            // rectangle is PARKED with its 64-bit word until kPark lanes wait (or the consumer does),
            // and then the wedge / tail path -- an extra uniform, exp or log1p -- runs once for all of
            // them (checked once per MDPP_NP_ATTEMPTS attempts).  In lockstep, 54 % of a wave's draws have some lane on that path and all 64 pay
            // for it (profiles/archive/r01_rng_microbench.txt).  Per lane the stream is consumed in exactly
            // numpy's order: a parked lane draws nothing until its own slow path has run.
            constexpr uint32_t kPark = MDPP_NP_PARK;
            const uint32_t nd = (a.has_p_noise ? (uint32_t)D : 0u) + (a.has_r_noise ? 1u : 0u);
            uint32_t kl = 0, jl = 0, pub = 0;        // this lane's step / draw within the step; steps published
            uint64_t pr = 0;
            bool parked = false;
            uint32_t cons = 0, spins = 0;
            auto put = [&](double x) __attribute__((always_inline)) {
                const uint32_t d = a.has_p_noise ? jl : (uint32_t)D;
                s_z[((size_t)(kl % (uint32_t)kNRing) * NPS + d) * kBlock + ln] = x;
                jl += 1;
                const bool wrap = jl == nd;
                jl = wrap ? 0u : jl;
                kl += wrap ? 1u : 0u;
            };
            uint32_t published = 0;
            for (;;) {
                cons = __hip_atomic_load(&s_cons[wv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t lim = min((uint32_t)K, cons + (uint32_t)kNRing);
                const uint64_t bcan = __builtin_amdgcn_ballot_w64(!parked && kl < lim);
#if MDPP_NP_BATCH
                // One round = up to MDPP_NP_ATTEMPTS fast attempts per lane, made as a batch: the words of the next states first
                // (they do not depend on what the attempts decide), then all table lookups at once (one LDS round trip instead of
                // one per attempt), then the accepted prefix is stored; the stream stops behind the first rejected word, which
                // parks the lane.  Per lane the stream is consumed in numpy's order exactly as in the attempt-by-attempt form.
                if (bcan != 0) {
                    constexpr int NB = MDPP_NP_ATTEMPTS;
                    uint32_t allowed = 0;
                    if (!parked && kl < lim) { const uint32_t rem = (lim - kl) * nd - jl; allowed = rem < (uint32_t)NB ? rem : (uint32_t)NB; }
                    uint64_t wd[NB], slo[NB], shi[NB];
                    Pcg64 t = hg;
#pragma unroll
                    for (int u = 0; u < NB; u++) { wd[u] = t.next64(); slo[u] = t.s_lo; shi[u] = t.s_hi; }
                    double xs[NB];
                    bool ok[NB];
#pragma unroll
                    for (int u = 0; u < NB; u++) {
                        uint64_t r = wd[u];
                        const int idx = (int)(r & 0xff);
                        r >>= 8;
                        const int sign = (int)(r & 0x1);
                        const uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
                        const double x = (double)rabs * zig.wi[idx];
                        xs[u] = sign ? -x : x;
                        ok[u] = rabs < zig.ki[idx];
                    }
                    uint32_t nacc = allowed;
                    bool rej = false;
#pragma unroll
                    for (int u = NB - 1; u >= 0; u--)                     // (descending: the FIRST rejection wins)
                        if ((uint32_t)u < allowed && !ok[u]) { nacc = (uint32_t)u; rej = true; pr = wd[u]; }
                    const uint32_t consumed = rej ? nacc + 1u : allowed;
#pragma unroll
                    for (int u = 0; u < NB; u++) {
                        if ((uint32_t)u < nacc) put(xs[u]);
                        if (consumed == (uint32_t)(u + 1)) { hg.s_lo = slo[u]; hg.s_hi = shi[u]; }
                    }
                    parked = parked || rej;
                }
#else
#pragma unroll
                for (int u = 0; u < MDPP_NP_ATTEMPTS; u++) {   // fast attempts per round of bookkeeping
                    if (!parked && kl < lim) {
                        uint64_t r = hg.next64();
                        const uint64_t r0 = r;
                        const int idx = (int)(r & 0xff);
                        r >>= 8;
                        const int sign = (int)(r & 0x1);
                        const uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
                        double x = (double)rabs * zig.wi[idx];
                        x = sign ? -x : x;
                        if (rabs < zig.ki[idx]) put(x);
                        else { parked = true; pr = r0; }
                    }
                }
#endif
                const uint64_t bpark = __builtin_amdgcn_ballot_w64(parked);
                if (bpark != 0) {
                    // (no "urgent" rule for a parked lane that holds the next step back: the producer is the
                    // slower side, so the consumer is always waiting, and serving such lanes at once is the
                    // lockstep behaviour again; the other lanes use the wait to run up to kNRing steps ahead)
                    if ((uint32_t)__builtin_popcountll(bpark) >= kPark || bcan == 0) {
                        if (parked) {
                            uint64_t r = pr;
                            const int idx = (int)(r & 0xff);
                            r >>= 8;
                            const int sign = (int)(r & 0x1);
                            const uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
                            double x = (double)rabs * zig.wi[idx];
                            x = sign ? -x : x;
                            if (idx == 0) put(np_zig_tail(hg, rabs));
                            else if (((zig.fi[idx - 1] - zig.fi[idx]) * np_random(hg) + zig.fi[idx]) < exp(-0.5 * x * x)) put(x);
                            parked = false;         // (a rejected wedge point: the draw starts over with a fresh word)
                        }
                    }
                }
                // publish the steps every lane has completed
                while (pub < (uint32_t)K && __builtin_amdgcn_ballot_w64(kl <= pub) == 0) pub++;
                if (pub != published) {
                    if ((ln & 63) == 0)
                        __hip_atomic_store(&s_prod[0][wv], pub, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    published = pub;
                }
                if (pub == (uint32_t)K) break;
                if (bcan == 0 && bpark == 0) {       // ring full for every lane: wait for the consumer
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > kCSpinLimit) { hstatus |= kCStatusInternal; break; }
                } else {
                    spins = 0;
                }
            }
            hg.store(a.env_s, i);
            if (hstatus) atomicOr(&a.status[i], hstatus);
            return;
        }
        for (int k = 0; k < K; k++) {
            if (k >= kNRing) {                  // wait until the consumer wave freed slot k % kNRing
                uint32_t spins = 0;
                while (__hip_atomic_load(&s_cons[wv], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <
                       (uint32_t)(k - kNRing + 1)) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > kCSpinLimit) { hstatus |= kCStatusInternal; break; }
                }
            }
            ZT *slot = s_z + (size_t)(k % kNRing) * NPS * kBlock + ln;
            if (a.has_p_noise) {
#pragma unroll
                for (int d = 0; d < D; d++) slot[d * kBlock] = np_standard_normal_lds(hg, zig);
            }
            if (a.has_r_noise) slot[D * kBlock] = np_standard_normal_lds(hg, zig);
            if ((ln & 63) == 0)
                __hip_atomic_store(&s_prod[0][wv], (uint32_t)(k + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        hg.store(a.env_s, i);
        if (hstatus) atomicOr(&a.status[i], hstatus);
        return;
    }
#ifdef MDPP_ABL_WK_NOCONS
    if (WALK) return;
#endif
    // (round 5: a noisy step at D <= 4 takes 0.8 us, less than a load's round trip -- one row ahead left the consumer waiting for its
    //  action every step; its step body is small enough to unroll four times)
    //  (c_d2_n0, us per launch at 1 / 2 / 4 / 8 rows ahead: numpy streams 458 / 426 / 402 / 382, Philox 399 / 365 / 342 / 355)
    constexpr int kCAhead = NOISE ? (D <= 4 ? (PHILOX ? MDPP_CAHEAD_SMALL_D : 2 * MDPP_CAHEAD_SMALL_D) : kCAheadNoise) : kCAheadQuiet;
    constexpr int V = (D == 2) ? 1 : D / 4;      // 16-byte pieces per action / observation row
    // Several producers per consumer: the consumer wave is the critical path (one dependent chain per step, while
    // the producers have a whole step of slack each), so its instructions go first whenever they are ready
    if (HELPER && NPROD > 1) __builtin_amdgcn_s_setprio(MDPP_CONSUMER_PRIO);
    if (HELPER && !PHILOX) __builtin_amdgcn_s_setprio(WALK ? MDPP_WK_CONSUMER_PRIO : MDPP_NP_CONSUMER_PRIO);

    float k1act[K1 ? D : 1];
    // PAR: three of the four waves only make stream words; the fourth also integrates.  WHICH one rotates with the workgroup
    // index: the four workgroups a CU holds are 256 apart (8 XCDs x 32 CUs, round robin), and with wave 0 everywhere the
    // integrating waves of all four sat on ONE SIMD while the other three idled through the second half of the launch.
    const int env_w = PAR ? (int)((blockIdx.x >> 8) & 3u) : 0;
    const bool env_wave = !PAR || (tid >> 6) == env_w;
    if constexpr (K1) if (env_wave) {           // the step's action row first: everything else of the launch waits for it
        if constexpr (D == 2) {
            const float2 v = ((const float2 *)actions)[i];
            k1act[0] = v.x; k1act[1] = v.y;
        } else {
#pragma unroll
            for (int q = 0; q < D / 4; q++) {
                const float4 v = ((const float4 *)actions)[(size_t)i * (D / 4) + q];
                k1act[4 * q] = v.x; k1act[4 * q + 1] = v.y; k1act[4 * q + 2] = v.z; k1act[4 * q + 3] = v.w;
            }
        }
    }
    float sd[ORDER + 1][D], cur[D];
    uint2 meta = make_uint2(0u, 0u);
    if (env_wave) {
#pragma unroll
        for (int k = 0; k <= ORDER; k++)
#pragma unroll
            for (int d = 0; d < D; d++) sd[k][d] = (K1 && k == ORDER) ? 0.0f : a.sd[((size_t)k * D + d) * N + i];
#pragma unroll
        for (int d = 0; d < D; d++) cur[d] = (K1 && d >= NREL) ? 0.0f : a.cur[(size_t)d * N + i];
        meta = a.meta[i];
    }
    Pcg64 g;
    if (ZIG && !HELPER) g.load(a.env_s, a.env_inc, i);
    if constexpr (PAR) {
        // ---- the words at positions 4w .. 4w + 3 (no tables needed yet: their loads are still in flight) ----
        const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
        // s_(4w+1) = A s + G inc (mod 2^128), A = M^(4w+1), G = 1 + M + ... + M^(4w)   [w = 0: A = M, G = 1]
        const uint64_t A_HI[4] = {0x2360ed051fc65da4ULL, 0x16c406e9fbe6c01fULL, 0x18f8f9f7734932a8ULL, 0xd008599933890bf1ULL};
        const uint64_t A_LO[4] = {0x4385df649fccf645ULL, 0x1712dd28ec4e2775ULL, 0x16509219b4cc2da5ULL, 0xfcce321a884738d5ULL};
        const uint64_t G_HI[4] = {0x0ULL, 0x55eb531472e35affULL, 0x85f34a8885b1db52ULL, 0x55f2070f3b269f3cULL};
        const uint64_t G_LO[4] = {0x1ULL, 0x53148145f0c4118dULL, 0xfacc3ec479366459ULL, 0x452c836ab6c04465ULL};
        uint64_t ahi = A_HI[0], alo = A_LO[0], ghi = G_HI[0], glo = G_LO[0];
#pragma unroll
        for (int q = 1; q < 4; q++) { const bool me = w4 == q; ahi = me ? A_HI[q] : ahi; alo = me ? A_LO[q] : alo; ghi = me ? G_HI[q] : ghi; glo = me ? G_LO[q] : glo; }
        auto mul128lo = [](uint64_t xlo, uint64_t xhi, uint64_t ylo, uint64_t yhi, uint64_t &rlo, uint64_t &rhi) __attribute__((always_inline)) {
            rlo = xlo * ylo;
            rhi = __umul64hi(xlo, ylo) + xlo * yhi + xhi * ylo;
        };
        auto xsl_rr = [](uint64_t lo, uint64_t hi) __attribute__((always_inline)) -> uint64_t {
            const uint64_t x = hi ^ lo;
            const unsigned rot = (unsigned)(hi >> 58);
            return (x >> rot) | (x << ((64u - rot) & 63u));
        };
        Pcg64 t = g;
        {
            uint64_t p0, p1, q0, q1;
            mul128lo(g.s_lo, g.s_hi, alo, ahi, p0, p1);
            mul128lo(g.inc_lo, g.inc_hi, glo, ghi, q0, q1);
            t.s_lo = p0 + q0;
            t.s_hi = p1 + q1 + (t.s_lo < p0 ? 1ULL : 0ULL);
        }
        uint64_t wd[4];
        wd[0] = xsl_rr(t.s_lo, t.s_hi);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (q > 0) wd[q] = t.next64();
            const int p = 4 * w4 + q;
            p_w[p * 64 + l] = wd[q];
            if (p >= D - 1) p_s[(p - (D - 1)) * 64 + l] = make_ulonglong2(t.s_lo, t.s_hi);
        }
        s_ki[tid] = k1_ki; s_wi[tid] = k1_wi; s_fi[tid] = k1_fi;
        __syncthreads();
#ifdef MDPP_ABL_PAR_RET1
        return;
#endif
        // ---- every position as the start of a draw: kind 0 accepted at once, 1 wedge accepted, 2 wedge rejected (two words
        // either way), 3 anything else (tail; a wedge whose uniform lies behind the last position) ----
        uint32_t kind[4], idxs[4];
        double xs[4];
        bool pend = false;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint64_t r = wd[q];
            const int idx = (int)(r & 0xff);
            r >>= 8;
            const int sign = (int)(r & 0x1);
            const uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
            double x = (double)rabs * s_wi[idx];
            x = sign ? -x : x;
            const bool ok = rabs < s_ki[idx];
            const int p = 4 * w4 + q;
            xs[q] = x; idxs[q] = (uint32_t)idx;
            kind[q] = ok ? 0u : ((idx == 0 || p + 1 >= kPP) ? 3u : 4u);      // 4: a wedge point, decided below
            pend = pend || kind[q] == 4u;
            p_v[p * 64 + l] = x;
        }
#ifdef MDPP_ABL_PAR_NOWEDGE
        pend = false;
#endif
        for (int pass = 0; pass < 4 && __builtin_amdgcn_ballot_w64(pend) != 0; pass++) {       // (one pass serves a lane's first open wedge point)
            int q = 0;
            double x = 0.0;
            uint32_t idx = 1;
#pragma unroll
            for (int u = 3; u >= 0; u--) { const bool me = kind[u] == 4u; q = me ? u : q; x = me ? xs[u] : x; idx = me ? idxs[u] : idx; }
            bool accf = false, sure = true;
            double y = 0.0;
            if (pend) {
                const uint64_t nw = p_w[(4 * w4 + q + 1) * 64 + l];
                const double u1 = (double)(nw >> 11) * (1.0 / 9007199254740992.0);
                y = (s_fi[idx - 1] - s_fi[idx]) * u1 + s_fi[idx];
                // exp(-x^2 / 2) = 2^t: a float32 estimate decides wherever y is further than 1e-5 (relative) from it, the float64
                // exp() the rest -- every decision is the one exp() gives (the walker of the rollout kernel does the same)
                const double e = (double)__builtin_amdgcn_exp2f((float)(x * x * -0.72134752044448170368));
                accf = y < e;
                sure = fabs(y - e) > 1.0e-5 * e;
            }
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(pend && !sure) != 0, 0)) {
                if (pend && !sure) accf = y < exp(-0.5 * x * x);
            }
            bool more = false;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (pend && u == q) kind[u] = accf ? 1u : 2u;
                more = more || kind[u] == 4u;
            }
            pend = more;
        }
        ((uint8_t *)p_k)[l * 4 + w4] = (uint8_t)(kind[0] | (kind[1] << 2) | (kind[2] << 4) | (kind[3] << 6));
        __syncthreads();
#ifdef MDPP_ABL_PAR_RET2
        return;
#endif
        // ---- ONE wave walks the stream like numpy: draw d starts at position `pos`.  Not the integrating wave: that one holds the
        // env's state in registers since the top of the launch, and numpy's loop (exp, log1p) inlined into its path spilled
        // them to scratch (the walk alone then took 4.6 us of a 13.5 us launch, tools/ablate_step1.py); a wave that only made
        // words has nothing live here.  It also stores the generator state behind the last word consumed. ----
        if (env_wave) __syncthreads();              // (the walker's normals: every wave of the block passes ONE more barrier)
        else if ((tid >> 6) != ((env_w + 1) & 3)) { __syncthreads(); return; }
        else {
        struct PosGen {                            // next64() = the word at the next position: posted, or made behind the last posted one
            uint32_t pos;
            const uint64_t *w;
            Pcg64 tg;
            __device__ __forceinline__ uint64_t next64() {
                uint64_t r = 0;
                if (pos < (uint32_t)kPP) r = w[pos * 64];
                else r = tg.next64();
                pos += 1;
                return r;
            }
        };
        PosGen pg;
        pg.pos = 0; pg.w = p_w + l;
        {
            const ulonglong2 last = p_s[(kPP - 1 - (D - 1)) * 64 + l];     // the state behind word 15
            pg.tg.s_lo = last.x; pg.tg.s_hi = last.y; pg.tg.inc_lo = g.inc_lo; pg.tg.inc_hi = g.inc_hi;
        }
        const uint32_t kinds = p_k[l];
        const int nd = D + (a.has_r_noise ? 1 : 0);
        // A ROLLED loop over the draws (unrolled, its D + 1 copies of numpy's loop -- exp, log1p -- are tens of KB of code that
        // every wave walks past).  Measured (tools/ablate_step1.py, cfg5 at 65 536 envs, replayed graph): words + kinds 4.1 us of
        // the launch, this walk 2.4, the integrator and its stores 4.9 -- and 2.9 for numpy's loop below: a TAIL draw (layer 0
        // beyond |x| = 3.65: two float64 log1p, 2.6 in 10 000 draws) costs about 2 us, a launch of 852 000 draws holds some 220
        // of them, and the launch ends with its slowest wave.  A variant that settled all draws in registers first and sent
        // the lanes with a tail through numpy's loop afterwards measured 14.3 us against this form's 12.7.
#pragma unroll 1
        for (int d = 0; d < NPS; d++) {
            if (d < nd) {
                uint32_t pos = pg.pos;
                uint32_t k = pos < (uint32_t)kPP ? (kinds >> (2u * pos)) & 3u : 3u;
#pragma unroll
                for (int rej = 0; rej < 2; rej++) {                      // up to two rejected wedge points in a row: the draw starts over
                    const bool r2 = k == 2u;
                    pos += r2 ? 2u : 0u;
                    k = r2 ? (pos < (uint32_t)kPP ? (kinds >> (2u * pos)) & 3u : 3u) : k;
                }
                const bool slow = k >= 2u;
                double z = p_v[(pos < (uint32_t)kPP ? pos : 0u) * 64 + l];
                pg.pos = slow ? pos : pos + (k == 0u ? 1u : 2u);
#ifndef MDPP_ABL_PAR_NOGEN
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(slow) != 0, 0)) {
                    if (slow) z = np_standard_normal_lds(pg, zig);      // numpy's own loop on the words from `pos` on
                }
#endif
                s_z[d * 64 + l] = z;            // (read back by the integrating wave where the step uses the normal)
            }
        }
        // the generator behind the last word consumed (pos >= D words were taken)
        if (pg.pos <= (uint32_t)kPP) {
            const ulonglong2 st = p_s[(pg.pos - 1u - (uint32_t)(D - 1)) * 64 + l];
            g.s_lo = st.x; g.s_hi = st.y;
        } else {
            g.s_lo = pg.tg.s_lo; g.s_hi = pg.tg.s_hi;
        }
        g.store(a.env_s, i);
        __syncthreads();
        return;
        }
#ifdef MDPP_ABL_PAR_RET3
        return;
#endif
    } else if constexpr (K1 && ZIG) {           // (launched with N % 256 == 0: every wave of the block reaches the barrier)
        s_ki[tid] = k1_ki; s_wi[tid] = k1_wi; s_fi[tid] = k1_fi;
        __syncthreads();
    }
    uint32_t steps = meta.x, flags = meta.y, status = 0;
    // gymnasium's next-step autoreset (the call after an episode's last step IS the reset: action ignored, reward 0, no
    // flags); "episode ended" travels in bit 1 of the flags word like in k_continuous_step.  Served here without noise or
    // with Philox streams: numpy noise streams are drawn ahead per step, and a reset call must not move them.
    const bool nextmode = a.autoreset == MDPP_AUTORESET_NEXT_STEP;
    bool pend = nextmode && (flags & 2u) != 0u;
    flags &= ~2u;

    const ZT *zslot = s_z + ln;                 // HELPER: this step's normals, set per step
    int zi = 0;
    float zf[NPS];                              // PHILOX without helper waves: this step's normals
    const uint32_t wk_nd = (a.has_p_noise ? (uint32_t)D : 0u) + (a.has_r_noise ? 1u : 0u);
    uint32_t wk_nb = 0;                         // WALK: count of this step's first normal (ring slot = count & 31; wave-uniform)

    const uint32_t total = (uint32_t)K * N;
    auto r_act = __builtin_amdgcn_make_buffer_rsrc((void *)actions, 0, total * (uint32_t)(D * 4), kCRsrc);
    auto r_obs = __builtin_amdgcn_make_buffer_rsrc((void *)obs, 0, total * (uint32_t)(D * 4), kCRsrc);
    auto r_rew = __builtin_amdgcn_make_buffer_rsrc((void *)reward, 0, total * 4u, kCRsrc);
    auto r_term = __builtin_amdgcn_make_buffer_rsrc((void *)term, 0, total, kCRsrc);
    auto r_trunc = __builtin_amdgcn_make_buffer_rsrc((void *)trunc, 0, total, kCRsrc);
    const uint32_t v1 = i, v4 = i * 4u, vrow = i * (uint32_t)(D * 4);
    const bool full_wave = (i - (i & 63u)) + 64u <= N;     // all 64 lanes of this wave own an env
    const bool crows = CROWS && (N % (uint32_t)kBlock) == 0u;            // whole-row stores: every wave of every block is there
    const int k_rows = K - K % kRS;                        // ... for the steps in whole groups (the rest: per-lane stores)
    const uint32_t row_bytes = N * (uint32_t)(D * 4);

    const float amax = a.amax32, smax = a.smax32, radius = a.radius32;
    const bool inertia_pow2 = a.inertia_pow2 != 0;
    const float inv_inertia = a.inv_inertia32;
    const bool has_max = a.max_steps > 0, autoreset = a.autoreset != 0;
    const uint32_t max_steps = (uint32_t)a.max_steps;
    const bool has_alw = a.alw32 != 0.0f;

    auto norm_rel = [&](const float (&s)[D]) -> float {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < NREL; j++) {
            float dd = s[j] - a.target[j];
            float p = dd * dd;
            acc += (double)p;
        }
        return sqrtf((float)acc);
    };
    // terminal hypercubes over the relevant (= first NREL) coordinates (:945-952)
    auto in_boxes = [&](const float (&s)[D]) -> bool {
        bool any = false;
#pragma unroll
        for (int b = 0; b < MDPP_MAX_BOXES; b++) {     // constant trip count: a run-time index into the
            if (b < a.n_boxes) {                       // argument struct would put it in scratch memory
                bool in = true;
#pragma unroll
                for (int j = 0; j < NREL; j++)
                    in = in && (s[j] >= a.box_lo[b * NREL + j]) && (s[j] <= a.box_hi[b * NREL + j]);
                any = any || in;
            }
        }
        return any;
    };
    // |x_d| <= bound for every d, false if any x_d is NaN: for non-negative floats the IEEE order
    // is the unsigned order of the bit patterns (inf above every finite value, NaN above inf)
    auto all_within = [&](const float (&v)[D], float bound) -> bool {
        uint32_t m = 0;
#pragma unroll
        for (int d = 0; d < D; d++) m = max(m, __float_as_uint(v[d]) & 0x7FFFFFFFu);
        return m <= __float_as_uint(bound);
    };
    auto normal = [&]() -> double {
        // WALK: normals are packed in draw order, slot = (count of the step's first normal + zi) & 31 -- scalar arithmetic;
        // read where they are used (holding a step's 13 doubles in registers spilled the step loop at 168 registers)
        if constexpr (WALK) {
            if (kZ0) { zi++; return 0.0; }           // (sigma 0: the term is 0.0 + 0.0 z = +0.0 for every z)
            return (double)s_z[(size_t)((wk_nb + (uint32_t)(zi++)) & (uint32_t)(kWRing - 1)) * kBlock + ln];
        }
        if (HELPER) return (double)zslot[(zi++) * kBlock];
        if constexpr (PAR) {
            return (double)s_z[(zi++) * 64 + ln];
        } else if constexpr (PHILOX) {
            float v = 0.0f;
#pragma unroll
            for (int d = 0; d < NPS; d++) v = (d == zi) ? zf[d] : v;      // (zi is a compile-time constant after unrolling)
            zi++;
            return (double)v;
        } else {
            return np_standard_normal_lds(g, zig);
        }
    };

    float dist_prev = norm_rel(cur);

    auto load_row = [&](int k, float (&dst)[D]) {
        const uint32_t kk = (uint32_t)min(k, K - 1);
#ifdef MDPP_ABL_NOLOAD
#pragma unroll
        for (int d = 0; d < D; d++) dst[d] = 0.001f * (float)((kk + d + i) & 1023u) - 0.5f;
        return;
#endif
        if (D == 2) {
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r_act, vrow, kk * row_bytes, 0);
            dst[0] = __uint_as_float(v.x); dst[1] = __uint_as_float(v.y);
        } else {
#pragma unroll
            for (int q = 0; q < V; q++) {
                u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r_act, vrow + 16u * q, kk * row_bytes, 0);
                dst[4 * q] = __uint_as_float(v.x); dst[4 * q + 1] = __uint_as_float(v.y);
                dst[4 * q + 2] = __uint_as_float(v.z); dst[4 * q + 3] = __uint_as_float(v.w);
            }
        }
    };

    // (round 6: at D <= 4 a step takes 0.45 us -- four rows ahead were less than a loaded HBM round trip: two buffers of four rows,
    //  c_d2 0.338 -> 0.391 of HBM; three: 0.393; at D = 12 two buffers measure the same as one and eight rows in ONE buffer go to scratch)
    constexpr int kCBufs = NOISE ? 1 : (D <= 4 && MDPP_CBUFS < 2) ? 2 : MDPP_CBUFS;
    static_assert(kCBufs >= 1 && kCBufs <= 3, "one to three named buffers");
    float pre[kCAhead][D], pre1[kCBufs > 1 ? kCAhead : 1][D], pre2[kCBufs > 2 ? kCAhead : 1][D];
    if constexpr (!K1) {
#pragma unroll
    for (int u = 0; u < kCAhead; u++) load_row(u, pre[u]);
    }
    if constexpr (kCBufs > 1 && !K1) {
#pragma unroll
        for (int u = 0; u < kCAhead; u++) load_row(kCAhead + u, pre1[u]);
    }
    if constexpr (kCBufs > 2 && !K1) {
#pragma unroll
        for (int u = 0; u < kCAhead; u++) load_row(2 * kCAhead + u, pre2[u]);
    }

    auto rmin4 = [&](const uint32_t *p) -> uint32_t {
        const uint64_t x = __hip_atomic_load((const uint64_t *)p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint64_t y = __hip_atomic_load((const uint64_t *)p + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return min(min((uint32_t)x, (uint32_t)(x >> 32)), min((uint32_t)y, (uint32_t)(y >> 32)));
    };
    static_assert(kBlock / 64 == 4, "four per-wave counters");
    // this wave's row (w) of group gg, the block's 256 envs: all four waves have staged the group
    auto flush_rows = [&](int gg) __attribute__((always_inline)) {
        uint32_t spins = 0;
        while (rmin4(s_rprod) < (uint32_t)(gg + 1)) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kCSpinLimit) { status |= kCStatusInternal; break; }
        }
        const uint32_t wvu = (uint32_t)__builtin_amdgcn_readfirstlane(wv), l64 = (uint32_t)ln & 63u;
        const uint32_t src = ((uint32_t)(gg % kRBufs) * kRS + wvu) * kBlock + 4u * l64;
        const uint32_t row = ((uint32_t)(gg * kRS) + wvu) * N + (i - (uint32_t)ln) + 4u * l64;
        // (128-bit store: whole offset in the VGPR, see mdpp_discrete_quiet.hip on the store-data hazard)
        __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4 *)&s_rw[CROWS ? src : 0], r_rew, row * 4u, 0, MDPP_ST_NT);
        __builtin_amdgcn_raw_buffer_store_b32(*(const uint32_t *)&s_tw[CROWS ? src : 0], r_term, row, 0, MDPP_ST_NT);
        __builtin_amdgcn_raw_buffer_store_b32(*(const uint32_t *)&s_uw[CROWS ? src : 0], r_trunc, row, 0, MDPP_ST_NT);
        if (l64 == 0) __hip_atomic_store(&s_rcons[wv], (uint32_t)(gg + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto step = [&](const float (&act)[D], int k) {
        const uint32_t so = (uint32_t)k * N;
        float nxt[D];
        if (HELPER) {
            uint32_t spins = 0;
            // (producer k % NPROD has made k / NPROD + 1 steps once step k is in the ring)
            // (WALK: the walker publishes whole steps in s_prod[0])
#ifdef MDPP_ABL_WK_NOWAIT
            if (!WALK)
#endif
            if (!kZ0)                    // (Z0: nothing of the walker's is read)
            while (__hip_atomic_load(&s_prod[WALK ? 0 : k % NPROD][wv], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <
                   (uint32_t)(WALK ? k + 1 : k / NPROD + 1)) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kCSpinLimit) { status |= kCStatusInternal; break; }
            }
            if constexpr (WALK) {
                wk_nb = (uint32_t)k * wk_nd;
                zi = 0;
            } else {
                zslot = s_z + (size_t)(k % kNRing) * NPS * kBlock + ln;
                zi = a.has_p_noise ? 0 : D;
            }
        }
        if (NOISE && PHILOX && !HELPER) { philox_step(k, zf); zi = 0; }
        if (PAR) zi = 0;
        // ---- C1: Box.contains(action)
        const bool ok = all_within(act, amax);
        const bool all_ok = __builtin_amdgcn_ballot_w64(!ok) == 0;
        if constexpr (K1) {
            if (__builtin_expect(!all_ok, 0)) {          // a rejected action keeps every derivative and returns the last observation
#pragma unroll
                for (int d = 0; d < D; d++) sd[ORDER][d] = a.sd[((size_t)ORDER * D + d) * N + i];
#pragma unroll
                for (int d = NREL; d < D; d++) cur[d] = a.cur[(size_t)d * N + i];
            }
        }
        // ---- C2
        // The reference updates state_derivatives in place, lowest order first (:1654-1669); row i
        // only reads rows above it, which are still this step's inputs (row n = a / inertia), so
        // the update is a pure function old rows -> new rows.  Written that way (per-element
        // float accumulators, every index a compile-time constant) the arrays stay in registers;
        // the earlier in-place form through an array reference left sd in scratch memory for
        // D = 12, order 2.
        float nacc[D];                                   // new highest row: a / inertia
        if (inertia_pow2) {                              // wave-uniform choice hoisted out of the loops
#pragma unroll
            for (int d = 0; d < D; d++) nacc[d] = act[d] * inv_inertia;
        } else {
            // (a real branch: without the volatile statement the compiler computes the D IEEE divisions -- ten instructions each --
            //  on BOTH paths and selects: 120 of the ~300 vector instructions of a cfg5 consumer step, for nothing at inertia 1)
            float dv = a.inertia32;
            asm volatile("" : "+v"(dv));
#pragma unroll
            for (int d = 0; d < D; d++) nacc[d] = act[d] / dv;
        }
        // (the whole wave's actions admitted -- the normal case: no per-element selects; otherwise a lane whose
        // action was rejected keeps every derivative, :1671-1679)
        auto integrate = [&](auto all_admitted) __attribute__((always_inline)) {
        constexpr bool ALL = decltype(all_admitted)::value;
#pragma unroll
        for (int ii = 0; ii < ORDER; ii++) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                float acc = sd[ii][d];
#pragma unroll
                for (int j = 0; j < ORDER; j++) {
                    if (j >= ORDER - ii) continue;        // constant trip count: `ORDER - ii` as the bound defeats the unroller
                    const float hi = (ii + j + 1 == ORDER) ? nacc[d] : sd[(ii + j + 1 < ORDER) ? ii + j + 1 : ORDER][d];
                    const float prod = hi * a.tpow32[j + 1];
                    // 1! and 2! are powers of two, so dividing by them is multiplying by 1 or 1/2 exactly -- known at
                    // compile time for the orders this kernel is built for (a run-time test here became two scalar
                    // branches around a float64 division per term: 72 of them per step at D = 12, order 2)
                    // ... and the float64 detour folds away: prod / k! is prod or prod / 2, a 24-bit value, and
                    // round32(round64(acc + prod / k!)) == round32(acc + prod / k!) for 24-bit operands (53 >= 2 * 24 + 2:
                    // double rounding is innocuous for a sum), which is what one fma -- exact product, one rounding -- gives
                    if constexpr (ORDER <= 2) acc = (j + 1 == 2) ? fmaf(prod, 0.5f, acc) : acc + prod;
                    else if ((a.fact_pow2_mask >> (j + 1)) & 1u) acc = (float)((double)acc + (double)prod * a.inv_fact[j + 1]);
                    else acc = (float)((double)acc + (double)prod / a.fact[j + 1]);
                }
                sd[ii][d] = (ALL || ok) ? acc : sd[ii][d];        // rejected action: "stay", nothing moves
            }
        }
#pragma unroll
        for (int d = 0; d < D; d++) sd[ORDER][d] = (ALL || ok) ? nacc[d] : sd[ORDER][d];
#pragma unroll
        for (int d = 0; d < D; d++) nxt[d] = (ALL || ok) ? sd[0][d] : cur[d];                  // "stay", :1671
        };
        if (all_ok) integrate(std::true_type{}); else integrate(std::false_type{});
        status |= (ok || pend) ? 0u : (uint32_t)MDPP_STATUS_BAD_ACTION;      // (an action the reset call ignores is no error)
        // ---- C3
        if (NOISE && a.has_p_noise) {           // (the wave-uniform test outside the per-dimension loop)
#pragma unroll
            for (int d = 0; d < D; d++) nxt[d] = (float)((double)nxt[d] + (0.0 + a.p_noise * normal()));
        } else {
#pragma unroll
            for (int d = 0; d < D; d++) nxt[d] = nxt[d] + 0.0f;      // float32 += float64 zeros: only turns -0 into +0
        }
        // ---- C4: one dimension outside the box clips the whole vector and zeroes every
        // derivative (:1694-1717).  Clipping an in-range coordinate is the identity, so the clip is
        // applied unconditionally (states are finite here: admitted actions and noise are finite and
        // the box is bounded, so NaN handling of the reference's np.clip cannot come into play).
        const bool inside = all_within(nxt, smax);
        if (__builtin_amdgcn_ballot_w64(!inside) != 0) {
#pragma unroll
            for (int d = 0; d < D; d++) nxt[d] = __builtin_amdgcn_fmed3f(nxt[d], -smax, smax);
#pragma unroll
            for (int kk = 0; kk <= ORDER; kk++)
#pragma unroll
                for (int d = 0; d < D; d++) sd[kk][d] = inside ? sd[kk][d] : (kk == 0 ? nxt[d] : 0.0f);
        }
        // ---- C5
        const float dist_new = norm_rel(nxt);
        flags |= (dist_new < radius) ? 1u : 0u;
        steps += 1;
        // ---- C6
        float r;
        if (a.make_denser) r = -dist_new + dist_prev;
        else r = (dist_new < radius) ? 1.0f : 0.0f;
        if (__builtin_expect(has_alw || !all_ok, 0)) {
            double acc = 0.0;
#pragma unroll
            for (int d = 0; d < D; d++) { float p = act[d] * act[d]; acc += (double)p; }
            r = r - a.alw32 * sqrtf((float)acc);
        } else {
            r = r - 0.0f;                       // alw * ||a|| == +0 exactly for an admitted action
        }
        // ---- C7 (!GEN: delay 0, every step pays: the reward stays np.float32 throughout)
        bool done = (flags & 1u) != 0;
        if (!GEN) {
            if (NOISE && a.has_r_noise) {
                if (HELPER || PHILOX || PAR) zi = D;
                if (WALK) zi = a.has_p_noise ? D : 0;
                r = r + (float)(0.0 + a.r_noise * normal());
            }
            if (HELPER && (ln & 63) == 0)   // this wave is done with the step's slot
                __hip_atomic_store(&s_cons[wv], (uint32_t)(k + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            r = r * a.scale32;
            r = r + a.shift32;
            // ---- C8
            r = done ? r + a.term_add32 : r;
        } else {
            double rv = (double)r;
            bool is32 = true;
            if (a.delay > 0) {                                                   // FIFO of float32 bit patterns
                uint32_t *slot = a.ring + (size_t)((rhead0 + (uint32_t)k) % (uint32_t)a.delay) * N + i;
                const uint32_t bits = *slot;
                *slot = __float_as_uint(r);
                if (bits == kRingPyZero) { rv = 0.0; is32 = false; }
                else { rv = (double)__uint_as_float(bits); is32 = true; }
            }
            if (steps % (uint32_t)a.every_n != 0) { rv = 0.0; is32 = false; }
            if (NOISE && a.has_r_noise) {
                if (HELPER || PHILOX || PAR) zi = D;
                if (WALK) zi = a.has_p_noise ? D : 0;
                const double nz = 0.0 + a.r_noise * normal();
                if (is32) rv = (double)((float)rv + (float)nz); else rv = rv + nz;
            }
            if (HELPER && (ln & 63) == 0)
                __hip_atomic_store(&s_cons[wv], (uint32_t)(k + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (is32) {
                rv = (double)((float)rv * a.scale32);
                rv = (double)((float)rv + a.shift32);
            } else {
                rv = rv * a.scale;
                rv = rv + a.shift;
            }
            // ---- C8: a terminal hypercube around the state ends the episode too
            if (a.n_boxes > 0) done = done || in_boxes(nxt);
            if (done) { if (is32) rv = (double)((float)rv + a.term_add32); else rv = rv + a.term_add; }
            r = (float)rv;
        }
        bool tr = has_max && steps >= max_steps;
        dist_prev = dist_new;
#pragma unroll
        for (int d = 0; d < D; d++) cur[d] = nxt[d];
        // ---- episode end: same-step autoreset (reset(), :2284-2323), rare
        bool need = autoreset && (done || tr);
        if (nextmode) {                  // ... or the reset one call later: whatever this lane just computed is dropped
            const bool ended = (done || tr) && !pend;
            need = pend;
            r = pend ? 0.0f : r;
            done = pend ? false : done;
            tr = pend ? false : tr;
            pend = ended;
        }
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(need) != 0, 0)) {
            if (need) {
                if (final_obs && !nextmode) {
#pragma unroll
                    for (int d = 0; d < D; d++) final_obs[((size_t)so + i) * D + d] = nxt[d];
                }
                typename std::conditional<PHILOX, Philox, Pcg64>::type sp;
                if constexpr (PHILOX) sp.init(a.philox_seed, genv, ptick0 + (uint64_t)k, MDPP_STREAM_SPACE);
                else sp.load(a.sp_s, a.sp_inc, i);
                for (int tries = 0;; tries++) {
                    // (unbounded boxes are served by the GEN instantiations, so that the plain kernels do not
                    // carry the normal sampler)
                    if (!GEN || a.bounded) {
#pragma unroll
                        for (int d = 0; d < D; d++) cur[d] = (float)(a.reset_lo + a.reset_range * np_random(sp));
                    } else if (D <= 4) {                      // unbounded Box: gymnasium samples a standard normal
#pragma unroll
                        for (int d = 0; d < D; d++) cur[d] = (float)(0.0 + 1.0 * np_standard_normal(sp));
                    } else {
                        // (a rolled loop through this lane's row of the wave's LDS tile: D inlined copies
                        // of the sampler would be most of the kernel's code)
                        float *row = s_tr + ((size_t)wv * 64 + (ln & 63)) * D;
#pragma unroll 1
                        for (int d = 0; d < D; d++) row[d] = (float)(0.0 + 1.0 * np_standard_normal(sp));
#pragma unroll
                        for (int d = 0; d < D; d++) cur[d] = row[d];
                    }
                    if (!GEN || a.n_boxes == 0 || !in_boxes(cur)) break;     // :2284-2307 resample out of terminal cubes
                    if (tries > 4096) { status |= 2u; break; }
                }
                if constexpr (!PHILOX) sp.store(a.sp_s, i);
                if (GEN) for (int dd = 0; dd < a.delay; dd++) a.ring[(size_t)dd * N + i] = kRingPyZero;
#pragma unroll
                for (int kk = 0; kk <= ORDER; kk++)
#pragma unroll
                    for (int d = 0; d < D; d++) sd[kk][d] = (kk == 0) ? cur[d] : 0.0f;
                steps = 0; flags = 0;
            }
            dist_prev = norm_rel(cur);
        }
        // ---- outputs
#ifdef MDPP_ABL_NOSTORE
        status ^= (__float_as_uint(cur[0]) + __float_as_uint(r) + (done ? 1 : 0) + (tr ? 1 : 0)) & 0x100u;
        return;
#endif
        if (D == 2) {
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(cur[0]), __float_as_uint(cur[1])},
                                                  r_obs, vrow, so * (uint32_t)(D * 4), MDPP_ST_NT);
        } else {
            if (V > 1 && full_wave) {
            // rows of D floats -> the wave's 64 rows as one contiguous block, through a wave-private
            // LDS tile: every store instruction then writes 1 KiB of consecutive bytes instead of 64
            // pieces of 16 B at a stride of 4 D bytes (+15 % on cfg3, tools/ablate_cont.py)
            u32x4 *tile = (u32x4 *)(s_tr + (size_t)wv * 64 * D);
            const int l = ln & 63;
#pragma unroll
            for (int q = 0; q < V; q++)
                tile[l * V + q] = u32x4{__float_as_uint(cur[4 * q]), __float_as_uint(cur[4 * q + 1]),
                                        __float_as_uint(cur[4 * q + 2]), __float_as_uint(cur[4 * q + 3])};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            const uint32_t wbase = (i - (uint32_t)l) * (uint32_t)(D * 4);
#pragma unroll
            for (int q = 0; q < V; q++)
                // (128-bit stores: whole offset in the VGPR, see mdpp_discrete_quiet.hip on the store-data hazard)
                __builtin_amdgcn_raw_buffer_store_b128(tile[q * 64 + l], r_obs,
                                                       wbase + (uint32_t)(q * 64 + l) * 16u + so * (uint32_t)(D * 4), 0, MDPP_ST_NT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            } else {                              // D = 4 rows are contiguous as they are; ragged last wave
#pragma unroll
                for (int q = 0; q < V; q++)
                    __builtin_amdgcn_raw_buffer_store_b128(
                        u32x4{__float_as_uint(cur[4 * q]), __float_as_uint(cur[4 * q + 1]),
                              __float_as_uint(cur[4 * q + 2]), __float_as_uint(cur[4 * q + 3])},
                        r_obs, vrow + 16u * q + so * (uint32_t)(D * 4), 0, MDPP_ST_NT);
            }
        }
#ifdef MDPP_ABL_NORF
        status ^= (__float_as_uint(r) + (done ? 1 : 0) + (tr ? 1 : 0)) & 0x100u;
        return;
#endif
        if constexpr (CROWS) {
            if (crows && k < k_rows) {                  // (wave-uniform)
                const int grp = k / kRS, slot = k & (kRS - 1);
                if (slot == 0 && grp >= kRBufs) {       // the buffer held group grp - 3: stored by all four waves?
                    uint32_t spins = 0;
                    while (rmin4(s_rcons) < (uint32_t)(grp - kRBufs + 1)) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > kCSpinLimit) { status |= kCStatusInternal; break; }
                    }
                }
                const int at = ((grp % kRBufs) * kRS + slot) * kBlock + ln;
                s_rw[at] = r; s_tw[at] = (uint8_t)(done ? 1 : 0); s_uw[at] = (uint8_t)(tr ? 1 : 0);
                if (slot == kRS - 1) {
                    if ((ln & 63) == 0) __hip_atomic_store(&s_rprod[wv], (uint32_t)(grp + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (grp >= 1) flush_rows(grp - 1);
                }
                return;
            }
        }
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), r_rew, v4, so * 4u, MDPP_ST_NT);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(done ? 1 : 0), r_term, v1, so, MDPP_ST_NT);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(tr ? 1 : 0), r_trunc, v1, so, MDPP_ST_NT);
    };

    // chunk c from buffer `buf`, which is refilled with chunk c + kCBufs
    auto chunk = [&](float (&buf)[kCAhead][D], int c) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < kCAhead; u++) {
            float act[D];
#pragma unroll
            for (int d = 0; d < D; d++) act[d] = buf[u][d];
            load_row((c + kCBufs) * kCAhead + u, buf[u]);
            step(act, c * kCAhead + u);
        }
    };
    if constexpr (K1) {
        float act[D];
#pragma unroll
        for (int d = 0; d < D; d++) act[d] = k1act[d];
        step(act, 0);
    }
    const int nfull = K1 ? 0 : K / kCAhead, ngrp = nfull / kCBufs;
    for (int gq = 0; gq < ngrp; gq++) {
        chunk(pre, gq * kCBufs);
        if constexpr (kCBufs > 1) chunk(pre1, gq * kCBufs + 1);
        if constexpr (kCBufs > 2) chunk(pre2, gq * kCBufs + 2);
    }
    // the last full chunks and the ragged tail: rows already in the buffers (loads past the end were clamped)
    for (int k = ngrp * kCBufs * kCAhead; k < (K1 ? 0 : K); k++) {
        float act[D];
        const int rel = k - ngrp * kCBufs * kCAhead, b = rel / kCAhead, u = rel % kCAhead;
#pragma unroll
        for (int uu = 0; uu < kCAhead; uu++)
            if (uu == u) {
#pragma unroll
                for (int d = 0; d < D; d++)
                    act[d] = (kCBufs > 2 && b == 2) ? pre2[kCBufs > 2 ? uu : 0][d] : (kCBufs > 1 && b == 1) ? pre1[kCBufs > 1 ? uu : 0][d] : pre[uu][d];
            }
        step(act, k);
    }

    if constexpr (CROWS) {
        if (crows && k_rows > 0) flush_rows(k_rows / kRS - 1);          // (the last group)
    }
    if constexpr (K1) {
        // One step leaves 150-250 B of state per env: as plain stores they sit dirty in the L2s until the launch ends and are
        // written back THEN, all at once, before the next launch may start (MI355X_MICROARCH.md, "boundary": + B / 6 TB/s for B
        // dirty bytes: 10 MB = 1.7 us per step); as non-temporal stores they leave while the other waves still compute.
#ifndef MDPP_K1_PLAIN_STORES
#ifdef MDPP_K1_ABL_NOSTATE          /* timing only: no state stores at all */
        if (steps == 0x7fffffffu) a.meta[i].x = __float_as_uint(sd[0][0] + cur[0]);
        return;
#endif
        // (where a cfg3 launch's 5.42 us go, tools/ablate_step1.py: without these state stores 3.78, without the observation /
        //  reward / flag stores 5.21, without both 2.71 -- the stores drain at about 7 TB/s, 13.5 MB of them.  Storing the rows as 16
        //  bytes per lane through an LDS tile, 9 instructions instead of 37, changed nothing: 5.40 against 5.48 -- bytes, not requests.)
#pragma unroll
        for (int k = 0; k <= ORDER; k++)
#pragma unroll
            for (int d = 0; d < D; d++) __builtin_nontemporal_store(sd[k][d], &a.sd[((size_t)k * D + d) * N + i]);
#pragma unroll
        for (int d = 0; d < D; d++) __builtin_nontemporal_store(cur[d], &a.cur[(size_t)d * N + i]);
        __builtin_nontemporal_store(steps, &a.meta[i].x);
        __builtin_nontemporal_store(flags | (pend ? 2u : 0u), &a.meta[i].y);
        if (ZIG && !PAR) g.store(a.env_s, i);       // (PAR: the walking wave stored it)
        if (status) atomicOr(&a.status[i], status);
        return;
#endif
    }
#pragma unroll
    for (int k = 0; k <= ORDER; k++)
#pragma unroll
        for (int d = 0; d < D; d++) a.sd[((size_t)k * D + d) * N + i] = sd[k][d];
#pragma unroll
    for (int d = 0; d < D; d++) a.cur[(size_t)d * N + i] = cur[d];
    a.meta[i] = make_uint2(steps, flags | (pend ? 2u : 0u));
    if (ZIG && !HELPER) g.store(a.env_s, i);
    if (status) atomicOr(&a.status[i], status);
}

#ifndef MDPP_CFAST_TU_K1
#define MDPP_CFAST_TU_K1 0         // 1: this translation unit holds the one-step instantiations (mdpp_continuous_step1.hip)
#endif
#if MDPP_CFAST_TU_K1
// mdpp_step (K = 1) on the fast shape: k_continuous_rollout_fast<..., K1 = true>.  Returns false when the shape is not built
// here or the launch needs what K1 leaves out (numpy noise streams: whole 256-env blocks) -- the caller then launches the
// rollout kernel with K = 1 as before.
template <int D, int ORDER, int NREL, bool NOISE, bool GEN, bool PHILOX>
static bool launch_k1(const ContinuousArgs &a, const float *actions, float *obs, float *reward, uint8_t *term, uint8_t *trunc,
                      float *final_obs, hipStream_t s, char *name_out) {
    constexpr bool ZIG = NOISE && !PHILOX;
    constexpr int WG = ZIG ? kBlock : 64;
    // numpy streams with transition noise: the step's draws side by side (PAR: 64 envs per 256-thread workgroup)
    // (D >= 8 only: at D = 2, three draws per step, the four-wave hand-over costs more than the draws it spreads -- c_d2_n0 7.9 us per
    //  step in a replayed graph against 6.4 for the rollout kernel with K = 1; the sequential one-step form below serves small D)
    if constexpr (ZIG && D + 1 <= 16 && D >= 8) {
        if (a.has_p_noise && (a.N % 64) == 0 && !(a.opts & MDPP_OPT_NO_HELPER)) {
            if (name_out) {
                snprintf(name_out, kNameLen, "k_continuous_step1<D=%d,ORDER=%d,NREL=%d,NOISE=%d,GEN=%d,PHILOX=%d,PAR=1>", D, ORDER, NREL, NOISE, GEN, PHILOX);
                return true;
            }
            hipLaunchKernelGGL((k_continuous_rollout_fast<D, ORDER, NREL, NOISE, false, GEN, PHILOX, 1, true, true>), dim3(a.N / 64), dim3(kBlock),
                               0, s, a, 1, actions, obs, reward, term, trunc, final_obs);
            return true;
        }
    }
    if (ZIG && (a.N % kBlock) != 0) return false;
    if (name_out) {
        snprintf(name_out, kNameLen, "k_continuous_step1<D=%d,ORDER=%d,NREL=%d,NOISE=%d,GEN=%d,PHILOX=%d,WG=%d>", D, ORDER, NREL, NOISE, GEN, PHILOX, WG);
        return true;
    }
    hipLaunchKernelGGL((k_continuous_rollout_fast<D, ORDER, NREL, NOISE, false, GEN, PHILOX, 1, true>), dim3((a.N + WG - 1) / WG), dim3(WG),
                       0, s, a, 1, actions, obs, reward, term, trunc, final_obs);
    return true;
}

bool launch_continuous_step1(const ContinuousArgs &a, const float *actions, float *obs, float *reward, uint8_t *term,
                             uint8_t *trunc, float *final_obs, hipStream_t s, char *name_out) {
    if (!a.fast_ok || (a.opts & (MDPP_OPT_NO_CFAST | MDPP_OPT_NO_STEP1)) || (a.philox && (a.opts & MDPP_OPT_NO_PHILOX_FAST))) return false;
    const bool gen = a.delay > 0 || a.every_n != 1 || a.n_boxes > 0 || !a.bounded;
    // (Philox streams carry no state: noise keys whose sigma is 0 add +0.0 whatever the normal is -- the noise-free instantiation serves them)
    const bool sig0 = a.philox && (!a.has_p_noise || a.p_noise == 0.0) && (!a.has_r_noise || a.r_noise == 0.0) && !(a.opts & MDPP_OPT_NO_SIGMA0);
    const bool noise = (a.has_p_noise || a.has_r_noise) && !sig0;
#define MDPP_K1(DD, OO, RR)                                                                                                  \
    if (a.D == DD && a.order == OO && a.n_rel == RR) {                                                                        \
        const int sel = (noise ? 4 : 0) | (gen ? 2 : 0) | (a.philox ? 1 : 0);                                                \
        switch (sel) {                                                                                                        \
        case 0: return launch_k1<DD, OO, RR, false, false, false>(a, actions, obs, reward, term, trunc, final_obs, s, name_out); \
        case 1: return launch_k1<DD, OO, RR, false, false, true>(a, actions, obs, reward, term, trunc, final_obs, s, name_out);  \
        case 2: return launch_k1<DD, OO, RR, false, true, false>(a, actions, obs, reward, term, trunc, final_obs, s, name_out);  \
        case 3: return launch_k1<DD, OO, RR, false, true, true>(a, actions, obs, reward, term, trunc, final_obs, s, name_out);   \
        case 4: return launch_k1<DD, OO, RR, true, false, false>(a, actions, obs, reward, term, trunc, final_obs, s, name_out);  \
        case 5: return launch_k1<DD, OO, RR, true, false, true>(a, actions, obs, reward, term, trunc, final_obs, s, name_out);   \
        case 6: return launch_k1<DD, OO, RR, true, true, false>(a, actions, obs, reward, term, trunc, final_obs, s, name_out);   \
        default: return launch_k1<DD, OO, RR, true, true, true>(a, actions, obs, reward, term, trunc, final_obs, s, name_out);   \
        }                                                                                                                     \
    }
    MDPP_K1(12, 1, 4) MDPP_K1(12, 2, 4)
#ifndef MDPP_CF_SHAPES_MIN
    MDPP_K1(2, 1, 2) MDPP_K1(2, 2, 2) MDPP_K1(4, 1, 4) MDPP_K1(4, 2, 4)
    MDPP_K1(8, 1, 8) MDPP_K1(8, 2, 8) MDPP_K1(12, 1, 12) MDPP_K1(12, 2, 12)
    MDPP_K1(4, 1, 2) MDPP_K1(4, 2, 2) MDPP_K1(8, 1, 4) MDPP_K1(8, 2, 4)
    MDPP_K1(2, 3, 2)                // (the reference's *_move_to_a_point_p_order_3 sweeps)
#endif
#undef MDPP_K1
    return false;
}
#else
template <int D, int ORDER, int NREL, bool GEN, bool PHILOX>
static void launch_g(const ContinuousArgs &a, int K, const float *actions, float *obs, float *reward,
                     uint8_t *term, uint8_t *trunc, float *final_obs, hipStream_t s, char *name_out) {
    const int grid = (a.N + kBlock - 1) / kBlock;
    constexpr bool can_help = (size_t)kNRing * (D + 1) * kBlock * 8 <= 120 * 1024;
    // (Philox streams carry no state: noise keys whose sigma is 0 add +0.0 whatever the normal is -- the noise-free instantiation serves them)
    const bool sig0 = PHILOX && (!a.has_p_noise || a.p_noise == 0.0) && (!a.has_r_noise || a.r_noise == 0.0) && !(a.opts & MDPP_OPT_NO_SIGMA0);
    const bool noise = (a.has_p_noise || a.has_r_noise) && !sig0;
    // producer/consumer split for long rollouts of full blocks (LDS ring: 4 * (D+1) * 2 KiB)
    const bool helper = noise && can_help && K >= 16 && (a.N % kBlock) == 0 && !(a.opts & MDPP_OPT_NO_HELPER);
    // Philox: several producer waves per consumer wave when the per-step draw count makes it worth it
    // numpy streams: generator + walker waves (WALK) under the same conditions
    // (round 5: D = 2 too -- the shape of every continuous experiment file of the reference, whose noise keys at 0 still draw
    //  D + 1 normals per step: tests/test_gpu_sweep.py, bench leg c_d2_n0)
    constexpr bool kWalkD = D >= 8 || D == 2;
    const bool walk = !PHILOX && helper && kWalkD && !(a.opts & (MDPP_OPT_NO_TRIO | MDPP_OPT_NO_PARK));
    const int nprod = (PHILOX && helper && D >= 8 && !(a.opts & MDPP_OPT_NO_TRIO)) ? kPhiloxProducers : (walk ? 2 : 1);
    // sigma-0 noise keys (kernel header, Z0): the D = 2 walker kernels without their normals ring
    const bool z0 = walk && D == 2 && (!a.has_p_noise || a.p_noise == 0.0) && (!a.has_r_noise || a.r_noise == 0.0) && !(a.opts & MDPP_OPT_NO_SIGMA0);
    if (name_out) {
        snprintf(name_out, kNameLen, "k_continuous_rollout_fast<D=%d,ORDER=%d,NREL=%d,NOISE=%d,HELPER=%d,GEN=%d,PHILOX=%d,NPROD=%d%s>", D,
                 ORDER, NREL, noise, helper, GEN, PHILOX, helper ? nprod : 0, z0 ? ",Z0=1" : "");
        return;
    }
    if (noise) {
        ContinuousArgs ap = a;
        ap.park = (a.opts & MDPP_OPT_NO_PARK) ? 0 : 1;
        if (can_help && helper && PHILOX && nprod > 1)
            hipLaunchKernelGGL((k_continuous_rollout_fast<D, ORDER, NREL, true, can_help, GEN, PHILOX, PHILOX ? kPhiloxProducers : 1>),
                               dim3(grid), dim3((1 + kPhiloxProducers) * kBlock), 0, s, ap, K, actions, obs, reward, term,
                               trunc, final_obs);
        else if (can_help && helper && walk && D == 2 && z0)
            hipLaunchKernelGGL((k_continuous_rollout_fast<D, ORDER, NREL, true, can_help && kWalkD, GEN, false, (can_help && kWalkD && !PHILOX) ? 2 : 1, false, false,
                                                          D == 2 && can_help && !PHILOX>), dim3(grid),
                               dim3(3 * kBlock), 0, s, ap, K, actions, obs, reward, term, trunc, final_obs);
        else if (can_help && helper && walk)
            hipLaunchKernelGGL((k_continuous_rollout_fast<D, ORDER, NREL, true, can_help && kWalkD, GEN, false, (can_help && kWalkD && !PHILOX) ? 2 : 1>), dim3(grid),
                               dim3(3 * kBlock), 0, s, ap, K, actions, obs, reward, term, trunc, final_obs);
        else if (can_help && helper)
            hipLaunchKernelGGL((k_continuous_rollout_fast<D, ORDER, NREL, true, can_help, GEN, PHILOX>), dim3(grid),
                               dim3(2 * kBlock), 0, s, ap, K, actions, obs, reward, term, trunc, final_obs);
        else
            hipLaunchKernelGGL((k_continuous_rollout_fast<D, ORDER, NREL, true, false, GEN, PHILOX>), dim3(grid),
                               dim3(kBlock), 0, s, a, K, actions, obs, reward, term, trunc, final_obs);
    } else {
        hipLaunchKernelGGL((k_continuous_rollout_fast<D, ORDER, NREL, false, false, GEN, PHILOX>), dim3(grid), dim3(kBlock),
                           0, s, a, K, actions, obs, reward, term, trunc, final_obs);
    }
}

template <int D, int ORDER, int NREL>
static void launch_t(const ContinuousArgs &a, int K, const float *actions, float *obs, float *reward,
                     uint8_t *term, uint8_t *trunc, float *final_obs, hipStream_t s, char *name_out) {
    const bool gen = a.delay > 0 || a.every_n != 1 || a.n_boxes > 0 || !a.bounded;
    if (a.philox) {
        if (gen) launch_g<D, ORDER, NREL, true, true>(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
        else launch_g<D, ORDER, NREL, false, true>(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
    } else {
        if (gen) launch_g<D, ORDER, NREL, true, false>(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
        else launch_g<D, ORDER, NREL, false, false>(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
    }
}

// Returns false when the shape does not qualify (caller falls back to k_continuous_step).
bool launch_continuous_fast(const ContinuousArgs &a, int K, const float *actions, float *obs,
                            float *reward, uint8_t *term, uint8_t *trunc, float *final_obs,
                            hipStream_t s, char *name_out) {
    if (!a.fast_ok || (a.opts & MDPP_OPT_NO_CFAST) || (a.philox && (a.opts & MDPP_OPT_NO_PHILOX_FAST))) return false;
#define MDPP_CF(DD, OO, RR) if (a.D == DD && a.order == OO && a.n_rel == RR) { launch_t<DD, OO, RR>(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out); return true; }
#ifdef MDPP_CF_SHAPES_D2             // (disassembly builds: the reference's sweep shape alone)
    MDPP_CF(2, 1, 2)
    return false;
#endif
    MDPP_CF(12, 1, 4) MDPP_CF(12, 2, 4)
#ifndef MDPP_CF_SHAPES_MIN          // (resource-usage / ablation builds compile the BASELINE shapes only)
    MDPP_CF(2, 1, 2) MDPP_CF(2, 2, 2) MDPP_CF(4, 1, 4) MDPP_CF(4, 2, 4)
    MDPP_CF(8, 1, 8) MDPP_CF(8, 2, 8) MDPP_CF(12, 1, 12) MDPP_CF(12, 2, 12)
    MDPP_CF(4, 1, 2) MDPP_CF(4, 2, 2) MDPP_CF(8, 1, 4) MDPP_CF(8, 2, 4)
    MDPP_CF(2, 3, 2)                // (the reference's *_move_to_a_point_p_order_3 sweeps)
#endif
#undef MDPP_CF
    return false;
}
#endif   // !MDPP_CFAST_TU_K1

} // namespace mdpp

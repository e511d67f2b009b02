// Four-role pipelined rollout for discrete shapes with at most 8 states: the work of
// k_discrete_rollout_pipe (mdpp_discrete_pipe.hip) re-encoded so that one env step costs about a
// third of its vector instructions, with the output stores marked non-temporal.  Same arithmetic
// and the same results as that kernel and as k_discrete_rollout_fast (reference
// rl_toy_env.py:1992-2125, reset :2250-2278); the same tests hold all three against the oracle.
//
// A 1024-thread workgroup steps 256 envs, lane l of waves w, w+4, w+8, w+12 serving the same env:
//   E   waves 0-3    state recurrence, same-step autoreset from the queue of pre-drawn start states;
//                    three dwords per env step into an LDS ring (13 instructions per step)
//   O1  waves 4-7    reward path: reward bit, delay line, reward value; stores `reward` (numpy streams, full blocks: hands
//                    the rewards to the O2 waves through LDS instead -- "whole-row stores" below)
//   O2  waves 8-11   stores `obs`, `terminated`, `truncated` -- as whole rows of the block where every wave of it is present
//   H   waves 12-15  own the envs' PCG64 streams for the launch and keep an LDS ring of pre-drawn
//                    rho_0 start states filled; un-draw what was not used at the end
// Hand-offs and spin bounds as in mdpp_discrete_pipe.hip (single-producer rings, monotonic counters,
// release/acquire at workgroup scope, polled once per kChunk steps).
//
// The encoding:
//   * history register: one nibble per state, newest lowest, bit 3 = "is a state" (0 = NaN).  A
//     transition is v_perm_b32 (byte `cur` of the action's 8-byte column) + v_lshl_or_b32; a reset
//     replaces the register with (s0 | 8), which also restarts the NaN prefix
//   * column byte = next state | 8 | is_terminal << 7: the terminal test is a compare of the byte
//     just fetched (bit 7 lands on a history bit that is set anyway)
//   * reward bit and NaN gate (:1822) in ONE lookup: a 65 536-bit LDS table indexed by the four
//     low nibbles, zero wherever one of the L + 1 nibbles it needs is not a state
//   * truncation: the step counter is biased so that bit 16 is set exactly when steps >= max_steps;
//     byte 2 of the counter is the `truncated` byte
//   * reward_every_n_steps: a down-counter in units of 16, kept by the O1 lane, that indexes the
//     reward-value table (only its row 0 hands out the bit): the gate (:1975-1976) is part of the lookup
//   * the queue of pre-drawn start states keeps bit 3 of every nibble it holds, so "queue empty"
//     is a test of its low nibble and no count is kept
//   * reward value: one LDS read of a table indexed by (steps to the next pay step, delayed bit, terminated)
//   * "reset happened" = the history words before and after it differ
// NZ (Philox streams only; bit 0 transition noise, bit 1 reward noise; mdpp_discrete_lean_noise.hip): everything random
// about tick t is one word / one float32 normal of a block that serves four ticks (mdpp_rng.hpp philox_pnoise_*,
// PhiloxTickNormals).  H makes the chunk's eight transition-noise nibbles (noisy << 3 | index of the re-drawn state among
// the others) beside its start states; the E lane applies one with seven instructions and re-encodes the column byte
// with a v_perm_b32 of a constant; the O1 lane makes its own eight normals per chunk (two blocks, two packed Box-Muller
// pairs) and forms the reward in float64 in the reference's order (:1980-1990, :2107).
// NZ on NUMPY streams (round 4; mdpp_discrete_lean_npnoise.hip): the reference's own draws, bit for bit, without putting a
// generator on the E lane (k_discrete_rollout_quiet<PN, RN> runs both PCG64 streams, the categorical search and the
// ziggurat inside the recurrence: 607 us per cfg2 launch).  The H wave owns both streams:
//   * transition noise (:1604-1622) takes exactly one word of the state space's stream per step, so H makes the chunk's
//     eight words ahead like the Philox form; the categorical around table entry n is searched ONCE for all n -- with
//     a = #{j : TL[j] <= r}, b = #{j : TU[j] <= r} the re-drawn state is min(a, n) + max(b - n, 0) (DiscreteArgs::pn_TL:
//     exact whenever the rows' thresholds agree, which the host checks) -- and E applies the byte (a | b << 4);
//   * reward noise (:1980-1984) and reset() (:2255) share the ENV stream: per step one ziggurat normal (1, 2 or more
//     words), then one word if the episode ended -- only E knows where a step's draws start.  H therefore evaluates
//     EVERY position p of the stream as if a draw started there: meta[p] = {the start state word p would give a reset,
//     kind: normal accepted at once / wedge point accepted (2 words) / wedge point rejected (2 words, the draw starts
//     over) / tail (count in the high byte, the start state behind its words in bits 5-7)} and x[p] = that draw's value, into 16-position rings per lane; the
//     ziggurat's slow path uses a COPY of the generator (its words are evaluated again as positions of their own).  E
//     walks positions: reads meta[p], meta[p + 1], meta[p + 2] (fetched one step ahead), copies x[p] into the step's
//     record for O1, and moves on by the draw's words (+ 1 if the episode ended); H runs up to 16 positions ahead of the
//     position E publishes and un-draws what E did not reach at the end of the launch.
// What the measurements said (profiles/archive/r02_ablation_lean_kernel.txt): with the default cache policy
// the time was set by the stores, whoever issued them (147 us per 512-step launch of 65 536 envs with
// one, two or three storing waves per SIMD); marked nt they cost 10 us on top of the 95 us the
// recurrence + start-state draws take, 106 us = 5.7 TB/s.
#include <stdlib.h>

#include <stdio.h>

#include <type_traits>

#ifndef MDPP_LEAN_TU_NEXT
#define MDPP_LEAN_TU_NEXT 0        // 1: this translation unit holds the next-step autoreset instantiations
#endif
#ifndef MDPP_LEAN_TU_NOISE
#define MDPP_LEAN_TU_NOISE 0       // 1: ... the transition- / reward-noise instantiations (Philox streams); 2: on numpy streams
#endif

#include "mdpp_internal.hpp"
#include "mdpp_rng.hpp"

namespace mdpp {

#ifndef MDPP_LEAN_CHUNK
#define MDPP_LEAN_CHUNK 8
#endif
#ifndef MDPP_LEAN_DEPTH
#define MDPP_LEAN_DEPTH 32
#endif
#ifndef MDPP_LEAN_AHEAD
#define MDPP_LEAN_AHEAD 4
#endif
#if defined(MDPP_LEAN_PRIO_E) || defined(MDPP_LEAN_PRIO_O) || defined(MDPP_LEAN_PRIO_H)
#define MDPP_LEAN_PRIO_FORCED 1    // (tools/ablate.py: the same priorities for every variant)
#else
#define MDPP_LEAN_PRIO_FORCED 0
#endif
#ifndef MDPP_LEAN_PRIO_E
#define MDPP_LEAN_PRIO_E 3
#endif
#ifndef MDPP_LEAN_PRIO_O
#define MDPP_LEAN_PRIO_O 2
#endif
#ifndef MDPP_LEAN_PRIO_H
#define MDPP_LEAN_PRIO_H 0
#endif
#ifndef MDPP_LEAN_PRIO_O2
#define MDPP_LEAN_PRIO_O2 -1       // (>= 0: the O2 waves' own priority; tools/ablate.py)
#endif
#ifndef MDPP_LEAN_ST_AUX
#define MDPP_LEAN_ST_AUX MDPP_ST_NT
#endif
#ifndef MDPP_LEAN_ST_AUX_BYTES
#define MDPP_LEAN_ST_AUX_BYTES MDPP_LEAN_ST_AUX
#endif
#ifndef MDPP_LEAN_LD_AUX
#define MDPP_LEAN_LD_AUX 0
#endif
#ifndef MDPP_LEAN_HMIN
#define MDPP_LEAN_HMIN 16
#endif
#ifndef MDPP_LEAN_PHILOX_PN_O2
#define MDPP_LEAN_PHILOX_PN_O2 1   // Philox transition noise: the chunk's noise nibbles are made by the O2 wave (1) / the H wave (0)
#endif
#ifndef MDPP_LEAN_H_LIMBS
#define MDPP_LEAN_H_LIMBS 0        // (the limb form of PCG64 on the numpy H wave: 137 us per cfg2 launch either way -- H is not what bounds it)
#endif
#ifndef MDPP_LEAN_H_SSTAB
#define MDPP_LEAN_H_SSTAB 0        // numpy H wave without noise: start state by the word's top 11 bits from lds_sstab (1) / eight compares (0)
#endif
#ifndef MDPP_LEAN_HSLEEP
#define MDPP_LEAN_HSLEEP 8
#endif
#ifndef MDPP_LEAN_XCD_CONTIG
#define MDPP_LEAN_XCD_CONTIG 1     // every XCD steps one contiguous eighth of the envs (0: blocks in launch order, round-robin over the XCDs)
#endif
#ifndef MDPP_LEAN_ROWS
#define MDPP_LEAN_ROWS 3           // whole-row stores (header): bit 0 obs and flags, bit 1 the rewards too, bit 2 also in the noise instantiations
#endif
namespace lean {
constexpr int kChunk = MDPP_LEAN_CHUNK;   // steps between hand-off polls; also the action prefetch distance
constexpr int kDepth = MDPP_LEAN_DEPTH;   // E->O ring depth in steps (multiple of kChunk)
constexpr int kAhead = MDPP_LEAN_AHEAD;   // action prefetch distance in chunks
constexpr int kPRsrc = 0x00020000;
constexpr uint32_t kSpinLimit = 1u << 22;
constexpr uint32_t kStatusInternal = 0x80000000u;
constexpr uint32_t kQueueCap = 6;
constexpr int kHChunks = 8;               // Philox streams: H runs up to this many chunks ahead of E
constexpr int kHChunksNp = 4;             // numpy transition noise: ... this many (LDS)
// (Measured and not kept, round 6: H running TWO words ahead so that a position's meta carries the start states of both words behind it
//  and E reads one entry per step instead of three -- d_s8_rn0 232 -> 248 us, cfg2 + both noises 274 -> 277: H is the long stage of
//  this form, work moved onto it costs more than E saves.  The other direction -- Z0: a 32-bit meta with the word's top 11 bits, E looks the
//  start state up itself, the words kept in LDS for the <= 8 buckets of 2 048 a threshold falls into -- 232 -> 253 us: + 40 KB of LDS,
//  one workgroup per CU instead of two.)
constexpr int kXR = 16;                   // numpy reward noise: stream positions H evaluates ahead of E (per lane)
constexpr int kXB = 4;                    // ... per batch (divides kXR; 8 with kXR = 16 leaves E too little lead: 497 -> 586 us)
constexpr int kDepthNp = 16;              // ... and the E->O ring depth of these instantiations (records carry the normal)
constexpr int kRoles = 4;                 // E, O1 (reward path), O2 (observation / flag stores), H
constexpr uint32_t kSelPad = 0x0c0c0c00u; // v_perm_b32 selector bytes 1-3: constant 0x00
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t wg_load_acq(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void wg_store_rel(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the words of the chunk's eight ticks: blocks b0, b0 + 1 (and b0 + 2 when the launch is r4 ticks into its first block)
__device__ __forceinline__ void chunk_words(uint64_t seed, uint64_t genv, uint64_t b0, uint32_t r4, uint32_t stream,
                                            uint32_t (&wd)[MDPP_LEAN_CHUNK]) {
    uint32_t o0[4], o1[4], o2[4] = {0u, 0u, 0u, 0u};
    philox_start_block(seed, genv, b0, stream, o0);
    philox_start_block(seed, genv, b0 + 1, stream, o1);
    if (r4 != 0u) philox_start_block(seed, genv, b0 + 2, stream, o2);
    const uint32_t all[12] = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3], o2[0], o2[1], o2[2], o2[3]};
#pragma unroll
    for (int u = 0; u < MDPP_LEAN_CHUNK; u++)
        wd[u] = r4 == 0u ? all[u] : r4 == 1u ? all[u + 1] : r4 == 2u ? all[u + 2] : all[u + 3];
}
// ... and their reward normals (four per block: PhiloxTickNormals)
__device__ __forceinline__ void chunk_normals(uint64_t seed, uint64_t genv, uint64_t b0, uint32_t r4, uint32_t stream,
                                              float (&z)[MDPP_LEAN_CHUNK]) {
    float all[12] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t o[4];
    philox_start_block(seed, genv, b0, stream, o);
    philox_box_muller2(o, all[0], all[1], all[2], all[3]);
    philox_start_block(seed, genv, b0 + 1, stream, o);
    philox_box_muller2(o, all[4], all[5], all[6], all[7]);
    if (r4 != 0u) {
        philox_start_block(seed, genv, b0 + 2, stream, o);
        philox_box_muller2(o, all[8], all[9], all[10], all[11]);
    }
#pragma unroll
    for (int u = 0; u < MDPP_LEAN_CHUNK; u++)
        z[u] = r4 == 0u ? all[u] : r4 == 1u ? all[u + 1] : r4 == 2u ? all[u + 2] : all[u + 3];
}
} // namespace lean
using namespace lean;

template <bool OBS64, bool DELAY, bool HASMAX, bool EVN, bool PHILOX, bool IRR, bool NEXT, int NZ = 0>
__global__ __launch_bounds__(kRoles * kBlock) void k_discrete_rollout_lean(DiscreteArgs a, int K,
                                                                      const int32_t *__restrict__ actions,
                                                                      void *__restrict__ obs,
                                                                      float *__restrict__ reward,
                                                                      uint8_t *__restrict__ term,
                                                                      uint8_t *__restrict__ trunc,
                                                                      void *__restrict__ final_obs) {
    const uint64_t ptick0 = tick_now(a);               // the step counter at this launch (through the device-side offset of a graph replay)
    // E -> O, three dwords per env step in ONE array (constant 32 KB apart: two of them go out as one ds_write2st64):
    //   A history before a reset   B history after it   C byte 0: the column entry (bit 7 terminated), byte 1 the
    //   irrelevant observation, byte 2 truncated
    constexpr bool PN = (NZ & 1) != 0, RN = (NZ & 2) != 0;
    constexpr bool NPN = PN && !PHILOX, NRN = RN && !PHILOX;          // noise on numpy streams
    // Z0 (NZ bit 2, numpy streams; round 6): the reward_noise key is PRESENT with sigma 0 -- what every discrete experiment file of
    // the reference passes.  The reference still draws rng.normal(0, 0) per step (rl_toy_env.py:398-403, :1982), so the stream
    // positions, the ziggurat's accept decisions and the reset words behind them are all kept -- but a normal's VALUE is only
    // ever multiplied by 0: r + (0.0 + 0 z) == r bit for bit for r in {0.0, 1.0} and finite z.  H then makes meta[p] alone (no
    // rabs * wi, no sign insert, no lds_x write; x is formed only on the wedge path, which needs it to decide), E copies nothing
    // per step, and O1 pays from the reward-value table like the noise-free kernel (a.rsel holds the same float64 arithmetic).
    constexpr bool Z0 = (NZ & 4) != 0;
    static_assert(!Z0 || NRN, "sigma-0 draws: numpy streams with the reward_noise key");
    constexpr bool NRX = NRN && !Z0;                                  // ... whose values are used
    // (Z0 without transition noise: the records carry no normal and the O waves are the noise-free kernel's -- the deep ring and the
    //  whole-row stores of the noise-free form, MDPP_LEAN_Z0_ROWS)
#ifndef MDPP_LEAN_Z0_ROWS
#define MDPP_LEAN_Z0_ROWS 0        /* measured: d_s8_rn0 236 -> 233-238 us, nothing -- the H / E position pipeline bounds this form, not its stores */
#endif
    constexpr bool ZROWS = MDPP_LEAN_Z0_ROWS && Z0 && !PN;
    constexpr int KD = (NRN && !ZROWS) ? kDepthNp : kDepth;           // E->O ring depth in steps
    __shared__ __align__(16) uint32_t lds_rec[3][KD][kBlock];
    __shared__ __align__(16) uint32_t lds_V[2048];            // reward bit & NaN gate, by the 4 low nibbles
    __shared__ __align__(16) uint2 lds_col[16];               // action a, byte s: P[s][a] | 8 | is_term[P[s][a]] << 7
    __shared__ __align__(16) uint32_t lds_R[128];             // 4096 reward bits by key (staging for lds_V)
    __shared__ __align__(16) uint64_t lds_T[8];               // rho_0 thresholds
    __shared__ __align__(16) float lds_rsel[EVN ? 4 * 64 : 4]; // reward by (steps to pay step, bit, terminated)
    __shared__ __align__(16) uint64_t lds_ring[kBlock];       // H -> E: {8 nibbles, #pushed}
    __shared__ uint32_t lds_head[kBlock];                     // E -> H: #popped
    __shared__ uint32_t lds_prod[kBlock / 64];                // steps published by E wave w
    __shared__ __align__(8) uint32_t lds_cons[kBlock / 64][2]; // steps consumed by the O1 / O2 wave w
    __shared__ uint32_t lds_done;                             // E waves that have finished
    // Whole-row stores (MDPP_LEAN_ROWS; tools/bench_store.hip: the same bytes leave 10 % faster when one wave writes the
    // workgroup's whole piece of an output row -- obs 2 KiB, reward 1 KiB, flags 256 B -- than when every wave writes its
    // own 64 envs of every row): of a chunk's 8 rows the O waves w take rows w and w + 4, for all 256 envs of the block.
    // The O2 wave reads the other E waves' records (E waits for ALL four O2 waves before it reuses a ring slot); the O1
    // wave -- its reward path carries state from step to step -- computes its own envs' 8 rewards as before and hands
    // them to the O2 waves through lds_rw (kRB chunks deep), which store them too.  Blocks with spare lanes (N % 256)
    // keep the per-lane stores.
    constexpr bool ROWS2 = (MDPP_LEAN_ROWS & 1) != 0 && !IRR && (NZ == 0 || ZROWS || (MDPP_LEAN_ROWS & 4) != 0);   // (noise: the O waves are long stages there -- no gain, measured)
    // (rewards through the O2 waves: numpy streams only -- with Philox streams, where E runs at the lowest priority, 124-128 us per cfg2
    //  launch became 132-133; numpy streams 115-120 -> 112-118; without whole-row stores 129-137, all on one lease)
    constexpr bool ROWS1 = (MDPP_LEAN_ROWS & 2) != 0 && ROWS2 && !PHILOX && (!NRN || ZROWS);
    constexpr int kRB = ZROWS ? 2 : 4;                        // chunks of staged rewards (the O1 waves run this far ahead of the stores; Z0: LDS)
    __shared__ __align__(16) float lds_rw[ROWS1 ? kRB : 1][kChunk][ROWS1 ? kBlock : 4];
    __shared__ __align__(16) uint32_t lds_rprod[kBlock / 64], lds_rcons[kBlock / 64];   // chunks staged by O1 wave w / stored by O2 wave w
    typedef typename std::conditional<IRR, uint64_t, uint32_t>::type S0Word;   // a nibble per tick; with an irrelevant sub-space two
    __shared__ __align__(16) S0Word lds_s0[PHILOX ? kHChunks : 1][kBlock];     // Philox: H -> E, the chunk's 8 start states
    __shared__ __align__(16) uint2 lds_col1[IRR ? 16 : 1];    // irrelevant sub-space: action a1, byte s1: P1[s1][a1]
    __shared__ __align__(16) uint64_t lds_T1[IRR ? 8 : 1];    // its rho_0 thresholds
    __shared__ uint32_t lds_hprod[kBlock / 64];               // Philox: chunks published by H wave w
    __shared__ uint32_t lds_pprod[kBlock / 64];               // Philox transition noise made by the O2 wave: chunks published
    // Wave priorities (s_setprio).  numpy streams: the serial recurrence (E) first, the H wave's PCG64 draws are filler work.
    // Philox streams: the waves that make Philox blocks are the long stages and go first -- H (start states, and with PN
    // the transition-noise words), with RN the O1 wave (normals) -- and the E wave, now the shortest stage, last:
    // cfg2 128 -> 122 us per launch, + transition noise 195 -> 144, + reward noise 181 -> 167, both 227 -> 184
    // (profiles/archive/r03_ablation_lean_priorities.txt).
    // numpy streams with noise: H (both generators) is the long stage: E 3 O 2 H 0 -> 497 us per cfg2 + noise launch, E 2 O 1 H 3 -> 420
    // (profiles/r04_ablation_npnoise.txt)
    constexpr bool kNpNoise = !PHILOX && NZ != 0;
    // Philox streams with noise (round 4, the transition-noise nibbles made by the O2 wave): E last, the three waves that make
    // Philox blocks level -- E 1, O1 2, H 2, O2 2 (3 without reward noise): 191 -> 168 us both noises, 143 -> 138 transition
    // noise alone (24 combinations: profiles/r04_ablation_lean_philox_noise.txt)
    constexpr bool kPhNoise = PHILOX && NZ != 0 && MDPP_LEAN_PHILOX_PN_O2;
    constexpr int kPrioE = MDPP_LEAN_PRIO_FORCED ? MDPP_LEAN_PRIO_E : kNpNoise ? 2 : !PHILOX ? MDPP_LEAN_PRIO_E : 1;
    constexpr int kPrioO = MDPP_LEAN_PRIO_FORCED ? MDPP_LEAN_PRIO_O : kNpNoise ? 1 : !PHILOX ? MDPP_LEAN_PRIO_O : kPhNoise ? 2 : (PN ? 2 : 3);
    constexpr int kPrioH = MDPP_LEAN_PRIO_FORCED ? MDPP_LEAN_PRIO_H : kNpNoise ? 3 : !PHILOX ? MDPP_LEAN_PRIO_H : kPhNoise ? 2 : (PN ? 3 : 2);
    static_assert(NZ == 0 || (!IRR && !NEXT), "noise on the lean kernel: one sub-space, same-step autoreset");
    __shared__ __align__(16) uint32_t lds_pn[(PN && PHILOX) ? kHChunks : 1][kBlock];       // H -> E: the chunk's 8 transition-noise nibbles
    // numpy streams (header): transition-noise bytes of a chunk; per stream position the draw's value and {start state, kind, words}
    __shared__ __align__(16) uint32_t lds_pn2[NPN ? kHChunksNp : 1][2][kBlock];
    __shared__ __align__(16) double lds_x[NRX ? kXR : 1][kBlock];
    __shared__ uint16_t lds_meta[NRN ? kXR : 1][kBlock];
    __shared__ __align__(16) double lds_rx[NRX ? KD : 1][kBlock];      // E -> O1: the step's normal
    __shared__ uint32_t lds_hhead[NRN ? kBlock : 1], lds_epos[NRN ? kBlock : 1];   // positions made by H / reached by E
    // the two categorical searches of H by the word's top bits: byte = the answer where the bucket holds no threshold, 0xFF
    // where it does (then the thresholds are counted: a few lanes per thousand)
    constexpr bool HSS = MDPP_LEAN_H_SSTAB && !PHILOX && NZ == 0 && !IRR;
    __shared__ uint8_t lds_sstab[(NRN || NPN || HSS) ? 2048 : 1];     // start state by r >> 53
    __shared__ uint8_t lds_pntab[NPN ? 4096 : 1];                     // a | b << 4 by r >> 52
    __shared__ ulonglong2 lds_kw[NRN ? 256 : 1];                       // ziggurat {ki, wi * 2^52}
    __shared__ double lds_fi[NRN ? 256 : 1];
    constexpr int kEN = IRR ? 2 : 1;                // nibbles per start-state entry (relevant, irrelevant)
    // gymnasium's next-step autoreset: the call after an episode's last step IS the reset (action ignored, reward 0,
    // no flags); the pending flag travels in bit 31 of the step counter like in k_discrete_step
    // (a template flag, instantiated in its own translation unit, mdpp_discrete_lean_next.hip: as a run-time
    //  flag its selects cost the same-step mode 9 us of 106 per launch)
    constexpr bool nextmode = NEXT;
    const bool ar = a.autoreset != 0;               // autoreset "disabled" (the reference's own behaviour): no resets, H idle
    const int tid = threadIdx.x;
    const int role = tid / kBlock;                  // 0 = E, 1 = O1, 2 = O2, 3 = H
    const int l = tid & (kBlock - 1);               // env slot inside the block
    const int w = l >> 6;
    const uint32_t S = (uint32_t)a.S, L = (uint32_t)a.L;
    if (tid < 16) {
        uint64_t cs = 0;
        if (tid < a.A)
            for (uint32_t s = 0; s < S; s++) {
                const uint32_t nx = a.P[s * a.A + tid] & 7u;
                cs |= (uint64_t)(nx | 8u | ((uint32_t)((a.term_mask >> nx) & 1ULL) << 7)) << (8 * s);
            }
        lds_col[tid] = make_uint2((uint32_t)cs, (uint32_t)(cs >> 32));
        if (tid < 8) lds_T[tid] = a.init_thr[tid];
        if (IRR) {                                              // (:2028-2035, :2063-2082, reset :2259-2264)
            uint64_t c1 = 0;
            if (tid < a.A1)
                for (int s1 = 0; s1 < a.S1; s1++) c1 |= (uint64_t)(a.P1[s1 * a.A1 + tid] & 7u) << (8 * s1);
            lds_col1[tid] = make_uint2((uint32_t)c1, (uint32_t)(c1 >> 32));
            if (tid < 8) lds_T1[tid] = tid < a.S1 ? (uint64_t)ceil(a.init_cdf1[tid] * 9007199254740992.0) : ~0ULL;
        }
        if (!EVN && tid < 4) lds_rsel[tid] = a.rsel[tid];
    }
    if (EVN)                                                  // only pay steps hand out the bit (:1975-1976)
        for (int k = tid; k < 4 * a.every_n; k += kRoles * kBlock) lds_rsel[k] = a.rsel[k < 4 ? k : (k & 1)];
    for (uint32_t k = tid; k < 128; k += kRoles * kBlock) {
        uint32_t wd = 0;
        for (int b = 0; b < 4; b++) {
            uint32_t byte = 4 * k + b;
            if (byte < a.rbits_stride) wd |= (uint32_t)a.rbits[byte] << (8 * b);
        }
        lds_R[k] = wd;
    }
    if (tid < kBlock) { lds_ring[tid] = 0; lds_head[tid] = 0; if (NRN) { lds_hhead[tid] = 0; lds_epos[tid] = 0; } }
    if (NRN || NPN || HSS) {
        for (uint32_t k = tid; k < 2048u; k += kRoles * kBlock) {
            const uint64_t lo = (uint64_t)k << 53, hi = lo + ((1ULL << 53) - 1ULL);       // the bucket's words
            uint32_t c0 = 0, c1 = 0;
            for (int j = 0; j < 8; j++) {
                const uint64_t t = a.init_thr[j] > (1ULL << 53) - 1ULL ? ~0ULL : a.init_thr[j] << 11;
                c0 += (t <= lo) ? 1u : 0u; c1 += (t <= hi) ? 1u : 0u;
            }
            lds_sstab[k] = (uint8_t)(c0 == c1 ? c0 : 0xFFu);
        }
    }
    if (NPN) {
        for (uint32_t k = tid; k < 4096u; k += kRoles * kBlock) {
            const uint64_t lo = (uint64_t)k << 52, hi = lo + ((1ULL << 52) - 1ULL);
            uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
            for (int j = 0; j < 7; j++) { a0 += (a.pn_TL[j] <= lo) ? 1u : 0u; a1 += (a.pn_TL[j] <= hi) ? 1u : 0u; }
            for (int j = 0; j < 8; j++) { b0 += (a.pn_TU[j] <= lo) ? 1u : 0u; b1 += (a.pn_TU[j] <= hi) ? 1u : 0u; }
            lds_pntab[k] = (uint8_t)((a0 == a1 && b0 == b1) ? (a0 | (b0 << 4)) : 0xFFu);
        }
    }
    if (NRN)
        for (int k = tid; k < 256; k += kRoles * kBlock) { lds_kw[k] = make_ulonglong2(d_zig_ki[k], (unsigned long long)(__double_as_longlong(d_zig_wi[k]) + (52LL << 52)));   /* {ki, W = wi 2^52} */ lds_fi[k] = d_zig_fi[k]; }
    if (tid < kBlock / 64) { lds_prod[tid] = 0; lds_cons[tid][0] = 0; lds_cons[tid][1] = 0; lds_hprod[tid] = 0; lds_pprod[tid] = 0; lds_rprod[tid] = 0; lds_rcons[tid] = 0; }
    if (tid == 0) lds_done = 0;
    __syncthreads();
    {   // lds_V: dword d holds indices 32 d .. 32 d + 31; nibble j of an index = (d * 32 + b) >> 4 j.
        // Non-zero only where nibbles 0..L all carry bit 3: nibbles 1..L are fixed by d.
        uint32_t dmask = 0;
        for (uint32_t j = 1; j <= L; j++) dmask |= 1u << (4 * j - 2);
        for (uint32_t d = tid; d < 2048; d += kRoles * kBlock) {
            uint32_t wd = 0;
            if ((d & dmask) == dmask) {
                for (uint32_t b = 8; b < 32; b++) {
                    if (!(b & 8u)) continue;
                    const uint32_t idx = d * 32u + b;
                    uint32_t key = 0, pw = 1;
                    for (uint32_t j = 0; j < L; j++) { key += ((idx >> (4 * j)) & 7u) * pw; pw *= S; }
                    if (key < a.nkeys) wd |= ((lds_R[key >> 5] >> (key & 31u)) & 1u) << b;
                }
            }
            lds_V[d] = wd;
        }
    }
    // (the episode step count at launch, for the O1 lane's every-n phase: read BEFORE the barrier -- the E lane
    //  writes the word back at the end of the launch, and with K <= kDepth it could get there before O1 starts)
    uint32_t steps_at_launch = 0;
    {
        const uint32_t eb0 = (MDPP_LEAN_XCD_CONTIG && (gridDim.x & 7u) == 0u) ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
        const uint32_t i0 = eb0 * kBlock + l;
        if (EVN && role == 1 && i0 < (uint32_t)a.N) steps_at_launch = ((const uint32_t *)&a.state[i0])[2] & 0x7FFFFFFFu;
    }
    __syncthreads();

    // Workgroup b runs on XCD b % 8 (round-robin dispatch).  Give every XCD one contiguous eighth of the
    // envs, so that what its L2 writes back per output row is one contiguous range, not every eighth
    // 256-env piece of it.
    const uint32_t eblk = (MDPP_LEAN_XCD_CONTIG && (gridDim.x & 7u) == 0u) ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const uint32_t i = eblk * kBlock + l;
    const uint32_t N = (uint32_t)a.N;
    const bool rows = ROWS2 && (N % (uint32_t)kBlock) == 0u;       // (every wave of every block is there)
    const uint32_t blk0 = eblk * kBlock, ln = (uint32_t)l & 63u;
    const uint32_t ws = (uint32_t)__builtin_amdgcn_readfirstlane(w);
    // min over the four waves' counters (two 64-bit LDS reads)
    auto min4 = [&](const uint32_t *p) -> uint32_t {
        const uint64_t x = __hip_atomic_load((const uint64_t *)p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint64_t y = __hip_atomic_load((const uint64_t *)p + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return min(min((uint32_t)x, (uint32_t)(x >> 32)), min((uint32_t)y, (uint32_t)(y >> 32)));
    };
    if (i >= N) {                                   // ragged last block: its spare lanes leave (every wave that stays keeps
        // lane 0, which publishes the hand-off counters; no barrier follows); an E wave that leaves entirely counts as done
        if (role == 0 && (l & 63) == 0) __hip_atomic_fetch_add(&lds_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    const uint32_t A = (uint32_t)a.A;
    constexpr int kMinLanes = MDPP_LEAN_HMIN;       // H draws for the whole wave once this many lanes have room (or one runs low)
    uint32_t status = 0;

    // =============================================================== H: start-state producer
    if (role == 3) {
#if defined(MDPP_ABL_NOH) && defined(MDPP_ABL_NORESET)
        return;         // (only together with NORESET: without the H waves every reset spins to its bound -- minutes per launch)
#endif
        if (!ar && !PN && !NRN) return;
        __builtin_amdgcn_s_setprio(kPrioH);
        if constexpr (PHILOX) {
            // Philox streams: the start state a reset at tick t draws is a function of (seed, env, t) alone -- ONE 32-bit
            // word of the start-state stream, four ticks to a block (mdpp_rng.hpp philox_start_*, as in k_discrete_step /
            // _quiet) --, so H makes the one of EVERY tick, a chunk of 8 at a time: two blocks (three when the launch
            // does not start on a multiple of four ticks), 31-bit thresholds in scalar registers; packed as 8 nibbles | 8
            const uint64_t genv = (uint64_t)(a.env_id_offset + (int64_t)i);
            const int nch = (K + kChunk - 1) / kChunk;
            static_assert(kChunk == 8, "two start-state blocks per chunk");
            uint32_t thr[8], thr1[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {                        // ceil(cdf 2^31) from ceil(cdf 2^53); padding: never reached
                const uint64_t t53 = lds_T[j];
                thr[j] = __builtin_amdgcn_readfirstlane(t53 > (1ULL << 53) ? 0x80000000u : (uint32_t)((t53 + 0x3FFFFFULL) >> 22));
                const uint64_t u53 = IRR ? lds_T1[j] : ~0ULL;
                thr1[j] = __builtin_amdgcn_readfirstlane(u53 > (1ULL << 53) ? 0x80000000u : (uint32_t)((u53 + 0x3FFFFFULL) >> 22));
            }
            const uint32_t r4 = (uint32_t)ptick0 & 3u;            // the launch's offset inside its first block (wave-uniform)
            for (int c = 0; c < nch; c++) {
                if (c >= kHChunks) {                             // slot c % kHChunks: E must be through chunk c - kHChunks
                    const uint32_t must = (uint32_t)min((c - kHChunks + 1) * kChunk, K);
                    uint32_t spins = 0;
                    while (wg_load_acq(&lds_prod[w]) < must) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > kSpinLimit) { status |= kStatusInternal; break; }
                    }
                }
                const uint64_t b0 = (ptick0 + (uint64_t)(c * kChunk)) >> 2;
                auto words = [&](uint32_t stream, uint32_t (&wd)[kChunk]) { chunk_words(a.philox_seed, genv, b0, r4, stream, wd); };
                uint32_t wd[kChunk], wd1[kChunk];
                words(kPhiloxStartStream, wd);
                if (IRR) words(kPhiloxStartIrrStream, wd1);
                S0Word pk = 0;
#pragma unroll
                for (int u = 0; u < kChunk; u++) {
                    const uint32_t m = wd[u] >> 1;
                    uint32_t s0 = 0;
#pragma unroll
                    for (int j = 0; j < 7; j++) s0 += (thr[j] <= m) ? 1u : 0u;       // (entry 7: cdf 1.0 or padding, 2^31 > m)
                    if (IRR) {
                        const uint32_t m1 = wd1[u] >> 1;
                        uint32_t s1 = 0;
#pragma unroll
                        for (int j = 0; j < 7; j++) s1 += (thr1[j] <= m1) ? 1u : 0u;
                        s0 |= (s1 | 8u) << 4;
                    }
                    pk |= (S0Word)(s0 | 8u) << (kEN * 4 * u);
                }
                lds_s0[c % kHChunks][l] = pk;
                if constexpr (PN && !MDPP_LEAN_PHILOX_PN_O2) {   // noisy << 3 | j (S <= 8: j <= 6), mdpp_rng.hpp philox_pnoise_index
                    uint32_t wp[kChunk], pn = 0;
                    words(kPhiloxPNoiseStream, wp);
#pragma unroll
                    for (int u = 0; u < kChunk; u++) {
                        const uint32_t e = philox_pnoise_index(wp[u], a.pn_T, a.pn_M);
                        pn |= ((e & 7u) | ((e >> 5) & 8u)) << (4 * u);
                    }
                    lds_pn[c % kHChunks][l] = pn;
                }
                if ((l & 63) == 0) wg_store_rel(&lds_hprod[w], (uint32_t)(c + 1));
            }
            if (status) atomicOr(&a.status[i], status);
            return;
        }
        if constexpr (NPN || NRN) {
            // ---- numpy streams with noise (header): H owns the env stream (RN: every position evaluated as a draw's start; else:
            // the start-state queue).  The state space's stream (transition noise) is the O2 wave's: H was the long chain
            // (68 vector instructions per wave and step at 66 % of the issue cycles, profiles/r04_cfg2_noise_sq.txt)
            Pcg64 g;
            g.load(a.env_s, a.env_inc, i);
            Pcg64LimbsLo ge;
            ge.from(g);
            uint32_t hq = 0;                        // RN: positions made
            uint64_t look = 0;                      // RN: the word of position hq (the generator runs one word ahead)
            if (NRN) look = ge.next64();
            uint32_t vals = 0, tail = 0, spins = 0; // !RN: the start-state queue (as on quiet handles)
            // start state of a reset whose word is r: #{j : ceil(cdf[j] 2^53) <= r >> 11}
            auto start_of = [&](uint64_t r) __attribute__((always_inline)) -> uint32_t {
#ifdef MDPP_ABL_NP_NOSS
                return (uint32_t)(r >> 61) % 6u;
#endif
                uint32_t s0 = lds_sstab[(uint32_t)(r >> 53)];
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(s0 == 0xFFu) != 0, 0)) {
                    if (s0 == 0xFFu) {
                        const uint64_t m = r >> 11;
                        s0 = 0;
#pragma unroll
                        for (int j = 0; j < 8; j++) s0 += (a.init_thr[j] <= m) ? 1u : 0u;
                    }
                }
                return s0;
            };
            for (;;) {
                if (wg_load_acq(&lds_done) == kBlock / 64) break;
                bool did = false;
                if constexpr (NRN) {
                    const uint32_t epos = __hip_atomic_load(&lds_epos[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const bool go = hq + (uint32_t)kXB <= epos + (uint32_t)kXR;
                    if (__builtin_amdgcn_ballot_w64(go) != 0) {
                        // The generator runs ONE word ahead (`look` = the word of position hq): the uniform a wedge point of
                        // position p takes is word p + 1, which the batch has in hand -- no copy of the generator, no saved states
                        uint32_t rej = 0;
                        uint64_t wdv[kXB + 1];
                        if (go) {
                            wdv[0] = look;
#pragma unroll
                            for (int u = 0; u < kXB; u++) {
                                const uint64_t wd = wdv[u];
                                wdv[u + 1] = ge.next64();
                                const uint64_t rabs = (wd >> 9) & 0x000fffffffffffffULL;
                                if constexpr (Z0) {                 // the accept decision alone (an integer compare); no value
                                    const uint64_t ki = lds_kw[(uint32_t)wd & 0xffu].x;
                                    const bool ok = rabs < ki;
                                    rej |= ok ? 0u : (1u << u);
                                    lds_meta[(hq + (uint32_t)u) & (uint32_t)(kXR - 1)][l] = (uint16_t)(start_of(wd) | (ok ? 0u : (2u << 3)));
                                    continue;
                                }
                                const ulonglong2 kw = lds_kw[(uint32_t)wd & 0xffu];
                                // rabs * wi in ONE rounding (as in mdpp_continuous_fast.hip's walker): m = 1 + rabs 2^-52 as bits,
                                // W = wi 2^52 from the table, fma(m, W, -W) = round(rabs wi); the sign (bit 8 of the word) by a bit-field insert
                                const double W = __longlong_as_double((long long)kw.y);
                                const double x = __builtin_fma(__longlong_as_double((long long)(rabs | 0x3FF0000000000000ULL)), W, -W);
                                const uint64_t xb = (uint64_t)__double_as_longlong(x);
                                const double xs = __longlong_as_double((long long)(((uint64_t)(((uint32_t)(xb >> 32) & 0x7FFFFFFFu) |
                                                                                   (((uint32_t)wd << 23) & 0x80000000u)) << 32) | (uint32_t)xb));
#ifdef MDPP_ABL_NP_NOSLOW
                                const bool ok = rabs < kw.x + 0x7fffffffffffffffULL;
#else
                                const bool ok = rabs < kw.x;
#endif
                                rej |= ok ? 0u : (1u << u);
                                const uint32_t slot = (hq + (uint32_t)u) & (uint32_t)(kXR - 1);
                                lds_x[slot][l] = xs;
                                lds_meta[slot][l] = (uint16_t)(start_of(wd) | (ok ? 0u : (2u << 3)));   // (rejected: patched below)
                            }
                            look = wdv[kXB];
                        }
                        // the ziggurat's slow path for the rejected words of the batch
                        while (__builtin_amdgcn_ballot_w64(rej != 0u) != 0) {
                            if (rej != 0u) {
                                const uint32_t j = (uint32_t)__builtin_ctz(rej);
                                rej &= rej - 1u;
                                uint64_t wd = 0, wn = 0;            // the rejected word and the one behind it
#pragma unroll
                                for (int u = 0; u < kXB; u++)
                                    if (j == (uint32_t)u) { wd = wdv[u]; wn = wdv[u + 1]; }
                                const uint32_t idx = (uint32_t)wd & 0xffu;
                                const uint32_t slot = (hq + j) & (uint32_t)(kXR - 1);
                                const uint32_t ss = lds_meta[slot][l] & 7u;
                                if (__builtin_expect(idx == 0u, 0)) {   // tail: two uniforms per try (np_zig_tail), 3 in 10^4 draws
                                    const double nor_r = 3.6541528853610087963519472518, nor_inv_r = 0.27366123732975827203338247596;
                                    // its words: what the batch still holds behind position j, then a copy of the generator
                                    Pcg64LimbsLo t = ge;
                                    uint32_t have = (uint32_t)kXB - j, cnt = 1u;
                                    auto word = [&]() -> uint64_t {
                                        uint64_t r = 0;
                                        if (have != 0u) {
#pragma unroll
                                            for (int u = 1; u <= kXB; u++) if ((uint32_t)kXB - have + 1u == (uint32_t)u) r = wdv[u];
                                            have -= 1u;
                                        } else {
                                            r = t.next64();
                                        }
                                        cnt += 1u;
                                        return r;
                                    };
                                    double val = 0.0;
                                    for (;;) {
                                        const double u1 = (double)(word() >> 11) * (1.0 / 9007199254740992.0);
                                        const double u2 = (double)(word() >> 11) * (1.0 / 9007199254740992.0);
                                        const double xx = -nor_inv_r * log1p(-u1);
                                        const double yy = -log1p(-u2);
                                        if (yy + yy > xx * xx) { val = ((wd >> 17) & 0x1) ? -(nor_r + xx) : nor_r + xx; break; }
                                        if (cnt > 250u) { status |= kStatusInternal; break; }
                                    }
                                    if constexpr (!Z0) lds_x[slot][l] = val;
                                    // (bits 5-7: the start state of the word BEHIND the tail's words -- E must not have to
                                    //  find it in the ring, a long tail reaches beyond the window H keeps ahead of E)
                                    const uint32_t words = cnt, ssb = start_of(word());
                                    lds_meta[slot][l] = (uint16_t)(ss | (3u << 3) | (ssb << 5) | (words << 8));
                                } else {                    // wedge: one uniform; a rejected point starts the draw over two words on
                                    double x;
                                    if constexpr (Z0) {             // (formed here, where the decision needs it: |x| suffices, it is squared)
                                        const double W = __longlong_as_double((long long)lds_kw[idx].y);
                                        x = __builtin_fma(__longlong_as_double((long long)(((wd >> 9) & 0x000fffffffffffffULL) | 0x3FF0000000000000ULL)), W, -W);
                                    } else {
                                        x = lds_x[slot][l];
                                    }
                                    const double u1 = (double)(wn >> 11) * (1.0 / 9007199254740992.0);
                                    const double y = (lds_fi[idx - 1] - lds_fi[idx]) * u1 + lds_fi[idx];
                                    // exp(-x^2 / 2) by a float32 estimate (relative error < 2e-6); float64 exp() within 1e-5 of it
                                    const double e = (double)__builtin_amdgcn_exp2f((float)(x * x * -0.72134752044448170368));
                                    bool acc = y < e;
                                    if (__builtin_expect(!(fabs(y - e) > 1.0e-5 * e), 0)) acc = y < exp(-0.5 * x * x);
                                    lds_meta[slot][l] = (uint16_t)(ss | ((acc ? 1u : 2u) << 3));
                                }
                            }
                        }
                        if (go) hq += (uint32_t)kXB;
                        __hip_atomic_store(&lds_hhead[l], hq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        did = true;
                    }
                } else if (ar) {
                    // (transition noise only: resets draw their start states from the env stream ahead of need, as without noise)
                    const uint32_t head = wg_load_acq(&lds_head[l]);
                    const uint32_t cnt = tail - head;
                    const bool want = cnt + 1u <= 8u;
                    const uint64_t bw = __builtin_amdgcn_ballot_w64(want);
                    const bool urgent = __builtin_amdgcn_ballot_w64(want && cnt <= 2) != 0;
                    if (__builtin_popcountll(bw) >= kMinLanes || urgent) {
                        Pcg64LimbsLo n = ge;
                        const uint32_t s0 = start_of(n.next64()) | 8u;
                        if (want) {
                            const uint32_t sh = (tail & 7u) * 4u;
                            ge = n;
                            vals = (vals & ~(0xFu << sh)) | (s0 << sh);
                            tail += 1u;
                        }
                        __hip_atomic_store(&lds_ring[l], (uint64_t)vals | ((uint64_t)tail << 32), __ATOMIC_RELEASE,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                        did = true;
                    }
                }
                if (__builtin_amdgcn_ballot_w64(did) == 0) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > kSpinLimit * 4u) { status |= kStatusInternal; break; }
                } else {
                    spins = 0;
                }
            }
            // un-draw what was made and not used: s_prev = (s - inc) * M^-1 (mod 2^128)
            auto undraw = [&](Pcg64 &gg, uint32_t q) {
                for (; q > 0; q--) {
                    uint64_t lo = gg.s_lo - gg.inc_lo;
                    uint64_t hi = gg.s_hi - gg.inc_hi - (gg.s_lo < gg.inc_lo ? 1ULL : 0ULL);
                    gg.s_lo = lo * a.minv_lo;
                    gg.s_hi = __umul64hi(lo, a.minv_lo) + lo * a.minv_hi + hi * a.minv_lo;
                }
            };
            ge.to(g);
            if (NRN) undraw(g, hq + 1u - __hip_atomic_load(&lds_epos[l], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));   // (+ the word in hand)
            else undraw(g, tail - wg_load_acq(&lds_head[l]));
            g.store(a.env_s, i);
            if (status) atomicOr(&a.status[i], status);
            return;
        }
        Pcg64 g;
        g.load(a.env_s, a.env_inc, i);
#if MDPP_LEAN_H_LIMBS
        // (the hand-scheduled limb form of the generator: 31 instead of 46 vector instructions per word on the filler wave)
        Pcg64LimbsLo gl;
        gl.from(g);
        auto draw = [&](Pcg64LimbsLo &gg) -> uint32_t {
#else
        auto draw = [&](Pcg64 &gg) -> uint32_t {
#endif
            const uint64_t r0 = gg.next64();
            const uint64_t m = r0 >> 11;
            uint32_t s0 = 0;
            if constexpr (HSS) {                                 // (the bucket's answer; 0xFF = a threshold inside it: count)
                s0 = lds_sstab[(uint32_t)(r0 >> 53)];
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(s0 == 0xFFu) != 0, 0)) {
                    if (s0 == 0xFFu) {
                        s0 = 0;
#pragma unroll
                        for (int j = 0; j < 8; j++) s0 += (lds_T[j] <= m) ? 1u : 0u;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) s0 += (lds_T[j] <= m) ? 1u : 0u;
            }
            if (IRR) {                                           // relevant, then irrelevant, like reset() (:2255-2264)
                const uint64_t m1 = gg.next64() >> 11;
                uint32_t s1 = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) s1 += (lds_T1[j] <= m1) ? 1u : 0u;
                s0 |= (s1 | 8u) << 4;
            }
            return s0;
        };
        uint32_t vals = 0, tail = 0;
        for (;;) {
            if (wg_load_acq(&lds_done) == kBlock / 64) break;
            const uint32_t head = wg_load_acq(&lds_head[l]);
            const uint32_t cnt = tail - head;
            const bool want = cnt + (uint32_t)kEN <= 8u;
            const uint64_t bw = __builtin_amdgcn_ballot_w64(want);
            const bool urgent = __builtin_amdgcn_ballot_w64(want && cnt <= 2) != 0;
            if (__builtin_popcountll(bw) >= kMinLanes || urgent) {
#if MDPP_LEAN_H_LIMBS
                Pcg64LimbsLo n = gl;
#else
                Pcg64 n = g;
#endif
                const uint32_t s0 = draw(n) | 8u;
                if (want) {
                    const uint32_t sh = (tail & 7u) * 4u;
#if MDPP_LEAN_H_LIMBS
                    gl = n;
#else
                    g = n;
#endif
                    vals = (vals & ~((IRR ? 0xFFu : 0xFu) << sh)) | (s0 << sh);
                    tail += (uint32_t)kEN;                       // (in nibbles = in draws of the stream)
                }
                __hip_atomic_store(&lds_ring[l], (uint64_t)vals | ((uint64_t)tail << 32), __ATOMIC_RELEASE,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                __builtin_amdgcn_s_sleep(MDPP_LEAN_HSLEEP);
            }
        }
        // un-draw what the env lane did not take: s_prev = (s - inc) * M^-1 (mod 2^128)
#if MDPP_LEAN_H_LIMBS
        gl.to(g);
#endif
        const uint32_t head = wg_load_acq(&lds_head[l]);
        for (uint32_t q = tail - head; q > 0; q--) {
            uint64_t lo = g.s_lo - g.inc_lo;
            uint64_t hi = g.s_hi - g.inc_hi - (g.s_lo < g.inc_lo ? 1ULL : 0ULL);
            g.s_lo = lo * a.minv_lo;
            g.s_hi = __umul64hi(lo, a.minv_lo) + lo * a.minv_hi + hi * a.minv_lo;
        }
        g.store(a.env_s, i);
        return;
    }

    const uint32_t total = (uint32_t)K * N;
    const int nchunks = (K + kChunk - 1) / kChunk;

    // =============================================================== O1: reward path
    if (role == 1) {
#ifdef MDPP_ABL_NOO1
        return;
#endif
        __builtin_amdgcn_s_setprio(kPrioO);
        auto r_rew = __builtin_amdgcn_make_buffer_rsrc((void *)reward, 0, total * 4u, kPRsrc);
        const uint32_t v4 = i * 4u;
        const uint32_t dsh = (uint32_t)(a.delay > 0 ? a.delay - 1 : 0);
        uint32_t ring = ((const uint32_t *)&a.state[i])[3];
        const uint8_t *rselb = (const uint8_t *)lds_rsel;

        // reward_every_n_steps (:1975-1976): 16 x (steps to the next pay step), the row index of the reward-value table
        const uint32_t ph_full = 16u * (uint32_t)a.every_n;
        uint32_t ph = EVN ? ph_full - 16u * (steps_at_launch % (uint32_t)a.every_n) : 0u;
        const uint64_t genv = (uint64_t)(a.env_id_offset + (int64_t)i);
        const uint32_t r4 = (uint32_t)ptick0 & 3u;
        float zc[kChunk] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};         // RN: the chunk's reward normals
        const bool plain = a.scale == 1.0 && a.shift == 0.0;                        // (wave-uniform)
        auto emit = [&](uint32_t ra, uint32_t rb, uint32_t rc, uint32_t so, double z, float *stage) {
            uint32_t bit = lds_V[(ra >> 5) & 2047u] >> (ra & 31u);                   // reward bit, NaN-gated (:1822)
            uint32_t out;
            if (DELAY) {                                                             // FIFO (:1970-1973)
                out = (ring >> dsh) & 1u;
                ring = (ring << 1) | (bit & 1u);
                ring = (ra != rb) ? 0u : ring;                                       // reset clears it (:2250)
            } else {
                out = bit & 1u;
            }
            const uint32_t tb = (rc >> 7) & 1u;
            uint32_t rd = 0;
            if (EVN) {
                ph -= 16u;
                rd = ph;
                ph = (ra != rb || ph == 0u) ? ph_full : ph;
            }
            float rout;
            if constexpr (RN && !Z0) {                                               // :1975-1990, :2107 in float64
                double r = (out != 0u && (!EVN || rd == 0u)) ? 1.0 : 0.0;
                r += 0.0 + a.r_noise * z;
                if (!plain) {                // (scale 1, shift 0: r * 1.0 and r + 0.0 are r -- it is never -0.0 here)
                    r *= a.scale;
                    r += a.shift;
                }
                if (tb) r += a.term_add;
                rout = (float)r;
            } else {
                rout = EVN ? *(const float *)(rselb + (rd | (out << 3) | (tb << 2)))
                           : *(const float *)(rselb + (((out << 1) | tb) << 2));
            }
            if (nextmode) rout = (ra != rb) ? 0.0f : rout;                           // the reset call returns reward 0
            if (stage) { *stage = rout; return; }
#if defined(MDPP_ABL_NOSTORE) || defined(MDPP_ABL_NOREW)
            status ^= __float_as_uint(rout) & 0x100u;
#else
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rout), r_rew, v4, so * 4u, MDPP_LEAN_ST_AUX);
#endif
        };
        for (int c = 0; c < nchunks; c++) {
            const int kbase = c * kChunk;
            const uint32_t upto = (uint32_t)min(kbase + kChunk, K);
            if constexpr (RN && PHILOX)       // (before the wait: independent of E)
                chunk_normals(a.philox_seed, genv, (ptick0 + (uint64_t)kbase) >> 2, r4, kPhiloxRNoiseStream, zc);
            uint32_t spins = 0;
#ifdef MDPP_ABL_FREEO
            while (false) {
#else
            while (wg_load_acq(&lds_prod[w]) < upto) {
#endif
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit) { status |= kStatusInternal; break; }
            }
            const bool rowc = ROWS1 && rows && kbase + kChunk <= K;                  // (wave-uniform)
            if (kbase + kChunk <= K) {
                uint32_t ra[kChunk], rb[kChunk], rc[kChunk];
#pragma unroll
                for (int u = 0; u < kChunk; u++) {
                    ra[u] = lds_rec[0][(kbase + u) % KD][l];
                    rb[u] = lds_rec[1][(kbase + u) % KD][l];
                    rc[u] = lds_rec[2][(kbase + u) % KD][l];
                }
                double zd[kChunk];
#pragma unroll
                for (int u = 0; u < kChunk; u++) zd[u] = NRX ? lds_rx[NRX ? (kbase + u) % KD : 0][l] : (double)zc[u];    // (numpy streams: the normal E found)
                if (rowc && c >= kRB) {                     // the staging rows of chunk c - kRB: stored by all four O2 waves
                    uint32_t sp2 = 0;
                    while (min4(lds_rcons) < (uint32_t)(c - kRB + 1)) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++sp2 > kSpinLimit) { status |= kStatusInternal; break; }
                    }
                }
#pragma unroll
                for (int u = 0; u < kChunk; u++)
                    emit(ra[u], rb[u], rc[u], (uint32_t)(kbase + u) * N, zd[u], rowc ? &lds_rw[ROWS1 ? (c % kRB) : 0][u][ROWS1 ? l : 0] : nullptr);
            } else {
#pragma unroll
                for (int u = 0; u < kChunk; u++)
                    if (kbase + u < K)
                        emit(lds_rec[0][(kbase + u) % KD][l], lds_rec[1][(kbase + u) % KD][l],
                             lds_rec[2][(kbase + u) % KD][l], (uint32_t)(kbase + u) * N,
                             NRX ? lds_rx[NRX ? (kbase + u) % KD : 0][l] : (double)zc[u], nullptr);
            }
            if ((l & 63) == 0) wg_store_rel(&lds_cons[w][0], upto);
            if constexpr (ROWS1) {
                if (rowc && (l & 63) == 0) wg_store_rel(&lds_rprod[w], (uint32_t)(c + 1));
            }
        }
        ((uint32_t *)&a.state[i])[3] = ring;
        if (status) atomicOr(&a.status[i], status);
        return;
    }

    // =============================================================== O2: observation, terminated, truncated
    if (role == 2) {
#ifdef MDPP_ABL_NOO2
        return;
#endif
        // (numpy transition noise: this wave runs the state space's generator -- a long stage like H)
        // (3 beside an H wave that only keeps the start-state queue, 2 beside one that evaluates the env stream: 159 / 306 us per
        //  cfg2 launch with transition noise / both noises, against 202 / 313 at the O waves' priority)
        __builtin_amdgcn_s_setprio(MDPP_LEAN_PRIO_O2 >= 0 ? MDPP_LEAN_PRIO_O2 : (NPN && !MDPP_LEAN_PRIO_FORCED) ? (NRN ? 2 : 3)
                                   : (PN && PHILOX && MDPP_LEAN_PHILOX_PN_O2 && !MDPP_LEAN_PRIO_FORCED) ? (RN ? 2 : 3) : kPrioO);
        auto r_obs = __builtin_amdgcn_make_buffer_rsrc(obs, 0, total * (OBS64 ? 8u : 4u) * (uint32_t)kEN, kPRsrc);
        auto r_term = __builtin_amdgcn_make_buffer_rsrc((void *)term, 0, total, kPRsrc);
        auto r_rew2 = __builtin_amdgcn_make_buffer_rsrc((void *)reward, 0, total * 4u, kPRsrc);
        auto r_trunc = __builtin_amdgcn_make_buffer_rsrc((void *)trunc, 0, total, kPRsrc);
        const uint32_t v1 = i, v4 = i * 4u, v8 = i * 8u, v16 = i * 16u;
        auto emit = [&](uint32_t rb, uint32_t rc, uint32_t so) {
            const uint32_t o = rb & 7u;
#ifdef MDPP_ABL_NOSTORE
            status ^= (o + rc) & 0x100u;
            return;
#endif
#ifdef MDPP_ABL_NOOBS
            status ^= o & 0x100u;
#else
            if (IRR) {
                const uint32_t o1 = (rc >> 8) & 7u;
                // (128-bit stores: the whole offset in the VGPR, see the store-data hazard note in mdpp_discrete_quiet.hip)
                if (OBS64) __builtin_amdgcn_raw_buffer_store_b128(u32x4{o, 0u, o1, 0u}, r_obs, v16 + so * 16u, 0, MDPP_LEAN_ST_AUX);
                else __builtin_amdgcn_raw_buffer_store_b64(u32x2{o, o1}, r_obs, v8, so * 8u, MDPP_LEAN_ST_AUX);
            } else if (OBS64) __builtin_amdgcn_raw_buffer_store_b64(u32x2{o, 0u}, r_obs, v8, so * 8u, MDPP_LEAN_ST_AUX);
            else __builtin_amdgcn_raw_buffer_store_b32(o, r_obs, v4, so * 4u, MDPP_LEAN_ST_AUX);
#endif
#ifdef MDPP_ABL_NOBYTES
            status ^= rc & 0x100u;
#else
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)((rc >> 7) & 1u), r_term, v1, so, MDPP_LEAN_ST_AUX_BYTES);
            __builtin_amdgcn_raw_buffer_store_b8(HASMAX ? (uint8_t)(rc >> 16) : (uint8_t)0, r_trunc, v1, so, MDPP_LEAN_ST_AUX_BYTES);
#endif
        };
        // numpy transition noise (header): this wave owns the state space's stream and makes the noise bytes of chunk pc --
        // one word per step -- up to kHChunksNp chunks ahead of the chunk E has finished (slot pc % kHChunksNp is free then)
        Pcg64 sp;
        Pcg64LimbsLo gs;
        int pc = 0;
        if constexpr (NPN) { sp.load(a.sp_s, a.sp_inc, i); gs.from(sp); }
        auto make_pn = [&](int upto_chunk) __attribute__((always_inline)) {
            for (; pc < nchunks && pc < upto_chunk; pc++) {
                uint32_t pk[2] = {0u, 0u};
#pragma unroll
                for (int u = 0; u < kChunk; u++) {
                    const uint64_t r = gs.next64();
                    uint32_t by = lds_pntab[(uint32_t)(r >> 52)];
#ifdef MDPP_ABL_NP_NOPNC
                    by = 7u | ((uint32_t)(r >> 63) << 4);
#endif
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(by == 0xFFu) != 0, 0)) {
                        if (by == 0xFFu) {
                            uint32_t na = 0, nb = 0;
#pragma unroll
                            for (int j = 0; j < 7; j++) na += (a.pn_TL[j] <= r) ? 1u : 0u;
#pragma unroll
                            for (int j = 0; j < 8; j++) nb += (a.pn_TU[j] <= r) ? 1u : 0u;
                            by = na | (nb << 4);
                        }
                    }
                    pk[u >> 2] |= by << (8 * (u & 3));
                }
                lds_pn2[pc % kHChunksNp][0][l] = pk[0];
                lds_pn2[pc % kHChunksNp][1][l] = pk[1];
                if ((l & 63) == 0) wg_store_rel(&lds_hprod[w], (uint32_t)(pc + 1));
            }
        };
        // Philox transition noise: the chunk's eight nibbles (noisy << 3 | index among the other states) are a function of
        // (seed, env, tick) alone -- made here, up to kHChunks chunks ahead of the chunk E has finished, instead of on the
        // H wave, whose chain (start states AND noise words: four Philox blocks per chunk) set the pace
        constexpr bool PPN = PN && PHILOX && MDPP_LEAN_PHILOX_PN_O2;
        const uint64_t genv2 = (uint64_t)(a.env_id_offset + (int64_t)i);
        int ppc = 0;
        auto make_ppn = [&](int upto_chunk) __attribute__((always_inline)) {
            for (; ppc < nchunks && ppc < upto_chunk; ppc++) {
                uint32_t wp[kChunk], pn = 0;
                chunk_words(a.philox_seed, genv2, (ptick0 + (uint64_t)(ppc * kChunk)) >> 2, (uint32_t)ptick0 & 3u, kPhiloxPNoiseStream, wp);
#pragma unroll
                for (int u = 0; u < kChunk; u++) {
                    const uint32_t e = philox_pnoise_index(wp[u], a.pn_T, a.pn_M);
                    pn |= ((e & 7u) | ((e >> 5) & 8u)) << (4 * u);
                }
                lds_pn[ppc % kHChunks][l] = pn;
                if ((l & 63) == 0) wg_store_rel(&lds_pprod[w], (uint32_t)(ppc + 1));
            }
        };
        if constexpr (PPN) make_ppn(kHChunks);
        if constexpr (NPN) make_pn(kHChunksNp);
        for (int c = 0; c < nchunks; c++) {
            const int kbase = c * kChunk;
            const uint32_t upto = (uint32_t)min(kbase + kChunk, K);
            uint32_t spins = 0;
#ifdef MDPP_ABL_FREEO
            while (false) {
#else
            while (wg_load_acq(&lds_prod[w]) < upto) {
#endif
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit) { status |= kStatusInternal; break; }
            }
            if constexpr (NPN) make_pn(c + 1 + kHChunksNp);          // (E is through chunk c)
            if constexpr (PPN) make_ppn(c + 1 + kHChunks);
            if (ROWS2 && rows && kbase + kChunk <= K) {
                // rows w and w + 4 of the chunk, the block's 256 envs (header "whole-row stores"): all four E waves are through it
                uint32_t sp2 = 0;
                while (min4(lds_prod) < upto) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++sp2 > kSpinLimit) { status |= kStatusInternal; break; }
                }
                if constexpr (ROWS1) {              // ... and all four O1 waves have staged its rewards
                    while (min4(lds_rprod) < (uint32_t)(c + 1)) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++sp2 > kSpinLimit) { status |= kStatusInternal; break; }
                    }
                }
#pragma unroll
                for (int h = 0; h < kChunk / 4; h++) {
                    const uint32_t kk = (uint32_t)kbase + ws + 4u * (uint32_t)h, slot = kk % (uint32_t)KD;
#if !defined(MDPP_ABL_NOSTORE) && !defined(MDPP_ABL_NOREW)
                    if constexpr (ROWS1) {          // (whole offset in the VGPR: mdpp_discrete_quiet.hip's hazard note)
                        const u32x4 v = *(const u32x4 *)&lds_rw[c % kRB][ws + 4u * (uint32_t)h][4u * ln];
                        __builtin_amdgcn_raw_buffer_store_b128(v, r_rew2, (blk0 + 4u * ln) * 4u + kk * N * 4u, 0, MDPP_LEAN_ST_AUX);
                    }
#endif
#ifdef MDPP_ABL_NOSTORE
                    status ^= lds_rec[1][slot][l] & 0x100u;
                    continue;
#endif
#ifndef MDPP_ABL_NOOBS
                    if (OBS64) {                    // lane ln: envs 2 ln, 2 ln + 1 and 128 + 2 ln, 129 + 2 ln -- each store 1 KiB in one piece
                        const u32x2 b0 = *(const u32x2 *)&lds_rec[1][slot][2u * ln];
                        const u32x2 b1 = *(const u32x2 *)&lds_rec[1][slot][128u + 2u * ln];
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{b0.x & 7u, 0u, b0.y & 7u, 0u}, r_obs, (blk0 + 2u * ln) * 8u + kk * N * 8u, 0, MDPP_LEAN_ST_AUX);
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{b1.x & 7u, 0u, b1.y & 7u, 0u}, r_obs, (blk0 + 128u + 2u * ln) * 8u + kk * N * 8u, 0, MDPP_LEAN_ST_AUX);
                    } else {
                        const u32x4 b = *(const u32x4 *)&lds_rec[1][slot][4u * ln];
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{b.x & 7u, b.y & 7u, b.z & 7u, b.w & 7u}, r_obs, (blk0 + 4u * ln) * 4u + kk * N * 4u, 0, MDPP_LEAN_ST_AUX);
                    }
#endif
#ifndef MDPP_ABL_NOBYTES
                    const u32x4 c4 = *(const u32x4 *)&lds_rec[2][slot][4u * ln];
                    // byte j of the flag words = the flag of env 4 ln + j: bit 7 of byte 0 / byte 2 of its record
                    const uint32_t t01 = __builtin_amdgcn_perm(c4.y, c4.x, 0x0c0c0400u), t23 = __builtin_amdgcn_perm(c4.w, c4.z, 0x04000c0cu);
                    const uint32_t tw = ((t01 | t23) >> 7) & 0x01010101u;
                    __builtin_amdgcn_raw_buffer_store_b32(tw, r_term, blk0 + 4u * ln, kk * N, MDPP_LEAN_ST_AUX_BYTES);
                    uint32_t uw = 0u;
                    if (HASMAX) uw = __builtin_amdgcn_perm(c4.y, c4.x, 0x0c0c0602u) | __builtin_amdgcn_perm(c4.w, c4.z, 0x06020c0cu);
                    __builtin_amdgcn_raw_buffer_store_b32(uw, r_trunc, blk0 + 4u * ln, kk * N, MDPP_LEAN_ST_AUX_BYTES);
#endif
                }
            } else if (kbase + kChunk <= K) {
                uint32_t rb[kChunk], rc[kChunk];
#pragma unroll
                for (int u = 0; u < kChunk; u++) {
                    rb[u] = lds_rec[1][(kbase + u) % KD][l];
                    rc[u] = lds_rec[2][(kbase + u) % KD][l];
                }
#pragma unroll
                for (int u = 0; u < kChunk; u++) emit(rb[u], rc[u], (uint32_t)(kbase + u) * N);
            } else {
                for (int k = kbase; k < K; k++) emit(lds_rec[1][k % KD][l], lds_rec[2][k % KD][l], (uint32_t)k * N);
            }
            if ((l & 63) == 0) {
                wg_store_rel(&lds_cons[w][1], upto);
                if (ROWS1) wg_store_rel(&lds_rcons[w], (uint32_t)(c + 1));
            }
        }
        if constexpr (NPN) {                        // un-draw the words of a ragged last chunk: s_prev = (s - inc) * M^-1 (mod 2^128)
            gs.to(sp);
            for (uint32_t q = (uint32_t)(pc * kChunk) - (uint32_t)K; q > 0; q--) {
                uint64_t lo = sp.s_lo - sp.inc_lo;
                uint64_t hi = sp.s_hi - sp.inc_hi - (sp.s_lo < sp.inc_lo ? 1ULL : 0ULL);
                sp.s_lo = lo * a.minv_lo;
                sp.s_hi = __umul64hi(lo, a.minv_lo) + lo * a.minv_hi + hi * a.minv_lo;
            }
            sp.store(a.sp_s, i);
        }
        if (status) atomicOr(&a.status[i], status);
        return;
    }

    // =============================================================== E: state recurrence
    __builtin_amdgcn_s_setprio(kPrioE);   // the serial recurrence is the critical path; H is filler work
    // hist: bytes newest first, 0xFF = NaN  ->  nibbles newest first, bit 3 = is a state
    uint32_t k2, qv, cnt, steps0, last_reset = 0, badq[2] = {0u, 0u};
    bool pend = false;                              // next-step mode: the episode ended on the previous step
    {
        uint4 st = a.state[i];
        k2 = 0;
        for (int j = 3; j >= 0; j--) {
            const uint32_t b = (st.x >> (8 * j)) & 0xFFu;
            k2 = (k2 << 4) | (b == 0xFFu ? 0u : ((b & 7u) | 8u));
        }
        const bool own_q = !PHILOX && !IRR && !nextmode && NZ == 0;   // word 1 of the state is this kernel's draw queue (fast_ok handles)
        const uint32_t qc = own_q ? (st.y >> 24) & 7u : 0u;
        qv = own_q ? (st.y & 0x00777777u) | (0x00888888u & ((1u << (4u * qc)) - 1u)) : 0u;
        steps0 = st.z & 0x7FFFFFFFu;
        pend = nextmode && (st.z >> 31) != 0u;
        const uint32_t ms = (uint32_t)a.max_steps;
        cnt = HASMAX ? (0x10000u - ms) + (steps0 < ms ? steps0 : ms) : 0u;
    }
    const uint32_t c0 = HASMAX ? 0x10000u - (uint32_t)a.max_steps : 0u;
    auto r_act = __builtin_amdgcn_make_buffer_rsrc((void *)actions, 0, total * 4u * (uint32_t)kEN, kPRsrc);
    const uint32_t v4 = i * 4u * (uint32_t)kEN;
    uint32_t head_local = 0;
    S0Word s0c = 0;
    uint32_t sel = (k2 & 7u) | kSelPad;
    uint32_t c1 = IRR ? (a.irr_state[i] & 7u) : 0u;                 // the irrelevant part of curr_state
    const uint32_t A1 = IRR ? (uint32_t)a.A1 : 1u;

    uint32_t pnc = 0, pnc_hi = 0;                   // PN: this chunk's transition-noise nibbles (numpy streams: bytes a | b << 4)
    // NRN: the env stream by position (header).  ep = position of the next draw; (m0, m1, m2, xv) = meta[ep .. ep + 2], x[ep],
    // fetched at the end of the previous step; hh = positions H has made, as last read
    uint32_t ep = 0, hh = 0, m0 = 0, m1 = 0, m2 = 0;
    double xv = 0.0;
    auto ensure = [&](uint32_t upto) __attribute__((always_inline)) {       // positions < upto are made
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(upto > hh) != 0, 0)) {
            uint32_t spins = 0;
            hh = __hip_atomic_load(&lds_hhead[l], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__builtin_amdgcn_ballot_w64(upto > hh) != 0) {
                __hip_atomic_store(&lds_epos[l], ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);    // (H's window follows E)
                __builtin_amdgcn_s_sleep(1);
                hh = __hip_atomic_load(&lds_hhead[l], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (++spins > kSpinLimit) { status |= kStatusInternal; hh = upto; break; }
            }
        }
    };
    auto fetch = [&]() __attribute__((always_inline)) {
        ensure(ep + 3u);
        m0 = lds_meta[ep & (uint32_t)(kXR - 1)][l];
        m1 = lds_meta[(ep + 1u) & (uint32_t)(kXR - 1)][l];
        m2 = lds_meta[(ep + 2u) & (uint32_t)(kXR - 1)][l];
        if constexpr (NRX) xv = lds_x[ep & (uint32_t)(kXR - 1)][l];
    };
    if constexpr (NRN) fetch();
    // PN: byte s of {enc_lo, enc_hi} = s | 8 | is_terminal[s] << 7, the column byte of a re-drawn state
    uint32_t enc_lo = 0, enc_hi = 0;
    if constexpr (PN) {
        uint64_t enc = 0;
        for (uint32_t s_ = 0; s_ < 8u; s_++) enc |= (uint64_t)(s_ | 8u | ((uint32_t)((a.term_mask >> s_) & 1ULL) << 7)) << (8 * s_);
        enc_lo = (uint32_t)enc; enc_hi = (uint32_t)(enc >> 32);
    }
    auto pull = [&](int c) {
        if (!ar && !PN) return;                     // (no resets: nothing is drawn, the H lanes have left)
        if (NPN && c >= 0) {                        // this chunk's transition-noise bytes, made by the H wave from the space stream
            uint32_t spins = 0;
            while (wg_load_acq(&lds_hprod[w]) < (uint32_t)(c + 1)) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit) { status |= kStatusInternal; break; }
            }
            pnc = lds_pn2[c % kHChunksNp][0][l];
            pnc_hi = lds_pn2[c % kHChunksNp][1][l];
        }
        if constexpr (NRN) return;                  // (start states come with the positions)
        if (!PHILOX && !ar) return;
        if constexpr (PHILOX) {                     // this chunk's start states, made by the H wave
            uint32_t spins = 0;
            while (wg_load_acq(&lds_hprod[w]) < (uint32_t)(c + 1)) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit) { status |= kStatusInternal; break; }
            }
            s0c = lds_s0[c % kHChunks][l];
            if constexpr (PN) {
                if (MDPP_LEAN_PHILOX_PN_O2) {
                    spins = 0;
                    while (wg_load_acq(&lds_pprod[w]) < (uint32_t)(c + 1)) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > kSpinLimit) { status |= kStatusInternal; break; }
                    }
                }
                pnc = lds_pn[c % kHChunks][l];
            }
            return;
        }
        const uint64_t rt = __hip_atomic_load(&lds_ring[l], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t vals = (uint32_t)rt, tail = (uint32_t)(rt >> 32);
        const uint32_t qc = (uint32_t)__builtin_popcount(qv & 0x88888888u);
        const uint32_t avail = tail - head_local, room = kQueueCap - qc;       // (IRR: both even, entries are nibble pairs)
        const uint32_t take = avail < room ? avail : room;
        const uint32_t rot = __builtin_amdgcn_alignbit(vals, vals, (head_local & 7u) * 4u);
        const uint32_t m = (1u << (4u * take)) - 1u;
        qv |= (rot & m) << (4u * qc);
        head_local += take;
        __hip_atomic_store(&lds_head[l], head_local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    typedef typename std::conditional<IRR, uint2, int>::type Act;      // (action, irrelevant action)
    typedef typename std::conditional<IRR, uint4, uint2>::type Col;    // their columns
    auto column = [&](Act act, uint32_t &badm, int u) -> Col {
        int action;
        if constexpr (IRR) action = (int)act.x; else action = act;
        uint32_t ua = (uint32_t)action;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(ua >= A) != 0, 0)) {
            ua = (uint32_t)(action + ((action >> 31) & (int)A));
            const bool bad = ua >= A;
            // (next-step mode: an action the reset call ignores is not an error; decided at the step)
            if (nextmode) badm |= bad ? (1u << u) : 0u; else status |= bad ? (uint32_t)MDPP_STATUS_BAD_ACTION : 0u;
            ua = bad ? 0u : ua;
        }
        if constexpr (IRR) {
            const int action1 = (int)act.y;
            uint32_t ub = (uint32_t)action1;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(ub >= A1) != 0, 0)) {
                ub = (uint32_t)(action1 + ((action1 >> 31) & (int)A1));
                const bool bad = ub >= A1;
                if (nextmode) badm |= bad ? (1u << u) : 0u; else status |= bad ? (uint32_t)MDPP_STATUS_BAD_ACTION : 0u;
                ub = bad ? 0u : ub;
            }
            const uint2 ca = lds_col[ua], cb = lds_col1[ub];
            return make_uint4(ca.x, ca.y, cb.x, cb.y);
        } else {
            return lds_col[ua];
        }
    };
    auto stepE = [&](const Col &col, int k, uint32_t badbit) {
        uint32_t entry = __builtin_amdgcn_perm(col.y, col.x, sel);                    // D1: P[cur][a] | 8 | terminal << 7
        if constexpr (PN && PHILOX) {                                                 // D2 (:1604-1622), philox_pnoise_*
            const uint32_t nib = (pnc >> (4 * (k % kChunk))) & 0xFu, j = nib & 7u, nx = entry & 7u;
            const uint32_t ns = nib > 7u ? j + (j >= nx ? 1u : 0u) : nx;
            entry = __builtin_amdgcn_perm(enc_hi, enc_lo, ns | kSelPad);
        }
        if constexpr (NPN) {                                                          // ... on the space stream's word: min(a, n) + max(b - n, 0)
            const uint32_t by = (((k % kChunk) < 4 ? pnc : pnc_hi) >> (8 * (k & 3))) & 0xFFu, nx = entry & 7u;
            const uint32_t ns = min(by & 15u, nx) + (uint32_t)max((int)(by >> 4) - (int)nx, 0);
            entry = __builtin_amdgcn_perm(enc_hi, enc_lo, ns | kSelPad);
        }
        const uint32_t k2n = (k2 << 4) | entry;       // (bit 7 of the sum is set anyway: the nibble below was a state)
        bool need = entry > 0x7Fu;                                                    // D7: is_terminal[next]
        uint32_t rc = entry;                          // (O1 / O2 take bit 7 out)
        if (HASMAX) {
            cnt += 1;
            need = need || cnt >= 0x10000u;
            rc = __builtin_amdgcn_perm(cnt, entry, 0x0c060c00u);                       // byte 0 the entry, byte 2 truncated
        }
#ifdef MDPP_ABL_NORESET
        need = false;
#endif
        need = need && ar;
        uint32_t rec_a = k2n;
        if (nextmode) {
            const bool ended = need && !pend;
            need = pend;                                         // reset now: this call is reset() (:2250-2278) ...
            rc = pend ? 0u : rc;                                 // ... which returns no flags,
            rec_a = pend ? 0u : k2n;                             // no reward (history word 0: O1 sees a reset, pays 0.0)
            status |= (badbit != 0u && !pend) ? (uint32_t)MDPP_STATUS_BAD_ACTION : 0u;
            pend = ended;
        }
        uint32_t s0v = PHILOX ? (uint32_t)(s0c >> (kEN * 4 * (k % kChunk))) & (IRR ? 0xFFu : 0xFu) : qv & (IRR ? 0xFFu : 0xFu);
        if constexpr (NRN) {
            // this step's draws on the env stream: the normal that starts at position ep, then -- if the episode ended -- one word
            uint32_t kind = (m0 >> 3) & 3u;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(kind >= 2u) != 0, 0)) {
                uint32_t guard = 0;
                while (__builtin_amdgcn_ballot_w64(kind == 2u) != 0) {          // wedge point rejected: the draw starts over two words on
                    const bool red = kind == 2u;
                    ep += red ? 2u : 0u;
                    uint32_t o0 = m0, o1 = m1, o2 = m2;
                    double ox = xv;
                    fetch();
                    if (!red) { m0 = o0; m1 = o1; m2 = o2; xv = ox; }
                    kind = (m0 >> 3) & 3u;
                    if (++guard > 64u) { status |= kStatusInternal; break; }
                }
            }
            uint32_t cntw = kind == 0u ? 1u : 2u, ssm = kind == 0u ? m1 : m2;
            // tail: its words are counted in the high byte, the start state of the word behind them is in bits 5-7
            if (kind == 3u) { cntw = m0 >> 8; ssm = m0 >> 5; }
            if constexpr (NRX) lds_rx[k % KD][l] = xv;
            s0v = (ssm & 7u) | 8u;
            ep += cntw + (need ? 1u : 0u);
            __hip_atomic_store(&lds_epos[l], ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (H's window follows E)
            fetch();
        }
        if (!PHILOX && !NRN && __builtin_expect(__builtin_amdgcn_ballot_w64(need && s0v == 0u) != 0, 0)) {
            uint32_t spins = 0;
            while (__builtin_amdgcn_ballot_w64(need && (qv & 0xFu) == 0u) != 0) {
                pull(-1);                                        // (the queue only: not the chunk's noise bytes)
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit) { status |= kStatusInternal; qv |= 8u; break; }
            }
            s0v = qv & (IRR ? 0xFFu : 0xFu);
        }
        if constexpr (IRR) {                                     // the irrelevant sub-space steps on its own table
            const uint32_t n1 = __builtin_amdgcn_perm(col.w, col.z, c1 | kSelPad);
            c1 = need ? ((s0v >> 4) & 7u) : n1;
            rc |= c1 << 8;                                       // byte 1: the irrelevant observation
            s0v &= 0xFu;
        }
        k2 = need ? s0v : k2n;
        if (!PHILOX && !NRN) qv = need ? (qv >> (4 * kEN)) : qv;
        if (HASMAX) cnt = need ? c0 : cnt;
        else last_reset = need ? (uint32_t)(k + 1) : last_reset;
        sel = (k2 & 7u) | kSelPad;
        lds_rec[0][k % KD][l] = rec_a;
        lds_rec[1][k % KD][l] = k2;
        lds_rec[2][k % KD][l] = rc;
    };

    auto load_act = [&](int k) -> Act {
        const uint32_t kk = (uint32_t)min(k, K - 1);
        if constexpr (IRR) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r_act, v4, kk * N * 8u, MDPP_LEAN_LD_AUX);
            return make_uint2(v.x, v.y);
        } else {
#ifdef MDPP_ABL_NOLOAD
            return (int)((kk * 7u + i) & 7u);
#endif
            return __builtin_amdgcn_raw_buffer_load_b32(r_act, v4, kk * N * 4u, MDPP_LEAN_LD_AUX);
        }
    };
    // Actions are fetched kAhead chunks ahead of their use, columns one chunk ahead.  The buffers rotate
    // by NAME (the chunk loop is unrolled kAhead times): copying a register whose load is still in
    // flight makes the wave wait for it, which would put one HBM latency into every chunk.
    constexpr int kAh = IRR ? 2 : kAhead;      // (action pairs: half the depth, for registers)
    static_assert(kAh % 2 == 0, "the column double buffer alternates with the chunk index");
    Act actq[kAh][kChunk];       // slot (m - 1) % kAhead holds the actions of chunk m
    Col colq[2][kChunk];            // slot m & 1 holds the columns of chunk m
#pragma unroll
    for (int u = 0; u < kChunk; u++) colq[0][u] = column(load_act(u), badq[0], u);
#pragma unroll
    for (int q = 0; q < kAh; q++)
#pragma unroll
        for (int u = 0; u < kChunk; u++) actq[q][u] = load_act((q + 1) * kChunk + u);

    const int nfull = K / kChunk;   // full chunks; a ragged tail (K % kChunk steps) follows the loop
    auto wait_room = [&](int kend) {                // do not run more than kDepth steps ahead of the O wave
        if (kend > KD) {
            const uint32_t must = (uint32_t)(kend - KD);
            uint32_t spins = 0;
            for (;;) {
                const uint64_t cc = __hip_atomic_load((const uint64_t *)&lds_cons[w][0], __ATOMIC_ACQUIRE,
                                                      __HIP_MEMORY_SCOPE_WORKGROUP);
                uint32_t have = min((uint32_t)cc, (uint32_t)(cc >> 32));
#ifdef MDPP_ABL_NOO1
                have = (uint32_t)(cc >> 32);
#endif
#ifdef MDPP_ABL_NOO2
                have = (uint32_t)cc;
#ifdef MDPP_ABL_NOO1
                have = must;
#endif
#else
                if (rows) {                         // whole-row stores: every O2 wave reads this wave's records
#pragma unroll
                    for (int j = 0; j < kBlock / 64; j++) have = min(have, wg_load_acq(&lds_cons[j][1]));
                }
#endif
                if (have >= must) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit) { status |= kStatusInternal; break; }
            }
        }
    };
    // one full chunk: slot j's actions -> next chunk's columns, (re)fill slot j, step, publish
    auto chunkE = [&](int c, int j, bool refill) {
        const int kbase = c * kChunk;
        badq[(j + 1) & 1] = 0u;
#pragma unroll
        for (int u = 0; u < kChunk; u++) colq[(j + 1) & 1][u] = column(actq[j][u], badq[(j + 1) & 1], u);
        if (refill) {
#pragma unroll
            for (int u = 0; u < kChunk; u++) actq[j][u] = load_act(kbase + (kAh + 1) * kChunk + u);
        }
        wait_room(kbase + kChunk);
        pull(c);
#pragma unroll
        for (int u = 0; u < kChunk; u++) stepE(colq[j & 1][u], kbase + u, (badq[j & 1] >> u) & 1u);
        if (NRN) __hip_atomic_store(&lds_epos[l], ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((l & 63) == 0) wg_store_rel(&lds_prod[w], (uint32_t)(kbase + kChunk));
    };
    // (single-exit loop body, no global load in an inner loop: the compiler then counts its vmcnt waits
    //  exactly, vmcnt(8 (kAhead - 1) + ...), instead of waiting for every load in flight)
    const int ngrp = nfull / kAh;
    for (int g = 0; g < ngrp; g++) {
#pragma unroll
        for (int j = 0; j < kAh; j++) chunkE(g * kAh + j, j, true);
    }
#pragma unroll
    for (int j = 0; j < kAh - 1; j++)
        if (ngrp * kAh + j < nfull) chunkE(ngrp * kAh + j, j, false);
    if (K % kChunk) {               // (no global load inside the loop above: its waits stay counted, not vmcnt(0))
        const int kbase = nfull * kChunk;
        Act ta[kChunk];
#pragma unroll
        for (int u = 0; u < kChunk; u++) ta[u] = load_act(kbase + u);
        wait_room(kbase + kChunk);
        pull(nfull);
#pragma unroll
        for (int u = 0; u < kChunk; u++)
            if (kbase + u < K) {
                uint32_t bm = 0;
                const Col cc = column(ta[u], bm, 0);
                stepE(cc, kbase + u, bm);
            }
        if ((l & 63) == 0) wg_store_rel(&lds_prod[w], (uint32_t)K);
    }

    // back to the handle's state words: hist bytes, queue nibbles + count, episode steps
    uint32_t hist = 0;
    for (int j = 3; j >= 0; j--) {
        const uint32_t nb = (k2 >> (4 * j)) & 0xFu;
        hist = (hist << 8) | ((nb & 8u) ? (nb & 7u) : 0xFFu);
    }
    const uint32_t qc = (uint32_t)__builtin_popcount(qv & 0x00888888u);
    const bool own_q = !PHILOX && !IRR && !nextmode && NZ == 0;
    if (NRN) __hip_atomic_store(&lds_epos[l], ep, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // (H un-draws what lies beyond)
    if (!own_q && !PHILOX && !NRN) {
        // word 1 of the state is not a queue for these handles: what sits in the register queue goes back to the H
        // lane (it un-draws everything not taken)
        __hip_atomic_store(&lds_head[l], head_local - qc, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (IRR) a.irr_state[i] = c1;                   // the irrelevant state has its own array
    uint32_t steps;
    if (HASMAX) steps = cnt - c0;
    else steps = last_reset ? (uint32_t)K - last_reset : steps0 + (uint32_t)K;
    uint32_t *st = (uint32_t *)&a.state[i];
    st[0] = hist; st[2] = steps | (pend ? 0x80000000u : 0u);        // word 3 (delay line) belongs to the O1 lane
    if (own_q) st[1] = (qv & 0x00777777u) | (qc << 24);   // (other handles: word 1 is older history, unused at L <= 3)
    if ((l & 63) == 0) __hip_atomic_fetch_add(&lds_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (status) atomicOr(&a.status[i], status);
}

// Returns false when the shape does not qualify (caller tries k_discrete_rollout_pipe, then _fast).
#if MDPP_LEAN_TU_NEXT
bool launch_discrete_lean_next(const DiscreteArgs &a, int K, const int32_t *actions, void *obs,
                               float *reward, uint8_t *term, uint8_t *trunc, void *final_obs,
                               hipStream_t s, char *name_out) {
    constexpr bool kNext = true;
    if (a.autoreset != MDPP_AUTORESET_NEXT_STEP) return false;
    const int nz = 0;
#elif MDPP_LEAN_TU_NOISE == 1
bool launch_discrete_lean_noise(const DiscreteArgs &a, int K, const int32_t *actions, void *obs,
                                float *reward, uint8_t *term, uint8_t *trunc, void *final_obs,
                                hipStream_t s, char *name_out) {
    constexpr bool kNext = false;
    const int nz = (a.has_p_noise ? 1 : 0) | (a.has_r_noise ? 2 : 0);
    if (nz == 0 || a.autoreset == MDPP_AUTORESET_NEXT_STEP || !a.philox || a.irr) return false;
#elif MDPP_LEAN_TU_NOISE == 2
bool launch_discrete_lean_npnoise(const DiscreteArgs &a, int K, const int32_t *actions, void *obs,
                                  float *reward, uint8_t *term, uint8_t *trunc, void *final_obs,
                                  hipStream_t s, char *name_out) {
    constexpr bool kNext = false;
    const int nz = (a.has_p_noise ? 1 : 0) | (a.has_r_noise ? 2 : 0);
    if (nz == 0 || a.autoreset == MDPP_AUTORESET_NEXT_STEP || a.philox || a.irr) return false;
#else
bool launch_discrete_lean_next(const DiscreteArgs &a, int K, const int32_t *actions, void *obs,
                               float *reward, uint8_t *term, uint8_t *trunc, void *final_obs,
                               hipStream_t s, char *name_out);
bool launch_discrete_lean_noise(const DiscreteArgs &a, int K, const int32_t *actions, void *obs,
                                float *reward, uint8_t *term, uint8_t *trunc, void *final_obs,
                                hipStream_t s, char *name_out);
bool launch_discrete_lean_npnoise(const DiscreteArgs &a, int K, const int32_t *actions, void *obs,
                                  float *reward, uint8_t *term, uint8_t *trunc, void *final_obs,
                                  hipStream_t s, char *name_out);
bool launch_discrete_lean(const DiscreteArgs &a, int K, const int32_t *actions, void *obs,
                          float *reward, uint8_t *term, uint8_t *trunc, void *final_obs,
                          hipStream_t s, char *name_out) {
    constexpr bool kNext = false;
    if (a.autoreset == MDPP_AUTORESET_NEXT_STEP)
        return launch_discrete_lean_next(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
    if ((a.has_p_noise || a.has_r_noise) && !a.philox)
        return launch_discrete_lean_npnoise(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
    // (Philox streams carry no state: a reward-noise key whose sigma is 0 adds 0.0 + 0.0 z = +0.0 whatever the normal is -- the
    //  noise-free instantiation, whose reward-value table holds the same float64 arithmetic, serves the handle)
    const bool ph_sig0 = a.philox && !a.has_p_noise && a.has_r_noise && a.r_noise == 0.0 && !(a.opts & MDPP_OPT_NO_SIGMA0);
    if ((a.has_p_noise || a.has_r_noise) && !ph_sig0)
        return launch_discrete_lean_noise(a, K, actions, obs, reward, term, trunc, final_obs, s, name_out);
    const int nz = 0;
#endif
    const bool ph = a.philox != 0, irr = a.irr != 0;
    if (final_obs) return false;        // (rollouts of K >= 32 steps never ask for final observations: mdpp_step does, K = 1)
    const bool shape = a.autoreset == MDPP_AUTORESET_NEXT_STEP ? a.lean_next_ok != 0
                       : (nz && !ph) ? a.shape_ok_noise_np != 0
                       : nz ? a.shape_ok_noise != 0
                       : irr ? a.shape_ok_irr != 0 : (ph ? (a.shape_ok != 0 || (MDPP_LEAN_TU_NOISE == 0 && !MDPP_LEAN_TU_NEXT && a.shape_ok_noise != 0 && a.r_noise == 0.0 && !a.has_p_noise)) : a.fast_ok != 0);
    if (!shape || (ph && (a.opts & MDPP_OPT_NO_PHILOX_FAST)) || K < 32 || a.N < kBlock ||
        (a.opts & (MDPP_OPT_NO_PIPE | MDPP_OPT_NO_LEAN)))
        return false;
    if (nz && (a.opts & MDPP_OPT_NO_QUIET_NOISE)) return false;
    if (!a.autoreset && a.max_steps > 0) return false;      // (the biased step counter saturates only through resets)
    if (a.S > 8 || a.A > 16 || a.every_n > 64 || a.max_steps >= 65536) return false;
    if (irr && (a.S1 > 8 || a.A1 > 16 || (8ULL * 2 * a.N * (unsigned long long)K) >= (1ULL << 32)))
        return false;
    const int grid = (a.N + kBlock - 1) / kBlock;
    const bool dl = a.delay > 0, hm = a.max_steps > 0, evn = a.every_n > 1;
    // sigma-0 reward noise on numpy streams (kernel header, Z0): the draws without their values
    const bool z0 = MDPP_LEAN_TU_NOISE == 2 && (nz & 2) && a.r_noise == 0.0 && !(a.opts & MDPP_OPT_NO_SIGMA0);
    (void)z0;
    if (name_out) {
        if (nz && z0)
            snprintf(name_out, kNameLen, "k_discrete_rollout_lean<OBS64=%d,DELAY=%d,HASMAX=%d,EVN=%d,PHILOX=%d,IRR=%d,NEXT=%d,PN=%d,RN=%d,Z0=1>",
                     !a.obs_i32, dl, hm, evn, ph, irr, kNext, nz & 1, (nz >> 1) & 1);
        else if (nz)
            snprintf(name_out, kNameLen, "k_discrete_rollout_lean<OBS64=%d,DELAY=%d,HASMAX=%d,EVN=%d,PHILOX=%d,IRR=%d,NEXT=%d,PN=%d,RN=%d>",
                     !a.obs_i32, dl, hm, evn, ph, irr, kNext, nz & 1, (nz >> 1) & 1);
        else
            snprintf(name_out, kNameLen, "k_discrete_rollout_lean<OBS64=%d,DELAY=%d,HASMAX=%d,EVN=%d,PHILOX=%d,IRR=%d,NEXT=%d>", !a.obs_i32, dl, hm, evn, ph, irr, kNext);
        return true;
    }
#if MDPP_LEAN_TU_NOISE
#define MDPP_LEAN_GO(O64, DL, HM, EV, NZ_)                                                                   \
    hipLaunchKernelGGL((k_discrete_rollout_lean<O64, DL, HM, EV, MDPP_LEAN_TU_NOISE == 1, false, false, NZ_>), dim3(grid), dim3(kRoles * kBlock), \
                       0, s, a, K, actions, obs, reward, term, trunc, final_obs)
#define MDPP_LEAN_LAUNCH(O64, DL, HM, EV)                                                                   \
    do {                                                                                                   \
        if (MDPP_LEAN_TU_NOISE == 2 && z0 && nz == 3) MDPP_LEAN_GO(O64, DL, HM, EV, (MDPP_LEAN_TU_NOISE == 2 ? 7 : 3));   \
        else if (MDPP_LEAN_TU_NOISE == 2 && z0) MDPP_LEAN_GO(O64, DL, HM, EV, (MDPP_LEAN_TU_NOISE == 2 ? 6 : 2));          \
        else if (nz == 3) MDPP_LEAN_GO(O64, DL, HM, EV, 3);                                                \
        else if (nz == 2) MDPP_LEAN_GO(O64, DL, HM, EV, 2);                                                \
        else MDPP_LEAN_GO(O64, DL, HM, EV, 1);                                                             \
    } while (0)
#else
#define MDPP_LEAN_GO(O64, DL, HM, EV, PH, IR)                                                                \
    hipLaunchKernelGGL((k_discrete_rollout_lean<O64, DL, HM, EV, PH, IR, kNext>), dim3(grid), dim3(kRoles * kBlock), \
                       0, s, a, K, actions, obs, reward, term, trunc, final_obs)
#define MDPP_LEAN_LAUNCH(O64, DL, HM, EV)                                                                   \
    do {                                                                                                   \
        if (ph && irr) MDPP_LEAN_GO(O64, DL, HM, EV, true, true);                                          \
        else if (ph) MDPP_LEAN_GO(O64, DL, HM, EV, true, false);                                           \
        else if (irr) MDPP_LEAN_GO(O64, DL, HM, EV, false, true);                                          \
        else MDPP_LEAN_GO(O64, DL, HM, EV, false, false);                                                  \
    } while (0)
#endif
#define MDPP_LEAN_L3(O64, DL, HM) do { if (evn) MDPP_LEAN_LAUNCH(O64, DL, HM, true); else MDPP_LEAN_LAUNCH(O64, DL, HM, false); } while (0)
#define MDPP_LEAN_L2(O64, DL) do { if (hm) MDPP_LEAN_L3(O64, DL, true); else MDPP_LEAN_L3(O64, DL, false); } while (0)
#ifdef MDPP_LEAN_SHAPES_MIN       // (ablation builds: the bench shape only)
    if (a.obs_i32 || !dl || hm || !evn) return false;
    MDPP_LEAN_LAUNCH(true, true, false, true);
#else
    if (a.obs_i32) { if (dl) MDPP_LEAN_L2(false, true); else MDPP_LEAN_L2(false, false); }
    else { if (dl) MDPP_LEAN_L2(true, true); else MDPP_LEAN_L2(true, false); }
#endif
#undef MDPP_LEAN_GO
#undef MDPP_LEAN_L2
#undef MDPP_LEAN_L3
#undef MDPP_LEAN_LAUNCH
    return true;
}

} // namespace mdpp

// Internal definitions shared by the translation units of libmdpp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <string>

#include "../../include/mdpp.h"

namespace mdpp {

// Cache policy of the rollout kernels' OUTPUT stores (aux of the raw-buffer store builtins; 2 = nt).  Every
// output row is written once and never read by the kernel; left at the default policy the write-back
// traffic of these rows throttled the whole memory pipeline of the CU (cfg2 fused rollout: 147 -> 106 us
// per launch with nt, profiles/archive/r02_ablation_lean_kernel.txt).  NOT for the picture kernels: their 16-byte
// stores of whole pictures ran 0.51 -> 0.44 (polygon pictures) and 0.54 -> 0.18 (continuous pictures) with nt.
#ifndef MDPP_ST_NT
#define MDPP_ST_NT 2
#endif
constexpr int kBlock = 256;         // 4 wavefronts of 64 lanes; one lane per env instance
constexpr uint32_t kNoKey = 0xFFFFFFFFu;
constexpr uint32_t kRingPyZero = 0x7FC0DE1Au; // float32 ring slot holding Python's float 0.0
// Philox stream ids beyond the MDPP_STREAM_* indices (keys are (seed, env id, tick, stream id)):
constexpr uint32_t kPhiloxResetStream = 3;    // an explicit reset(), keyed by the reset tick
constexpr uint32_t kPhiloxIrrStream = 4;      // P-noise of the irrelevant sub-space
constexpr uint32_t kPhiloxActionStream = 5;   // grid: the re-drawn noisy action
// (6-8: the post-processor's streams, mdpp_post.hip)
constexpr uint32_t kPhiloxStartStream = 9;    // discrete: start state of an in-rollout reset, one word per tick (mdpp_rng.hpp)
constexpr uint32_t kPhiloxStartIrrStream = 10; // ... of the irrelevant sub-space
constexpr uint32_t kPhiloxPNoiseStream = 12;  // discrete: transition noise, one word per tick (mdpp_rng.hpp)
constexpr uint32_t kPhiloxRNoiseStream = 13;  // discrete: reward noise, one float32 normal per tick (four per block)

// ---- per-episode noise statistics (cfg.episode_stats; general kernels only) ------------------------------------
// What the reference accumulates per env object and logs at every reset() (rl_toy_env.py:2231-2247; cleared
// :2360-2369): row 0 total_abs_noise_in_reward_episode (:1984), row 1 total_reward_episode (:1985: the reward after the
// delay line and the every-n mask, before noise / scale / shift), row 2 total_noisy_transitions_episode (discrete :1620,
// grid :1746), rows 3.. total_abs_noise_in_transition_episode per dimension (continuous :1686).  cur [nk][N] is the running
// episode, last [nk + 1][N] the episode a reset() ended (row nk: its total_transitions_episode).  cur == nullptr: off.
struct EpisodeStatsDev {
    double *cur, *last;
    int32_t nk;
};
__device__ __forceinline__ void est_add(const EpisodeStatsDev &e, long N, long i, int k, double v) {
    e.cur[(size_t)k * N + i] += v;
}
__device__ __forceinline__ void est_roll(const EpisodeStatsDev &e, long N, long i, uint32_t steps) {   // reset(): log, then clear
    for (int k = 0; k < e.nk; k++) {
        e.last[(size_t)k * N + i] = e.cur[(size_t)k * N + i];
        e.cur[(size_t)k * N + i] = 0.0;
    }
    e.last[(size_t)e.nk * N + i] = (double)steps;
}

// ---- discrete: kernel arguments (passed by value; wave-uniform => SGPRs) -------------------
struct DiscreteArgs {
    int32_t N;
    int32_t S, A, L, delay, every_n;
    int32_t shared_tables;      // 1: tables staged in LDS once per block; 0: one table set per env
    int32_t unit_rewards;       // reward table is a bitmask of keys, every reward = 1.0
    int32_t rew_sa;             // reward table keyed by (state, action) of the transition (custom reward matrix)
    int32_t has_p_noise, has_r_noise;
    int32_t autoreset, max_steps, obs_i32;
    int32_t philox;
    uint32_t pn_T;              // Philox streams: transition-noise threshold ceil(p 2^32) (mdpp_rng.hpp philox_pnoise_*)
    uint64_t pn_M, pn_M1;       // ... and ceil(2^64 (S - 1) / T) for the relevant / the irrelevant sub-space
    uint32_t nkeys;             // S^L
    uint32_t tick;              // head of the delay ring at this launch: env steps taken so far mod delay
    uint64_t ptick;             // env steps taken by this handle before this launch (Philox counter)
    const uint64_t *dtick;      // launches captured into a HIP graph: device word added to ptick at run time (tick_from_device)
    uint32_t opts;              // MDPP_OPT_* (host side: kernel selection only)
    uint64_t philox_seed;
    int64_t env_id_offset;
    double r_noise, scale, shift, term_add; // term_add = term_state_reward * reward_scale
    // tables (device)
    const uint8_t *P;           // [T][S][A]
    const double *rtable;       // [T][nkeys]      (unit_rewards == 0)
    const uint8_t *rbits;       // [T][rbits_stride] (unit_rewards == 1)
    const uint8_t *is_term;     // [T][S]
    const double *init_cdf;     // [T][S]
    const double *noise_cdf;    // [S][S]
    uint32_t rbits_stride;
    // LDS carve (byte offsets, 16-aligned), shared_tables only
    uint32_t lds_P, lds_term, lds_rew, lds_init, lds_noise, lds_bytes;
    uint32_t rew_in_lds, noise_in_lds;
    // per-env state (device)
    uint4 *state;               // {hist bytes 0-3, hist bytes 4-7, steps, ring bits}
    uint64_t *hist_hi;          // S > 255 (mdpp_discrete_wide.hip): states are 16-bit fields, the four oldest of the L + 1 live here; L > 7 (mdpp_discrete_long.hip): the eight oldest bytes
    uint32_t *ring_keys;        // [delay][N] keys awaiting payout (unit_rewards == 0)
    ulonglong2 *env_s, *env_inc, *sp_s, *sp_inc; // PCG64 streams
    uint32_t *status;
    EpisodeStatsDev est;
    // ---- irrelevant sub-space (cfg.irrelevant; general kernel only) ----
    int32_t irr, S1, A1;
    const uint8_t *P1;          // [T][S1][A1]
    const double *init_cdf1;    // [T][S1]
    const double *noise_cdf1;   // [S1][S1]
    uint32_t *irr_state;        // [N] irrelevant part of curr_state
    ulonglong2 *sp1_s, *sp1_inc; // observation_spaces[1] PCG64 streams
    // ---- precomputed on the host for the fused fast path (mdpp_discrete_fast.hip) ----
    uint32_t fast_ok;           // shape qualifies: shared LDS tables, unit rewards, no noise, L <= 3, S <= 16
    uint32_t shape_ok;          // fast_ok without its "numpy streams" condition (Philox handles: k_discrete_rollout_lean)
    uint32_t shape_ok_irr;      // the same shape with an irrelevant sub-space of at most 8 states (k_discrete_rollout_lean<IRR>)
    uint32_t shape_ok_noise;    // the lean shape (at most 8 states) with transition and / or reward noise on Philox streams (k_discrete_rollout_lean<..., NZ>)
    uint32_t lean_next_ok;      // ... with next-step autoreset (at most 8 states, with or without the irrelevant sub-space)
    uint32_t shape_ok_noise_np; // the lean shape with transition and / or reward noise on NUMPY streams (k_discrete_rollout_lean<..., PHILOX=0, NZ>)
    // transition noise on numpy streams, row-independent form of the S categoricals (:1604-1622): with m = r >> 11 the 53-bit
    // draw, the state re-drawn around table entry n is  min(a, n) + max(b - n, 0),  a = #{j <= S - 2 : pn_TL[j] <= r},
    // b = #{j <= S - 1 : pn_TU[j] <= r}  (thresholds pre-shifted by 11: compared with the 64-bit word itself).  Valid when
    // ceil(cdf_n[j] 2^53) is the same for every row n > j (TL) and for every row n <= j (TU): checked on the host, else
    // shape_ok_noise_np stays 0 and the general / quiet kernels search the row's own thresholds.
    uint64_t pn_TL[8], pn_TU[8];
    uint32_t s_shift;           // log2(S) when S is a power of two, else 0xFFFFFFFF
    uint32_t key_mask;          // S^L - 1 (power-of-two S)
    uint32_t spow;              // S^(L-1)
    uint64_t term_mask;         // bit s set <=> state s terminal (S <= 64)
    uint64_t init_thr[16];      // ceil(init_cdf[j] * 2^53): cdf[j] <= u  <=>  init_thr[j] <= (r >> 11)
    float rsel[4];              // reward for {paid*2 + terminal}, formed in float64 like :1987-1990,:2107
    uint64_t minv_lo, minv_hi;  // inverse of the PCG64 LCG multiplier mod 2^128 (un-drawing)
};

// ---- discrete, one launch = one step (mdpp_discrete_step1.hip): everything the kernel reads, nothing else -- built once at
// mdpp_upload_discrete_tables (the caller's buffers are filled in per launch)
#ifndef MDPP_S1_REPLICAS
#define MDPP_S1_REPLICAS 1
#endif
// 1 KiB rounds of the wide one-step kernel's table blob that ONE wave stages into its LDS (k_discrete_step1w<NZ>: with a
// noise key / without): the upload refuses the one-step path for a longer blob, the launcher checks again
constexpr uint32_t kS1wRounds = 8, kS1wRoundsNoise = 12;
// copies of the one-step kernels' table blob (power of two), one per group of workgroups.  1: sixteen copies measured the
// same (S = 50: 4.07 against 4.04 us per step) -- 1 024 waves reading the same 4 KiB is not what made that launch slow, the
// compiler serialising the blob's loads was (mdpp_discrete_step1.hip)
constexpr int kS1Replicas = MDPP_S1_REPLICAS;
struct Step1Args {
    int32_t N;
    uint32_t A, S, L, every_n, max_steps, delay, autoreset;
    uint32_t term32;            // bit s: state s terminal (S <= 16)
    uint32_t nan_mask;          // 0xFF << 8 L: history byte L is the NaN test (:1822)
    uint32_t topup_rounds, topup_fill;   // start-state queue: at most this many rounds of one draw per lane after a step, up to this many entries
    float rsel[4];              // reward for {paid*2 + terminal} (DiscreteArgs::rsel)
    double inv_every_n;
    uint64_t ptick;             // Philox streams: env steps taken by this handle before this launch ...
    const uint64_t *dtick;      // ... plus this device word when the launch is replayed from a graph (tick_now)
    uint64_t philox_seed;
    int64_t env_id_offset;
    const uint4 *blob;          // 1 KiB: P columns as nibbles, rho_0 thresholds, reward bits (layout: mdpp_discrete_step1.hip)
    // k_discrete_step1w (any S <= 255 whose tables fit 8 KiB): byte offsets into the blob, its size in 1 KiB rounds
    uint32_t wide, blob_rounds, off_term, off_thr, off_thr31, off_rew, S8, off_bk;
    // ... with rewards that are not all 1.0 (reward_dist): a float64 table by sequence key at off_rew, the delay line of KEYS in HBM
    uint32_t unit, ring_head;   // ring_head: env steps taken so far mod delay (a graph replay: from ptick + *dtick)
    uint32_t *ring_keys;        // [delay][N]
    double scale, shift, term_add;
    // ... with transition and / or reward noise (k_discrete_step1w<..., NZ = true>): thresholds of the S noise categoricals
    // [S][S8] (numpy streams) at off_tn, numpy's ziggurat tables ki / wi / fi (3 x 2 KiB) at off_zig; the state space's stream
    uint32_t has_p_noise, has_r_noise, off_tn, off_zig, pn_T;
    uint64_t pn_M;
    double r_noise;
    ulonglong2 *sp_s;
    const ulonglong2 *sp_inc;
    const int32_t *actions;
    void *obs;
    float *reward;
    uint8_t *term, *trunc;
    void *final_obs;
    uint4 *state;
    ulonglong2 *env_s;
    const ulonglong2 *env_inc;
    uint32_t *status;
};

struct ContinuousArgs {
    int32_t N, D, n_rel, order, delay, every_n;
    int32_t make_denser, has_p_noise, has_r_noise, bounded;
    int32_t autoreset, max_steps, philox, n_boxes, rel_prefix;
    uint32_t tick;              // head of the delay ring at this launch: env steps taken so far mod delay
    uint64_t ptick;             // env steps taken by this handle before this launch (Philox counter)
    const uint64_t *dtick;      // launches captured into a HIP graph: device word added to ptick at run time (tick_from_device)
    uint32_t opts;              // MDPP_OPT_* (host side: kernel selection only)
    uint64_t philox_seed;
    int64_t env_id_offset;
    float inertia32, amax32, smax32, radius32, alw32, scale32, shift32, term_add32;
    float tpow32[MDPP_MAX_ORDER + 1];
    double fact[MDPP_MAX_ORDER + 1];
    double p_noise, r_noise, scale, shift, term_add, reset_lo, reset_range;
    int32_t rel[MDPP_MAX_DIM];
    float target[MDPP_MAX_DIM];
    float box_lo[MDPP_MAX_BOXES * MDPP_MAX_DIM];
    float box_hi[MDPP_MAX_BOXES * MDPP_MAX_DIM];
    // per-env state (device), struct-of-arrays: consecutive lanes -> consecutive addresses
    float *sd;                  // [order+1][D][N] state_derivatives
    float *cur;                 // [D][N] last returned (noisy, clipped) state
    uint2 *meta;                // {steps, flags: bit0 reached_terminal}
    uint32_t *ring;             // [delay][N] float32 bit patterns (kRingPyZero = Python 0.0)
    // reward_function move_along_a_line (0 = move_to_a_point): sequence_length, the last line_L states'
    // relevant coordinates [slot = s % line_L][line_NL][N] (s = transitions made when the state was
    // reached), and a float64 delay line (these rewards are Python floats)
    int32_t line_L;
    int32_t line_NL;            // row width of line_hist: 4 (at most 4 relevant dimensions) or 8
    int32_t line_lds;           // this launch mirrors the L points of every lane in dynamic LDS (set by launch_step_t)
    float *line_hist;
    double *line_ws;            // more than 8 relevant dimensions: the fit's matrices in HBM, [(2 n n + 3 n)][N] (c_line_reward_big); else null
    double *ring64;             // [delay][N]
    // the reference's DEFAULT target_point, float64 zeros over every dimension (:652-654): float64 distances, target
    // latch and dense reward; rew64 = the reward is float64 throughout (line reward, or default target + make_denser)
    int32_t target64, rew64;
    double radius;
    ulonglong2 *env_s, *env_inc, *sp_s, *sp_inc;
    uint32_t *status;
    EpisodeStatsDev est;
    // ---- precomputed on the host for the fused fast path (mdpp_continuous_fast.hip) ----
    int32_t park;               // helper waves park lanes that leave the ziggurat's fast path (mdpp_continuous_fast.hip)
    int32_t image_quirk;        // image observations: every step clips and zeroes the derivatives (see k_continuous_step C4)
    uint32_t fast_ok;           // PCG64, no hypercubes, relevant dims = a prefix, bounded, delay 0, every_n 1
    uint32_t inertia_pow2;      // inertia is a power of two: a / inertia == a * inv_inertia32 exactly
    float inv_inertia32;
    uint32_t fact_pow2_mask;    // bit k: k! is a power of two (k = 1, 2)
    double inv_fact[MDPP_MAX_ORDER + 1];
};

// A launch captured into a HIP graph replays with the argument values it was captured with -- the step counter among
// them (Philox keys; the head of a delay line kept in memory).  In capture mode (mdpp_graph_capture) the launches carry a
// pointer to a device word instead: the difference between the counter NOW and the counter at capture, written by
// mdpp_graph_set_tick_offset before a replay; every step / rollout kernel reads its counter through tick_now() /
// ring_head_now() (a wave-uniform scalar load).  (NOT by patching the argument struct inside the kernel: a write to a
// by-value kernel argument makes the compiler copy the whole struct to private memory -- the cfg2 rollout went from 129 to
// 195 us per launch with exactly that.)
template <class A>
__device__ __forceinline__ uint64_t tick_now(const A &a) { return a.ptick + (a.dtick ? *a.dtick : 0ULL); }
template <class A>
__device__ __forceinline__ uint32_t ring_head_now(const A &a, uint64_t ptick) {
    return a.dtick ? (a.delay > 0 ? (uint32_t)(ptick % (uint64_t)a.delay) : 0u) : a.tick;
}

// A kernel that asks for more than the default 32 KiB of dynamic LDS: raise its limit, and say whether the launch can go
// ahead (static + dynamic LDS within what a workgroup may have; the attribute call succeeded).  false = the caller takes
// its path without the LDS mirror.
inline bool dynamic_lds_ok(const void *kernel, size_t dyn_bytes) {
    if (dyn_bytes <= 32 * 1024) return true;
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, kernel) != hipSuccess) { (void)hipGetLastError(); return false; }
    int dev = 0, max_lds = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (fa.sharedSizeBytes + dyn_bytes > (size_t)max_lds) return false;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
    return true;
}

// ---- grid (mdpp_grid.hip) ----
struct GridArgs {
    int32_t N, G;               // envs; state dimensions (2 or 4)
    int32_t shape[4], target[2];
    int32_t make_denser, has_p_noise, has_r_noise, every_n;
    int32_t autoreset, max_steps, obs_i32, philox;
    uint64_t ptick;             // env steps taken by this handle before this launch (Philox counter)
    const uint64_t *dtick;      // launches captured into a HIP graph: device word added to ptick at run time (tick_from_device)
    uint32_t opts;              // MDPP_OPT_* (host side: kernel selection only)
    uint64_t philox_seed;
    int64_t env_id_offset;
    double p_noise, r_noise, scale, shift, term_add;
    uint4 *state;               // {cells (one byte per dimension), steps, flags: bit0 reached target, -}
    ulonglong2 *env_s, *env_inc, *sp_s, *sp_inc, *act_s, *act_inc;
    uint2 *act_half;            // numpy's buffered 32-bit half of the action stream
    uint64_t minv_lo, minv_hi;  // inverse of the PCG64 LCG multiplier mod 2^128 (un-drawing queued reset cells)
    uint32_t *status;
    EpisodeStatsDev est;
};

} // namespace mdpp

// The handle.  Plain struct; all members are host-side bookkeeping + device allocations.
struct mdpp_env {
    mdpp_config cfg;
    int device;
    int num_cus;                // compute units of `device` (persistent-kernel grids)
    std::string err;
    uint64_t tick;              // env steps taken so far (ring head = tick mod delay; Philox counter)
    uint64_t reset_tick;        // reset() calls so far (Philox counter)
    uint32_t opts;              // MDPP_OPT_* kernel-selection switches (mdpp_set_options)
    char kname[192];            // mdpp_kernel_name()
    // device allocations
    void *d_P, *d_rtable, *d_rbits, *d_is_term, *d_init_cdf, *d_noise_cdf;
    void *d_state, *d_ring, *d_status;
    void *d_hist_hi;            // discrete, S > 255
    void *d_line_hist, *d_line_ws, *d_ring64;                             // continuous, move_along_a_line
    void *d_est_cur, *d_est_last;                             // cfg.episode_stats: EpisodeStatsDev rows
    int32_t est_nk;
    void *d_P1, *d_init_cdf1, *d_noise_cdf1, *d_irr_state;   // irrelevant sub-space
    bool irr_ready;
    bool graph_capture;         // mdpp_graph_capture(h, 1): launches take the step counter's offset from d_tick_off
    void *d_tick_off;           // uint64: counter now - counter at capture (mdpp_graph_set_tick_offset)
    bool line_hist_stale;       // move_along_a_line: set_state_continuous restored the step counters, the fit's window not yet
    void *d_sd, *d_cur, *d_meta;
    void *d_rng_s[MDPP_NUM_STREAMS], *d_rng_inc[MDPP_NUM_STREAMS], *d_rng_half;
    void *d_img_tpl, *d_img_tplp, *d_img_clsx, *d_img_clsy, *d_img_rot, *d_img_state_out, *d_img_state_final;
    void *d_img_near;           // fast renderer: table of the near dwords of a polygon (mdpp_image.hip render_fast_eval), or null
    void *d_img_rec;            // ImgRec [2][img_chunk][N] per-image records (mdpp_image.hip), 64 B each
    void *d_img_ctr;            // uint32 [2][2]: the fast renderer's work counters (scratch set x render launch)
    int32_t img_chunk;          // env steps per state-kernel + draw + render batch
    uint32_t imgc_disc_rows[32]; // continuous image observations: the disc raster, one bitmask per row
    void *d_imgc_boxes;          // ... and the terminal hypercubes / cells it draws: [n_boxes]{lo0, lo1, hi0, hi1} float32 (grid: {c0, c1, 0, 0})
    bool img_ready, img_fast_ok, img_lines_ready;   // img_fast_ok: k_image_obs<true> applies (mdpp_image.hip)
    int32_t img_n_radii, img_n_cls_x, img_n_cls_y;
    int32_t img_colb;            // fast renderer: 64 (k_image_obs_fast) or 128 (k_image_obs_wide) bytes of an LDS row per wave
    uint32_t nkeys, rbits_stride;
    bool tables_ready, streams_ready[MDPP_NUM_STREAMS];
    hipEvent_t ev0, ev1;
    // image rollouts: the next batch's state / draw / record kernels run on a side stream under the
    // rendering of the current batch (two scratch sets, events both ways; mdpp_capi.hip)
    hipStream_t side_stream;
    hipEvent_t ev_entry, ev_side[2], ev_render[2];
    mdpp::DiscreteArgs dargs;
    mdpp::Step1Args s1args;     // k_discrete_step1's argument block (blob == nullptr: the shape does not qualify)
    void *d_s1_blob;
    mdpp::ContinuousArgs cargs;
    mdpp::GridArgs gargs;
};

namespace mdpp {
// implemented in the kernel translation units
// name_out != nullptr: a dry run -- the launcher writes the name of the kernel it would launch (at most
// kNameLen bytes) and launches nothing
constexpr int kNameLen = 192;
int launch_discrete_step(mdpp_env *h, int K, const int32_t *actions, void *obs, float *reward,
                         uint8_t *term, uint8_t *trunc, void *final_obs, hipStream_t s, char *name_out = nullptr);
int launch_discrete_reset(mdpp_env *h, const uint8_t *mask, void *obs, hipStream_t s);
// state spaces of 256 ... 65 535 states: the general kernel alone, 16-bit table entries and history fields (mdpp_discrete_wide.hip)
int launch_discrete_step_wide(mdpp_env *h, int K, const int32_t *actions, void *obs, float *reward,
                              uint8_t *term, uint8_t *trunc, void *final_obs, hipStream_t s, char *name_out);
int launch_discrete_reset_wide(mdpp_env *h, const uint8_t *mask, void *obs, hipStream_t s);
// sequence_length 8 ... 15: the same with a history of sixteen byte fields (mdpp_discrete_long.hip)
int launch_discrete_step_long(mdpp_env *h, int K, const int32_t *actions, void *obs, float *reward,
                              uint8_t *term, uint8_t *trunc, void *final_obs, hipStream_t s, char *name_out);
int launch_discrete_reset_long(mdpp_env *h, const uint8_t *mask, void *obs, hipStream_t s);
bool launch_discrete_fast(const DiscreteArgs &a, int K, const int32_t *actions, void *obs, float *reward,
                          uint8_t *term, uint8_t *trunc, void *final_obs, hipStream_t s, char *name_out = nullptr);
int launch_continuous_step(mdpp_env *h, int K, const float *actions, float *obs, float *reward,
                           uint8_t *term, uint8_t *trunc, float *final_obs, hipStream_t s, char *name_out = nullptr);
int launch_continuous_reset(mdpp_env *h, const uint8_t *mask, float *obs, hipStream_t s);
bool launch_discrete_step1(const DiscreteArgs &d, const Step1Args &proto, const int32_t *actions, void *obs, float *reward,
                           uint8_t *term, uint8_t *trunc, void *final_obs, hipStream_t s, char *name_out = nullptr);
bool launch_discrete_quiet(const DiscreteArgs &a, int K, const int32_t *actions, void *obs, float *reward,
                           uint8_t *term, uint8_t *trunc, void *final_obs, hipStream_t s, char *name_out = nullptr);
bool launch_discrete_pipe(const DiscreteArgs &a, int K, const int32_t *actions, void *obs, float *reward,
                          uint8_t *term, uint8_t *trunc, void *final_obs, hipStream_t s, char *name_out = nullptr);
bool launch_discrete_lean(const DiscreteArgs &a, int K, const int32_t *actions, void *obs, float *reward,
                          uint8_t *term, uint8_t *trunc, void *final_obs, hipStream_t s, char *name_out = nullptr);
bool launch_continuous_fast(const ContinuousArgs &a, int K, const float *actions, float *obs, float *reward,
                            uint8_t *term, uint8_t *trunc, float *final_obs, hipStream_t s, char *name_out = nullptr);
bool launch_continuous_step1(const ContinuousArgs &a, const float *actions, float *obs, float *reward, uint8_t *term,
                             uint8_t *trunc, float *final_obs, hipStream_t s, char *name_out = nullptr);
bool launch_continuous_line(const ContinuousArgs &a, int K, const float *actions, float *obs, float *reward,
                            uint8_t *term, uint8_t *trunc, float *final_obs, hipStream_t s, char *name_out = nullptr);
int launch_imagec_obs(mdpp_env *h, int K, const void *states, const void *final_states, const uint8_t *term,
                      const uint8_t *trunc, const uint8_t *mask, uint8_t *img_out, uint8_t *img_final, hipStream_t s);
int launch_grid_step(mdpp_env *h, int K, const int32_t *actions, void *obs, float *reward, uint8_t *term,
                     uint8_t *trunc, void *final_obs, hipStream_t s, char *name_out = nullptr);
int launch_grid_reset(mdpp_env *h, const uint8_t *mask, void *obs, hipStream_t s);
// phase bits: 1 = draw + records, 2 = render (phase == 2 exactly: pipelined, the persistent grid leaves slots
// free for the next batch's state kernel; 6 = render only on the full grid); buf: scratch set (0 / 1)
int launch_image_obs(mdpp_env *h, int K, const int32_t *state_out, const int32_t *state_final,
                     const uint8_t *term, const uint8_t *trunc, const uint8_t *mask,
                     uint8_t *img_out, uint8_t *img_final, hipStream_t s, int phase = 3, int buf = 0);
// One step's draw + record + render in one kernel (mdpp_image.hip k_image_step1): 1 = launched, 0 = this handle keeps
// launch_image_obs, < 0 = error
int launch_image_step1(mdpp_env *h, const int32_t *state_out, const int32_t *state_final, const uint8_t *term,
                       const uint8_t *trunc, uint8_t *img_out, uint8_t *img_final, hipStream_t s);
const char *image_obs_kernel_name(const mdpp_env *h, int K);    // the renderer launch_image_obs() uses
} // namespace mdpp

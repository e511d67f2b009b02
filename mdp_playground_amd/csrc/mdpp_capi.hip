// C ABI of libmdpp_hip.so (see include/mdpp.h): handle lifetime, table/stream upload, dispatch.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "mdpp_internal.hpp"
#include "mdpp_rng.hpp"

using namespace mdpp;

static std::string g_create_err;

#define HIPCHK(h, expr)                                                                  \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                \
            return MDPP_EHIP;                                                            \
        }                                                                                \
    } while (0)

static int fail(mdpp_env *h, int code, const std::string &msg) {
    if (h) h->err = msg; else g_create_err = msg;
    return code;
}

static uint32_t align16(uint32_t x) { return (x + 15u) & ~15u; }

// Inverse of PCG64's 128-bit LCG multiplier modulo 2^128 (Newton iteration; the multiplier is odd).
static unsigned __int128 pcg_mult_inverse() {
    static unsigned __int128 inv = 0;
    if (inv == 0) {
        const unsigned __int128 m = (((unsigned __int128)0x2360ED051FC65DA4ULL) << 64) | 0x4385DF649FCCF645ULL;
        unsigned __int128 x = m; // correct to 3 bits
        for (int it = 0; it < 7; it++) x *= 2 - m * x;
        inv = x;
    }
    return inv;
}

extern "C" int mdpp_abi_version(void) { return MDPP_ABI_VERSION; }

extern "C" const char *mdpp_last_error(const mdpp_env *h) {
    return h ? h->err.c_str() : g_create_err.c_str();
}

static void free_all(mdpp_env *h) {
    void *ptrs[] = {h->d_P, h->d_rtable, h->d_rbits, h->d_is_term, h->d_init_cdf, h->d_noise_cdf,
                    h->d_state, h->d_ring, h->d_status, h->d_sd, h->d_cur, h->d_meta, h->d_rng_half,
                    h->d_P1, h->d_init_cdf1, h->d_noise_cdf1, h->d_irr_state,
                    h->d_img_tpl, h->d_img_tplp, h->d_img_clsx, h->d_img_clsy, h->d_img_rot, h->d_img_state_out,
                    h->d_img_state_final, h->d_img_rec, h->d_img_ctr, h->d_line_hist, h->d_line_ws, h->d_ring64, h->d_est_cur, h->d_est_last, h->d_tick_off,
                    h->d_img_near, h->d_s1_blob, h->d_imgc_boxes, h->d_hist_hi};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (int s = 0; s < MDPP_NUM_STREAMS; s++) {
        if (h->d_rng_s[s]) (void)hipFree(h->d_rng_s[s]);
        if (h->d_rng_inc[s]) (void)hipFree(h->d_rng_inc[s]);
    }
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    for (hipEvent_t e : {h->ev_entry, h->ev_side[0], h->ev_side[1], h->ev_render[0], h->ev_render[1]})
        if (e) (void)hipEventDestroy(e);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
}

extern "C" void mdpp_destroy(mdpp_env *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    free_all(h);
    delete h;
}

static int alloc_zero(mdpp_env *h, void **p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    HIPCHK(h, hipMalloc(p, bytes));
    HIPCHK(h, hipMemset(*p, 0, bytes));
    return MDPP_OK;
}

extern "C" int mdpp_create(const mdpp_config *cfg, int device, mdpp_env **out) {
    if (!cfg || !out) return fail(nullptr, MDPP_EINVAL, "mdpp_create: null argument");
    if (cfg->abi_version != MDPP_ABI_VERSION)
        return fail(nullptr, MDPP_EINVAL, "mdpp_create: abi_version mismatch");
    if (cfg->num_envs <= 0) return fail(nullptr, MDPP_EINVAL, "mdpp_create: num_envs <= 0");
    if (cfg->delay < 0 || cfg->every_n < 1)
        return fail(nullptr, MDPP_EINVAL, "mdpp_create: need delay >= 0 and reward_every_n_steps >= 1");
    if (cfg->autoreset < MDPP_AUTORESET_DISABLED || cfg->autoreset > MDPP_AUTORESET_NEXT_STEP)
        return fail(nullptr, MDPP_EINVAL, "mdpp_create: unknown autoreset mode");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess)
        return fail(nullptr, MDPP_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    mdpp_env *h = new mdpp_env();
    memset(&h->cfg, 0, sizeof(h->cfg));
    h->cfg = *cfg;
    h->device = device;
    h->num_cus = 256;
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) h->num_cus = v; }
    h->tick = 0; h->reset_tick = 0;
    h->opts = 0; h->kname[0] = 0;
    h->d_P = h->d_rtable = h->d_rbits = h->d_is_term = h->d_init_cdf = h->d_noise_cdf = nullptr;
    h->d_state = h->d_ring = h->d_status = h->d_sd = h->d_cur = h->d_meta = h->d_rng_half = nullptr;
    h->d_P1 = h->d_init_cdf1 = h->d_noise_cdf1 = h->d_irr_state = nullptr;
    h->d_line_hist = h->d_line_ws = h->d_ring64 = nullptr;
    h->d_est_cur = h->d_est_last = nullptr; h->est_nk = 0;
    h->irr_ready = false;
    h->line_hist_stale = false;
    h->graph_capture = false;
    h->d_tick_off = nullptr;
    h->d_img_tpl = h->d_img_tplp = h->d_img_clsx = h->d_img_clsy = h->d_img_rot = nullptr;
    h->d_img_near = nullptr;
    h->d_s1_blob = nullptr;
    memset(&h->s1args, 0, sizeof(h->s1args));
    h->d_img_state_out = h->d_img_state_final = h->d_img_rec = h->d_img_ctr = nullptr;
    h->d_imgc_boxes = nullptr;
    h->d_hist_hi = nullptr;
    // env steps per batch of an image rollout, while the records of two batches stay below about 1 GiB: 64 (cfg4, round 3:
    // 7 740 us per 512 steps with batches of 16, 7 440 with 32; with the renderer's waves claiming their images, 7 250 / 6 830 /
    // 6 660 with 16 / 32 / 64: fewer kernel tails and hand-overs; a long rollout starts with batches of 8 and 16 because
    // nothing hides the first batch's serial draw kernel, image_batches)
    {
        const size_t imgs = (size_t)cfg->num_envs * (cfg->irrelevant ? 2 : 1);
        h->img_chunk = imgs <= 32768 ? 64 : imgs <= 65536 ? 32 : 16;
    }
#ifdef MDPP_ABL_IMG_CHUNK
    h->img_chunk = MDPP_ABL_IMG_CHUNK;
#endif
    h->img_ready = false; h->img_fast_ok = false; h->img_lines_ready = false; h->img_colb = 64;
    for (int r = 0; r < 32; r++) h->imgc_disc_rows[r] = 0;
    h->img_n_radii = h->img_n_cls_x = h->img_n_cls_y = 0;
    for (int s = 0; s < MDPP_NUM_STREAMS; s++) { h->d_rng_s[s] = h->d_rng_inc[s] = nullptr; h->streams_ready[s] = false; }
    h->tables_ready = false;
    h->ev0 = h->ev1 = nullptr;
    h->side_stream = nullptr;
    h->ev_entry = h->ev_side[0] = h->ev_side[1] = h->ev_render[0] = h->ev_render[1] = nullptr;
    const size_t N = (size_t)cfg->num_envs;
    int rc = MDPP_OK;
#define TRY(x) do { rc = (x); if (rc != MDPP_OK) { g_create_err = h->err; free_all(h); delete h; return rc; } } while (0)
#define TRYHIP(x) do { hipError_t e2_ = (x); if (e2_ != hipSuccess) { g_create_err = std::string(#x) + ": " + hipGetErrorString(e2_); free_all(h); delete h; return MDPP_EHIP; } } while (0)
    TRYHIP(hipEventCreate(&h->ev0));
    TRYHIP(hipEventCreate(&h->ev1));
    TRY(alloc_zero(h, &h->d_status, N * sizeof(uint32_t)));
    TRY(alloc_zero(h, &h->d_tick_off, sizeof(uint64_t)));
    if (cfg->episode_stats) {       // per-episode noise statistics (EpisodeStatsDev): running episode + the one a reset() ended
        h->est_nk = 3 + (cfg->kind == MDPP_KIND_CONTINUOUS ? cfg->D : 0);
        TRY(alloc_zero(h, &h->d_est_cur, (size_t)h->est_nk * N * sizeof(double)));
        TRY(alloc_zero(h, &h->d_est_last, (size_t)(h->est_nk + 1) * N * sizeof(double)));
    }
    if (cfg->rng_mode == MDPP_RNG_NUMPY_PCG64) {
        for (int s = 0; s < MDPP_NUM_STREAMS; s++) {
            if (s == MDPP_STREAM_IMAGE && !(cfg->image && cfg->kind == MDPP_KIND_DISCRETE)) continue;
            if (s == MDPP_STREAM_SPACE_IRR && !(cfg->kind == MDPP_KIND_DISCRETE && cfg->irrelevant)) continue;
            if (s == MDPP_STREAM_ACTION && cfg->kind != MDPP_KIND_GRID) continue;
            TRY(alloc_zero(h, &h->d_rng_s[s], N * 16));
            TRY(alloc_zero(h, &h->d_rng_inc[s], N * 16));
        }
        if (cfg->kind == MDPP_KIND_GRID) TRY(alloc_zero(h, &h->d_rng_half, N * 8));   // action stream's 32-bit half
        if (cfg->image && cfg->kind == MDPP_KIND_DISCRETE) TRY(alloc_zero(h, &h->d_rng_half, N * 8));
    } else if (cfg->rng_mode != MDPP_RNG_PHILOX) {
        g_create_err = "mdpp_create: unknown rng_mode"; free_all(h); delete h; return MDPP_EINVAL;
    }
    if (cfg->image && cfg->kind == MDPP_KIND_DISCRETE) {      // (either RNG mode)
        // scratch of one batch of img_chunk env steps: states in, transform records in between
        const size_t sub = cfg->irrelevant ? 2 : 1;       // images per observation (one per sub-space)
        // two sets of everything: batch b + 1 is prepared while batch b is rendered
        TRY(alloc_zero(h, &h->d_img_state_out, 2 * (size_t)h->img_chunk * N * 4 * sub));
        TRY(alloc_zero(h, &h->d_img_state_final, 2 * (size_t)h->img_chunk * N * 4 * sub));
        TRY(alloc_zero(h, &h->d_img_rec, 2 * 2 * (size_t)h->img_chunk * N * 64 * sub));
        TRY(alloc_zero(h, &h->d_img_ctr, 2 * 2 * 64 * 128));     // [scratch set][render launch][kImgCtrs] counters, 128 B apart
    }

    if (cfg->image && cfg->kind == MDPP_KIND_DISCRETE && (cfg->img_w < 1 || cfg->img_h < 1 || cfg->img_tpl_size < 1)) {
        g_create_err = "mdpp_create: image observations of a discrete env need img_w, img_h, img_tpl_size >= 1";
        free_all(h); delete h; return MDPP_EUNSUPPORTED;
    }
    if (cfg->image && cfg->kind == MDPP_KIND_CONTINUOUS &&
        ((cfg->D != 2 && cfg->D != 4) || !isfinite(cfg->state_space_max) || cfg->img_w < 1 || cfg->img_h < 1 ||
         cfg->img_r0 < 1 || cfg->img_r0 > 15)) {
        g_create_err = "mdpp_create: ImageContinuous observations need 2 or 4 bounded state dimensions "
                       "and a disc radius of 1..15";
        free_all(h); delete h; return MDPP_EUNSUPPORTED;
    }
    if (cfg->image) {      // image rollouts pipeline their batches over a side stream (step_common)
        // (at the highest priority: streams of one priority share a few hardware queues, and a side stream that lands in the
        //  caller's queue serialises the two pipeline stages -- 7.4 or 8.4 ms per cfg4 launch from run to run when bench.py
        //  ran its legs on a non-default stream; the queues of another priority are not the caller's unless it asks for them)
        {
            int lo = 0, hi = 0;
            TRYHIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
#ifdef MDPP_ABL_SIDE_NORMAL
            hi = 0;
#endif
            TRYHIP(hipStreamCreateWithPriority(&h->side_stream, hipStreamNonBlocking, hi));
        }
        for (hipEvent_t *e : {&h->ev_entry, &h->ev_side[0], &h->ev_side[1], &h->ev_render[0], &h->ev_render[1]})
            TRYHIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    if (cfg->kind == MDPP_KIND_DISCRETE) {
        // (S > 255 -- round 6: 16-bit table entries and history fields, served by the general kernel alone, mdpp_discrete_wide.hip;
        //  without picture observations; an irrelevant sub-space keeps its own limit of 255 states)
        // (sequence_length 8 ... 15 -- round 6: a history of sixteen byte fields, mdpp_discrete_long.hip; S <= 255 there)
        const bool wide = cfg->S > 255, lng = cfg->L > 7;
        if (cfg->S < 2 || cfg->S > 65535 || cfg->A < 1 || cfg->L < 1 || cfg->L > 15 || ((wide || lng) && cfg->image) || (wide && lng)) {
            g_create_err = "mdpp_create: discrete needs 2 <= S <= 65535, A >= 1, 1 <= L <= 15 (S <= 255 and L <= 7 with image observations; "
                           "not S > 255 together with L > 7)";
            free_all(h); delete h; return MDPP_EUNSUPPORTED;
        }
        if (cfg->has_transition_noise && cfg->S > 8192) {
            // (the S x S table of the noise categoricals' cdfs is uploaded whole, 8 S^2 bytes: 512 MiB at 8 192 states)
            g_create_err = "mdpp_create: transition noise needs S <= 8192 (the S x S cdf table)";
            free_all(h); delete h; return MDPP_EUNSUPPORTED;
        }
        if (cfg->num_tables != 1 && cfg->num_tables != cfg->num_envs) {
            g_create_err = "mdpp_create: num_tables must be 1 or num_envs"; free_all(h); delete h; return MDPP_EINVAL;
        }
        const bool rew_sa = cfg->reward_kind == MDPP_REWARD_STATE_ACTION;
        if ((cfg->reward_kind != MDPP_REWARD_SEQUENCES && !rew_sa) || (rew_sa && cfg->unit_rewards)) {
            g_create_err = "mdpp_create: reward_kind must be MDPP_REWARD_SEQUENCES or, with unit_rewards = 0, MDPP_REWARD_STATE_ACTION";
            free_all(h); delete h; return MDPP_EINVAL;
        }
        double nk = rew_sa ? (double)cfg->S * (double)cfg->A : pow((double)cfg->S, (double)cfg->L);
        if (nk > 4.0e9) { g_create_err = "mdpp_create: S^L too large"; free_all(h); delete h; return MDPP_EUNSUPPORTED; }
        if (cfg->unit_rewards && cfg->delay > 32) {
            g_create_err = "mdpp_create: unit_rewards needs delay <= 32"; free_all(h); delete h; return MDPP_EINVAL;
        }
        h->nkeys = (uint32_t)nk;
        h->rbits_stride = (h->nkeys + 7u) / 8u;
        const size_t T = (size_t)cfg->num_tables;
        if (cfg->irrelevant) {
            if (cfg->S_irr < 2 || cfg->S_irr > 255 || cfg->A_irr < 1) {
                g_create_err = "mdpp_create: irrelevant sub-space needs 2 <= S_irr <= 255, A_irr >= 1";
                free_all(h); delete h; return MDPP_EUNSUPPORTED;
            }
            TRY(alloc_zero(h, &h->d_P1, T * cfg->S_irr * cfg->A_irr));
            TRY(alloc_zero(h, &h->d_init_cdf1, T * cfg->S_irr * sizeof(double)));
            if (cfg->has_transition_noise)
                TRY(alloc_zero(h, &h->d_noise_cdf1, (size_t)cfg->S_irr * cfg->S_irr * sizeof(double)));
            TRY(alloc_zero(h, &h->d_irr_state, N * sizeof(uint32_t)));
        }
        TRY(alloc_zero(h, &h->d_P, T * cfg->S * cfg->A * (wide ? 2 : 1)));
        if (wide || lng) {
            TRY(alloc_zero(h, &h->d_hist_hi, N * sizeof(uint64_t)));
            TRYHIP(hipMemset(h->d_hist_hi, 0xFF, N * sizeof(uint64_t)));
        }
        TRY(alloc_zero(h, &h->d_is_term, T * cfg->S));
        TRY(alloc_zero(h, &h->d_init_cdf, T * cfg->S * sizeof(double)));
        if (cfg->unit_rewards) TRY(alloc_zero(h, &h->d_rbits, T * h->rbits_stride));
        else TRY(alloc_zero(h, &h->d_rtable, T * (size_t)h->nkeys * sizeof(double)));
        if (cfg->has_transition_noise) TRY(alloc_zero(h, &h->d_noise_cdf, (size_t)cfg->S * cfg->S * sizeof(double)));
        TRY(alloc_zero(h, &h->d_state, N * sizeof(uint4)));
        if (!cfg->unit_rewards && cfg->delay > 0) {
            TRY(alloc_zero(h, &h->d_ring, (size_t)cfg->delay * N * sizeof(uint32_t)));
            TRYHIP(hipMemset(h->d_ring, 0xFF, (size_t)cfg->delay * N * sizeof(uint32_t)));
        }
        DiscreteArgs &a = h->dargs;
        memset(&a, 0, sizeof(a));
        a.N = cfg->num_envs; a.S = cfg->S; a.A = cfg->A; a.L = cfg->L; a.delay = cfg->delay;
        a.every_n = cfg->every_n; a.shared_tables = (cfg->num_tables == 1); a.unit_rewards = cfg->unit_rewards;
        a.rew_sa = rew_sa ? 1 : 0;
        a.has_p_noise = cfg->has_transition_noise; a.has_r_noise = cfg->has_reward_noise;
        a.pn_T = cfg->has_transition_noise ? philox_pnoise_threshold(cfg->transition_noise) : 0u;
        a.pn_M = philox_pnoise_magic(a.pn_T, (uint32_t)cfg->S);
        a.pn_M1 = cfg->irrelevant ? philox_pnoise_magic(a.pn_T, (uint32_t)cfg->S_irr) : 0ull;
        a.autoreset = cfg->autoreset; a.max_steps = cfg->max_episode_steps;
        a.obs_i32 = (cfg->obs_dtype == MDPP_OBS_I32 || cfg->image);
        a.philox = (cfg->rng_mode == MDPP_RNG_PHILOX);
        a.nkeys = h->nkeys; a.philox_seed = cfg->philox_seed; a.env_id_offset = cfg->env_id_offset;
        a.r_noise = cfg->reward_noise; a.scale = cfg->reward_scale; a.shift = cfg->reward_shift;
        a.term_add = cfg->term_state_reward * cfg->reward_scale;
        a.P = (const uint8_t *)h->d_P; a.rtable = (const double *)h->d_rtable;
        a.rbits = (const uint8_t *)h->d_rbits; a.is_term = (const uint8_t *)h->d_is_term;
        a.init_cdf = (const double *)h->d_init_cdf; a.noise_cdf = (const double *)h->d_noise_cdf;
        a.rbits_stride = h->rbits_stride;
        // LDS carve for the shared-table path
        uint32_t off = 0;
        a.lds_P = off; off = align16(off + cfg->S * cfg->A);
        a.lds_term = off; off = align16(off + cfg->S);
        a.lds_init = off; off = align16(off + cfg->S * 8);
        uint32_t rew_bytes = cfg->unit_rewards ? h->rbits_stride : h->nkeys * 8u;
        a.rew_in_lds = rew_bytes <= 48u * 1024u;
        a.lds_rew = off; if (a.rew_in_lds) off = align16(off + rew_bytes);
        uint32_t noise_bytes = (uint32_t)cfg->S * cfg->S * 8u;
        a.noise_in_lds = cfg->has_transition_noise && noise_bytes <= 32u * 1024u;
        a.lds_noise = off; if (a.noise_in_lds) off = align16(off + noise_bytes);
        a.lds_bytes = off;
        a.state = (uint4 *)h->d_state; a.ring_keys = (uint32_t *)h->d_ring; a.hist_hi = (uint64_t *)h->d_hist_hi;
        a.env_s = (ulonglong2 *)h->d_rng_s[MDPP_STREAM_ENV]; a.env_inc = (ulonglong2 *)h->d_rng_inc[MDPP_STREAM_ENV];
        a.sp_s = (ulonglong2 *)h->d_rng_s[MDPP_STREAM_SPACE]; a.sp_inc = (ulonglong2 *)h->d_rng_inc[MDPP_STREAM_SPACE];
        a.status = (uint32_t *)h->d_status;
        a.est = mdpp::EpisodeStatsDev{(double *)h->d_est_cur, (double *)h->d_est_last, h->est_nk};
        a.irr = cfg->irrelevant ? 1 : 0; a.S1 = cfg->S_irr; a.A1 = cfg->A_irr;
        a.P1 = (const uint8_t *)h->d_P1; a.init_cdf1 = (const double *)h->d_init_cdf1;
        a.noise_cdf1 = (const double *)h->d_noise_cdf1; a.irr_state = (uint32_t *)h->d_irr_state;
        a.sp1_s = (ulonglong2 *)h->d_rng_s[MDPP_STREAM_SPACE_IRR];
        a.sp1_inc = (ulonglong2 *)h->d_rng_inc[MDPP_STREAM_SPACE_IRR];
    } else if (cfg->kind == MDPP_KIND_CONTINUOUS) {
        if (cfg->D < 1 || cfg->D > MDPP_MAX_DIM || cfg->order < 1 || cfg->order > MDPP_MAX_ORDER ||
            cfg->n_rel < 1 || cfg->n_rel > cfg->D || cfg->n_boxes < 0 ||
            // terminal hypercubes: [n_boxes][n_rel] packed into box_lo / box_hi -- as many as fit the arrays (64 at four
            // relevant dimensions); handles with picture observations draw them from a device list (mdpp_imagec.hip; round 6: no cap of its own)
            cfg->n_boxes > (MDPP_MAX_BOXES * MDPP_MAX_DIM) / cfg->n_rel) {   // (no int32 product to wrap)
            g_create_err = "mdpp_create: continuous needs 1 <= D <= 32, 1 <= order <= 4, n_boxes * n_rel <= 256";
            free_all(h); delete h; return MDPP_EUNSUPPORTED;
        }
        const bool line = cfg->reward_function == MDPP_CREWARD_MOVE_ALONG_A_LINE;
        if ((cfg->reward_function != MDPP_CREWARD_MOVE_TO_A_POINT && !line) ||
            (line && (cfg->L < 1 || cfg->L > 64 || cfg->image))) {
            g_create_err = "mdpp_create: move_along_a_line needs 1 <= L <= 64 and no image observations";
            free_all(h); delete h; return MDPP_EUNSUPPORTED;
        }
        const size_t D = (size_t)cfg->D;
        if (cfg->target_f64) {      // (ddot's summation order is the sequential one below 16 elements only)
            bool all_rel = !line && cfg->n_rel == cfg->D && cfg->D < 16;
            for (int j = 0; all_rel && j < cfg->n_rel; j++) all_rel = cfg->rel_idx[j] == j && cfg->target[j] == 0.0f;
            if (!all_rel) {
                g_create_err = "mdpp_create: target_f64 (the default target_point) needs move_to_a_point, every dimension relevant, state_space_dim < 16";
                free_all(h); delete h; return MDPP_EUNSUPPORTED;
            }
        }
        const bool rew64 = line || (cfg->target_f64 && cfg->make_denser);
        // (more than 8 relevant dimensions, or 5 to 8 of more than 12: the fit's matrices in an HBM workspace, rows as wide as n_rel)
        const bool line_big = line && (cfg->n_rel > 8 || (cfg->n_rel > 4 && cfg->D > 12));
        const size_t line_nl = line_big ? (size_t)cfg->n_rel : cfg->n_rel > 4 ? 8 : 4;   // row width of the history of relevant coordinates
        if (line) TRY(alloc_zero(h, &h->d_line_hist, (size_t)cfg->L * line_nl * N * sizeof(float)));
        if (line_big) TRY(alloc_zero(h, &h->d_line_ws, (size_t)(2 * cfg->n_rel * cfg->n_rel + 3 * cfg->n_rel) * N * sizeof(double)));
        if (rew64 && cfg->delay > 0) TRY(alloc_zero(h, &h->d_ring64, (size_t)cfg->delay * N * sizeof(double)));
        TRY(alloc_zero(h, &h->d_sd, (size_t)(cfg->order + 1) * D * N * sizeof(float)));
        TRY(alloc_zero(h, &h->d_cur, D * N * sizeof(float)));
        TRY(alloc_zero(h, &h->d_meta, N * sizeof(uint2)));
        if (cfg->delay > 0) {
            std::vector<uint32_t> init((size_t)cfg->delay * N, kRingPyZero);
            TRY(alloc_zero(h, &h->d_ring, init.size() * sizeof(uint32_t)));
            TRYHIP(hipMemcpy(h->d_ring, init.data(), init.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        ContinuousArgs &a = h->cargs;
        memset(&a, 0, sizeof(a));
        a.N = cfg->num_envs; a.D = cfg->D; a.n_rel = cfg->n_rel; a.order = cfg->order;
        a.delay = cfg->delay; a.every_n = cfg->every_n; a.make_denser = cfg->make_denser;
        a.has_p_noise = cfg->has_p_noise; a.has_r_noise = cfg->has_reward_noise;
        a.bounded = isfinite(cfg->state_space_max) ? 1 : 0;
        a.autoreset = cfg->autoreset; a.max_steps = cfg->max_episode_steps;
        a.philox = (cfg->rng_mode == MDPP_RNG_PHILOX); a.n_boxes = cfg->n_boxes;
        a.philox_seed = cfg->philox_seed; a.env_id_offset = cfg->env_id_offset;
        a.inertia32 = (float)cfg->inertia; a.amax32 = (float)cfg->action_space_max;
        a.smax32 = (float)cfg->state_space_max; a.radius32 = (float)cfg->target_radius;
        a.alw32 = (float)cfg->action_loss_weight;
        a.scale = cfg->reward_scale; a.shift = cfg->reward_shift;
        a.term_add = cfg->term_state_reward * cfg->reward_scale;
        a.scale32 = (float)a.scale; a.shift32 = (float)a.shift; a.term_add32 = (float)a.term_add;
        double f = 1.0;
        for (int k = 1; k <= cfg->order; k++) {
            f *= (double)k; a.fact[k] = f;
            a.tpow32[k] = (float)pow(cfg->time_unit, (double)k);
        }
        a.p_noise = cfg->p_noise; a.r_noise = cfg->reward_noise;
        a.reset_lo = (double)(-a.smax32); a.reset_range = (double)a.smax32 - (double)(-a.smax32);
        a.rel_prefix = 1;
        for (int j = 0; j < cfg->n_rel; j++) {
            if (cfg->rel_idx[j] < 0 || cfg->rel_idx[j] >= cfg->D) {
                g_create_err = "mdpp_create: relevant index out of range"; free_all(h); delete h; return MDPP_EINVAL;
            }
            a.rel[j] = cfg->rel_idx[j]; a.target[j] = cfg->target[j];
            if (cfg->rel_idx[j] != j) a.rel_prefix = 0;
        }
        for (int b = 0; b < cfg->n_boxes * cfg->n_rel; b++) { a.box_lo[b] = cfg->box_lo[b]; a.box_hi[b] = cfg->box_hi[b]; }
        {
            auto is_pow2 = [](double v) { int e; return v > 0 && isfinite(v) && frexp(v, &e) == 0.5; };
            a.inertia_pow2 = is_pow2((double)a.inertia32) ? 1u : 0u;
            a.inv_inertia32 = 1.0f / a.inertia32;
            a.fact_pow2_mask = 0;
            for (int k = 1; k <= cfg->order; k++) {
                a.inv_fact[k] = 1.0 / a.fact[k];
                if (is_pow2(a.fact[k])) a.fact_pow2_mask |= 1u << k;
            }
            // (an unbounded box is fine as long as the actions are bounded: states then stay finite, which
            // the fast kernel's clip relies on -- np.clip's NaN propagation lives in the general kernel)
            // (next-step autoreset: without noise or with Philox streams -- numpy noise streams are drawn ahead per step)
            // (has_p_noise: the continuous handle's transition-noise flag.  Until round 6 this line tested has_transition_noise, the
            //  discrete / grid field, and a next-step handle with transition noise ALONE on numpy streams went to the fused kernel,
            //  whose walker draws during the reset call too: found by tests/test_gpu_sweep.py's random configurations)
            const bool next_ok = cfg->autoreset != MDPP_AUTORESET_NEXT_STEP || cfg->rng_mode != MDPP_RNG_NUMPY_PCG64 ||
                                 (!cfg->has_p_noise && !cfg->has_reward_noise);
            // (the fused kernels test the hypercubes in an unrolled loop of MDPP_MAX_BOXES: more than that -> general kernel)
            a.fast_ok = (a.rel_prefix && !cfg->image && !line && !cfg->target_f64 && next_ok && cfg->n_boxes <= MDPP_MAX_BOXES &&
                         (a.bounded || isfinite(cfg->action_space_max))) ? 1u : 0u;
            a.image_quirk = cfg->image ? 1 : 0;
        }
        if (cfg->image) {   // scratch of one batch of img_chunk env steps: the states the pictures are made from
            TRY(alloc_zero(h, &h->d_img_state_out, 2 * (size_t)h->img_chunk * N * D * sizeof(float)));     // two sets (pipeline)
            {   // the rectangles the pictures draw: the first two relevant dimensions of every hypercube
                std::vector<float> bx((size_t)(cfg->n_boxes > 0 ? cfg->n_boxes : 1) * 4, 0.0f);
                for (int b = 0; b < cfg->n_boxes; b++)
                    for (int d = 0; d < 2 && d < cfg->n_rel; d++) {
                        bx[4 * b + d] = cfg->box_lo[b * cfg->n_rel + d];
                        bx[4 * b + 2 + d] = cfg->box_hi[b * cfg->n_rel + d];
                    }
                TRY(alloc_zero(h, &h->d_imgc_boxes, bx.size() * sizeof(float)));
                TRYHIP(hipMemcpy(h->d_imgc_boxes, bx.data(), bx.size() * sizeof(float), hipMemcpyHostToDevice));
            }
            TRY(alloc_zero(h, &h->d_img_state_final, 2 * (size_t)h->img_chunk * N * D * sizeof(float)));
        }
        a.sd = (float *)h->d_sd; a.cur = (float *)h->d_cur; a.meta = (uint2 *)h->d_meta;
        a.ring = (uint32_t *)h->d_ring;
        a.line_L = line ? cfg->L : 0; a.line_NL = (int32_t)line_nl; a.line_hist = (float *)h->d_line_hist; a.line_ws = (double *)h->d_line_ws; a.ring64 = (double *)h->d_ring64;
        a.target64 = cfg->target_f64 ? 1 : 0; a.rew64 = rew64 ? 1 : 0; a.radius = cfg->target_radius;
        a.est = mdpp::EpisodeStatsDev{(double *)h->d_est_cur, (double *)h->d_est_last, h->est_nk};
        if (cfg->episode_stats) a.fast_ok = 0;
        a.env_s = (ulonglong2 *)h->d_rng_s[MDPP_STREAM_ENV]; a.env_inc = (ulonglong2 *)h->d_rng_inc[MDPP_STREAM_ENV];
        a.sp_s = (ulonglong2 *)h->d_rng_s[MDPP_STREAM_SPACE]; a.sp_inc = (ulonglong2 *)h->d_rng_inc[MDPP_STREAM_SPACE];
        a.status = (uint32_t *)h->d_status;
        h->tables_ready = true;
    } else if (cfg->kind == MDPP_KIND_GRID) {
        bool ok = (cfg->grid_dims == 2 || cfg->grid_dims == 4) && cfg->delay == 0;
        for (int d = 0; ok && d < cfg->grid_dims; d++) ok = cfg->grid_shape[d] >= 1 && cfg->grid_shape[d] <= 254;
        if (ok && cfg->image)
            ok = cfg->img_w >= 1 && cfg->img_h >= 1 &&
                 cfg->img_r0 >= 1 && cfg->img_r0 <= 15 && cfg->n_boxes >= 0 && cfg->n_boxes <= (MDPP_MAX_BOXES * MDPP_MAX_DIM) / 2;
        if (!ok) {
            g_create_err = "mdpp_create: grid needs 2 (or 4) dimensions of 1..254 cells and delay 0; with image "
                           "observations also a disc radius of 1..15 and <= 128 terminal cells";
            free_all(h); delete h; return MDPP_EUNSUPPORTED;
        }
        TRY(alloc_zero(h, &h->d_state, N * sizeof(uint4)));
        if (cfg->image) {   // scratch of one batch of img_chunk env steps: the cells the pictures are made from
            TRY(alloc_zero(h, &h->d_img_state_out, 2 * (size_t)h->img_chunk * N * cfg->grid_dims * 4));     // two sets (pipeline)
            {   // the terminal cells the pictures draw (they ride in box_lo, two coordinates each)
                std::vector<float> bx((size_t)(cfg->n_boxes > 0 ? cfg->n_boxes : 1) * 4, 0.0f);
                for (int b = 0; b < cfg->n_boxes; b++) { bx[4 * b] = cfg->box_lo[2 * b]; bx[4 * b + 1] = cfg->box_lo[2 * b + 1]; }
                TRY(alloc_zero(h, &h->d_imgc_boxes, bx.size() * sizeof(float)));
                TRYHIP(hipMemcpy(h->d_imgc_boxes, bx.data(), bx.size() * sizeof(float), hipMemcpyHostToDevice));
            }
            TRY(alloc_zero(h, &h->d_img_state_final, 2 * (size_t)h->img_chunk * N * cfg->grid_dims * 4));
            TRY(alloc_zero(h, &h->d_img_tpl, (((size_t)(cfg->grid_dims / 2) * cfg->img_w * cfg->img_h + 15) / 16) * 2));   // grid-line bits
        }
        GridArgs &a = h->gargs;
        memset(&a, 0, sizeof(a));
        a.N = cfg->num_envs; a.G = cfg->grid_dims;
        for (int d = 0; d < cfg->grid_dims; d++) a.shape[d] = cfg->grid_shape[d];
        a.target[0] = cfg->grid_target[0]; a.target[1] = cfg->grid_target[1];
        a.make_denser = cfg->make_denser; a.has_p_noise = cfg->has_transition_noise;
        a.has_r_noise = cfg->has_reward_noise; a.every_n = cfg->every_n;
        a.autoreset = cfg->autoreset; a.max_steps = cfg->max_episode_steps;
        a.obs_i32 = (cfg->obs_dtype == MDPP_OBS_I32 || cfg->image);
        a.philox = (cfg->rng_mode == MDPP_RNG_PHILOX);
        a.philox_seed = cfg->philox_seed; a.env_id_offset = cfg->env_id_offset;
        a.p_noise = cfg->transition_noise; a.r_noise = cfg->reward_noise;
        a.scale = cfg->reward_scale; a.shift = cfg->reward_shift;
        a.term_add = cfg->term_state_reward * cfg->reward_scale;
        a.state = (uint4 *)h->d_state;
        a.env_s = (ulonglong2 *)h->d_rng_s[MDPP_STREAM_ENV]; a.env_inc = (ulonglong2 *)h->d_rng_inc[MDPP_STREAM_ENV];
        a.sp_s = (ulonglong2 *)h->d_rng_s[MDPP_STREAM_SPACE]; a.sp_inc = (ulonglong2 *)h->d_rng_inc[MDPP_STREAM_SPACE];
        a.act_s = (ulonglong2 *)h->d_rng_s[MDPP_STREAM_ACTION]; a.act_inc = (ulonglong2 *)h->d_rng_inc[MDPP_STREAM_ACTION];
        a.act_half = (uint2 *)h->d_rng_half;
        a.status = (uint32_t *)h->d_status;
        a.est = mdpp::EpisodeStatsDev{(double *)h->d_est_cur, (double *)h->d_est_last, h->est_nk};
        a.minv_lo = (uint64_t)pcg_mult_inverse(); a.minv_hi = (uint64_t)(pcg_mult_inverse() >> 64);
        h->tables_ready = true;      // a grid env has no tables
    } else {
        g_create_err = "mdpp_create: unknown kind"; free_all(h); delete h; return MDPP_EINVAL;
    }
#undef TRY
#undef TRYHIP
    *out = h;
    return MDPP_OK;
}

static inline bool t_fits53(uint64_t t) { return t <= (1ULL << 53) - 1; }

extern "C" int mdpp_upload_discrete_tables(mdpp_env *h, const uint8_t *P, const double *rtable,
                                           const uint8_t *rbits, const uint8_t *is_term,
                                           const double *init_cdf, const double *noise_cdf) {
    if (!h) return MDPP_EINVAL;
    if (h->cfg.kind != MDPP_KIND_DISCRETE) return fail(h, MDPP_EINVAL, "upload_discrete_tables: not a discrete handle");
    if (!P || !is_term || !init_cdf) return fail(h, MDPP_EINVAL, "upload_discrete_tables: null table");
    if (h->cfg.unit_rewards ? !rbits : !rtable) return fail(h, MDPP_EINVAL, "upload_discrete_tables: reward table missing");
    if (h->cfg.has_transition_noise && !noise_cdf) return fail(h, MDPP_EINVAL, "upload_discrete_tables: noise_cdf missing");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t T = (size_t)h->cfg.num_tables, S = (size_t)h->cfg.S, A = (size_t)h->cfg.A;
    const bool wide = S > 255;          // (P is then uint16[T][S][A] behind the same pointer)
    // every P entry must be a valid state id: the kernels index tables with it unchecked
    for (size_t k = 0; k < T * S * A; k++)
        if ((wide ? (size_t)((const uint16_t *)P)[k] : (size_t)P[k]) >= S) return fail(h, MDPP_EINVAL, "upload_discrete_tables: P entry out of range");
    HIPCHK(h, hipMemcpy(h->d_P, P, T * S * A * (wide ? 2 : 1), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_is_term, is_term, T * S, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_init_cdf, init_cdf, T * S * sizeof(double), hipMemcpyHostToDevice));
    if (h->cfg.unit_rewards)
        HIPCHK(h, hipMemcpy(h->d_rbits, rbits, T * h->rbits_stride, hipMemcpyHostToDevice));
    else
        HIPCHK(h, hipMemcpy(h->d_rtable, rtable, T * (size_t)h->nkeys * sizeof(double), hipMemcpyHostToDevice));
    if (h->cfg.has_transition_noise)
        HIPCHK(h, hipMemcpy(h->d_noise_cdf, noise_cdf, S * S * sizeof(double), hipMemcpyHostToDevice));
    if (wide || h->cfg.L > 7) {     // no specialised kernel serves such a handle (launch_discrete_step: mdpp_discrete_wide.hip / _long.hip)
        DiscreteArgs &a = h->dargs;
        a.shape_ok = a.fast_ok = a.shape_ok_irr = a.lean_next_ok = a.shape_ok_noise = a.shape_ok_noise_np = 0u;
        a.minv_lo = (uint64_t)pcg_mult_inverse(); a.minv_hi = (uint64_t)(pcg_mult_inverse() >> 64);
        memset(&h->s1args, 0, sizeof(h->s1args));
        h->tables_ready = true;
        return MDPP_OK;
    }
    // ---- derived constants of the fused fast path (see DiscreteArgs) ----
    {
        DiscreteArgs &a = h->dargs;
        const mdpp_config &c = h->cfg;
        a.shape_ok = (T == 1 && c.unit_rewards && !c.has_transition_noise && !c.has_reward_noise &&
                      c.L <= 3 && c.S <= 16 && c.delay <= 32 && c.autoreset != MDPP_AUTORESET_NEXT_STEP &&
                      a.rew_in_lds && !c.irrelevant) ? 1u : 0u;  // (image handles: the state kernel of a batch)
        a.fast_ok = (a.shape_ok && c.rng_mode == MDPP_RNG_NUMPY_PCG64) ? 1u : 0u;
        a.shape_ok_irr = (T == 1 && c.unit_rewards && !c.has_transition_noise && !c.has_reward_noise &&
                          c.L <= 3 && c.S <= 8 && c.delay <= 32 && c.autoreset != MDPP_AUTORESET_NEXT_STEP &&
                          a.rew_in_lds && c.irrelevant && !c.image) ? 1u : 0u;
        a.lean_next_ok = (T == 1 && c.unit_rewards && !c.has_transition_noise && !c.has_reward_noise &&
                          c.L <= 3 && c.S <= 8 && c.delay <= 32 && c.autoreset == MDPP_AUTORESET_NEXT_STEP &&
                          a.rew_in_lds && !c.image) ? 1u : 0u;
        a.shape_ok_noise = (T == 1 && c.unit_rewards && (c.has_transition_noise || c.has_reward_noise) &&
                            c.rng_mode == MDPP_RNG_PHILOX && c.L <= 3 && c.S <= 8 && c.S >= 2 && c.delay <= 32 &&
                            c.autoreset != MDPP_AUTORESET_NEXT_STEP && a.rew_in_lds && !c.irrelevant && !c.image) ? 1u : 0u;
        // numpy streams: the lean kernel serves the noisy shape when the S transition-noise categoricals have the
        // row-independent threshold form (DiscreteArgs::pn_TL / pn_TU)
        a.shape_ok_noise_np = (T == 1 && c.unit_rewards && (c.has_transition_noise || c.has_reward_noise) &&
                               c.rng_mode == MDPP_RNG_NUMPY_PCG64 && c.L <= 3 && c.S <= 8 && c.S >= 2 && c.delay <= 32 &&
                               c.autoreset != MDPP_AUTORESET_NEXT_STEP && a.rew_in_lds && !c.irrelevant && !c.image) ? 1u : 0u;
        for (int j = 0; j < 8; j++) a.pn_TL[j] = a.pn_TU[j] = ~0ULL;
        if (a.shape_ok_noise_np && c.has_transition_noise) {
            bool same = true;
            for (size_t j = 0; j < S; j++) {
                uint64_t tl = 0, tu = 0;
                bool hl = false, hu = false;
                for (size_t n = 0; n < S; n++) {
                    const uint64_t t = (uint64_t)ceil(ldexp(noise_cdf[n * S + j], 53));
                    if (n > j) { if (hl && tl != t) same = false; tl = t; hl = true; }
                    else { if (hu && tu != t) same = false; tu = t; hu = true; }
                }
                if (t_fits53(tl) && hl) a.pn_TL[j] = tl << 11;
                if (hu) a.pn_TU[j] = tu > (1ULL << 53) - 1 ? ~0ULL : tu << 11;      // (cdf 1.0: never reached by a 53-bit draw)
            }
            if (!same) a.shape_ok_noise_np = 0u;
        }
        if (c.episode_stats) a.shape_ok = a.fast_ok = a.shape_ok_irr = a.lean_next_ok = a.shape_ok_noise = a.shape_ok_noise_np = 0u;   // (general kernel keeps the statistics)
        a.s_shift = 0xFFFFFFFFu;
        for (uint32_t b = 1; b < 8; b++) if ((1u << b) == (uint32_t)c.S) a.s_shift = b;
        a.key_mask = h->nkeys - 1u;
        a.spow = 1; for (int j = 0; j < c.L - 1; j++) a.spow *= (uint32_t)c.S;
        a.term_mask = 0;
        if (c.S <= 64) for (size_t s = 0; s < S; s++) if (is_term[s]) a.term_mask |= 1ULL << s;
        for (int j = 0; j < 16; j++) a.init_thr[j] = ~0ULL;
        if (c.S <= 16)
            for (size_t j = 0; j < S; j++) a.init_thr[j] = (uint64_t)ceil(ldexp(init_cdf[j], 53));
        a.minv_lo = (uint64_t)pcg_mult_inverse(); a.minv_hi = (uint64_t)(pcg_mult_inverse() >> 64);
        for (int q = 0; q < 4; q++) {
            double r = (q & 2) ? 1.0 : 0.0;
            r *= a.scale; r += a.shift;
            if (q & 1) r += a.term_add;
            a.rsel[q] = (float)r;
        }
        // ---- one launch = one step (mdpp_discrete_step1.hip): the tables as ONE 1 KiB blob, the arguments as one small block ----
        Step1Args &s1 = h->s1args;
        memset(&s1, 0, sizeof(s1));
        const bool rew_sa_ = c.reward_kind == MDPP_REWARD_STATE_ACTION;
        if (a.shape_ok && c.every_n < (1 << 20)) {
            std::vector<uint32_t> blob(256, 0u);
            for (size_t act = 0; act < A && act < 16; act++) {
                uint64_t col = 0;
                for (size_t st = 0; st < S; st++) col |= (uint64_t)(P[st * A + act] & 0xF) << (4 * st);
                blob[2 * act] = (uint32_t)col; blob[2 * act + 1] = (uint32_t)(col >> 32);
            }
            for (int j = 0; j < 16; j++) { blob[32 + 2 * j] = (uint32_t)a.init_thr[j]; blob[32 + 2 * j + 1] = (uint32_t)(a.init_thr[j] >> 32); }
            for (uint32_t b = 0; b < h->rbits_stride && b < 512; b++) blob[64 + b / 4] |= (uint32_t)rbits[b] << (8 * (b % 4));
            for (size_t j = 0; j < 16; j++) {     // searchsorted(cdf, m31 2^-31, 'right') = #{j : ceil(cdf[j] 2^31) <= m31}
                const double t = j < S ? ceil(ldexp(init_cdf[j], 31)) : 4294967295.0;
                blob[192 + j] = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
            }
            // (kS1Replicas copies, one per group of workgroups: see mdpp_internal.hpp -- one copy by default)
            if (h->d_s1_blob) { (void)hipFree(h->d_s1_blob); h->d_s1_blob = nullptr; }
            HIPCHK(h, hipMalloc(&h->d_s1_blob, (size_t)kS1Replicas * 1024));
            for (int r = 0; r < kS1Replicas; r++)
                HIPCHK(h, hipMemcpy((char *)h->d_s1_blob + (size_t)r * 1024, blob.data(), 1024, hipMemcpyHostToDevice));
            s1.N = c.num_envs; s1.A = (uint32_t)c.A; s1.S = (uint32_t)c.S; s1.L = (uint32_t)c.L;
            s1.every_n = (uint32_t)c.every_n; s1.inv_every_n = 1.0 / (double)c.every_n;
            s1.max_steps = c.max_episode_steps > 0 ? (uint32_t)c.max_episode_steps : 0u;
            s1.delay = (uint32_t)c.delay; s1.autoreset = c.autoreset != MDPP_AUTORESET_DISABLED ? 1u : 0u;
            s1.term32 = (uint32_t)a.term_mask; s1.nan_mask = 0xFFu << (8 * c.L);
            // (one round per launch up to six entries; 1 / 6 / 0 rounds and a fill level of 1 measured the same, 2.48-2.54 us per
            //  step in a replayed graph of cfg2: profiles/r05_step1_variants.txt)
            s1.topup_rounds = 1; s1.topup_fill = 6;
            s1.philox_seed = a.philox_seed; s1.env_id_offset = a.env_id_offset;
            for (int q = 0; q < 4; q++) s1.rsel[q] = a.rsel[q];
            s1.blob = (const uint4 *)h->d_s1_blob;
            s1.state = a.state; s1.env_s = a.env_s; s1.env_inc = a.env_inc; s1.status = a.status;
        } else if (T == 1 && (c.unit_rewards ? c.delay <= 32 : (!rew_sa_ && !c.has_transition_noise && !c.has_reward_noise)) && c.L <= 3 &&
                   c.autoreset != MDPP_AUTORESET_NEXT_STEP && !c.irrelevant && !c.episode_stats && c.every_n < (1 << 20)) {
            // the same for state spaces beyond 16 states (k_discrete_step1w; the reference's 24- and 50-state sweeps): P as bytes,
            // terminal flags, rho_0 thresholds (64-bit for numpy's draw, 31-bit for a Philox word), reward bits -- one blob of at
            // most 8 KiB that a wave stages for itself
            const uint32_t S8 = ((uint32_t)S + 7u) & ~7u;
            const uint32_t off_term = align16((uint32_t)(S * A)), off_thr = align16(off_term + (uint32_t)S);
            const uint32_t off_thr31 = off_thr + S8 * 8u, off_rew = align16(off_thr31 + S8 * 4u);
            // + 256 buckets over the top 8 bits of the uniform (the handle's RNG: 53-bit numpy draw / 31-bit Philox word):
            //   {thresholds at or below the bucket's first value, thresholds strictly inside it}
            // (rewards that are not all 1.0: the float64 table by sequence key instead of the bit table)
            const uint32_t rew_bytes = c.unit_rewards ? h->rbits_stride : h->nkeys * 8u;
            const uint32_t off_bk = align16(off_rew + rew_bytes);
            // (noise: the thresholds of the S transition-noise categoricals, numpy's ziggurat tables for the reward noise)
            const bool np_streams = c.rng_mode == MDPP_RNG_NUMPY_PCG64;
            const uint32_t off_tn = off_bk + 512u;
            const uint32_t off_zig = off_tn + ((c.has_transition_noise && np_streams) ? (uint32_t)S * S8 * 8u : 0u);
            const uint32_t bytes = (off_zig + ((c.has_reward_noise && np_streams) ? 6144u : 0u) + 1023u) & ~1023u;
            // (what the selected instantiation stages: k_discrete_step1w<NZ> holds 12 rounds of 1 KiB in registers, the
            //  noise-free ones 8 -- a longer blob would leave the tail of the reward table and the buckets unwritten in LDS)
            const uint32_t max_rounds = (c.has_transition_noise || c.has_reward_noise) ? kS1wRoundsNoise : kS1wRounds;
            if (bytes <= max_rounds * 1024u) {
                std::vector<uint8_t> blob(bytes, 0);
                memcpy(blob.data(), P, S * A);
                memcpy(blob.data() + off_term, is_term, S);
                for (uint32_t j = 0; j < S8; j++) {
                    const uint64_t t64 = j < S ? (uint64_t)ceil(ldexp(init_cdf[j], 53)) : ~0ULL;
                    const double t31 = j < S ? ceil(ldexp(init_cdf[j], 31)) : 4294967295.0;
                    const uint32_t t32 = t31 >= 4294967295.0 ? 4294967295u : (uint32_t)t31;
                    memcpy(blob.data() + off_thr + 8u * j, &t64, 8);
                    memcpy(blob.data() + off_thr31 + 4u * j, &t32, 4);
                }
                if (c.unit_rewards) memcpy(blob.data() + off_rew, rbits, h->rbits_stride);
                else memcpy(blob.data() + off_rew, rtable, (size_t)h->nkeys * 8u);
                {
                    const bool ph = c.rng_mode == MDPP_RNG_PHILOX;
                    const int shift = ph ? 23 : 45;
                    for (uint32_t b = 0; b < 256u; b++) {
                        const uint64_t lo = (uint64_t)b << shift, hi = lo + (1ULL << shift);
                        uint32_t c0 = 0, nin = 0;
                        for (uint32_t j = 0; j < S; j++) {
                            uint64_t t;
                            if (ph) { uint32_t t32; memcpy(&t32, blob.data() + off_thr31 + 4u * j, 4); t = t32; }
                            else memcpy(&t, blob.data() + off_thr + 8u * j, 8);
                            c0 += t <= lo ? 1u : 0u;
                            nin += (t > lo && t < hi) ? 1u : 0u;
                        }
                        const uint16_t e = (uint16_t)(c0 | (nin << 8));
                        memcpy(blob.data() + off_bk + 2u * b, &e, 2);
                    }
                }
                if (c.has_transition_noise && np_streams)
                    for (uint32_t row = 0; row < S; row++)
                        for (uint32_t j = 0; j < S8; j++) {
                            const uint64_t t64 = j < S ? (uint64_t)ceil(ldexp(noise_cdf[row * S + j], 53)) : ~0ULL;
                            memcpy(blob.data() + off_tn + 8u * (row * S8 + j), &t64, 8);
                        }
                if (c.has_reward_noise && np_streams) {
                    static const uint64_t ki[256] = NPZ_KI_INIT;
                    static const double wi[256] = NPZ_WI_INIT, fi[256] = NPZ_FI_INIT;
                    memcpy(blob.data() + off_zig, ki, 2048); memcpy(blob.data() + off_zig + 2048, wi, 2048);
                    memcpy(blob.data() + off_zig + 4096, fi, 2048);
                }
                if (h->d_s1_blob) { (void)hipFree(h->d_s1_blob); h->d_s1_blob = nullptr; }
                HIPCHK(h, hipMalloc(&h->d_s1_blob, (size_t)kS1Replicas * bytes));
                for (int r = 0; r < kS1Replicas; r++)
                    HIPCHK(h, hipMemcpy((char *)h->d_s1_blob + (size_t)r * bytes, blob.data(), bytes, hipMemcpyHostToDevice));
                s1.N = c.num_envs; s1.A = (uint32_t)c.A; s1.S = (uint32_t)c.S; s1.L = (uint32_t)c.L;
                s1.every_n = (uint32_t)c.every_n; s1.inv_every_n = 1.0 / (double)c.every_n;
                s1.max_steps = c.max_episode_steps > 0 ? (uint32_t)c.max_episode_steps : 0u;
                s1.delay = (uint32_t)c.delay; s1.autoreset = c.autoreset != MDPP_AUTORESET_DISABLED ? 1u : 0u;
                s1.nan_mask = 0xFFu << (8 * c.L);
                s1.philox_seed = a.philox_seed; s1.env_id_offset = a.env_id_offset;
                for (int q = 0; q < 4; q++) s1.rsel[q] = a.rsel[q];
                s1.blob = (const uint4 *)h->d_s1_blob;
                s1.wide = 1; s1.blob_rounds = bytes / 1024u; s1.off_term = off_term; s1.off_thr = off_thr; s1.off_thr31 = off_thr31;
                s1.off_rew = off_rew; s1.S8 = S8; s1.off_bk = off_bk;
                s1.has_p_noise = c.has_transition_noise ? 1u : 0u; s1.has_r_noise = c.has_reward_noise ? 1u : 0u;
                s1.off_tn = off_tn; s1.off_zig = off_zig; s1.pn_T = a.pn_T; s1.pn_M = a.pn_M; s1.r_noise = a.r_noise;
                s1.sp_s = a.sp_s; s1.sp_inc = a.sp_inc;
                s1.unit = c.unit_rewards ? 1u : 0u; s1.ring_keys = a.ring_keys;
                s1.scale = a.scale; s1.shift = a.shift; s1.term_add = a.term_add;
                s1.state = a.state; s1.env_s = a.env_s; s1.env_inc = a.env_inc; s1.status = a.status;
            }
        }
    }
    h->tables_ready = true;
    return MDPP_OK;
}

extern "C" int mdpp_upload_discrete_irrelevant(mdpp_env *h, const uint8_t *P1, const double *init_cdf1,
                                               const double *noise_cdf1) {
    if (!h) return MDPP_EINVAL;
    if (h->cfg.kind != MDPP_KIND_DISCRETE || !h->cfg.irrelevant)
        return fail(h, MDPP_EINVAL, "upload_discrete_irrelevant: not a handle with an irrelevant sub-space");
    if (!P1 || !init_cdf1) return fail(h, MDPP_EINVAL, "upload_discrete_irrelevant: null table");
    if (h->cfg.has_transition_noise && !noise_cdf1) return fail(h, MDPP_EINVAL, "upload_discrete_irrelevant: noise_cdf missing");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t T = (size_t)h->cfg.num_tables, S1 = (size_t)h->cfg.S_irr, A1 = (size_t)h->cfg.A_irr;
    for (size_t k = 0; k < T * S1 * A1; k++)
        if (P1[k] >= S1) return fail(h, MDPP_EINVAL, "upload_discrete_irrelevant: P entry out of range");
    HIPCHK(h, hipMemcpy(h->d_P1, P1, T * S1 * A1, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_init_cdf1, init_cdf1, T * S1 * sizeof(double), hipMemcpyHostToDevice));
    if (h->cfg.has_transition_noise)
        HIPCHK(h, hipMemcpy(h->d_noise_cdf1, noise_cdf1, S1 * S1 * sizeof(double), hipMemcpyHostToDevice));
    h->irr_ready = true;
    return MDPP_OK;
}

extern "C" int mdpp_upload_image_disc(mdpp_env *h, const uint8_t *disc) {
    if (!h || !disc) return MDPP_EINVAL;
    if ((h->cfg.kind != MDPP_KIND_CONTINUOUS && h->cfg.kind != MDPP_KIND_GRID) || !h->cfg.image)
        return fail(h, MDPP_EINVAL, "upload_image_disc: not a continuous / grid handle with image observations");
    const int T = 2 * h->cfg.img_r0 + 1;
    for (int r = 0; r < 32; r++) h->imgc_disc_rows[r] = 0;
    for (int dy = 0; dy < T; dy++)
        for (int dx = 0; dx < T; dx++)
            if (disc[dy * T + dx]) h->imgc_disc_rows[dy] |= 1u << dx;
    h->img_ready = h->cfg.kind == MDPP_KIND_CONTINUOUS || h->img_lines_ready;
    return MDPP_OK;
}

extern "C" int mdpp_upload_image_lines(mdpp_env *h, const uint8_t *lines) {
    if (!h || !lines) return MDPP_EINVAL;
    if (h->cfg.kind != MDPP_KIND_GRID || !h->cfg.image)
        return fail(h, MDPP_EINVAL, "upload_image_lines: not a grid handle with image observations");
    const size_t npix = (size_t)(h->cfg.grid_dims / 2) * h->cfg.img_w * h->cfg.img_h;
    std::vector<uint16_t> bits((npix + 15) / 16, 0);
    for (size_t p = 0; p < npix; p++)
        if (lines[p]) bits[p >> 4] |= (uint16_t)(1u << (p & 15));
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(h->d_img_tpl, bits.data(), bits.size() * 2, hipMemcpyHostToDevice));
    h->img_lines_ready = true;
    bool disc = false;
    for (int r = 0; r < 32; r++) disc = disc || h->imgc_disc_rows[r] != 0;
    h->img_ready = disc;
    return MDPP_OK;
}

extern "C" int mdpp_get_state_grid(mdpp_env *h, int32_t *cells, int32_t *steps, uint8_t *reached) {
    if (!h || !cells || !steps || !reached) return MDPP_EINVAL;
    if (h->cfg.kind != MDPP_KIND_GRID) return fail(h, MDPP_EINVAL, "get_state_grid: not a grid handle");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    const size_t N = (size_t)h->cfg.num_envs;
    const int G = h->cfg.grid_dims;
    std::vector<uint32_t> st(4 * N);
    HIPCHK(h, hipMemcpy(st.data(), h->d_state, N * 16, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; i++) {
        for (int d = 0; d < G; d++) cells[i * G + d] = (int32_t)((st[4 * i] >> (8 * d)) & 0xFF);
        steps[i] = (int32_t)st[4 * i + 1];
        reached[i] = (uint8_t)(st[4 * i + 2] & 1u);                  // (bit 1: next-step autoreset pending)
    }
    return MDPP_OK;
}

extern "C" int mdpp_set_state_grid(mdpp_env *h, const int32_t *cells, const int32_t *steps, const uint8_t *reached) {
    if (!h || !cells || !steps) return MDPP_EINVAL;
    if (h->cfg.kind != MDPP_KIND_GRID) return fail(h, MDPP_EINVAL, "set_state_grid: not a grid handle");
    const size_t N = (size_t)h->cfg.num_envs;
    const int G = h->cfg.grid_dims;
    std::vector<uint32_t> st(4 * N, 0);
    for (size_t i = 0; i < N; i++) {
        for (int d = 0; d < G; d++) {
            // a reset can leave the index one past the grid (Box.sample of an int box), like the reference
            if (cells[i * G + d] < 0 || cells[i * G + d] > h->cfg.grid_shape[d])
                return fail(h, MDPP_EINVAL, "set_state_grid: cell out of range");
            st[4 * i] |= (uint32_t)cells[i * G + d] << (8 * d);
        }
        st[4 * i + 1] = (uint32_t)steps[i];
        st[4 * i + 2] = reached ? (uint32_t)(reached[i] & 1u) : 0u;
    }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(h->d_state, st.data(), N * 16, hipMemcpyHostToDevice));
    return MDPP_OK;
}

extern "C" int mdpp_get_state_irrelevant(mdpp_env *h, int32_t *irr) {
    if (!h || !irr) return MDPP_EINVAL;
    if (h->cfg.kind != MDPP_KIND_DISCRETE || !h->cfg.irrelevant) return fail(h, MDPP_EINVAL, "get_state_irrelevant: no irrelevant sub-space");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(irr, h->d_irr_state, (size_t)h->cfg.num_envs * 4, hipMemcpyDeviceToHost));
    return MDPP_OK;
}

extern "C" int mdpp_set_state_irrelevant(mdpp_env *h, const int32_t *irr) {
    if (!h || !irr) return MDPP_EINVAL;
    if (h->cfg.kind != MDPP_KIND_DISCRETE || !h->cfg.irrelevant) return fail(h, MDPP_EINVAL, "set_state_irrelevant: no irrelevant sub-space");
    for (int i = 0; i < h->cfg.num_envs; i++)
        if (irr[i] < 0 || irr[i] >= h->cfg.S_irr) return fail(h, MDPP_EINVAL, "set_state_irrelevant: state out of range");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(h->d_irr_state, irr, (size_t)h->cfg.num_envs * 4, hipMemcpyHostToDevice));
    return MDPP_OK;
}

extern "C" int mdpp_seed_streams(mdpp_env *h, int stream, const uint64_t *words) {
    if (!h || !words) return MDPP_EINVAL;
    if (h->cfg.rng_mode != MDPP_RNG_NUMPY_PCG64) return fail(h, MDPP_ESTATE, "seed_streams: handle is in Philox mode");
    if (stream < 0 || stream >= MDPP_NUM_STREAMS || !h->d_rng_s[stream]) return fail(h, MDPP_EINVAL, "seed_streams: bad stream");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t N = (size_t)h->cfg.num_envs;
    std::vector<uint64_t> st(2 * N), inc(2 * N);
    std::vector<uint32_t> half(2 * N);
    for (size_t i = 0; i < N; i++) {
        st[2 * i] = words[6 * i]; st[2 * i + 1] = words[6 * i + 1];
        inc[2 * i] = words[6 * i + 2]; inc[2 * i + 1] = words[6 * i + 3];
        half[2 * i] = (uint32_t)words[6 * i + 4]; half[2 * i + 1] = (uint32_t)words[6 * i + 5];
    }
    HIPCHK(h, hipMemcpy(h->d_rng_s[stream], st.data(), N * 16, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_rng_inc[stream], inc.data(), N * 16, hipMemcpyHostToDevice));
    if ((stream == MDPP_STREAM_IMAGE || stream == MDPP_STREAM_ACTION) && h->d_rng_half)
        HIPCHK(h, hipMemcpy(h->d_rng_half, half.data(), N * 8, hipMemcpyHostToDevice));
    if (stream == MDPP_STREAM_ENV && h->cfg.kind == MDPP_KIND_DISCRETE && h->dargs.fast_ok) {
        // start states drawn ahead from the old stream are void: empty every env's queue.  (Only the
        // packed-nibble kernels keep a queue in word 1 of the record; for every other handle that word
        // holds history bytes 4-7 and must survive a re-seed.)
        HIPCHK(h, hipDeviceSynchronize());
        HIPCHK(h, hipMemset2D((char *)h->d_state + 4, 16, 0, 4, N));
    }
    h->streams_ready[stream] = true;
    return MDPP_OK;
}

extern "C" int mdpp_get_streams(mdpp_env *h, int stream, uint64_t *words) {
    if (!h || !words) return MDPP_EINVAL;
    if (h->cfg.rng_mode != MDPP_RNG_NUMPY_PCG64) return fail(h, MDPP_ESTATE, "get_streams: handle is in Philox mode");
    if (stream < 0 || stream >= MDPP_NUM_STREAMS || !h->d_rng_s[stream]) return fail(h, MDPP_EINVAL, "get_streams: bad stream");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    const size_t N = (size_t)h->cfg.num_envs;
    std::vector<uint64_t> st(2 * N), inc(2 * N);
    std::vector<uint32_t> half(2 * N, 0);
    HIPCHK(h, hipMemcpy(st.data(), h->d_rng_s[stream], N * 16, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(inc.data(), h->d_rng_inc[stream], N * 16, hipMemcpyDeviceToHost));
    if ((stream == MDPP_STREAM_IMAGE || stream == MDPP_STREAM_ACTION) && h->d_rng_half)
        HIPCHK(h, hipMemcpy(half.data(), h->d_rng_half, N * 8, hipMemcpyDeviceToHost));
    std::vector<uint32_t> rec;
    const bool queued = stream == MDPP_STREAM_ENV && h->cfg.kind == MDPP_KIND_DISCRETE && h->dargs.fast_ok;
    if (queued) {
        rec.resize(4 * N);
        HIPCHK(h, hipMemcpy(rec.data(), h->d_state, N * 16, hipMemcpyDeviceToHost));
    }
    for (size_t i = 0; i < N; i++) {
        unsigned __int128 s = ((unsigned __int128)st[2 * i + 1] << 64) | st[2 * i];
        const unsigned __int128 c = ((unsigned __int128)inc[2 * i + 1] << 64) | inc[2 * i];
        if (queued) // un-draw the start states still waiting in the queue: s_prev = (s - inc) * M^-1
            for (uint32_t q = (rec[4 * i + 1] >> 24) & 7u; q > 0; q--) s = (s - c) * pcg_mult_inverse();
        words[6 * i] = (uint64_t)s; words[6 * i + 1] = (uint64_t)(s >> 64);
        words[6 * i + 2] = inc[2 * i]; words[6 * i + 3] = inc[2 * i + 1];
        words[6 * i + 4] = half[2 * i]; words[6 * i + 5] = half[2 * i + 1];
    }
    return MDPP_OK;
}

extern "C" int mdpp_set_options(mdpp_env *h, uint32_t disable_mask) {
    if (!h) return MDPP_EINVAL;
    h->opts = disable_mask;
    return MDPP_OK;
}

extern "C" int mdpp_get_episode_stats(mdpp_env *h, double *current, double *last) {
    if (!h) return MDPP_EINVAL;
    if (!h->d_est_cur) return fail(h, MDPP_ESTATE, "get_episode_stats: the handle was not created with episode_stats");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    const size_t N = (size_t)h->cfg.num_envs;
    if (current) HIPCHK(h, hipMemcpy(current, h->d_est_cur, (size_t)h->est_nk * N * sizeof(double), hipMemcpyDeviceToHost));
    if (last) HIPCHK(h, hipMemcpy(last, h->d_est_last, (size_t)(h->est_nk + 1) * N * sizeof(double), hipMemcpyDeviceToHost));
    return MDPP_OK;
}

// ---- HIP graphs of single steps: the host-side step counter ---------------------------------------------
// mdpp_step hands the counter (`tick`) to its launch BY VALUE: the head of a delay line kept in memory is
// tick mod delay, and Philox streams are keyed by it.  A captured launch therefore replays with the value it was
// captured with.
extern "C" int mdpp_graph_replay_exact(mdpp_env *h, int K) {
    if (!h || K < 1) return MDPP_EINVAL;
    const int d = h->cfg.delay;
    const bool ring_in_memory = d > 0 && (h->cfg.kind == MDPP_KIND_CONTINUOUS ||
                                          (h->cfg.kind == MDPP_KIND_DISCRETE && !h->cfg.unit_rewards));
    // by value: numpy streams (a Philox key would repeat), and no ring head that moves between replays
    if (h->cfg.rng_mode != MDPP_RNG_PHILOX && !(ring_in_memory && K % d != 0)) return 1;
    // otherwise: through the device-side offset (mdpp_graph_capture / mdpp_graph_set_tick_offset) -- every step and rollout
    // kernel reads it, and (round 5) the kernels that draw the image transforms of a step do too
    return 2;
}

namespace mdpp {
__global__ void k_set_tick_offset(uint64_t *p, uint64_t v) { *p = v; }
}

// Capture mode: launches made while it is on carry a pointer to the handle's device word `tick offset` and add it to
// the step counter they were given by value (mdpp_internal.hpp tick_from_device).
extern "C" int mdpp_graph_capture(mdpp_env *h, int on) {
    if (!h) return MDPP_EINVAL;
    h->graph_capture = on != 0;
    return MDPP_OK;
}

// offset = (step counter now) - (step counter when the graph was captured), enqueued on `stream` BEFORE the graph launch.
extern "C" int mdpp_graph_set_tick_offset(mdpp_env *h, int64_t offset, void *stream) {
    if (!h || !h->d_tick_off) return MDPP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(mdpp::k_set_tick_offset, dim3(1), dim3(1), 0, (hipStream_t)stream, (uint64_t *)h->d_tick_off, (uint64_t)offset);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("k_set_tick_offset: ") + hipGetErrorString(e); return MDPP_EHIP; }
    return MDPP_OK;
}

extern "C" int mdpp_tick(mdpp_env *h, int64_t advance, uint64_t *tick_out) {
    if (!h) return MDPP_EINVAL;
    if (advance < 0 && (uint64_t)(-advance) > h->tick) return fail(h, MDPP_EINVAL, "mdpp_tick: the counter would become negative");
    h->tick = (uint64_t)((int64_t)h->tick + advance);
    if (tick_out) *tick_out = h->tick;
    return MDPP_OK;
}

// ---- next-step autoreset: the "episode ended on the previous call" flag of every env ---------------------
// (bit 31 of the discrete step counter, bit 1 of the grid / continuous flag words; the state getters mask it out)
static int pending_word(const mdpp_env *h, void **buf, size_t *stride_words, size_t *word, uint32_t *bit) {
    if (h->cfg.kind == MDPP_KIND_DISCRETE) { *buf = h->d_state; *stride_words = 4; *word = 2; *bit = 0x80000000u; }
    else if (h->cfg.kind == MDPP_KIND_GRID) { *buf = h->d_state; *stride_words = 4; *word = 2; *bit = 2u; }
    else { *buf = h->d_meta; *stride_words = 2; *word = 1; *bit = 2u; }
    return MDPP_OK;
}

extern "C" int mdpp_get_reset_pending(mdpp_env *h, uint8_t *pending) {
    if (!h || !pending) return MDPP_EINVAL;
    void *buf; size_t sw, wd; uint32_t bit;
    pending_word(h, &buf, &sw, &wd, &bit);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    const size_t N = (size_t)h->cfg.num_envs;
    std::vector<uint32_t> st(sw * N);
    HIPCHK(h, hipMemcpy(st.data(), buf, st.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; i++) pending[i] = (st[sw * i + wd] & bit) ? 1 : 0;
    return MDPP_OK;
}

extern "C" int mdpp_set_reset_pending(mdpp_env *h, const uint8_t *pending) {
    if (!h || !pending) return MDPP_EINVAL;
    const size_t N = (size_t)h->cfg.num_envs;
    if (h->cfg.autoreset != MDPP_AUTORESET_NEXT_STEP) {
        for (size_t i = 0; i < N; i++)
            if (pending[i]) return fail(h, MDPP_EINVAL, "set_reset_pending: the handle does not use next-step autoreset");
        return MDPP_OK;
    }
    void *buf; size_t sw, wd; uint32_t bit;
    pending_word(h, &buf, &sw, &wd, &bit);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    std::vector<uint32_t> st(sw * N);
    HIPCHK(h, hipMemcpy(st.data(), buf, st.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; i++) st[sw * i + wd] = (st[sw * i + wd] & ~bit) | (pending[i] ? bit : 0u);
    HIPCHK(h, hipMemcpy(buf, st.data(), st.size() * 4, hipMemcpyHostToDevice));
    return MDPP_OK;
}

static int check_ready(mdpp_env *h, const char *what) {
    if (!h->tables_ready) return fail(h, MDPP_ESTATE, std::string(what) + ": tables not uploaded");
    if (h->cfg.rng_mode == MDPP_RNG_NUMPY_PCG64) {
        if (!h->streams_ready[MDPP_STREAM_ENV] || !h->streams_ready[MDPP_STREAM_SPACE])
            return fail(h, MDPP_ESTATE, std::string(what) + ": RNG streams not seeded");
        if (h->cfg.image && h->cfg.kind == MDPP_KIND_DISCRETE && !h->streams_ready[MDPP_STREAM_IMAGE])
            return fail(h, MDPP_ESTATE, std::string(what) + ": image RNG stream not seeded");
        if (h->cfg.image && h->cfg.kind == MDPP_KIND_DISCRETE && !h->img_ready)
            return fail(h, MDPP_ESTATE, std::string(what) + ": image templates not uploaded");
        if (h->cfg.kind == MDPP_KIND_GRID && !h->streams_ready[MDPP_STREAM_ACTION])
            return fail(h, MDPP_ESTATE, std::string(what) + ": action-space RNG stream not seeded");
        if (h->cfg.kind == MDPP_KIND_DISCRETE && h->cfg.irrelevant && !h->streams_ready[MDPP_STREAM_SPACE_IRR])
            return fail(h, MDPP_ESTATE, std::string(what) + ": irrelevant sub-space RNG stream not seeded");
    }
    if (h->cfg.image && h->cfg.kind != MDPP_KIND_DISCRETE && !h->img_ready)
        return fail(h, MDPP_ESTATE, std::string(what) + ": image disc raster (and grid-line mask) not uploaded");
    if (h->cfg.kind == MDPP_KIND_DISCRETE && h->cfg.irrelevant && !h->irr_ready)
        return fail(h, MDPP_ESTATE, std::string(what) + ": irrelevant sub-space tables not uploaded");
    return MDPP_OK;
}

extern "C" int mdpp_reset(mdpp_env *h, const uint8_t *mask_dev, void *obs_dev, void *stream) {
    if (!h) return MDPP_EINVAL;
    int rc = check_ready(h, "mdpp_reset");
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    if (h->cfg.kind == MDPP_KIND_DISCRETE) {
        if (h->cfg.image) {
            rc = launch_discrete_reset(h, mask_dev, h->d_img_state_out, s);
            if (rc) return rc;
            return launch_image_obs(h, 1, (const int32_t *)h->d_img_state_out, nullptr, nullptr, nullptr,
                                    mask_dev, (uint8_t *)obs_dev, nullptr, s);
        }
        return launch_discrete_reset(h, mask_dev, obs_dev, s);
    }
    if (h->cfg.kind == MDPP_KIND_GRID) {
        if (!h->cfg.image) return launch_grid_reset(h, mask_dev, obs_dev, s);
        rc = launch_grid_reset(h, mask_dev, h->d_img_state_out, s);
        if (rc || !obs_dev) return rc;
        return launch_imagec_obs(h, 1, h->d_img_state_out, nullptr, nullptr, nullptr, mask_dev, (uint8_t *)obs_dev, nullptr, s);
    }
    if (h->cfg.image) {
        rc = launch_continuous_reset(h, mask_dev, (float *)h->d_img_state_out, s);
        if (rc || !obs_dev) return rc;
        return launch_imagec_obs(h, 1, h->d_img_state_out, nullptr, nullptr, nullptr, mask_dev,
                                 (uint8_t *)obs_dev, nullptr, s);
    }
    return launch_continuous_reset(h, mask_dev, (float *)obs_dev, s);
}

// Image rollouts run as batches of up to img_chunk steps.  With two or more batches they form a
// two-stage pipeline: `prepare` (the state kernel of batch b + 1, and for polygon images its draw and
// record kernels: latency-bound, a few dozen waves) goes to the handle's side stream while `render`
// of batch b owns the memory system on the caller's stream.  The side stream starts behind everything
// already queued on `s`, every prepared batch is waited for by its renderer on `s` (so all work is
// joined back into `s`, also under graph capture), and a scratch set (buf = b & 1) is reused only
// after its renderer has finished.
template <class Prepare, class Render>
static int image_batches(mdpp_env *h, int K, hipStream_t s, Prepare prepare, Render render) {
    // batch sizes: img_chunk, but a long rollout starts with 8, 16, 32, ... steps -- nothing hides the first batch's prepare
    // stage (its draw kernel walks the steps serially), so it is kept short and the renderer starts early; every later
    // batch's prepare stage (about 4.6 us per step) then fits behind the render of the batch before it (about 12 us per step)
    const bool ramp = K >= 4 * h->img_chunk && h->img_chunk >= 32 && !(h->opts & MDPP_OPT_NO_IMG_OVERLAP);
    auto size_of = [&](int b, int k0) {
        int want = h->img_chunk;
        if (ramp && b < 4 && (8 << b) < want) want = 8 << b;
        return K - k0 < want ? K - k0 : want;
    };
    int nb = 0;
    for (int k0 = 0; k0 < K; nb++) k0 += size_of(nb, k0);
    const bool overlap = nb >= 2 && h->side_stream && !(h->opts & MDPP_OPT_NO_IMG_OVERLAP);
    hipStream_t s2 = overlap ? h->side_stream : s;
    if (overlap) {
        HIPCHK(h, hipEventRecord(h->ev_entry, s));
        HIPCHK(h, hipStreamWaitEvent(s2, h->ev_entry, 0));
    }
    int k0 = 0;
    for (int b = 0; b < nb; b++) {
        const int buf = overlap ? (b & 1) : 0;
        const int kc = size_of(b, k0);
        if (overlap && b >= 2) HIPCHK(h, hipStreamWaitEvent(s2, h->ev_render[buf], 0));
        int rc = prepare(k0, kc, buf, s2);
        if (rc) return rc;
        if (overlap) {
            HIPCHK(h, hipEventRecord(h->ev_side[buf], s2));
            HIPCHK(h, hipStreamWaitEvent(s, h->ev_side[buf], 0));
        }
        rc = render(k0, kc, buf, s, overlap);
        if (rc) return rc;
        if (overlap) HIPCHK(h, hipEventRecord(h->ev_render[buf], s));
        k0 += kc;
    }
    return MDPP_OK;
}

static int step_common(mdpp_env *h, int K, const void *actions, void *obs, float *reward,
                       uint8_t *term, uint8_t *trunc, void *final_obs, void *stream) {
    if (!h) return MDPP_EINVAL;
    if (K < 1) return fail(h, MDPP_EINVAL, "step: K < 1");
    if (!actions || !obs || !reward || !term || !trunc) return fail(h, MDPP_EINVAL, "step: null buffer");
    int rc = check_ready(h, "mdpp_step");
    if (rc) return rc;
    if (h->line_hist_stale)
        return fail(h, MDPP_EUNSUPPORTED, "step: mdpp_set_state_continuous on a move_along_a_line handle must be followed by mdpp_set_line_history (the window of the line fit)");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    if (h->cfg.kind == MDPP_KIND_DISCRETE) {
        if (h->cfg.image) {
            // batches of up to img_chunk env steps: one state kernel (states only, a few bytes per
            // env step), one transform-draw kernel, one render kernel over steps x envs images
            // (W*H bytes each)
            const size_t N = (size_t)h->cfg.num_envs, aw = h->cfg.irrelevant ? 2 : 1;
            const size_t isz = aw * h->cfg.img_w * h->cfg.img_h;
            const size_t sset = (size_t)h->img_chunk * N * aw;            // int32 per scratch set
            auto scratch = [&](int buf, int32_t **so, int32_t **sf) {
                *so = (int32_t *)h->d_img_state_out + buf * sset;
                *sf = (int32_t *)h->d_img_state_final + buf * sset;
            };
            if (K == 1) {
                // one step: the state kernel, then ONE kernel that draws, builds the record and renders (k_image_step1)
                int32_t *so, *sf;
                scratch(0, &so, &sf);
                int r = launch_discrete_step(h, 1, (const int32_t *)actions, so, reward, term, trunc, sf, s);
                if (r) return r;
                r = launch_image_step1(h, so, sf, term, trunc, (uint8_t *)obs, (uint8_t *)final_obs, s);
                if (r != 0) return r < 0 ? r : MDPP_OK;
                {
                    r = launch_image_obs(h, 1, so, sf, term, trunc, nullptr, nullptr, nullptr, s, 1, 0);
                    if (r) return r;
                    r = launch_image_obs(h, 1, so, sf, term, trunc, nullptr, (uint8_t *)obs, (uint8_t *)final_obs, s, 6, 0);
                }
                return r;
            }
            return image_batches(
                h, K, s,
                [&](int k0, int kc, int buf, hipStream_t st) {
                    const size_t off = (size_t)k0 * N;
                    int32_t *so, *sf;
                    scratch(buf, &so, &sf);
                    int r = launch_discrete_step(h, kc, (const int32_t *)actions + off * aw, so, reward + off, term + off,
                                                 trunc + off, sf, st);
                    if (r) return r;
                    return launch_image_obs(h, kc, so, sf, term + off, trunc + off, nullptr, nullptr, nullptr, st, 1, buf);
                },
                [&](int k0, int kc, int buf, hipStream_t st, bool overlap) {
                    const size_t off = (size_t)k0 * N;
                    int32_t *so, *sf;
                    scratch(buf, &so, &sf);
                    // (phase 2 leaves render slots free for the next batch's state kernel: only when pipelined)
                    return launch_image_obs(h, kc, so, sf, term + off, trunc + off, nullptr, (uint8_t *)obs + off * isz,
                                            final_obs ? (uint8_t *)final_obs + off * isz : nullptr, st, overlap ? 2 : 6, buf);
                });
        }
        return launch_discrete_step(h, K, (const int32_t *)actions, obs, reward, term, trunc, final_obs, s);
    }
    if (h->cfg.kind == MDPP_KIND_GRID && !h->cfg.image)
        return launch_grid_step(h, K, (const int32_t *)actions, obs, reward, term, trunc, final_obs, s);
    if (h->cfg.kind == MDPP_KIND_GRID) {
        const size_t N = (size_t)h->cfg.num_envs, G = (size_t)h->cfg.grid_dims;
        const size_t isz = (G / 2) * h->cfg.img_w * h->cfg.img_h * 3;
        const size_t sset = (size_t)h->img_chunk * N * G;                 // int32 per scratch set
        return image_batches(
            h, K, s,
            [&](int k0, int kc, int buf, hipStream_t st) {
                const size_t off = (size_t)k0 * N;
                return launch_grid_step(h, kc, (const int32_t *)actions + off * G, (int32_t *)h->d_img_state_out + buf * sset,
                                        reward + off, term + off, trunc + off, (int32_t *)h->d_img_state_final + buf * sset, st);
            },
            [&](int k0, int kc, int buf, hipStream_t st, bool) {
                const size_t off = (size_t)k0 * N;
                return launch_imagec_obs(h, kc, (int32_t *)h->d_img_state_out + buf * sset,
                                         (int32_t *)h->d_img_state_final + buf * sset, term + off, trunc + off, nullptr,
                                         (uint8_t *)obs + off * isz, final_obs ? (uint8_t *)final_obs + off * isz : nullptr, st);
            });
    }
    if (h->cfg.image) {
        // batches of up to img_chunk env steps: one state kernel, one render kernel over steps x envs
        // pictures (3 W H bytes per 2-D sub-space each)
        const size_t N = (size_t)h->cfg.num_envs, D = (size_t)h->cfg.D;
        const size_t isz = (size_t)(D > 2 ? 2 : 1) * h->cfg.img_w * h->cfg.img_h * 3;
        const size_t sset = (size_t)h->img_chunk * N * D;                 // floats per scratch set
        return image_batches(
            h, K, s,
            [&](int k0, int kc, int buf, hipStream_t st) {
                const size_t off = (size_t)k0 * N;
                return launch_continuous_step(h, kc, (const float *)actions + off * D, (float *)h->d_img_state_out + buf * sset,
                                              reward + off, term + off, trunc + off, (float *)h->d_img_state_final + buf * sset, st);
            },
            [&](int k0, int kc, int buf, hipStream_t st, bool) {
                const size_t off = (size_t)k0 * N;
                return launch_imagec_obs(h, kc, (float *)h->d_img_state_out + buf * sset, (float *)h->d_img_state_final + buf * sset,
                                         term + off, trunc + off, nullptr, (uint8_t *)obs + off * isz,
                                         final_obs ? (uint8_t *)final_obs + off * isz : nullptr, st);
            });
    }
    return launch_continuous_step(h, K, (const float *)actions, (float *)obs, reward, term, trunc,
                                  (float *)final_obs, s);
}

extern "C" int mdpp_step(mdpp_env *h, const void *actions, void *obs, float *reward, uint8_t *term,
                         uint8_t *trunc, void *final_obs, void *stream) {
    return step_common(h, 1, actions, obs, reward, term, trunc, final_obs, stream);
}

extern "C" int mdpp_step_n(mdpp_env *h, int K, const void *actions, void *obs, float *reward,
                           uint8_t *term, uint8_t *trunc, void *stream) {
    return step_common(h, K, actions, obs, reward, term, trunc, nullptr, stream);
}

extern "C" const char *mdpp_kernel_name(mdpp_env *h, int K) {
    if (!h) return "";
    h->kname[0] = 0;
    if (K < 1 || check_ready(h, "mdpp_kernel_name")) return h->kname;
    if (h->cfg.image) {      // the renderer is the dominant kernel of an image rollout
        snprintf(h->kname, sizeof h->kname, "%s", h->cfg.kind == MDPP_KIND_DISCRETE ? image_obs_kernel_name(h, K)
                 : h->cfg.kind == MDPP_KIND_GRID ? "k_imagec_obs<GRID=1>" : "k_imagec_obs<GRID=0>");
        return h->kname;
    }
    if (h->cfg.kind == MDPP_KIND_DISCRETE)
        (void)launch_discrete_step(h, K, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, h->kname);
    else if (h->cfg.kind == MDPP_KIND_GRID)
        (void)launch_grid_step(h, K, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, h->kname);
    else
        (void)launch_continuous_step(h, K, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, h->kname);
    return h->kname;
}

// Python round(v, 15) for |v| <= 1: correctly rounded decimal conversion, like Pillow's rotate().
static double round15(double v) {
    char buf[64];
    snprintf(buf, sizeof buf, "%.15f", v);
    return strtod(buf, nullptr);
}

extern "C" int mdpp_upload_image_templates(mdpp_env *h, const uint8_t *tpl, int32_t n_radii,
                                           int32_t n_cls_x, int32_t n_cls_y, const int16_t *cls_x,
                                           const int16_t *cls_y) {
    if (!h || !tpl || !cls_x || !cls_y) return MDPP_EINVAL;
    if (!h->cfg.image) return fail(h, MDPP_EINVAL, "upload_image_templates: not an image handle");
    const mdpp_config &c = h->cfg;
    if (n_radii != c.img_r_max - c.img_r_min + 1 || n_cls_x < 1 || n_cls_y < 1)
        return fail(h, MDPP_EINVAL, "upload_image_templates: radii/classes do not match the config");
    // templates cover the states of the larger sub-space (state s is drawn as an (s + 3)-gon in either)
    const size_t S = (size_t)((c.irrelevant && c.S_irr > c.S) ? c.S_irr : c.S);
    if (S * n_radii * n_cls_x * n_cls_y >= (1u << 20) || c.img_r_max > 1023 || c.img_w > 65535 || c.img_h > 65535)
        return fail(h, MDPP_EUNSUPPORTED, "upload_image_templates: more than 2^20 templates, R > 1023 or a side > 65535");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t W = (size_t)c.img_w, H = (size_t)c.img_h;
    const size_t tb = S * n_radii * n_cls_x * n_cls_y * (size_t)c.img_tpl_size * c.img_tpl_size;
    for (size_t k = 0; k < S * n_radii * W; k++)
        if (cls_x[k] >= n_cls_x) return fail(h, MDPP_EINVAL, "upload_image_templates: cls_x out of range");
    for (size_t k = 0; k < S * n_radii * H; k++)
        if (cls_y[k] >= n_cls_y) return fail(h, MDPP_EINVAL, "upload_image_templates: cls_y out of range");
    for (void **p : {&h->d_img_tpl, &h->d_img_tplp, &h->d_img_clsx, &h->d_img_clsy, &h->d_img_rot, &h->d_img_near})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    HIPCHK(h, hipMalloc(&h->d_img_tpl, tb));
    HIPCHK(h, hipMemcpy(h->d_img_tpl, tpl, tb, hipMemcpyHostToDevice));
    // k_image_obs_fast (mdpp_image.hip render_fast) applies when: dword rows (H % 4 == 0) and
    // 16-byte chunks (W H % 16 == 0); the template inside its zero border fits a 64-byte LDS
    // column; the image columns a polygon's bounding circle can span fit the wave's 6 KiB of LDS;
    // every polygon raster stays within +-R of its centre and the centre keeps R + 1 clear of the
    // image edge (shift draws |v| <= W/2 - R - 1, image_multi_discrete.py:172-181), so the polygon
    // never leaves the image.
    {
        const int PADW = 8, t = c.img_tpl_size, tp = t + 2 * PADW, half = t / 2;
        const int span = 2 * c.img_r_max + 9 + 4 + 1;      // columns of the near circle's box, at most
        // (round 5) templates past 64 bytes: 128-byte LDS columns, two waves per workgroup, no image columns in LDS
        // (k_image_obs_wide; its chunk -> column division by multiply-shift needs W (H / 4)^2 < 2^24)
        const int colb = tp <= 64 ? 64 : 128;
        const int coldw = span * (c.img_h / 4) + 8 + 4;                                                   /* (+ 4: the zero chunk of render_fast_store) */
        bool ok = (c.img_h % 4 == 0) && ((size_t)c.img_w * c.img_h) % 16 == 0 && tp <= 128 &&
                  (size_t)c.img_w * c.img_h <= (1u << 20) &&
                  (colb == 64 ? coldw <= 1536 : (size_t)c.img_w * (c.img_h / 4) * (c.img_h / 4) < (1u << 24)) &&
                  c.img_r_max <= (c.img_w < c.img_h ? c.img_w : c.img_h) / 2 - 1;
        h->img_colb = colb;
        const size_t ntpl = S * n_radii * n_cls_x * n_cls_y;
        for (size_t k = 0; ok && k < ntpl; k++) {
            const int R = c.img_r_min + (int)((k / ((size_t)n_cls_x * n_cls_y)) % n_radii);
            const uint8_t *src = tpl + k * (size_t)t * t;
            for (int y = 0; y < t && ok; y++)
                for (int x = 0; x < t; x++)
                    if (src[y * t + x] && (abs(x - half) > R || abs(y - half) > R)) { ok = false; break; }
        }
        h->img_fast_ok = ok;
        if (ok) {
            std::vector<uint8_t> padded(ntpl * (size_t)tp * colb, 0);
            for (size_t k = 0; k < ntpl; k++)
                for (int y = 0; y < t; y++)
                    memcpy(&padded[(k * tp + y + PADW) * colb + PADW], tpl + (k * t + y) * (size_t)t, t);
            HIPCHK(h, hipMalloc(&h->d_img_tplp, padded.size()));
            HIPCHK(h, hipMemcpy(h->d_img_tplp, padded.data(), padded.size(), hipMemcpyHostToDevice));
            // The near dwords of a polygon as a TABLE (one radius only: no scale transform).  The renderer evaluates the dwords
            // (column x, dword row q) whose centre lies within R + 4.5 of the polygon's centre; relative to the centre rounded to
            // whole pixels (off by <= 0.71 px) that set is contained in the disc of radius R + 4.5 + 0.71 -- enumerated here once
            // per phase of the centre row in the dword grid (cy mod 4), in two orders (x fastest / y fastest: the renderer picks
            // the one along which its lanes read along template rows), 8 entries (dx, dq as int8 pairs) per lane:
            // [order][phase][lane][8].  Every pixel of a listed dword is within R + 5.21 + 1.5 + 0.71 < R + 8 of the centre
            // after the map's rounding: inside the template's zero border.
            if (n_radii == 1 && colb == 64) {
                const int R = c.img_r_max;
                const double rt = (double)R + 4.5 + 0.7072, rt2 = rt * rt;
                std::vector<uint16_t> tab((size_t)2 * 4 * 64 * 8, (uint16_t)0x8080u);      // (dx = dq = -128: never inside a box)
                bool fits = true;
                for (int o = 0; o < 2 && fits; o++)
                    for (int ph = 0; ph < 4 && fits; ph++) {
                        std::vector<std::pair<int, int>> e;
                        const int lim = R + 6;
                        for (int dq = -(lim / 4 + 2); dq <= lim / 4 + 2; dq++)
                            for (int dx = -lim; dx <= lim; dx++) {
                                const double dy = 4.0 * dq + 1.5 - (double)ph;
                                if ((double)dx * dx + dy * dy <= rt2) e.emplace_back(dx, dq);
                            }
                        if (e.size() > 512) { fits = false; break; }
                        if (o == 1) std::sort(e.begin(), e.end());                                       // y fastest: by (dx, dq)
                        else std::sort(e.begin(), e.end(), [](const std::pair<int, int> &u, const std::pair<int, int> &v) {
                                 return u.second != v.second ? u.second < v.second : u.first < v.first; });    // x fastest: by (dq, dx)
                        for (size_t k = 0; k < e.size(); k++)
                            tab[(((size_t)o * 4 + ph) * 64 + (k & 63)) * 8 + (k >> 6)] =
                                (uint16_t)(((uint32_t)e[k].first & 0xFFu) | (((uint32_t)e[k].second & 0xFFu) << 8));
                    }
                if (fits) {
                    HIPCHK(h, hipMalloc(&h->d_img_near, tab.size() * 2));
                    HIPCHK(h, hipMemcpy(h->d_img_near, tab.data(), tab.size() * 2, hipMemcpyHostToDevice));
                }
            }
        }
    }
    HIPCHK(h, hipMalloc(&h->d_img_clsx, S * n_radii * W * 2));
    HIPCHK(h, hipMemcpy(h->d_img_clsx, cls_x, S * n_radii * W * 2, hipMemcpyHostToDevice));
    HIPCHK(h, hipMalloc(&h->d_img_clsy, S * n_radii * H * 2));
    HIPCHK(h, hipMemcpy(h->d_img_clsy, cls_y, S * n_radii * H * 2, hipMemcpyHostToDevice));
    // Pillow Image.rotate(angle) -> transform(AFFINE, NEAREST): fixed-point coefficients per angle
    // (PIL/Image.py rotate(); libImaging/Geometry.c affine_fixed, third-party, Pillow 12.2.0)
    std::vector<int32_t> rot(360 * 6);
    const double cxr = c.img_w / 2.0, cyr = c.img_h / 2.0, PI = 3.141592653589793;
    for (int ang = 0; ang < 360; ang++) {
        const double a = -((double)ang * (PI / 180.0));
        const double m0 = round15(cos(a)), m1 = round15(sin(a)), m3 = round15(-sin(a)), m4 = round15(cos(a));
        const double m2 = (m0 * (-cxr) + m1 * (-cyr) + 0.0) + cxr;
        const double m5 = (m3 * (-cxr) + m4 * (-cyr) + 0.0) + cyr;
        auto FIX = [](double v) { return (int32_t)floor(v * 65536.0 + 0.5); };
        int32_t *r = &rot[ang * 6];
        r[0] = FIX(m0); r[1] = FIX(m1); r[3] = FIX(m3); r[4] = FIX(m4);
        r[2] = FIX(m2 + m0 * 0.5 + m1 * 0.5); r[5] = FIX(m5 + m3 * 0.5 + m4 * 0.5);
        // Image.rotate() transposes instead for these angles: exact integer maps, same form
        const int W1 = c.img_w - 1, H1 = c.img_h - 1, ONE = 65536;
        if (ang == 0) { r[0] = ONE; r[1] = 0; r[2] = 0; r[3] = 0; r[4] = ONE; r[5] = 0; }
        if (ang == 180) { r[0] = -ONE; r[1] = 0; r[2] = W1 * ONE; r[3] = 0; r[4] = -ONE; r[5] = H1 * ONE; }
        if (ang == 90 && c.img_w == c.img_h) { r[0] = 0; r[1] = -ONE; r[2] = W1 * ONE; r[3] = ONE; r[4] = 0; r[5] = 0; }
        if (ang == 270 && c.img_w == c.img_h) { r[0] = 0; r[1] = ONE; r[2] = 0; r[3] = -ONE; r[4] = 0; r[5] = H1 * ONE; }
    }
    HIPCHK(h, hipMalloc(&h->d_img_rot, rot.size() * 4));
    HIPCHK(h, hipMemcpy(h->d_img_rot, rot.data(), rot.size() * 4, hipMemcpyHostToDevice));
    h->img_n_radii = n_radii; h->img_n_cls_x = n_cls_x; h->img_n_cls_y = n_cls_y;
    h->img_ready = true;
    return MDPP_OK;
}

// ---- state export / import ------------------------------------------------------------------
extern "C" int mdpp_get_state_discrete(mdpp_env *h, int32_t *hist, int32_t *steps, double *ring) {
    if (!h || h->cfg.kind != MDPP_KIND_DISCRETE) return MDPP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    const size_t N = (size_t)h->cfg.num_envs;
    const int L = h->cfg.L, d = h->cfg.delay;
    std::vector<uint32_t> st(4 * N);
    HIPCHK(h, hipMemcpy(st.data(), h->d_state, N * 16, hipMemcpyDeviceToHost));
    std::vector<uint32_t> keys;
    std::vector<double> rt;
    if (!h->cfg.unit_rewards && d > 0) {
        keys.resize((size_t)d * N);
        HIPCHK(h, hipMemcpy(keys.data(), h->d_ring, keys.size() * 4, hipMemcpyDeviceToHost));
        const size_t T = (size_t)h->cfg.num_tables;
        rt.resize(T * h->nkeys);
        HIPCHK(h, hipMemcpy(rt.data(), h->d_rtable, rt.size() * 8, hipMemcpyDeviceToHost));
    }
    const bool wide = h->cfg.S > 255;   // (16-bit history fields, the older four in d_hist_hi)
    const bool lng = h->cfg.L > 7;      // (sixteen byte fields, the older eight in d_hist_hi)
    std::vector<uint64_t> hhi;
    if (wide || lng) { hhi.resize(N); HIPCHK(h, hipMemcpy(hhi.data(), h->d_hist_hi, N * 8, hipMemcpyDeviceToHost)); }
    for (size_t i = 0; i < N; i++) {
        uint64_t hb = ((uint64_t)st[4 * i + 1] << 32) | st[4 * i];
        if (hist && wide)
            for (int j = 0; j <= L; j++) {
                const int f = L - j;
                const uint32_t b = (uint32_t)((f < 4 ? hb >> (16 * f) : hhi[i] >> (16 * (f - 4))) & 0xFFFFu);
                hist[i * (L + 1) + j] = (b == 0xFFFFu) ? -1 : (int32_t)b;
            }
        else if (hist)
            for (int j = 0; j <= L; j++) { // hist[0] oldest ... hist[L] newest
                const int f = L - j;
                uint32_t b = (uint32_t)((f < 8 ? hb >> (8 * f) : hhi[i] >> (8 * (f - 8))) & 0xFF);
                hist[i * (L + 1) + j] = (b == 0xFF) ? -1 : (int32_t)b;
            }
        if (steps) steps[i] = (int32_t)(st[4 * i + 2] & 0x7FFFFFFFu);    // (bit 31: next-step autoreset pending)
        if (ring) {
            for (int j = 0; j < d; j++) { // ring[0] pays out next
                double v;
                if (h->cfg.unit_rewards) v = ((st[4 * i + 3] >> (d - 1 - j)) & 1u) ? 1.0 : 0.0;
                else {
                    uint32_t k = keys[(size_t)((h->tick + j) % d) * N + i];
                    v = (k == kNoKey) ? 0.0 : rt[(h->cfg.num_tables == 1 ? 0 : i) * h->nkeys + k];
                }
                ring[i * d + j] = v;
            }
        }
    }
    return MDPP_OK;
}

extern "C" int mdpp_set_state_discrete(mdpp_env *h, const int32_t *hist, const int32_t *steps,
                                       const double *ring) {
    if (!h || h->cfg.kind != MDPP_KIND_DISCRETE || !hist || !steps) return MDPP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    const size_t N = (size_t)h->cfg.num_envs;
    const int L = h->cfg.L, d = h->cfg.delay;
    std::vector<uint32_t> st(4 * N);
    HIPCHK(h, hipMemcpy(st.data(), h->d_state, N * 16, hipMemcpyDeviceToHost));
    // the kernels rely on: every entry a valid state id or a NaN slot, NaN slots a contiguous OLDEST
    // prefix (what reset() + steps produce, rl_toy_env.py:2275-2278, :2050-2052), the newest entry valid
    for (size_t i = 0; i < N; i++) {
        bool seen_valid = false;
        for (int j = 0; j <= L; j++) {
            const int32_t v = hist[i * (L + 1) + j];
            if (v >= h->cfg.S) return fail(h, MDPP_EINVAL, "set_state_discrete: state id out of range");
            if (v >= 0) seen_valid = true;
            else if (seen_valid) return fail(h, MDPP_EINVAL, "set_state_discrete: a NaN slot newer than a valid state");
        }
        if (!seen_valid) return fail(h, MDPP_EINVAL, "set_state_discrete: the current state is a NaN slot");
    }
    if (ring && !h->cfg.unit_rewards && d > 0) {
        // the delay line of non-unit rewards holds sequence KEYS (values are looked up at pop time): a
        // value is imported as any key that pays exactly that value, 0.0 as "no key"
        const size_t T = (size_t)h->cfg.num_tables;
        std::vector<double> rt(T * h->nkeys);
        HIPCHK(h, hipMemcpy(rt.data(), h->d_rtable, rt.size() * 8, hipMemcpyDeviceToHost));
        std::vector<uint32_t> keys((size_t)d * N);
        for (size_t i = 0; i < N; i++) {
            const double *tab = &rt[(T == 1 ? 0 : i) * (size_t)h->nkeys];
            for (int j = 0; j < d; j++) {
                const double v = ring[i * d + j];
                uint32_t key = kNoKey;
                if (v != 0.0) {
                    for (uint32_t k = 0; k < h->nkeys && key == kNoKey; k++) if (tab[k] == v) key = k;
                    if (key == kNoKey)
                        return fail(h, MDPP_EINVAL, "set_state_discrete: a reward_buffer value that no rewardable sequence pays");
                }
                keys[(size_t)((h->tick + j) % d) * N + i] = key;
            }
        }
        HIPCHK(h, hipMemcpy(h->d_ring, keys.data(), keys.size() * 4, hipMemcpyHostToDevice));
    }
    const bool wide = h->cfg.S > 255, lng = L > 7;
    std::vector<uint64_t> hhi((wide || lng) ? N : 0);
    for (size_t i = 0; i < N; i++) {
        uint64_t hb = ~0ULL, hh = ~0ULL;
        for (int j = 0; j <= L; j++) {
            int32_t v = hist[i * (L + 1) + j];
            if (wide) { hh = (hh << 16) | (hb >> 48); hb = (hb << 16) | (uint64_t)(v < 0 ? 0xFFFF : v); }
            else { hh = (hh << 8) | (hb >> 56); hb = (hb << 8) | (uint64_t)(v < 0 ? 0xFF : v); }
        }
        if (wide || lng) hhi[i] = hh;
        st[4 * i] = (uint32_t)hb;
        if (!h->dargs.fast_ok) st[4 * i + 1] = (uint32_t)(hb >> 32); // fast path: word 1 is the draw queue
        st[4 * i + 2] = (uint32_t)steps[i];
        if (ring && h->cfg.unit_rewards) {
            uint32_t bits = 0;
            for (int j = 0; j < d; j++) bits |= (ring[i * d + j] != 0.0 ? 1u : 0u) << (d - 1 - j);
            st[4 * i + 3] = bits;
        }
    }
    HIPCHK(h, hipMemcpy(h->d_state, st.data(), N * 16, hipMemcpyHostToDevice));
    if (wide || lng) HIPCHK(h, hipMemcpy(h->d_hist_hi, hhi.data(), N * 8, hipMemcpyHostToDevice));
    return MDPP_OK;
}

extern "C" int mdpp_get_state_continuous(mdpp_env *h, float *derivs, float *cur, int32_t *steps,
                                         double *ring, uint8_t *ring_is32, uint8_t *reached) {
    if (!h || h->cfg.kind != MDPP_KIND_CONTINUOUS) return MDPP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    const size_t N = (size_t)h->cfg.num_envs, D = (size_t)h->cfg.D;
    const int n = h->cfg.order, d = h->cfg.delay;
    if (derivs) {
        std::vector<float> sd((size_t)(n + 1) * D * N);
        HIPCHK(h, hipMemcpy(sd.data(), h->d_sd, sd.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < N; i++)
            for (int k = 0; k <= n; k++)
                for (size_t c = 0; c < D; c++) derivs[(i * (n + 1) + k) * D + c] = sd[((size_t)k * D + c) * N + i];
    }
    if (cur) {
        std::vector<float> cu(D * N);
        HIPCHK(h, hipMemcpy(cu.data(), h->d_cur, cu.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < N; i++) for (size_t c = 0; c < D; c++) cur[i * D + c] = cu[c * N + i];
    }
    if (steps || reached) {
        std::vector<uint32_t> me(2 * N);
        HIPCHK(h, hipMemcpy(me.data(), h->d_meta, N * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < N; i++) {
            if (steps) steps[i] = (int32_t)me[2 * i];
            if (reached) reached[i] = (uint8_t)(me[2 * i + 1] & 1u);
        }
    }
    if (ring && d > 0 && h->cargs.rew64) {           // move_along_a_line / the default float64 target: a float64 delay line
        std::vector<double> rg((size_t)d * N);
        HIPCHK(h, hipMemcpy(rg.data(), h->d_ring64, rg.size() * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < N; i++)
            for (int j = 0; j < d; j++) {
                ring[i * d + j] = rg[(size_t)((h->tick + j) % d) * N + i];
                if (ring_is32) ring_is32[i * d + j] = 0;
            }
    } else if (ring && d > 0) {
        std::vector<uint32_t> rg((size_t)d * N);
        HIPCHK(h, hipMemcpy(rg.data(), h->d_ring, rg.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < N; i++)
            for (int j = 0; j < d; j++) {
                uint32_t b = rg[(size_t)((h->tick + j) % d) * N + i];
                float f; memcpy(&f, &b, 4);
                ring[i * d + j] = (b == kRingPyZero) ? 0.0 : (double)f;
                if (ring_is32) ring_is32[i * d + j] = (b != kRingPyZero);
            }
    }
    return MDPP_OK;
}

extern "C" int mdpp_set_state_continuous(mdpp_env *h, const float *derivs, const float *cur,
                                         const int32_t *steps, const double *ring,
                                         const uint8_t *ring_is32, const uint8_t *reached) {
    if (!h || h->cfg.kind != MDPP_KIND_CONTINUOUS || !derivs || !cur || !steps) return MDPP_EINVAL;
    // move_along_a_line fits the last sequence_length states, which this call does not carry: stepping is refused
    // until mdpp_set_line_history() has restored them for the restored step counters
    if (h->cargs.line_L) h->line_hist_stale = true;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    const size_t N = (size_t)h->cfg.num_envs, D = (size_t)h->cfg.D;
    const int n = h->cfg.order, d = h->cfg.delay;
    std::vector<float> sd((size_t)(n + 1) * D * N), cu(D * N);
    std::vector<uint32_t> me(2 * N);
    for (size_t i = 0; i < N; i++) {
        for (int k = 0; k <= n; k++)
            for (size_t c = 0; c < D; c++) sd[((size_t)k * D + c) * N + i] = derivs[(i * (n + 1) + k) * D + c];
        for (size_t c = 0; c < D; c++) cu[c * N + i] = cur[i * D + c];
        me[2 * i] = (uint32_t)steps[i];
        me[2 * i + 1] = reached ? (reached[i] ? 1u : 0u) : 0u;
    }
    HIPCHK(h, hipMemcpy(h->d_sd, sd.data(), sd.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_cur, cu.data(), cu.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_meta, me.data(), N * 8, hipMemcpyHostToDevice));
    if (ring && d > 0 && h->cargs.rew64) {          // the default float64 target with a dense reward: a float64 delay line
        std::vector<double> rg((size_t)d * N);
        for (size_t i = 0; i < N; i++)
            for (int j = 0; j < d; j++) rg[(size_t)((h->tick + j) % d) * N + i] = ring[i * d + j];
        HIPCHK(h, hipMemcpy(h->d_ring64, rg.data(), rg.size() * 8, hipMemcpyHostToDevice));
    } else if (ring && d > 0) {
        std::vector<uint32_t> rg((size_t)d * N);
        for (size_t i = 0; i < N; i++)
            for (int j = 0; j < d; j++) {
                uint32_t b = kRingPyZero;
                if (!ring_is32 || ring_is32[i * d + j]) { float f = (float)ring[i * d + j]; memcpy(&b, &f, 4); }
                rg[(size_t)((h->tick + j) % d) * N + i] = b;
            }
        HIPCHK(h, hipMemcpy(h->d_ring, rg.data(), rg.size() * 4, hipMemcpyHostToDevice));
    }
    return MDPP_OK;
}

// move_along_a_line: the window of the line fit -- the relevant coordinates of the last L = sequence_length states of
// every env (rl_toy_env.py:1865-1872 reads them from augmented_state) --, oldest first, [N][L][n_rel] float32 on the
// host.  Slots older than the running episode (fewer than L - 1 transitions since its reset) hold NaN, as the
// reference's NaN-filled list does.  Device layout: line_hist[slot = s % L][NL][N], s = transitions made when the
// state was reached (mdpp_continuous.hip c_line_put), so the export reads the per-env step counters.
extern "C" int mdpp_get_line_history(mdpp_env *h, float *hist) {
    if (!h || h->cfg.kind != MDPP_KIND_CONTINUOUS || !hist) return MDPP_EINVAL;
    if (!h->cargs.line_L) return fail(h, MDPP_EUNSUPPORTED, "get_line_history: reward_function is not move_along_a_line");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    const size_t N = (size_t)h->cfg.num_envs, L = (size_t)h->cargs.line_L, NL = (size_t)h->cargs.line_NL, nr = (size_t)h->cfg.n_rel;
    std::vector<float> lh(L * NL * N);
    std::vector<uint32_t> me(2 * N);
    HIPCHK(h, hipMemcpy(lh.data(), h->d_line_hist, lh.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(me.data(), h->d_meta, N * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; i++) {
        const size_t steps = me[2 * i];
        for (size_t j = 0; j < L; j++) {                    // j-th oldest of the L newest: reached after steps - (L - 1) + j transitions
            const bool live = steps + j + 1 >= L;
            const size_t slot = (steps + 1 + j) % L;
            for (size_t c = 0; c < nr; c++)
                hist[(i * L + j) * nr + c] = live ? lh[(slot * NL + c) * N + i] : __builtin_nanf("");
        }
    }
    return MDPP_OK;
}

// Inverse of mdpp_get_line_history for the step counters the handle holds NOW: call it after mdpp_set_state_continuous.
extern "C" int mdpp_set_line_history(mdpp_env *h, const float *hist) {
    if (!h || h->cfg.kind != MDPP_KIND_CONTINUOUS || !hist) return MDPP_EINVAL;
    if (!h->cargs.line_L) return fail(h, MDPP_EUNSUPPORTED, "set_line_history: reward_function is not move_along_a_line");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    const size_t N = (size_t)h->cfg.num_envs, L = (size_t)h->cargs.line_L, NL = (size_t)h->cargs.line_NL, nr = (size_t)h->cfg.n_rel;
    std::vector<float> lh(L * NL * N, 0.0f);
    std::vector<uint32_t> me(2 * N);
    HIPCHK(h, hipMemcpy(me.data(), h->d_meta, N * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; i++) {
        const size_t steps = me[2 * i];
        for (size_t j = 0; j < L; j++) {
            const size_t slot = (steps + 1 + j) % L;
            for (size_t c = 0; c < nr; c++) {
                const float v = hist[(i * L + j) * nr + c];
                lh[(slot * NL + c) * N + i] = (v == v) ? v : 0.0f;      // (slots before the episode are never read: steps < L gates the fit)
            }
        }
    }
    HIPCHK(h, hipMemcpy(h->d_line_hist, lh.data(), lh.size() * 4, hipMemcpyHostToDevice));
    h->line_hist_stale = false;
    return MDPP_OK;
}

__global__ void k_philox_normals(uint64_t seed, int64_t env0, uint64_t tick, uint32_t stream, int n_envs, int n_per_env,
                                 double *out) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_envs) return;
    mdpp::Philox g;
    g.init(seed, (uint64_t)(env0 + e), tick, stream);
    for (int j = 0; j < n_per_env; j++) out[e * n_per_env + j] = g.normal();
}

extern "C" int mdpp_philox_normals(uint64_t seed, int64_t env_id0, uint64_t tick, uint32_t stream, int32_t n_envs,
                                   int32_t n_per_env, double *out_dev, void *hip_stream) {
    if (!out_dev || n_envs < 1 || n_per_env < 1) return MDPP_EINVAL;
    hipLaunchKernelGGL(k_philox_normals, dim3((n_envs + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)hip_stream, seed,
                       env_id0, tick, stream, n_envs, n_per_env, out_dev);
    return hipGetLastError() == hipSuccess ? MDPP_OK : MDPP_EHIP;
}

extern "C" int mdpp_status(mdpp_env *h, uint32_t *flags) {
    if (!h || !flags) return MDPP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    const size_t N = (size_t)h->cfg.num_envs;
    HIPCHK(h, hipMemcpy(flags, h->d_status, N * 4, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemset(h->d_status, 0, N * 4));
    return MDPP_OK;
}

extern "C" int mdpp_timer_begin(mdpp_env *h, void *stream) {
    if (!h) return MDPP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventRecord(h->ev0, (hipStream_t)stream));
    return MDPP_OK;
}

extern "C" int mdpp_timer_end(mdpp_env *h, void *stream, float *ms) {
    if (!h || !ms) return MDPP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventRecord(h->ev1, (hipStream_t)stream));
    HIPCHK(h, hipEventSynchronize(h->ev1));
    HIPCHK(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return MDPP_OK;
}

// ---- what the memory system gives plain streaming kernels (bench.py: the ceiling `roofline.frac` is priced beside) ----
// 16 bytes per lane: the float4 copy MI355X_MICROARCH.md measures at 6.29 TB/s, a fill and a read of the same shape.
// ONE-SHOT grids (round 5): a workgroup takes one contiguous 16 KiB tile -- four 16-byte pieces per lane, 4 KiB apart -- and
// ends; the tiles are dispatched in address order, so the chip's write front stays compact.  (Rounds 3-4 ran the same tiles
// on a persistent grid of 8 workgroups per CU: 25 % slower for fills, tools/bench_store.hip, profiles/r04_store_pattern.txt
// -- the product's rollout kernel then BEAT its own "measured ceiling")  NT: non-temporal stores.
namespace mdpp {
typedef unsigned int pu4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void probe_st(pu4 *p, const pu4 &v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
template <bool NT>
__global__ __launch_bounds__(256) void k_probe_copy(const pu4 *__restrict__ s, pu4 *__restrict__ d, size_t n) {
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    if (i + 768 < n) {
        const pu4 a = s[i], b = s[i + 256], c = s[i + 512], e = s[i + 768];
        probe_st<NT>(d + i, a); probe_st<NT>(d + i + 256, b); probe_st<NT>(d + i + 512, c); probe_st<NT>(d + i + 768, e);
    } else {
        for (size_t j = i; j < n; j += 256) probe_st<NT>(d + j, s[j]);
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void k_probe_fill(pu4 *__restrict__ d, size_t n) {
    const pu4 v = pu4{threadIdx.x, blockIdx.x, 3u, 4u};
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    if (i + 768 < n) {
        probe_st<NT>(d + i, v); probe_st<NT>(d + i + 256, v); probe_st<NT>(d + i + 512, v); probe_st<NT>(d + i + 768, v);
    } else {
        for (size_t j = i; j < n; j += 256) probe_st<NT>(d + j, v);
    }
}
__global__ __launch_bounds__(256) void k_probe_read(const pu4 *__restrict__ s, size_t n, uint32_t *out) {
    uint32_t acc = 0;
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    if (i + 768 < n) {
        const pu4 a = s[i], b = s[i + 256], c = s[i + 512], e = s[i + 768];
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ e.x ^ e.y ^ e.z ^ e.w;
    } else {
        for (size_t j = i; j < n; j += 256) { const pu4 v = s[j]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345u) out[0] = acc;          // (keeps the loads alive)
}
__global__ void k_probe_empty(uint32_t *p) { if (p && threadIdx.x == 0xFFFFu) p[0] = 1u; }
// The renderer's store shape without anything else (mode 5 / 6): a wave writes ONE picture of `pic` bytes (84 x 84 = 7 056:
// 16-byte aligned, not cache-line aligned) front to back, 1 KiB per store instruction; four pictures per workgroup, one-shot.
// mode 6: the WORKGROUP writes its four pictures as one region, 4 KiB per round of its four waves, starting at a 128-byte
// boundary (the bytes before it belong to the previous workgroup's last round).
template <int COOP>
__global__ __launch_bounds__(256) void k_probe_pictures(pu4 *__restrict__ d, size_t nbytes, uint32_t pic) {
    const pu4 v = pu4{threadIdx.x, blockIdx.x, 5u, 6u};
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (COOP == 3) {                                    // (mode 8: eight pictures per workgroup -- 441 whole lines --, two per wave)
        for (uint32_t r = 0; r < 2; r++) {
            const size_t base = ((size_t)blockIdx.x * 8 + r * 4 + wave) * pic;
            if (base + pic > nbytes) return;
            for (uint32_t off = lane * 16u; off < pic; off += 1024u) d[(base + off) >> 4] = v;
        }
    } else if (!COOP) {
        const size_t base = ((size_t)blockIdx.x * 4 + wave) * pic;
        if (base + pic > nbytes) return;
        for (uint32_t off = lane * 16u; off < pic; off += 1024u) d[(base + off) >> 4] = v;
    } else {
        const size_t lo = (size_t)blockIdx.x * 4 * pic, hi = lo + 4ull * pic;
        if (hi > nbytes) return;
        size_t a0 = blockIdx.x == 0 ? 0 : ((lo + 127) & ~(size_t)127), a1 = (hi + 127) & ~(size_t)127;   // [a0, a1): this workgroup's lines
        if (COOP == 2) { a0 = lo; a1 = hi; }            // (mode 7: the workgroup's own bytes only, no line alignment)
        for (size_t off = a0 + (size_t)threadIdx.x * 16u; off < a1 && off + 16 <= nbytes; off += 4096u) d[off >> 4] = v;
    }
}
} // namespace mdpp

extern "C" int mdpp_probe_hbm(int mode, void *dst_dev, const void *src_dev, size_t nbytes, int reps, void *stream, float *ms_out) {
    // mode: 0 copy, 1 fill, 2 read (dst_dev: one word of scratch), 3 copy with non-temporal stores, 4 fill with non-temporal stores,
    // 5 / 6: the picture renderer's store shape, per wave / per workgroup (tools only)
    if (!ms_out || reps < 1 || nbytes < 16 || mode < 0 || mode > 8) return MDPP_EINVAL;
    if ((mode != 2 && !dst_dev) || (mode != 1 && mode != 4 && mode < 5 && !src_dev)) return MDPP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const size_t n = nbytes / 16;
    const dim3 grid((unsigned)((n + 1023) / 1024)), block(256);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return MDPP_EHIP;
    auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(mdpp::k_probe_copy<false>, grid, block, 0, s, (const mdpp::pu4 *)src_dev, (mdpp::pu4 *)dst_dev, n);
        else if (mode == 3) hipLaunchKernelGGL(mdpp::k_probe_copy<true>, grid, block, 0, s, (const mdpp::pu4 *)src_dev, (mdpp::pu4 *)dst_dev, n);
        else if (mode == 1) hipLaunchKernelGGL(mdpp::k_probe_fill<false>, grid, block, 0, s, (mdpp::pu4 *)dst_dev, n);
        else if (mode == 4) hipLaunchKernelGGL(mdpp::k_probe_fill<true>, grid, block, 0, s, (mdpp::pu4 *)dst_dev, n);
        else if (mode == 5) hipLaunchKernelGGL(mdpp::k_probe_pictures<0>, dim3((unsigned)(nbytes / (4 * 7056))), block, 0, s, (mdpp::pu4 *)dst_dev, nbytes, 7056u);
        else if (mode == 6) hipLaunchKernelGGL(mdpp::k_probe_pictures<1>, dim3((unsigned)(nbytes / (4 * 7056))), block, 0, s, (mdpp::pu4 *)dst_dev, nbytes, 7056u);
        else if (mode == 8) hipLaunchKernelGGL(mdpp::k_probe_pictures<3>, dim3((unsigned)(nbytes / (8 * 7056))), block, 0, s, (mdpp::pu4 *)dst_dev, nbytes, 7056u);
        else if (mode == 7) hipLaunchKernelGGL(mdpp::k_probe_pictures<2>, dim3((unsigned)(nbytes / (4 * 7056))), block, 0, s, (mdpp::pu4 *)dst_dev, nbytes, 7056u);
        else hipLaunchKernelGGL(mdpp::k_probe_read, grid, block, 0, s, (const mdpp::pu4 *)src_dev, n, (uint32_t *)dst_dev);
    };
    for (int w = 0; w < 2; w++) launch();
    int rc = MDPP_OK;
    if (hipEventRecord(e0, s) != hipSuccess) rc = MDPP_EHIP;
    for (int r = 0; r < reps; r++) launch();
    if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
        hipEventElapsedTime(ms_out, e0, e1) != hipSuccess || hipGetLastError() != hipSuccess) rc = MDPP_EHIP;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return rc;
}

// The device's own floor for ONE launch per step (bench.py `single_step.launch_floor`): n launches of an empty kernel
// (`workgroups` x 64 threads) back to back on `stream` -- the host's time per hipLaunchKernel call and the device's time per
// launch (HIP events around all of them; when the host is the slower side the two agree: eager launches are host-bound).
extern "C" int mdpp_probe_launch(int n, int workgroups, void *stream, float *host_us_out, float *device_us_out) {
    if (n < 1 || workgroups < 1 || !host_us_out || !device_us_out) return MDPP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return MDPP_EHIP;
    for (int w = 0; w < 20; w++) hipLaunchKernelGGL(mdpp::k_probe_empty, dim3(workgroups), dim3(64), 0, s, (uint32_t *)nullptr);
    int rc = MDPP_OK;
    if (hipStreamSynchronize(s) != hipSuccess || hipEventRecord(e0, s) != hipSuccess) rc = MDPP_EHIP;
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int r = 0; r < n; r++) hipLaunchKernelGGL(mdpp::k_probe_empty, dim3(workgroups), dim3(64), 0, s, (uint32_t *)nullptr);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    float ms = 0.0f;
    if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
        hipEventElapsedTime(&ms, e0, e1) != hipSuccess || hipGetLastError() != hipSuccess) rc = MDPP_EHIP;
    *host_us_out = (float)(((double)(t1.tv_sec - t0.tv_sec) * 1e9 + (double)(t1.tv_nsec - t0.tv_nsec)) * 1e-3 / n);
    *device_us_out = ms * 1e3f / (float)n;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return rc;
}

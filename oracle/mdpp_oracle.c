/* ORACLE — TEST INFRASTRUCTURE ONLY (see mdpp_oracle.h for scope, sources, pinning).
 * Scalar restatement, one env instance at a time, written for clarity not speed. */
#include "mdpp_oracle.h"
#include "np_random.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORA_MAX_L 16
#define ORA_MAX_DELAY 64
#define ORA_MAX_DIM 64
#define ORA_MAX_ORDER 8

/* ======================================================================
 * Discrete
 * ==================================================================== */
struct ora_discrete {
    int S, A, L, delay, every_n;
    int has_p_noise, has_r_noise;
    double p_noise, r_noise, scale, shift, term_reward;
    int32_t *P;
    double *rtable;
    uint8_t *is_term;
    double *init_cdf;
    /* per-instance state */
    int hist[ORA_MAX_L + 1]; /* last L+1 states of augmented_state; -1 = NaN slot */
    double ring[ORA_MAX_DELAY];
    int steps;               /* total_transitions_episode */
    np_pcg64 env_rng;        /* self._np_random */
    np_pcg64 space_rng;      /* self.observation_spaces[0].np_random */
    int philox; uint64_t ph_seed, ph_env; uint64_t tick, reset_tick;
    int ph_explicit;         /* Philox mode: the next reset() is a call of its own (stream 3), not the end of a step */
    /* per-episode noise statistics (:1620, :1984-1985; logged at reset :2231-2247, cleared :2360-2369):
     * [0] total_abs_noise_in_reward_episode, [1] total_reward_episode, [2] total_noisy_transitions_episode; st_last: the
     * episode the latest reset() ended, [3] its total_transitions_episode */
    double st[3], st_last[4];
    /* irrelevant sub-space (irrelevant_features=True), rl_toy_env.py:2028-2035, :2063-2092 */
    int irr, S1, A1, irr_state;
    int32_t *P1;
    double *init_cdf1;
    np_pcg64 space1_rng;     /* self.observation_spaces[1].np_random */
    /* use_custom_mdp with a reward MATRIX: R(s, a) of the transition (:1259-1267, :1817-1818) */
    double *rmat;            /* [S*A] or NULL */
};

static long ipow(long b, int e) { long r = 1; while (e-- > 0) r *= b; return r; }

ora_discrete *ora_d_create(int S, int A, int L, int delay, int every_n,
                           int has_p_noise, double p_noise, int has_r_noise, double r_noise,
                           double scale, double shift, double term_reward,
                           const int32_t *P, const double *rtable, const uint8_t *is_term,
                           const double *init_dist) {
    if (L < 1 || L > ORA_MAX_L || delay < 0 || delay > ORA_MAX_DELAY) return NULL;
    ora_discrete *e = (ora_discrete *)calloc(1, sizeof(*e));
    e->S = S; e->A = A; e->L = L; e->delay = delay; e->every_n = every_n;
    e->has_p_noise = has_p_noise; e->p_noise = p_noise;
    e->has_r_noise = has_r_noise; e->r_noise = r_noise;
    e->scale = scale; e->shift = shift; e->term_reward = term_reward;
    long nk = ipow(S, L);
    e->P = (int32_t *)malloc(sizeof(int32_t) * S * A);
    memcpy(e->P, P, sizeof(int32_t) * S * A);
    e->rtable = (double *)malloc(sizeof(double) * nk);
    memcpy(e->rtable, rtable, sizeof(double) * nk);
    e->is_term = (uint8_t *)malloc(S);
    memcpy(e->is_term, is_term, S);
    e->init_cdf = (double *)malloc(sizeof(double) * S);
    np_build_cdf(init_dist, S, e->init_cdf);
    return e;
}

void ora_d_destroy(ora_discrete *e) {
    if (!e) return;
    free(e->P); free(e->rtable); free(e->is_term); free(e->init_cdf); free(e->P1); free(e->init_cdf1);
    free(e->rmat); free(e);
}

/* use_custom_mdp=True with "reward_function" given as an S x A array: the reference wraps it as
 * lambda s, a: reward_matrix[s[-2], a] (:1259-1267) and reward_function() calls that instead of
 * the rewardable-sequence lookup, without the NaN gate (:1817-1818); delay FIFO, every-n, noise and
 * the affine map then apply as for every env (:1968-1990). */
void ora_d_set_reward_matrix(ora_discrete *e, const double *R) {
    free(e->rmat);
    e->rmat = (double *)malloc(sizeof(double) * e->S * e->A);
    memcpy(e->rmat, R, sizeof(double) * e->S * e->A);
}

void ora_d_set_irrelevant(ora_discrete *e, int S1, int A1, const int32_t *P1, const double *init_dist1) {
    e->irr = 1; e->S1 = S1; e->A1 = A1;
    e->P1 = (int32_t *)malloc(sizeof(int32_t) * S1 * A1);
    memcpy(e->P1, P1, sizeof(int32_t) * S1 * A1);
    e->init_cdf1 = (double *)malloc(sizeof(double) * S1);
    np_build_cdf(init_dist1, S1, e->init_cdf1);
}
void ora_d_set_rng_irr(ora_discrete *e, const uint64_t w[6]) { np_pcg64_load(&e->space1_rng, w); }
void ora_d_get_rng_irr(const ora_discrete *e, uint64_t w[6]) { np_pcg64_store(&e->space1_rng, w); }

void ora_d_set_rng(ora_discrete *e, const uint64_t a[6], const uint64_t b[6]) {
    np_pcg64_load(&e->env_rng, a); np_pcg64_load(&e->space_rng, b);
}
void ora_d_get_rng(const ora_discrete *e, uint64_t a[6], uint64_t b[6]) {
    np_pcg64_store(&e->env_rng, a); np_pcg64_store(&e->space_rng, b);
}

/* reset(): rl_toy_env.py:2250 (ring cleared), :2255-2257 (choice from rho_0 on the
 * env RNG), :2275-2278 (NaN-filled history), :2358-2369 (counters). */
/* Philox streams (the build's own, not the reference's): a reset that ends a step of a rollout takes the start-state
 * word of that step's tick (stream 9; the irrelevant sub-space stream 10); an explicit reset() draws 53-bit uniforms
 * from stream 3 keyed by the reset count. */
#define ORA_PHILOX_START 9
#define ORA_PHILOX_START_IRR 10
#define ORA_PHILOX_PNOISE 12     /* transition noise, one word per tick (irrelevant sub-space: id 4) */
#define ORA_PHILOX_RNOISE 13     /* reward noise, one float32 normal per tick */
void ora_d_get_stats(const ora_discrete *e, double cur[3], double last[4]) {
    for (int k = 0; k < 3; k++) cur[k] = e->st[k];
    for (int k = 0; k < 4; k++) last[k] = e->st_last[k];
}
int64_t ora_d_reset(ora_discrete *e) {
    for (int k = 0; k < 3; k++) { e->st_last[k] = e->st[k]; e->st[k] = 0.0; }
    e->st_last[3] = (double)e->steps;
    for (int i = 0; i < e->delay; i++) e->ring[i] = 0.0;
    int s0;
    if (e->philox && !e->ph_explicit)
        s0 = np_philox_start_state(e->ph_seed, e->ph_env, e->tick - 1, ORA_PHILOX_START, e->init_cdf, e->S);
    else
        s0 = np_choice_cdf(&e->env_rng, e->init_cdf, e->S);
    for (int i = 0; i < e->L; i++) e->hist[i] = -1;
    e->hist[e->L] = s0;
    e->steps = 0;
    e->ph_explicit = 0;
    return s0;
}

void ora_d_set_philox(ora_discrete *e, uint64_t seed, uint64_t env_id, uint64_t tick, uint64_t reset_tick) {
    e->philox = 1; e->ph_seed = seed; e->ph_env = env_id; e->tick = tick; e->reset_tick = reset_tick;
}
void ora_d_philox_explicit_reset(ora_discrete *e) {
    np_philox_init(&e->env_rng, e->ph_seed, e->ph_env, e->reset_tick, 3);
    e->reset_tick += 1;
    e->ph_explicit = 1;
}

void ora_d_step(ora_discrete *e, int action, int64_t *obs, double *reward, uint8_t *done) {
    const int S = e->S, L = e->L;
    const uint64_t t0 = e->tick;              /* Philox streams: this step's tick */
    if (e->philox) e->tick += 1;
    /* D1: table lookup, :1603 */
    const int prev = e->hist[L];
    int nxt = e->P[prev * e->A + action];
    /* D2: categorical P-noise on the state-space RNG, :1604-1622 + discrete_extended.py:11-23 */
    if (e->has_p_noise) {
        int noisy;
        if (e->philox) {     /* the build's own streams: one word of the tick, no cdf (np_random.c np_philox_pnoise_state) */
            noisy = np_philox_pnoise_state(np_philox_tick_word(e->ph_seed, e->ph_env, t0, ORA_PHILOX_PNOISE), e->p_noise, S, nxt);
        } else {             /* numpy streams: the state space's generator, as the reference */
            double cdf_s[256], probs_s[256];     /* (S > 255, round 6: on the heap) */
            double *cdf = S <= 256 ? cdf_s : (double *)malloc(sizeof(double) * (size_t)S);
            double *probs = S <= 256 ? probs_s : (double *)malloc(sizeof(double) * (size_t)S);
            for (int i = 0; i < S; i++) probs[i] = 1.0 * e->p_noise / (double)(S - 1);
            probs[nxt] = 1 - e->p_noise;
            np_build_cdf(probs, S, cdf);
            noisy = np_choice_cdf(&e->space_rng, cdf, S);
            if (S > 256) { free(cdf); free(probs); }
        }
        if (noisy != nxt) e->st[2] += 1.0;              /* :1620 */
        nxt = noisy;
    }
    /* D3: history shift, :2050-2052; :2058 */
    for (int i = 0; i < L; i++) e->hist[i] = e->hist[i + 1];
    e->hist[L] = nxt;
    e->steps += 1;
    /* D4: rewardable-sequence lookup, :1821-1845 */
    double r = 0.0;
    if (e->rmat) {
        r = e->rmat[prev * e->A + action];   /* s[-2] is the state the transition started from */
    } else if (e->hist[0] >= 0) {
        long key = 0;
        for (int i = 1; i <= L; i++) key = key * S + e->hist[i];
        r = e->rtable[key];
    }
    /* D5: delay FIFO, :1970-1973 */
    if (e->delay > 0) {
        double out = e->ring[0];
        memmove(e->ring, e->ring + 1, sizeof(double) * (e->delay - 1));
        e->ring[e->delay - 1] = r;
        r = out;
    }
    /* D6: every-n mask, noise, affine, :1975-1990 */
    if (e->steps % e->every_n != 0) r = 0.0;
    e->st[1] += r;                                       /* :1985 */
    if (e->has_r_noise) {
        const double zn = e->philox ? (double)np_philox_tick_normal(e->ph_seed, e->ph_env, t0, ORA_PHILOX_RNOISE)
                                    : np_standard_normal(&e->env_rng);
        const double nz = 0.0 + e->r_noise * zn;
        e->st[0] += fabs(nz);                            /* :1984 */
        r += nz;
    }
    r *= e->scale;
    r += e->shift;
    /* D7: :2102-2109 */
    uint8_t d = e->is_term[nxt];
    if (d) r += e->term_reward * e->scale;
    *obs = nxt; *reward = r; *done = d;
}

/* Tuple spaces: reset() draws the irrelevant start state right after the relevant one, from the
 * same env generator (:2259-2264); step() moves the irrelevant part with its own table and its own
 * P-noise generator after the reward was computed (:2063-2082).  obs = (relevant, irrelevant). */
void ora_d_reset2(ora_discrete *e, int64_t out[2]) {
    const int in_step = e->philox && !e->ph_explicit;
    out[0] = ora_d_reset(e);
    if (in_step)
        e->irr_state = np_philox_start_state(e->ph_seed, e->ph_env, e->tick - 1, ORA_PHILOX_START_IRR, e->init_cdf1, e->S1);
    else
        e->irr_state = np_choice_cdf(&e->env_rng, e->init_cdf1, e->S1);
    e->ph_explicit = 0;
    out[1] = e->irr_state;
}

void ora_d_step2(ora_discrete *e, int action, int action_irr, int64_t obs[2], double *reward, uint8_t *done) {
    ora_d_step(e, action, &obs[0], reward, done);
    int nxt = e->P1[e->irr_state * e->A1 + action_irr];
    if (e->has_p_noise && e->philox) {
        nxt = np_philox_pnoise_state(np_philox_tick_word(e->ph_seed, e->ph_env, e->tick - 1, 4), e->p_noise, e->S1, nxt);
    } else if (e->has_p_noise) {
        double cdf[256], probs[256];
        for (int i = 0; i < e->S1; i++) probs[i] = 1.0 * e->p_noise / (double)(e->S1 - 1);
        probs[nxt] = 1 - e->p_noise;
        np_build_cdf(probs, e->S1, cdf);
        nxt = np_choice_cdf(&e->space1_rng, cdf, e->S1);
    }
    e->irr_state = nxt;
    obs[1] = nxt;
}

void ora_d_rollout2(ora_discrete *e, int T, const int32_t *actions /* [T][2] */, const uint8_t *reset_after,
                    int64_t *obs /* [T][2] */, double *reward, uint8_t *done, int64_t *reset_obs /* [T][2] */) {
    for (int t = 0; t < T; t++) {
        ora_d_step2(e, actions[2 * t], actions[2 * t + 1], &obs[2 * t], &reward[t], &done[t]);
        int rs = reset_after ? reset_after[t] : done[t];
        int64_t ro[2] = {0, 0};
        if (rs) ora_d_reset2(e, ro);
        if (reset_obs) { reset_obs[2 * t] = ro[0]; reset_obs[2 * t + 1] = ro[1]; }
    }
}

void ora_d_rollout(ora_discrete *e, int T, const int32_t *actions, const uint8_t *reset_after,
                   int64_t *obs, double *reward, uint8_t *done, int64_t *reset_obs) {
    for (int t = 0; t < T; t++) {
        ora_d_step(e, actions[t], &obs[t], &reward[t], &done[t]);
        int rs = reset_after ? reset_after[t] : done[t];
        int64_t ro = 0;
        if (rs) ro = ora_d_reset(e);
        if (reset_obs) reset_obs[t] = ro;
    }
}

/* ======================================================================
 * Grid, move_to_a_point (rl_toy_env.py:1727-1778 transition, :1947-1965 reward,
 * :2325-2345 reset; spaces/grid_action_space.py:13-39)
 * ==================================================================== */
struct ora_grid {
    int G;                   /* state dimensions: 2, or 4 with irrelevant_features (:604-608) */
    int shape[4], target[2];
    int make_denser, has_p_noise, has_r_noise, every_n;
    double p_noise, r_noise, scale, shift, term_reward;
    int state[4], steps, reached;
    double st[3], st_last[4]; /* per-episode noise statistics, as in ora_discrete ([2]: :1746) */
    np_pcg64 env_rng;        /* self._np_random: noise trigger (:1736), reward noise */
    np_pcg64 space_rng;      /* self.feature_space.np_random: reset() sample (:2326) */
    np_pcg64 action_rng;     /* self.action_space.np_random: the noisy action (:1738) */
    int philox; uint64_t ph_seed, ph_env; uint64_t tick, reset_tick;
};

ora_grid *ora_g_create(int G, const int32_t *shape, const int32_t *target, int make_denser,
                       int has_p_noise, double p_noise, int has_r_noise, double r_noise,
                       int every_n, double scale, double shift, double term_reward) {
    if (G != 2 && G != 4) return NULL;
    ora_grid *e = (ora_grid *)calloc(1, sizeof(*e));
    e->G = G;
    for (int i = 0; i < G; i++) e->shape[i] = shape[i];
    e->target[0] = target[0]; e->target[1] = target[1];
    e->make_denser = make_denser; e->has_p_noise = has_p_noise; e->p_noise = p_noise;
    e->has_r_noise = has_r_noise; e->r_noise = r_noise; e->every_n = every_n;
    e->scale = scale; e->shift = shift; e->term_reward = term_reward;
    return e;
}
void ora_g_destroy(ora_grid *e) { free(e); }
void ora_g_set_rng(ora_grid *e, const uint64_t env[6], const uint64_t space[6], const uint64_t action[6]) {
    np_pcg64_load(&e->env_rng, env); np_pcg64_load(&e->space_rng, space); np_pcg64_load(&e->action_rng, action);
}
void ora_g_get_rng(const ora_grid *e, uint64_t env[6], uint64_t space[6], uint64_t action[6]) {
    np_pcg64_store(&e->env_rng, env); np_pcg64_store(&e->space_rng, space); np_pcg64_store(&e->action_rng, action);
}
void ora_g_set_philox(ora_grid *e, uint64_t seed, uint64_t env_id, uint64_t tick, uint64_t reset_tick) {
    e->philox = 1; e->ph_seed = seed; e->ph_env = env_id; e->tick = tick; e->reset_tick = reset_tick;
}
void ora_g_philox_explicit_reset(ora_grid *e) {
    np_philox_init(&e->space_rng, e->ph_seed, e->ph_env, e->reset_tick, 3);
    e->reset_tick += 1;
}

/* reset(): feature_space.sample() of a Box(0, grid_shape, int64) (:780-788, :2326): per dimension
 * floor(uniform(0, g + 1)) -- so the cell index g, one past the grid, can come out, as in the
 * reference; the terminal-state resampling loop never triggers (Box(int64).contains(float64 array)
 * is False under gymnasium's dtype check, :973-982). */
void ora_g_reset(ora_grid *e, int64_t *obs) {
    for (int k = 0; k < 3; k++) { e->st_last[k] = e->st[k]; e->st[k] = 0.0; }
    e->st_last[3] = (double)e->steps;
    for (int i = 0; i < e->G; i++) {
        double v = 0.0 + ((double)(e->shape[i] + 1) - 0.0) * np_random(&e->space_rng);
        e->state[i] = (int)floor(v);
        obs[i] = e->state[i];
    }
    e->steps = 0; e->reached = 0;
}
void ora_g_get_stats(const ora_grid *e, double cur[3], double last[4]) {
    for (int k = 0; k < 3; k++) cur[k] = e->st[k];
    for (int k = 0; k < 4; k++) last[k] = e->st_last[k];
}

void ora_g_step(ora_grid *e, const int32_t *action, int64_t *obs, double *reward, uint8_t *done) {
    const int G = e->G;
    if (e->philox) {
        np_philox_init(&e->env_rng, e->ph_seed, e->ph_env, e->tick, 0);
        np_philox_init(&e->space_rng, e->ph_seed, e->ph_env, e->tick, 1);
        np_philox_init(&e->action_rng, e->ph_seed, e->ph_env, e->tick, 5);
        e->tick += 1;
    }
    int a[4], old0 = e->state[0], old1 = e->state[1];
    /* GridActionSpace.contains: every entry in {-1, 0, 1} and at most one non-zero */
    int ok = 1, nz = 0;
    for (int i = 0; i < G; i++) { a[i] = action[i]; if (a[i] < -1 || a[i] > 1) ok = 0; nz += a[i] != 0; }
    if (nz > 1) ok = 0;
    if (ok) {
        if (e->has_p_noise) {
            if (np_random(&e->env_rng) < e->p_noise) {                /* :1736 uniform() */
                for (;;) {                                               /* :1737-1749 */
                    int ind = (int)np_integers(&e->action_rng, 0, G);
                    int val = (int)np_integers(&e->action_rng, 0, 3);
                    int na[4] = {0, 0, 0, 0}, same = 1;
                    na[ind] = val - 1;
                    for (int i = 0; i < G; i++) if (na[i] != a[i]) same = 0;
                    if (!same) { for (int i = 0; i < G; i++) a[i] = na[i]; e->st[2] += 1.0; break; }   /* :1746 */
                }
            }
        }
        for (int i = 0; i < G; i++) {                                    /* :1751-1761 */
            int n = e->state[i] + a[i];
            if (n < 0) n = 0;
            if (n >= e->shape[i]) n = e->shape[i] - 1;
            e->state[i] = n;
        }
    }                                                                    /* else: noop, :1763-1768 */
    if (e->state[0] == e->target[0] && e->state[1] == e->target[1]) e->reached = 1;   /* :1770-1776 */
    e->steps += 1;
    double r = 0.0;
    if (e->make_denser) {                                                /* :1949-1960 */
        int d_old = abs(old0 - e->target[0]) + abs(old1 - e->target[1]);
        int d_new = abs(e->state[0] - e->target[0]) + abs(e->state[1] - e->target[1]);
        r += (double)(d_old - d_new);
    } else if (e->state[0] == e->target[0] && e->state[1] == e->target[1]) r += 1.0;  /* :1962-1965 */
    if (e->steps % e->every_n != 0) r = 0.0;                             /* :1975-1978 */
    e->st[1] += r;                                                       /* :1985 */
    if (e->has_r_noise) {
        const double nz = 0.0 + e->r_noise * np_standard_normal(&e->env_rng);
        e->st[0] += fabs(nz);                                            /* :1984 */
        r += nz;
    }
    r *= e->scale;
    r += e->shift;
    uint8_t d = (uint8_t)e->reached;                                     /* :2102-2104 */
    if (d) r += e->term_reward * e->scale;
    for (int i = 0; i < G; i++) obs[i] = e->state[i];
    *reward = r; *done = d;
}

void ora_g_rollout(ora_grid *e, int T, const int32_t *actions, const uint8_t *reset_after,
                   int64_t *obs, double *reward, uint8_t *done, int64_t *reset_obs) {
    const int G = e->G;
    for (int t = 0; t < T; t++) {
        ora_g_step(e, actions + (size_t)t * G, obs + (size_t)t * G, &reward[t], &done[t]);
        int rs = reset_after ? reset_after[t] : done[t];
        int64_t ro[4] = {0, 0, 0, 0};
        if (rs) ora_g_reset(e, ro);
        if (reset_obs) for (int i = 0; i < G; i++) reset_obs[(size_t)t * G + i] = ro[i];
    }
}

/* ======================================================================
 * Continuous, move_to_a_point
 * ==================================================================== */
typedef struct { double v; int is32; } rew_t; /* np.float32 vs Python float */

struct ora_continuous {
    int D, n_rel, order, make_denser, has_p_noise, has_r_noise, delay, every_n, n_boxes;
    int image_quirk;
    int target64;            /* the DEFAULT target_point: float64 zeros of length state_space_dim (:652-654) */
    double radius;
    int rel[ORA_MAX_DIM];
    float inertia32, amax32, smax32, radius32, alw32;
    float tpow32[ORA_MAX_ORDER + 1];  /* float32(time_unit ** k) */
    double fact[ORA_MAX_ORDER + 1];   /* k! as float64 */
    double smax, amax;
    float target[ORA_MAX_DIM];
    double p_noise, r_noise, scale, shift, term_reward;
    float *box_lo, *box_hi;
    /* state */
    float sd[ORA_MAX_ORDER + 1][ORA_MAX_DIM]; /* state_derivatives */
    float cur[ORA_MAX_DIM];                   /* curr_state == augmented_state[-1] */
    rew_t ring[ORA_MAX_DELAY];
    int steps, reached;
    /* per-episode noise statistics (:1686, :1984-1985; logged at reset :2231-2247, cleared :2360-2369): [0]
     * total_abs_noise_in_reward_episode, [1] total_reward_episode (np.float32 running sum where the reward is np.float32:
     * int 0 + float32 -> float32, float32 + Python 0.0 -> float32; float64 for the line reward / default target), [2] unused,
     * [3 + d] total_abs_noise_in_transition_episode[d]; st_last: the episode the latest reset() ended, [3 + D] its transitions */
    double st[3 + ORA_MAX_DIM], st_last[4 + ORA_MAX_DIM];
    np_pcg64 env_rng;    /* self._np_random */
    np_pcg64 space_rng;  /* self.feature_space.np_random */
    int philox; uint64_t ph_seed, ph_env; uint64_t tick, reset_tick;
    /* reward_function == "move_along_a_line" (:1864-1910) */
    int line_L;                               /* sequence_length; 0 = move_to_a_point */
    ora_line_fit_fn line_fit;
    float lhist[ORA_MAX_LINE][ORA_MAX_DIM];   /* the last line_L states, oldest first */
};

ora_continuous *ora_c_create(int D, int n_rel, const int32_t *rel_idx, int order,
                             double inertia, double time_unit, double state_max,
                             double action_max, const float *target, double target_radius,
                             int make_denser, double action_loss_weight,
                             int has_p_noise, double p_noise, int has_r_noise, double r_noise,
                             int delay, int every_n, double scale, double shift,
                             double term_reward, int n_boxes, const float *box_lo,
                             const float *box_hi) {
    if (D > ORA_MAX_DIM || order > ORA_MAX_ORDER || order < 1 || delay > ORA_MAX_DELAY) return NULL;
    ora_continuous *e = (ora_continuous *)calloc(1, sizeof(*e));
    e->D = D; e->n_rel = n_rel; e->order = order; e->make_denser = make_denser;
    for (int i = 0; i < n_rel; i++) { e->rel[i] = rel_idx[i]; e->target[i] = target[i]; }
    e->inertia32 = (float)inertia;
    e->amax = action_max; e->smax = state_max;
    e->amax32 = (float)action_max; e->smax32 = (float)state_max;
    e->radius32 = (float)target_radius; e->alw32 = (float)action_loss_weight;
    e->radius = target_radius;
    double f = 1.0;
    for (int k = 1; k <= order; k++) {
        f *= (double)k;
        e->fact[k] = f;
        e->tpow32[k] = (float)pow(time_unit, (double)k); /* Python float ** int */
    }
    e->has_p_noise = has_p_noise; e->p_noise = p_noise;
    e->has_r_noise = has_r_noise; e->r_noise = r_noise;
    e->delay = delay; e->every_n = every_n;
    e->scale = scale; e->shift = shift; e->term_reward = term_reward;
    e->n_boxes = n_boxes;
    if (n_boxes > 0) {
        size_t nb = sizeof(float) * n_boxes * n_rel;
        e->box_lo = (float *)malloc(nb); memcpy(e->box_lo, box_lo, nb);
        e->box_hi = (float *)malloc(nb); memcpy(e->box_hi, box_hi, nb);
    }
    return e;
}

void ora_c_set_image_quirk(ora_continuous *e, int on) { e->image_quirk = on; }
/* No "target_point" in the config: the reference falls back to np.zeros(shape=(state_space_dim,)) -- FLOAT64, and
 * of the full state dimension, so it only broadcasts when every dimension is relevant (:652-654).  The float32 state
 * minus that target is a float64 vector: np.linalg.norm, the target latch (:1719-1725) and a dense reward (:1926-1929)
 * are float64 (np.float64 behaves like a Python float in what follows); a sparse reward is the Python float 1.0 / 0.0
 * and turns np.float32 at `reward -= action_loss_weight * norm(action)` as with an explicit target. */
void ora_c_set_target64(ora_continuous *e) { e->target64 = 1; }

void ora_c_destroy(ora_continuous *e) {
    if (!e) return;
    free(e->box_lo); free(e->box_hi); free(e);
}
void ora_c_set_rng(ora_continuous *e, const uint64_t a[6], const uint64_t b[6]) {
    np_pcg64_load(&e->env_rng, a); np_pcg64_load(&e->space_rng, b);
}
void ora_c_get_rng(const ora_continuous *e, uint64_t a[6], uint64_t b[6]) {
    np_pcg64_store(&e->env_rng, a); np_pcg64_store(&e->space_rng, b);
}
void ora_c_get_derivs(const ora_continuous *e, float *out) {
    for (int k = 0; k <= e->order; k++)
        for (int i = 0; i < e->D; i++) out[k * e->D + i] = e->sd[k][i];
}

/* is_terminal_state for hypercubes, :945-952 (Box.contains on the relevant dims) */
static int in_term_box(const ora_continuous *e, const float *s) {
    for (int b = 0; b < e->n_boxes; b++) {
        int in = 1;
        for (int j = 0; j < e->n_rel; j++) {
            float x = s[e->rel[j]];
            if (!(x >= e->box_lo[b * e->n_rel + j] && x <= e->box_hi[b * e->n_rel + j])) { in = 0; break; }
        }
        if (in) return 1;
    }
    return 0;
}

/* np.linalg.norm of a contiguous float32 vector = sqrt(x.dot(x)) in float32.
 * numpy's FLOAT_dot hands the vector to cblas_sdot (scipy-openblas 0.3.29,
 * third-party, not under /root/reference); for n < 32 OpenBLAS's x86_64 sdot
 * runs its scalar tail: float32 products accumulated sequentially in a double,
 * rounded to float32 once at the end (checked against numpy for n = 1..31 in
 * tests/test_np_random.py::test_float32_norm_semantics). */
static float norm32_rel_minus_target(const ora_continuous *e, const float *s) {
    double acc = 0.0;
    for (int j = 0; j < e->n_rel; j++) {
        float d = s[e->rel[j]] - e->target[j];
        float p = d * d;
        acc += (double)p;
    }
    return sqrtf((float)acc);
}
/* np.linalg.norm of (float32 state - float64 zeros) = sqrt(x.dot(x)) in float64: cblas_ddot; for n < 16 OpenBLAS's
 * x86_64 ddot runs its scalar tail, a sequential sum (every product of two float32-valued doubles is exact, so an
 * fma chain and multiply-then-add give the same bits; checked against numpy for n <= 12, mismatches from n = 16 on:
 * the host builder refuses the default target for state_space_dim >= 16). */
static double norm64_rel(const ora_continuous *e, const float *s) {
    double acc = 0.0;
    for (int j = 0; j < e->n_rel; j++) {
        double d = (double)s[e->rel[j]] - 0.0;
        acc += d * d;
    }
    return sqrt(acc);
}
static float norm32(const float *x, int n) {
    double acc = 0.0;
    for (int j = 0; j < n; j++) { float p = x[j] * x[j]; acc += (double)p; }
    return sqrtf((float)acc);
}

/* reset(): :2250 ring, :2284-2307 rejection sampling from feature_space
 * (gymnasium Box.sample: uniform(low, high) per bounded dim, normal() when
 * unbounded), :2311-2323 derivatives / history, :2358-2369 counters. */
/* move_along_a_line (:1864-1910): the last sequence_length states (relevant dimensions, float32)
 * are fitted with a line through their mean along the first right-singular vector of the centred
 * data (np.linalg.svd on float32: LAPACK sgesdd -- third-party arithmetic, obtained through the
 * callback from the same numpy the reference uses, together with the float32 mean); the reward is
 * minus the mean float64 distance of the points from that line (dist_of_pt_from_line, :2546-2576). */
void ora_c_set_line_reward(ora_continuous *e, int L, ora_line_fit_fn fit) {
    e->line_L = (L >= 1 && L <= ORA_MAX_LINE) ? L : 0;
    e->line_fit = fit;
}

static void line_push(ora_continuous *e, const float *s) {
    for (int k = 0; k + 1 < e->line_L; k++) memcpy(e->lhist[k], e->lhist[k + 1], sizeof(float) * e->D);
    memcpy(e->lhist[e->line_L - 1], s, sizeof(float) * e->D);
}

/* numpy's scalar `x ** 2` is libm pow(x, 2.0), which is not always the correctly rounded x * x;
 * called through a volatile pointer so that the compiler does not fold it into a multiplication */
static double (*volatile libm_pow)(double, double) = pow;

static double line_reward(ora_continuous *e) {
    const int L = e->line_L, n = e->n_rel, D = e->D;
    float pts[ORA_MAX_LINE * ORA_MAX_DIM] = {0};
    double mean[ORA_MAX_DIM], v0[ORA_MAX_DIM], ptA[ORA_MAX_DIM], ptB[ORA_MAX_DIM];
    for (int k = 0; k < L; k++) memcpy(pts + (size_t)k * D, e->lhist[k], sizeof(float) * D);
    e->line_fit(pts, L, D, mean, v0);        /* float32 values, widened */
    for (int j = 0; j < n; j++) {             /* vv[0] * linspace(-1, 1, 2)[:, None] + data_mean, float64 */
        ptA[j] = v0[j] * -1.0 + mean[j];
        ptB[j] = v0[j] * 1.0 + mean[j];
    }
    double total = 0.0;
    for (int k = 0; k < L; k++) {
        double ab[ORA_MAX_DIM], ap[ORA_MAX_DIM], dot = 0.0, nab = 0.0, nap = 0.0;
        for (int j = 0; j < n; j++) {
            ab[j] = ptA[j] - ptB[j];
            ap[j] = ptA[j] - (double)pts[(size_t)k * D + e->rel[j]];
        }
        /* np.dot / np.linalg.norm on short float64 vectors: OpenBLAS ddot's scalar tail, a chain of
         * fused multiply-adds (checked against numpy for n <= 8 in tests/test_np_random.py) */
        for (int j = 0; j < n; j++) { dot = fma(ab[j], ap[j], dot); nab = fma(ab[j], ab[j], nab); nap = fma(ap[j], ap[j], nap); }
        nab = sqrt(nab);
        double dist = 0.0;
        if (!(nab < 1e-13)) {
            const double proj = dot / nab;
            double sq = libm_pow(sqrt(nap), 2.0) - libm_pow(proj, 2.0);
            if (sq < 0) sq = 0;
            dist = sqrt(sq);
        }
        total += dist;
    }
    return 0.0 + -total / (double)L;
}

void ora_c_get_stats(const ora_continuous *e, double *cur /* [3 + D] */, double *last /* [4 + D] */) {
    for (int k = 0; k < 3 + e->D; k++) { cur[k] = e->st[k]; last[k] = e->st_last[k]; }
    last[3 + e->D] = e->st_last[3 + e->D];
}
void ora_c_reset(ora_continuous *e, float *obs) {
    for (int k = 0; k < 3 + e->D; k++) { e->st_last[k] = e->st[k]; e->st[k] = 0.0; }
    e->st_last[3 + e->D] = (double)e->steps;
    for (int i = 0; i < e->delay; i++) { e->ring[i].v = 0.0; e->ring[i].is32 = 0; }
    const int bounded = isfinite(e->smax);
    const double lo = (double)(-e->smax32), range = (double)e->smax32 - (double)(-e->smax32);
    for (;;) {
        for (int i = 0; i < e->D; i++) {
            double v = bounded ? lo + range * np_random(&e->space_rng)
                               : 0.0 + 1.0 * np_standard_normal(&e->space_rng);
            e->cur[i] = (float)v;
        }
        if (!in_term_box(e, e->cur)) break;
    }
    for (int k = 0; k <= e->order; k++)
        for (int i = 0; i < e->D; i++) e->sd[k][i] = 0.0f;
    for (int i = 0; i < e->D; i++) { e->sd[0][i] = e->cur[i]; obs[i] = e->cur[i]; }
    e->steps = 0; e->reached = 0;
    if (e->line_L) line_push(e, e->cur);       /* augmented_state = [NaN]*(len-1) + [curr_state], :2313-2323 */
}

void ora_c_set_philox(ora_continuous *e, uint64_t seed, uint64_t env_id, uint64_t tick, uint64_t reset_tick) {
    e->philox = 1; e->ph_seed = seed; e->ph_env = env_id; e->tick = tick; e->reset_tick = reset_tick;
}
void ora_c_philox_explicit_reset(ora_continuous *e) {
    np_philox_init(&e->space_rng, e->ph_seed, e->ph_env, e->reset_tick, 3);
    e->reset_tick += 1;
}

void ora_c_step(ora_continuous *e, const float *a, float *obs, double *reward,
                int *reward_is32, uint8_t *done) {
    const int D = e->D, n = e->order;
    if (e->philox) {
        np_philox_init(&e->env_rng, e->ph_seed, e->ph_env, e->tick, 0);
        np_philox_init(&e->space_rng, e->ph_seed, e->ph_env, e->tick, 1);
        e->tick += 1;
    }
    float nxt[ORA_MAX_DIM];
    /* C1: Box.contains(action), :1640 */
    int ok = 1;
    for (int i = 0; i < D; i++) if (!(a[i] >= -e->amax32 && a[i] <= e->amax32)) ok = 0;
    if (ok) {
        /* C2: :1654-1669.  float32 * weak Python float stays float32; dividing by the
         * np.float64 factorial promotes to float64; += rounds back to float32. */
        for (int i = 0; i < D; i++) e->sd[n][i] = a[i] / e->inertia32;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n - i; j++)
                for (int c = 0; c < D; c++) {
                    float prod = e->sd[i + j + 1][c] * e->tpow32[j + 1];
                    double term = (double)prod / e->fact[j + 1];
                    e->sd[i][c] = (float)((double)e->sd[i][c] + term);
                }
        for (int i = 0; i < D; i++) nxt[i] = e->sd[0][i];
    } else {
        for (int i = 0; i < D; i++) nxt[i] = e->cur[i]; /* "stay", :1671-1672 */
    }
    /* C3: :1682-1691 (D normals from the env RNG, or float64 zeros) */
    for (int i = 0; i < D; i++) {
        double nz = e->has_p_noise ? 0.0 + e->p_noise * np_standard_normal(&e->env_rng) : 0.0;
        e->st[3 + i] += fabs(nz);                        /* :1686 */
        nxt[i] = (float)((double)nxt[i] + nz);
    }
    /* C4: :1694-1717 */
    int inside = 1;
    for (int i = 0; i < D; i++) if (!(nxt[i] >= -e->smax32 && nxt[i] <= e->smax32)) inside = 0;
    /* image_representations=True: observation_space is the ImageContinuous space, whose contains()
     * returns None for a state vector (spaces/image_continuous.py:292-302), so the reference takes
     * the out-of-bounds branch on EVERY step: clip (a no-op inside the box) and zero all derivatives */
    if (e->image_quirk) inside = 0;
    if (!inside) {
        for (int i = 0; i < D; i++) {
            float x = nxt[i]; /* np.clip = minimum(maximum(x, lo), hi); NaN propagates */
            if (x < -e->smax32) x = -e->smax32;
            if (x > e->smax32) x = e->smax32;
            nxt[i] = x;
        }
        for (int k = 0; k <= n; k++)
            for (int i = 0; i < D; i++) e->sd[k][i] = 0.0f;
        for (int i = 0; i < D; i++) e->sd[0][i] = nxt[i];
    }
    /* C5: :1719-1725 (the target latch exists for move_to_a_point only) */
    float dist_new = e->line_L ? 0.0f : norm32_rel_minus_target(e, nxt);
    double dist_new64 = e->target64 ? norm64_rel(e, nxt) : 0.0;
    const int within = e->target64 ? (dist_new64 < e->radius) : (dist_new < e->radius32);
    if (!e->line_L && within) e->reached = 1;
    e->steps += 1; /* :2058 */
    /* C6: :1912-1945 */
    rew_t r;
    if (e->line_L) {
        /* :1864-1910; gate :1856: the state delay+sequence_length transitions back must exist */
        line_push(e, nxt);
        r.v = (e->steps >= e->line_L) ? line_reward(e) : 0.0;
        r.is32 = 0;
    } else if (e->make_denser && e->target64) {
        r.v = -dist_new64;                       /* np.float64 (:1926) */
        r.v = r.v + norm64_rel(e, e->cur);       /* :1929 */
    } else if (e->make_denser) {
        float dist_old = norm32_rel_minus_target(e, e->cur);
        r.v = (double)(float)(-dist_new + dist_old);
    } else {
        r.v = within ? 1.0 : 0.0;
    }
    if (!e->line_L) {
        float pen = e->alw32 * norm32(a, D);     /* Python float * np.float32 -> np.float32 */
        if (e->make_denser && e->target64) { r.v = r.v - (double)pen; r.is32 = 0; }   /* np.float64 - np.float32 */
        else { r.v = (double)((float)r.v - pen); r.is32 = 1; }
    }
    /* C7: :1968-1990 */
    if (e->delay > 0) {
        rew_t out = e->ring[0];
        memmove(e->ring, e->ring + 1, sizeof(rew_t) * (e->delay - 1));
        e->ring[e->delay - 1] = r;
        r = out;
    }
    if (e->steps % e->every_n != 0) { r.v = 0.0; r.is32 = 0; }
    /* :1985 */
    if (e->line_L || (e->target64 && e->make_denser)) e->st[1] += r.v;
    else if (r.is32) e->st[1] = (double)((float)e->st[1] + (float)r.v);
    if (e->has_r_noise) {
        double nz = 0.0 + e->r_noise * np_standard_normal(&e->env_rng);
        e->st[0] += fabs(nz);                            /* :1984 */
        if (r.is32) r.v = (double)((float)r.v + (float)nz); else r.v = r.v + nz;
    }
    if (r.is32) {
        r.v = (double)((float)r.v * (float)e->scale);
        r.v = (double)((float)r.v + (float)e->shift);
    } else {
        r.v = r.v * e->scale;
        r.v = r.v + e->shift;
    }
    /* C8 + :2102-2109 */
    uint8_t d = (uint8_t)(in_term_box(e, nxt) || e->reached);
    if (d) {
        double add = e->term_reward * e->scale;
        if (r.is32) r.v = (double)((float)r.v + (float)add); else r.v = r.v + add;
    }
    for (int i = 0; i < D; i++) { e->cur[i] = nxt[i]; obs[i] = nxt[i]; }
    *reward = r.v; *reward_is32 = r.is32; *done = d;
}

void ora_c_rollout(ora_continuous *e, int T, const float *actions, const uint8_t *reset_after,
                   float *obs, double *reward, uint8_t *done, float *reset_obs) {
    const int D = e->D;
    for (int t = 0; t < T; t++) {
        int is32;
        ora_c_step(e, actions + (size_t)t * D, obs + (size_t)t * D, &reward[t], &is32, &done[t]);
        int rs = reset_after ? reset_after[t] : done[t];
        if (reset_obs) memset(reset_obs + (size_t)t * D, 0, sizeof(float) * D);
        if (rs) {
            float tmp[ORA_MAX_DIM];
            ora_c_reset(e, tmp);
            if (reset_obs) memcpy(reset_obs + (size_t)t * D, tmp, sizeof(float) * D);
        }
    }
}

/* ======================================================================
 * Image observations
 * ==================================================================== */
static long floordiv(long a, long b) { long q = a / b; if ((a % b != 0) && ((a < 0) != (b < 0))) q--; return q; }

static void i_draw_with(const ora_image_cfg *c, np_pcg64 *gp, int *R, int *cx, int *cy, int *angle, int *flip);
void ora_i_draw(const ora_image_cfg *c, uint64_t rngw[6], int *R, int *cx, int *cy,
                int *angle, int *flip) {
    np_pcg64 g; np_pcg64_load(&g, rngw);
    i_draw_with(c, &g, R, cx, cy, angle, flip);
    np_pcg64_store(&g, rngw);
}
/* Philox streams (the build's own): the transforms of the `n` images drawn at one tick -- the step's observation, one
 * image per sub-space, then, where the step ended the episode, the images of reset()'s observation -- come in that order
 * from stream (seed, env, tick, stream id): id 2 for steps, id 11 for an explicit reset() (keyed by the reset count).
 * out: n rows of {R, cx, cy, angle, flip}. */
void ora_i_draw_philox(const ora_image_cfg *c, uint64_t seed, uint64_t env, uint64_t tick, uint32_t stream, int n, int *out) {
    np_pcg64 g;
    np_philox_init(&g, seed, env, tick, stream);
    for (int k = 0; k < n; k++) i_draw_with(c, &g, &out[5 * k], &out[5 * k + 1], &out[5 * k + 2], &out[5 * k + 3], &out[5 * k + 4]);
}
static void i_draw_with(const ora_image_cfg *c, np_pcg64 *gp, int *R, int *cx, int *cy, int *angle, int *flip) {
    np_pcg64 g = *gp;
    int r = c->R0;
    *cx = (int)(c->W / 2.0); *cy = (int)(c->H / 2.0);
    if (c->has_scale) { /* :149-169 */
        double ls = c->log_min_R + np_random(&g) * (c->log_max_R - c->log_min_R);
        r = (int)exp(ls);
    }
    if (c->has_shift) { /* :172-181: integers(-max_shift + 1, max_shift), Python // quantisation */
        double mw = c->W / 2.0 - r, mh = c->H / 2.0 - r;
        long aw = np_integers(&g, (int64_t)(-mw + 1), (int64_t)mw);
        long ah = np_integers(&g, (int64_t)(-mh + 1), (int64_t)mh);
        aw = floordiv(aw, c->sh_quant) * c->sh_quant;
        ah = floordiv(ah, c->sh_quant) * c->sh_quant;
        *cx += (int)aw; *cy += (int)ah;
    }
    *angle = 0;
    if (c->has_rotate) { /* :247-254 */
        long rot = np_integers(&g, 0, 360);
        *angle = (int)(floordiv(rot, c->ro_quant) * c->ro_quant);
    }
    *flip = 0;
    if (c->has_flip) { /* :257-262 */
        if (np_integers(&g, 0, 2) == 0) *flip = (np_integers(&g, 0, 2) == 0) ? 1 : 2;
    }
    *R = r;
    *gp = g;
}

static double round15(double v) { /* Python round(v, 15) for |v| <= 1 */
    /* round-half-even on the decimal representation; use the correctly rounded
     * decimal conversion the C library provides. */
    char buf[64];
    snprintf(buf, sizeof buf, "%.15f", v);
    return strtod(buf, NULL);
}

void ora_i_rotate_flip_transpose(int W, int H, const uint8_t *src, int angle, int flip, uint8_t *obs) {
    uint8_t *rot = (uint8_t *)calloc((size_t)W * H, 1);
    angle = ((angle % 360) + 360) % 360;
    if (angle == 0) {
        memcpy(rot, src, (size_t)W * H);
    } else if (angle == 180) {
        for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) rot[y * W + x] = src[(H - 1 - y) * W + (W - 1 - x)];
    } else if ((angle == 90 || angle == 270) && W == H) {
        for (int y = 0; y < H; y++) for (int x = 0; x < W; x++)
            rot[y * W + x] = (angle == 90) ? src[x * W + (W - 1 - y)] : src[(H - 1 - x) * W + y];
    } else {
        const double PI = 3.141592653589793;
        double a = -((double)angle * (PI / 180.0)); /* -math.radians(angle) */
        double m0 = round15(cos(a)), m1 = round15(sin(a)), m3 = round15(-sin(a)), m4 = round15(cos(a));
        double cxr = W / 2.0, cyr = H / 2.0;
        double m2 = (m0 * (-cxr) + m1 * (-cyr) + 0.0) + cxr;
        double m5 = (m3 * (-cxr) + m4 * (-cyr) + 0.0) + cyr;
#define ORA_FIX(v) ((int)floor((v) * 65536.0 + 0.5))
        int a0 = ORA_FIX(m0), a1 = ORA_FIX(m1), a3 = ORA_FIX(m3), a4 = ORA_FIX(m4);
        int a2 = ORA_FIX(m2 + m0 * 0.5 + m1 * 0.5), a5 = ORA_FIX(m5 + m3 * 0.5 + m4 * 0.5);
        for (int y = 0; y < H; y++) {
            int xx = a2, yy = a5;
            for (int x = 0; x < W; x++) {
                int xin = xx >> 16;
                if (xin >= 0 && xin < W) {
                    int yin = yy >> 16;
                    if (yin >= 0 && yin < H) rot[y * W + x] = src[yin * W + xin];
                }
                xx += a0; yy += a3;
            }
            a2 += a1; a5 += a4;
        }
    }
    /* flip (:257-262), then the transpose of :264-266: obs[x][y] = img[y][x] */
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
        int sx = (flip == 1) ? W - 1 - x : x, sy = (flip == 2) ? H - 1 - y : y;
        obs[x * H + y] = rot[sy * W + sx];
    }
    free(rot);
}

/* ======================================================================
 * ImageContinuous.get_image_representation (spaces/image_continuous.py:116-277) for a continuous
 * env: per 2-D sub-space one W x H RGB picture -- background (208,208,208); the terminal
 * hypercubes as black rectangles (corner pixels from convert_to_pixel, inclusive, :176-188); the
 * target as a green disc and the agent as a blue disc, both Pillow ellipses with the integer
 * bounding box centre +- R (:190-207), i.e. a fixed (2R+1)^2 raster `disc` (made by the caller
 * with Pillow, the third-party rasteriser the reference calls) at an integer position; the
 * irrelevant sub-space's picture shows only its agent disc; pictures are concatenated along the
 * first axis after the transpose (:209-212, :239-250).  out: uint8 [n_sub * W][H][3].
 * convert_to_pixel (:252-277): ((v - min) / (max - min)) in float32, times the image size in
 * float64, truncated toward zero. */
static void ic_pixel(float vx, float vy, float smax, int W, int H, int *px, int *py) {
    const float lo = -smax, hi = smax;
    const float fx = (vx - lo) / (hi - lo), fy = (vy - lo) / (hi - lo);
    *px = (int)((double)fx * (double)W);
    *py = (int)((double)fy * (double)H);
}

void ora_ic_render(int W, int H, int R, const uint8_t *disc /* [(2R+1)*(2R+1)], [dy][dx] */,
                   int D, const float *state, float smax, const float *target /* [2] */,
                   int n_boxes, const float *box_lo, const float *box_hi /* [n_boxes*2] */,
                   uint8_t *out) {
    const int n_sub = D > 2 ? 2 : 1, T = 2 * R + 1;
    for (int sub = 0; sub < n_sub; sub++) {
        uint8_t *img = out + (size_t)sub * W * H * 3;      /* [x][y][c] */
        for (size_t k = 0; k < (size_t)W * H * 3; k++) img[k] = 208;
        if (sub == 0) {
            for (int b = 0; b < n_boxes; b++) {
                int x0, y0, x1, y1;
                ic_pixel(box_lo[2 * b], box_lo[2 * b + 1], smax, W, H, &x0, &y0);
                ic_pixel(box_hi[2 * b], box_hi[2 * b + 1], smax, W, H, &x1, &y1);
                for (int x = x0 < 0 ? 0 : x0; x <= x1 && x < W; x++)
                    for (int y = y0 < 0 ? 0 : y0; y <= y1 && y < H; y++)
                        img[((size_t)x * H + y) * 3] = img[((size_t)x * H + y) * 3 + 1] = img[((size_t)x * H + y) * 3 + 2] = 0;
            }
        }
        for (int pass = (sub == 0 ? 0 : 1); pass < 2; pass++) {    /* 0: target (green), 1: agent (blue) */
            int cx, cy;
            if (pass == 0) ic_pixel(target[0], target[1], smax, W, H, &cx, &cy);
            else ic_pixel(state[2 * sub], state[2 * sub + 1], smax, W, H, &cx, &cy);
            for (int dy = 0; dy < T; dy++)
                for (int dx = 0; dx < T; dx++) {
                    if (!disc[dy * T + dx]) continue;
                    const int x = cx - R + dx, y = cy - R + dy;
                    if (x < 0 || x >= W || y < 0 || y >= H) continue;
                    uint8_t *p = img + ((size_t)x * H + y) * 3;
                    p[0] = 0; p[1] = pass == 0 ? 255 : 0; p[2] = pass == 0 ? 0 : 255;
                }
        }
    }
}

/* ImageContinuous of a GRID env (draw_grid, spaces/image_continuous.py:139-207): white grid lines
 * (the caller's `lines` mask, drawn with Pillow exactly as :145-165 does), terminal cells as black
 * rectangles from convert_to_pixel(cell) to convert_to_pixel(cell + 1) inclusive, the target and the
 * agent as discs at convert_to_pixel(cell + 0.5); convert_to_pixel here is float64:
 * int((v - 0) / (g - 0) * size).  out: uint8 [n_sub * W][H][3], n_sub = G / 2. */
static int ig_px(double v, int g, int size) { return (int)((v - 0.0) / (double)(g - 0) * (double)size); }

void ora_ig_render(int W, int H, int R, const uint8_t *disc, int G, const int32_t *shape, const int32_t *cells,
                   const int32_t *target, int n_term, const int32_t *term_cells /* [n_term*2] */,
                   const uint8_t *lines /* [n_sub*W*H], [x][y] */, uint8_t *out) {
    const int n_sub = G / 2, T = 2 * R + 1;
    for (int sub = 0; sub < n_sub; sub++) {
        uint8_t *img = out + (size_t)sub * W * H * 3;
        const uint8_t *ln = lines + (size_t)sub * W * H;
        for (size_t k = 0; k < (size_t)W * H; k++) img[3 * k] = img[3 * k + 1] = img[3 * k + 2] = ln[k] ? 255 : 208;
        if (sub == 0) {
            for (int b = 0; b < n_term; b++) {
                const int x0 = ig_px(term_cells[2 * b], shape[0], W), y0 = ig_px(term_cells[2 * b + 1], shape[1], H);
                const int x1 = ig_px(term_cells[2 * b] + 1.0, shape[0], W), y1 = ig_px(term_cells[2 * b + 1] + 1.0, shape[1], H);
                for (int x = x0 < 0 ? 0 : x0; x <= x1 && x < W; x++)
                    for (int y = y0 < 0 ? 0 : y0; y <= y1 && y < H; y++)
                        img[((size_t)x * H + y) * 3] = img[((size_t)x * H + y) * 3 + 1] = img[((size_t)x * H + y) * 3 + 2] = 0;
            }
        }
        for (int pass = (sub == 0 ? 0 : 1); pass < 2; pass++) {
            const double vx = pass == 0 ? target[0] + 0.5 : cells[2 * sub] + 0.5;
            const double vy = pass == 0 ? target[1] + 0.5 : cells[2 * sub + 1] + 0.5;
            /* both sub-spaces are scaled with the RELEVANT grid's size (feature_space.high[relevant_indices]) */
            const int cx = ig_px(vx, shape[0], W), cy = ig_px(vy, shape[1], H);
            for (int dy = 0; dy < T; dy++)
                for (int dx = 0; dx < T; dx++) {
                    if (!disc[dy * T + dx]) continue;
                    const int x = cx - R + dx, y = cy - R + dy;
                    if (x < 0 || x >= W || y < 0 || y >= H) continue;
                    uint8_t *p = img + ((size_t)x * H + y) * 3;
                    p[0] = 0; p[1] = pass == 0 ? 255 : 0; p[2] = pass == 0 ? 0 : 255;
                }
        }
    }
}

/* ======================================================================
 * GymEnvWrapper-style post-processor, gym_env_wrapper.py:350-439, :441-486, :523-618
 * ==================================================================== */
struct ora_post {
    int continuous, n_actions, obs_dim, obs_f64, delay;
    int has_p, has_r;
    double p_noise, r_noise, scale, shift, term;
    double *noise_cdf;            /* [n][n]: row a = cdf of the categorical around action a (:356-364) */
    int image, H, W, C, pad, has_shift, sh_quant;
    double *ring; int ring_n;     /* self.reward_buffer (a list: append at the end, pop the front) */
    np_pcg64 rng;
    int philox; uint64_t ph_seed, ph_env, tick, reset_tick, action_tick;
};

/* numpy's pairwise summation of a contiguous float64 vector (loops_utils.h.src, n <= 128) == np.sum */
double ora_np_pairwise_sum(const double *a, int n) {
    if (n < 8) { double res = 0.; for (int i = 0; i < n; i++) res += a[i]; return res; }
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

ora_post *ora_p_create(int continuous, int n_actions, int obs_dim, int obs_f64, int delay,
                       int has_p_noise, double p_noise, int has_r_noise, double r_noise,
                       double scale, double shift, double term_state_reward,
                       int image, int H, int W, int C, int pad, int has_shift, int sh_quant) {
    ora_post *e = (ora_post *)calloc(1, sizeof *e);
    e->continuous = continuous; e->n_actions = n_actions; e->obs_dim = obs_dim; e->obs_f64 = obs_f64;
    e->delay = delay; e->has_p = has_p_noise; e->p_noise = p_noise; e->has_r = has_r_noise; e->r_noise = r_noise;
    e->scale = scale; e->shift = shift; e->term = term_state_reward;
    e->image = image; e->H = H; e->W = W; e->C = C; e->pad = pad; e->has_shift = has_shift; e->sh_quant = sh_quant;
    e->ring = (double *)calloc((size_t)(delay > 0 ? delay : 1), sizeof(double));
    e->ring_n = delay;
    if (!continuous && has_p_noise && p_noise != 0.0) {        /* `if self.transition_noise:` (:355) */
        const int n = n_actions;
        e->noise_cdf = (double *)malloc(sizeof(double) * (size_t)n * n);
        double *p = (double *)malloc(sizeof(double) * (size_t)n);
        for (int a = 0; a < n; a++) {                            /* :356-361 */
            for (int j = 0; j < n; j++) p[j] = 1.0 * p_noise / (double)(n - 1);
            p[a] = 1 - p_noise;
            np_build_cdf(p, n, e->noise_cdf + (size_t)a * n);
        }
        free(p);
    }
    return e;
}
void ora_p_destroy(ora_post *e) { if (e) { free(e->ring); free(e->noise_cdf); free(e); } }
void ora_p_set_rng(ora_post *e, const uint64_t w[6]) { np_pcg64_load(&e->rng, w); }
void ora_p_get_rng(const ora_post *e, uint64_t w[6]) { np_pcg64_store(&e->rng, w); }
void ora_p_set_philox(ora_post *e, uint64_t seed, uint64_t env_id, uint64_t tick, uint64_t reset_tick) {
    e->philox = 1; e->ph_seed = seed; e->ph_env = env_id; e->tick = tick; e->reset_tick = reset_tick;
    e->action_tick = tick;
}

/* get_transformed_image, :523-618 (only "shift" does anything upstream; square RGB images) */
static void post_image(ora_post *e, const uint8_t *in, uint8_t *out) {
    const int H = e->H, W = e->W, C = e->C, pad = e->pad;
    const int tot_w = W + 2 * pad, tot_h = H + 2 * pad;
    int shift_w = (int)(tot_w / 2.0), shift_h = (int)(tot_h / 2.0);          /* :560-561 */
    if (e->has_shift) {                                                      /* :584-594 */
        const int R = W;
        const long max_w = (tot_w - R) / 2, max_h = (tot_h - R) / 2;
        long aw = np_integers(&e->rng, -max_w + 1, max_w);
        long ah = np_integers(&e->rng, -max_h + 1, max_h);
        aw = (aw / e->sh_quant) * e->sh_quant;                               /* int(a / q) * q: truncation */
        ah = (ah / e->sh_quant) * e->sh_quant;
        shift_w += (int)aw; shift_h += (int)ah;
    }
    const int top = shift_h - H / 2, left = shift_w - W / 2;                 /* :604-611 */
    memset(out, 0, (size_t)tot_w * tot_h * C);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < C; c++)                                      /* canvas[row][col] then transpose(1, 0, 2), :616 */
                out[((size_t)(left + x) * tot_h + (top + y)) * C + c] = in[((size_t)y * W + x) * C + c];
}

void ora_p_reset(ora_post *e, const uint8_t *img_in, uint8_t *img_out) {
    for (int i = 0; i < e->delay; i++) e->ring[i] = 0.0;                     /* :456 */
    e->ring_n = e->delay;
    if (e->philox) { np_philox_init(&e->rng, e->ph_seed, e->ph_env, e->reset_tick, 8); e->reset_tick += 1; }
    if (e->image) post_image(e, img_in, img_out);                            /* :481-482 */
}

int ora_p_action(ora_post *e, int action) {
    /* (Philox streams: keyed by the number of action calls, so that K calls before one fused K-step call differ) */
    if (e->philox) { np_philox_init(&e->rng, e->ph_seed, e->ph_env, e->action_tick, 6); e->action_tick += 1; }
    if (e->continuous || !e->noise_cdf) return action;
    return np_choice_cdf(&e->rng, e->noise_cdf + (size_t)action * e->n_actions, e->n_actions);   /* :364 */
}

void ora_p_step(ora_post *e, const void *obs_in, double reward, int done, void *obs_out, double *reward_out) {
    if (e->philox) { np_philox_init(&e->rng, e->ph_seed, e->ph_env, e->tick, 7); e->tick += 1; }
    if (e->continuous) {
        /* :367-373 noise drawn before the inner step (it does not touch this generator), added to a copy
         * of the observation in its own dtype :400-402; without noise `next_obs += 0.0` */
        for (int d = 0; d < e->obs_dim; d++) {
            const double nz = e->has_p ? 0.0 + e->p_noise * np_standard_normal(&e->rng) : 0.0;
            if (e->obs_f64) ((double *)obs_out)[d] = ((const double *)obs_in)[d] + nz;
            else ((float *)obs_out)[d] = (float)((double)((const float *)obs_in)[d] + nz);
        }
    }
    if (e->image) post_image(e, (const uint8_t *)obs_in, (uint8_t *)obs_out);   /* :404-405 */
    if (done) {                                                              /* :407-414 */
        double tmp[128];
        for (int i = 0; i < e->ring_n; i++) tmp[i] = e->ring[i] * e->scale + e->shift;
        reward += ora_np_pairwise_sum(tmp, e->ring_n);
        reward += e->term * e->scale;
    } else {                                                                 /* :415-420 */
        if (e->delay > 0) {
            const double out = e->ring[0];
            memmove(e->ring, e->ring + 1, sizeof(double) * (size_t)(e->delay - 1));
            e->ring[e->delay - 1] = reward;
            reward = out;
        }
    }
    const double nz = e->has_r ? 0.0 + e->r_noise * np_standard_normal(&e->rng) : 0.0;   /* :426 */
    reward += nz;                                                            /* :430-432 */
    reward *= e->scale;
    reward += e->shift;
    *reward_out = reward;
}

/* mdpp.h — C ABI of libmdpp_hip.so: batched RLToyEnv.step()/reset() on MI355X (gfx950).
 *
 * One handle = one shard of env instances on one GPU.  Every `*_dev` pointer is a
 * DEVICE pointer owned by the caller (the Python layer passes torch tensors'
 * data_ptr()); every `*_host` pointer is host memory.  All work is enqueued on the
 * caller's HIP stream (`stream` is a hipStream_t passed as void*; NULL = default
 * stream) and is asynchronous with respect to the host (image rollouts additionally fork to a side
 * stream owned by the handle and join back into the caller's stream before they return).  The library owns only its
 * per-env state and tables (freed by mdpp_destroy); nothing is allocated inside
 * mdpp_step / mdpp_step_n / mdpp_reset, so they can be captured into a HIP graph
 * (tests/test_gpu_parity.py::test_rollout_is_graph_capturable).  The handle's step counter
 * travels to the kernels by value: a replayed graph is exact for numpy-stream handles with
 * unit rewards; Philox keys and the key ring of non-unit rewards read the counter.  A handle is not
 * thread-safe.  Every function returns 0 on success or a negative MDPP_E* code and
 * never throws; mdpp_last_error() gives the message.
 *
 * What each entry point replaces in the reference (/root/reference):
 *   mdpp_create + mdpp_upload_*   RLToyEnv.__init__            mdp_playground/envs/rl_toy_env.py:216-853
 *                                 (the tables themselves are generated on the host by
 *                                  mdp_playground_amd/mdp.py, restating :855-1575)
 *   mdpp_seed_streams             RLToyEnv.seed / Space.seed   rl_toy_env.py:2379-2406,
 *                                                              spaces/discrete_extended.py:7-9
 *   mdpp_reset                    RLToyEnv.reset               rl_toy_env.py:2217-2377
 *   mdpp_step                     RLToyEnv.step                rl_toy_env.py:1992-2125
 *                                 (transition_function :1577-1725, reward_function :1782-1990,
 *                                  ImageMultiDiscrete.get_image_representation
 *                                  spaces/image_multi_discrete.py:129-288)
 *                                 (grid: transition :1727-1778, reward :1947-1965,
 *                                  GridActionSpace spaces/grid_action_space.py:13-39)
 *   mdpp_step_n                   a Python loop of K step() calls (e.g. example.py:69-86)
 *   mdpp_get_state/set_state      get/set_augmented_state      rl_toy_env.py:2127-2215
 *                                 (plus the RNG streams, ring and counters the reference omits)
 */
#ifndef MDPP_H
#define MDPP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDPP_ABI_VERSION 8

enum { MDPP_OK = 0, MDPP_EINVAL = -1, MDPP_EHIP = -2, MDPP_ENOMEM = -3, MDPP_ESTATE = -4,
       MDPP_EUNSUPPORTED = -5 };

enum { MDPP_KIND_DISCRETE = 0, MDPP_KIND_CONTINUOUS = 1, MDPP_KIND_GRID = 2 };
enum { MDPP_RNG_NUMPY_PCG64 = 0,   /* per-env numpy Generator(PCG64) streams: reference-exact */
       MDPP_RNG_PHILOX = 1 };      /* counter-based Philox4x32-10 keyed by (seed, global env id, tick, stream id): no reference for
                                      this mode, its draws are defined in mdp_playground_amd/csrc/mdpp_rng.hpp and restated in
                                      oracle/np_random.c.  Stream ids: 0-4 the MDPP_STREAM_* below (whole blocks per tick:
                                      continuous / grid noise, image transforms; 4 also the irrelevant sub-space's transition
                                      noise), 3 an explicit reset(), 5 grid noisy action, 6-8 the post-processor, 9 / 10 the
                                      start state of an in-rollout reset (relevant / irrelevant), 11 an explicit reset()'s image,
                                      12 discrete transition noise, 13 discrete reward noise -- ids 4 (discrete), 9, 10, 12, 13
                                      take ONE word (one float32 normal) per tick from a block that serves four ticks */
enum { MDPP_AUTORESET_DISABLED = 0,    /* the reference's own behaviour: keeps stepping after done */
       MDPP_AUTORESET_SAME_STEP = 1,   /* gymnasium 0.29 SyncVectorEnv: the step that ends an episode returns the next episode's first obs */
       MDPP_AUTORESET_NEXT_STEP = 2 }; /* gymnasium >= 1.0 vector envs: the step() AFTER the one that ended an episode ignores that env's
                                          action, resets it and returns (first obs, reward 0, no flags) */
enum { MDPP_OBS_I64 = 0, MDPP_OBS_I32 = 1, MDPP_OBS_F32 = 2, MDPP_OBS_IMAGE_U8 = 3 };
/* RNG streams, named after the generator object they mirror in the reference */
enum { MDPP_STREAM_ENV = 0,        /* RLToyEnv._np_random: reset draw, reward noise, continuous P-noise */
       MDPP_STREAM_SPACE = 1,      /* discrete: observation_spaces[0] (P-noise); continuous, grid: feature_space (reset) */
       MDPP_STREAM_IMAGE = 2,      /* observation_space (ImageMultiDiscrete transforms) */
       MDPP_STREAM_SPACE_IRR = 3,  /* discrete, irrelevant_features: observation_spaces[1] (its P-noise) */
       MDPP_STREAM_ACTION = 4,     /* grid: action_space (GridActionSpace.sample of a noisy action) */
       MDPP_NUM_STREAMS = 5 };
/* per-env status bits (mdpp_status) */
enum { MDPP_STATUS_BAD_ACTION = 1u,     /* discrete: action out of range (reference: IndexError);
                                           continuous: action rejected by Box.contains -> "stay" (:1671) */
       MDPP_STATUS_RESET_GAVE_UP = 2u,  /* continuous reset(): 4096 draws all fell into terminal hypercubes */
       MDPP_STATUS_INTERNAL = 0x80000000u }; /* a bounded in-kernel wait expired (never expected) */

/* Kernel-selection switches (mdpp_set_options): each bit takes one specialised rollout kernel (or
 * one of its multi-wave forms) out of the dispatch, so that the same handle runs on the more general
 * kernel of the same arithmetic.  Results never depend on them (tests compare both sides); they exist
 * for those tests, for profiling and for ablations.  Per handle; nothing is read from the environment. */
enum { MDPP_OPT_NO_PIPE = 1u << 0,         /* discrete: no three-role k_discrete_rollout_pipe */
       MDPP_OPT_NO_HELPER = 1u << 1,       /* no helper (producer) waves in k_*_rollout_fast */
       MDPP_OPT_NO_PARK = 1u << 2,         /* continuous helper waves draw in lockstep (no parked lanes) */
       MDPP_OPT_NO_CFAST = 1u << 3,        /* continuous: k_continuous_step instead of k_continuous_rollout_fast */
       MDPP_OPT_NO_QUIET = 1u << 4,        /* discrete: k_discrete_step instead of k_discrete_rollout_quiet */
       MDPP_OPT_NO_QUIET_NOISE = 1u << 5,  /* ... only for handles with P- or reward noise */
       MDPP_OPT_NO_DUO = 1u << 6,          /* k_discrete_rollout_quiet: one role only */
       MDPP_OPT_NO_TRIO = 1u << 7,         /* k_discrete_rollout_quiet: at most two roles */
       MDPP_OPT_NO_GFAST = 1u << 8,        /* grid: k_grid_step instead of k_grid_rollout_fast */
       MDPP_OPT_NO_GFAST_NOISE = 1u << 9,  /* ... only for handles with noise */
       MDPP_OPT_NO_IMGFAST = 1u << 10,     /* polygon images: k_image_obs instead of k_image_obs_fast */
       MDPP_OPT_NO_IMG_OVERLAP = 1u << 11, /* image rollouts: no side-stream pipeline of the batches */
       MDPP_OPT_NO_PHILOX_FAST = 1u << 12, /* Philox handles: general kernels only */
       MDPP_OPT_NO_LEAN = 1u << 13,        /* discrete: no k_discrete_rollout_lean (S <= 8 re-encoding of _pipe) */
       MDPP_OPT_NO_IMG_NEARTAB = 1u << 14, /* polygon images: k_image_obs_fast walks the bounding box instead of the near-dword table */
       MDPP_OPT_NO_STEP1 = 1u << 15,       /* mdpp_step (K = 1): the rollout kernels with K = 1 instead of k_discrete_step1 / k_continuous_step1 */
       MDPP_OPT_NO_QUIET_SF = 1u << 17,    /* k_discrete_rollout_quiet: no compile-time form of the sweep defaults (sequence_length 1, same-step autoreset, ...) */
       MDPP_OPT_NO_SIGMA0 = 1u << 16       /* noise keys present with sigma 0 (the reference draws rng.normal(0, 0): rl_toy_env.py:398-403, :1982):
                                              form the normals' values anyway instead of advancing the streams alone */ };

/* what a discrete env's reward table is keyed by */
enum { MDPP_REWARD_SEQUENCES = 0,     /* the last L states (rewardable_sequences, rl_toy_env.py:1837-1841) */
       MDPP_REWARD_STATE_ACTION = 1 }; /* (s, a) of the transition: use_custom_mdp with a reward MATRIX
                                          (:1259-1267, :1817-1818); no NaN gate, needs unit_rewards = 0 */

/* reward_function of a continuous env */
enum { MDPP_CREWARD_MOVE_TO_A_POINT = 0,    /* rl_toy_env.py:1912-1945 */
       MDPP_CREWARD_MOVE_ALONG_A_LINE = 1 }; /* :1864-1910: minus the mean distance of the last L states from
                                                the line fitted through them (first right-singular vector);
                                                n_rel <= 8 (state_space_dim <= 12 beyond 4), L <= 64, no image observations */

typedef struct mdpp_env mdpp_env;

#define MDPP_MAX_DIM 32
#define MDPP_MAX_ORDER 4
#define MDPP_MAX_BOXES 8

typedef struct {
    int32_t abi_version;        /* MDPP_ABI_VERSION */
    int32_t kind;               /* MDPP_KIND_* */
    int32_t num_envs;           /* env instances in this shard */
    int64_t env_id_offset;      /* global id of local env 0 (multi-GPU sharding; Philox keys use the global id) */
    int32_t rng_mode;           /* MDPP_RNG_* */
    int32_t autoreset;          /* MDPP_AUTORESET_* */
    int32_t max_episode_steps;  /* 0 = never truncate (RLToyFiniteHorizon-v0: 100) */
    int32_t obs_dtype;          /* MDPP_OBS_* */
    uint64_t philox_seed;       /* MDPP_RNG_PHILOX only */

    /* reward post-processing shared by both kinds, rl_toy_env.py:1968-1990, :2105-2109 */
    int32_t delay;              /* reward_buffer length */
    int32_t every_n;            /* reward_every_n_steps */
    int32_t has_reward_noise;   /* "reward_noise" in config (a normal is drawn even for std 0) */
    double reward_noise;        /* std */
    double reward_scale, reward_shift, term_state_reward;

    /* ---- discrete ---- */
    int32_t S, A, L;            /* state_space_size (<= 65 535), action_space_size, sequence_length (<= 15; S^L < 4e9) -- S > 255 and L > 7: the general
                                   kernel alone, not together, without image observations */
    int32_t num_tables;         /* 1 = one MDP shared by all envs; num_envs = one MDP per env */
    int32_t unit_rewards;       /* 1: every rewardable sequence pays exactly 1.0 (bitmask table) */
    int32_t reward_kind;        /* MDPP_REWARD_*: what the reward table is keyed by */
    int32_t has_transition_noise;
    double transition_noise;
    /* irrelevant_features=True (rl_toy_env.py:2028-2035, :2063-2092): a second, reward-irrelevant
     * sub-space with its own transition table and P-noise generator; actions and observations
     * become pairs [N][2] = (relevant, irrelevant) */
    int32_t irrelevant;
    int32_t S_irr, A_irr;

    /* ---- continuous ---- */
    int32_t D, n_rel, order;    /* state_space_dim, len(relevant_indices), transition_dynamics_order */
    int32_t reward_function;    /* MDPP_CREWARD_*; move_along_a_line takes sequence_length from L above */
    int32_t rel_idx[MDPP_MAX_DIM];
    int32_t make_denser;
    int32_t has_p_noise;        /* "transition_noise" in config (D normals drawn even for std 0) */
    double p_noise;             /* std */
    double inertia, time_unit, state_space_max, action_space_max;
    double target_radius, action_loss_weight;
    float target[MDPP_MAX_DIM];
    int32_t n_boxes;            /* terminal hypercubes, rl_toy_env.py:908-952: [n_boxes][n_rel] packed, n_boxes * n_rel <= 256 (ABI 8; 8 before) */
    float box_lo[MDPP_MAX_BOXES * MDPP_MAX_DIM];
    float box_hi[MDPP_MAX_BOXES * MDPP_MAX_DIM];

    /* ---- grid (move_to_a_point), rl_toy_env.py:1727-1778, :1947-1965; uses make_denser,
     * has_transition_noise / transition_noise (probability of a re-drawn action) from above ---- */
    int32_t grid_dims;          /* 2, or 4 with irrelevant_features (the grid repeated, :604-608) */
    int32_t grid_shape[4];
    int32_t grid_target[2];

    /* ---- image observations: discrete envs (ImageMultiDiscrete: polygons, random transforms), or
     * continuous envs (ImageContinuous, spaces/image_continuous.py:116-277: RGB pictures, uses image,
     * img_w, img_h and img_r0 = radius of the agent / target discs only) ---- */
    int32_t image;              /* 1: obs is uint8[W][H][1] per env (discrete) or uint8[n_sub W][H][3] (continuous, grid) */
    int32_t img_w, img_h;
    int32_t img_has_scale, img_has_shift, img_has_rotate, img_has_flip;
    int32_t img_sh_quant, img_ro_quant;
    int32_t img_r0;             /* circle_radius */
    int32_t img_r_min, img_r_max; /* radii that can occur (scale transform), templates cover [r_min, r_max] */
    double img_log_min_r, img_log_max_r;
    int32_t img_tpl_size;       /* templates are (2*tpl_half+1)^2 bitmaps */

    /* ---- ABI 6 ---- */
    int32_t episode_stats;      /* 1: keep the reference's per-episode noise statistics per env (mdpp_get_episode_stats); such
                                   handles run on the general kernels */
    int32_t target_f64;         /* continuous, move_to_a_point, no "target_point" in the config: the reference's default,
                                   np.zeros(shape=(state_space_dim,)) -- float64, every dimension relevant
                                   (rl_toy_env.py:652-654): distances, the target latch and a dense reward are float64 */
} mdpp_config;

/* Lifetime */
int mdpp_abi_version(void);
int mdpp_create(const mdpp_config *cfg, int device, mdpp_env **out);
void mdpp_destroy(mdpp_env *h);
const char *mdpp_last_error(const mdpp_env *h);   /* h may be NULL: last create() error */

/* Discrete tables (host pointers; T = cfg.num_tables):
 *   P        uint8 [T][S][A]      transition matrix                       rl_toy_env.py:1050-1151
 *            (cfg.S > 255, up to 65 535 -- round 6: uint16 [T][S][A] behind the same pointer; such handles run the general
 *            kernel alone, without image observations: mdpp_discrete_wide.hip)
 *   rtable   double[T][S^L] (MDPP_REWARD_SEQUENCES) or double[T][S][A] (MDPP_REWARD_STATE_ACTION:
 *            use_custom_mdp with a reward matrix, R(s, a) of the transition s, a -> s', :1259-1267)
 *            or, when cfg.unit_rewards, NULL with
 *   rbits    uint8 [T][ceil(S^L/8)] bit k set <=> sequence with key k is rewardable (:1508)
 *   is_term  uint8 [T][S]                                                 :868-889
 *   init_cdf double[T][S]  cumsum(rho_0)/cumsum(rho_0)[-1]                :1003-1018, :2255
 *   noise_cdf double[S][S] row n = normalised cdf of the P-noise categorical whose mode is n
 *            (:1605-1612); NULL when cfg.has_transition_noise == 0 */
int mdpp_upload_discrete_tables(mdpp_env *h, const uint8_t *P_host, const double *rtable_host,
                                const uint8_t *rbits_host, const uint8_t *is_term_host,
                                const double *init_cdf_host, const double *noise_cdf_host);

/* Irrelevant sub-space tables (host; shared by all envs or one set per env like the others):
 *   P_irr uint8 [T][S_irr][A_irr]   transition_function_irrelevant            rl_toy_env.py:1153-1228
 *   init_cdf_irr double[T][S_irr]   cumsum(irrelevant_init_state_dist) normalised  :1025-1037, :2260
 *   noise_cdf_irr double[S_irr][S_irr]  as noise_cdf, for the irrelevant P-noise  :2068-2076; NULL without noise */
int mdpp_upload_discrete_irrelevant(mdpp_env *h, const uint8_t *P_irr_host, const double *init_cdf_irr_host,
                                    const double *noise_cdf_irr_host);

/* Image templates (host): uint8 [S][n_radii][n_cls][tpl][tpl], polygon rasters centred in the
 * template; cls_x/cls_y int16 [S][n_radii][W or H] map a centre coordinate to its template class.
 * A centre whose polygon is CUT by the picture's edge (a quantised shift (v // q) * q rounds towards -inf and can leave the
 * draw's own range, image_multi_discrete.py:172-181) must be a class of its own whose template is cut the same way: the
 * specialised renderers apply Pillow's "source outside the picture reads 0" rule through the template alone
 * (mdp_playground_amd/image_obs.py _edge_clip; golden i_shq5_rot). */
int mdpp_upload_image_templates(mdpp_env *h, const uint8_t *tpl_host, int32_t n_radii, int32_t n_cls_x,
                                int32_t n_cls_y, const int16_t *cls_x_host, const int16_t *cls_y_host);

/* Continuous image observations: disc uint8 [(2 R + 1)][(2 R + 1)], non-zero = covered, R = cfg.img_r0:
 * the raster of Pillow's ellipse with the integer bounding box centre +- R (image_continuous.py:190-207). */
int mdpp_upload_image_disc(mdpp_env *h, const uint8_t *disc_host);
/* Grid envs with image observations additionally: lines uint8 [n_sub W][H], non-zero = white grid line
 * (Pillow's draw.line from the reference's end points, image_continuous.py:145-165); the terminal
 * cells drawn as rectangles ride in cfg.box_lo[2 b + d] (cell coordinates), cfg.n_boxes of them. */
int mdpp_upload_image_lines(mdpp_env *h, const uint8_t *lines_host);

/* RNG streams.  words_host: uint64 [num_envs][6] = PCG64 {state_lo, state_hi, inc_lo, inc_hi,
 * has_uint32, uinteger} exactly as numpy's bit_generator.state reports them. */
int mdpp_seed_streams(mdpp_env *h, int stream, const uint64_t *words_host);
int mdpp_get_streams(mdpp_env *h, int stream, uint64_t *words_host);

/* reset(): mask_dev == NULL resets every env, else only envs with mask_dev[i] != 0.
 * obs_dev receives the new first observation of the reset envs (others untouched). */
int mdpp_reset(mdpp_env *h, const uint8_t *mask_dev, void *obs_dev, void *stream);

/* step(): actions int32[N] (discrete; int32[N][2] with cfg.irrelevant), float32[N][D]
 * (continuous) or int32[N][grid_dims] (grid); obs per cfg.obs_dtype ([N], [N][2] with
 * cfg.irrelevant, [N][D], [N][grid_dims] or [N][W][H]);
 * reward float32[N]; terminated/truncated uint8[N].
 * final_obs_dev (nullable): with same-step autoreset, the last observation of episodes that ended. */
int mdpp_step(mdpp_env *h, const void *actions_dev, void *obs_dev, float *reward_dev,
              uint8_t *terminated_dev, uint8_t *truncated_dev, void *final_obs_dev, void *stream);

/* K fused steps in one launch, per-env state held in registers between steps.
 * actions [K][N](…), outputs [K][N](…), time-major. */
int mdpp_step_n(mdpp_env *h, int K, const void *actions_dev, void *obs_dev, float *reward_dev,
                uint8_t *terminated_dev, uint8_t *truncated_dev, void *stream);

/* Per-env internal state <-> host (synchronous; checkpoint / set_augmented_state).
 * Discrete: hist int32[N][L+1] (-1 = NaN slot), steps int32[N], ring double[N][delay].
 * Continuous: derivs float[N][order+1][D], cur float[N][D], steps int32[N],
 *             ring double[N][delay] + ring_is32 uint8[N][delay], reached uint8[N]. */
int mdpp_get_state_discrete(mdpp_env *h, int32_t *hist_host, int32_t *steps_host, double *ring_host);
int mdpp_set_state_discrete(mdpp_env *h, const int32_t *hist_host, const int32_t *steps_host,
                            const double *ring_host);
/* cfg.irrelevant: the irrelevant part of curr_state, int32[N] (curr_state[1], :2089) */
int mdpp_get_state_irrelevant(mdpp_env *h, int32_t *irr_host);
int mdpp_set_state_irrelevant(mdpp_env *h, const int32_t *irr_host);
/* Grid: cells int32[N][grid_dims], steps int32[N], reached uint8[N] (the latched target flag, :1776) */
int mdpp_get_state_grid(mdpp_env *h, int32_t *cells_host, int32_t *steps_host, uint8_t *reached_host);
int mdpp_set_state_grid(mdpp_env *h, const int32_t *cells_host, const int32_t *steps_host,
                        const uint8_t *reached_host);
int mdpp_get_state_continuous(mdpp_env *h, float *derivs_host, float *cur_host, int32_t *steps_host,
                              double *ring_host, uint8_t *ring_is32_host, uint8_t *reached_host);
int mdpp_set_state_continuous(mdpp_env *h, const float *derivs_host, const float *cur_host,
                              const int32_t *steps_host, const double *ring_host,
                              const uint8_t *ring_is32_host, const uint8_t *reached_host);
/* reward_function = move_along_a_line (rl_toy_env.py:1864-1910): the window of the line fit -- the relevant coordinates
 * of the last sequence_length states of every env, which the reference keeps in `augmented_state` and returns from
 * get_augmented_state() (:2147-2156) --, oldest first, float32 [N][L][n_rel] on the host; slots older than the running
 * episode are NaN like the reference's NaN-filled list.  mdpp_set_state_continuous on such a handle restores the step
 * counters but not this window: stepping is refused (MDPP_EUNSUPPORTED) until mdpp_set_line_history has been called
 * AFTER it (the slot of a state depends on the step counter).  ABI 7. */
int mdpp_get_line_history(mdpp_env *h, float *hist_host);
int mdpp_set_line_history(mdpp_env *h, const float *hist_host);

/* HIP graphs of single steps (RLToyVectorEnv.step_graph).  mdpp_step hands the handle's step counter to its
 * launch by value (ring head = counter mod delay for delay lines kept in memory; Philox keys), so a captured
 * launch replays with the counter it was captured with.
 * mdpp_graph_replay_exact: 1 when a graph of K captured mdpp_step launches replays exactly for this handle
 * (numpy streams, and either no delay line in memory or K a multiple of the delay), 2 when it does through the
 * device-side offset below, < 0 on error (0, "does not": image observations keyed by the counter until round 4; no handle now).
 * mdpp_tick: adds `advance` (may be negative) to the step counter and returns the new value in *tick_out (may be
 * NULL): the capture advances the counter although nothing ran (take it back with -K), a replay runs K steps the
 * counter has not seen (add K). */
int mdpp_graph_replay_exact(mdpp_env *h, int K);
/* ABI 7 -- graphs that replay exactly for EVERY handle (Philox streams, any delay line; round 5: image observations too -- the
 * kernels that draw a step's image transforms read the same device word):
 * mdpp_graph_replay_exact returns 2 where a by-value capture would not be exact but this protocol is:
 *   mdpp_graph_capture(h, 1);  capture the K mdpp_step launches;  mdpp_graph_capture(h, 0);  mdpp_tick(h, -K, NULL);
 *   per replay:  mdpp_graph_set_tick_offset(h, counter_now - counter_at_capture, stream);  launch the graph on `stream`;
 *                mdpp_tick(h, K, NULL).
 * Launches made in capture mode add the device word that mdpp_graph_set_tick_offset writes to the step counter they
 * were captured with (ring head of a delay line in memory, Philox keys), first thing in the kernel. */
int mdpp_graph_capture(mdpp_env *h, int on);
int mdpp_graph_set_tick_offset(mdpp_env *h, int64_t offset, void *stream);
int mdpp_tick(mdpp_env *h, int64_t advance, uint64_t *tick_out);

/* MDPP_AUTORESET_NEXT_STEP: the per-env flag "the episode ended on the previous call, the next call is the reset"
 * (gymnasium >= 1.0 vector envs; the reference itself never autoresets).  mdpp_get_state_* leave it out and
 * mdpp_set_state_* clear it; a checkpoint taken between the two calls carries it through these (uint8[N], host).
 * Call mdpp_set_reset_pending AFTER mdpp_set_state_*. */
int mdpp_get_reset_pending(mdpp_env *h, uint8_t *pending_host);
int mdpp_set_reset_pending(mdpp_env *h, const uint8_t *pending_host);

/* cfg.episode_stats: what the reference accumulates per env object and logs at every reset() (rl_toy_env.py:2231-2247,
 * cleared :2360-2369).  Rows of `current_host` (double [nk][N], the running episode): 0 total_abs_noise_in_reward_episode
 * (:1984), 1 total_reward_episode (:1985: the reward after the delay line and the every-n mask, before noise, scale and shift;
 * accumulated in float32 where the reference's reward is np.float32), 2 total_noisy_transitions_episode (discrete :1620, grid
 * :1746; 0 for continuous envs), 3 .. 3 + D - 1 total_abs_noise_in_transition_episode per dimension (continuous :1686);
 * nk = 3, or 3 + D for continuous envs.  `last_host` (double [nk + 1][N]): the same rows of the episode the latest reset()
 * of each env ended -- in-rollout autoresets and mdpp_reset alike -- and in row nk its total_transitions_episode.  Either
 * pointer may be NULL.  Synchronises the device. */
int mdpp_get_episode_stats(mdpp_env *h, double *current_host, double *last_host);

/* Kernel selection (see MDPP_OPT_*): disable_mask replaces the handle's current mask (0 = default dispatch). */
int mdpp_set_options(mdpp_env *h, uint32_t disable_mask);
/* Name (with template arguments) of the kernel mdpp_step_n(h, K, ...) would launch for this handle
 * right now -- K = 1: mdpp_step -- decided by the same code that launches it, nothing is launched.
 * Image handles: the renderer (the dominant kernel).  The string lives in the handle until the next call. */
const char *mdpp_kernel_name(mdpp_env *h, int K);

/* Philox-mode streams made visible (diagnostics, offline reproduction of the noise): out_dev[e][j]
 * (double, device) = the j-th standard normal of stream (seed, env_id0 + e, tick, stream id) -- a float32
 * Box-Muller pair per 64-bit draw, see mdp_playground_amd/csrc/mdpp_rng.hpp.  Runs on the current device.
 * (The reward noise of a DISCRETE env at tick t is normal t & 3 of stream (seed, env, t >> 2, 13): call with tick = t >> 2,
 *  stream = 13, n_per_env = 4.) */
int mdpp_philox_normals(uint64_t seed, int64_t env_id0, uint64_t tick, uint32_t stream, int32_t n_envs,
                        int32_t n_per_env, double *out_dev, void *hip_stream);

/* Per-env sticky status bits (MDPP_STATUS_*), cleared by the call. flags_host: uint32[N]. */
int mdpp_status(mdpp_env *h, uint32_t *flags_host);

/* Kernel timing with HIP events on the caller's stream (bench.py roofline leg):
 * begin/end bracket any number of launches; returns elapsed milliseconds. */
int mdpp_timer_begin(mdpp_env *h, void *stream);
int mdpp_timer_end(mdpp_env *h, void *stream, float *ms_out);

/* ---- peer-copy gather of the observation shards (ABI 7; SURVEY.md 8e "Collective") ----------------------------------
 * The path's one collective -- every rank gets the concatenated current-observation tensor -- as device-to-device copies
 * on the copy engines instead of a collective KERNEL: the rollout kernels hold every compute unit, so an RCCL all-gather
 * can only run between two launches, a copy engine runs beside them.  No counterpart in the reference (one process, one
 * env); `dist.PeerGatherer` is the binding, `ObsGatherer` (RCCL all_gather_into_tensor) stays the default and `value`.
 *   mdpp_peer_create   a buffer [slots][world][shard_bytes] on `device` (+ one 64-bit flag per slot and rank)
 *   mdpp_peer_handle   its hipIpcMemHandle_t (MDPP_PEER_HANDLE_BYTES bytes); exchange them over any channel
 *   mdpp_peer_open     maps the other ranks' buffers: `handles` = world x MDPP_PEER_HANDLE_BYTES bytes in rank order
 *   mdpp_peer_push     behind what `stream` has enqueued: shard_dev (shard_bytes on this device) -> row `rank` of slot
 *                      `slot` of EVERY rank's buffer, then `seq` into this rank's flag there (side stream of the handle)
 *   mdpp_peer_fence    `stream` waits (an event) until THIS rank's copies of the slot's latest push have left: the shard
 *                      buffer may be overwritten after that
 *   mdpp_peer_wait     `stream` waits until every rank's flag of `slot` is >= seq (a one-wave kernel polling device
 *                      memory, bounded: a timeout sets a status bit per missing rank -- mdpp_peer_status -- never hangs)
 *   mdpp_peer_buffer   device pointer of slot `slot`: [world][shard_bytes], rank-major = global env-id order */
#define MDPP_PEER_HANDLE_BYTES 64
typedef struct mdpp_peer mdpp_peer;
int mdpp_peer_create(int device, int world, int rank, size_t shard_bytes, int slots, mdpp_peer **out);
int mdpp_peer_handle(mdpp_peer *p, void *handle_out);
int mdpp_peer_open(mdpp_peer *p, const void *handles);
int mdpp_peer_push(mdpp_peer *p, int slot, const void *shard_dev, uint64_t seq, void *stream);
int mdpp_peer_fence(mdpp_peer *p, int slot, void *stream);
int mdpp_peer_wait(mdpp_peer *p, int slot, uint64_t seq, void *stream);
void *mdpp_peer_buffer(mdpp_peer *p, int slot);
int mdpp_peer_status(mdpp_peer *p, uint32_t *status_out, int *finegrained_out);
const char *mdpp_peer_last_error(mdpp_peer *p);
int mdpp_peer_destroy(mdpp_peer *p);

/* What the memory system of the current device gives plain streaming kernels (no handle; nothing of the reference):
 * `reps` launches of a 16-bytes-per-lane kernel over `nbytes` on `stream` (one 16 KiB tile per workgroup, tiles in address
 * order), HIP events around them -> *ms_out (all reps).  mode 0: copy src -> dst (2 x nbytes moved per launch; the float4
 * copy MI355X_MICROARCH.md measures at 6.29 TB/s), 1: fill dst (src unused), 2: read src (dst: a device pointer to 4 scratch
 * bytes); ABI 8: 3 = copy, 4 = fill with non-temporal stores.  bench.py prices `roofline.frac` beside the fastest form of
 * each (`peak_measured`). */
int mdpp_probe_hbm(int mode, void *dst_dev, const void *src_dev, size_t nbytes, int reps, void *stream, float *ms_out);
/* ABI 8.  The floor of the one-launch-per-step API on this device: n launches of an EMPTY kernel (`workgroups` x 64 threads)
 * back to back on `stream` -> the host's time per launch call and the device's time per launch (HIP events), microseconds.
 * bench.py reports mdpp_step beside it (`single_step.launch_floor`): below about 3.5 us per launch eager stepping is bound
 * by the host's launch call, not by the kernel. */
int mdpp_probe_launch(int n, int workgroups, void *stream, float *host_us_out, float *device_us_out);

/* ---- GymEnvWrapper-style post-processor ---------------------------------------
 * What the reference's mdp_playground/envs/gym_env_wrapper.py does around ANY inner env, for a batch of N
 * independent instances whose (obs, reward, done) the caller supplies as device tensors -- from
 * libmdpp_hip's own envs or from any other batched simulator.  Instance i is one GymEnvWrapper object:
 * one generator (the wrapper's _np_random, seeded by mdpp_post_seed_streams or a Philox stream), one reward
 * FIFO.  Replaces, per entry point:
 *   mdpp_post_actions   discrete action noise                                   gym_env_wrapper.py:354-366
 *   mdpp_post_step(_n)  continuous observation noise :367-373, :400-402; image canvas + shift :404-405,
 *                       :523-618; reward delay / flush on done / terminal reward / noise / scale / shift :407-432
 *   mdpp_post_reset     reset(): buffer refilled with zeros, image of the first observation :441-486
 * Rewards are float64 in and out (the reference computes them as Python floats).  Upstream's `done` branch
 * raises TypeError (list * float, :410); what is implemented is its evident numpy meaning, see INTEGRATION.md. */
typedef struct mdpp_post mdpp_post;
typedef struct {
    int32_t abi_version;        /* MDPP_ABI_VERSION */
    int32_t num_envs;
    int64_t env_id_offset;      /* global id of local instance 0 (Philox keys) */
    int32_t rng_mode;           /* MDPP_RNG_* */
    uint64_t philox_seed;
    int32_t continuous;         /* config["state_space_type"] == "continuous" */
    int32_t n_actions;          /* discrete: env.action_space.n */
    int32_t obs_dim, obs_f64;   /* continuous: length of an observation and its dtype (0 float32, 1 float64) */
    int32_t delay;              /* 0..128 */
    int32_t has_transition_noise; double transition_noise;   /* discrete: P(action replaced), continuous: std of the obs noise */
    int32_t has_reward_noise; double reward_noise;
    double reward_scale, reward_shift, term_state_reward;
    int32_t autoreset;          /* 1: the caller's env resets itself at done -> the buffer is refilled with zeros after a done step */
    /* image_transforms (discrete envs with uint8 [H][W][C] observations): canvas uint8 [W + 2 pad][H + 2 pad][C] */
    int32_t image, img_h, img_w, img_c, img_pad, img_has_shift, img_sh_quant;
} mdpp_post_config;
int mdpp_post_create(const mdpp_post_config *cfg, int device, mdpp_post **out);
void mdpp_post_destroy(mdpp_post *h);
const char *mdpp_post_last_error(const mdpp_post *h);
int mdpp_post_seed_streams(mdpp_post *h, const uint64_t *words_host);   /* uint64 [N][6], as mdpp_seed_streams */
int mdpp_post_get_streams(mdpp_post *h, uint64_t *words_host);
int mdpp_post_get_reward_buffer(mdpp_post *h, double *ring_host);       /* double [N][delay], [0] pays out next */
/* mask_dev NULL = every instance.  Image handles: obs_in uint8 [N][H][W][C] -> obs_out uint8 [N][W+2p][H+2p][C]
 * (instances outside the mask untouched); others: both NULL. */
int mdpp_post_reset(mdpp_post *h, const uint8_t *mask_dev, const void *obs_in_dev, void *obs_out_dev, void *stream);
/* the actions the inner envs receive: int32 [N] -> int32 [N] (may alias) */
int mdpp_post_actions(mdpp_post *h, const int32_t *actions_in_dev, int32_t *actions_out_dev, void *stream);
/* obs_in / obs_out: continuous float32|float64 [K][N][obs_dim]; image uint8 [K][N][H][W][C] -> [K][N][W+2p][H+2p][C];
 * otherwise NULL (observations pass through).  reward double [K][N], done uint8 [K][N] (terminated). */
int mdpp_post_step(mdpp_post *h, const void *obs_in_dev, const double *reward_in_dev, const uint8_t *done_dev,
                   void *obs_out_dev, double *reward_out_dev, void *stream);
int mdpp_post_step_n(mdpp_post *h, int K, const void *obs_in_dev, const double *reward_in_dev, const uint8_t *done_dev,
                     void *obs_out_dev, double *reward_out_dev, void *stream);

/* Episode statistics of a block of per-step outputs (what RLlib reports per training iteration and the reference's
 * callbacks write to the stats CSV, config_processor.py:275-407; mdp_playground_amd/stats_csv.py EpisodeStats): for
 * every instance the running return (double [N]) and length (int64 [N]) are carried through the K rows of
 * reward [K][N] (float32, or float64 when reward_is_f64) with end flags ended | ended2 (uint8 [K][N]; ended2 may be
 * NULL), and over the episodes that end inside the block *sum_ret, *sum_len, *count (device scalars) grow by the sum of
 * their returns, the sum of their lengths and their number.  Deterministic (no atomics).  scratch_dev: at least
 * 24 * ceil(N / 256) bytes.  Runs on the current device. */
int mdpp_episode_stats(int32_t K, int32_t N, const void *reward_dev, int32_t reward_is_f64, const uint8_t *ended_dev,
                       const uint8_t *ended2_dev, double *ret_dev, int64_t *len_dev, double *sum_ret_dev,
                       int64_t *sum_len_dev, int64_t *count_dev, void *scratch_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MDPP_H */

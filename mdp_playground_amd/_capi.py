"""ctypes binding of libmdpp_hip.so (include/mdpp.h).  No fallback: if the HIP library is
missing or does not export the ABI, importing the vector env raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmdpp_hip.so")

MDPP_ABI_VERSION = 8
MAX_DIM, MAX_ORDER, MAX_BOXES = 32, 4, 8
KIND_DISCRETE, KIND_CONTINUOUS, KIND_GRID = 0, 1, 2
REWARD_SEQUENCES, REWARD_STATE_ACTION = 0, 1
CREWARD_MOVE_TO_A_POINT, CREWARD_MOVE_ALONG_A_LINE = 0, 1
RNG_NUMPY_PCG64, RNG_PHILOX = 0, 1
AUTORESET_DISABLED, AUTORESET_SAME_STEP, AUTORESET_NEXT_STEP = 0, 1, 2
OBS_I64, OBS_I32, OBS_F32, OBS_IMAGE_U8 = 0, 1, 2, 3
STREAM_ENV, STREAM_SPACE, STREAM_IMAGE, STREAM_SPACE_IRR, STREAM_ACTION = 0, 1, 2, 3, 4
STATUS_BAD_ACTION = 1
PEER_HANDLE_BYTES = 64
# MDPP_OPT_* kernel-selection switches (mdpp_set_options)
OPTIONS = {"NO_PIPE": 1 << 0, "NO_HELPER": 1 << 1, "NO_PARK": 1 << 2, "NO_CFAST": 1 << 3, "NO_QUIET": 1 << 4,
           "NO_QUIET_NOISE": 1 << 5, "NO_DUO": 1 << 6, "NO_TRIO": 1 << 7, "NO_GFAST": 1 << 8,
           "NO_GFAST_NOISE": 1 << 9, "NO_IMGFAST": 1 << 10, "NO_IMG_OVERLAP": 1 << 11, "NO_PHILOX_FAST": 1 << 12, "NO_LEAN": 1 << 13,
           "NO_IMG_NEARTAB": 1 << 14, "NO_STEP1": 1 << 15, "NO_SIGMA0": 1 << 16, "NO_QUIET_SF": 1 << 17}

EXPORTS = [
    "mdpp_abi_version", "mdpp_create", "mdpp_destroy", "mdpp_last_error",
    "mdpp_upload_discrete_tables", "mdpp_upload_image_templates", "mdpp_seed_streams",
    "mdpp_get_streams", "mdpp_reset", "mdpp_step", "mdpp_step_n", "mdpp_get_state_discrete",
    "mdpp_set_state_discrete", "mdpp_get_state_continuous", "mdpp_set_state_continuous",
    "mdpp_status", "mdpp_timer_begin", "mdpp_timer_end",
    "mdpp_upload_discrete_irrelevant", "mdpp_get_state_irrelevant", "mdpp_set_state_irrelevant",
    "mdpp_get_state_grid", "mdpp_set_state_grid", "mdpp_upload_image_disc", "mdpp_upload_image_lines",
    "mdpp_set_options", "mdpp_kernel_name", "mdpp_philox_normals",
    "mdpp_graph_replay_exact", "mdpp_graph_capture", "mdpp_graph_set_tick_offset", "mdpp_tick", "mdpp_get_reset_pending", "mdpp_set_reset_pending",
    "mdpp_get_episode_stats", "mdpp_get_line_history", "mdpp_set_line_history",
    "mdpp_post_create", "mdpp_post_destroy", "mdpp_post_last_error", "mdpp_post_seed_streams", "mdpp_post_get_streams",
    "mdpp_post_get_reward_buffer", "mdpp_post_reset", "mdpp_post_actions", "mdpp_post_step", "mdpp_post_step_n",
    "mdpp_episode_stats", "mdpp_probe_hbm", "mdpp_probe_launch",
    "mdpp_peer_create", "mdpp_peer_handle", "mdpp_peer_open", "mdpp_peer_push", "mdpp_peer_fence", "mdpp_peer_wait", "mdpp_peer_buffer",
    "mdpp_peer_status", "mdpp_peer_last_error", "mdpp_peer_destroy",
]


class MdppConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("kind", C.c_int32), ("num_envs", C.c_int32),
        ("env_id_offset", C.c_int64), ("rng_mode", C.c_int32), ("autoreset", C.c_int32),
        ("max_episode_steps", C.c_int32), ("obs_dtype", C.c_int32), ("philox_seed", C.c_uint64),
        ("delay", C.c_int32), ("every_n", C.c_int32), ("has_reward_noise", C.c_int32),
        ("reward_noise", C.c_double), ("reward_scale", C.c_double), ("reward_shift", C.c_double),
        ("term_state_reward", C.c_double),
        ("S", C.c_int32), ("A", C.c_int32), ("L", C.c_int32), ("num_tables", C.c_int32),
        ("unit_rewards", C.c_int32), ("reward_kind", C.c_int32), ("has_transition_noise", C.c_int32),
        ("transition_noise", C.c_double),
        ("irrelevant", C.c_int32), ("S_irr", C.c_int32), ("A_irr", C.c_int32),
        ("D", C.c_int32), ("n_rel", C.c_int32), ("order", C.c_int32), ("reward_function", C.c_int32),
        ("rel_idx", C.c_int32 * MAX_DIM), ("make_denser", C.c_int32), ("has_p_noise", C.c_int32),
        ("p_noise", C.c_double), ("inertia", C.c_double), ("time_unit", C.c_double),
        ("state_space_max", C.c_double), ("action_space_max", C.c_double),
        ("target_radius", C.c_double), ("action_loss_weight", C.c_double),
        ("target", C.c_float * MAX_DIM), ("n_boxes", C.c_int32),
        ("box_lo", C.c_float * (MAX_BOXES * MAX_DIM)), ("box_hi", C.c_float * (MAX_BOXES * MAX_DIM)),
        ("grid_dims", C.c_int32), ("grid_shape", C.c_int32 * 4), ("grid_target", C.c_int32 * 2),
        ("image", C.c_int32), ("img_w", C.c_int32), ("img_h", C.c_int32),
        ("img_has_scale", C.c_int32), ("img_has_shift", C.c_int32), ("img_has_rotate", C.c_int32),
        ("img_has_flip", C.c_int32), ("img_sh_quant", C.c_int32), ("img_ro_quant", C.c_int32),
        ("img_r0", C.c_int32), ("img_r_min", C.c_int32), ("img_r_max", C.c_int32),
        ("img_log_min_r", C.c_double), ("img_log_max_r", C.c_double), ("img_tpl_size", C.c_int32),
        ("episode_stats", C.c_int32), ("target_f64", C.c_int32),
    ]


class MdppPostConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("num_envs", C.c_int32), ("env_id_offset", C.c_int64), ("rng_mode", C.c_int32),
        ("philox_seed", C.c_uint64), ("continuous", C.c_int32), ("n_actions", C.c_int32), ("obs_dim", C.c_int32),
        ("obs_f64", C.c_int32), ("delay", C.c_int32), ("has_transition_noise", C.c_int32),
        ("transition_noise", C.c_double), ("has_reward_noise", C.c_int32), ("reward_noise", C.c_double),
        ("reward_scale", C.c_double), ("reward_shift", C.c_double), ("term_state_reward", C.c_double),
        ("autoreset", C.c_int32), ("image", C.c_int32), ("img_h", C.c_int32), ("img_w", C.c_int32),
        ("img_c", C.c_int32), ("img_pad", C.c_int32), ("img_has_shift", C.c_int32), ("img_sh_quant", C.c_int32),
    ]


class MdppError(RuntimeError):
    pass


_lib = None


def load():
    """Load libmdpp_hip.so and declare every prototype of include/mdpp.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MdppError(
            f"{LIB_PATH} is missing: build it with `python -m mdp_playground_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    # PyTorch-ROCm ships its own libamdhip64: load it first, so that the library binds to the HIP
    # runtime the tensors live in (a second copy of the runtime in one process sees no device)
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise MdppError(f"{LIB_PATH} does not export {name}")
    vp, i32 = C.c_void_p, C.c_int
    L.mdpp_abi_version.restype = i32
    L.mdpp_create.argtypes = [C.POINTER(MdppConfig), i32, C.POINTER(vp)]
    L.mdpp_destroy.argtypes = [vp]
    L.mdpp_destroy.restype = None
    L.mdpp_last_error.argtypes = [vp]
    L.mdpp_last_error.restype = C.c_char_p
    L.mdpp_upload_discrete_tables.argtypes = [vp] * 7
    L.mdpp_upload_image_templates.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    L.mdpp_seed_streams.argtypes = [vp, i32, vp]
    L.mdpp_get_streams.argtypes = [vp, i32, vp]
    L.mdpp_reset.argtypes = [vp, vp, vp, vp]
    L.mdpp_step.argtypes = [vp] * 8
    L.mdpp_step_n.argtypes = [vp, i32] + [vp] * 6
    L.mdpp_get_state_discrete.argtypes = [vp] * 4
    L.mdpp_set_state_discrete.argtypes = [vp] * 4
    L.mdpp_get_state_continuous.argtypes = [vp] * 7
    L.mdpp_set_state_continuous.argtypes = [vp] * 7
    L.mdpp_upload_discrete_irrelevant.argtypes = [vp] * 4
    L.mdpp_get_state_irrelevant.argtypes = [vp, vp]
    L.mdpp_set_state_irrelevant.argtypes = [vp, vp]
    L.mdpp_upload_image_disc.argtypes = [vp, vp]
    L.mdpp_upload_image_lines.argtypes = [vp, vp]
    L.mdpp_get_state_grid.argtypes = [vp] * 4
    L.mdpp_set_state_grid.argtypes = [vp] * 4
    L.mdpp_status.argtypes = [vp, vp]
    L.mdpp_set_options.argtypes = [vp, C.c_uint32]
    L.mdpp_kernel_name.argtypes = [vp, i32]
    L.mdpp_kernel_name.restype = C.c_char_p
    L.mdpp_graph_replay_exact.argtypes = [vp, i32]
    L.mdpp_graph_capture.argtypes = [vp, i32]
    L.mdpp_graph_set_tick_offset.argtypes = [vp, C.c_int64, vp]
    L.mdpp_tick.argtypes = [vp, C.c_int64, C.POINTER(C.c_uint64)]
    L.mdpp_get_episode_stats.argtypes = [vp, vp, vp]
    L.mdpp_get_line_history.argtypes = [vp, vp]
    L.mdpp_set_line_history.argtypes = [vp, vp]
    L.mdpp_get_reset_pending.argtypes = [vp, vp]
    L.mdpp_set_reset_pending.argtypes = [vp, vp]
    L.mdpp_philox_normals.argtypes = [C.c_uint64, C.c_int64, C.c_uint64, C.c_uint32, C.c_int32, C.c_int32, vp, vp]
    L.mdpp_post_create.argtypes = [C.POINTER(MdppPostConfig), i32, C.POINTER(vp)]
    L.mdpp_post_destroy.argtypes = [vp]
    L.mdpp_post_destroy.restype = None
    L.mdpp_post_last_error.argtypes = [vp]
    L.mdpp_post_last_error.restype = C.c_char_p
    L.mdpp_post_seed_streams.argtypes = [vp, vp]
    L.mdpp_post_get_streams.argtypes = [vp, vp]
    L.mdpp_post_get_reward_buffer.argtypes = [vp, vp]
    L.mdpp_post_reset.argtypes = [vp] * 5
    L.mdpp_post_actions.argtypes = [vp] * 4
    L.mdpp_post_step.argtypes = [vp] * 7
    L.mdpp_post_step_n.argtypes = [vp, i32] + [vp] * 6
    L.mdpp_episode_stats.argtypes = [i32, i32, vp, i32] + [vp] * 9
    L.mdpp_probe_hbm.argtypes = [i32, vp, vp, C.c_size_t, i32, vp, C.POINTER(C.c_float)]
    L.mdpp_probe_launch.argtypes = [i32, i32, vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.mdpp_peer_create.argtypes = [i32, i32, i32, C.c_size_t, i32, C.POINTER(vp)]
    L.mdpp_peer_handle.argtypes = [vp, vp]
    L.mdpp_peer_open.argtypes = [vp, vp]
    L.mdpp_peer_push.argtypes = [vp, i32, vp, C.c_uint64, vp]
    L.mdpp_peer_fence.argtypes = [vp, i32, vp]
    L.mdpp_peer_wait.argtypes = [vp, i32, C.c_uint64, vp]
    L.mdpp_peer_buffer.argtypes = [vp, i32]
    L.mdpp_peer_buffer.restype = vp
    L.mdpp_peer_status.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(i32)]
    L.mdpp_peer_last_error.argtypes = [vp]
    L.mdpp_peer_last_error.restype = C.c_char_p
    L.mdpp_peer_destroy.argtypes = [vp]
    L.mdpp_timer_begin.argtypes = [vp, vp]
    L.mdpp_timer_end.argtypes = [vp, vp, C.POINTER(C.c_float)]
    if L.mdpp_abi_version() != MDPP_ABI_VERSION:
        raise MdppError("libmdpp_hip.so ABI version mismatch")
    _lib = L
    return L


def check(lib, handle, rc, what):
    if rc != 0:
        msg = lib.mdpp_last_error(handle)
        raise MdppError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def nptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)

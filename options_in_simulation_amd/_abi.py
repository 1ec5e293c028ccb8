"""ctypes mirror of include/ois.h (struct layout, enums). Pure data, no library load."""
import ctypes as C

ABI_VERSION = 6

OK, EINVAL, ENOMEM, EHIP, ESTATE = 0, -1, -2, -3, -4

MIX_SWITCH, MIX_BLEND = 0, 1
GAMMA_SCHEDULE, GAMMA_CONST = 0, 1
GUIDE_EXECUTED, GUIDE_POLICY = 0, 1
CMD_RAND, CMD_BOX, CMD_GRID, CMD_EVAL, CMD_LEVEL = 0, 1, 2, 3, 4
POLICY_ZERO, POLICY_RANDOM, POLICY_EXTERNAL = 0, 1, 2

(BUF_OBS, BUF_REW, BUF_DONE, BUF_TIMEOUT, BUF_WHEEL_TORQS, BUF_POSE, BUF_QUAT, BUF_LINVEL,
 BUF_ANGVEL, BUF_WHEEL_VEL, BUF_COMMANDS, BUF_COUNTERS, BUF_SUMS_A, BUF_SUMS_B, BUF_PARAMS,
 BUF_EPISODE_INFO, BUF_ENV0_DIAG, BUF_SUMS_C, BUF_LAMBDA_N, BUF_LAMBDA_L, BUF_LAMBDA_T,
 BUF_ENV_ORIGINS, BUF_TERRAIN_IDX, BUF_CMD_FRACTION, BUF_COUNT) = range(25)
HEIGHT_POINTS = 140

EPISODE_INFO_LEN = 32
ENV0_DIAG_LEN = 32

# slot index inside BUF_EPISODE_INFO -> key of extras["episode"] (KAMMA.py:205-221, 470-488)
EI_COUNT = 0
EI_STEP = 19
EI_TERRAIN_LEVEL = 18
EPISODE_INFO_KEYS = {
    1: "rew_r1: Tracking error reward (squared errors)",
    2: "rew_r2: Convergence reward (squared accelerations)",
    3: "rew_r3: Torque penalty (sum of squared torques)",
    4: "rew_Dense reward",
    5: "rew_Sparse reward",
    6: "rew_Guiding reward",
    7: "rew_Observed reward",
    8: "rew_Final reward",
    9: "rew_r1/Final reward",
    10: "rew_r2/Final reward",
    11: "rew_r3/Final reward",
    12: "rew_Dense/Final reward",
    13: "rew_Sparse/Final reward",
    14: "rew_Guiding/Final reward",
    15: "rew_Observed/Final reward",
    16: "gamma_1 action assist",
    17: "gamma_2 reward assist",
    18: "terrain_level",
}
# eval-variant extras (KAMMA_eval.py:224-226)
EPISODE_INFO_EVAL_KEYS = {20: "rew_Smoothness", 21: "rew_Tracking error", 22: "rew_Desired velocity"}
# keys the reference pre-declares but never writes (KAMMA.py:217-220): always 0
EPISODE_INFO_ZERO_KEYS = (
    "rew_Dense reward/Final reward",
    "rew_Sparse reward/Final reward",
    "rew_Guiding reward/Final reward",
    "rew_Observed reward/Final reward",
)


# the KA-DDPG files keep ONE assist factor and log it under one key (KA_DDPG_4D.py:484; 1D / 2D and their eval twins alike)
EPISODE_INFO_SINGLE_GAMMA_KEY = "gamma assist"


def episode_info_layout(eval_keys: bool, single_gamma: bool):
    """[(key, slot)] of extras["episode"] in the ORDER the reference's dict has (KAMMA.py:204-221, 470-488): the declared
    episode sums (the four `... reward/Final reward` keys are declared and never written: slot 31, always 0; the eval
    files declare three more, KAMMA_eval.py:224-226), then the four ratio keys reset_idx adds under new names, the assist
    factor(s), terrain_level. One key set per variant: `vecenv` and the multi-rank merge of `dist` both build from this."""
    out = [(EPISODE_INFO_KEYS[s], s) for s in range(1, 12)]
    out += [(k, 31) for k in EPISODE_INFO_ZERO_KEYS]
    if eval_keys:
        out += [(k, s) for s, k in EPISODE_INFO_EVAL_KEYS.items()]
    out += [(EPISODE_INFO_KEYS[s], s) for s in range(12, 16)]
    out += [(EPISODE_INFO_SINGLE_GAMMA_KEY, 16)] if single_gamma else [(EPISODE_INFO_KEYS[16], 16), (EPISODE_INFO_KEYS[17], 17)]
    out.append((EPISODE_INFO_KEYS[18], 18))
    return out


EPISODE_INFO_FIRST_SLOT, EPISODE_INFO_LAST_SLOT = 1, 22  # the slots a key can name, besides the always-zero slot 31


def episode_info_items(vec, eval_keys: bool, single_gamma: bool):
    """(key, value) pairs of extras["episode"] (KAMMA.py:480-488) from the 32-slot vector of OIS_BUF_EPISODE_INFO, in the
    key set and order of the reference file the variant stands for: `vec` is anything indexable (numpy row, 1-D tensor)."""
    for key, slot in episode_info_layout(eval_keys, single_gamma):
        yield key, vec[slot]


class OisConfig(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_uint32),
        ("num_envs", C.c_int32),
        ("arena_offset", C.c_int64),
        ("seed", C.c_uint64),
        ("action_dim", C.c_int32),
        ("mix_mode", C.c_int32),
        ("gamma1_mode", C.c_int32),
        ("gamma2_mode", C.c_int32),
        ("gamma1_const", C.c_float),
        ("gamma2_const", C.c_float),
        ("guiding_mode", C.c_int32),
        ("cmd_sampler", C.c_int32),
        ("max_sim_steps", C.c_float),
        ("eval_sums", C.c_int32),
        ("sim_dt", C.c_float),
        ("decimation", C.c_int32),
        ("extra_substeps", C.c_int32),
        ("policy_dt", C.c_float),
        ("max_episode_length", C.c_int32),
        ("episode_length_s", C.c_float),
        ("Kp", C.c_float),
        ("Kp_omega", C.c_float),
        ("vehicle_mass", C.c_float),
        ("vehicle_inertia", C.c_float),
        ("action_scale", C.c_float),
        ("torque_clip", C.c_float),
        ("w1", C.c_float),
        ("w2", C.c_float),
        ("w3", C.c_float),
        ("sparse_bonus", C.c_float),
        ("sparse_threshold", C.c_float),
        ("terminal_reward", C.c_float),
        ("cmd_x_lo", C.c_float),
        ("cmd_x_hi", C.c_float),
        ("cmd_y_lo", C.c_float),
        ("cmd_y_hi", C.c_float),
        ("cmd_yaw_lo", C.c_float),
        ("cmd_yaw_hi", C.c_float),
        ("cmd_x_span", C.c_float),
        ("cmd_y_span", C.c_float),
        ("cmd_yaw_span", C.c_float),
        ("init_pos", C.c_float * 3),
        ("init_quat", C.c_float * 4),
        ("init_lin_vel", C.c_float * 3),
        ("init_ang_vel", C.c_float * 3),
        ("init_dof_vel", C.c_float * 4),
        ("spawn_xy_range", C.c_float),
        ("bounds", C.c_float),
        ("clip_obs", C.c_float),
        ("eval_ramp_steps", C.c_float),
        ("gravity", C.c_float),
        ("body_mass", C.c_float),
        ("body_inertia", C.c_float * 3),
        ("com_height", C.c_float),
        ("wheel_pos", C.c_float * 3),
        ("wheel_radius", C.c_float),
        ("wheel_spin_inertia", C.c_float),
        ("wheel_damping", C.c_float),
        ("friction", C.c_float),
        ("baumgarte", C.c_float),
        ("max_depenetration_velocity", C.c_float),
        ("max_angular_velocity", C.c_float),
        ("solver_iterations", C.c_int32),
        ("wheel_limit_velocity", C.c_float),
        ("wheel_limit_damping", C.c_float),
        ("curriculum_threshold_high", C.c_float),
        ("env0_reward_guard", C.c_int32),
        ("reserved_", C.c_int32 * 4),
    ]


class OisTerrain(C.Structure):
    """mirror of `ois_terrain` (include/ois.h)"""
    _fields_ = [
        ("struct_bytes", C.c_uint32), ("rows", C.c_int32), ("cols", C.c_int32),
        ("horizontal_scale", C.c_float), ("vertical_scale", C.c_float), ("border_size", C.c_float),
        ("num_levels", C.c_int32), ("num_types", C.c_int32), ("curriculum", C.c_int32),
        ("threshold_low", C.c_float), ("threshold_high", C.c_float), ("max_level", C.c_int32),
        ("footprint_stride", C.c_int32),
    ]


class OisStats(C.Structure):
    _fields_ = [
        ("episodes", C.c_double),
        ("return_sum", C.c_double),
        ("return_sqsum", C.c_double),
        ("return_min", C.c_double),
        ("return_max", C.c_double),
        ("length_sum", C.c_double),
        ("steps", C.c_double),
        ("reserved", C.c_double),
    ]

"""ctypes binding of librlcontrol_hip.so (include/rlcontrol_hip.h).

The HIP library is the only compute path of this package: if it is missing, or no MI355X is
visible, construction fails loudly -- there is no CPU fallback (the CPU restatement under oracle/
is test infrastructure and is never imported from here).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RLCONTROL_HIP_LIB", os.path.join(_HERE, "librlcontrol_hip.so"))

# every symbol include/rlcontrol_hip.h declares (tests check the .so exports all of them)
EXPORTS = (
    "rlc_last_error", "rlc_version", "rlc_device_count",
    "rlc_destroy", "rlc_sync",
    "rlc_ddpg_create", "rlc_ddpg_param_count",
    "rlc_ddpg_set_blob", "rlc_ddpg_get_blob", "rlc_ddpg_set_beta_powers", "rlc_ddpg_get_beta_powers",
    "rlc_ddpg_init_target",
    "rlc_replay_add", "rlc_replay_add_batch", "rlc_replay_fill_all_dev", "rlc_replay_size",
    "rlc_replay_gather", "rlc_replay_sample_indices",
    "rlc_ddpg_act", "rlc_ddpg_act_queue", "rlc_ddpg_act_fetch", "rlc_ddpg_act_explore", "rlc_ddpg_reset_noise", "rlc_ddpg_qval",
    "rlc_ddpg_update", "rlc_ddpg_update_batch", "rlc_ddpg_set_kernel", "rlc_ddpg_get_kernel", "rlc_ddpg_set_split",
    "rlc_ddpg_last_tap", "rlc_ddpg_enable_grad_taps", "rlc_debug_fail_next_split",
    "rlc_timer_begin", "rlc_timer_end",
    "rlc_sac_create", "rlc_sac_param_count", "rlc_sac_set_blob", "rlc_sac_get_blob", "rlc_sac_set_beta_powers",
    "rlc_sac_get_beta_powers", "rlc_sac_init_target", "rlc_sac_act", "rlc_sac_act_queue", "rlc_sac_act_fetch", "rlc_sac_update", "rlc_sac_update_batch",
    "rlc_sac_last_tap", "rlc_sac_enable_grad_taps", "rlc_sac_set_kernel", "rlc_sac_get_kernel",
    "rlc_kl_create", "rlc_kl_param_count", "rlc_kl_set_blob", "rlc_kl_get_blob", "rlc_kl_set_step", "rlc_kl_get_step",
    "rlc_kl_init_target", "rlc_kl_act", "rlc_kl_act_queue", "rlc_kl_act_fetch", "rlc_kl_update", "rlc_kl_update_batch", "rlc_kl_last_tap",
    "rlc_kl_enable_grad_taps", "rlc_kl_set_kernel", "rlc_kl_get_kernel", "rlc_kl_set_split",
    "rlc_naf_create", "rlc_naf_param_count", "rlc_naf_set_blob", "rlc_naf_get_blob", "rlc_naf_get_beta_powers",
    "rlc_naf_init_target", "rlc_naf_act", "rlc_naf_act_queue", "rlc_naf_act_fetch", "rlc_naf_update", "rlc_naf_update_batch", "rlc_naf_last_tap",
    "rlc_naf_enable_grad_taps", "rlc_naf_set_kernel", "rlc_naf_get_kernel",
    "rlc_ddpg_rollout_create", "rlc_ddpg_rollout_run", "rlc_sac_rollout_create", "rlc_sac_rollout_run", "rlc_kl_rollout_create", "rlc_kl_rollout_run",
    "rlc_naf_rollout_create", "rlc_naf_rollout_run", "rlc_rollout_counts", "rlc_rollout_train_log",
    "rlc_rollout_eval_log", "rlc_rollout_observation",
)


class RlcError(RuntimeError):
    pass


class rlc_ddpg_config(ctypes.Structure):
    _fields_ = [
        ("device", ctypes.c_int32), ("n_agents", ctypes.c_int32),
        ("state_dim", ctypes.c_int32), ("action_dim", ctypes.c_int32),
        ("shared_l1_dim", ctypes.c_int32), ("actor_l2_dim", ctypes.c_int32), ("critic_l2_dim", ctypes.c_int32),
        ("batch_size", ctypes.c_int32), ("buffer_size", ctypes.c_int64),
        ("clip_state", ctypes.c_int32), ("norm_type", ctypes.c_int32),
        ("tau", ctypes.c_float), ("reserved1", ctypes.c_float),
        ("state_min", ctypes.POINTER(ctypes.c_float)), ("state_max", ctypes.POINTER(ctypes.c_float)),
        ("action_min", ctypes.POINTER(ctypes.c_float)), ("action_max", ctypes.POINTER(ctypes.c_float)),
        ("actor_lr", ctypes.POINTER(ctypes.c_float)), ("critic_lr", ctypes.POINTER(ctypes.c_float)),
        ("seed", ctypes.POINTER(ctypes.c_uint64)),
        ("ou_theta", ctypes.c_float), ("ou_mu", ctypes.c_float), ("ou_sigma", ctypes.c_float),
        ("separate_networks", ctypes.c_int32),
    ]


class rlc_sac_config(ctypes.Structure):
    _fields_ = [
        ("device", ctypes.c_int32), ("n_agents", ctypes.c_int32), ("state_dim", ctypes.c_int32),
        ("action_dim", ctypes.c_int32),
        ("actor_l1_dim", ctypes.c_int32), ("actor_l2_dim", ctypes.c_int32), ("critic_l1_dim", ctypes.c_int32),
        ("critic_l2_dim", ctypes.c_int32),
        ("batch_size", ctypes.c_int32), ("clip_state", ctypes.c_int32), ("buffer_size", ctypes.c_int64),
        ("tau", ctypes.c_float), ("state_min0", ctypes.c_float), ("state_max0", ctypes.c_float),
        ("action_max0", ctypes.c_float),
        ("pi_lr", ctypes.POINTER(ctypes.c_float)), ("qf_vf_lr", ctypes.POINTER(ctypes.c_float)),
        ("entropy_scale", ctypes.POINTER(ctypes.c_float)), ("seed", ctypes.POINTER(ctypes.c_uint64)),
        ("norm_type", ctypes.c_int32), ("reserved0", ctypes.c_int32),
    ]


class rlc_kl_config(ctypes.Structure):
    _fields_ = [
        ("device", ctypes.c_int32), ("n_agents", ctypes.c_int32), ("state_dim", ctypes.c_int32),
        ("action_dim", ctypes.c_int32),
        ("actor_l1_dim", ctypes.c_int32), ("actor_l2_dim", ctypes.c_int32), ("critic_l1_dim", ctypes.c_int32),
        ("critic_l2_dim", ctypes.c_int32),
        ("batch_size", ctypes.c_int32), ("kind", ctypes.c_int32), ("optim_type", ctypes.c_int32),
        ("q_update_type", ctypes.c_int32), ("n_nodes", ctypes.c_int32), ("reserved0", ctypes.c_int32),
        ("buffer_size", ctypes.c_int64),
        ("tau", ctypes.c_float), ("action_max0", ctypes.c_float),
        ("node_actions", ctypes.POINTER(ctypes.c_float)), ("node_weights", ctypes.POINTER(ctypes.c_float)),
        ("pi_lr", ctypes.POINTER(ctypes.c_float)), ("qf_vf_lr", ctypes.POINTER(ctypes.c_float)),
        ("entropy_scale", ctypes.POINTER(ctypes.c_float)), ("seed", ctypes.POINTER(ctypes.c_uint64)),
    ]


class rlc_naf_config(ctypes.Structure):
    _fields_ = [
        ("device", ctypes.c_int32), ("n_agents", ctypes.c_int32), ("state_dim", ctypes.c_int32),
        ("action_dim", ctypes.c_int32), ("l1_dim", ctypes.c_int32), ("l2_dim", ctypes.c_int32),
        ("batch_size", ctypes.c_int32), ("clip_state", ctypes.c_int32), ("buffer_size", ctypes.c_int64),
        ("tau", ctypes.c_float), ("norm_type", ctypes.c_int32),
        ("state_min", ctypes.POINTER(ctypes.c_float)), ("state_max", ctypes.POINTER(ctypes.c_float)),
        ("action_max", ctypes.POINTER(ctypes.c_float)), ("learning_rate", ctypes.POINTER(ctypes.c_float)),
        ("seed", ctypes.POINTER(ctypes.c_uint64)), ("action_min", ctypes.POINTER(ctypes.c_float)),
    ]


class rlc_rollout_config(ctypes.Structure):
    _fields_ = [
        ("env_id", ctypes.c_int32), ("episode_steps_limit", ctypes.c_int32),
        ("total_steps_limit", ctypes.c_int64), ("eval_interval", ctypes.c_int64),
        ("eval_episodes", ctypes.c_int32), ("warmup_steps", ctypes.c_int32),
        ("max_train_episodes", ctypes.c_int32), ("reserved0", ctypes.c_int32),
        ("gamma", ctypes.c_double),
    ]


ENV_IDS = {"Pendulum-v0": 1}


_lib = None


def load():
    """Load the shared library; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RlcError("HIP library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); rlcontrol_amd has no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.rlc_last_error.restype = ctypes.c_char_p
    for name in EXPORTS:
        if name != "rlc_last_error":
            getattr(lib, name).restype = ctypes.c_int
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise RlcError(load().rlc_last_error().decode("utf-8", "replace"))


def fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def dptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def iptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)

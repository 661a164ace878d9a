"""ctypes loader of libddrl_pop.so.  Fails loudly: there is no fallback."""
import ctypes as C
import os

import numpy as np
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.environ.get("DDRL_POP_SO") or os.path.join(HERE, "libddrl_pop.so")   # DDRL_POP_SO: dev experiments only
DDRL_MAX_HIDDEN = 3
REAL = np.float32            # what a `float` of the C ABI is in the loaded library (float64 in the fp64 debug build)


class LibraryMissing(RuntimeError):
    pass


class DdrlError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ddrl_pop error %d: %s" % (code, msg))
        self.code = code


class DdpgConfigC(C.Structure):
    _fields_ = [
        ("n_agents", C.c_int), ("nb_sensors", C.c_int), ("nb_motors", C.c_int),
        ("n_hidden_a", C.c_int), ("hidden_unit_a", C.c_int * DDRL_MAX_HIDDEN),
        ("n_hidden_q", C.c_int), ("hidden_unit_q", C.c_int * DDRL_MAX_HIDDEN),
        ("mini_batch_size", C.c_int), ("replay_memory", C.c_int), ("inverting_grad", C.c_int),
        ("force_more_update", C.c_int), ("hidden_layer_type", C.c_int),
        ("actor_output_layer_type", C.c_int), ("momentum", C.c_int), ("nstep_return", C.c_int),
        ("batch_norm_actor", C.c_int), ("batch_norm_critic", C.c_int),
        ("gamma", C.c_double), ("tau_soft_update", C.c_double), ("alpha_a", C.c_double),
        ("alpha_v", C.c_double), ("decay_v", C.c_double),
        ("ewc", C.c_double), ("ewc_decay", C.c_double), ("ewc_fisher_beta", C.c_double),
        ("ewc_best_param_method", C.c_int), ("ewc_for_critic", C.c_int),
        ("device", C.c_int),
        ("td3", C.c_int), ("policy_freq", C.c_int), ("policy_noise", C.c_double), ("noise_clip", C.c_double),
        ("test_net", C.c_int), ("bn_adapt", C.c_int),
    ]


class NfacConfigC(C.Structure):
    _fields_ = [
        ("n_agents", C.c_int), ("nb_sensors", C.c_int), ("nb_motors", C.c_int),
        ("n_hidden_a", C.c_int), ("hidden_unit_a", C.c_int * DDRL_MAX_HIDDEN),
        ("n_hidden_v", C.c_int), ("hidden_unit_v", C.c_int * DDRL_MAX_HIDDEN),
        ("update_delta_neg", C.c_int), ("update_critic_first", C.c_int),
        ("number_fitted_iteration", C.c_int), ("stoch_iter_actor", C.c_int), ("stoch_iter_critic", C.c_int),
        ("batch_norm_actor", C.c_int), ("batch_norm_critic", C.c_int),
        ("actor_output_layer_type", C.c_int), ("hidden_layer_type", C.c_int), ("momentum", C.c_int),
        ("gae", C.c_int), ("inverting_gradient", C.c_int), ("max_trajectory", C.c_int),
        ("gamma", C.c_double), ("lambda_", C.c_double), ("alpha_a", C.c_double), ("alpha_v", C.c_double),
        ("ewc", C.c_double), ("ewc_decay", C.c_double), ("ewc_fisher_beta", C.c_double),
        ("ewc_best_param_method", C.c_int), ("ewc_for_critic", C.c_int),
        ("device", C.c_int),
        ("vnn_from_scratch", C.c_int), ("corrected_update_ac", C.c_int), ("corrected_update_ac_factor", C.c_double),
    ]


class EnvVtable(C.Structure):
    """ddrl_env_vtable: the reference's Environment interface as C callbacks (one instance per agent)"""
    RESET = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int)
    PERCEPTIONS = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_double))
    PERFORMANCE = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int)
    APPLY = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_double))
    FLAG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
    NEXT = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int)
    DUMP = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p, C.c_int)      # (a c_char_p argument would arrive as a copy)
    _fields_ = [("user", C.c_void_p), ("reset_episode", RESET), ("perceptions", PERCEPTIONS), ("performance", PERFORMANCE),
                ("apply", APPLY), ("running", FLAG), ("final_state", FLAG), ("next_instance", NEXT), ("dump", DUMP)]


# every symbol include/ddrl_pop.h declares: (name, restype, argtypes)
_vp, _i, _cp = C.c_void_p, C.c_int, C.c_char_p
SYMBOLS = [
    ("ddrl_last_error", _cp, []),
    ("ddrl_version", _cp, []),
    ("ddrl_ddpg_config_default", _i, [C.POINTER(DdpgConfigC)]),
    ("ddrl_ddpg_config_from_ini", _i, [_cp, C.POINTER(DdpgConfigC)]),
    ("ddrl_ddpg_create", _i, [C.POINTER(DdpgConfigC), C.c_uint64, C.POINTER(_vp)]),
    ("ddrl_pop_destroy", _i, [_vp]),
    ("ddrl_pop_sync", _i, [_vp]),
    ("ddrl_pop_shape", _i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    ("ddrl_pop_set_hyper", _i, [_vp, _cp, _vp]),
    ("ddrl_pop_num_params", _i, [_vp, _i]),
    ("ddrl_pop_set_weights", _i, [_vp, _i, _vp]),
    ("ddrl_pop_get_weights", _i, [_vp, _i, _vp]),
    ("ddrl_pop_get_adam", _i, [_vp, _i, _vp, _vp, _vp]),
    ("ddrl_pop_set_adam", _i, [_vp, _i, _vp, _vp, _vp]),
    ("ddrl_pop_seed_replay", _i, [_vp, _vp]),
    ("ddrl_pop_replay_append", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("ddrl_pop_replay_load", _i, [_vp, _i, _i, _vp]),
    ("ddrl_pop_replay_len", _i, [_vp, _vp]),
    ("ddrl_pop_infer", _i, [_vp, _vp, _vp]),
    ("ddrl_pop_infer_dev", _i, [_vp, _vp, _vp]),
    ("ddrl_pop_forward_batch", _i, [_vp, _i, _i, _vp, _vp]),
    ("ddrl_pop_ddpg_update", _i, [_vp, _vp, _vp]),
    ("ddrl_pop_ddpg_update_n", _i, [_vp, _i, _vp, _vp]),
    ("ddrl_pop_ddpg_last", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("ddrl_td3_config_from_ini", _i, [_cp, C.POINTER(DdpgConfigC)]),
    ("ddrl_td3_create", _i, [C.POINTER(DdpgConfigC), C.c_uint64, C.POINTER(_vp)]),
    ("ddrl_pop_td3_seed_noise", _i, [_vp, _vp]),
    ("ddrl_pop_td3_update", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("ddrl_pop_td3_last", _i, [_vp, _vp, _vp]),
    ("ddrl_agents_create_td3", _i, [_cp, _i, _i, _i, _i, C.c_uint64, C.POINTER(_vp)]),
    ("ddrl_pop_set_gates", _i, [_vp, _vp, _vp]),
    ("ddrl_pop_get_gates", _i, [_vp, _vp, _vp]),
    ("ddrl_pop_set_gate_mode", _i, [_vp, _i, _i, _i, _i, C.c_uint32]),
    ("ddrl_pop_ewc_setup_net", _i, [_vp, _i, _vp]),
    ("ddrl_pop_replay_clear", _i, [_vp, _vp]),
    ("ddrl_pop_update_best_param", _i, [_vp, _vp, _vp]),
    ("ddrl_pop_ewc_setup", _i, [_vp, _vp]),
    ("ddrl_pop_ewc_decay_update", _i, [_vp, _vp]),
    ("ddrl_pop_ewc_cost", _i, [_vp, _i, _vp]),
    ("ddrl_pop_get_ewc", _i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    ("ddrl_pop_save_run", _i, [_vp, _cp]),
    ("ddrl_pop_load_previous_run", _i, [_vp, _cp]),
    ("ddrl_cmaes_save_run", _i, [_vp, _cp]),
    ("ddrl_cmaes_load_previous_run", _i, [_vp, _cp]),
    ("ddrl_agents_save_run", _i, [_vp, _cp]),
    ("ddrl_agents_load_previous_run", _i, [_vp, _cp]),
    ("ddrl_agents_shape", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("ddrl_sim_create", _i, [_vp, C.POINTER(EnvVtable), _cp, _cp, _i, _i, C.POINTER(_vp)]),
    ("ddrl_sim_destroy", _i, [_vp]),
    ("ddrl_sim_run", _i, [_vp]),
    ("ddrl_sim_run_episode", _i, [_vp, _i, C.c_long, C.c_long]),
    ("ddrl_pop_set_profiling", _i, [_vp, _i]),
    ("ddrl_pop_profile_read", _i, [_vp, C.POINTER(C.c_double), C.POINTER(_i)]),
    ("ddrl_pop_debug_stamps", _i, [_vp, _i, _vp]),
    ("ddrl_pop_set_option", _i, [_vp, _cp, _i]),
    ("ddrl_agents_create_ddpg", _i, [_cp, _i, _i, _i, _i, C.c_uint64, C.POINTER(_vp)]),
    ("ddrl_agents_create_nfac", _i, [_cp, _i, _i, _i, _i, _i, C.c_uint64, C.POINTER(_vp)]),
    ("ddrl_agents_create_cmaes", _i, [_cp, _i, _i, _i, C.c_uint64, C.POINTER(_vp)]),
    ("ddrl_agents_cmaes", _vp, [_vp]),
    ("ddrl_agents_cmaes_state", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("ddrl_policy_create", _i, [_i, _i, _i, _vp, _i, _i, _i, _i, C.c_uint64, C.POINTER(_vp)]),
    ("ddrl_pop_load_flat_dev", _i, [_vp, _i, _vp, _i, _vp]),
    ("ddrl_pop_set_gate_layout", _i, [_vp, _vp, _i, _vp, _i, _i]),
    ("ddrl_pop_dev_num_params", _i, [_vp, _i, _i, C.POINTER(C.c_int)]),
    ("ddrl_pop_dev_get_weights", _i, [_vp, _i, _i, _vp]),
    ("ddrl_pop_dev_set_weights", _i, [_vp, _i, _i, _vp]),
    ("ddrl_agents_destroy", _i, [_vp]),
    ("ddrl_agents_population", _vp, [_vp]),
    ("ddrl_agents_start_episode", _i, [_vp, _vp, _vp, _vp]),
    ("ddrl_agents_run", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("ddrl_agents_end_episode", _i, [_vp, _vp, _vp]),
    ("ddrl_agents_stats", _i, [_vp, _vp, _vp, _vp]),
    ("DDPGAg_new", C.c_int64, [C.c_uint, C.c_uint]),
    ("DDPGAg_unique_invoke", None, [C.c_int64, _i, C.POINTER(_cp)]),
    ("DDPGAg_start_episode", None, [C.c_int64, _vp, C.c_bool]),
    ("DDPGAg_run", _vp, [C.c_int64, C.c_double, _vp, C.c_bool, C.c_bool, C.c_bool]),
    ("DDPGAg_end_episode", None, [C.c_int64, C.c_bool]),
    ("DDPGAg_dump", None, [C.c_int64, C.c_bool, _i, _i, C.c_double]),
    ("DDPGAg_display", None, [C.c_int64, C.c_bool, _i, _i, C.c_double]),
    ("DDPGAg_save", None, [C.c_int64, _i]),
    ("DDPGAg_load", None, [C.c_int64, _i]),
    ("DDPGAg_delete", None, [C.c_int64]),
    ("OfflineCaclaAg_new", C.c_int64, [C.c_uint, C.c_uint]),
    ("OfflineCaclaAg_unique_invoke", None, [C.c_int64, _i, C.POINTER(_cp)]),
    ("OfflineCaclaAg_start_episode", None, [C.c_int64, _vp, C.c_bool]),
    ("OfflineCaclaAg_run", _vp, [C.c_int64, C.c_double, _vp, C.c_bool, C.c_bool, C.c_bool]),
    ("OfflineCaclaAg_end_episode", None, [C.c_int64, C.c_bool]),
    ("OfflineCaclaAg_dump", None, [C.c_int64, C.c_bool, _i, _i, C.c_double]),
    ("OfflineCaclaAg_display", None, [C.c_int64, C.c_bool, _i, _i, C.c_double]),
    ("OfflineCaclaAg_save", None, [C.c_int64, _i]),
    ("OfflineCaclaAg_load", None, [C.c_int64, _i]),
    ("OfflineCaclaAg_delete", None, [C.c_int64]),
    ("TD3Ag_new", C.c_int64, [C.c_uint, C.c_uint]),
    ("TD3Ag_unique_invoke", None, [C.c_int64, _i, C.POINTER(_cp)]),
    ("TD3Ag_start_episode", None, [C.c_int64, _vp, C.c_bool]),
    ("TD3Ag_run", _vp, [C.c_int64, C.c_double, _vp, C.c_bool, C.c_bool, C.c_bool]),
    ("TD3Ag_end_episode", None, [C.c_int64, C.c_bool]),
    ("TD3Ag_dump", None, [C.c_int64, C.c_bool, _i, _i, C.c_double]),
    ("TD3Ag_display", None, [C.c_int64, C.c_bool, _i, _i, C.c_double]),
    ("TD3Ag_save", None, [C.c_int64, _i]),
    ("TD3Ag_load", None, [C.c_int64, _i]),
    ("TD3Ag_delete", None, [C.c_int64]),
    ("ddrl_nfac_config_default", _i, [C.POINTER(NfacConfigC)]),
    ("ddrl_nfac_config_from_ini", _i, [_cp, C.POINTER(NfacConfigC)]),
    ("ddrl_nfac_create", _i, [C.POINTER(NfacConfigC), C.c_uint64, C.POINTER(_vp)]),
    ("ddrl_nfac_traj_append", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("ddrl_nfac_traj_load", _i, [_vp, _i, _i, _vp, _i, _vp]),
    ("ddrl_nfac_episode_end", _i, [_vp, _vp]),
    ("ddrl_nfac_traj_len", _i, [_vp, _vp]),
    ("ddrl_nfac_max_trajectory", _i, [_vp]),
    ("ddrl_nfac_update", _i, [_vp, _vp]),
    ("ddrl_nfac_last", _i, [_vp, _vp, _vp, _vp]),
    ("ddrl_nfac_learn_batch_lw", _i, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    ("ddrl_cmaes_create", _i, [_i, _vp, _vp, C.c_long, _i, _i, C.POINTER(_vp)]),
    ("ddrl_cmaes_destroy", _i, [_vp]),
    ("ddrl_cmaes_sample", _i, [_vp]),
    ("ddrl_cmaes_resample", _i, [_vp, _i]),
    ("ddrl_cmaes_make_feasible", _i, [_vp, C.c_double, C.POINTER(_i)]),
    ("ddrl_cmaes_get_population", _i, [_vp, _vp]),
    ("ddrl_cmaes_population_dev", _i, [_vp, C.POINTER(_vp), C.POINTER(_i)]),
    ("ddrl_cmaes_update", _i, [_vp, _vp]),
    ("ddrl_cmaes_update_dev", _i, [_vp, _vp]),
    ("ddrl_cmaes_rank", _i, [_vp, _vp, _i, _vp]),
    ("ddrl_cmaes_rank_noalloc", _i, [_vp, _i, _vp, _vp]),
    ("ddrl_cmaes_get", _i, [_vp, _cp, C.POINTER(C.c_double)]),
    ("ddrl_cmaes_get_vec", _i, [_vp, _cp, _vp]),
    ("ddrl_cmaes_get_index", _i, [_vp, _vp]),
    ("ddrl_cmaes_test_for_termination", _i, [_vp, _cp, _i]),
    ("ddrl_cmaes_set_option", _i, [_vp, _cp, _i]),
]

_lib = None


def build():
    """Compile libddrl_pop.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(HERE, "csrc")])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise LibraryMissing(
            "%s not found: build it with `make -C ddrl_amd/csrc` (or __graft_entry__.build()). "
            "ddrl_amd has no CPU path." % SO)
    L = C.CDLL(SO)
    L.ddrl_version.restype = _cp
    global REAL
    partial = b"+f64" in L.ddrl_version()    # the fp64 debug build (csrc/pop_internal.h): the one-tile DDPG path only,
    REAL = np.float64 if partial else np.float32   # and the ABI's float arrays are double arrays
    for name, res, args in SYMBOLS:
        if (partial or os.environ.get("DDRL_POP_SO")) and not hasattr(L, name):    # (dev experiments: an older build)
            continue
        fn = getattr(L, name)     # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise DdrlError(rc, load().ddrl_last_error().decode(errors="replace"))

"""ctypes binding of libic3rollout.so (include/ic3_rollout.h).  No fallback: import errors are fatal."""
import ctypes as C
import os

# torch must load first: it ships its own libamdhip64 (same soname as /opt/rocm's); loading ours first would
# bind the process to a second HIP runtime that cannot see torch's device allocations.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("IC3_ROLLOUT_LIB") or os.path.join(_HERE, "csrc", "libic3rollout.so")   # (override: A/B builds)

ABI_VERSION = 601        # IC3_VERSION of include/ic3_rollout.h this binding was written against (checked at load)
ENV_PP, ENV_TJ = 1, 2
PP_MODES = {"mixed": 0, "cooperative": 1, "competitive": 2}
TJ_DIFFICULTY = {"easy": 0, "medium": 1, "hard": 2}


class PPCfg(C.Structure):
    _fields_ = [("E", C.c_int32), ("N", C.c_int32), ("nprey", C.c_int32), ("dim", C.c_int32), ("vision", C.c_int32),
                ("mode", C.c_int32), ("stay", C.c_int32), ("moving_prey", C.c_int32), ("enemy_comm", C.c_int32),
                ("seed", C.c_uint32),
                ("env_id_offset", C.c_uint32)]


class TJCfg(C.Structure):
    _fields_ = [("E", C.c_int32), ("N", C.c_int32), ("dim", C.c_int32), ("vision", C.c_int32),
                ("difficulty", C.c_int32), ("vocab_type", C.c_int32), ("add_rate_min", C.c_double),
                ("add_rate_max", C.c_double), ("curr_start", C.c_double), ("curr_end", C.c_double),
                ("seed", C.c_uint32), ("env_id_offset", C.c_uint32)]


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kind", "E", "N", "obs_dim", "vocab", "naction", "window", "npath",
                                         "narrival", "max_route_len", "grid_h", "grid_w", "state_words")]


class Stats(C.Structure):
    _fields_ = [("success_sum", C.c_double), ("add_rate", C.c_double), ("episodes", C.c_int64),
                ("live_env_steps", C.c_int64), ("auto_success_sum", C.c_double), ("auto_episodes", C.c_int64),
                ("auto_env_steps", C.c_int64)]


class Policy(C.Structure):
    """ic3_policy (include/ic3_rollout.h): host struct of device pointers for ic3_policy_step."""
    _fields_ = [("struct_size", C.c_uint32), ("H", C.c_int32), ("nheads", C.c_int32), ("head_sizes", C.c_int32 * 4), ("mode_avg", C.c_int32),
                ("comm_zero", C.c_int32), ("enc_wt", C.c_void_p), ("enc_bias", C.c_void_p), ("loc_table", C.c_void_p),
                ("c_wp", C.c_void_p), ("lstm_wp", C.c_void_p), ("lstm_bias", C.c_void_p), ("head_w", C.c_void_p),
                ("head_b", C.c_void_p), ("pass_index", C.c_int32), ("inner_pass", C.c_int32), ("gate_split", C.c_int32),
                ("npasses", C.c_int32), ("lstm_wp3", C.c_void_p), ("c_wp_pass", C.c_void_p * 4), ("enc_bias_pass", C.c_void_p * 4)]

    def __init__(self, *a, **kw):             # positional arguments start at the field behind struct_size
        super().__init__(*((0,) + a if a else a), **kw)
        self.struct_size = C.sizeof(self)


class Episode(C.Structure):
    """ic3_episode (include/ic3_rollout.h): episode buffers in, masks and reduced statistics out."""
    _fields_ = [("struct_size", C.c_uint32), ("n", C.c_int32), ("E", C.c_int32), ("N", C.c_int32), ("auto_reset", C.c_int32),
                ("forced_last", C.c_int32), ("gate_ones", C.c_int32), ("done", C.c_void_p), ("alive", C.c_void_p),
                ("is_completed", C.c_void_p), ("reward", C.c_void_p), ("gate", C.c_void_p), ("gate_stride", C.c_int64),
                ("live", C.c_void_p), ("alive_mask", C.c_void_p), ("episode_mask", C.c_void_p),
                ("episode_mini_mask", C.c_void_p), ("live_after", C.c_void_p), ("stats", C.c_void_p),
                ("scratch", C.c_void_p), ("counter", C.c_void_p)]

    def __init__(self, *a, **kw):             # positional arguments start at the field behind struct_size
        super().__init__(*((0,) + a if a else a), **kw)
        self.struct_size = C.sizeof(self)


class Bptt(C.Structure):
    """ic3_bptt (include/ic3_rollout.h): one window of the recorded-gates backward through time."""
    _fields_ = [("struct_size", C.c_uint32)] + [(n, C.c_int32) for n in ("T", "E", "N", "H", "OT", "mode_avg", "comm_zero",
                                                                        "detach_gap", "enc_first")] + \
               [("gates", C.c_void_p), ("hs", C.c_void_p), ("cs", C.c_void_p), ("dhead", C.c_void_p), ("snaps", C.c_void_p),
                ("snap_words", C.c_int64), ("alive", C.POINTER(C.c_void_p)), ("gate", C.POINTER(C.c_void_p)),
                ("row_live", C.c_void_p), ("row_keep", C.c_void_p), ("lstm_wp3_bwd", C.c_void_p), ("w_heads", C.c_void_p),
                ("c_weight", C.c_void_p), ("dh", C.c_void_p), ("dc", C.c_void_p), ("dxh", C.c_void_p),
                ("dbias_partials", C.c_void_p), ("dcw_partials", C.c_void_p), ("enc_work", C.c_void_p),
                ("dxh_step", C.c_int64), ("two_chains", C.c_int32), ("gate_events", C.POINTER(C.c_void_p))]


EXPORTS = {
    # name: (restype, argtypes)
    "ic3_version": (C.c_int, []),
    "ic3_abi_check": (C.c_int, [C.c_int, C.c_size_t, C.c_size_t]),
    "ic3_last_error": (C.c_char_p, []),
    "ic3_pp_create": (C.c_int, [C.POINTER(PPCfg), C.c_int, C.POINTER(C.c_void_p)]),
    "ic3_tj_create": (C.c_int, [C.POINTER(TJCfg), C.c_int, C.POINTER(C.c_void_p)]),
    "ic3_env_destroy": (C.c_int, [C.c_void_p]),
    "ic3_env_dims": (C.c_int, [C.c_void_p, C.POINTER(Dims)]),
    "ic3_env_reset": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ic3_env_reset_to": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ic3_env_step": (C.c_int, [C.c_void_p] * 7 + [C.c_void_p]),
    "ic3_env_set_auto_reset": (C.c_int, [C.c_void_p, C.c_int]),
    "ic3_env_observe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ic3_env_observe_at": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ic3_env_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_void_p]),
    "ic3_env_encode_table": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ic3_env_set_incremental_obs": (C.c_int, [C.c_void_p, C.c_int]),
    "ic3_env_encode_at": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_void_p]),
    "ic3_env_snapshot": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ic3_env_encode_backward_work": (C.c_int64, [C.c_void_p, C.c_int]),
    "ic3_env_encode_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    "ic3_env_encode_backward_accumulate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                     C.c_void_p]),
    "ic3_env_encode_backward_finish": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ic3_env_encode_backward_window_work": (C.c_int64, [C.c_void_p, C.c_int]),
    "ic3_env_encode_backward_window": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int,
                                                 C.c_void_p, C.c_int, C.c_void_p]),
    "ic3_env_encode_backward_window_finish": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ic3_env_check": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ic3_env_get_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ic3_env_set_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ic3_env_state_field": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ic3_tj_get_tables": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ic3_tj_build_tables": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(Dims), C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_size_t]),
    "ic3_tj_get_add_rate": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ic3_env_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats), C.c_void_p]),
    "ic3_comm_masked_mean": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p]),
    "ic3_comm_masked_mean_add": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
                                 + [C.c_int] * 5 + [C.c_void_p]),
    "ic3_lstm_cell": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ic3_lstm_cell_backward": (C.c_int, [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_void_p]),
    "ic3_policy_pack_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ic3_policy_pack_split_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ic3_lstm_gates_backward_given": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ic3_comm_backward_partials": (C.c_int, [C.c_int, C.c_int]),
    "ic3_comm_backward": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 7 + [C.c_int] * 6 + [C.c_void_p]),
    "ic3_lstm_weight_grad_scratch_floats": (C.c_size_t, [C.c_longlong, C.c_int]),
    "ic3_lstm_weight_grad": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p,
                                       C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ic3_bptt_first_chain_envs": (C.c_int, [C.c_int, C.c_int]),
    "ic3_bptt_backward_supported": (C.c_int, [C.c_void_p, C.c_int]),
    "ic3_bptt_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ic3_env_set_record_out": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ic3_lstm_gates_backward_dx": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 11 + [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ic3_commnet_forward_supported": (C.c_int, [C.c_int, C.c_int]),
    "ic3_commnet_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ic3_commnet_pack_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ic3_commnet_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6 +
                            [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5),
    "ic3_commnet_step_supported": (C.c_int, [C.c_void_p, C.c_int]),
    "ic3_commnet_step": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int] +
                         [C.c_void_p] * 12),
    "ic3_lstm_gates_backward_supported": (C.c_int, [C.c_int]),
    "ic3_lstm_gates_backward": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 10 + [C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ic3_heads_grad_scratch_floats": (C.c_size_t, [C.c_int]),
    "ic3_heads_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p]),
    "ic3_env_set_hidden_out": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ic3_lstm_cell_heads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ic3_policy_heads": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                   C.c_int, C.c_void_p]),
    "ic3_sample_actions": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ic3_env_sample_actions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    "ic3_policy_pack": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_void_p]),
    "ic3_gate_product_probe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ic3_policy_step_supported": (C.c_int, [C.c_void_p, C.c_int]),
    "ic3_policy_forward": (C.c_int, [C.POINTER(Policy), C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6),
    "ic3_policy_step": (C.c_int, [C.c_void_p, C.POINTER(Policy)] + [C.c_void_p] * 12),
    "ic3_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "ic3_event_destroy": (C.c_int, [C.c_void_p]),
    "ic3_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "ic3_env_set_step_events": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ic3_episode_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "ic3_episode_finalize": (C.c_int, [C.POINTER(Episode), C.c_void_p]),
    "ic3_loss_gradients_partials": (C.c_int, [C.c_longlong, C.c_longlong]),
    "ic3_loss_gradients": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ic3_returns_scan": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p]),
    "ic3_random_actions": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                     C.c_int, C.c_void_p]),
}

_lib = None


class IC3Error(RuntimeError):
    pass


def lib():
    """Load libic3rollout.so (built by __graft_entry__.build() / `make -C ic3net_amd/csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise IC3Error("%s is missing: build it with `make -C ic3net_amd/csrc` (hipcc, gfx950). "
                           "There is no CPU fallback." % SO_PATH)
        l = C.CDLL(SO_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        # the boundary checks itself: same header version, same struct layouts (a stale .so would read garbage otherwise)
        if l.ic3_version() != ABI_VERSION or l.ic3_abi_check(ABI_VERSION, C.sizeof(Policy), C.sizeof(Episode)) != 0:
            raise IC3Error("%s is version %d, this binding needs %d with sizeof(ic3_policy) = %d, sizeof(ic3_episode) = %d: %s"
                           % (SO_PATH, l.ic3_version(), ABI_VERSION, C.sizeof(Policy), C.sizeof(Episode),
                              l.ic3_last_error().decode("utf-8", "replace")))
        _lib = l
    return _lib


def check(rc, exc=IC3Error):
    if rc < 0:
        msg = lib().ic3_last_error().decode("utf-8", "replace")
        if rc == -38:
            raise NotImplementedError(msg)
        if rc == -22 and exc is IC3Error:
            raise ValueError(msg)
        raise exc(msg)
    return rc


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)

"""ctypes binding of libcpnative.so (include/cpnative.h).

There is deliberately no CPU fallback: if the HIP library is missing or a call
fails, an exception is raised (the product path must never silently run anything
but the gfx950 kernels).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CPNATIVE_LIB: another build of the same library (A/B measurements of kernel variants)
LIB_PATH = os.environ.get("CPNATIVE_LIB") or os.path.join(_HERE, "libcpnative.so")

CP_F32, CP_BF16, CP_FP8 = 0, 1, 2
FP8_STATE_BYTES = 1024            # scale table at the start of a CP_FP8 workspace (csrc/fp8.cuh): zero once after allocating
CP_TASKS, CP_EMG_DIM, CP_D_E, CP_N_BN, CP_N_FC = 41, 12, 16, 9, 7

_fp = C.c_void_p


class cp_params(C.Structure):
    _fields_ = [
        ("conv1_w", _fp), ("conv1_b", _fp), ("conv2_w", _fp), ("conv2_b", _fp),
        ("fc_w", _fp * CP_N_FC), ("fc_b", _fp * CP_N_FC),
        ("bn_g", _fp * CP_N_BN), ("bn_b", _fp * CP_N_BN),
        ("last_w", _fp), ("easy_w", _fp), ("easy_b", _fp),
    ]


class cp_bn_buffers(C.Structure):
    _fields_ = [("running_mean", _fp * CP_N_BN), ("running_var", _fp * CP_N_BN)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)     # cp_allreduce_fn

# cp_config.options bits (include/cpnative.h CP_OPT_*): test / measurement switches of ONE call, 0 in production
OPTIONS = {"unfused_bn_bwd": 1, "unpaired_wgrad": 2, "fp8_bridge": 4, "no_small": 8, "fp8_head_f32": 16}
CP_TILES_STATIC, CP_TILES_DYNAMIC = 0, 1


class cp_config(C.Structure):
    """Everything a call depends on besides its tensors: the library keeps no per-process state for the training path."""
    _fields_ = [
        ("n_windows", C.c_int64), ("dtype", C.c_int32), ("adabn", C.c_int32),
        ("training", C.c_int32), ("step_state_lo", C.c_uint32),
        ("dp_emg", C.c_float), ("bn_momentum", C.c_float), ("bn_eps", C.c_float), ("step_state_hi", C.c_uint32),
        ("seed", C.c_uint64), ("step", C.c_uint64),
        ("options", C.c_uint32), ("tile_schedule", C.c_int32),
        ("stats_allreduce", C.c_void_p), ("stats_user", C.c_void_p), ("stats_world", C.c_int32), ("reserved0", C.c_int32),
        ("grad_tap", C.c_void_p), ("grad_tap_bytes", C.c_size_t),
        ("aux_stream", C.c_void_p), ("aux_fork", C.c_void_p), ("aux_join", C.c_void_p),
    ]


class cp_adam_hyper(C.Structure):
    _fields_ = [(n, C.c_float) for n in
                ("lr_emg", "lr_glove", "reg_emg", "reg_glove", "beta1", "beta2", "eps", "grad_scale")]


# every symbol include/cpnative.h declares: (restype, argtypes)
_P = C.POINTER
class cp_glove_params(C.Structure):
    _fields_ = [("w1", C.c_void_p), ("bn_g", C.c_void_p), ("bn_b", C.c_void_p), ("last_w", C.c_void_p),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p)]


SYMBOLS = {
    "cp_version": (C.c_int, []),
    "cp_last_error": (C.c_char_p, []),
    "cp_has_variants": (C.c_int, []),
    "cp_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_float]),
    "cp_gather_groups": (C.c_int, [_fp, C.c_int64, _fp, C.c_int64, _fp, C.c_int64, C.c_int32, _fp, _fp]),
    "cp_gather_oob_count": (C.c_int, [_fp, C.c_int32, _fp]),
    "cp_encoder_forward": (C.c_int, [_P(cp_config), _P(cp_params), _P(cp_bn_buffers), _fp, _fp, C.c_size_t, _fp, _fp]),
    "cp_head": (C.c_int, [_P(cp_config), _P(cp_params), _fp, _fp, C.c_int64, C.c_int32, C.c_int32, _fp, C.c_size_t,
                          _fp, _fp, _fp, _P(cp_params), _fp]),
    "cp_global_negatives_scratch_floats": (C.c_size_t, [C.c_int64]),
    "cp_global_negatives": (C.c_int, [_P(cp_params), _fp, C.c_int64, _fp, _fp, _fp, _fp]),
    "cp_global_negatives_g": (C.c_int, [_P(cp_params), _fp, C.c_int64, _fp, _fp, _fp, _fp]),
    "cp_global_negatives_h": (C.c_int, [C.c_int64, _fp, _fp, _fp, _fp]),
    "cp_head_gneg": (C.c_int, [_P(cp_config), _P(cp_params), _fp, _fp, C.c_int64, C.c_int32, C.c_int32, _fp, C.c_size_t,
                               _fp, _fp, _fp, _P(cp_params), _fp, _fp]),
    "cp_encoder_backward": (C.c_int, [_P(cp_config), _P(cp_params), _fp, _fp, C.c_size_t, _P(cp_params), _fp]),
    "cp_encoder_backward_ev": (C.c_int, [_P(cp_config), _P(cp_params), _fp, _fp, C.c_size_t, _P(cp_params), _fp, _fp]),
    "cp_vote": (C.c_int, [_fp, _fp, C.c_int64, C.c_int32, _fp, _fp, _fp]),
    "cp_subset_vote": (C.c_int, [_fp, _fp, C.c_int64, C.c_int32, _fp, C.c_int64, _fp, _fp, _fp]),
    "cp_confusion": (C.c_int, [_fp, _fp, C.c_int64, _fp, _fp]),
    "cp_preprocess_emg": (C.c_int, [_fp, C.c_int64, C.c_int32, _P(C.c_double), _P(C.c_double), C.c_int32, C.c_int32, C.c_float,
                                    _P(C.c_int32), C.c_int32, _fp, _fp]),
    "cp_emg_stats": (C.c_int, [_fp, C.c_int64, C.c_int32, _fp, C.c_int32, _fp, _fp, _fp]),
    "cp_emg_normalize": (C.c_int, [_fp, C.c_int64, _fp, _fp]),
    "cp_glove_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "cp_glove_forward": (C.c_int, [_P(cp_config), _P(cp_glove_params), _fp, C.c_int64, _fp, C.c_size_t, _fp, _fp]),
    "cp_head_glove": (C.c_int, [_P(cp_config), _fp, _fp, _fp, C.c_int64, C.c_int32, C.c_int32, _fp, C.c_size_t, _fp,
                                C.c_size_t, _fp, _fp, _fp, _fp]),
    "cp_glove_backward": (C.c_int, [_P(cp_config), _P(cp_glove_params), C.c_int64, _fp, C.c_size_t, _P(cp_glove_params), _fp]),
    "cp_optimizer_scratch_floats": (C.c_size_t, [_P(C.c_int64), C.c_int32]),
    "cp_l2_norms": (C.c_int, [_fp, _P(C.c_int64), _P(C.c_int64), _P(C.c_int32), _P(C.c_int32), C.c_int32,
                              _P(cp_adam_hyper), _fp, _fp, _fp]),
    "cp_l2_adam_step": (C.c_int, [_fp, _fp, _fp, _fp, _P(C.c_int64), _P(C.c_int64), _P(C.c_int32), _P(C.c_int32),
                                  C.c_int32, _P(cp_adam_hyper), C.c_int64, _fp, _fp, _fp]),
    "cp_l2_adam_step_graph": (C.c_int, [_fp, _fp, _fp, _fp, _P(C.c_int64), _P(C.c_int64), _P(C.c_int32), _P(C.c_int32),
                                        C.c_int32, _P(cp_adam_hyper), _fp, _fp, _fp, _fp]),
    "cp_debug_hog": (C.c_int, [C.c_int32, C.c_int32, _fp]),
    "cp_profile_enable": (C.c_int, [C.c_uint64, C.c_int32]),
    "cp_profile_disable": (C.c_int, []),
    "cp_profile_resume": (C.c_int, []),
    "cp_profile_summary": (C.c_int, [C.c_int32, _P(C.c_double), _P(C.c_int64)]),
    "cp_debug_activation": (C.c_int, [_P(cp_config), _P(cp_params), _fp, _fp, C.c_size_t, C.c_int32, _fp, _fp]),
    "cp_debug_bn_stats": (C.c_int, [_P(cp_config), _fp, C.c_size_t, C.c_int32, _fp, _fp]),
    "cp_debug_gemm": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _fp, _fp, _fp, _fp, _fp, _fp,
                                C.c_int32, _fp]),
}

KERNEL_KINDS = ["gather", "prep", "conv1_fwd", "bn_finalize", "conv2_fwd", "fold", "fc_fwd", "dropout", "proj_fwd",
                "head", "proj_bwd", "bn_bwd", "fc_wgrad", "reduce_slabs", "fc_dgrad", "conv2_wgrad", "conv2_dgrad",
                "conv1_bwd", "optimizer", "fc_dgrad_stats", "fc_dgrad_bn", "fc_fwd_ws", "fc_dgrad_conv"]

_lib = None


class CpNativeError(RuntimeError):
    pass


def load():
    """Load libcpnative.so (built by ``__graft_entry__.build()`` / ``csrc/Makefile``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CpNativeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # torch first: its wheel carries its own HIP runtime (torch/lib/libamdhip64.so), and the process must end up with ONE runtime -- the
    # one that owns torch's allocations and streams.  With libcpnative.so loaded first the dynamic loader binds it to /opt/rocm's copy,
    # torch then brings its own, and every launch of this library fails with hipErrorNoDevice on the box (seen: build() and smoke() in
    # one process).  Loaded behind torch, the library resolves against the runtime that is already there.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.cp_version() < 110:
        raise CpNativeError("libcpnative.so is older than this binding")
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        msg = load().cp_last_error().decode("utf-8", "replace")
        raise CpNativeError(f"{what} failed: {msg}")

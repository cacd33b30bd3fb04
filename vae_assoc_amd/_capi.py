"""ctypes binding of libavae.so (C ABI declared in include/avae.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded, importing a
symbol from here raises.  ``import torch`` happens first on purpose: the PyTorch-ROCm wheel
bundles its own libamdhip64.so.7, and loading it first makes libavae (linked against the same
soname) resolve to that single HIP runtime, so device pointers of torch tensors are valid
inside the library.
"""
import ctypes as C
import os
import threading

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

AVAE_ABI_VERSION = 4
AVAE_MAX_MODALITIES = 4
AVAE_MAX_HIDDEN = 8
AVAE_MAX_WORLD = 8
AVAE_IPC_HANDLE_BYTES = 128
COMM_NONE, COMM_RCCL, COMM_IPC = 0, 1, 2

ACT_IDS = {"identity": 0, "relu": 1, "softplus": 2, "sigmoid": 3, "tanh": 4}
DTYPE_IDS = {"fp32": 0, "f32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libavae.so")

# every symbol include/avae.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "avae_workspace_bytes", "avae_create", "avae_destroy", "avae_last_error", "avae_param_count",
    "avae_get_params", "avae_set_params", "avae_get_grads", "avae_get_opt_state", "avae_set_opt_state",
    "avae_train_step", "avae_train_steps", "avae_stage_batches", "avae_grad_buffer", "avae_cost_history",
    "avae_dp_plan", "avae_dp_backward", "avae_dp_apply", "avae_comm_unique_id", "avae_comm_ipc_handle", "avae_comm_ipc_attach",
    "avae_eval_cost", "avae_encode", "avae_decode", "avae_generate", "avae_reconstruct", "avae_save", "avae_load",
    "avae_synchronize", "avae_timing_enable", "avae_timing_report", "avae_debug_fetch", "avae_comm_allreduce",
]


class Modality(C.Structure):
    _fields_ = [("n_input", C.c_int32), ("n_hidden_layers", C.c_int32),
                ("n_hidden", C.c_int32 * AVAE_MAX_HIDDEN), ("binary", C.c_int32),
                ("weight", C.c_float), ("hidden_conv", C.c_int32), ("conv_gener", C.c_int32 * 2),
                ("reserved", C.c_int32)]


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("n_modalities", C.c_int32),
                ("mod", Modality * AVAE_MAX_MODALITIES),
                ("n_z", C.c_int32), ("batch_size", C.c_int32), ("batch_global", C.c_int32),
                ("row_offset", C.c_int32), ("activation", C.c_int32), ("compute_dtype", C.c_int32),
                ("device", C.c_int32), ("use_graph", C.c_int32),
                ("assoc_lambda", C.c_float), ("learning_rate", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
                ("seed", C.c_uint64), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("use_comm", C.c_int32), ("world_size", C.c_int32), ("rank", C.c_int32), ("comm_buckets", C.c_int32),
                ("nccl_id", C.c_uint8 * 128), ("wire_dtype", C.c_int32), ("reserved2", C.c_int32 * 3)]


_lib = None
_lock = threading.Lock()


def lib():
    """The loaded library; raises (never falls back) when it is absent."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    "libavae.so is missing (%s): build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "-- vae_assoc_amd has no CPU fallback" % LIB_PATH)
            L = C.CDLL(LIB_PATH)
            fp, vp, i32, sz = C.POINTER(C.c_float), C.c_void_p, C.c_int32, C.c_size_t
            L.avae_workspace_bytes.argtypes = [C.POINTER(Config), C.POINTER(sz)]
            L.avae_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
            L.avae_destroy.argtypes = [vp]
            L.avae_destroy.restype = None
            L.avae_last_error.argtypes = [vp]
            L.avae_last_error.restype = C.c_char_p
            L.avae_param_count.argtypes = [vp, C.POINTER(sz)]
            L.avae_get_params.argtypes = [vp, vp]
            L.avae_set_params.argtypes = [vp, vp]
            L.avae_get_grads.argtypes = [vp, vp]
            L.avae_get_opt_state.argtypes = [vp, vp, vp, C.POINTER(C.c_int64)]
            L.avae_set_opt_state.argtypes = [vp, vp, vp, C.c_int64]
            L.avae_train_step.argtypes = [vp, C.POINTER(vp), C.POINTER(i32), vp, fp, vp]
            L.avae_train_steps.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i32), vp, fp, vp]
            L.avae_stage_batches.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i32), vp, vp]
            L.avae_grad_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
            L.avae_cost_history.argtypes = [vp, i32, vp, C.POINTER(C.c_int64)]
            L.avae_dp_plan.argtypes = [C.POINTER(Config), C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
            L.avae_dp_backward.argtypes = [vp, i32, i32, vp]
            L.avae_dp_apply.argtypes = [vp, i32, fp, vp]
            L.avae_comm_unique_id.argtypes = [vp]
            L.avae_comm_ipc_handle.argtypes = [vp, vp]
            L.avae_comm_ipc_attach.argtypes = [vp, vp]
            L.avae_eval_cost.argtypes = [vp, C.POINTER(vp), C.POINTER(i32), vp, fp, vp]
            L.avae_encode.argtypes = [vp, i32, vp, i32, i32, vp, vp, vp]
            L.avae_decode.argtypes = [vp, i32, vp, i32, vp, vp]
            L.avae_generate.argtypes = [vp, vp, i32, C.POINTER(vp), vp]
            L.avae_reconstruct.argtypes = [vp, i32, vp, i32, vp, i32, vp, vp]
            L.avae_save.argtypes = [vp, C.c_char_p]
            L.avae_load.argtypes = [vp, C.c_char_p]
            L.avae_synchronize.argtypes = [vp]
            L.avae_timing_enable.argtypes = [vp, i32]
            L.avae_timing_report.argtypes = [vp, C.c_char_p, sz]
            L.avae_debug_fetch.argtypes = [vp, C.c_char_p, vp, sz, C.POINTER(sz)]
            L.avae_comm_allreduce.argtypes = [vp, i32, vp]
            for name in SYMBOLS:
                if name not in ("avae_destroy", "avae_last_error"):
                    getattr(L, name).restype = C.c_int
            _lib = L
        return _lib


def check(handle, rc, what):
    if rc != 0:
        msg = lib().avae_last_error(handle)
        raise RuntimeError("%s failed (%d): %s" % (what, rc, msg.decode("utf-8", "replace") if msg else "?"))

"""ctypes binding of libwekws_hip.so (include/wekws_hip.h).  This is the ONLY compute path of the package:
there is no CPU / PyTorch fallback, and a missing or stale library is a hard error."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_LIB_PATH = os.environ.get("WEKWS_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                      "libwekws_hip.so")  # override: kernel experiments only
ABI_VERSION = 2


class HipLibraryError(RuntimeError):
    pass


class Desc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "abi_version", "backbone", "idim", "hdim", "odim", "num_layers", "num_stack", "stack_size", "kernel_size",
        "preproc_relu", "head", "head_hidden", "activation", "precision", "aux0", "aux1")]


class FbankCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_bins", "sample_rate", "frame_length", "frame_shift", "window")] + \
               [("reserved", C.c_int32 * 3)]


# name -> (restype, argtypes); the CPU test-suite checks every symbol of the header is exported
SIGNATURES = {
    "wekws_hip_last_error": (C.c_char_p, []),
    "wekws_hip_abi_version": (C.c_int, []),
    "wekws_hip_blob_elems": (C.c_size_t, [C.POINTER(Desc)]),
    "wekws_hip_create": (C.c_int, [C.POINTER(Desc), C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "wekws_hip_destroy": (None, [C.c_void_p]),
    "wekws_hip_cache_dim": (C.c_int, [C.c_void_p]),
    "wekws_hip_cache_len": (C.c_int, [C.c_void_p]),
    "wekws_hip_cache_elems": (C.c_size_t, [C.c_void_p, C.c_int]),
    "wekws_hip_output_elems": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "wekws_hip_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "wekws_hip_effective_precision": (C.c_int, [C.c_void_p]),
    "wekws_hip_weight_spread_log2": (C.c_float, [C.c_void_p]),
    "wekws_hip_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "wekws_hip_reserve": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "wekws_hip_release": (C.c_int, [C.c_void_p, C.c_void_p]),
    "wekws_hip_forward_status": (C.c_int, [C.c_void_p, C.c_void_p]),
    "wekws_hip_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_void_p]),
    "wekws_hip_fbank_create": (C.c_int, [C.POINTER(FbankCfg), C.c_int, C.POINTER(C.c_void_p)]),
    "wekws_hip_fbank_destroy": (None, [C.c_void_p]),
    "wekws_hip_fbank_num_frames": (C.c_int, [C.c_void_p, C.c_int]),
    "wekws_hip_fbank_compute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "wekws_hip_fbank_compute_i16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "wekws_hip_splice_frames": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "wekws_hip_dct_lifter": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "wekws_hip_softmax_topk": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "wekws_hip_score_maxpool": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "wekws_hip_det_false_alarms": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_int, C.c_void_p, C.c_void_p]),
    "wekws_hip_det_false_alarms_text": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_int, C.c_void_p, C.c_void_p]),
    "wekws_hip_splice": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p]),
}

OPTIONS = {"w16": 0, "mdtc16": 1, "stream": 2, "mm": 3, "head_slices": 4, "g16": 5, "envelope": 6, "gru_pipe": 7}   # enum wekws_hip_option

_lib: Optional[C.CDLL] = None


def lib_path() -> str:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load (once) and type the library.  Raises HipLibraryError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise HipLibraryError(
            f"{_LIB_PATH} is missing: build it with `make -C wekws_amd/csrc` (or __graft_entry__.build()). "
            "wekws_amd has no CPU fallback.")
    try:
        lib = C.CDLL(_LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 not found
        raise HipLibraryError(f"cannot load {_LIB_PATH}: {e}") from e
    # the version first: a stale library then fails with the ABI message, not with a missing-symbol error of a newer entry
    try:
        lib.wekws_hip_abi_version.restype = C.c_int
        lib.wekws_hip_abi_version.argtypes = []
        have = lib.wekws_hip_abi_version()
    except AttributeError as e:
        raise HipLibraryError(f"{_LIB_PATH} does not export wekws_hip_abi_version; rebuild it") from e
    if have != ABI_VERSION:
        raise HipLibraryError(f"ABI version mismatch: library {have}, binding {ABI_VERSION}; rebuild with `make -C wekws_amd/csrc`")
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{_LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().wekws_hip_last_error().decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise HipLibraryError(f"{what} failed (code {rc}): {last_error()}")


def make_desc(fields: dict) -> Desc:
    d = Desc()
    for n, _ in Desc._fields_:
        setattr(d, n, int(fields.get(n, 0)))
    return d

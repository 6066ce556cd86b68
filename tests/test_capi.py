"""CPU: the C-ABI library loads, exports every symbol include/wekws_hip.h declares, and its host-side
validation (no compute, no GPU needed) behaves: descriptor checks, blob sizing, error strings."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from wekws_amd import _capi, pack
from wekws_amd.utils import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "wekws_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wekws_hip_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _capi.load()
    syms = header_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/wekws_hip.h but not exported"
        assert s in _capi.SIGNATURES, f"{s} has no ctypes signature in wekws_amd/_capi.py"
    assert sorted(_capi.SIGNATURES) == syms


def test_every_entry_point_is_documented_for_integrators():
    """INTEGRATION.md's table names, for every C entry point, the reference interface it replaces (or says that it has
    none): a symbol added to the header without a row there is a gap in the drop-in story."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    names = set(re.findall(r"wekws_hip_[a-z_0-9]+", doc))
    # the table abbreviates families as `wekws_hip_fbank_create` / `_compute` / `_compute_i16`: expand the short forms
    for fam, rest in re.findall(r"`(wekws_hip_[a-z0-9]+(?:_[a-z0-9]+)*)`((?:\s*/\s*`_[a-z_0-9]+`)+)", doc):
        stem = fam.rsplit("_", 1)[0]
        for short in re.findall(r"`(_[a-z_0-9]+)`", rest):
            names.add(stem + short)
            names.add("wekws_hip" + short)
    missing = [s for s in header_symbols() if s not in names]
    assert not missing, f"not mentioned in INTEGRATION.md: {missing}"


def test_abi_version_and_desc_layout():
    lib = _capi.load()
    assert lib.wekws_hip_abi_version() == _capi.ABI_VERSION == pack.ABI_VERSION
    assert C.sizeof(_capi.Desc) == 16 * 4 and C.sizeof(_capi.FbankCfg) == 8 * 4
    src = open(os.path.join(ROOT, "include", "wekws_hip.h")).read()
    assert f"#define WEKWS_HIP_ABI_VERSION {_capi.ABI_VERSION}" in src


@pytest.mark.parametrize("name", sorted(synth.MODEL_CONFIGS))
def test_blob_elems_matches_packer(name):
    lib = _capi.load()
    cfg = synth.MODEL_CONFIGS[name]
    sd = synth.synth_state_dict(pack.model_spec(cfg), 1)
    desc, blob = pack.pack(cfg, sd)
    d = _capi.make_desc(desc)
    assert lib.wekws_hip_blob_elems(C.byref(d)) == blob.size == pack.blob_elems(desc)
    assert blob.dtype == np.float32 and np.isfinite(blob).all()


def test_invalid_descriptors_are_rejected_with_a_message():
    lib = _capi.load()
    cfg = synth.MODEL_CONFIGS["ds_tcn_h64"]
    desc, blob = pack.pack(cfg, synth.synth_state_dict(pack.model_spec(cfg), 1))
    for field, bad in (("abi_version", 99), ("backbone", 7), ("head", 9), ("hdim", 0), ("num_layers", 0)):
        d = _capi.make_desc({**desc, field: bad})
        assert lib.wekws_hip_blob_elems(C.byref(d)) == 0
        assert _capi.last_error() != ""
    h = C.c_void_p()
    d = _capi.make_desc(desc)
    # wrong blob length -> EINVAL before any device work
    assert lib.wekws_hip_create(C.byref(d), blob.ctypes.data, blob.size - 1, 0, C.byref(h)) == -1
    assert "floats" in _capi.last_error() and not h
    # Shapes beyond every specialised kernel (wider, longer kernels; refused with EUNSUPPORTED until round 4) now run on the
    # any-shape path (csrc/generic.hip.h): create must get as far as the device -- here, without a GPU, that is EDEVICE, never
    # EUNSUPPORTED / EINVAL; with one it succeeds (parity: tests/test_hip_generic.py).  (96, 9) / (96, 12) were the round-3
    # advisor finding: an odd width AND a kernel size above the built one once slipped into the zero-padding path.
    import torch
    mdtc = synth.MODEL_CONFIGS["mdtc_h64"]
    for cfg2 in (dict(cfg, hidden_dim=320),
                 dict(cfg, backbone=dict(cfg["backbone"], kernel_size=9)),
                 dict(cfg, hidden_dim=96, backbone=dict(cfg["backbone"], kernel_size=9)),
                 dict(cfg, hidden_dim=96, backbone=dict(cfg["backbone"], kernel_size=12)),
                 dict(mdtc, hidden_dim=48, backbone=dict(mdtc["backbone"], hidden_dim=48, kernel_size=7)),
                 dict(mdtc, hidden_dim=160, backbone=dict(mdtc["backbone"], hidden_dim=160))):
        desc2, blob2 = pack.pack(cfg2, synth.synth_state_dict(pack.model_spec(cfg2), 1))
        d2 = _capi.make_desc(desc2)
        h2 = C.c_void_p()
        rc = lib.wekws_hip_create(C.byref(d2), blob2.ctypes.data, blob2.size, 0, C.byref(h2))
        if torch.cuda.is_available():
            assert rc == 0 and h2, _capi.last_error()
            assert lib.wekws_hip_effective_precision(h2) == pack.PRECISION["f32"]
            lib.wekws_hip_destroy(h2)
        else:
            assert rc == -3 and not h2, (rc, _capi.last_error())
    # NULL arguments
    assert lib.wekws_hip_create(None, blob.ctypes.data, blob.size, 0, C.byref(h)) == -1
    assert lib.wekws_hip_forward(None, None, 1, 1, None, None, None, 0, None) == -1
    assert lib.wekws_hip_cache_elems(None, 4) == 0 and lib.wekws_hip_output_elems(None, 1, 1) == 0


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _capi.load()
    cfg = synth.MODEL_CONFIGS["ds_tcn_h64"]
    desc, blob = pack.pack(cfg, synth.synth_state_dict(pack.model_spec(cfg), 1))
    h = C.c_void_p()
    d = _capi.make_desc(desc)
    rc = lib.wekws_hip_create(C.byref(d), blob.ctypes.data, blob.size, 0, C.byref(h))
    assert rc in (-3, -2) and not h and _capi.last_error() != ""
    f = C.c_void_p()
    fc = _capi.FbankCfg()
    fc.num_bins, fc.sample_rate, fc.frame_length, fc.frame_shift, fc.window = 40, 16000, 400, 160, 0
    assert lib.wekws_hip_fbank_create(C.byref(fc), 0, C.byref(f)) == -3 and not f
    fc.num_bins = 0
    assert lib.wekws_hip_fbank_create(C.byref(fc), 0, C.byref(f)) == -1


def test_handle_queries_reject_null():
    """Entry points that take a model handle say EINVAL for NULL instead of dereferencing it (no GPU needed)."""
    lib = _capi.load()
    assert lib.wekws_hip_forward_status(None, None) == -1 and "NULL" in _capi.last_error()
    assert lib.wekws_hip_release(None, None) == -1
    assert lib.wekws_hip_reserve(None, 1, 1, None) == -1
    assert lib.wekws_hip_effective_precision(None) == -1


def test_model_refuses_cpu_tensors_and_missing_library(monkeypatch):
    import torch
    from wekws_amd.model.kws_model import init_model
    m = init_model(synth.MODEL_CONFIGS["mdtc_small"])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 5, 40))
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "_LIB_PATH", "/nonexistent/libwekws_hip.so")
    with pytest.raises(_capi.HipLibraryError, match="no CPU fallback"):
        _capi.load()

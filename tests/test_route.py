"""CPU: the routing layer (wekws_amd/csrc/route.h) -- which shape a conv model runs as (as it is / zero-padded / any-shape path) and
which kernel family serves a call -- swept WITHOUT a GPU through the hooks library's wekws_hip_debug_conv_route.  All three defects
the round-5 fuzz found lived in this layer and needed a GPU to show; their configurations are explicit cases here, and the fuzz
generator's configurations are swept against the invariants of every choice.  (The GPU side -- that wekws_hip_forward takes exactly
the family this function names: it switches on its result; every family is parity-green in tests/test_hip_parity.py.)"""
import copy
import ctypes as C
import os

import numpy as np
import pytest

from tests.helpers import random_model_config
from wekws_amd import _capi, pack
from wekws_amd.utils import synth

FAMILIES = ["none", "ds256_stream", "ds256_g32", "ds256_mm", "ds256_g16", "ds256_w16", "ds64_g4", "mdtc64_stream", "mdtc64_g4",
            "mdtc64_w16", "mdtc32_g4", "dense_stack_f16", "conv_stack_f16", "conv_stack"]
KEYS = ("plan", "C", "ks", "family", "nt", "split", "ctx", "fast", "grid", "threads", "lds", "utts_per_wg", "cache_len", "max_pad")


@pytest.fixture(scope="module")
def hooks():
    path = os.path.join(os.path.dirname(_capi.lib_path()), "libwekws_hip_hooks.so")
    lib = C.CDLL(path)
    lib.wekws_hip_debug_conv_route.restype = C.c_int
    lib.wekws_hip_debug_conv_route.argtypes = [C.POINTER(_capi.Desc), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                               C.c_char_p, C.c_int]
    return lib


def route(lib, cfg, B, T, has_in=False, has_out=True, precision="default", x16=1, cache16=1, cus=256, ntiles=1, opts=None):
    cfg = dict(cfg)
    cfg["_precision"] = precision
    d = _capi.make_desc(dict(pack.parse_config(cfg), abi_version=_capi.ABI_VERSION))
    call = (C.c_int * 8)(B, T, ntiles, int(has_in), int(has_out), x16, cache16, cus)
    out = (C.c_int * 14)()
    why = C.create_string_buffer(256)
    o = (C.c_int * 9)(*opts) if opts is not None else None
    assert lib.wekws_hip_debug_conv_route(C.byref(d), o, call, out, why, 256) == 0
    r = dict(zip(KEYS, list(out)))
    r["family"] = FAMILIES[r["family"]]
    r["plan"] = ["as_is", "padded", "generic"][r["plan"]]
    r["why"] = why.value.decode()
    return r


M = synth.MODEL_CONFIGS


@pytest.mark.parametrize("name,kw,family,nt", [
    ("ds_tcn_h256", dict(B=1024, T=98), "ds256_g16", 7),                                   # the headline
    ("ds_tcn_h256", dict(B=1024, T=98, precision="f32"), "ds256_g32", 7),
    ("ds_tcn_h256", dict(B=3, T=80, has_in=True), "ds256_g16", 7),                         # later chunk: the context variant
    ("ds_tcn_h256", dict(B=3, T=20, has_in=True), "ds256_g16", 4),                         # (the context tile is one 16-lane row: >= 4 tiles)
    ("ds_tcn_h256", dict(B=4096, T=10, has_in=True), "ds256_stream", 1),
    ("ds_tcn_h256", dict(B=1, T=10, has_in=False, has_out=True), "ds256_stream", 1),        # first chunk of a stream
    ("ds_tcn_h256", dict(B=1, T=10, has_in=True, cache16=0), "ds256_w16", 1),              # unaligned cache: no 16-byte moves
    ("ds_tcn_h256", dict(B=2, T=98, x16=0), "ds256_g16", 7),                               # unaligned features: its general instantiation
    ("ds_tcn_h256_ctc300", dict(B=2, T=40), "ds256_mm", 4),
    ("ds_tcn_h64", dict(B=1024, T=98), "ds64_g4", 7),
    ("ds_tcn_h64", dict(B=1, T=80, has_in=True), "ds64_g4", 7),
    ("ds_tcn_h64", dict(B=1, T=10, has_in=True), "conv_stack_f16", 1),
    ("ds_tcn_h64", dict(B=8, T=98, precision="f32"), "conv_stack", 7),
    ("tcn_h64", dict(B=8, T=98), "dense_stack_f16", 7),
    ("mdtc_h64", dict(B=1024, T=98), "mdtc64_g4", 7),
    ("mdtc_h64", dict(B=1024, T=10, has_in=True), "mdtc64_stream", 1),
    ("mdtc_h64", dict(B=1, T=98, has_in=True), "mdtc64_w16", 7),                           # one or two streams at 65 .. 112 frames
    ("mdtc_h64", dict(B=3, T=98, has_in=True), "mdtc64_g4", 7),
    ("mdtc_h64_global12", dict(B=8, T=98), "mdtc64_g4", 7),
    ("mdtc_h64_global12", dict(B=8, T=98, has_in=True), "mdtc64_w16", 7),                  # pooled heads with a cache: the LDS-tile kernel
    ("mdtc_small", dict(B=1024, T=98), "mdtc32_g4", 7),
    ("mdtc_small", dict(B=8, T=10, has_in=True), "conv_stack_f16", 1),
])
def test_recipes_take_the_kernel_they_were_built_for(hooks, name, kw, family, nt):
    r = route(hooks, M[name], **kw)
    assert (r["plan"], r["family"], r["nt"]) == ("as_is", family, nt), r


def test_round5_defects_are_visible_without_a_gpu(hooks):
    # (1) DS-TCN h256 with a FIFTH block (dilation 16, padding 112): ds256_w16's hand-over wrote one 64-column pass.  The shape is
    #     outside the register-resident / streaming kernels (dilations 1 / 2 / 4 / 8) and must land on the kernel that walks any padding
    cfg = copy.deepcopy(M["ds_tcn_h256"])
    cfg["backbone"]["num_layers"] = 5
    for kw in (dict(T=98), dict(T=10, has_in=True), dict(T=80, has_in=True)):
        r = route(hooks, cfg, B=3, **kw)
        assert r["family"] == "ds256_w16" and r["max_pad"] == 112, r
    # (2) DS-TCN / TCN with hidden_dim 32 were taken for a built width (32 is built for MDTC only): they run as 64
    for ds in (True, False):
        cfg = copy.deepcopy(M["ds_tcn_h64"])
        cfg["hidden_dim"] = 32
        cfg["backbone"]["ds"] = ds
        r = route(hooks, cfg, B=3, T=50)
        assert (r["plan"], r["C"]) == ("padded", 64) and r["family"] != "none", r
    cfg = copy.deepcopy(M["mdtc_small"])
    assert route(hooks, cfg, B=3, T=50)["plan"] == "as_is"
    # (3) (the empty mel filter was a front-end check: tests/test_hip_fbank.py)


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_configurations_keep_the_invariants_of_every_choice(hooks, seed):
    rng = np.random.default_rng([0x207E, seed])
    seen = set()
    for _ in range(120):
        cfg, head = random_model_config(rng)
        if cfg["backbone"]["type"] == "gru":
            continue
        B = int(rng.choice([1, 2, 3, 9, 260, 1024, 5000]))
        T = int(rng.integers(1, 113))
        has_in = bool(rng.integers(0, 2))
        precision = str(rng.choice(["default", "f32", "f16"]))
        kw = dict(B=B, T=T, has_in=has_in, has_out=bool(rng.integers(0, 2)) or has_in, precision=precision, x16=int(rng.integers(0, 4) > 0),
                  cache16=int(rng.integers(0, 4) > 0), cus=256)
        r = route(hooks, cfg, **kw)
        what = (cfg, kw, r)
        C0, ks0 = cfg["hidden_dim"], cfg["backbone"]["kernel_size"]
        mdtc = cfg["backbone"]["type"] == "mdtc"
        if r["plan"] == "generic":
            assert r["why"], what
            continue
        # the shape the kernels run: a built width that holds the model's, the built kernel size
        assert r["C"] in ((32, 64, 128) if mdtc else (64, 128, 256)) and r["C"] >= C0 and r["ks"] == (5 if mdtc else 8) and ks0 <= r["ks"], what
        assert (r["plan"] == "padded") == (r["C"] != C0 or r["ks"] != ks0), what
        # every shape wekws_hip_create takes has a kernel for every call
        assert r["family"] != "none", what
        assert r["nt"] in (1, 2, 4, 7) and 16 * r["nt"] >= T and 0 <= r["lds"] <= 160 * 1024 and r["threads"] in (128, 256, 512, 1024), what
        assert r["grid"] >= 1 and r["grid"] * r["utts_per_wg"] >= min(B, r["grid"] * r["utts_per_wg"]), what
        if r["grid"] * r["utts_per_wg"] < B:                     # fewer workgroups than utterances: a persistent kernel
            assert r["family"] in ("ds256_g16", "ds256_g32") and r["fast"] and r["grid"] == 256, what
        if r["ctx"]:
            assert has_in or r["family"].endswith("stream"), what
        if r["family"].endswith("stream"):
            assert T <= 16 and (has_in or kw["has_out"]) and kw["cache16"] and precision != "f32", what
        if r["family"] in ("ds256_g16", "ds256_g32", "ds64_g4", "mdtc64_g4", "mdtc32_g4", "ds256_stream", "mdtc64_stream"):
            assert r["max_pad"] <= (r["ks"] - 1) * 8, what        # register-resident / streaming kernels: dilations 1 / 2 / 4 / 8
        if r["family"] in ("ds256_g32", "conv_stack"):
            assert precision == "f32", what
        if precision == "f32":
            assert r["family"] in ("ds256_g32", "conv_stack"), what
        if has_in and r["family"] in ("ds256_g16", "ds64_g4", "mdtc64_g4", "mdtc32_g4"):
            assert r["ctx"] and r["nt"] >= 4 and head == "linear", what
        seen.add(r["family"])
    assert len(seen) >= 4

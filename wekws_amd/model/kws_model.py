"""MI355X drop-in for ``wekws.model.kws_model`` (reference: wekws/model/kws_model.py:33-214).

Same surface as the reference:
    model = init_model(configs['model'])          # kws_model.py:97-214, same config dict
    load_checkpoint(model, path)                  # wekws/utils/checkpoint.py:23-36 works unchanged: the
                                                  # module exposes the reference's state_dict key names
    model = model.to('cuda').eval()
    y, out_cache = model(feats, in_cache)         # kws_model.py:65-76
    y, out_cache = model.forward_softmax(...)     # kws_model.py:78-90
but ``forward`` does not run torch ops: it folds + packs the weights once (wekws_amd/pack.py), hands them
to libwekws_hip.so and launches the fused gfx950 kernels on the caller's current HIP stream.  PyTorch is
used for tensor allocation / stream interop only.  There is deliberately NO CPU fallback: a CPU tensor, a
missing library or an unsupported configuration raises.
"""
from __future__ import annotations

import ctypes
import warnings
import math
import sys
from typing import Mapping, Optional, Tuple

import numpy as np
import torch
from torch import nn

from wekws_amd import _capi, pack

__all__ = ["KWSModel", "init_model"]

_BUFFER_LEAVES = ("running_mean", "running_var", "num_batches_tracked", "mean", "istd")


def _default_init(name: str, shape: tuple) -> torch.Tensor:
    """PyTorch-default-like initial values so a freshly built model is usable (training is out of scope)."""
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.long)
    if leaf in ("running_mean", "mean"):
        return torch.zeros(shape)
    if leaf in ("running_var", "istd"):
        return torch.ones(shape)
    if leaf == "weight" and len(shape) == 1:  # BatchNorm gamma
        return torch.ones(shape)
    if leaf == "bias" and (".bn" in name or name.split(".")[-2] in ("1", "4") and "cnn" in name):
        return torch.zeros(shape)
    if leaf.startswith(("weight_ih", "weight_hh", "bias_ih", "bias_hh")):
        bound = 1.0 / math.sqrt(shape[0] / 3.0)
    elif leaf == "weight":
        bound = 1.0 / math.sqrt(float(np.prod(shape[1:])))
    else:  # bias of a Linear / Conv: bound from the sibling weight is not known here; use a small range
        bound = 0.05
    return (torch.rand(shape) * 2.0 - 1.0) * bound


class _Holder(nn.Module):
    """Pure parameter container: gives nested names like ``backbone.network.0.cnn.3.weight``."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container; the model runs through libwekws_hip.so")


class _HipHandle:
    """Owns one wekws_hip_model*; destroyed with the Python object."""

    def __init__(self, desc_fields: dict, blob: np.ndarray, device_index: int, options: Optional[dict] = None):
        lib = _capi.load()
        self._lib = lib
        self.ptr = ctypes.c_void_p()
        desc = _capi.make_desc(desc_fields)
        need = lib.wekws_hip_blob_elems(ctypes.byref(desc))
        if need != blob.size:
            raise _capi.HipLibraryError(f"packer produced {blob.size} floats, library expects {need}: "
                                        f"{_capi.last_error()}")
        _capi.check(lib.wekws_hip_create(ctypes.byref(desc), blob.ctypes.data, blob.size, device_index,
                                         ctypes.byref(self.ptr)), "wekws_hip_create")
        for name, value in (options or {}).items():
            _capi.check(lib.wekws_hip_set_option(self.ptr, _capi.OPTIONS[name], int(value)), "wekws_hip_set_option")
        # A DEFAULT / F16X3 request whose weights leave the split-fp16 envelope is served by the exact-f32 kernels (several
        # times slower for the conv backbones): say so once instead of leaving it to effective_precision() to be asked.
        req = int(desc_fields.get("precision", 0))
        if req in (pack.PRECISION["default"], pack.PRECISION["f16x3"]) and \
                lib.wekws_hip_effective_precision(self.ptr) == pack.PRECISION["f32"]:
            spread = float(lib.wekws_hip_weight_spread_log2(self.ptr))
            if spread > 20:
                warnings.warn(f"wekws_amd: the row / column magnitudes of a weight matrix spread over 2^{spread:.1f} "
                              f"(> 2^20): this model runs the exact-f32 kernels, not the split-fp16 ones "
                              f"(wekws_hip_effective_precision; set_option('envelope', 0) keeps the fast kernels at "
                              f"reduced accuracy in the small rows)", RuntimeWarning, stacklevel=3)

    def __del__(self):
        try:
            if self.ptr:
                self._lib.wekws_hip_destroy(self.ptr)
                self.ptr = ctypes.c_void_p()
        except Exception:
            pass


class KWSModel(nn.Module):
    """Reference: wekws/model/kws_model.py:33-95.  idim/odim/hdim attributes as there."""

    def __init__(self, configs: Mapping):
        super().__init__()
        self._cfg = dict(configs)
        self._d = pack.parse_config(configs)
        self.idim, self.odim, self.hdim = self._d["idim"], self._d["odim"], int(configs["hidden_dim"])
        for name, shape in pack.model_spec(configs):
            parts = name.split(".")
            mod = self
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, _Holder())
                mod = mod._modules[p]
            t = _default_init(name, shape)
            if parts[-1] in _BUFFER_LEAVES:
                mod.register_buffer(parts[-1], t)
            else:
                mod.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))
        self._handle: Optional[_HipHandle] = None
        self._handle_key = None
        self._tlist = None
        self._frozen = False
        self._packed_blob = None
        self._packed_versions = None
        self._options = {}

    # ------------------------------------------------------------------ weights -> device library
    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float(): buffers become new tensor objects (version counters restart at 0) and storages move.
        # Moving or casting the module's tensors is not a modification of the weights: a blob installed by load_packed()
        # that was still valid before the move stays what runs (and what packed() / checkpoints report) after it.
        blob = getattr(self, "_packed_blob", None)
        keep = blob is not None and (self._frozen or self._versions() == self._packed_versions)
        self._tlist = None
        self._handle = None
        self._frozen = False
        out = super()._apply(fn, *args, **kwargs)
        if keep:
            self._packed_versions = self._versions()
        elif blob is not None:
            self._packed_blob = None
        return out

    def load_state_dict(self, *args, **kwargs):
        # assign=True swaps the Parameter objects themselves: drop the cached tensor list and re-pack on the next call
        self._frozen = False
        self._tlist = None
        self._handle = None
        self._packed_blob = None          # a blob installed by load_packed() is superseded by the new tensors
        return super().load_state_dict(*args, **kwargs)

    def freeze(self) -> "KWSModel":
        """Promise that the weights will not be modified in place any more: forward stops comparing tensor versions
        (a streaming loop over MDTC's 363 tensors otherwise spends more host time there than the GPU needs for the
        chunk).  load_state_dict / .to() / set_precision / load_packed lift the promise again."""
        if getattr(self, "_packed_blob", None) is not None and self._versions() != self._packed_versions:
            self._packed_blob = None      # edited after load_packed(): the module's own tensors win (as in _get_handle)
            self._handle = None
        self._frozen = True
        return self

    def _weights_key(self, device: torch.device):
        # every parameter / buffer counts its in-place modifications: a flat cached list makes the check O(tensors)
        # attribute reads per call instead of rebuilding the state_dict
        if self._tlist is None:
            self._tlist = list(self.state_dict(keep_vars=True).values())
        lst = self._tlist
        return (device.index, lst[0].data_ptr() if lst else 0) + tuple(t._version for t in lst)

    def load_packed(self, blob: np.ndarray) -> None:
        """Install an already folded weight blob (e.g. received through the RCCL broadcast of
        wekws_amd.parallel.broadcast_weights) instead of packing this module's own tensors."""
        desc = {k: int(self._d[k]) for k in pack.DESC_FIELDS}
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        if blob.size != pack.blob_elems(desc):
            raise ValueError(f"blob has {blob.size} floats, config needs {pack.blob_elems(desc)}")
        self._packed_blob = blob
        self._packed_versions = self._versions()   # in-place edits of the module's tensors after this point supersede it
        self._handle = None
        self._frozen = False

    def _versions(self):
        if self._tlist is None:
            self._tlist = list(self.state_dict(keep_vars=True).values())
        return tuple(t._version for t in self._tlist)

    def _get_handle(self, device: torch.device) -> _HipHandle:
        if getattr(self, "_packed_blob", None) is not None and not self._frozen and \
                self._versions() != self._packed_versions:
            self._packed_blob = None      # the module's own weights were modified after load_packed(): they win
            self._handle = None
        if getattr(self, "_packed_blob", None) is not None:
            if self._handle is None or self._handle_key != ("packed", device.index):
                desc = {k: int(self._d[k]) for k in pack.DESC_FIELDS}
                self._handle = _HipHandle(desc, self._packed_blob, device.index if device.index is not None else 0,
                                          self._options)
                self._handle_key = ("packed", device.index)
            return self._handle
        if self._frozen and self._handle is not None and self._handle_key and self._handle_key[0] == device.index:
            return self._handle
        key = self._weights_key(device)
        if self._handle is None or key != self._handle_key:
            sd = {k: v.detach().cpu().numpy() for k, v in self.state_dict().items()}
            desc, blob = pack.pack(self._cfg, sd)
            self._handle = _HipHandle(desc, blob, device.index if device.index is not None else 0, self._options)
            self._handle_key = key
        return self._handle

    def set_precision(self, mode: str) -> "KWSModel":
        """'default' | 'f32' (exact-f32 matrix instructions) | 'f16x3' (fp16 hi/lo split, fp32-level accuracy,
        ~5x the matrix rate) -- enum wekws_hip_precision.  Can also be given as configs['_precision']."""
        if mode not in pack.PRECISION:
            raise ValueError(f"precision must be one of {sorted(pack.PRECISION)}")
        self._cfg["_precision"] = mode
        self._d["precision"] = pack.PRECISION[mode]
        self._handle = None
        self._frozen = False
        return self

    def effective_precision(self, device: Optional[torch.device] = None) -> str:
        """The arithmetic the calls really run in -- 'f32' | 'f16x3' | 'f16' (wekws_hip_effective_precision).  set_precision
        is a request: FSMN has the split-fp16 kernel only (an 'f32' request gets 'f16x3': same 1e-4 bar, not the
        reference's rounding), 'f16' is honoured by the 16-wave DS-TCN / MDTC kernels only."""
        dev = device or next(self.parameters()).device
        code = _capi.load().wekws_hip_effective_precision(self._get_handle(dev).ptr)
        _capi.check(min(code, 0), "wekws_hip_effective_precision")
        return {v: k for k, v in pack.PRECISION.items()}[code]

    def weight_spread_log2(self, device: Optional[torch.device] = None) -> float:
        """Largest spread (binades) between the row / column magnitudes of any matrix that feeds the matrix cores
        (wekws_hip_weight_spread_log2); above 20 a default-precision model runs the exact-f32 kernels."""
        dev = device or next(self.parameters()).device
        return float(_capi.load().wekws_hip_weight_spread_log2(self._get_handle(dev).ptr))

    def set_option(self, name: str, value: int) -> "KWSModel":
        """Kernel-selection override (enum wekws_hip_option: 'w16', 'mdtc16', 'stream', 'mm', 'head_slices') -- for A/B
        measurements and the tests that keep every kernel family parity-green; defaults are the product choice."""
        if name not in _capi.OPTIONS:
            raise ValueError(f"option must be one of {sorted(_capi.OPTIONS)}")
        self._options[name] = int(value)
        self._handle = None
        self._frozen = False
        return self

    def reserve(self, B: int, T: int, device: Optional[torch.device] = None) -> "KWSModel":
        """Size the scratch buffer of the current stream for calls of up to (B, T) now (wekws_hip_reserve): later calls
        then never allocate or synchronise -- required before capturing them into a HIP graph."""
        dev = device or next(self.parameters()).device
        h = self._get_handle(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _capi.check(_capi.load().wekws_hip_reserve(h.ptr, int(B), int(T), ctypes.c_void_p(stream)), "wekws_hip_reserve")
        return self

    def check(self, device: Optional[torch.device] = None) -> "KWSModel":
        """Synchronise the current stream and raise if a device-side wait of an earlier forward gave up
        (wekws_hip_forward_status; only the GRU wavefront has such waits).  Optional: a forward that gave up is also
        reported by the NEXT forward() on the same stream, which raises without being asked -- this is the call for the last
        forward of a script, and a cheap synchronisation point where results are read anyway."""
        dev = device or next(self.parameters()).device
        h = self._get_handle(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _capi.check(_capi.load().wekws_hip_forward_status(h.ptr, ctypes.c_void_p(stream)), "wekws_hip_forward_status")
        return self

    def packed(self) -> Tuple[dict, np.ndarray]:
        """(descriptor, folded float32 blob) -- what wekws_hip_create consumes; used by the multi-GPU
        weight broadcast (wekws_amd/parallel.py) and the model-file writer."""
        if getattr(self, "_packed_blob", None) is not None and self._versions() == self._packed_versions:
            return {k: int(self._d[k]) for k in pack.DESC_FIELDS}, self._packed_blob     # what is actually running
        sd = {k: v.detach().cpu().numpy() for k, v in self.state_dict().items()}
        return pack.pack(self._cfg, sd)

    # ------------------------------------------------------------------ forward
    def _run(self, x: torch.Tensor, in_cache: Optional[torch.Tensor], softmax: bool, want_cache: bool = True):
        if not isinstance(x, torch.Tensor) or x.dim() != 3 or x.size(2) != self.idim:
            raise ValueError(f"expected x of shape (B, T, {self.idim}), got {tuple(x.shape) if hasattr(x, 'shape') else x}")
        # (NaN / Inf features or caches need no check here: the kernels return what the reference returns for them -- NaN / Inf in
        # that utterance's causal receptive field, the other utterances untouched -- wekws_amd/csrc/nonfinite.hip.h)
        if not x.is_cuda:
            raise RuntimeError("wekws_amd.KWSModel runs on the MI355X HIP path only (no CPU fallback): "
                               "move the model and its inputs to a ROCm device")
        if x.dtype != torch.float32:
            raise TypeError(f"x must be float32, got {x.dtype}")
        dev = x.device
        B, T = int(x.size(0)), int(x.size(1))
        if T <= 0:                                           # an empty time axis: the reference's own errors, by backbone
            bb = self._d["backbone"]
            if bb == pack.BACKBONE["gru"]:
                raise RuntimeError("Expected sequence length to be larger than 0 in RNN")          # torch.nn.GRU, kws_model.py:73
            if bb == pack.BACKBONE["fsmn"]:
                raise RuntimeError("Kernel size can't be greater than actual input size (empty time axis)")   # fsmn.py's memory conv
            raise AssertionError()                            # tcn.py:53 / mdtc.py:112 `assert y.size(2) > self.padding`
        x = x.contiguous()
        h = self._get_handle(dev)
        lib = _capi.load()
        cshape = pack.cache_shape(self._d, B)
        cin = None
        if in_cache is not None and in_cache.numel() > 0:
            if tuple(in_cache.shape) != cshape:
                raise ValueError(f"in_cache shape {tuple(in_cache.shape)} != {cshape}")
            cin = in_cache.to(device=dev, dtype=torch.float32).contiguous()
        per_frame = self._d["head"] in (pack.HEAD["linear"], pack.HEAD["identity"])
        y = torch.empty((B, T, self.odim) if per_frame else (B, self.odim), dtype=torch.float32, device=dev)
        out_cache = torch.empty(cshape if want_cache else (0,), dtype=torch.float32, device=dev)
        if B > 0:
            stream = torch.cuda.current_stream(dev).cuda_stream
            _capi.check(lib.wekws_hip_forward(h.ptr, x.data_ptr(), B, T, cin.data_ptr() if cin is not None else None,
                                              y.data_ptr(), out_cache.data_ptr() if out_cache.numel() else None,
                                              1 if softmax else 0, ctypes.c_void_p(stream)), "wekws_hip_forward")
        return y, out_cache

    def forward(self, x: torch.Tensor,
                in_cache: torch.Tensor = torch.zeros(0, 0, 0, dtype=torch.float)) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._run(x, in_cache, False)

    def forward_softmax(self, x: torch.Tensor,
                        in_cache: torch.Tensor = torch.zeros(0, 0, 0, dtype=torch.float)
                        ) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._d["head"] in (pack.HEAD["glob"], pack.HEAD["last"]):
            raise IndexError("Dimension out of range (expected to be in range of [-2, 1], but got 2)")  # x.softmax(2) on (B, K)
        return self._run(x, in_cache, True)

    def posteriors(self, x: torch.Tensor, softmax: bool = False) -> torch.Tensor:
        """``model(x)[0]`` for callers that drop the cache anyway (wekws/bin/score.py:125 ``logits, _ = model(feats)``):
        the C ABI takes ``out_cache = NULL`` and the kernels then skip the cache hand-over and its HBM writes
        (110 MB per 1024 utterances for DS-TCN h256)."""
        return self._run(x, None, softmax, want_cache=False)[0]

    def forward_stream(self, x: torch.Tensor, in_cache: Optional[torch.Tensor] = None
                       ) -> Tuple[torch.Tensor, torch.Tensor]:
        """One streaming step: ``y, cache = model.forward_stream(chunk, cache)``; ``None`` / an empty tensor starts a
        stream.  The reference has no method of this name -- its streaming callers call ``forward(x, in_cache)``
        and rebind the cache (wekws/bin/stream_kws_ctc.py:486-487, runtime/core/kws/keyword_spotting.cc:63-94) --
        so this is that call under the name BASELINE.json's north_star uses."""
        return self._run(x, in_cache, False)

    def fuse_modules(self):
        """Reference: kws_model.py:92-94 (quantisation-time Conv+BN+ReLU fusion).  Here BN is always folded
        at pack time, so this is a no-op kept for API compatibility."""
        return None


def init_model(configs: Mapping) -> KWSModel:
    """Reference: wekws/model/kws_model.py:97-214.  Unknown types exit like the reference does."""
    try:
        model = KWSModel(configs)
    except pack.ConfigError as e:
        print(str(e))
        sys.exit(1)
    cmvn = configs.get("cmvn", {}) or {}
    if cmvn.get("cmvn_file"):
        from wekws_amd.utils.cmvn import load_cmvn, load_kaldi_cmvn
        import os
        if os.path.exists(cmvn["cmvn_file"]):
            loader = load_kaldi_cmvn if "kaldi" in cmvn["cmvn_file"] else load_cmvn
            mean, istd = loader(cmvn["cmvn_file"])
            model.global_cmvn.mean.copy_(torch.from_numpy(mean).float())
            model.global_cmvn.istd.copy_(torch.from_numpy(istd).float())
        # else: the statistics file recorded in config.yaml is not on this machine (the reference would raise here);
        # the global_cmvn.mean / .istd buffers are still created and a checkpoint's state_dict fills them
    return model


def load_exported(path: str) -> KWSModel:
    """A model file written by the reference's exporters -- the ``.onnx`` of wekws/bin/export_onnx.py:62-77 or its
    ORT-format conversion (runtime/android/app/src/main/assets/kws.ort) -- as a KWSModel on the HIP path.

    The file is recognised, not executed (wekws_amd/utils/onnx_lower.py): the result is an ordinary KWSModel whose
    config / state_dict were read off the graph, BatchNorm already folded.  ``model(x, cache)`` then equals
    ``ort_sess.run(None, {'input': x, 'cache': cache})`` of export_onnx.py:80-85 -- including the softmax of CTC
    exports -- for any batch size, with the cache optional as in KWSModel.forward."""
    from wekws_amd.utils.onnx_lower import load_model_file
    cfg, sd, info = load_model_file(path)
    if info["softmax"]:
        cfg["_exported_softmax"] = True
    model = KWSModel(cfg)
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
    model.exported_info = info
    return model.eval()

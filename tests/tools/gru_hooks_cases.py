"""Cases of tests/test_hip_gru_safety.py that need the TEST build of the library (libwekws_hip_hooks.so: the product library plus
two debug entry points, `make -C wekws_amd/csrc hooks`).  Run as a subprocess with WEKWS_HIP_LIB pointing at it:

    python tests/tools/gru_hooks_cases.py epoch_wrap | starved | squeezed | worker <seconds>

Exit code 0 = the case passed; anything else is a failure (assertion text on stderr)."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from wekws_amd import _capi, pack  # noqa: E402
from wekws_amd.model.kws_model import init_model  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402


def gru_cfg(layers):
    cfg = dict(synth.MODEL_CONFIGS["gru_2x128"])
    cfg["backbone"] = dict(cfg["backbone"], num_layers=layers)
    return cfg


def build(cfg, sd):
    m = init_model(cfg)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return m.to("cuda").eval()


def hooks():
    lib = _capi.load()
    assert _capi.lib_path().endswith("libwekws_hip_hooks.so"), _capi.lib_path()
    lib.wekws_hip_debug_set_gru_epoch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
    lib.wekws_hip_debug_set_gru_epoch.restype = ctypes.c_int
    lib.wekws_hip_debug_hog.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.wekws_hip_debug_hog.restype = ctypes.c_int
    return lib


def epoch_wrap():
    """The hand-over tags come from a 32-bit launch counter on the device; a streaming server reaches its wrap after about a
    day of back-to-back chunks.  Tags are compared for equality and 0 is never a tag, so nothing may change when the
    counter passes 2^32: launches of every kind (one chunk, full tiles beyond a lap of the rings, several rounds) walked
    across the wrap from a few steps before it, every one compared with the layer-major kernels."""
    cfg = gru_cfg(2)
    sd = synth.synth_state_dict(pack.model_spec(cfg), 4246)
    pipe, major = build(cfg, sd).set_option("gru_pipe", 2), build(cfg, sd).set_option("gru_pipe", 0)
    lib, dev = hooks(), torch.device("cuda", torch.cuda.current_device())
    shapes = [(1, 10), (300, 40), (3, 98), (2100, 21), (256, 10)]
    ref = {}
    for B, T in shapes:
        x = torch.from_numpy(synth.synth_feats(B, T, cfg["input_dim"], seed=B + T)).cuda()
        ref[(B, T)] = (x,) + tuple(major(x))
    pipe(ref[(1, 10)][0])                                       # (the control block exists now)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for start in (0xFFFFFFF0, 0xFFFFFFFF, 0x7FFFFFFC, 0x0FFFFFFE):     # 2^32, the substitute for tag 0, the 28-bit XCD words
        assert lib.wekws_hip_debug_set_gru_epoch(pipe._get_handle(dev).ptr, stream, start) == 0, _capi.last_error()
        for rep in range(4):
            for B, T in shapes:
                x, y0, c0 = ref[(B, T)]
                y1, c1 = pipe(x)
                assert torch.equal(y1, y0) and torch.equal(c1, c0), (hex(start), rep, B, T)
    pipe.check()


def _same(pipe, got, want, what):
    """Bit-identical, or say exactly how not: a mismatch the library REPORTS (a bounded wait gave up) and one it does not are
    different failures."""
    (y1, c1), (y0, c0) = got, want
    if torch.equal(y1, y0) and torch.equal(c1, c0):
        return
    torch.cuda.synchronize()
    ny, nc = int((y1 != y0).sum()), int((c1 != c0).sum())
    try:
        pipe.check()
        rep = "NOT reported by the health word (silent)"
    except _capi.HipLibraryError as e:
        rep = f"reported: {e}"
    raise AssertionError(f"{what}: {ny} of {y0.numel()} posteriors and {nc} of {c0.numel()} states differ "
                         f"(max |dy| {float((y1 - y0).abs().max()):.3e}); {rep}")


def _hog_setup():
    cfg = gru_cfg(2)
    sd = synth.synth_state_dict(pack.model_spec(cfg), 4250)
    pipe, major = build(cfg, sd).set_option("gru_pipe", 2), build(cfg, sd).set_option("gru_pipe", 0)
    x = torch.from_numpy(synth.synth_feats(1024, 98, cfg["input_dim"], seed=5)).cuda()
    y0, c0 = major(x)
    y1, c1 = pipe(x)                                            # (workspaces exist; the launch below allocates nothing)
    assert torch.equal(y1, y0) and torch.equal(c1, c0)
    torch.cuda.synchronize()
    return pipe, x, y0, c0


def starved():
    """A tenant holds 250 of the 256 CUs for 0.6 s; a full-grid wavefront launch (B = 1024 x 98 frames: 256 workgroups, ring
    credits) gets six.  Its resident producers run out of ring positions and wait for consumers that have no CU -- longer than
    the bounded waits allow.  Required: the launch ENDS (no hang: the whole case takes about a second), and the caller hears
    of it WITHOUT asking -- the next forward on the stream raises, and the call after that works again."""
    lib = hooks()
    pipe, x, y0, c0 = _hog_setup()
    dev = torch.device("cuda", torch.cuda.current_device())
    side = torch.cuda.Stream()
    assert lib.wekws_hip_debug_hog(dev.index, 250, 600, ctypes.c_void_p(side.cuda_stream)) == 0, _capi.last_error()
    time.sleep(0.05)                                            # the hog is resident
    t0 = time.time()
    y1, c1 = pipe(x)                                            # issued while the device is squeezed: returns at once
    torch.cuda.synchronize()
    dt = time.time() - t0
    assert dt < 3.0, f"the starved launch took {dt:.2f} s: its waits are not bounded"
    raised = False
    try:
        pipe(x)                                                 # the NEXT call reports (no check() ritual)
    except _capi.HipLibraryError as e:
        raised = "bounded wait" in str(e)
    if not raised:
        # the squeeze may have been survivable (consumers found CUs in time): then the results must be right
        assert torch.equal(y1, y0) and torch.equal(c1, c0), "no error reported, but wrong results"
        print("starved: the launch survived the squeeze (results bit-identical); nothing to report")
        return
    y2, c2 = pipe(x)                                            # reported once; the stream works again
    torch.cuda.synchronize()
    assert torch.equal(y2, y0) and torch.equal(c2, c0)
    pipe.check()
    print(f"starved: launch ended after {dt:.2f} s, failure raised by the next forward, stream healthy afterwards")


def squeezed():
    """The same squeeze for 30 ms -- well inside the bound: the launch waits for its CUs and returns bit-identical results."""
    lib = hooks()
    pipe, x, y0, c0 = _hog_setup()
    dev = torch.device("cuda", torch.cuda.current_device())
    side = torch.cuda.Stream()
    for it in range(5):
        assert lib.wekws_hip_debug_hog(dev.index, 250, 30, ctypes.c_void_p(side.cuda_stream)) == 0, _capi.last_error()
        time.sleep(0.005)
        got = pipe(x)
        torch.cuda.synchronize()
        _same(pipe, got, (y0, c0), f"squeezed launch {it}")
    pipe.check()


def worker(seconds):
    """A second PROCESS on the same GPU (tests/test_hip_gru_safety.py::test_two_processes_share_the_gpu): full-grid wavefront
    forwards back to back for `seconds`, every result compared with the layer-major kernels."""
    cfg = gru_cfg(2)
    sd = synth.synth_state_dict(pack.model_spec(cfg), 4251)
    pipe, major = build(cfg, sd).set_option("gru_pipe", 2), build(cfg, sd).set_option("gru_pipe", 0)
    x = torch.from_numpy(synth.synth_feats(1024, 98, cfg["input_dim"], seed=6)).cuda()
    y0, c0 = major(x)
    torch.cuda.synchronize()
    print("ready", flush=True)
    sys.stdin.readline()                                        # both processes start together
    n, t0, reported = 0, time.time(), []
    while time.time() - t0 < seconds:
        try:
            outs = [pipe(x) for _ in range(8)]
            torch.cuda.synchronize()
            for got in outs:
                _same(pipe, got, (y0, c0), f"worker launch {n}")
                n += 1
        except (AssertionError, _capi.HipLibraryError) as e:
            if "NOT reported" in str(e):
                raise
            reported.append(str(e))                              # a REPORTED give-up (the failure contract): tolerated twice, printed
            if len(reported) > 2:
                raise
    if reported:
        print("reported give-ups in the worker:", reported, flush=True)
    else:
        pipe.check()
    print(f"done {n}", flush=True)


if __name__ == "__main__":
    case = sys.argv[1]
    if case == "worker":
        worker(float(sys.argv[2]))
    else:
        {"epoch_wrap": epoch_wrap, "starved": starved, "squeezed": squeezed}[case]()
    print("OK")

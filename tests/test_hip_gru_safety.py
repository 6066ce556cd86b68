"""GPU tests of the GRU layer wavefront (wekws_amd/csrc/gru_pipe.hip.h) as a TENANT of a shared device: its launches hand
sequences between workgroups, so they must stay correct -- and must end -- whatever else runs on the GPU.

  * two FULL-GRID wavefront launches at once (two streams of one model, two models, a second process): every word compared
    with the layer-major kernels (torch.nn.GRU as built at wekws/model/kws_model.py:128-133; same instructions per column,
    so bit-identical), then the health check;
  * a launch that is starved of CUs for longer than its bounded waits ENDS and is reported by the next forward;
  * wekws_hip_forward_status synchronises for every model.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.test_hip_parity import _gru_cfg, build
from wekws_amd.utils import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOOKS = os.path.join(ROOT, "wekws_amd", "lib", "libwekws_hip_hooks.so")
CASES = os.path.join(ROOT, "tests", "tools", "gru_hooks_cases.py")


def _pair(seed, layers=2):
    from wekws_amd import pack
    cfg = _gru_cfg(layers)
    sd = synth.synth_state_dict(pack.model_spec(cfg), seed)
    return cfg, build(cfg, sd).set_option("gru_pipe", 2), build(cfg, sd).set_option("gru_pipe", 0)


@pytest.mark.parametrize("B", [1024, 4096])
def test_two_full_grid_wavefronts_on_two_streams(B):
    """ONE model, two streams, B x 98 frames each: 2 x 256 workgroups for 256 CUs, ring credits in use (98 > 16 steps; B = 4096:
    four rounds per slot).  50 iterations, both streams' launches in flight together."""
    cfg, pipe, major = _pair(4301)
    xs = [torch.from_numpy(synth.synth_feats(B, 98, cfg["input_dim"], seed=s)).cuda() for s in (11, 12)]
    ref = [tuple(t.clone() for t in major(x)) for x in xs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for i in range(2):                                          # each stream's workspace exists before the race
        with torch.cuda.stream(streams[i]):
            pipe(xs[i])
    torch.cuda.synchronize()
    for it in range(50):
        outs = []
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                outs.append(pipe(xs[i]))
        torch.cuda.synchronize()
        for i, (y, c) in enumerate(outs):
            assert torch.equal(y, ref[i][0]) and torch.equal(c, ref[i][1]), (B, it, i)
    for s in streams:
        with torch.cuda.stream(s):
            pipe.check()


@pytest.mark.parametrize("B", [1024, 4096])
def test_two_models_full_grid_wavefronts(B):
    """TWO models (2 and 3 layers: 4 and 6 stages) on two streams, plus the default stream joining in every fifth iteration:
    up to three credit-using launches share the CUs."""
    cfg2, pipe2, major2 = _pair(4302, 2)
    cfg3, pipe3, major3 = _pair(4303, 3)
    x2 = torch.from_numpy(synth.synth_feats(B, 98, 40, seed=21)).cuda()
    x3 = torch.from_numpy(synth.synth_feats(B, 98, 40, seed=22)).cuda()
    r2, r3 = tuple(t.clone() for t in major2(x2)), tuple(t.clone() for t in major3(x3))
    s2, s3 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s2):
        pipe2(x2)
    with torch.cuda.stream(s3):
        pipe3(x3)
    pipe2(x3)
    torch.cuda.synchronize()
    r2b = tuple(t.clone() for t in major2(x3))
    for it in range(50):
        with torch.cuda.stream(s2):
            o2 = pipe2(x2)
        with torch.cuda.stream(s3):
            o3 = pipe3(x3)
        o2b = pipe2(x3) if it % 5 == 0 else None
        torch.cuda.synchronize()
        assert torch.equal(o2[0], r2[0]) and torch.equal(o2[1], r2[1]), (B, it, "2 layers")
        assert torch.equal(o3[0], r3[0]) and torch.equal(o3[1], r3[1]), (B, it, "3 layers")
        if o2b is not None:
            assert torch.equal(o2b[0], r2b[0]) and torch.equal(o2b[1], r2b[1]), (B, it, "default stream")
    with torch.cuda.stream(s2):
        pipe2.check()
    with torch.cuda.stream(s3):
        pipe3.check()
    pipe2.check()


def test_pipelined_wavefronts_without_synchronisation():
    """Many launches queued on three streams before anything is waited for (3+ pending launches: the case ADVICE r4 calls likely
    to interleave), chunk-sized launches of a streaming caller in between."""
    cfg, pipe, major = _pair(4304)
    x = torch.from_numpy(synth.synth_feats(1024, 98, 40, seed=31)).cuda()
    xc = torch.from_numpy(synth.synth_feats(1, 10, 40, seed=32)).cuda()
    ref, refc = tuple(t.clone() for t in major(x)), tuple(t.clone() for t in major(xc))
    streams = [torch.cuda.Stream() for _ in range(3)]
    for s in streams:
        with torch.cuda.stream(s):
            pipe(x), pipe(xc)
    torch.cuda.synchronize()
    outs = []
    for it in range(20):
        for s in streams:
            with torch.cuda.stream(s):
                outs.append((pipe(x), ref))
                outs.append((pipe(xc), refc))
    torch.cuda.synchronize()
    for n, ((y, c), (ry, rc)) in enumerate(outs):
        assert torch.equal(y, ry) and torch.equal(c, rc), n
    for s in streams:
        with torch.cuda.stream(s):
            pipe.check()


def _run_case(args, hooks=True, timeout=600):
    env = dict(os.environ)
    if hooks:
        assert os.path.exists(HOOKS), f"{HOOKS} is missing: make -C wekws_amd/csrc hooks (or __graft_entry__.build())"
        env["WEKWS_HIP_LIB"] = HOOKS
    else:
        env.pop("WEKWS_HIP_LIB", None)
    return subprocess.run([sys.executable, CASES] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_two_processes_share_the_gpu():
    """A second PROCESS drives its own full-grid wavefront launches on the same GPU while this one does (different HSA queues:
    nothing orders the two processes' launches); both compare every result with the layer-major kernels."""
    cfg, pipe, major = _pair(4305)
    x = torch.from_numpy(synth.synth_feats(1024, 98, 40, seed=41)).cuda()
    y0, c0 = major(x)
    pipe(x)
    torch.cuda.synchronize()
    env = dict(os.environ)
    env.pop("WEKWS_HIP_LIB", None)
    p = subprocess.Popen([sys.executable, CASES, "worker", "4"], env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True)
    try:
        assert p.stdout.readline().strip() == "ready", p.stderr.read()
        p.stdin.write("go\n")
        p.stdin.flush()
        n, reported = 0, []
        while p.poll() is None:
            try:
                outs = [pipe(x) for _ in range(8)]
                torch.cuda.synchronize()
            except Exception as e:       # noqa: BLE001  (a forward behind a give-up returns the error by itself: the contract)
                reported.append(f"forward raised: {e}")
                assert len(reported) <= 2, reported
                continue
            for y1, c1 in outs:
                if not (torch.equal(y1, y0) and torch.equal(c1, c0)):
                    try:
                        pipe.check()
                    except Exception as e:       # noqa: BLE001
                        # A bounded wait gave up under the other process's load and the library SAID so: the failure contract of
                        # DESIGN.md section 1, not a wrong result.  (Round 6: this test failed once in 29 runs with only its summary line
                        # kept; a reported give-up is tolerated twice per process and printed, a silent mismatch never.)
                        reported.append(f"parent launch {n}: {int((y1 != y0).sum())} posteriors differ; reported: {e}")
                        assert len(reported) <= 2, reported
                        break
                    raise AssertionError(f"parent launch {n}: {int((y1 != y0).sum())} posteriors differ; NOT reported by the health word (silent)")
                n += 1
        if reported:
            print("reported give-ups in the parent:", reported)
        out, err = p.communicate(timeout=60)
    finally:
        if p.poll() is None:
            p.kill()
    assert p.returncode == 0 and "OK" in out, (out, err[-2000:])
    assert n >= 8
    if not reported:
        pipe.check()


def test_gru_wavefront_epoch_wrap():
    r = _run_case(["epoch_wrap"])
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_starved_wavefront_ends_and_is_reported_by_the_next_call():
    r = _run_case(["starved"])
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    print(r.stdout)


def test_squeezed_wavefront_waits_and_is_correct():
    r = _run_case(["squeezed"])
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_product_library_exports_no_debug_entry():
    import ctypes
    from wekws_amd import _capi
    lib = ctypes.CDLL(_capi.lib_path())
    for name in ("wekws_hip_debug_set_gru_epoch", "wekws_hip_debug_hog"):
        assert not hasattr(lib, name), name


@pytest.mark.parametrize("name", ["ds_tcn_h256", "mdtc_h64", "fsmn_small", "gru_2x128"])
def test_forward_status_synchronises_every_model(name):
    """wekws_hip_forward_status is documented to SYNCHRONISE the stream; round 4 returned early for streams without wavefront
    control words (every non-GRU model), and the C++ runtime reads its host buffer right behind the call.  An asynchronous copy
    into PINNED memory behind a long queue of forwards: after check() -- and nothing else -- the host buffer holds the result."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS[name])
    m = build(cfg, synth.synth_state_dict(pack.model_spec(cfg), 5))
    x = torch.from_numpy(synth.synth_feats(512, 64, cfg["input_dim"], seed=3)).cuda()
    want = m(x)[0].cpu()
    s = torch.cuda.Stream()
    host = torch.zeros(want.shape, dtype=torch.float32).pin_memory()
    with torch.cuda.stream(s):
        for _ in range(30):
            y = m(x)[0]
        host.copy_(y, non_blocking=True)
        m.check()
    assert torch.equal(host, want), name
    torch.cuda.synchronize()

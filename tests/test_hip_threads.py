"""GPU: the boundary's re-entrancy promise ("`forward` re-entrant across streams / threads as long as each call has
its own cache / y buffers") driven from HOST THREADS: eight threads, each with its own HIP stream, share ONE model object (one
wekws_hip_model, one weight image, the per-(model, stream) workspace map under its mutex) and run 200 forwards each with
their own inputs and carried caches; every output must equal the single-threaded result bit for bit.  (ctypes releases the GIL
for the duration of a call, so the calls really overlap.)"""
import threading

import numpy as np
import pytest
import torch

from tests.test_hip_parity import build
from wekws_amd import pack
from wekws_amd.utils import synth

pytestmark = pytest.mark.gpu
NTHREADS, ITERS = 8, 200


def _work(model, x, cache, chunks):
    """A thread's script: a streaming pass (carried cache) repeated; returns the last outputs (deterministic per thread)."""
    ys = None
    for _ in range(ITERS // len(chunks)):
        c, ys, t = cache, [], 0
        for n in chunks:
            y, c = model(x[:, t:t + n].contiguous()) if c is None else model(x[:, t:t + n].contiguous(), c)
            ys.append(y)
            t += n
    return torch.cat(ys, 1), c


@pytest.mark.parametrize("name,B,chunks", [("ds_tcn_h256", 3, [98]), ("ds_tcn_h256", 2, [10] * 8), ("mdtc_h64", 3, [40, 40]),
                                           ("gru_2x128", 3, [10] * 5), ("gru_2x128", 40, [98]), ("fsmn_small", 2, [20, 20]),
                                           ("tcn_h64", 2, [30, 7, 30])])
def test_one_model_many_host_threads(name, B, chunks):
    cfg = dict(synth.MODEL_CONFIGS[name])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
    model = build(cfg, sd).freeze()
    T = sum(chunks)
    gru = cfg["backbone"]["type"] == "gru"
    xs = [torch.from_numpy(synth.synth_feats(B, T, cfg["input_dim"], seed=50 + i)).cuda() for i in range(NTHREADS)]
    c0 = [torch.from_numpy((0.3 * np.random.default_rng(i).standard_normal((cfg["backbone"]["num_layers"], B, cfg["hidden_dim"])))
                           .astype(np.float32)).cuda() if gru else None for i in range(NTHREADS)]
    want = [_work(model, xs[i], c0[i], chunks) for i in range(NTHREADS)]          # single-threaded, default stream
    torch.cuda.synchronize()
    got, errs = [None] * NTHREADS, []
    go = threading.Barrier(NTHREADS)

    def run(i):
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.default_stream())
            go.wait()
            with torch.cuda.stream(s):
                got[i] = _work(model, xs[i], c0[i], chunks)
            s.synchronize()
        except Exception as e:                                                     # (surface it in the main thread)
            errs.append((i, repr(e)))

    ths = [threading.Thread(target=run, args=(i,)) for i in range(NTHREADS)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    torch.cuda.synchronize()
    for i in range(NTHREADS):
        assert torch.equal(got[i][0], want[i][0]), f"thread {i}: y differs"
        assert torch.equal(got[i][1], want[i][1]), f"thread {i}: cache differs"

#!/usr/bin/env python3
"""Secondary measurements for the other BASELINE.json configs (one JSON line each, not the bench.py contract):
batch throughput of every backbone, streaming per-frame latency (B=1, 10-frame chunks, carried cache), fbank
front-end throughput.  GPU only.   python tools/bench_configs.py > gpurun_out/configs.jsonl"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wekws_amd import pack  # noqa: E402
from wekws_amd.frontend import Fbank  # noqa: E402
from wekws_amd.model.kws_model import init_model  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402


def build(name):
    cfg = dict(synth.MODEL_CONFIGS[name])
    m = init_model(cfg)
    sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return cfg, m.cuda().eval().freeze()      # weights are final: skip the per-call version check


def timeit(fn, warm=5, reps=30, group=10):
    """Throughput timing: `group` back-to-back launches between two events (the stream never drains, so host launch
    overhead is hidden as in bench.py), repeated `reps` times; returns median / p10 / p90 of the per-launch time."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t_pre = time.perf_counter()       # an idle MI355X runs its first ~0.25 s of work at lower clocks: measure behind that
    while time.perf_counter() - t_pre < 0.3:
        for _ in range(group):
            fn()
        torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(group):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / group)
    return float(np.median(ts)), float(np.percentile(ts, 10)), float(np.percentile(ts, 90))


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ""   # optional substring filter on the model name
    out = []
    # FSMN-CTC: 400-d spliced features at frame_skip 3 (fsmn_ctc.yaml:21-25): 98 fbank frames (1 s) -> T = 32; 2 s -> 64
    for name, B, T in (("fsmn_ctc", 1024, 32), ("fsmn_ctc", 1024, 64), ("fsmn_ctc", 4096, 32)):
        if only not in name:
            continue
        cfg, m = build(name)
        x = torch.from_numpy(synth.synth_feats(B, T, cfg["input_dim"], seed=1)).cuda()
        med, p10, p90 = timeit(lambda: m(x), reps=6, group=6)
        out.append(dict(kind="batch", model=name, B=B, T=T, ms=round(med, 4), p10=round(p10, 4), p90=round(p90, 4),
                        utts_per_s=round(B / med * 1e3, 1), frames_per_s=round(B * T / med * 1e3, 1)))
        print(json.dumps(out[-1]), flush=True)
    # "@f16": WEKWS_HIP_PRECISION_F16, the reduced-precision mode of BASELINE.json config 5 (fp16 weights + fp16 MFMA
    # pointwise conv; MDTC + 12-class GlobalClassifier, 1024 utterances per GPU of the 8192)
    for name, B in (("ds_tcn_h256", 1024), ("ds_tcn_h256", 8192), ("mdtc_h64", 1024), ("mdtc_h64", 8192),
                    ("mdtc_h64_global12", 1024), ("mdtc_h64_global12@f16", 1024), ("mdtc_h64_global12@f16", 8192),
                    ("mdtc_h64@f16", 1024), ("ds_tcn_h256@f16", 1024),
                    ("mdtc_small", 1024), ("ds_tcn_h64", 1024), ("tcn_h64", 1024),
                    ("gru_2x128", 256), ("gru_2x128", 1024), ("gru_2x128", 16384)):
        if only not in name:
            continue
        cfg, m = build(name.split("@")[0])
        if "@" in name:
            m.set_precision(name.split("@")[1]).freeze()
        x = torch.from_numpy(synth.synth_feats(B, 98, cfg["input_dim"], seed=1)).cuda()
        med, p10, p90 = timeit(lambda: m(x), reps=10 if B <= 1024 else 4, group=10 if B <= 1024 else 4)
        out.append(dict(kind="batch", model=name, B=B, T=98, ms=round(med, 4), p10=round(p10, 4), p90=round(p90, 4),
                        utts_per_s=round(B / med * 1e3, 1)))
        print(json.dumps(out[-1]), flush=True)
    # streaming latency: B streams, 10-frame chunks, cache carried, host-timed per chunk (includes launch overhead)
    for name, B in (("gru_2x128", 1), ("gru_2x128", 256), ("ds_tcn_h256", 1), ("ds_tcn_h256", 256), ("mdtc_h64", 1),
                    ("mdtc_h64", 256), ("fsmn_ctc", 1), ("fsmn_ctc", 256)):
        if only not in name:
            continue
        cfg, m = build(name)
        x = torch.from_numpy(synth.synth_feats(B, 10, cfg["input_dim"], seed=2)).cuda()
        _, cache = m(x)
        t_pre = time.perf_counter()   # (steady clocks, as in timeit)
        while time.perf_counter() - t_pre < 0.3:
            for _ in range(20):
                _, cache = m(x, cache)
            torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            y, cache = m(x, cache)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e3
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            y, cache = m(x, cache)
        b.record()
        torch.cuda.synchronize()
        dev = a.elapsed_time(b) / n
        out.append(dict(kind="stream", model=name, B=B, chunk=10, ms_per_chunk_wall=round(wall, 4),
                        ms_per_chunk_stream=round(dev, 4), us_per_frame=round(dev * 100, 2)))
        print(json.dumps(out[-1]), flush=True)
    # streaming THROUGHPUT with many concurrent streams (10-frame chunks, caches carried): stream-chunks per second;
    # lds_cache = the streaming kernel that keeps each stream's cache in LDS (ds256_stream.hip.h, the default for
    # chunks of <= 16 frames); False = option stream = 0, the batch kernel fed the same chunks
    if only in "manystreams":
        for mname, B in [("ds_tcn_h256", b) for b in (1, 256, 1024, 4096, 16384)] + [("mdtc_h64", b) for b in (1, 256, 4096)]:
            for packed in (True, False):
                cfg, m = build(mname)
                m.set_option("stream", 1 if packed else 0).freeze()
                x = torch.from_numpy(synth.synth_feats(B, 10, 40, seed=2)).cuda()
                _, cache = m(x)
                state = {"c": cache}

                def step():
                    _, state["c"] = m(x, state["c"])
                med, p10, p90 = timeit(step, reps=6, group=6)
                out.append(dict(kind="manystreams", model=mname, B=B, chunk=10, lds_cache=packed, ms=round(med, 4),
                                chunks_per_s=round(B / med * 1e3, 1), frames_per_s=round(B * 10 / med * 1e3, 1)))
                print(json.dumps(out[-1]), flush=True)
    # end-to-end on the device, PCM resident in HBM: (a) fbank40 -> DS-TCN h256 posteriors; (b) fbank80 -> context
    # expansion(2, 2) / skip 3 -> FSMN-CTC logits -> fused softmax + top-3 (what stream_kws_ctc.py's decoder consumes)
    if only in "e2e":
        from wekws_amd import ctc
        from wekws_amd.frontend import splice_skip
        B = 1024
        pcm = torch.from_numpy(synth.synth_pcm(B, 16000, seed=3)).cuda()
        _, m1 = build("ds_tcn_h256")
        fb40 = Fbank(40)
        med, p10, p90 = timeit(lambda: m1(fb40(pcm)), reps=6, group=6)
        out.append(dict(kind="e2e", pipeline="pcm -> fbank40 -> ds_tcn_h256 -> posteriors", B=B, ms=round(med, 4),
                        utts_per_s=round(B / med * 1e3, 1)))
        print(json.dumps(out[-1]), flush=True)
        _, m2 = build("fsmn_ctc")
        fb80 = Fbank(80)
        med, p10, p90 = timeit(lambda: ctc.softmax_topk(m2(splice_skip(fb80(pcm), 2, 2, 3))[0], 3), reps=6, group=6)
        out.append(dict(kind="e2e", pipeline="pcm -> fbank80 -> splice(2,2)/skip3 -> fsmn_ctc -> softmax+top3", B=B,
                        ms=round(med, 4), utts_per_s=round(B / med * 1e3, 1)))
        print(json.dumps(out[-1]), flush=True)
    # host buffers in, host scores out (what a caller without device residency pays): pinned feats -> H2D -> DS-TCN h256
    # -> D2H of the posteriors.  "serial": one stream.  "overlapped": copies on their own streams, 3 batches in flight.
    if only in "pcie":
        B, T = 1024, 98
        _, m = build("ds_tcn_h256")
        nbuf = 3
        hx = [torch.from_numpy(synth.synth_feats(B, T, 40, seed=i)).pin_memory() for i in range(nbuf)]
        hy = [torch.empty(B, T, 2).pin_memory() for _ in range(nbuf)]
        dx = [torch.empty(B, T, 40, device="cuda") for _ in range(nbuf)]
        dy = [None] * nbuf

        def serial(n):
            for i in range(n):
                j = i % nbuf
                dx[j].copy_(hx[j], non_blocking=True)
                y, _ = m(dx[j])
                hy[j].copy_(y, non_blocking=True)

        up, comp, down = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        ev_up = [torch.cuda.Event() for _ in range(nbuf)]
        ev_c = [torch.cuda.Event() for _ in range(nbuf)]
        ev_d = [torch.cuda.Event() for _ in range(nbuf)]

        def overlapped(n):
            for i in range(n):
                j = i % nbuf
                with torch.cuda.stream(up):
                    up.wait_event(ev_c[j])                      # the previous forward on this buffer has read it
                    dx[j].copy_(hx[j], non_blocking=True)
                    ev_up[j].record(up)
                with torch.cuda.stream(comp):
                    comp.wait_event(ev_up[j])
                    comp.wait_event(ev_d[j])                    # its previous scores have left the device
                    dy[j], _ = m(dx[j])
                    ev_c[j].record(comp)
                with torch.cuda.stream(down):
                    down.wait_event(ev_c[j])
                    hy[j].copy_(dy[j], non_blocking=True)
                    ev_d[j].record(down)

        for label, fn in (("serial", serial), ("overlapped", overlapped)):
            fn(6)
            torch.cuda.synchronize()
            n = 60
            t0 = time.perf_counter()
            fn(n)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            out.append(dict(kind="pcie", mode=label, model="ds_tcn_h256", B=B, ms=round(ms, 4),
                            utts_per_s=round(B / ms * 1e3, 1), h2d_GBs=round(B * T * 40 * 4 / ms / 1e6, 1)))
            print(json.dumps(out[-1]), flush=True)
    fb = Fbank(40)
    for B in ((1024, 8192) if only in "fbank" else ()):
        pcm = torch.from_numpy(synth.synth_pcm(B, 16000, seed=3)).cuda()
        med, p10, p90 = timeit(lambda: fb(pcm))
        out.append(dict(kind="fbank", B=B, nsamp=16000, ms=round(med, 4), utts_per_s=round(B / med * 1e3, 1),
                        GBs=round((B * 16000 * 4 + B * 98 * 40 * 4) / med / 1e6, 1)))
        print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()

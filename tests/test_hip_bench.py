"""GPU: bench.py prints the one JSON line the driver parses, with every field of the contract and the two extra objects
(`roofline`, `cpu_baseline`) -- run small, but through the real code path."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(*flags, extras_out=None):
    """Runs bench.py as the driver does and reads its stdout as the driver does: the LAST line is the contract line; it must
    parse, be the only line starting with '{', and be small enough to sit whole in the 8 KB tail the driver keeps (round 5's
    22.5 KB line was unparseable for the driver: BENCH_r05.json `parsed: null`)."""
    if extras_out:
        flags = (*flags, "--extras-out", extras_out)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    out_lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    last = out_lines[-1]
    assert last.startswith('{"metric"'), last[:200]
    assert len(last.encode()) < 6000, f"contract line is {len(last.encode())} bytes"
    assert [ln for ln in out_lines if ln.startswith("{")] == [last], "exactly one stdout line may start with '{': the contract line"
    tail = p.stdout.encode()[-8192:].decode(errors="ignore")                       # what the driver's record keeps
    assert json.loads(tail[tail.rindex('{"metric"'):]) == json.loads(last)
    return json.loads(last)


def test_contract_line_small_run(tmp_path):
    ex = str(tmp_path / "extras.json")
    d = run_bench("--gpus", "1", "--steps", "4", "--warmup", "2", "--preheat", "0.05", "--no-extras", extras_out=ex)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "step_ms"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = utterances of the K timed steps / their wall time
    assert abs(d["value"] - d["config"]["batch_per_gpu"] / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and 0.0 < r["frac"] < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0           # only from a PMC profile of this very library build
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1
    side = json.load(open(ex))                                         # the verbose records live in the side file
    assert side["contract_line"]["value"] == d["value"]
    rows = side["extras"]["cpu_baseline_detail"]["rows"]
    pinned = rows["pinned_one_thread_per_physical_core"]               # one OpenMP thread per physical core, affinity set
    assert pinned and all(("utts_per_s" in r) or ("error" in r) for r in pinned.values())
    assert c["value"] >= max([r.get("utts_per_s", 0) for r in pinned.values()] + [rows["one_core"]["utts_per_s"]]) - 1e-6
    # both allocator patterns of the headline are in the record
    v1 = side["extras"]["value_single_output_buffer"]
    assert v1["value"] > 0 and "median" in v1["step_ms"]
    assert d["value_no_preheat"] > 0
    # the product is (much) faster than the CPU path it replaces; not a quality claim, a sanity check of both numbers
    assert d["value"] > 10 * c["value"]


def test_multi_rank_statements_run_on_a_gpu():
    """No session and no driver run ever had more than one GPU, so the N > 1 statements of bench.py (rank rendezvous, weight
    broadcast into rank > 0's module, barriers, MAX over ranks, per-rank placement, `comm`) had only ever run in the CPU stub.
    `--share-gpu` lets two ranks share cuda:0 over gloo: every one of those statements executes, with the real forward, on the GPU
    box.  The number is meaningless (two processes time-share one device) and the line says so."""
    d = run_bench("--gpus", "2", "--steps", "4", "--warmup", "2", "--preheat", "0.05", "--no-extras", "--no-cpu-baseline",
                  "--share-gpu")
    assert d["n_gpus"] == 2 and d["steps"] == 4 and "test_mode" in d
    assert d["comm"]["world_size"] == 2 and d["comm"]["collectives_in_timed_region"] == 0 and d["comm"]["broadcast_bytes"] > 1_000_000
    assert [p.split(":")[0] for p in d["comm"]["ranks"]] == ["r0", "r1"] and len(d["per_rank_utts_per_s"]["all"]) == 2
    assert d["value"] > 0 and d["value_no_preheat"] > 0
    # whole-job value = utterances of both ranks / the slowest rank's time
    assert abs(d["value"] - 2 * d["config"]["batch_per_gpu"] / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]


def test_default_run_line_fits_the_driver_tail(tmp_path):
    """The driver's own command (BENCH_r0N.json `cmd`), every secondary workload included: the last stdout line is the compact
    contract line with `roofline`, `cpu_baseline` and the `summary` of the extras; the extras are in the side file."""
    ex = str(tmp_path / "extras.json")
    d = run_bench("--gpus", "1", "--steps", "20", "--warmup", "5", extras_out=ex)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "summary", "value_no_preheat"):
        assert k in d, k
    assert d["steps"] == 20 and d["warmup"] == 5 and d["roofline"]["bound"] == "mfma"
    extras = json.load(open(ex))["extras"]
    for k in ("f32", "also", "gru", "config4_shard", "config5", "latency", "latency_chunk80", "rooflines_other", "audio_to_posteriors"):
        assert k in extras, k
    assert d["summary"]["mdtc_h64"]["value"] == extras["also"]["value"]

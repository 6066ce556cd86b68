"""CPU: the shape of bench.py's stdout -- the contract line is LAST, compact, the only line that starts with '{'; the verbose
records go to the side file and to `extras <key> = ...` lines.  (Round 5's single 22.5 KB line could not be parsed by the
driver; the GPU run of the same check is tests/test_hip_bench.py.)"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _line(**more):
    roof = bench.mfma_roofline("ds_tcn_h256", 1024, 0.2, "f16x3")
    roof.update(kernel="ds256_g16_kernel<7,true,true,false>", traffic=207_100_000, traffic_unit="x" * 900,
                traffic_source="profiles/r06_x.txt", kernel_avg_ms_rocprof=0.197)
    out = {"metric": "m", "value": 5.0e6, "unit": "utts/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.2,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "w"}, "roofline": bench.compact_roofline(roof),
           "cpu_baseline": {"value": 1.0, "unit": "utts/s", "cores": 1, "kind": "port", "sample": "s"}}
    out.update(more)
    return out, roof


def test_emit_puts_the_compact_line_last(tmp_path):
    out, roof = _line()
    extra = {"roofline_verbose": roof, "big": {"rows": ["y" * 100] * 300}}          # ~30 KB of extras
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(out, extra, str(tmp_path / "x" / "extras.json"))
    lines = buf.getvalue().splitlines()
    assert lines[-1].startswith('{"metric"') and len(lines[-1]) < bench.MAX_LINE_BYTES
    assert sum(ln.startswith("{") for ln in lines) == 1
    d = json.loads(lines[-1])
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes_per_launch"):
        assert k in d["roofline"], k
    assert "traffic_unit" not in d["roofline"]
    side = json.load(open(tmp_path / "x" / "extras.json"))
    assert side["extras"]["big"]["rows"][0] == "y" * 100 and side["contract_line"]["value"] == d["value"]
    tail = buf.getvalue().encode()[-8192:].decode()
    assert json.loads(tail[tail.rindex('{"metric"'):]) == d                          # survives the driver's 8 KB tail


def test_emit_refuses_a_line_that_outgrew_the_tail(tmp_path):
    out, _ = _line(bloat="z" * 7000)
    with pytest.raises(AssertionError):
        with redirect_stdout(io.StringIO()):
            bench.emit(out, {}, str(tmp_path / "e.json"))

"""C++ runtime over the C ABI (runtime/: wekws::KeywordSpotting, wenet::FeaturePipeline, kws_main).
CPU: it builds and rejects a wrong command line like the reference's kws_main.  GPU: wav file -> kws_main ->
per-frame probabilities must match  oracle fbank (C port of the reference front-end) -> oracle streaming forward
with the same chunking, i.e. the same numbers the reference runtime would print for this model and audio."""
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import fbank_oracle, kws_oracle
from wekws_amd import pack
from wekws_amd.utils import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KWS_MAIN = os.path.join(ROOT, "runtime", "build", "kws_main")
MODEL_CONVERT = os.path.join(ROOT, "runtime", "build", "model_convert")
FP_TEST = os.path.join(ROOT, "runtime", "build", "feature_pipeline_test")
REF_ORT = "/root/reference/runtime/android/app/src/main/assets/kws.ort"


def build_runtime():
    subprocess.run(["make", "-C", os.path.join(ROOT, "runtime")], check=True, capture_output=True)
    assert os.path.exists(KWS_MAIN) and os.path.exists(MODEL_CONVERT)


def write_wav(path, pcm_int16, rate=16000):
    data = np.asarray(pcm_int16, dtype="<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * 2, 2, 16))
        f.write(b"data" + struct.pack("<I", len(data)) + data)


def test_builds_and_rejects_bad_usage():
    build_runtime()
    r = subprocess.run([KWS_MAIN, "40"], capture_output=True, text=True)
    assert r.returncode != 0 and "Usage: kws_main fbank_dim(int) batch_size(int)" in r.stderr
    r = subprocess.run([KWS_MAIN, "40", "80", "/nonexistent/model", "/nonexistent.wav"], capture_output=True, text=True)
    assert r.returncode != 0 and "cannot read" in r.stderr


EXPORTED = ["ds_tcn_h64_cmvn", "tcn_h32", "mdtc_small", "mdtc_small_global12", "fsmn_small_ctc", "ds_tcn_h40_nopre_cmvn",
            "fsmn_lorder1_ctc"]


def _python_packed(path):
    from wekws_amd.utils.onnx_lower import load_model_file
    cfg, sd, info = load_model_file(path)
    if info["softmax"]:
        cfg["_exported_softmax"] = True
    return pack.pack(cfg, sd)


@pytest.mark.parametrize("src", [os.path.join(ROOT, "tests", "golden", "onnx", n + ".onnx") for n in EXPORTED] + [REF_ORT],
                         ids=EXPORTED + ["reference_kws_ort"])
def test_cpp_model_reader_equals_python_reader(tmp_path, src):
    """runtime/kws/model_file.cc (what KeywordSpotting(model_path) uses for .onnx / .ort files) must produce the very
    descriptor and folded blob the Python reader + packer produce: same 16 ints, same float32 bits."""
    if not os.path.exists(src):
        pytest.skip("reference tree not present (GPU box)")
    build_runtime()
    out = str(tmp_path / "m.wekwship")
    r = subprocess.run([MODEL_CONVERT, src, out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    desc, blob = pack.load_packed(out)
    pdesc, pblob = _python_packed(src)
    assert {k: int(desc[k]) for k in pack.DESC_FIELDS} == {k: int(pdesc[k]) for k in pack.DESC_FIELDS}
    assert blob.dtype == np.float32 and np.array_equal(blob.view(np.uint32), pblob.view(np.uint32))
    # a packed file passes through unchanged
    out2 = str(tmp_path / "m2.wekwship")
    assert subprocess.run([MODEL_CONVERT, out, out2], capture_output=True).returncode == 0
    assert open(out, "rb").read() == open(out2, "rb").read()


def test_cpp_model_reader_refuses_bad_files(tmp_path):
    build_runtime()
    src = os.path.join(ROOT, "tests", "golden", "onnx", "tcn_h32.onnx")
    data = open(src, "rb").read()
    cases = {"truncated.onnx": data[:len(data) // 2], "empty.onnx": b"", "junk.ort": b"\x10\x00\x00\x00ORTM" + b"\xff" * 64,
             "short.wekwship": b"WEKWSHIP" + b"\x00" * 20}
    # metadata that contradicts the graph (appended metadata_props entry: later cache_len wins)
    entry = b"\x0a\x09cache_len\x12\x03104"              # StringStringEntryProto{key=1, value=2}
    cases["badmeta.onnx"] = data + b"\x72" + bytes([len(entry)]) + entry      # ModelProto.metadata_props = 14
    for name, blob in cases.items():
        p = str(tmp_path / name)
        open(p, "wb").write(blob)
        r = subprocess.run([MODEL_CONVERT, p, str(tmp_path / "o")], capture_output=True, text=True)
        assert r.returncode == 2 and r.stderr.strip(), name
    r = subprocess.run([MODEL_CONVERT, str(tmp_path / "missing.onnx"), str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 2 and "cannot read" in r.stderr


def test_model_readers_survive_corrupted_files(tmp_path):
    """Model files are untrusted input: random byte corruption of real exports must end in a clean refusal or a valid
    model -- never a crash (C++: exit code 0 or 2, no signal) and never an exception other than ModelFileError (Python)."""
    from wekws_amd.utils import onnx_model
    from wekws_amd.utils.onnx_lower import lower
    build_runtime()
    rng = np.random.default_rng(0)
    srcs = [os.path.join(ROOT, "tests", "golden", "onnx", n + ".onnx") for n in ("tcn_h32", "fsmn_small_ctc", "mdtc_small")]
    from tests.helpers import build_tiny_ort
    blobs = [open(s, "rb").read() for s in srcs] + [build_tiny_ort()[0]]
    outcomes = {0: 0, 2: 0}
    for trial in range(120):
        data = bytearray(blobs[trial % len(blobs)])
        for _ in range(int(rng.integers(1, 6))):
            kind = rng.integers(0, 3)
            pos = int(rng.integers(0, len(data)))
            if kind == 0:
                data[pos] = int(rng.integers(0, 256))
            elif kind == 1:
                del data[pos:pos + int(rng.integers(1, 64))]
            else:
                data[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 16)), dtype=np.uint8))
        p = str(tmp_path / "fuzz.bin")
        open(p, "wb").write(bytes(data))
        r = subprocess.run([MODEL_CONVERT, p, str(tmp_path / "o")], capture_output=True, text=True, timeout=60)
        assert r.returncode in (0, 2), (trial, r.returncode, r.stderr[-200:])
        outcomes[r.returncode] += 1
        try:
            g = onnx_model.parse_ort(bytes(data)) if bytes(data[4:8]) == b"ORTM" else onnx_model.parse_onnx(bytes(data))
            lower(g)
        except onnx_model.ModelFileError:
            pass
    assert outcomes[2] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name,chunk", [("ds_tcn_h64", 80), ("mdtc_small", 33), ("ds_tcn_h256", 10), ("gru_2x128", 25),
                                        ("ds_tcn_h256", 1), ("mdtc_h64", 200), ("tcn_h64", 113), ("ds_tcn_h256", 80)])
def test_kws_main_matches_oracle(tmp_path, name, chunk):
    build_runtime()
    cfg = dict(synth.MODEL_CONFIGS[name])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
    desc, blob = pack.pack(cfg, sd)
    model = str(tmp_path / "model.wekwship")
    pack.save_packed(model, desc, blob)
    pcm = (synth.synth_pcm(1, 40000, seed=11, kind="noise")[0] * 0.5 + synth.synth_pcm(1, 40000, kind="sine")[0])
    pcm = np.clip(np.round(pcm), -32768, 32767).astype(np.int16)
    wav = str(tmp_path / "t.wav")
    write_wav(wav, pcm)
    r = subprocess.run([KWS_MAIN, "40", str(chunk), model, wav], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    rows = [l.split() for l in r.stdout.strip().splitlines()]
    assert all(w[0] == "frame" and w[2] == "prob" for w in rows)
    assert [int(w[1]) for w in rows] == list(range(len(rows)))  # reference numbering: offset + i
    got = np.array([[float(v) for v in w[3:]] for w in rows], np.float32)

    feats = fbank_oracle.fbank(pcm.astype(np.float32), 40)
    T = feats.shape[0]
    chunks = [chunk] * (T // chunk) + ([T % chunk] if T % chunk else [])
    ref, _ = kws_oracle.forward_streaming(cfg, sd, feats[None], chunks, None)
    assert got.shape == ref[0].shape == (T, cfg["output_dim"])
    assert float(np.abs(got - ref[0]).max()) <= 1e-4


@pytest.mark.gpu
def test_exported_onnx_to_kws_main(tmp_path):
    """The reference's deployment flow with the middle piece swapped: export_onnx.py's .onnx -> export_packed
    --exported -> kws_main.  Printed probabilities == the exported graph (numpy executor) run chunk by chunk with the
    carried cache, which is what keyword_spotting.cc:56-95 does with onnxruntime."""
    from oracle import onnx_graph_oracle
    from wekws_amd.bin import export_packed
    from wekws_amd.utils import onnx_model
    build_runtime()
    src = os.path.join(ROOT, "tests", "golden", "onnx", "ds_tcn_h64_cmvn.onnx")
    model = str(tmp_path / "model.wekwship")
    export_packed.main(["--exported", src, "--output", model])
    pcm = (synth.synth_pcm(1, 24000, seed=5, kind="noise")[0] * 0.5 + synth.synth_pcm(1, 24000, kind="sine")[0])
    pcm = np.clip(np.round(pcm), -32768, 32767).astype(np.int16)
    wav = str(tmp_path / "t.wav")
    write_wav(wav, pcm)
    chunk = 40
    r = subprocess.run([KWS_MAIN, "40", str(chunk), model, wav], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = np.array([[float(v) for v in l.split()[3:]] for l in r.stdout.strip().splitlines()], np.float32)
    g = onnx_model.load_graph(src)
    feats = fbank_oracle.fbank(pcm.astype(np.float32), 40)
    cache = np.zeros((1, int(g.meta["cache_dim"]), int(g.meta["cache_len"])), np.float32)   # keyword_spotting.cc:47-53
    ref = []
    for t in range(0, feats.shape[0], chunk):
        out = onnx_graph_oracle.run(g, dict(input=feats[None, t:t + chunk], cache=cache))
        ref.append(out["output"][0])
        cache = out["r_cache"]
    ref = np.concatenate(ref)
    assert got.shape == ref.shape and float(np.abs(got - ref).max()) <= 1e-4
    # ... and kws_main takes the exporter's file itself, like the reference's kws_main does (model_path = the .onnx)
    r2 = subprocess.run([KWS_MAIN, "40", str(chunk), src, wav], capture_output=True, text=True, timeout=120)
    assert r2.returncode == 0, r2.stderr
    assert r2.stdout == r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "i16", "mixed"])
def test_feature_pipeline_uneven_pushes(tmp_path, mode):
    """the framing / leftover rule of FeaturePipeline::AcceptWaveform (feature_pipeline.cc:30-47)
    through the HIP pipeline under uneven pushes, through the float overload, the int16 overload (int16 upload, widened
    on the GPU) and a mix of both.  Every push pattern must give the frames of the whole signal -- the goldens recorded
    from the compiled reference, one of them with the reference itself fed in two pushes -- and the SAME bits whatever
    the pattern and overload (frames are functions of the same samples)."""
    from tests.golden.fbank_cases import FBANK_CASES, fbank_input
    build_runtime()
    assert os.path.exists(FP_TEST)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "fbank_golden.npz"))
    first = None
    for cname, patterns in (("noise_1s_two_pushes", [[16000], [5000, 11000], [1, 399, 160, 161, 7000, 3], [240, 160], [399, 1, 15600]]),
                            ("noise_ragged_tail", [[16097], [400, 15697], [16096, 1], [777]]),
                            ("noise_exactly_one_frame", [[400], [399, 1], [1]])):
        case = [c for c in FBANK_CASES if c["name"] == cname][0]
        pcm = fbank_input(case)[0]
        i16 = np.clip(np.round(pcm), -32768, 32767).astype("<i2")
        assert np.array_equal(i16.astype(np.float32), pcm)          # the synthetic PCM is integral
        raw = str(tmp_path / (cname + ".raw"))
        i16.tofile(raw)
        ref = gold[cname][0]
        first = None
        for sizes in patterns:
            out = str(tmp_path / "o.f32")
            r = subprocess.run([FP_TEST, "40", mode, raw, out] + [str(v) for v in sizes], capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stderr
            got = np.fromfile(out, np.float32).reshape(-1, 40)
            assert got.shape == ref.shape, (cname, sizes, got.shape)
            assert float(np.abs(got - ref).max()) <= 1e-4, (cname, sizes)
            if first is None:
                first = got
            assert np.array_equal(got, first), (cname, sizes)       # push pattern / overload do not change a bit


def test_packed_files_written_under_abi_1_still_load(tmp_path):
    """ABI version 2 (round 5) added entry points, not descriptor fields or blob sections: a `.wekwship` file written by an ABI-1
    build must keep loading -- in Python (pack.load_packed) and in the C++ runtime's reader (a rocprofv3 trace of kws_main on such a
    file is how the C++ side's rejection was found)."""
    build_runtime()
    cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h64"])
    desc, blob = pack.pack(cfg, synth.synth_state_dict(pack.model_spec(cfg), 3))
    old = str(tmp_path / "old.wekwship")
    pack.save_packed(old, dict(desc, abi_version=1), blob)
    d2, b2 = pack.load_packed(old)
    assert d2["abi_version"] == pack.ABI_VERSION and np.array_equal(b2, blob)
    new = str(tmp_path / "new.wekwship")
    r = subprocess.run([MODEL_CONVERT, old, new], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d3, b3 = pack.load_packed(new)
    assert d3["abi_version"] == pack.ABI_VERSION and np.array_equal(b3, blob)


STREAM_TEST = os.path.join(ROOT, "runtime", "build", "stream_kws_test")


@pytest.mark.gpu
@pytest.mark.parametrize("name,chunk", [("ds_tcn_h64", 80), ("mdtc_small", 10), ("gru_2x128", 10)])
def test_producer_and_consumer_threads_equal_the_offline_run(tmp_path, name, chunk):
    """wenet::FeaturePipeline is a hand-off between a producer thread (AcceptWaveform) and a consumer thread (Read)
    (runtime/core/frontend/feature_pipeline.h:48-54; the reference's stream_kws_main.cc:63-93).  Four streams at once, each
    with its own HIP stream, FeaturePipeline and KeywordSpotting, a producer pushing uneven PCM pieces and a consumer blocking in
    Read(chunk) -> Forward: every stream prints exactly what the single-threaded offline tool prints."""
    build_runtime()
    cfg = dict(synth.MODEL_CONFIGS[name])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
    desc, blob = pack.pack(cfg, sd)
    model = str(tmp_path / "model.wekwship")
    pack.save_packed(model, desc, blob)
    pcm = (synth.synth_pcm(1, 48000, seed=21, kind="noise")[0] * 0.5 + synth.synth_pcm(1, 48000, kind="sine")[0])
    pcm = np.clip(np.round(pcm), -32768, 32767).astype(np.int16)
    wav = str(tmp_path / "t.wav")
    write_wav(wav, pcm)
    off = subprocess.run([KWS_MAIN, "40", str(chunk), model, wav], capture_output=True, text=True, timeout=120)
    assert off.returncode == 0, off.stderr
    thr = subprocess.run([STREAM_TEST, "40", str(chunk), model, wav, "4", "1600", "37", "4001", "160", "799", "12000"],
                         capture_output=True, text=True, timeout=180)
    assert thr.returncode == 0, thr.stderr
    assert thr.stdout == off.stdout and len(off.stdout.splitlines()) == 1 + (48000 - 400) // 160

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built_artifacts():
    """The compiled pieces are git-ignored: a fresh checkout builds them once (what __graft_entry__.build() does --
    hipcc cross-compiles without a GPU).  Only MISSING artefacts trigger a build: a shipped tree (the GPU box) is used
    as it is, whatever the file times say."""
    import subprocess
    need = {os.path.join("wekws_amd", "csrc"): os.path.join("wekws_amd", "lib", "libwekws_hip.so"),
            "oracle": os.path.join("oracle", "_build", "libfbank_oracle.so")}
    for sub, artefact in need.items():
        if not os.path.exists(os.path.join(ROOT, artefact)):
            subprocess.run(["make", "-C", os.path.join(ROOT, sub), "-j", "8"], check=False, capture_output=True)
    # the reference's shipped trained model (an ORT file), converted once for test_reference_android_asset_hip: only
    # where the reference tree exists (the build container); build/ is git-ignored but travels to the GPU box
    asset = "/root/reference/runtime/android/app/src/main/assets/kws.ort"
    if os.path.exists(asset) and not os.path.exists(os.path.join(ROOT, "build", "ref_asset", "expect.npz")):
        subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "make_ref_asset.py")], check=False,
                       capture_output=True)
    yield


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "model_golden.npz"))


@pytest.fixture(scope="session")
def scale_golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "scale_golden.npz"))


@pytest.fixture(scope="session")
def error_report():
    """Measured errors of the parity tests, written to gpurun_out/parity_errors.json when the session ends (the bar
    is an assertion; the margin is worth knowing: VERDICT r1)."""
    import json
    rec = {}
    yield rec
    if rec:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_errors.json"), "w") as f:
            json.dump(rec, f, indent=1, sort_keys=True)

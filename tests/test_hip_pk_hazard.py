"""GPU: the operand-select forms the library's packed-f32 code relies on are clean beside MFMA waves (wekws_amd/csrc/pk_safe.hip.h).

tools/probe/pk_opsel_probe4.hip runs every op_sel / op_sel_hi combination of v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 in waves that
share their SIMDs with MFMA waves and compares with plain VOP3 instructions.  The forms with op_sel:[0,1,..] are the known-bad ones
(reported, not asserted: a part without the hazard is welcome); every OTHER form -- the ones head_fma4 and the fbank butterflies use
among them -- must not produce a single wrong value."""
import os
import re
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def test_safe_packed_f32_forms_beside_mfma_waves(tmp_path):
    exe = tmp_path / "pk_probe4"
    subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", os.path.join(ROOT, "tools", "probe", "pk_opsel_probe4.hip"), "-o", str(exe)],
                   check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=600).stdout
    sections = re.split(r"(?m)^== ", out)[1:]
    assert len(sections) == 2 and sections[1].startswith("2 MFMA waves"), out[:400]
    rows = re.findall(r"low wrong\s+(\d+) \(lanes 48..63:\s+(\d+);[^)]*\)\s+high wrong\s+(\d+)\s+(v_pk_\w+ .*)", sections[1])
    assert len(rows) == 96
    known_bad, unexpected = [], []
    for lo, lo3, hi, form in rows:
        wrong = int(lo) + int(hi)
        if re.search(r"op_sel:\[0,1", form):
            known_bad.append((form, int(lo), int(lo3), int(hi)))
        elif wrong:
            unexpected.append((form, int(lo), int(hi)))
    assert not unexpected, unexpected[:4]
    # without MFMA neighbours nothing is ever wrong
    assert all(int(lo) + int(hi) == 0 for lo, _, hi, _ in re.findall(
        r"low wrong\s+(\d+) \(lanes 48..63:\s+(\d+);[^)]*\)\s+high wrong\s+(\d+)\s+(v_pk_\w+ .*)", sections[0]))
    hit = [k for k in known_bad if k[1]]
    print(f"\nop_sel:[0,1,..] forms: {len(hit)} of {len(known_bad)} returned wrong low halves beside MFMA waves"
          + (f" (all in lanes 48..63: {all(k[1] == k[2] and k[3] == 0 for k in hit)})" if hit else " -- this part does not show the hazard"))

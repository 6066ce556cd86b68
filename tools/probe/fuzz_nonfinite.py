"""Development tool (GPU box): tests/test_hip_nonfinite.py::test_nonfinite_fuzz over seeds the suite does not run.
    python tools/probe/fuzz_nonfinite.py LO HI [minutes]"""
import sys
import time

sys.path.insert(0, '.')
import tests.test_hip_nonfinite as t

lo, hi = int(sys.argv[1]), int(sys.argv[2])
budget = float(sys.argv[3]) * 60 if len(sys.argv) > 3 else 1e9
t0, bad, n = time.time(), 0, 0
for seed in range(lo, hi):
    if time.time() - t0 > budget:
        break
    for prec in ("f16x3", "f32", "f16"):
        n += 1
        try:
            t.test_nonfinite_fuzz(seed, prec)
        except AssertionError as e:
            bad += 1
            print("FAIL seed", seed, prec, str(e)[:700], flush=True)
        except Exception as e:
            bad += 1
            print("ERROR seed", seed, prec, repr(e)[:700], flush=True)
print(f"non-finite fuzz: seeds {lo} .. {seed}: {n} (seed, precision) runs x 4 configurations, {bad} failures ({time.time() - t0:.0f} s)", flush=True)

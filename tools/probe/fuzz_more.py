"""Development tool: tests/test_hip_parity.py::test_random_model_shapes_against_the_oracle over many more seeds than the suite runs\n(a few hundred random configurations per minute on the GPU box)."""
import sys, traceback
sys.path.insert(0, '.')
import tests.test_hip_parity as t
bad = 0
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 60)
for seed in range(lo, hi):
    try:
        t.test_random_model_shapes_against_the_oracle(seed)
    except AssertionError as e:
        bad += 1
        print("FAIL seed", seed, str(e)[:600], flush=True)
    except Exception as e:
        bad += 1
        print("ERROR seed", seed, repr(e)[:600], flush=True)
print("model-shape fuzz, seeds", lo, "..", hi - 1, ":", bad, "failures")

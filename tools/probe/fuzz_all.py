"""Development tool (GPU box): the suite's seeded random tests over seed ranges the suite does not run.
    python tools/probe/fuzz_all.py LO HI [minutes [names]]   -- the fuzz tests (all, or those whose name contains one of the comma-separated
    `names`) for seeds LO .. HI-1; every test gets an equal share of `minutes`"""
import sys
import time

sys.path.insert(0, '.')
import tests.test_hip_det as t_det
import tests.test_hip_fbank as t_fb
import tests.test_hip_parity as t_par
import tests.test_hip_splice as t_spl
import tests.test_hip_topk as t_top

lo, hi = int(sys.argv[1]), int(sys.argv[2])
budget = float(sys.argv[3]) * 60 if len(sys.argv) > 3 else 1e9
tests = [("model shapes", t_par.test_random_model_shapes_against_the_oracle), ("fsmn shapes", t_par.test_random_fsmn_shapes_against_the_oracle),
         ("fbank framings", t_fb.test_random_framings_against_the_c_oracle), ("det shapes", t_det.test_random_det_shapes),
         ("splice shapes", t_spl.test_random_splice_shapes), ("topk shapes", t_top.test_random_topk_shapes)]
if len(sys.argv) > 4:
    tests = [t for t in tests if any(k in t[0] for k in sys.argv[4].split(","))]
t0 = time.time()
for k, (name, fn) in enumerate(tests):
    bad = n = 0
    for seed in range(lo, hi):
        if time.time() - t0 > budget * (k + 1) / len(tests):
            break
        n += 1
        try:
            fn(seed)
        except AssertionError as e:
            bad += 1
            print("FAIL", name, "seed", seed, str(e)[:500], flush=True)
        except Exception as e:
            bad += 1
            print("ERROR", name, "seed", seed, repr(e)[:500], flush=True)
    print(f"{name}: seeds {lo} .. {lo + n - 1}: {bad} failures ({time.time() - t0:.0f} s)", flush=True)

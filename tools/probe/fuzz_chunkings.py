"""Development tool (GPU box): tests/test_hip_parity.py::test_random_chunkings_cross_the_kernel_families under other seeds
(WEKWS_FUZZ_SEED = LO .. HI-1), every model x {default, f32}, for at most `minutes`.
    python tools/probe/fuzz_chunkings.py LO HI [minutes]"""
import os
import sys
import time

sys.path.insert(0, '.')
import tests.test_hip_parity as t

lo, hi = int(sys.argv[1]), int(sys.argv[2])
budget = float(sys.argv[3]) * 60 if len(sys.argv) > 3 else 1e9
names = ["ds_tcn_h256", "ds_tcn_h64", "mdtc_h64", "mdtc_h64_80d", "mdtc_small", "tcn_h64", "gru_2x128", "gru_1x128", "ds_tcn_h64_ctc20",
         "ds_tcn_h256_ctc300", "fsmn_ctc300", "fsmn_small"]
t0, bad, n = time.time(), 0, 0
for seed in range(lo, hi):
    os.environ["WEKWS_FUZZ_SEED"] = str(seed)
    for name in names:
        for prec in ("default", "f32"):
            if time.time() - t0 > budget:
                break
            n += 1
            try:
                t.test_random_chunkings_cross_the_kernel_families(name, prec)
            except AssertionError as e:
                bad += 1
                print("FAIL seed", seed, name, prec, str(e)[:500], flush=True)
            except Exception as e:
                bad += 1
                print("ERROR seed", seed, name, prec, repr(e)[:500], flush=True)
    print(f"seed {seed} done: {n} (model, precision) runs x 16 random chunkings so far, {bad} failures ({time.time() - t0:.0f} s)", flush=True)
    if time.time() - t0 > budget:
        break

"""Development tool, BUILD CONTAINER ONLY (executes the reference's own context_expansion / frame_skip, lifted from
/root/reference/wekws/dataset/init_dataset.py like tests/golden/make_splice_golden.py does): the numpy oracle against them on random
shapes INCLUDING utterances no longer than their context -- same output bit for bit, same lengths, IndexError where the reference raises.
    python tools/probe/fuzz_splice_vs_reference.py [n]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import splice_oracle  # noqa: E402
from tests.golden.make_splice_golden import load_reference_functions  # noqa: E402

ctx, skip_fn = load_reference_functions()
rng = np.random.default_rng(20260930)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
raised = degenerate = 0
for it in range(n):
    B, T, F = int(rng.integers(0, 5)), int(rng.integers(0, 12) if rng.random() < 0.7 else rng.integers(12, 140)), int(rng.choice([1, 3, 8]))
    left, right, skip = int(rng.integers(0, 7)), int(rng.integers(0, 9)), int(rng.integers(1, 6))
    x = rng.standard_normal((B, T, F)).astype(np.float32)
    lens = rng.integers(0, T + 1, size=B).astype(np.int32)
    what = (it, B, T, F, left, right, skip)
    try:
        s = skip_fn(ctx({"feats": torch.from_numpy(x.copy()), "feats_lengths": torch.from_numpy(lens.copy())}, left=left, right=right), skip_rate=skip)
        want, wlens = s["feats"].contiguous().numpy(), s["feats_lengths"].numpy()
    except IndexError:
        raised += 1
        try:
            splice_oracle.splice_skip(x, left, right, skip)
        except IndexError:
            assert left >= 1 and left >= T, what          # the rule the product implements (wekws_hip_splice: EINVAL)
            continue
        raise AssertionError(("the reference raises, the oracle does not", what))
    assert not (left >= 1 and left >= T), ("the product would refuse, the reference does not", what)
    got = splice_oracle.splice_skip(x, left, right, skip)
    degenerate += T < right
    assert got.shape == want.shape and np.array_equal(got, want), what
    assert np.array_equal(splice_oracle.lengths(lens, right, skip), wlens), what
    kept = T - right if T >= right else max(2 * T - right, 0)          # wekws_hip_splice_frames
    assert got.shape[1] == (kept + skip - 1) // skip, what
print(f"{n} random shapes: oracle == reference everywhere ({raised} where both raise IndexError, {degenerate} with T < right)")

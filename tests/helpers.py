"""Shared by the CPU and GPU tests: seeded weights / inputs for a golden case (no reference import)."""
import numpy as np

from tests.golden.cases import CASES, case_config, case_in_cache, case_input  # noqa: F401
from wekws_amd import pack
from wekws_amd.utils import synth


def case_weights(case):
    cfg = case_config(case)
    sd = synth.synth_state_dict(pack.model_spec(cfg), case["wseed"])
    return cfg, sd


def by_name(name):
    for c in CASES:
        if c["name"] == name:
            return c
    raise KeyError(name)


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) if np.size(a) else 0.0

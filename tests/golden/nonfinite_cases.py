"""Non-finite inputs: what the reference returns when a feature or a carried cache holds NaN / +Inf / -Inf
(torch.relu(nan) = nan, IEEE arithmetic everywhere else: wekws/model/tcn.py:101-114, kws_model.py:65-76).  Shared by the golden
generator (runs the live reference) and the tests (oracle / HIP path).  No reference import here."""
import numpy as np

from tests.golden.cases import case_config, case_input, case_in_cache  # noqa: F401  (re-exported for the tests)

VAL = {"nan": np.float32(np.nan), "+inf": np.float32(np.inf), "-inf": np.float32(-np.inf)}


def _c(name, model, B, T, poison, **kw):
    d = dict(name=name, model=model, B=B, T=T, wseed=1234, xseed=5, cmvn=False, chunks=None, cache="empty", softmax=False,
             odim=None, poison=poison)
    d.update(kw)
    return d


def _x(b, t, f, v):
    return ("x", b, t, f, v)


# the standard batch: utterance 0 clean, 1 NaN, 2 +Inf, 3 -Inf, 4 all three at separated frames, 5 clean
def _std(T, idim):
    a, bq, c = T // 3, T // 2, (3 * T) // 4
    return [_x(1, bq, 3 % idim, "nan"), _x(2, a, 7 % idim, "+inf"), _x(3, c, 0, "-inf"),
            _x(4, c, 1 % idim, "nan"), _x(4, a, 5 % idim, "+inf"), _x(4, bq, 2 % idim, "-inf")]


CASES = []
for _m, _idim in (("ds_tcn_h256", 40), ("ds_tcn_h64", 40), ("tcn_h64", 40), ("mdtc_h64", 40), ("mdtc_small", 40),
                  ("mdtc_h64_global12", 40), ("mdtc_small_last12", 40), ("gru_2x128", 40), ("fsmn_small", 40),
                  ("ds_tcn_h256_ctc300", 40)):
    _cache = "zeros" if _m.startswith("gru") else "empty"
    CASES.append(_c(f"{_m}/full", _m, 6, 98, _std(98, _idim), cache=_cache))
CASES += [
    # the last frame / the first frame / a frame right of the last block's reach
    _c("ds_tcn_h256/edges", "ds_tcn_h256", 4, 98, [_x(0, 97, 39, "nan"), _x(1, 0, 0, "+inf"), _x(2, 96, 1, "-inf")]),
    _c("mdtc_h64/edges", "mdtc_h64", 4, 98, [_x(0, 97, 39, "nan"), _x(1, 0, 0, "+inf"), _x(2, 96, 1, "-inf")]),
    # streaming: the poison travels to the later chunks through the carried cache (stream_kws_ctc.py:486-487)
    _c("ds_tcn_h256/stream10", "ds_tcn_h256", 3, 60, [_x(1, 23, 4, "nan"), _x(2, 35, 9, "+inf")], chunks=[10] * 6),
    _c("ds_tcn_h256/stream_mixed", "ds_tcn_h256", 2, 98, [_x(1, 12, 4, "-inf")], chunks=[1, 3, 10, 7, 30, 47]),
    _c("ds_tcn_h64/stream80", "ds_tcn_h64", 3, 160, [_x(1, 70, 4, "nan"), _x(2, 100, 9, "+inf")], chunks=[80, 80]),
    _c("mdtc_h64/stream10", "mdtc_h64", 3, 60, [_x(1, 23, 4, "nan"), _x(2, 35, 9, "-inf")], chunks=[10] * 6),
    _c("mdtc_h64/stream80", "mdtc_h64", 3, 160, [_x(1, 70, 4, "+inf"), _x(2, 100, 9, "nan")], chunks=[80, 80]),
    _c("mdtc_small/stream_mixed", "mdtc_small", 2, 98, [_x(1, 5, 4, "nan")], chunks=[1, 7, 10, 80]),
    _c("tcn_h64/stream_mixed", "tcn_h64", 2, 98, [_x(1, 12, 4, "+inf")], chunks=[1, 3, 10, 7, 30, 47]),
    _c("gru_2x128/stream10", "gru_2x128", 3, 60, [_x(1, 23, 4, "nan"), _x(2, 35, 9, "+inf")], chunks=[10] * 6, cache="zeros"),
    _c("fsmn_small/stream_mixed", "fsmn_small", 2, 98, [_x(1, 12, 4, "nan")], chunks=[1, 3, 10, 7, 30, 47]),
    # a poisoned incoming cache (random cache, one element each)
    _c("ds_tcn_h256/cache", "ds_tcn_h256", 4, 30, [("cache", 1, 17, 104, "nan"), ("cache", 2, 200, 3, "+inf"), ("cache", 3, 5, 60, "-inf")],
       cache="random"),
    _c("ds_tcn_h256/cache_T100", "ds_tcn_h256", 3, 100, [("cache", 1, 17, 104, "nan"), ("cache", 2, 200, 40, "+inf")], cache="random"),
    _c("mdtc_h64/cache", "mdtc_h64", 4, 30, [("cache", 1, 17, 243, "nan"), ("cache", 2, 60, 3, "+inf"), ("cache", 3, 5, 100, "-inf")],
       cache="random"),
    _c("gru_2x128/h0", "gru_2x128", 4, 20, [("cache", 1, 1, 17, "nan"), ("cache", 0, 2, 5, "+inf"), ("cache", 1, 3, 100, "-inf")],
       cache="random"),                                          # GRU cache is (L, B, H): index = (layer, b, unit)
    # long inputs: consecutive 112-frame tiles hand the context over inside one call
    _c("ds_tcn_h256/T250", "ds_tcn_h256", 3, 250, [_x(1, 100, 3, "nan"), _x(2, 230, 3, "+inf")]),
    _c("mdtc_h64_global12/T250", "mdtc_h64_global12", 3, 250, [_x(1, 100, 3, "nan"), _x(2, 230, 3, "-inf")]),
    # forward_softmax on a CTC head
    _c("ds_tcn_h64_ctc20/softmax", "ds_tcn_h64_ctc20", 4, 40, [_x(1, 20, 3, "nan"), _x(2, 10, 3, "+inf"), _x(3, 30, 3, "-inf")],
       softmax=True),
    # CMVN folded into the first Linear: (Inf - mean) * istd
    _c("ds_tcn_h256/cmvn", "ds_tcn_h256", 4, 98, [_x(1, 50, 3, "nan"), _x(2, 30, 7, "+inf"), _x(3, 60, 0, "-inf")], cmvn=True, xseed=1),
]


def poisoned_input(case, cfg=None):
    """(x, in_cache) of the case with the poison applied (copies)."""
    cfg = cfg or case_config(case)
    x = case_input(case).copy()
    cache = case_in_cache(case, cfg)
    cache = None if cache is None else cache.copy()
    for p in case["poison"]:
        if p[0] == "x":
            _, b, t, f, v = p
            x[b, t, f] = VAL[v]
        else:
            _, i, j, k, v = p
            cache[i, j, k] = VAL[v]
    return x, cache


def classify(a):
    """0 finite, 1 NaN, 2 +Inf, 3 -Inf (int8)."""
    a = np.asarray(a)
    c = np.zeros(a.shape, np.int8)
    c[np.isnan(a)] = 1
    c[np.isposinf(a)] = 2
    c[np.isneginf(a)] = 3
    return c

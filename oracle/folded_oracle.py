"""numpy evaluator of the FOLDED weight blob (the layout include/wekws_hip.h documents)  --  TEST
INFRASTRUCTURE, NOT PRODUCT.  It lets the CPU suite check the host packer (BatchNorm / CMVN folding, blob
order, descriptor) against the unfolded oracle without a GPU: pack() -> this evaluator must reproduce
oracle/kws_oracle.forward.  Conv backbones, GRU and FSMN, one-shot (empty cache) only."""
import numpy as np

from oracle import kws_oracle as ko

F32 = np.float32


class _Reader:
    def __init__(self, blob):
        self.b, self.p = np.asarray(blob, F32), 0

    def take(self, *shape):
        n = int(np.prod(shape))
        v = self.b[self.p:self.p + n].reshape(shape)
        self.p += n
        return v


def _fsmn(desc, r, x):
    """FSMN blob (include/wekws_hip.h): the memory block as ONE tap vector per channel (identity path folded in)."""
    C, I, K = desc["hdim"], desc["idim"], desc["odim"]
    A1, A2, D, nt = desc["aux0"], desc["aux1"], desc["num_stack"], desc["kernel_size"] + desc["stack_size"]
    h = ko.linear(np.asarray(x, F32), r.take(A1, I), r.take(A1))
    h = ko.relu(ko.linear(h, r.take(C, A1), r.take(C)))
    B, T, _ = h.shape
    for _ in range(desc["num_layers"]):
        p = np.matmul(h, r.take(D, C).T).astype(F32)
        taps = r.take(D, nt)
        xp = np.concatenate([np.zeros((B, nt - 1, D), F32), p], axis=1)
        m = np.zeros((B, T, D), F32)
        for j in range(nt):
            m += taps[None, None, :, j] * xp[:, j:j + T]
        h = ko.relu(ko.linear(m, r.take(C, D), r.take(C)))
    h = ko.linear(h, r.take(A2, C), r.take(A2))
    y = ko.linear(h, r.take(K, A2), r.take(K))
    assert r.p == r.b.size, "blob not fully consumed"
    return y


def forward(desc, blob, x, mm_dtype=None):
    """mm_dtype=np.float16 restates WEKWS_HIP_PRECISION_F16 (include/wekws_hip.h): both operands of the input Linear
    and of every pointwise convolution are rounded to fp16 before the (fp32-accumulated) product; everything else
    stays float32.  DS-TCN / MDTC only."""
    r = _Reader(blob)
    if desc["backbone"] == 4:
        return _fsmn(desc, r, x)
    q = (lambda a: a) if mm_dtype is None else (lambda a: np.asarray(a, F32).astype(mm_dtype).astype(F32))
    assert mm_dtype is None or desc["backbone"] in (0, 2)
    C, I, K, ks = desc["hdim"], desc["idim"], desc["odim"], desc["kernel_size"]
    W, b = r.take(C, I), r.take(C)
    h = ko.linear(q(np.asarray(x, F32)), q(W), b)
    if desc["preproc_relu"]:
        h = ko.relu(h)
    bb = desc["backbone"]
    if bb == 3:
        sd = {}
        for l in range(desc["num_layers"]):
            sd[f"backbone.weight_ih_l{l}"], sd[f"backbone.weight_hh_l{l}"] = r.take(3 * C, C), r.take(3 * C, C)
            sd[f"backbone.bias_ih_l{l}"], sd[f"backbone.bias_hh_l{l}"] = r.take(3 * C), r.take(3 * C)
        h, _ = ko.gru_forward(dict(backbone=dict(num_layers=desc["num_layers"])), sd, h, None)
    else:
        h = np.transpose(h, (0, 2, 1))
        if bb == 2:
            dils = [1] + [2 ** j for _ in range(desc["num_stack"]) for j in range(desc["stack_size"])]
        else:
            dils = [2 ** i for i in range(desc["num_layers"])]
        z = None
        for bi, d in enumerate(dils):
            u, _ = ko.causal_concat(h, None, (ks - 1) * d)
            if bb == 0:
                a = ko.relu(ko.depthwise_conv(u, r.take(C, 1, ks), r.take(C), d))
                h = ko.relu(ko.pointwise_conv(q(a), q(r.take(C, C, 1)), r.take(C))) + h
            elif bb == 1:
                h = ko.relu(ko.full_conv(u, r.take(C, C, ks), r.take(C), d)) + h
            else:
                a = ko.depthwise_conv(u, r.take(C, 1, ks), r.take(C), d)
                a = ko.relu(ko.pointwise_conv(q(a), q(r.take(C, C, 1)), r.take(C)))
                h = ko.relu(ko.pointwise_conv(q(a), q(r.take(C, C, 1)), r.take(C)) + h)
                if bi > 0 and (bi - 1) % desc["stack_size"] == desc["stack_size"] - 1:
                    z = h.copy() if z is None else z + h
        h = np.transpose(z if bb == 2 else h, (0, 2, 1))
    if desc["head"] == 0:
        y = ko.linear(h, r.take(K, C), r.take(K))
    elif desc["head"] in (1, 2):
        m = h.mean(axis=1, dtype=F32) if desc["head"] == 1 else h[:, -1, :]
        hh = desc["head_hidden"]
        m = ko.relu(ko.linear(m, r.take(hh, C), r.take(hh)))
        y = ko.linear(m, r.take(K, hh), r.take(K))
    else:
        y = h
    assert r.p == r.b.size, "blob not fully consumed"
    return ko.sigmoid(y) if desc["activation"] == 1 else y

"""TEST / BASELINE INFRASTRUCTURE, NOT PRODUCT -- the reference's CPU path restated on torch's own CPU operators.

The reference's CPU inference *is* PyTorch: KWSModel.forward (wekws/model/kws_model.py:65-76) dispatches to ATen's
linear / conv1d / batch_norm / sigmoid kernels (oneDNN + OpenMP).  /root/reference does not travel to the GPU box, but
torch does -- so for the `cpu_baseline` leg of bench.py this module issues the same ATen calls, in the same order, on
the same shapes as the reference modules would (functional form, weights taken from a reference-named state_dict):
the time it measures is the reference's CPU kernel time, which the numpy oracle's is not.

Conv backbones with a per-frame linear head (the benchmarked recipes: DS-TCN, TCN, MDTC) and the GRU (torch.nn.GRU, the
very module the reference builds: kws_model.py:128-133), one-shot or streaming with a carried cache (bench.py's CPU
per-frame latency).  Pinned like the numpy oracle: tests/test_oracle.py runs it over the live-reference goldens of those
recipes, streaming traces included.
Only tests/ and bench.py's cpu_baseline may import it.
"""
import torch
import torch.nn.functional as F


def _bn(x, sd, p):                       # nn.BatchNorm1d, eval (tcn.py:81,108,111 ; mdtc.py:47,86,92)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, 1e-5)


def _left_ctx(x, pad, cache, off):       # Block.forward: F.pad with an empty cache, else cat (tcn.py:49-52 ; mdtc.py:108-111)
    if cache is None:
        return F.pad(x, (pad, 0), "constant", 0.0)
    return torch.cat((cache[:, :, off:off + pad], x), dim=2)


_GRU_CACHE = {}


def _gru(cfg, sd, hdim):
    """torch.nn.GRU(hdim, hdim, num_layers, batch_first=True) carrying the state_dict's backbone.* tensors."""
    key = (id(sd), hdim, cfg["backbone"]["num_layers"])
    if key not in _GRU_CACHE:
        g = torch.nn.GRU(hdim, hdim, num_layers=cfg["backbone"]["num_layers"], batch_first=True)
        g.load_state_dict({k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")})
        _GRU_CACHE.clear()
        _GRU_CACHE[key] = g.eval()
    return _GRU_CACHE[key]


@torch.no_grad()
def forward(cfg, sd, x, in_cache=None):
    """cfg: configs['model']; sd: {name: torch.Tensor}; x: (B, T, idim) float32 CPU tensor; in_cache: None (the
    reference's empty-cache call) or the carried cache -> (y, out_cache)."""
    if "global_cmvn.mean" in sd:                                             # cmvn.py:45-48
        x = x - sd["global_cmvn.mean"]
        if cfg.get("cmvn", {}).get("norm_var", True):
            x = x * sd["global_cmvn.istd"]
    h = F.relu(F.linear(x, sd["preprocessing.out.0.weight"], sd["preprocessing.out.0.bias"]))   # subsampling.py:53-57
    bb = cfg["backbone"]
    if bb["type"] == "gru":                                                  # kws_model.py:128-133, forward :70-74
        out, hn = _gru(cfg, sd, h.size(2))(h) if in_cache is None else _gru(cfg, sd, h.size(2))(h, in_cache)
        y = F.linear(out, sd["classifier.linear.weight"], sd["classifier.linear.bias"])
        if "classifier" not in cfg and cfg.get("activation", {}).get("type") != "identity":
            y = torch.sigmoid(y)
        return y, hn
    h = h.transpose(1, 2)                                                    # tcn.py:153 ; mdtc.py:249
    caches = []
    off = 0
    if bb["type"] == "tcn":
        k, C = bb.get("kernel_size", 8), h.size(1)
        for i in range(bb["num_layers"]):                                    # tcn.py:155-163
            d = 2 ** i
            pad = (k - 1) * d
            u = _left_ctx(h, pad, in_cache, off)
            off += pad
            caches.append(u[:, :, -pad:])
            p = "backbone.network.%d.cnn." % i
            if bb.get("ds", False):                                          # tcn.py:101-114
                a = F.conv1d(u, sd[p + "0.weight"], sd[p + "0.bias"], dilation=d, groups=C)
                a = F.relu(_bn(a, sd, p + "1"))
                a = F.conv1d(a, sd[p + "3.weight"], sd[p + "3.bias"])
                a = F.relu(_bn(a, sd, p + "4"))
            else:                                                            # tcn.py:75-84
                a = F.relu(_bn(F.conv1d(u, sd[p + "0.weight"], sd[p + "0.bias"], dilation=d), sd, p + "1"))
            h = a + h                                                        # tcn.py:60
    elif bb["type"] == "mdtc":
        k, C = bb["kernel_size"], h.size(1)

        def block(h, p, d):                                                  # mdtc.py:95-121, 55-59
            nonlocal off
            pad = (k - 1) * d
            u = _left_ctx(h, pad, in_cache, off)
            off += pad
            caches.append(u[:, :, -pad:])
            a = F.conv1d(u, sd[p + "conv1.conv.weight"], sd[p + "conv1.conv.bias"], dilation=d, groups=C)
            a = _bn(a, sd, p + "conv1.bn")
            a = F.conv1d(a, sd[p + "conv1.pointwise.weight"], sd[p + "conv1.pointwise.bias"])
            a = F.relu(_bn(a, sd, p + "bn1"))
            a = _bn(F.conv1d(a, sd[p + "conv2.weight"], sd[p + "conv2.bias"]), sd, p + "bn2")
            return F.relu(a + h)
        h = F.relu(block(h, "backbone.preprocessor.", 1))                    # mdtc.py:251-253
        total = torch.zeros_like(h)
        for s in range(bb["num_stack"]):                                     # mdtc.py:255-273
            for j in range(bb["stack_size"]):
                h = block(h, "backbone.blocks.%d.res_blocks.%d." % (s, j), 2 ** j)
            total = total + h
        h = total
    else:
        raise NotImplementedError(bb["type"])
    h = h.transpose(1, 2)
    y = F.linear(h, sd["classifier.linear.weight"], sd["classifier.linear.bias"])               # classifier.py:63-67
    if "classifier" not in cfg and cfg.get("activation", {}).get("type") != "identity":
        y = torch.sigmoid(y)                                                 # kws_model.py:196-199
    return y, torch.cat(caches, dim=2)

# -*- coding: utf-8 -*-
"""oracle/detgen.py -- TEST INFRASTRUCTURE ONLY.

Bit-reproducible pseudo-random tensors from integer hashing (no RNG state, no libm):
the golden-vector generator (oracle/gen_golden.py, run once in the build container
against /root/reference) and the tests (run anywhere) rebuild identical weights and
inputs from (shape, seed), so fixtures only store the expected OUTPUTS.
"""
import math
import zlib

import torch


def det_uniform(shape, seed, scale=1.0):
    """U(-scale, scale)-like f32 tensor, a pure function of (shape, seed, scale)."""
    n = 1
    for s in shape:
        n *= int(s)
    i = torch.arange(n, dtype=torch.int64)
    x = (i * 2654435761 + (int(seed) & 0xFFFFFFFF) * 40503 + 12345) & 0xFFFFFFFF
    x = x ^ (x >> 15)
    x = (x * 73244475) & 0xFFFFFFFF
    x = x ^ (x >> 13)
    x = (x * 73244475) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    return ((x.double() / 4294967296.0 - 0.5) * (2.0 * float(scale))).float().view(*shape)


def det_ints(shape, seed, high):
    n = 1
    for s in shape:
        n *= int(s)
    i = torch.arange(n, dtype=torch.int64)
    x = (i * 2654435761 + (int(seed) & 0xFFFFFFFF) * 40503 + 977) & 0xFFFFFFFF
    x = x ^ (x >> 15)
    x = (x * 73244475) & 0xFFFFFFFF
    x = x ^ (x >> 13)
    return (x % int(high)).view(*shape)


def key_seed(key):
    return zlib.crc32(key.encode("utf-8"))


def det_fill_state(state, skip_prefixes=()):
    """Deterministic values for every floating entry of a state_dict-like mapping.

    SRU weights (n_in, d*k): amplitude sqrt(3/n_in); other >=2-D weights:
    sqrt(3/fan_in); embeddings: 1; 1-D (biases, BN affine): 0.1 (BN weight 1 +- 0.1,
    running_var 1 +- 0.1 so that nothing degenerates).
    """
    out = {}
    for key in sorted(state.keys()):
        t = state[key]
        if any(key.startswith(p) for p in skip_prefixes) or not t.dtype.is_floating_point:
            out[key] = t.clone()
            continue
        shp = tuple(t.shape)
        sd = key_seed(key)
        if t.dim() >= 2:
            if ".rnn_lst." in key:
                amp = math.sqrt(3.0 / shp[0])
            elif key.endswith("emb.weight"):
                amp = 1.0
            else:
                fan = 1
                for s in shp[1:]:
                    fan *= s
                amp = math.sqrt(3.0 / fan)
            out[key] = det_uniform(shp, sd, amp)
        else:
            v = det_uniform(shp, sd, 0.1)
            if key.endswith("running_var") or (key.endswith(".weight") and ".bn" in key):
                v = v + 1.0
            out[key] = v
    return out

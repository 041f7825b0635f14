"""Deterministic parameter recipe shared by the embedding-extractor fixture generator and its tests.

The extractor fixtures (`embed_*.npz`) hold inputs and expected outputs only; the parameters (5-20 M floats) are
re-created on both sides from this recipe, a pure function of the state_dict's key order and shapes, and pinned by the
checksum stored in the fixture.  Nothing here comes from the reference: it is test data generation.
"""
import hashlib

import numpy as np
import torch


def fill_state(module, seed=0):
    """Overwrite every floating entry of module.state_dict() in key order from one CPU generator."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        if not v.is_floating_point():
            new[k] = v.clone()
            continue
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "running_var":
            t = torch.rand(v.shape, generator=g) + 0.5
        elif leaf == "running_mean":
            t = torch.randn(v.shape, generator=g) * 0.1
        elif v.dim() == 1 and leaf == "weight":                      # norm scales
            t = torch.rand(v.shape, generator=g) + 0.5
        elif v.dim() == 1:                                             # biases
            t = torch.randn(v.shape, generator=g) * 0.1
        else:
            fan_in = int(np.prod(v.shape[1:]))
            t = torch.randn(v.shape, generator=g) / fan_in ** 0.5
        new[k] = t.to(v.dtype)
    module.load_state_dict(new, strict=True)
    return module


def state_checksum(module):
    h = hashlib.sha256()
    for k, v in module.state_dict().items():
        h.update(k.encode())
        h.update(str(tuple(v.shape)).encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()

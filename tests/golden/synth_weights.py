"""Deterministic synthetic weights derived from state_dict key names (no files to commit).

Both the reference modules (in make_golden.py) and the oracle / HIP modules (in tests) call
``fill_state_dict`` so that they hold bit-identical parameters without shipping ~80 MB of tensors."""
import zlib

import torch


def synth_tensor(key, shape, seed=0):
    g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + 7919 * seed) % (2 ** 31))
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_var":
        return 0.5 + torch.rand(shape, generator=g)
    if leaf == "running_mean":
        return 0.1 * torch.randn(shape, generator=g)
    if len(shape) == 1 and leaf == "weight":           # BatchNorm gamma
        return 0.5 + torch.rand(shape, generator=g)
    if leaf == "bias":
        return 0.1 * torch.randn(shape, generator=g)
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    return torch.randn(shape, generator=g) * (1.5 / max(fan_in, 1)) ** 0.5


def fill_state_dict(module, seed=0, prefix=""):
    sd = module.state_dict()
    new = {k: synth_tensor(prefix + k, tuple(v.shape), seed).to(v.dtype) for k, v in sd.items()}
    module.load_state_dict(new)
    return module

"""Generator-defined synthetic weights for benchmarking and profiling (there are no shipped checkpoints: README.md:58
of the reference).  ``synthetic_init_(module, seed)`` overwrites every parameter of a ``TemporalUnet1D`` / ``Unet``
in place with values that depend only on (seed, parameter name): conv / linear weights and biases uniform in
``+-1/sqrt(fan_in)`` (PyTorch's default bound), GroupNorm weights / LayerNorm gains ``1 + 0.1 u``, GroupNorm biases
``0.1 u``, u ~ U(-1, 1).  The same ``state_dict`` can then be handed to any other implementation of the model."""
import math
import zlib

import torch


def synthetic_init_(module, seed=0):
    params = dict(module.named_parameters())
    fan = {k[:-len(".weight")]: int(torch.tensor(p.shape[1:]).prod()) if p.dim() > 1 else 0
           for k, p in params.items() if k.endswith(".weight")}
    with torch.no_grad():
        for k, p in params.items():
            g = torch.Generator().manual_seed(zlib.crc32(f"{int(seed)}:{k}".encode()))
            u = torch.rand(p.shape, generator=g) * 2.0 - 1.0
            base = k.rsplit(".", 1)[0]
            is_norm_w = k.endswith(".g") or (p.dim() == 1 and k.endswith(".weight"))
            is_norm_b = p.dim() == 1 and k.endswith(".bias") and fan.get(base, 0) == 0
            if is_norm_w:
                v = 1.0 + 0.1 * u
            elif is_norm_b:
                v = 0.1 * u
            else:
                v = u / math.sqrt(max(fan.get(base, 0), 1))
            p.copy_(v.to(p.dtype).to(p.device))
    return module

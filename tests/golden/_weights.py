"""Deterministic, RNG-free parameter fill shared by make_golden.py (reference
side) and the tests (build side), so module fixtures need no stored weights."""
import math

import torch


def fill_state_dict_(module: torch.nn.Module, seed: float = 0.0, keys=None, prefix: str = "") -> None:
    """In-place: every float parameter/buffer gets a smooth pseudo-random
    pattern scaled by 1/sqrt(fan_in); BN running_var kept positive.

    The pattern of a tensor depends on its index in the sorted key list.  When the
    module is a sub-module of the model that was filled on the reference side, pass
    that model's sorted ``keys`` and the sub-module's ``prefix`` to get the same values."""
    order = None if keys is None else {str(k): i for i, k in enumerate(keys)}
    with torch.no_grad():
        for pos, (name, t) in enumerate(sorted(module.state_dict().items())):
            t_idx = pos if order is None else order[prefix + name]
            if not t.is_floating_point():
                continue
            n = t.numel()
            i = torch.arange(n, dtype=torch.float64)
            v = torch.sin(i * 0.37 + 1.3 * t_idx + seed) + 0.5 * torch.cos(i * 0.011 + 0.7 * t_idx)
            fan_in = 1
            if t.dim() > 1:
                fan_in = int(t[0].numel())
            v = v / math.sqrt(max(fan_in, 1))
            if name.endswith("running_var"):
                v = v.abs() + 0.5
            elif name.endswith("running_mean") or name.endswith("bias"):
                v = 0.1 * v
            elif t.dim() == 1:  # BN weight
                v = 1.0 + 0.1 * v
            t.copy_(v.reshape(t.shape).to(t.dtype))

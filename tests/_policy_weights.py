"""Deterministic weights for the policy-network parity test: a closed formula instead of a stored state_dict."""
import math

import torch


def fill_(module):
    """Same values for any module with the reference's parameter names (conv1, conv2, fc1, fc2, action_head, value_head)."""
    with torch.no_grad():
        for k, (name, p) in enumerate(sorted(module.named_parameters())):
            n = p.numel()
            idx = torch.arange(n, dtype=torch.float64)
            fan_in = p.shape[-1] if p.dim() == 2 else (p[0].numel() if p.dim() > 1 else 4)
            scale = 3.0 / math.sqrt(max(1, fan_in))  # large enough that sparse one-hot inputs move the outputs
            vals = torch.sin(idx * idx * 0.0137 + idx * 0.37 + k * 1.3) * scale
            p.copy_(vals.reshape(p.shape).to(torch.float32))
    return module

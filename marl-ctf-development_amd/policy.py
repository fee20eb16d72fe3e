"""Policy / value network for the batched rollout collector — the consumer of the observation buffer.

Same architecture and parameter names as the reference's ``Agent`` (agent_network.py:5-81: conv3x3(C->16) tanh,
conv3x3(16->32) tanh, flatten ++ metadata, fc 256, fc 128, action / value heads), so a reference ``state_dict`` loads
unchanged.  What differs is how it is fed and run at scale: observations arrive as the env's native uint8 planes and
float16 metadata for tens of thousands of agents at once, are cast on the fly, and the body can run in bfloat16
(``compute_dtype``); the action-mask rule ``logits + (mask - 1) * 1e9`` (agent_network.py:66-75) is applied in float32.
Stock PyTorch-ROCm (MIOpen / hipBLASLt underneath): this module is outside the env hot path.
"""
import torch
import torch.nn as nn
from torch.distributions.categorical import Categorical


class CtfPolicy(nn.Module):
    def __init__(self, n_actions, n_channels, grid_size, metadata_size, compute_dtype=torch.float32):
        super().__init__()
        self.n_actions, self.compute_dtype = n_actions, compute_dtype
        side = grid_size - 4  # two valid 3x3 convolutions
        self.flat = 32 * side * side
        self.conv1 = nn.Conv2d(n_channels, 16, kernel_size=3)
        self.conv2 = nn.Conv2d(16, 32, kernel_size=3)
        self.fc1 = nn.Linear(self.flat + metadata_size, 256)
        self.fc2 = nn.Linear(256, 128)
        self.action_head = nn.Linear(128, n_actions)
        self.value_head = nn.Linear(128, 1)
        self.register_buffer("mask_5", torch.tensor([1.0] * 5 + [0.0] * (n_actions - 5)))

    def trunk(self, grid, metadata):
        """grid: [B, C, G, G] any dtype (uint8 from the env), metadata: [B, M] any dtype -> (value [B, 1], logits [B, A]) float32."""
        dt = self.compute_dtype
        with torch.autocast("cuda", dtype=dt, enabled=grid.is_cuda and dt != torch.float32):
            x = torch.tanh(self.conv1(grid.to(dt if grid.is_cuda else torch.float32)))
            x = torch.tanh(self.conv2(x))
            x = torch.cat((x.reshape(-1, self.flat), metadata.to(x.dtype)), dim=1)
            x = torch.tanh(self.fc1(x))
            x = torch.tanh(self.fc2(x))
            value, logits = self.value_head(x), self.action_head(x)
        return value.float(), logits.float()

    forward = trunk

    def get_value(self, grid, metadata):
        return self.trunk(grid, metadata)[0]

    def get_action_and_value(self, grid, metadata, masking_decision_tensor, action=None):
        """The reference's signature and mask rule: decision 1 -> only actions 0..4 are legal, 0 -> all."""
        value, logits = self.trunk(grid, metadata)
        decision = masking_decision_tensor.reshape(-1, 1).to(logits.dtype)
        mask = torch.where(decision == 1, self.mask_5.unsqueeze(0), torch.ones_like(logits))
        logits = logits + (mask - 1.0) * 1e9
        dist = Categorical(logits=logits)
        if action is None:
            action = dist.sample()
        return action, dist.log_prob(action), dist.entropy(), value

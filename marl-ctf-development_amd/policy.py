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
import torch.nn.functional as F
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
        # row b = the planes of standardise_state that code byte b of the compact observation switches on (gridworld_ctf.expand_codes:
        # bit 7 -> plane 0, the agent's own position; low bits k > 0 -> plane k); not part of the state_dict
        byte = torch.arange(256)
        low = byte & 0x7F
        planes = torch.zeros(256, n_channels)
        sel = (low > 0) & (low < n_channels)
        planes[byte[sel], low[sel]] = 1.0
        planes[byte >= 128, 0] = 1.0
        self.register_buffer("code_planes", planes, persistent=False)

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

    def trunk_codes(self, codes, metadata):
        """The same function as ``trunk(expand_codes(codes), metadata)`` for the COMPACT observation (uint8 [B, G, G], one byte per
        cell), arranged for the learner's forward + backward on MI355X (tools/learner_profile.py):
        * the one-hot planes are one table lookup (256 rows of C values, row = code byte) that lands directly in channels-last
          memory in the compute dtype — no compare / select / cast chain, no float32 planes;
        * both convolutions run on channels-last tensors: MIOpen's bf16 kernels are NHWC implicit GEMMs, and fed NCHW tensors it
          wraps every one of them (forward, data gradient, weight gradient) in transposes that cost more than the convolutions;
        * conv2's output is flattened as the channels-last VIEW it already is, and fc1's weight columns are permuted to that
          (y, x, c) order instead (2 MB instead of the activations); the metadata columns are a second small GEMM instead of a
          concatenated copy of the activation matrix.
        Same parameters, same summands as ``trunk``; only the order of float additions inside the library kernels may differ."""
        b = codes.shape[0]
        dt = self.compute_dtype if codes.is_cuda else torch.float32
        cl = torch.channels_last
        with torch.autocast("cuda", dtype=dt, enabled=codes.is_cuda and dt != torch.float32):
            x = F.embedding(codes.int(), self.code_planes.to(dt)).permute(0, 3, 1, 2)  # [B, C, G, G], channels-last in memory
            x = torch.tanh(F.conv2d(x, self.conv1.weight.to(dt).contiguous(memory_format=cl), self.conv1.bias.to(dt)))
            x = torch.tanh(F.conv2d(x, self.conv2.weight.to(dt).contiguous(memory_format=cl), self.conv2.bias.to(dt)))
            positions = self.flat // 32
            flat = x.permute(0, 2, 3, 1).reshape(b, self.flat)  # (y, x, c) order: a view of channels-last memory
            w = self.fc1.weight
            w_flat = w[:, :self.flat].reshape(-1, 32, positions).permute(0, 2, 1).reshape(-1, self.flat)
            x = torch.tanh(F.linear(flat, w_flat, self.fc1.bias) + F.linear(metadata.to(flat.dtype), w[:, self.flat:]))
            x = torch.tanh(self.fc2(x))
            value, logits = self.value_head(x), self.action_head(x)
        return value.float(), logits.float()

    def forward(self, grid, metadata):
        """planes [B, C, G, G] (the reference's input) or compact codes uint8 [B, G, G]"""
        if grid.dim() == 3 and grid.dtype == torch.uint8:
            return self.trunk_codes(grid, metadata)
        return self.trunk(grid, metadata)

    def get_value(self, grid, metadata):
        return self.forward(grid, metadata)[0]

    def get_action_and_value(self, grid, metadata, masking_decision_tensor, action=None):
        """The reference's signature and mask rule: decision 1 -> only actions 0..4 are legal, 0 -> all."""
        value, logits = self.forward(grid, metadata)
        decision = masking_decision_tensor.reshape(-1, 1).to(logits.dtype)
        mask = torch.where(decision == 1, self.mask_5.unsqueeze(0), torch.ones_like(logits))
        logits = logits + (mask - 1.0) * 1e9
        dist = Categorical(logits=logits)
        if action is None:
            action = dist.sample()
        return action, dist.log_prob(action), dist.entropy(), value

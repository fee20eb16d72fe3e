"""A deterministic stand-in for the reference's policy network (agent_network.py:63-81) used to pin the rollout
layout: every output is an exact function of the observation, identical whether called per agent (batch of 1, as
ppo.py does) or for thousands of envs at once."""
import torch


class StubPolicy:
    def __init__(self, salt, n_actions=9):
        self.salt, self.n_actions = int(salt), int(n_actions)

    def _score(self, x):
        c, g = x.shape[-3], x.shape[-1]
        ci = torch.arange(c, device=x.device, dtype=torch.float32).view(c, 1, 1)
        yi = torch.arange(g, device=x.device, dtype=torch.float32).view(1, g, 1)
        xi = torch.arange(g, device=x.device, dtype=torch.float32).view(1, 1, g)
        w = torch.remainder(ci * 31 + yi * 7 + xi * 3 + self.salt, 11.0) + 1.0  # small integers: sums are exact in f32
        return (x.to(torch.float32) * w).sum(dim=(-3, -2, -1))

    def get_action_and_value(self, x, x2, masking_decision_tensor, action=None):
        x = x.reshape((-1,) + tuple(x.shape[-3:]))
        x2 = x2.reshape(x.shape[0], -1)
        s = self._score(x)
        mask = masking_decision_tensor.reshape(-1).to(torch.float32).expand(x.shape[0]) if masking_decision_tensor.numel() == 1 \
            else masking_decision_tensor.reshape(-1).to(torch.float32)
        n_allowed = torch.where(mask == 1, torch.full_like(s, 5.0), torch.full_like(s, float(self.n_actions)))
        act = torch.remainder(s, n_allowed).to(torch.int64)
        logprob = -torch.remainder(s, 7.0) / 8.0
        value = (torch.remainder(s, 13.0) / 16.0 + x2[:, 0].to(torch.float32) * 2.0).reshape(-1, 1)
        return act, logprob, torch.zeros_like(logprob), value


class StubDuelPolicy(StubPolicy):
    """The same stub with the single-observation ``get_action`` the reference's duel harness calls (agent_network.py:42-58)."""

    def get_action(self, x, x2, masking_decision_tensor):
        return int(self.get_action_and_value(x, x2, masking_decision_tensor)[0].reshape(-1)[0].item())


class StubCodesPolicy(StubDuelPolicy):
    """The same stub behind the compact-observation interface of policy_native.CtfPolicyNative: it expands the code bytes
    to the one-hot planes and scores those, so a compact rollout must reproduce the plane rollout bit for bit."""

    def __init__(self, salt, n_channels, expand_codes, n_actions=9):
        super().__init__(salt, n_actions)
        self.n_channels, self.expand_codes = int(n_channels), expand_codes

    def act_from_codes(self, codes, meta, agent_idx, masking_decision_tensor, action=None, shared_view=False, self_cells=None):
        idx = torch.tensor(list(agent_idx), device=codes.device)
        g = codes.shape[-1]
        planes = self.expand_codes(codes.index_select(1, idx).transpose(0, 1).reshape(-1, g, g), self.n_channels)
        md = meta.index_select(1, idx).transpose(0, 1).reshape(planes.shape[0], -1).to(torch.float32)
        return self.get_action_and_value(planes, md, masking_decision_tensor)

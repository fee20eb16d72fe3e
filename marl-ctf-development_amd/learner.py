"""The learner side of BASELINE configs[4] (``ppo.py`` self-play end to end): GAE and the PPO update of the reference
(ppo.py:133-242), stock PyTorch, consuming the batched COMPACT rollout of ``rollout.BatchedRolloutCollector`` — the
observations stay one byte per cell in the rollout buffer (7.5 GB instead of 106 GB for 128 steps x 65 536 envs) and are
expanded to the one-hot planes per minibatch, right before the network's stock ``forward``.

Nothing here is on the accelerated env path; it exists so that the end-to-end number (rollout + update) can be measured
and so that a user of the reference's ``PPOTrainer`` finds the same two functions with the same semantics:

* ``calculate_advantages``  ppo.py:133-172, vectorised over the env axis (the reference runs it on [num_steps, num_envs]
  tensors as well); bit-identical arithmetic per element (pinned by tests/golden/learner_ref.npz);
* ``PPOLearner.optimise``   ppo.py:174-242: minibatch order from ``np.random.shuffle`` (the reference's own source of
  order), ratio / clip-fraction / KL bookkeeping, advantage normalisation per minibatch, clipped value loss, entropy
  bonus, gradient-norm clipping, early stop on ``target_kl``.
"""
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

try:
    from .gridworld_ctf import expand_codes
except ImportError:  # pragma: no cover
    from gridworld_ctf import expand_codes

# the defaults of the reference's experiment scripts (e.g. 8_arena.py:76-100)
DEFAULT_ARGS = dict(learning_rate=2.5e-4, gamma=0.99, gae_lambda=0.95, gae=True, update_epochs=4, num_minibatches=4, norm_adv=True,
                    clip_coef=0.2, clip_vloss=True, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, target_kl=None)


def _reverse_scan(first, coeff, tail):
    """x[t] = first[t] + coeff[t] * x[t + 1] for t = S-1 .. 0 with x[S] = tail: the one sequential pass of either estimator."""
    out = torch.empty_like(first)
    carry = tail
    for t in range(first.shape[0] - 1, -1, -1):
        carry = first[t] + coeff[t] * carry
        out[t] = carry
    return out


def calculate_advantages(next_value, rewards, next_done, dones, values, gamma=0.99, gae_lambda=0.95, gae=True):
    """The estimators of ppo.py:145-170 as one backward recurrence over whole-rollout tensors.
    rewards / dones / values: [S, ...]; next_value / next_done: [...] -> (advantages, returns) [S, ...].

    What follows a step — its value and whether the episode goes on — is the same tensor shifted by one step with the bootstrap
    entry appended, so the TD residuals of all steps come out of one expression; the only sequential part is the discounted
    reverse sum.  Every element goes through the reference's operations in the reference's order (tests/golden/learner_ref.npz
    holds it to the last bits)."""
    with torch.no_grad():
        boot = next_value.reshape(rewards.shape[1:])
        alive_after = 1.0 - torch.cat([dones[1:], next_done.reshape((1,) + tuple(rewards.shape[1:])).to(dones.dtype)], dim=0)
        if gae:
            value_after = torch.cat([values[1:], boot.unsqueeze(0)], dim=0)
            residual = rewards + gamma * value_after * alive_after - values
            advantages = _reverse_scan(residual, gamma * gae_lambda * alive_after, torch.zeros_like(boot))
            returns = advantages + values
        else:
            returns = _reverse_scan(rewards, gamma * alive_after, boot)
            advantages = returns - values
    return advantages, returns


class PPOLearner:
    def __init__(self, agent, n_channels, **args):
        """agent: a module with the reference's ``get_action_and_value`` / ``get_value`` (policy.CtfPolicy,
        policy_native.CtfPolicyNative or the reference's own Agent).  n_channels: C of the observation (to expand codes)."""
        self.agent, self.n_channels = agent, int(n_channels)
        self.codes_direct = hasattr(agent, "trunk_codes")  # False: always expand (the reference's own Agent, or to compare the two paths)
        self.args = SimpleNamespace(**dict(DEFAULT_ARGS, **args))
        self.optimizer = torch.optim.Adam(agent.parameters(), lr=self.args.learning_rate, eps=1e-5)  # ppo.py:283

    def _planes(self, grids):
        """What the network is fed: uint8 codes [B, G, G] as they are when the agent evaluates them directly (policy.CtfPolicy.trunk_codes:
        the convolutions as GEMMs over 3x3 patches, no one-hot planes), else expanded to float32 planes [B, C, G, G]; planes pass through."""
        if grids.dim() == 3:
            if self.codes_direct:
                return grids
            return expand_codes(grids, self.n_channels).to(torch.float32)
        return grids.to(torch.float32)

    def advantages(self, rollout):
        """GAE on a collected rollout (the dict ``BatchedRolloutCollector.collect`` returns)."""
        a = self.args
        nxt = rollout["next_grid_codes"] if "next_grid_codes" in rollout else rollout["next_grid_state"]
        with torch.no_grad():
            next_value = self.agent.get_value(self._planes(nxt), rollout["next_metadata_state"].to(torch.float32))
        return calculate_advantages(next_value, rollout["rewards"], rollout["next_done"], rollout["dones"], rollout["values"],
                                    a.gamma, a.gae_lambda, a.gae)

    def optimise(self, b_grids, b_metadata_states, b_logprobs, b_actions, b_use_action_mask, b_advantages, b_returns, b_values,
                 micro_batch=None, progress=None):
        """ppo.py:174-242 on a flattened batch.  b_grids: uint8 codes [B, G, G] (expanded per minibatch) or planes [B, C, G, G].

        micro_batch: evaluate a minibatch in pieces of at most this many samples, accumulating gradients — the same update
        (the minibatch's advantage statistics and means are taken over the WHOLE minibatch, one optimiser step per minibatch),
        for batches of 10^5..10^6 samples per minibatch where a single convolution call is impractically slow (MIOpen on
        N = 262 144: 6 k samples/s; in pieces of 16 384: 4.5 M samples/s).  progress: callable(str), called per minibatch."""
        a, agent = self.args, self.agent
        batch_size = b_logprobs.shape[0]
        minibatch_size = int(batch_size // a.num_minibatches)
        b_inds = np.arange(batch_size)
        v_loss = pg_loss = entropy_loss = approx_kl = None
        stats = None
        for epoch in range(a.update_epochs):
            np.random.shuffle(b_inds)
            inds_dev = torch.from_numpy(b_inds).to(b_logprobs.device)  # the epoch's order, uploaded once
            for start in range(0, batch_size, minibatch_size):
                mb_all = inds_dev[start:start + minibatch_size]
                n_mb = mb_all.numel()
                mb_adv_all = b_advantages[mb_all]
                if a.norm_adv:
                    mb_adv_all = (mb_adv_all - mb_adv_all.mean()) / (mb_adv_all.std() + 1e-8)
                self.optimizer.zero_grad()
                piece = n_mb if not micro_batch else int(micro_batch)
                sums = torch.zeros(5, dtype=torch.float64, device=b_logprobs.device)  # pg, v, entropy, kl, clip: summed on the device
                for lo in range(0, n_mb, piece):
                    mb, mb_advantages = mb_all[lo:lo + piece], mb_adv_all[lo:lo + piece]
                    _, newlogprob, entropy, newvalue = agent.get_action_and_value(self._planes(b_grids[mb]), b_metadata_states[mb].to(torch.float32),
                                                                                  b_use_action_mask[mb], b_actions[mb].long())
                    logratio = newlogprob - b_logprobs[mb]
                    ratio = logratio.exp()
                    with torch.no_grad():
                        kl_sum = ((ratio - 1) - logratio).sum()
                        clip_sum = ((ratio - 1.0).abs() > a.clip_coef).float().sum()
                    pg_loss1 = -mb_advantages * ratio
                    pg_loss2 = -mb_advantages * torch.clamp(ratio, 1 - a.clip_coef, 1 + a.clip_coef)
                    pg_sum = torch.max(pg_loss1, pg_loss2).sum()
                    newvalue = newvalue.view(-1)
                    if a.clip_vloss:
                        v_loss_unclipped = (newvalue - b_returns[mb]) ** 2
                        v_clipped = b_values[mb] + torch.clamp(newvalue - b_values[mb], -a.clip_coef, a.clip_coef)
                        v_loss_clipped = (v_clipped - b_returns[mb]) ** 2
                        v_sum = 0.5 * torch.max(v_loss_unclipped, v_loss_clipped).sum()
                    else:
                        v_sum = 0.5 * ((newvalue - b_returns[mb]) ** 2).sum()
                    ent_sum = entropy.sum()
                    # loss = pg_loss - ent_coef * entropy_loss + v_loss * vf_coef with every term a mean over the minibatch
                    ((pg_sum - a.ent_coef * ent_sum + v_sum * a.vf_coef) / n_mb).backward()
                    sums += torch.stack([pg_sum.detach(), v_sum.detach(), ent_sum.detach(), kl_sum, clip_sum]).double()
                stats = sums / n_mb  # stays on the device: the host reads the numbers once per epoch (or per minibatch for `progress`)
                nn.utils.clip_grad_norm_(agent.parameters(), a.max_grad_norm)
                self.optimizer.step()
                if progress is not None:
                    pg_loss, v_loss, entropy_loss, approx_kl, _ = stats.tolist()
                    progress(f"epoch {epoch} minibatch at {start}: v {v_loss:.4g} pg {pg_loss:.4g} entropy {entropy_loss:.4g}")
            if stats is not None:
                pg_loss, v_loss, entropy_loss, approx_kl, _ = stats.tolist()  # of the epoch's last minibatch, as the reference keeps them
            if a.target_kl is not None and approx_kl > a.target_kl:
                break
        return v_loss, pg_loss, entropy_loss

    def update(self, rollout, micro_batch=None, progress=None):
        """GAE + PPO update on one collected rollout -> (v_loss, pg_loss, entropy_loss).  The batch is the rollout flattened
        slot-major, as the reference flattens its [num_steps, num_envs] tensors (ppo.py:372-380)."""
        adv, ret = self.advantages(rollout)
        grids = rollout["grid_codes"] if "grid_codes" in rollout else rollout["grid_states"]
        flat = lambda t: t.reshape((-1,) + tuple(t.shape[2:]))
        return self.optimise(flat(grids), flat(rollout["metadata_states"]), flat(rollout["logprobs"]), flat(rollout["actions"]),
                             flat(rollout["use_action_mask"]), flat(adv), flat(ret), flat(rollout["values"]), micro_batch, progress)

"""The learner side of BASELINE configs[4] (``ppo.py`` self-play end to end): GAE and the PPO update of the reference
(ppo.py:133-242) consuming the batched COMPACT rollout of ``rollout.BatchedRolloutCollector`` — the observations stay one byte per
cell in the rollout buffer (7.5 GB instead of 106 GB for 128 steps x 65 536 envs) and go into the network as they are: a
``policy_native.CtfPolicyNative`` evaluates a minibatch of code bytes with native MFMA kernels in both directions (forward
``ctf_policy_features_train``, data gradient ``ctf_policy_front_dgrad``, weight gradients ``ctf_policy_front_wgrad``; the dense
layers and the optimiser are PyTorch's), a ``policy.CtfPolicy`` through a table lookup + channels-last library convolutions; only an
agent without ``trunk_codes`` (the reference's own ``Agent``) gets the codes expanded to one-hot planes per piece.

* ``calculate_advantages``  ppo.py:133-172, vectorised over the env axis (the reference runs it on [num_steps, num_envs]
  tensors as well); bit-identical arithmetic per element (pinned by tests/golden/learner_ref.npz);
* ``PPOLearner.optimise``   ppo.py:174-242: minibatch order from ``np.random.shuffle`` (the reference's own source of
  order), ratio / clip-fraction / KL bookkeeping, advantage normalisation per minibatch, clipped value loss, entropy
  bonus, gradient-norm clipping, early stop on ``target_kl``;
* **N ranks** (``PPOLearner(..., world=N, rank=r)``, one process per GPU): the reference concatenates the rollouts of its N workers
  into ONE update (ppo.py:359-376 -> :174-242).  Here every rank keeps its env shard's rollout where it was collected and the update
  is data-parallel over the SAME global minibatches: one global permutation per epoch (identical on every rank), a rank evaluates the
  samples of each minibatch that lie in its own env columns, the minibatch's advantage mean / std come from an all-reduce of
  (sum, sum of squares, count), every mean of the loss is over the GLOBAL minibatch, and ONE flat float32 all-reduce per optimiser
  step sums the gradients before ``clip_grad_norm_`` and Adam — so the N ranks' update is the single-process update on the union of
  their shards (tests/test_learner_dp_gloo.py: equal to 1e-6 in float32, 1e-10 in float64).
"""
import contextlib
import os
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


def _via_host(t, group):
    """gloo moves host memory: a device tensor goes through a host copy (the one-GPU rehearsal of the N-rank path runs over gloo, RCCL
    refuses two ranks on one device); RCCL / gloo-on-CPU take the tensor as it is."""
    import torch.distributed as dist

    return t.is_cuda and dist.get_backend(group) == "gloo"


def broadcast_module(module, src=0, group=None):
    """Every rank takes rank ``src``'s parameters of ``module`` (one flat broadcast)."""
    import torch.distributed as dist

    ps = list(module.parameters())
    flat = torch.cat([p.detach().reshape(-1) for p in ps])
    if _via_host(flat, group):
        host = flat.cpu()
        dist.broadcast(host, src=src, group=group)
        flat = host.to(flat.device)
    else:
        dist.broadcast(flat, src=src, group=group)
    with torch.no_grad():
        off = 0
        for p in ps:
            p.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
    return module


class PPOLearner:
    DET_WORKSPACE_FLOATS = 24 << 20  # 96 MB: every launch of the 8_arena network on a 256-CU device (include/ctf_policy.h)

    def __init__(self, agent, n_channels, world=1, rank=0, group=None, order=None, sync_params=True, force_collective=False,
                 deterministic=None, **args):
        """agent: a module with the reference's ``get_action_and_value`` / ``get_value`` (policy.CtfPolicy,
        policy_native.CtfPolicyNative or the reference's own Agent).  n_channels: C of the observation (to expand codes).

        world / rank / group: the data-parallel job this learner is one rank of (``torch.distributed`` initialised by the caller:
        RCCL on GPUs, gloo in the CPU tests).  world = 1: the single-process update, no collective (``force_collective``: the N-rank
        code path all the same, in a one-rank group — how one GPU rehearses it over RCCL).
        order: where an epoch's minibatch order comes from — "numpy": ``np.random.shuffle`` of the batch indices on the host, the
        reference's own (ppo.py:192; what tests/golden/learner_ref.npz pins); "device": ``torch.randperm`` on the learner's device,
        seeded per epoch with one integer drawn from ``np.random`` (rank 0's, broadcast) — identical on every rank without moving
        a permutation of the global batch between hosts, and no 0.1-0.2 s host shuffle per 4 M samples in front of every epoch.
        Default: "numpy" alone, "device" in a job of several ranks.
        deterministic (default: the environment's CTF_DETERMINISTIC=1): the native network's weight / bias gradients are reduced in a
        FIXED order — every block of the gradient kernels stores its partial sums, a second launch adds them in block order
        (ctf_policy_set_deterministic) — instead of through float atomics: two identical updates are then bit-identical, as two runs
        of the reference's update under the same seeds are (ppo.py:174-242).  Costs a 96 MB workspace and a few per cent of the
        update (profiles/r05_deterministic_learner.md)."""
        self.agent, self.n_channels = agent, int(n_channels)
        self.deterministic = (os.environ.get("CTF_DETERMINISTIC", "0") not in ("", "0")) if deterministic is None else bool(deterministic)
        self._det_ws = None
        self.codes_direct = hasattr(agent, "trunk_codes")  # False: always expand (the reference's own Agent, or to compare the two paths)
        self.args = SimpleNamespace(**dict(DEFAULT_ARGS, **args))
        self.world, self.rank, self.group = int(world), int(rank), group
        self.dp = self.world > 1 or bool(force_collective)  # force_collective: the N-rank code path in a one-rank group (rehearsal on one GPU)
        self.order = order or ("numpy" if not self.dp else "device")
        if self.order not in ("numpy", "device"):
            raise ValueError("order: 'numpy' or 'device'")
        if self.dp and self.order == "numpy":
            raise ValueError("a job of several ranks draws its minibatch order on the device (order='device'): every rank needs the same one")
        self._flat = None
        if self.dp and sync_params:
            self.broadcast_parameters()
        self.optimizer = torch.optim.Adam(agent.parameters(), lr=self.args.learning_rate, eps=1e-5)  # ppo.py:283

    # -- the pieces of the N-rank update ---------------------------------------------------------------------------------------
    def broadcast_parameters(self, src=0):
        """Every rank starts from rank ``src``'s parameters (one flat broadcast)."""
        broadcast_module(self.agent, src, self.group)

    def _flat_grads(self):
        """All gradients as views of ONE flat buffer (allocated once): zeroing is one fill, the all-reduce of an optimiser step is one
        collective on the buffer in place, and autograd accumulates the pieces of a minibatch straight into it."""
        ps = [p for p in self.agent.parameters() if p.requires_grad]
        stamp = tuple((p.data_ptr(), p.dtype) for p in ps)
        if self._flat is None or self._flat[1] != stamp:
            dt, dev = ps[0].dtype, ps[0].device
            if any(p.dtype != dt or p.device != dev for p in ps):
                if not self.dp:  # alone: nothing is all-reduced, the parameters keep gradients of their own
                    return None
                raise ValueError("the flat gradient buffer of a job of several ranks needs parameters of one dtype on one device")
            self._flat = (torch.zeros(sum(p.numel() for p in ps), dtype=dt, device=dev), stamp)
        flat, off = self._flat[0], 0
        for p in ps:
            view = flat[off:off + p.numel()].view_as(p)
            if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                p.grad = view
            off += p.numel()
        return flat

    def _epoch_order(self, batch_size, device):
        """-> int64 device tensor: this epoch's permutation of the (global) batch indices."""
        if self.order == "numpy":
            np.random.shuffle(self._b_inds)  # the same array shuffled again every epoch, as the reference does (ppo.py:189-192)
            return torch.from_numpy(self._b_inds).to(device)
        seed = torch.tensor([int(np.random.randint(0, 2 ** 31 - 1)) if self.rank == 0 else 0], dtype=torch.int64, device=device)
        if self.dp:
            import torch.distributed as dist

            if _via_host(seed, self.group):
                host = seed.cpu()
                dist.broadcast(host, src=0, group=self.group)
                seed = host
            else:
                dist.broadcast(seed, src=0, group=self.group)
        gen = torch.Generator(device=device)
        gen.manual_seed(int(seed.item()))
        return torch.randperm(batch_size, generator=gen, device=device)

    def _all_reduce(self, t):
        if self.dp:
            import torch.distributed as dist

            if _via_host(t, self.group):
                host = t.cpu()
                dist.all_reduce(host, group=self.group)
                t.copy_(host)
            else:
                dist.all_reduce(t, group=self.group)
        return t

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
                 micro_batch=None, progress=None, n_envs=None):
        """ppo.py:174-242 on a flattened batch.  b_grids: uint8 codes [B, G, G] (expanded per minibatch) or planes [B, C, G, G].

        micro_batch: evaluate a minibatch in pieces of at most this many samples, accumulating gradients — the same update
        (the minibatch's advantage statistics and means are taken over the WHOLE minibatch, one optimiser step per minibatch),
        for batches of 10^5..10^6 samples per minibatch where a single convolution call is impractically slow (MIOpen on
        N = 262 144: 6 k samples/s; in pieces of 16 384: 4.5 M samples/s).  progress: callable(str), called per minibatch.
        n_envs: in a job of several ranks, the number of env columns of THIS rank's flattened [S, n_envs] rollout (``update`` passes
        it): local sample s * n_envs + e is global sample s * E_total + lo + e, lo = the env columns of the ranks before this one."""
        a, agent = self.args, self.agent
        dev = b_logprobs.device
        local_size = b_logprobs.shape[0]
        dp = self.dp
        if dp:
            if not n_envs or local_size % int(n_envs):
                raise ValueError("a job of several ranks needs n_envs, the env axis of this rank's [S, n_envs] rollout")
            counts = torch.zeros((self.world, 2), dtype=torch.int64, device=dev)
            counts[self.rank, 0], counts[self.rank, 1] = int(n_envs), local_size // int(n_envs)
            self._all_reduce(counts)
            counts, slots = counts[:, 0].tolist(), counts[:, 1].tolist()
            if len(set(slots)) != 1:  # (a rank with another rollout length would index out of range or hang a later collective)
                raise ValueError(f"every rank of the job must hold a rollout of the same number of slots; the ranks report {slots}")
            e_loc, e_tot, lo = int(n_envs), sum(counts), sum(counts[:self.rank])
            batch_size = (local_size // e_loc) * e_tot
        else:
            batch_size = local_size
        minibatch_size = int(batch_size // a.num_minibatches)
        starts = list(range(0, batch_size, minibatch_size))
        self._b_inds = np.arange(batch_size)
        flat_grads = self._flat_grads()
        v_loss = pg_loss = entropy_loss = approx_kl = None
        stats = None
        with self._deterministic_scope(dev):
            for epoch in range(a.update_epochs):
                inds_dev = self._epoch_order(batch_size, dev)  # the epoch's order (global), on the device once
                if dp:
                    # this rank's share of every global minibatch, for the whole epoch at once (one host round trip per epoch): the samples
                    # whose env column is one of its own, in the order the permutation lists them
                    col = inds_dev % e_tot
                    mine = (col >= lo) & (col < lo + e_loc)
                    local_all = ((inds_dev // e_tot) * e_loc + (col - lo))[mine]
                    bounds = torch.tensor(starts + [batch_size], device=dev)
                    cum = torch.cat([mine.new_zeros(1, dtype=torch.int64), mine.cumsum(0)])
                    offs = cum[bounds].tolist()
                    # the advantage statistics of every global minibatch: (sum, sum of squares, count) of the local shares, ONE all-reduce
                    adv_stats = torch.zeros((len(starts), 3), dtype=torch.float64, device=dev)
                    for k in range(len(starts)):
                        x = b_advantages[local_all[offs[k]:offs[k + 1]]].double()
                        adv_stats[k, 0], adv_stats[k, 1], adv_stats[k, 2] = x.sum(), (x * x).sum(), x.numel()
                    self._all_reduce(adv_stats)
                for k, start in enumerate(starts):
                    if dp:
                        mb_all = local_all[offs[k]:offs[k + 1]]
                        n_mb = min(start + minibatch_size, batch_size) - start  # of the GLOBAL minibatch: every mean below is over it
                        mb_adv_all = b_advantages[mb_all]
                        if a.norm_adv:  # ppo.py:206-208 over the global minibatch: mean, and torch.std's unbiased estimator
                            tot, sq, cnt = adv_stats[k, 0], adv_stats[k, 1], adv_stats[k, 2]
                            mean = tot / cnt
                            std = ((sq - cnt * mean * mean).clamp(min=0) / (cnt - 1)).sqrt()
                            mb_adv_all = ((mb_adv_all.double() - mean) / (std + 1e-8)).to(mb_adv_all.dtype)
                    else:
                        mb_all = inds_dev[start:start + minibatch_size]
                        n_mb = mb_all.numel()
                        mb_adv_all = b_advantages[mb_all]
                        if a.norm_adv and self.deterministic:
                            # the N-rank formula (float64 sums) alone too: a one-rank job and the single process are then the SAME
                            # computation, bit for bit (within 1 ulp of float32 of torch's mean / std below)
                            x = mb_adv_all.double()
                            cnt = x.numel()
                            mean = x.sum() / cnt
                            std = (((x * x).sum() - cnt * mean * mean).clamp(min=0) / (cnt - 1)).sqrt()
                            mb_adv_all = ((x - mean) / (std + 1e-8)).to(mb_adv_all.dtype)
                        elif a.norm_adv:
                            mb_adv_all = (mb_adv_all - mb_adv_all.mean()) / (mb_adv_all.std() + 1e-8)
                    if flat_grads is not None:
                        flat_grads.zero_()
                    else:
                        self.optimizer.zero_grad(set_to_none=True)
                    n_here = mb_all.numel()
                    piece = max(n_here, 1) if not micro_batch else int(micro_batch)
                    sums = torch.zeros(5, dtype=torch.float64, device=dev)  # pg, v, entropy, kl, clip: summed on the device
                    for lo_p in range(0, n_here, piece):
                        mb, mb_advantages = mb_all[lo_p:lo_p + piece], mb_adv_all[lo_p:lo_p + piece]
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
                    if dp:  # ONE flat all-reduce of the gradients per optimiser step (+ the five loss sums, so that every rank reports the
                        self._all_reduce(flat_grads)  # global minibatch's numbers and takes the same early-stop decision)
                        self._all_reduce(sums)
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

    @contextlib.contextmanager
    def _deterministic_scope(self, dev):
        """Registers the fixed-order reduction workspace for the update's duration (deterministic=True on a native network on a GPU)."""
        lib = None
        if self.deterministic and dev.type == "cuda" and hasattr(self.agent, "native_training"):
            if self._det_ws is None or self._det_ws.device != dev:
                self._det_ws = torch.empty(self.DET_WORKSPACE_FLOATS, dtype=torch.float32, device=dev)
            try:
                from . import _abi
            except ImportError:  # pragma: no cover
                import _abi
            lib = _abi.load_library()
            if lib.ctf_policy_set_deterministic(dev.index, self._det_ws.data_ptr(), self._det_ws.numel()) != 0:
                raise _abi.CtfLibraryError("ctf_policy_set_deterministic: " + (lib.ctf_policy_last_error() or b"").decode())
        try:
            yield
        finally:
            if lib is not None:
                torch.cuda.current_stream(dev).synchronize()  # (the update's last launches still write into the workspace)
                lib.ctf_policy_set_deterministic(dev.index, None, 0)

    def update(self, rollout, micro_batch=None, progress=None):
        """GAE + PPO update on one collected rollout -> (v_loss, pg_loss, entropy_loss).  The batch is the rollout flattened
        slot-major, as the reference flattens its [num_steps, num_envs] tensors (ppo.py:372-380).  In a job of several ranks every rank
        passes ITS shard's rollout ([S, E_rank, ...]: GAE needs only the rank's own columns) and the update is the one over the union."""
        adv, ret = self.advantages(rollout)
        grids = rollout["grid_codes"] if "grid_codes" in rollout else rollout["grid_states"]
        flat = lambda t: t.reshape((-1,) + tuple(t.shape[2:]))
        return self.optimise(flat(grids), flat(rollout["metadata_states"]), flat(rollout["logprobs"]), flat(rollout["actions"]),
                             flat(rollout["use_action_mask"]), flat(adv), flat(ret), flat(rollout["values"]), micro_batch, progress,
                             n_envs=int(rollout["logprobs"].shape[1]))

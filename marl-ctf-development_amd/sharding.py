"""Multi-GPU layout of the path: envs are independent, so each rank (one process per GPU) owns a
contiguous range of global env indices with its own handle and stream, and stepping / rendering needs
no communication.  The only exchange is the rollout hand-off to the learner: an all-gather of the
compact per-step tensors (rewards, done) over RCCL (backend "nccl" on ROCm), issued asynchronously once
per chunk of steps (ChunkedRolloutGather; RolloutGather is the per-step variant).  The same code runs
over gloo on CPU tensors in the tests.
"""
import os

import numpy as np


def world_from_env():
    """(rank, local_rank, world_size) as torchrun exports them (1-process defaults otherwise)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_range(n_global, rank, world):
    """Contiguous split of n_global envs over `world` ranks; the first n_global % world ranks get one more."""
    base, extra = divmod(int(n_global), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def env_seeds(run, lo, hi):
    """per-env seeds s_py[e] = s_np[e] = 1_000_003 * run + e over GLOBAL env indices
    (reduced modulo 2**32, np.random.seed's limit)."""
    e = np.arange(lo, hi, dtype=np.uint64)
    return (np.uint64(1_000_003) * np.uint64(run) + e) % np.uint64(2 ** 32)


class RolloutGather:
    """All-gather of the per-step compact rollout tensors (rewards [E, N] f32, done [E] u8) of every
    shard, double-buffered so that the collective of step t may still be in flight while step t+1 runs."""

    def __init__(self, rewards, done, world, group=None, force_collective=False):
        import torch

        self.world, self.group = world, group
        self.collective = world > 1 or force_collective
        self.bufs, self.src = [], []
        for _ in range(2):
            self.bufs.append((
                torch.empty((world * rewards.shape[0],) + tuple(rewards.shape[1:]), dtype=rewards.dtype, device=rewards.device),
                torch.empty((world * done.shape[0],), dtype=done.dtype, device=done.device),
            ))
            # the env overwrites its rewards / done buffers on the next step while a gather may still be reading:
            # every in-flight gather reads from its own snapshot
            self.src.append((torch.empty_like(rewards), torch.empty_like(done)))
        self.pending = [None, None]
        self.k = 0

    def start(self, rewards, done):
        """Enqueue the gather of this step's tensors (async); returns the slot it will land in."""
        import torch.distributed as dist

        slot = self.k & 1
        self.wait(slot)
        gr, gd = self.bufs[slot]
        if not self.collective:
            self.bufs[slot] = (rewards, done)  # one rank: the shard is the whole batch, nothing to exchange or copy
        else:
            sr, sd = self.src[slot]
            sr.copy_(rewards, non_blocking=True)
            sd.copy_(done, non_blocking=True)
            w1 = dist.all_gather_into_tensor(gr, sr, group=self.group, async_op=True)
            w2 = dist.all_gather_into_tensor(gd, sd, group=self.group, async_op=True)
            self.pending[slot] = (w1, w2)
        self.k += 1
        return slot

    def wait(self, slot=None):
        for s in ((0, 1) if slot is None else (slot,)):
            if self.pending[s] is not None:
                for w in self.pending[s]:
                    w.wait()
                self.pending[s] = None

    def result(self, slot):
        self.wait(slot)
        return self.bufs[slot]


class ChunkedRolloutGather:
    """The same hand-off once per CHUNK of steps: the env writes step t's rewards / done straight into slot t % C of a
    chunk buffer (``views``; no copy), and every C steps the whole chunk ([C, E, N] f32 + [C, E] u8) is all-gathered
    asynchronously while the next chunk fills the other buffer.  A learner needs the rollout only when it is complete, so
    nothing waits for the per-step latency of a collective, and its launch cost is paid once per C steps."""

    def __init__(self, n_envs, n_agents, device, world, chunk=16, group=None, force_collective=False, dtype=None):
        import torch

        self.world, self.group, self.C = int(world), group, int(chunk)
        self.collective = world > 1 or force_collective
        f32 = dtype or torch.float32
        self.local = [(torch.zeros((self.C, n_envs, n_agents), dtype=f32, device=device),
                       torch.zeros((self.C, n_envs), dtype=torch.uint8, device=device)) for _ in range(2)]
        self.glob = [(torch.zeros((self.world, self.C, n_envs, n_agents), dtype=f32, device=device),
                      torch.zeros((self.world, self.C, n_envs), dtype=torch.uint8, device=device)) if self.collective else None
                     for _ in range(2)]
        self.pending = [None, None]

    def views(self, t):
        """-> (rewards [E, N], done [E]) that step t must write (e.g. ``vec.rewards, vec.done = gather.views(t)``)."""
        b, i = (t // self.C) & 1, t % self.C
        if i == 0:
            self.wait(b)  # the gather that last read this buffer (two chunks ago) must be done before it is overwritten
        r, d = self.local[b]
        return r[i], d[i]

    def step_done(self, t):
        """Call after step t was enqueued; launches the chunk's gather when t closes a chunk."""
        if t % self.C == self.C - 1:
            self._launch((t // self.C) & 1)

    def flush(self, t_next):
        """Gather a partly filled last chunk (steps up to t_next - 1)."""
        if t_next % self.C:
            self._launch((t_next // self.C) & 1)

    def _launch(self, b):
        if not self.collective:
            return
        import torch.distributed as dist

        (lr, ld), (gr, gd) = self.local[b], self.glob[b]
        # (the output as the concatenation along dim 0 that every backend accepts; [world, C, ...] is a view of it)
        w1 = dist.all_gather_into_tensor(gr.view((self.world * self.C,) + tuple(lr.shape[1:])), lr, group=self.group, async_op=True)
        w2 = dist.all_gather_into_tensor(gd.view((self.world * self.C,) + tuple(ld.shape[1:])), ld, group=self.group, async_op=True)
        self.pending[b] = (w1, w2)

    def wait(self, b=None):
        for s in ((0, 1) if b is None else (b,)):
            if self.pending[s] is not None:
                for w in self.pending[s]:
                    w.wait()
                self.pending[s] = None

    def result(self, chunk_index):
        """-> (rewards [world, C, E, N], done [world, C, E]) of a gathered chunk (rank-major = global env order)."""
        b = chunk_index & 1
        self.wait(b)
        if not self.collective:
            r, d = self.local[b]
            return r[None], d[None]
        return self.glob[b]


class RolloutHandoff:
    """All-gather of the COMPACT rollout a centralised PPO learner consumes (what ppo.py:46-55,74-84,105-109
    stores per slot): reward f32, log-prob f32, value f32, action i32, use_action_mask i32 (any integer a caller's policy uses as an
    action or as a masking decision survives the trip: -1 stays -1) — and, optionally, the observation
    the policy saw in compact form (grid codes u8 [G][G], metadata f16 [M]) — for the trained team's slots of every shard.

    The collector hands over one CHUNK of slots at a time (``launch``), as soon as the chunk's last reward is stored; the
    collectives (two per chunk: one float32 pack, one uint8 pack; two more with observations) run asynchronously beside the
    following env steps (the "u8" pack of round 2 is an int32 pack now: gloo and RCCL both carry it).  ``result`` waits and returns tensors in GLOBAL env order: [S, world * E, ...] — rank-major, which
    is the global env index because rank r owns envs [r * E, (r + 1) * E).  One rank (and no ``force_collective``): the
    local tensors are returned as they are, nothing is copied."""

    F32 = ("rewards", "logprobs", "values")
    U8 = ("actions", "use_action_mask")

    def __init__(self, world, group=None, with_observations=False, force_collective=False):
        self.world, self.group = int(world), group
        self.collective = world > 1 or force_collective
        self.with_observations = bool(with_observations)
        self.pending = []   # (slot range, {pack name: (work, gathered tensor)}, shapes)
        self.local = None

    def launch(self, lo, hi, local):
        """local: dict name -> tensor [S, E, ...] of this rank (the collector's rollout buffers); slots [lo, hi) are final."""
        import torch

        self.local = local
        if not self.collective or hi <= lo:
            return
        import torch.distributed as dist

        packs = {
            "f32": torch.stack([local[k][lo:hi].to(torch.float32) for k in self.F32], dim=1).contiguous(),     # [K, 3, E]
            "i32": torch.stack([local[k][lo:hi].to(torch.int32) for k in self.U8], dim=1).contiguous(),        # [K, 2, E]
        }
        if self.with_observations:
            packs["codes"] = local["grid_codes"][lo:hi].contiguous()                                            # [K, E, G, G] u8
            packs["meta"] = local["metadata_states"][lo:hi].to(torch.float16).contiguous()                      # [K, E, M] (f16 is exact: the env emits f16)
        works = {}
        for name, t in packs.items():
            out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            w = dist.all_gather_into_tensor(out.view((self.world * t.shape[0],) + tuple(t.shape[1:])), t, group=self.group, async_op=True)
            works[name] = (w, out, t)  # (t is kept alive until the collective has read it)
        self.pending.append(((lo, hi), works))

    def result(self):
        """-> dict name -> tensor [S, world * E, ...] over all launched chunks (slots not launched are not included: call
        after the last ``launch``)."""
        import torch

        if not self.collective:
            keys = self.F32 + self.U8 + (("grid_codes", "metadata_states") if self.with_observations else ())
            return {k: self.local[k] for k in keys if k in self.local}
        chunks = {}
        for (lo, hi), works in self.pending:
            for name, (w, out, _) in works.items():
                w.wait()
                chunks.setdefault(name, []).append(out)
        self.pending = []
        if not chunks:  # collectives on, nothing launched: empty tensors of the right trailing shape, not a KeyError
            keys = self.F32 + self.U8 + (("grid_codes", "metadata_states") if self.with_observations else ())
            loc = self.local or {}
            return {k: loc[k].new_zeros((0, self.world * loc[k].shape[1]) + tuple(loc[k].shape[2:])) for k in keys if k in loc}

        def glob(parts, env_dim):
            # parts: [world, K, ..., E, ...] per chunk -> [sum K, ..., world * E, ...] with the rank axis merged into the env axis
            t = torch.cat(parts, dim=1)
            t = t.movedim(0, env_dim)  # [K, ..., world, E, ...]
            shape = list(t.shape)
            shape[env_dim:env_dim + 2] = [shape[env_dim] * shape[env_dim + 1]]
            return t.reshape(shape)

        f32 = glob(chunks["f32"], 2)  # [S, 3, world * E]
        i32 = glob(chunks["i32"], 2)
        out = {k: f32[:, i] for i, k in enumerate(self.F32)}
        out.update({k: i32[:, i].to(torch.float32) for i, k in enumerate(self.U8)})  # the reference stores actions / masks as float32
        if self.with_observations:
            out["grid_codes"] = glob(chunks["codes"], 1)
            out["metadata_states"] = glob(chunks["meta"], 1).to(torch.float32)
        return out

    def gather_once(self, t):
        """Blocking all-gather of one [E, ...] tensor -> [world * E, ...] (the rollout's next_* tensors)."""
        import torch

        if not self.collective:
            return t
        import torch.distributed as dist

        t = t.contiguous()
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=self.group)
        return out


def gather_rank_times(rank, my_ms, world):
    """-> (ranks_seen, per_rank_ms): what every rank of the job reports for itself, all-gathered (every rank gets the lists)."""
    if world == 1:
        return [rank], [my_ms]
    import torch.distributed as dist

    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, (rank, my_ms))
    return [g[0] for g in got], [g[1] for g in got]


def ranks_complete(ranks_seen, per_rank_ms, n):
    """True iff ranks 0..n-1 each reported exactly one positive time: a multi-GPU bench line verifies itself with this and
    exits non-zero otherwise."""
    return (sorted(ranks_seen) == list(range(n)) and len(per_rank_ms) == n
            and all(t is not None and t == t and t > 0 for t in per_rank_ms))


def max_over_ranks(value, device, world):
    """MAX over ranks of a python float (the timing rule of bench.py)."""
    if world == 1:
        return float(value)
    import torch
    import torch.distributed as dist

    if dist.get_backend() == "gloo":
        device = "cpu"  # (the one-GPU rehearsal runs N ranks over gloo: its collectives take host tensors)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def max_over_ranks_list(values, device, world):
    """Element-wise MAX over ranks of a list of python floats (bench.py: one entry per timed window)."""
    vals = [float(v) for v in values]
    if world == 1:
        return vals
    import torch
    import torch.distributed as dist

    if dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor(vals, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t.cpu()]

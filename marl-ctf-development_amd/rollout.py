"""Batched rollout collection: the counterpart of ``PPOTrainer.get_single_rollout`` (reference ppo.py:31-131)
for E envs at once, entirely on the GPU.

What it reproduces from the reference, per env (column ``e`` of every returned tensor is exactly what the
reference's ``get_single_rollout`` returns for that env, checked against a recorded reference rollout):

* team-0 agents see the raw grid, team-1 agents the flipped one, and team-1 agents' sampled actions are mapped
  back through ``REVERSED_ACTION_MAP`` before they reach ``step`` (ppo.py:69,80-83,87,90-93) — the stored action is
  the un-mapped one (ppo.py:77);
* ``use_action_mask`` per agent type (ppo.py:68,86);
* only the trained team's (grid, metadata, action, mask, logprob, value, reward) are stored, agent-major inside an env
  step: slot ``t * A + k`` for the k-th trained agent in ascending agent index (ppo.py:74-84, :105-109), with
  ``A = N // 2`` and ``num_steps`` env steps per rollout (the reference's ``self.num_steps = args.num_steps * A``);
* ``dones`` is never written and stays zero (ppo.py:53); ``next_*`` come from the lowest-index trained agent after the
  last step (ppo.py:117-119), ``next_done`` from the last ``step``.

The policy networks are the caller's (stock PyTorch modules with the reference's
``get_action_and_value(grid, metadata, use_action_mask)`` signature, agent_network.py:63-81); GAE and the PPO update
stay the reference's code — they can consume the returned tensors as they are.
"""
import ctypes as C

import numpy as np

try:
    from . import _abi
    from .gridworld_ctf import _REVERSED_ACTIONS
except ImportError:  # pragma: no cover
    import _abi
    from gridworld_ctf import _REVERSED_ACTIONS


class BatchedRolloutCollector:
    def __init__(self, vec, num_steps, team_to_train, obs_dtype=None, compact=None):
        """vec: VecGridworldCtf.  num_steps: ENV steps per rollout (the reference's ``args.num_steps``).
        obs_dtype: dtype of the stored grid states (default uint8, the env's native output; the reference stores float32).
        compact: feed the policies and fill the rollout buffer with the compact observation (``observe_codes``: one byte
        per cell instead of C one-hot bytes; the rollout then holds ``grid_codes`` [S, E, G, G] and ``expand_codes`` gives
        the planes of any minibatch).  None = whenever both policies offer ``act_from_codes`` (policy_native.py)."""
        import torch

        self.torch = torch
        self.vec = vec
        self.T = int(num_steps)
        self.team = int(team_to_train)
        self.compact = compact
        n, dev = vec.N_AGENTS, vec.device
        teams = [vec.AGENT_TEAMS[i] for i in range(n)]
        types = [vec.AGENT_TYPES[i] for i in range(n)]
        self.trained = [i for i in range(n) if teams[i] == self.team]
        self.others = [i for i in range(n) if teams[i] != self.team]
        self.A = n // 2  # the reference's num_agents_per_team
        if len(self.trained) != self.A:
            raise ValueError("teams must be balanced (the reference assumes N // 2 trained agents)")
        self.trained_idx = torch.tensor(self.trained, device=dev)
        self.others_idx = torch.tensor(self.others, device=dev)
        flag = {0: 1.0, 1: 1.0, 2: 0.0, 3: 0.0}  # AGENT_TYPE_ACTION_MASK (gridworld_ctf.py:218-223)
        self.mask_flag = torch.tensor([flag[t] for t in types], dtype=torch.float32, device=dev)
        flip = vec.derived["flip_axis"]
        self.rev_lut = torch.tensor(_REVERSED_ACTIONS[flip], dtype=torch.int8, device=dev)
        self.is_team1 = torch.tensor([t == 1 for t in teams], device=dev)
        self.obs_dtype = obs_dtype or torch.uint8
        E, S = vec.n_envs, self.T * self.A
        m = vec.META_LEN
        self.grid_states = None  # [S, E, C, G, G], allocated by the first plane-mode collect
        self.grid_codes = None   # [S, E, G, G], allocated by the first compact collect
        self.metadata_states = torch.zeros((S, E, m), dtype=torch.float32, device=dev)
        self.actions = torch.zeros((S, E), dtype=torch.float32, device=dev)
        self.use_action_mask = torch.zeros((S, E), dtype=torch.float32, device=dev)
        self.logprobs = torch.zeros((S, E), dtype=torch.float32, device=dev)
        self.rewards = torch.zeros((S, E), dtype=torch.float32, device=dev)
        self.dones = torch.zeros((S, E), dtype=torch.float32, device=dev)
        self.values = torch.zeros((S, E), dtype=torch.float32, device=dev)
        self._env_actions = torch.zeros((E, n), dtype=torch.int8, device=dev)
        # compact mode with two distinct native networks, OFF by default: the opponent's conv front on a second stream beside the
        # trained team's fc1 GEMM + head.  Identical results (tested), but measured slower — 22.6 M against 25.0 M env-steps/s at
        # 65 536 envs: the front writes and the GEMM reads the same 2.2 GB per team at once and the front at one wave per SIMD does not
        # tolerate co-resident waves (profiles/r03_policy_roofline.md)
        self.overlap_teams = False
        self._side_stream = None
        # compact mode: the per-step bookkeeping (ppo.py:74-93 — what is stored per trained agent, the joint action with team-1 agents'
        # actions mapped back) as ONE launch (ctf_rollout_store_step) instead of ~17 small tensor kernels (0.15 of a 2.1 ms step)
        self.native_store = True
        self._store_args = None

    def use_codes(self, agent, opponent):
        if self.compact is None:
            return hasattr(agent, "act_from_codes") and hasattr(opponent, "act_from_codes")
        return bool(self.compact)

    def _policy(self, net, obs, meta, idx):
        """Run one policy over agents `idx` of every env: batch = E * len(idx), agent-major."""
        torch = self.torch
        E = self.vec.n_envs
        grid = obs.index_select(1, idx).transpose(0, 1).reshape((-1,) + tuple(obs.shape[2:])).to(torch.float32)
        md = meta.index_select(1, idx).transpose(0, 1).reshape(-1, meta.shape[2]).to(torch.float32)
        mask = self.mask_flag.index_select(0, idx)[:, None].expand(-1, E).reshape(-1)
        action, logprob, _, value = net.get_action_and_value(grid, md, mask)
        k = idx.numel()
        return (action.reshape(k, E), logprob.reshape(k, E), value.reshape(k, E), grid.reshape((k, E) + tuple(obs.shape[2:])),
                md.reshape(k, E, -1), mask.reshape(k, E))

    def _policy_codes(self, net, codes, meta, idx, agents, want_inputs=True):
        """The same over the compact observation: the network reads the env's code bytes in place.  (`agents` = `idx` as a
        host list: no device round trip per step.)"""
        E, k = self.vec.n_envs, len(agents)
        mask = self.mask_flag.index_select(0, idx)[:, None].expand(-1, E).reshape(-1)
        # one team, default reversal (by team): its agents look at the same tile planes
        shared = len({self.vec.AGENT_TEAMS[i] for i in agents}) == 1
        action, logprob, _, value = net.act_from_codes(codes, meta, agents, mask, shared_view=shared, self_cells=self.vec.self_cells)
        grid = md = None
        if want_inputs:  # views where the agents form a regular slice: the rollout buffer is then filled by ONE strided copy
            sl = self._as_slice(agents)
            grid = (codes[:, sl] if sl is not None else codes.index_select(1, idx)).transpose(0, 1)
            md = (meta[:, sl] if sl is not None else meta.index_select(1, idx)).transpose(0, 1)
        return action.reshape(k, E), logprob.reshape(k, E), value.reshape(k, E), grid, md, mask.reshape(k, E)

    def _two_teams_overlapped(self, agent, opponent, codes, meta):
        """_policy_codes for both teams with the opponent's kernels on a side stream, started when the trained team's conv front is
        through: its front then runs beside the trained team's fc1 GEMM and head.  Same kernels on the same inputs as the
        one-stream order (each network keeps its own sampler offset): the results are identical."""
        torch, vec = self.torch, self.vec
        E, dev = vec.n_envs, codes.device
        main = torch.cuda.current_stream(dev)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=dev)
        side = self._side_stream
        one_team = lambda agents: len({vec.AGENT_TEAMS[i] for i in agents}) == 1
        mask_a = self.mask_flag.index_select(0, self.trained_idx)[:, None].expand(-1, E).reshape(-1)
        mask_o = self.mask_flag.index_select(0, self.others_idx)[:, None].expand(-1, E).reshape(-1).contiguous()
        feats_a = agent._features_tuned(codes, meta, self.trained, one_team(self.trained), vec.self_cells)
        side.wait_stream(main)  # codes, metadata and mask are there, and the trained team's front has had the chip to itself
        with torch.cuda.stream(side):
            feats_o = opponent._features_tuned(codes, meta, self.others, one_team(self.others), vec.self_cells)
            o_act = opponent._tail(feats_o, mask=mask_o)[0]
        action, logprob, _, value, _ = agent._tail(feats_a, mask=mask_a)
        main.wait_stream(side)
        o_act.record_stream(main)
        k = len(self.trained)
        sl = self._as_slice(self.trained)
        grid = (codes[:, sl] if sl is not None else codes.index_select(1, self.trained_idx)).transpose(0, 1)
        md = (meta[:, sl] if sl is not None else meta.index_select(1, self.trained_idx)).transpose(0, 1)
        return ((action.reshape(k, E), logprob.reshape(k, E), value.reshape(k, E), grid, md, mask_a.reshape(k, E)),
                o_act.reshape(len(self.others), E))

    @staticmethod
    def _as_slice(v):
        if len(v) == 1:
            return slice(v[0], v[0] + 1)
        step = v[1] - v[0]
        if step > 0 and all(b - a == step for a, b in zip(v, v[1:])):
            return slice(v[0], v[-1] + 1, step)
        return None

    def joint_actions(self, agent, opponent, use_codes):
        """One decision of every agent of every env -> (what the trained team's policy returned, env actions int8 [E, N]):
        team-1 agents' actions are mapped back through the flip (ppo.py:80-83,90-93)."""
        torch, vec = self.torch, self.vec
        if use_codes:
            codes, meta = vec.observe_codes()  # default reversal: team(i) == 1
            if (self.overlap_teams and agent is not opponent and codes.is_cuda and hasattr(agent, "_features_tuned")
                    and hasattr(opponent, "_features_tuned")):
                trained, o_act = self._two_teams_overlapped(agent, opponent, codes, meta)
            else:
                trained = self._policy_codes(agent, codes, meta, self.trained_idx, self.trained)
                o_act = self._policy_codes(opponent, codes, meta, self.others_idx, self.others, want_inputs=False)[0]
        else:
            obs, meta = vec.observe()
            trained = self._policy(agent, obs, meta, self.trained_idx)
            o_act = self._policy(opponent, obs, meta, self.others_idx)[0]
        env_act = self._env_actions
        env_act[:, self.trained_idx] = trained[0].to(torch.int8).transpose(0, 1)
        env_act[:, self.others_idx] = o_act.to(torch.int8).transpose(0, 1)
        mapped = self.rev_lut[env_act.long()]
        return trained, torch.where(self.is_team1[None, :], mapped, env_act).contiguous()

    def _step_native(self, agent, opponent, t):
        """One decision of every agent from the compact observation and its bookkeeping in one launch: rows t * A .. of the rollout
        buffers and the env's joint action (``self._env_actions``)."""
        torch, vec, A = self.torch, self.vec, self.A
        E, N = vec.n_envs, vec.N_AGENTS
        codes, meta = vec.observe_codes()  # default reversal: team(i) == 1
        assert codes.dtype == torch.uint8 and meta.dtype == torch.float16 and codes.is_contiguous() and meta.is_contiguous()
        one_team = lambda agents: len({vec.AGENT_TEAMS[i] for i in agents}) == 1
        if self._store_args is None:
            lib = _abi.load_library()
            mask_of = lambda idx: self.mask_flag.index_select(0, idx)[:, None].expand(-1, E).reshape(-1).contiguous()
            team1 = sum(1 << i for i in range(N) if vec.AGENT_TEAMS[i] == 1)
            self._store_args = dict(lib=lib, trained=(C.c_int32 * A)(*self.trained), others=(C.c_int32 * len(self.others))(*self.others),
                                    lut=(C.c_uint8 * 9)(*[int(x) for x in self.rev_lut.tolist()]), team1=team1,
                                    mask_t=mask_of(self.trained_idx), mask_o=mask_of(self.others_idx),
                                    shared_t=one_team(self.trained), shared_o=one_team(self.others))
        sa = self._store_args
        a_act, a_lp, _, a_val = agent.act_from_codes(codes, meta, self.trained, sa["mask_t"], shared_view=sa["shared_t"], self_cells=vec.self_cells)
        o_act = opponent.act_from_codes(codes, meta, self.others, sa["mask_o"], shared_view=sa["shared_o"], self_cells=vec.self_cells)[0]
        i32 = lambda x: x.reshape(-1).to(torch.int32).contiguous()
        f32 = lambda x: x.reshape(-1).to(torch.float32).contiguous()
        a_act, o_act, a_lp, a_val = i32(a_act), i32(o_act), f32(a_lp), f32(a_val)
        sl = slice(t * A, (t + 1) * A)
        ptr = lambda x: C.c_void_p(x.data_ptr())
        g = vec.GRID_SIZE
        rc = sa["lib"].ctf_rollout_store_step(
            ptr(codes), ptr(meta), E, N, g * g, vec.META_LEN, sa["trained"], A, sa["others"], len(self.others), ptr(a_act), ptr(a_lp), ptr(a_val),
            ptr(o_act), sa["lut"], sa["team1"], ptr(self.grid_codes[sl]), ptr(self.metadata_states[sl]), ptr(self.actions[sl]),
            ptr(self.logprobs[sl]), ptr(self.values[sl]), ptr(self._env_actions), vec.device.index,
            C.c_void_p(torch.cuda.current_stream(vec.device).cuda_stream))
        if rc != 0:
            raise _abi.CtfLibraryError("ctf_rollout_store_step: " + (sa["lib"].ctf_policy_last_error() or b"").decode())

    def preallocate(self, agent, opponent):
        """The observation part of the rollout buffer (29 GB of codes for 65 536 envs x 500 steps), allocated on the first collect
        otherwise — a hipMalloc of tens of GB takes 0.1-1.5 s: a caller that times its first ``collect`` calls this before.  -> whether
        the collector runs in compact mode for these two policies."""
        torch, vec = self.torch, self.vec
        use_codes = self.use_codes(agent, opponent)
        E, S, g = vec.n_envs, self.T * self.A, vec.GRID_SIZE
        if use_codes and self.grid_codes is None:
            self.grid_codes = torch.zeros((S, E, g, g), dtype=torch.uint8, device=vec.device)
        if not use_codes and self.grid_states is None:
            self.grid_states = torch.zeros((S, E, vec.N_CHANNELS, g, g), dtype=self.obs_dtype, device=vec.device)
        return use_codes

    def collect(self, agent, opponent, reset=True, handoff=None, handoff_chunk=16):
        """-> dict with the tensors ``get_single_rollout`` returns, each with an env axis after the slot axis.  In compact
        mode ``grid_codes`` / ``next_grid_codes`` stand in for ``grid_states`` / ``next_grid_state``.

        ``handoff``: a ``sharding.RolloutHandoff`` — the N > 1 path for a centralised learner.  Every ``handoff_chunk`` env
        steps the chunk's compact slots (reward, log-prob, value, action, mask; with ``with_observations`` also codes and
        metadata) are all-gathered asynchronously while the next steps run; the returned dict then carries under
        ``"global"`` the same tensors for ALL ranks' envs in global env order ([S, world * E, ...]) plus the gathered
        ``next_*`` tensors.  Compact mode only (the one-hot planes are 14x the bytes; a learner expands codes per minibatch)."""
        torch, vec, A = self.torch, self.vec, self.A
        use_codes = self.preallocate(agent, opponent)
        E, S, g = vec.n_envs, self.T * self.A, vec.GRID_SIZE
        if handoff is not None and handoff.with_observations and not use_codes:
            raise ValueError("the rollout hand-off carries observations in compact form only (use policies with act_from_codes)")
        local = dict(rewards=self.rewards, logprobs=self.logprobs, values=self.values, actions=self.actions,
                     use_action_mask=self.use_action_mask, metadata_states=self.metadata_states)
        if use_codes:
            local["grid_codes"] = self.grid_codes
        sent = 0
        if reset:
            vec.reset()  # ppo.py:57
        self.dones.zero_()
        done = None
        # the one-launch bookkeeping (ctf_rollout_store_step) takes raw pointers: contiguous uint8 codes / float16 metadata, N <= 16 agents,
        # at most 8 per list, metadata rows of <= 64 elements; any other configuration keeps the tensor-expression path below
        fast = bool(use_codes and self.native_store and vec.device.type == "cuda" and not self.overlap_teams
                    and vec.meta.dtype == torch.float16 and vec.meta.is_contiguous() and vec.codes.dtype == torch.uint8
                    and vec.codes.is_contiguous() and vec.N_AGENTS <= 16 and A <= 8 and len(self.others) <= 8 and vec.META_LEN <= 64)
        if fast:  # the masking decision of a trained slot is a constant of the collector: written once, not per step
            self.use_action_mask.view(self.T, A, E)[:] = self.mask_flag.index_select(0, self.trained_idx)[None, :, None]
        with torch.no_grad():
            for t in range(self.T):
                if fast:
                    self._step_native(agent, opponent, t)
                    rewards, done = vec.step(self._env_actions)
                    self.rewards[t * A:(t + 1) * A] = rewards.index_select(1, self.trained_idx).transpose(0, 1)
                    if handoff is not None and ((t + 1) % handoff_chunk == 0 or t + 1 == self.T):
                        handoff.launch(sent, (t + 1) * A, local)
                        sent = (t + 1) * A
                    continue
                (a_act, a_lp, a_val, a_grid, a_md, a_mask), env_act = self.joint_actions(agent, opponent, use_codes)
                sl = slice(t * A, (t + 1) * A)
                if use_codes:
                    self.grid_codes[sl] = a_grid
                else:
                    self.grid_states[sl] = a_grid.to(self.obs_dtype)
                self.metadata_states[sl] = a_md
                self.values[sl] = a_val
                self.actions[sl] = a_act.to(torch.float32)
                self.use_action_mask[sl] = a_mask
                self.logprobs[sl] = a_lp
                rewards, done = vec.step(env_act)
                self.rewards[sl] = rewards.index_select(1, self.trained_idx).transpose(0, 1)
                if handoff is not None and ((t + 1) % handoff_chunk == 0 or t + 1 == self.T):
                    handoff.launch(sent, (t + 1) * A, local)  # async: runs beside the next steps
                    sent = (t + 1) * A
            first = self.trained[0]
            if use_codes:
                codes, meta = vec.observe_codes()
                next_obs = dict(next_grid_codes=codes[:, first].clone())
            else:
                obs, meta = vec.observe()
                next_obs = dict(next_grid_state=obs[:, first].to(torch.float32))
            next_meta = meta[:, first].to(torch.float32)
            next_done = done.to(torch.float32)
        grids = dict(grid_codes=self.grid_codes) if use_codes else dict(grid_states=self.grid_states)
        extra = {}
        if handoff is not None:
            glob = handoff.result()
            glob["dones"] = torch.zeros_like(glob["rewards"])  # never written in the reference (ppo.py:53)
            glob["next_done"] = handoff.gather_once(next_done)
            if handoff.with_observations:
                glob["next_metadata_state"] = handoff.gather_once(next_meta)
                glob["next_grid_codes"] = handoff.gather_once(next_obs["next_grid_codes"])
            extra["global"] = glob
        return dict(**extra, metadata_states=self.metadata_states, actions=self.actions, use_action_mask=self.use_action_mask,
                    logprobs=self.logprobs, rewards=self.rewards, dones=self.dones, values=self.values,
                    next_metadata_state=next_meta, next_done=next_done, **grids, **next_obs)

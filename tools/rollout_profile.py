#!/usr/bin/env python3
"""Kernel-level profile (torch.profiler) of the self-play rollout's steps: what runs beside the env and policy kernels.
    python tools/rollout_profile.py [envs] [steps]      (GPU box)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity

pkg = importlib.import_module("marl-ctf-development_amd")
E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
vec = pkg.VecGridworldCtf(E, device=0, **kw)
dev = torch.device("cuda", 0)
nets = [pkg.policy_native.CtfPolicyNative(9, vec.N_CHANNELS, vec.GRID_SIZE, vec.META_LEN).to(dev).prepare() for _ in range(2)]
col = pkg.BatchedRolloutCollector(vec, T, 0)
col.collect(*nets)
torch.cuda.synchronize()
t0 = time.perf_counter()
col.collect(*nets)
torch.cuda.synchronize()
print(f"rollout {T} steps: {(time.perf_counter() - t0) / T * 1e3:.3f} ms per step")
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    col.collect(*nets)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))

#!/usr/bin/env python3
"""Phase timing of k_policy_features_fact (profiling only): builds a -DPOL_TRACE=1 library, runs the factored fc1 path for one team
and prints the s_memtime deltas (100 MHz ticks) between the phase stamps of wave 0 / block 0."""
import ctypes
import importlib
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "marl-ctf-development_amd", "csrc")
so = os.path.join(ROOT, "tools", "_ab", "libctf_hip_trace_fact.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
                           "-ffp-contract=off", "-DPOL_TRACE=1", "-shared", "-o", so] + [os.path.join(src, f) for f in
                          ("ctf_abi.hip", "ctf_kernels.hip", "ctf_policy.hip", "ctf_policy_fact.hip")])
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.exit(0)
os.environ["CTF_LIB_PATH"] = so
pkg = importlib.import_module("marl-ctf-development_amd")
E = 65536
kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
vec = pkg.VecGridworldCtf(E, device=0, **kw)
acts = torch.zeros((E, 8), dtype=torch.int8, device="cuda")
for t in range(40):
    vec.random_actions(acts, 3, t)
    vec.step(acts)
codes, meta = vec.observe_codes()
net = pkg.policy_native.CtfPolicyNative(9, vec.N_CHANNELS, 15, vec.META_LEN).cuda().prepare()
team = [i for i in range(8) if vec.AGENT_TEAMS[i] == 0]
for _ in range(3):
    net.fc1_from_codes_factored(codes, meta, team, vec.self_cells)
torch.cuda.synchronize()
lib = ctypes.CDLL(so)
buf = (ctypes.c_uint64 * (16 * 64))()
assert lib.ctf_policy_fact_trace_read(buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
names = ["h0 update", "conv1 shared", "wait own cells / meta (+ drain)", "conv1 patches x4", "conv2 shared + view stores", "conv2 patches x4 + row stores"]
n = int((t[:, 0] > 0).sum())
d = np.diff(t[:n, :7], axis=1)
nxt = t[1:n, 0] - t[:n - 1, 6]
print("envs traced", n, " ticks are 10 ns (100 MHz)")
for i, nm in enumerate(names):
    print(f"{nm:34s} mean {d[:, i].mean() * 10:8.0f} ns   min {d[:, i].min() * 10:6d}  max {d[:, i].max() * 10:6d}")
print(f"{'loop back':34s} mean {nxt.mean() * 10:8.0f} ns")
print("per env total", (t[1:n, 0] - t[:n - 1, 0]).mean() * 10, "ns")

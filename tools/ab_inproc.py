"""A/B of several builds of the library IN ONE PROCESS on the SAME output buffers (placement of a large allocation
alone moves the render by 20 %, so separate processes cannot be compared).  Usage (GPU box):
    python tools/ab_inproc.py "<hipcc -D flags of build A>" "<flags of build B>" ...
"""
import ctypes, importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if not os.environ.get("AB_BUILD_ONLY"):
    import torch
pkg = importlib.import_module("marl-ctf-development_amd")
abi = pkg._abi
CS = os.path.join(ROOT, "marl-ctf-development_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "ab")
os.makedirs(OUT, exist_ok=True)
E = int(os.environ.get("AB_ENVS", 65536))
kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
vecs = []
ENVS = {}  # variant -> environment variables set around its calls ("%VAR=VAL %VAR2=VAL2 <flags>")
for i, flags in enumerate(sys.argv[1:]):
    so = os.path.join(OUT, f"lib{i}.so")
    src = CS
    full = flags
    ENVS[full] = {}
    while flags.startswith("%"):
        tok, _, flags = flags[1:].partition(" ")
        k, _, v = tok.partition("=")
        ENVS[full][k] = v
    if flags.startswith("@"):  # "@dir [flags]": build another copy of csrc (e.g. tools/_prev_csrc = the previous commit)
        src, _, flags_only = flags[1:].partition(" ")
        src = os.path.join(ROOT, src)
    else:
        flags_only = flags
    pre = os.path.join(ROOT, "tools", "_ab", f"lib{i}.so")
    if os.environ.get("AB_PREBUILT") and os.path.exists(pre):
        so = pre  # built in the CPU container by `AB_BUILD_ONLY=1 python tools/ab_inproc.py ...` (hipcc cross-compiles; no GPU minutes)
    else:
        if os.environ.get("AB_BUILD_ONLY"):
            os.makedirs(os.path.dirname(pre), exist_ok=True)
            so = pre
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wl,-Bsymbolic",
                               "-I" + os.path.join(ROOT, "marl-ctf-development_amd", "csrc")]
                              + flags_only.split() + ["-shared", "-o", so, os.path.join(src, "ctf_abi.hip"), os.path.join(src, "ctf_kernels.hip")])
    if os.environ.get("AB_BUILD_ONLY"):
        continue
    os.environ.update(ENVS[full])  # variables that are read when the handle is created (CTF_STEP_W)
    vecs.append((full, pkg.VecGridworldCtf(E, device=0, tune_placement=(i == 0 and not os.environ.get("AB_NOTUNE")), _lib=abi.bind(so, mode=ctypes.RTLD_LOCAL, optional=("ctf_policy_", "ctf_rollout_", "ctf_set_rng_states", "ctf_get_rng_states")), **kw)))
    for k in ENVS[full]:
        del os.environ[k]
if os.environ.get("AB_BUILD_ONLY"):
    sys.exit(0)
shared_obs, shared_meta = vecs[0][1].obs, vecs[0][1].meta
acts = torch.zeros((E, 8), dtype=torch.int8, device="cuda")
vecs[0][1].random_actions(acts, seed=5, step=0)
for t in range(int(os.environ.get("AB_PRESTEP", 0))):  # spread the agents over the map before timing anything
    for flags, v in vecs:
        v.random_actions(acts, seed=5, step=t)
        v.step(acts, auto_reset=True)
if os.environ.get("AB_BENCHLIKE"):  # bench.py's protocol: a discarded first episode, staggered episode phases, fresh actions every step
    phase = torch.arange(E, device="cuda") % 500
    for flags, v in vecs:
        for s_ in range(500):
            v.random_actions(acts, seed=0x5747, step=s_)
            v.step(acts, auto_reset=True)
            v.reset((phase == s_).to(torch.uint8))
ACTS = torch.empty((64, E, 8), dtype=torch.int8, device="cuda")
for t in range(64):
    vecs[0][1].random_actions(ACTS[t], seed=0xC7F, step=t)
if not os.environ.get("AB_BENCHLIKE"):
    ACTS[:] = acts
res = {f: ([], [], [], []) for f, _ in vecs}
os.environ["CTF_FUSED"] = os.environ.get("AB_FUSED", "1")  # step_observe below: the single launch where the build has one
for rnd in range(4):
    for flags, v in vecs:
        os.environ.update(ENVS[flags])
        v.obs, v.meta = shared_obs, shared_meta
        for t in range(10):
            v.step(ACTS[t], auto_reset=True); v.observe()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(60)]
        for t, (a, b, c) in enumerate(ev):
            a.record(); v.step(ACTS[t % 64], auto_reset=True); b.record(); v.observe(); c.record()
        ev2 = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(60)]
        for t, (a, b) in enumerate(ev2):
            a.record(); v.step_observe(ACTS[(t + 7) % 64], auto_reset=True); b.record()
        torch.cuda.synchronize()
        for k in ENVS[flags]:
            del os.environ[k]
        if rnd:
            res[flags][0].append(np.median([a.elapsed_time(b) for a, b, c in ev]))
            res[flags][1].append(np.median([b.elapsed_time(c) for a, b, c in ev]))
            res[flags][2].append(np.median([a.elapsed_time(b) for a, b in ev2]))
            res[flags][3].append(np.mean([a.elapsed_time(b) for a, b, c in ev]))  # the step launch's MEAN (its tail blocks come in bursts)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
for a, b in ev:
    a.record(); shared_obs.fill_(1); b.record()
torch.cuda.synchronize()
print("reference: torch fill_ of the same buffer %.4f ms" % np.median([a.elapsed_time(b) for a, b in ev]), flush=True)
for flags, (st, ob, so, sm) in res.items():
    print(f"[{flags or 'default'}] step {np.mean(st):.4f} ms (mean of all launches {np.mean(sm):.4f})  observe {np.mean(ob):.4f} ms  step_observe {np.mean(so):.4f} ms  "
          f"(observe rounds: {', '.join('%.4f' % x for x in ob)})", flush=True)

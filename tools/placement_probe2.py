#!/usr/bin/env python3
"""Which way of LOOKING for a fast observation buffer finds one within a bounded amount of held memory?  (about one
1.65 GB hipMalloc allocation in ten is of the kind the render streams into at full speed; the shipped search holds up to 8 candidates.)

  held      k candidates allocated one after the other, all kept (the shipped search)
  release   allocate, time, free (torch.cuda.empty_cache) — the driver is expected to hand the same memory out again
  spacer    allocate, time, free, then pin a small spacer allocation (64-320 MiB) so that the next candidate cannot land on the very
            same physical range; holds one candidate + the spacers at a time
  churn     allocate and free a few odd-sized blocks between candidates (nothing kept)

    python tools/placement_probe2.py [tries]          (GPU box)
"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("marl-ctf-development_amd")
tries = int(sys.argv[1]) if len(sys.argv) > 1 else 16
kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
vec = pkg.VecGridworldCtf(65536, device=0, tune_placement=False, **kw)
stream = torch.cuda.current_stream()
shape = (vec.n_envs, vec.N_AGENTS, vec.N_CHANNELS, vec.GRID_SIZE, vec.GRID_SIZE)


def timed(fn, reps=3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    a.record(stream)
    for _ in range(reps):
        fn()
    b.record(stream)
    b.synchronize()
    return a.elapsed_time(b) / reps


def probe(buf):
    vec.obs = buf
    return timed(lambda: vec.observe(meta=False))


def new():
    return torch.empty(shape, dtype=torch.uint8, device="cuda")


def fmt(v):
    return " ".join("%.3f" % x for x in v)


MiB = 1 << 20
first = new()
print("fill_ of one candidate: %.3f ms" % timed(lambda: first.fill_(0)))
del first
torch.cuda.empty_cache()
for rnd in range(2):
    held = [new() for _ in range(min(tries, 12))]
    print("round %d held   :" % rnd, fmt([probe(b) for b in held]), flush=True)
    vec.obs = None
    del held
    torch.cuda.empty_cache()
    out = []
    for i in range(tries):
        b = new()
        out.append(probe(b))
        vec.obs = None
        del b
        torch.cuda.empty_cache()
    print("round %d release:" % rnd, fmt(out), flush=True)
    out, spacers = [], []
    for i in range(tries):
        b = new()
        out.append(probe(b))
        vec.obs = None
        del b
        torch.cuda.empty_cache()
        spacers.append(torch.empty((64 * (1 + i % 5)) * MiB, dtype=torch.uint8, device="cuda"))
    print("round %d spacer :" % rnd, fmt(out), "  (spacers held: %.1f GiB)" % (sum(s.numel() for s in spacers) / 2 ** 30), flush=True)
    del spacers
    torch.cuda.empty_cache()
    out = []
    for i in range(tries):
        junk = [torch.empty((37 + 61 * ((i + k) % 7)) * MiB, dtype=torch.uint8, device="cuda") for k in range(6)]
        b = new()
        del junk
        out.append(probe(b))
        vec.obs = None
        del b
        torch.cuda.empty_cache()
    print("round %d churn  :" % rnd, fmt(out), flush=True)

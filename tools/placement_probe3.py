#!/usr/bin/env python3
"""Does the STATE of the process's device memory decide which kind of observation buffer the driver hands out?  (Round 3, fresh boxes:
on two of ten the headline's search met only slow candidates — 8 and 48 in a row — while the secondaries later in the same process
found fast ones at once.)  One mode per process; every mode ends with `tries` release-style candidates (allocate, time the render, free).

    python tools/placement_probe3.py MODE [tries]          (GPU box)
      plain      nothing before the candidates (what the headline's search sees in a fresh process)
      churn_big  allocate and free buffers of 2.9, 5.8 and 1.2 GB first (what the arena20 / split secondaries do)
      hold_big   allocate a 5.8 GB buffer and KEEP it while searching
      comb       allocate 48 blocks of 160 MB, free every other one, keep the rest while searching
      odd_sizes  candidates padded by 0..7 x 36 MB (their first 1.65 GB used)
"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("marl-ctf-development_amd")
mode = sys.argv[1]
tries = int(sys.argv[2]) if len(sys.argv) > 2 else 24
kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
vec = pkg.VecGridworldCtf(65536, device=0, tune_placement=False, **kw)
stream = torch.cuda.current_stream()
shape = (vec.n_envs, vec.N_AGENTS, vec.N_CHANNELS, vec.GRID_SIZE, vec.GRID_SIZE)
nbytes = 1
for d in shape:
    nbytes *= d
MiB = 1 << 20
u8 = lambda n: torch.empty(int(n), dtype=torch.uint8, device="cuda")


def probe(buf):
    vec.obs = buf
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    vec.observe(meta=False)
    a.record(stream)
    for _ in range(3):
        vec.observe(meta=False)
    b.record(stream)
    b.synchronize()
    vec.obs = None
    return a.elapsed_time(b) / 3


keep = []
if mode == "churn_big":
    for gb in (2.9, 5.8, 1.2):
        t = u8(gb * 1e9)
        t.fill_(1)
        del t
    torch.cuda.empty_cache()
elif mode == "hold_big":
    keep.append(u8(5.8e9))
elif mode == "comb":
    blocks = [u8(160 * MiB) for _ in range(48)]
    keep = blocks[::2]
    del blocks
    torch.cuda.empty_cache()
out = []
for i in range(tries):
    pad = (i % 8) * 36 * MiB if mode == "odd_sizes" else 0
    raw = u8(nbytes + pad)
    out.append(probe(raw[:nbytes].view(shape)))
    del raw
    torch.cuda.empty_cache()
print("%-10s" % mode, " ".join("%.3f" % x for x in out), "  fast: %d of %d" % (sum(x < 0.272 for x in out), len(out)), flush=True)

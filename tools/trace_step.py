#!/usr/bin/env python3
"""Phase trace of k_step in the bench's state: builds the library with -DSTEP_TRACE=1 (lane 0 of every block stamps the
100 MHz wall clock at fixed points of the step), runs the bench protocol for a few steps and prints where one wave's time goes.

    python tools/trace_step.py            (GPU box; AB_BUILD_ONLY=1 prebuilds tools/_ab/libtrace.so in the CPU container)
"""
import ctypes, importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pkg = importlib.import_module("marl-ctf-development_amd")
abi = pkg._abi
CS = os.path.join(ROOT, "marl-ctf-development_amd", "csrc")
so = os.path.join(ROOT, "tools", "_ab", "libtrace%s.so" % os.environ.get("STEP_TRACE", "1"))
if not os.path.exists(so) or os.environ.get("AB_BUILD_ONLY"):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wl,-Bsymbolic",
                           "-I" + CS, "-DSTEP_TRACE=" + os.environ.get("STEP_TRACE", "1")] + sys.argv[1:] + ["-shared", "-o", so, os.path.join(CS, "ctf_abi.hip"), os.path.join(CS, "ctf_kernels.hip")])
if os.environ.get("AB_BUILD_ONLY"):
    sys.exit(0)
import torch
import bench
E = 65536
kw = bench.WORKLOADS["arena"][1](pkg)
lib = abi.bind(so, mode=ctypes.RTLD_LOCAL, optional=("ctf_policy_",))
vec = pkg.VecGridworldCtf(E, device=0, tune_placement=False, _lib=lib, **kw)
acts = torch.empty((64, E, vec.N_AGENTS), dtype=torch.int8, device="cuda")
for t in range(64):
    vec.random_actions(acts[t], seed=0xC7F, step=t)
vec.observe()
bench.stagger_phases(vec, torch, 0, kw["GAME_STEPS"])
raw = ctypes.CDLL(so, mode=ctypes.RTLD_LOCAL)  # same handle: dlopen returns the loaded library
raw.ctf_debug_step_trace.argtypes = [ctypes.c_void_p]
acc = []
for t in range(40):
    vec.step(acts[t % 64], auto_reset=True)
    vec.observe()
    if t >= 8:
        torch.cuda.synchronize()
        buf = np.zeros((8192, 40), dtype=np.uint64)
        assert raw.ctf_debug_step_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        acc.append(buf[:E // 16].astype(np.int64))
bpc, lpc, lpb = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
for lds in (8192, 9920, 10240, 12288):
    rc = raw.ctf_debug_step_occupancy(lds, ctypes.byref(bpc), ctypes.byref(lpc), ctypes.byref(lpb))
    print(f"runtime occupancy query: {lds} B of LDS per block -> {bpc.value} blocks per CU (rc {rc}); device LDS per CU {lpc.value}, per block {lpb.value}")
span = np.zeros((8192, 3), dtype=np.uint64)
raw.ctf_debug_step_span.argtypes = [ctypes.c_void_p]
assert raw.ctf_debug_step_span(span.ctypes.data_as(ctypes.c_void_p)) == 0
span = span[:E // 16].astype(np.int64)
t0 = span[:, 0].min()
st, en = (span[:, 0] - t0) / 100.0, (span[:, 1] - t0) / 100.0
print("last step, all %d blocks: start p0/p50/p90/p99/max = %s us; end p0/p50/p90/max = %s us; life mean %.2f us" % (
    len(span), np.round(np.percentile(st, [0, 50, 90, 99, 100]), 2), np.round(np.percentile(en, [0, 50, 90, 100]), 2), (en - st).mean()))
hw = span[:, 2] & 0xFFFFFFFF
xcc = (span[:, 2] >> 32) & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = xcc * 1000 + se * 100 + sh * 16 + cu
u, cnt = np.unique(key, return_counts=True)
print("distinct (xcc, se, sh, cu):", len(u), " blocks per CU min/mean/max:", cnt.min(), cnt.mean(), cnt.max(), " histogram:", dict(zip(*np.unique(cnt, return_counts=True))))
late = st > 20
print("blocks starting later than 20 us:", int(late.sum()))
if os.environ.get('STEP_TRACE', '1') != '1':
    sys.exit(0)
a = np.stack(acc)  # [steps, blocks, stamps]
names = {0: "start", 1: "staged (loads -> LDS, barrier)", 33: "prologue (positions, reset check, perm / flag bits)", 2: "py refill 1", 3: "shuffle 1", 28: "py refill 2", 29: "shuffle 2", 30: "heal + rewards + vis log",
         31: "flushes + rngpos", 32: "write-back issued"}
for k in range(8):
    names[4 + 3 * k] = f"turn {k}: act"
    names[5 + 3 * k] = f"turn {k}: tagging"
    names[6 + 3 * k] = f"turn {k}: metrics"
order = [0, 1, 33, 2, 3] + [4 + i for i in range(24)] + [28, 29, 30, 31, 32]
prev = None
tot = {}
print("phase                                 mean us   p10    p90   (10 ns ticks of the 100 MHz clock, over steps x blocks)")
for k in order:
    if prev is not None:
        d = (a[:, :, k] - a[:, :, prev]).reshape(-1) / 100.0
        print(f"{names[k]:36s} {d.mean():7.2f} {np.percentile(d, 10):6.2f} {np.percentile(d, 90):6.2f}")
        key = names[k].split(": ")[-1] if names[k].startswith("turn") else names[k]
        tot[key] = tot.get(key, 0.0) + d.mean()
    prev = k
print("sum by kind:", {k: round(v, 2) for k, v in tot.items()})
life = (a[:, :, 32] - a[:, :, 0]) / 100.0
print("whole (stamp 0 -> 32): mean %.2f us, p50 %.2f, p90 %.2f, p99 %.2f, max %.2f" % (life.mean(), *np.percentile(life, [50, 90, 99, 100])))
# where do the slowest 2 % of the waves lose their time against the median half?
thr_hi, thr_lo = np.percentile(life, 98), np.percentile(life, 50)
slow, fast = life >= thr_hi, life <= thr_lo
print("slowest 2 %% of the waves (>= %.1f us) against the faster half, per phase:" % thr_hi)
prev = None
for k in order:
    if prev is not None:
        d = (a[:, :, k] - a[:, :, prev]) / 100.0
        ds, df = d[slow].mean(), d[fast].mean()
        if abs(ds - df) > 0.3:
            print(f"  {names[k]:36s} slow {ds:6.2f}  fast {df:6.2f}  (+{ds - df:.2f})")
    prev = k

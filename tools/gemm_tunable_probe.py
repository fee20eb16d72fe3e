#!/usr/bin/env python3
"""Do the library's GEMMs of the learner / the view GEMM get faster when PyTorch's TunableOp picks the hipBLASLt / rocBLAS solution by
measurement instead of the heuristic's first answer?  (profiling only)    python tools/gemm_tunable_probe.py [rows]"""
import json, os, sys, time
import torch
import torch.cuda.tunable as tun

M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
K, N = 4160, 256
dev, bf = "cuda", torch.bfloat16

def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

act = torch.randn(M, K, device=dev).to(bf)
w = torch.randn(N, K, device=dev).to(bf)
bias = torch.randn(N, device=dev).to(bf)
dy = torch.randn(M, N, device=dev).to(bf)
view = torch.randn(65536, 4096, device=dev).to(bf)
wv = torch.randn(4096, 256, device=dev).to(bf)
F = torch.nn.functional
cases = {
    "fc1_fwd_linear_nobias": lambda: F.linear(act, w),
    "fc1_dgrad_mm": lambda: torch.mm(dy, w),
    "fc1_wgrad_mm_T": lambda: torch.mm(dy.t(), act),
    "view_gemm_f32out": lambda: torch.mm(view, wv, out_dtype=torch.float32),
    "view_gemm_bf16out": lambda: torch.mm(view, wv),
}
out = {"rows": M, "untuned": {k: round(timed(f), 4) for k, f in cases.items()}}
tun.enable(True)
tun.tuning_enable(True)
tun.set_max_tuning_duration(int(os.environ.get("TUNE_MS", "400")))
tun.set_max_tuning_iterations(int(os.environ.get("TUNE_ITERS", "20")))
tun.set_filename(os.environ.get("TUNE_FILE", "/tmp/tunableop_results.csv"))
t0 = time.perf_counter()
for k, f in cases.items():
    t1 = time.perf_counter()
    f(); torch.cuda.synchronize()
    print(f"tuned {k} in {time.perf_counter() - t1:.1f} s", file=sys.stderr, flush=True)
out["tuning_s"] = round(time.perf_counter() - t0, 1)
tun.tuning_enable(False)
out["tuned"] = {k: round(timed(f), 4) for k, f in cases.items()}
out["results"] = [r for r in tun.get_results()][:12]
print(json.dumps(out, default=str))

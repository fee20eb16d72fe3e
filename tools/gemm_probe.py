#!/usr/bin/env python3
"""The learner's dense GEMMs in isolation (profiling only): fc1 forward / data gradient / weight gradient at the piece sizes the update
uses, by how they are called.     python tools/gemm_probe.py [rows]"""
import sys, json, torch
M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
K, N = 4160, 256
dev = "cuda"
bf = torch.bfloat16
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
wt = w.t().contiguous()
bias = torch.randn(N, device=dev).to(bf)
dy = torch.randn(M, N, device=dev).to(bf)
F = torch.nn.functional
out = {"rows": M}
fl = 2.0 * M * K * N / 1e12
out["fwd_linear_bias"] = timed(lambda: F.linear(act, w, bias))
out["fwd_linear_nobias"] = timed(lambda: F.linear(act, w))
out["fwd_mm_wt"] = timed(lambda: torch.mm(act, wt))
out["fwd_mm_f32out"] = timed(lambda: torch.mm(act, wt, out_dtype=torch.float32))
out["dgrad_mm"] = timed(lambda: torch.mm(dy, w))
out["wgrad_mm_T"] = timed(lambda: torch.mm(dy.t(), act))
dyT = dy.t().contiguous()
out["wgrad_mm_contigT"] = timed(lambda: torch.mm(dyT, act))
out["wgrad_f32out"] = timed(lambda: torch.mm(dy.t(), act, out_dtype=torch.float32))
out["bias_grad_sum"] = timed(lambda: dy.sum(0))
out["bias_grad_sum_f32"] = timed(lambda: dy.float().sum(0))
ones = torch.ones(1, M, device=dev, dtype=bf)
out["bias_grad_mm_ones"] = timed(lambda: torch.mm(ones, dy))
out["tanh_bwd_elem"] = timed(lambda: dy * (1 - dy * dy))
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}), "TFLOP per GEMM", round(fl, 3))
# autograd path as the learner calls it
w32 = torch.randn(N, K, device=dev, requires_grad=True)
b32 = torch.randn(N, device=dev, requires_grad=True)
def step():
    with torch.autocast("cuda", dtype=bf):
        y = torch.tanh(F.linear(act, w32, b32))
    y.float().sum().backward()
print("autograd fc1+tanh fwd+bwd", round(timed(step, 5), 4))

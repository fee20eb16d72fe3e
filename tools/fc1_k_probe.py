#!/usr/bin/env python3
"""How the library's bf16 fc1 GEMM ([B, K] x [K, 256]) depends on K (the activation row length the policy front chooses)."""
import json
import sys

import torch

F = torch.nn.functional


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


out = {}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
for K in (3904, 3968, 4032, 4096, 4160, 4224, 4288, 4352):
    x = torch.randn((B, K), device="cuda", dtype=torch.bfloat16)
    w = torch.randn((256, K), device="cuda", dtype=torch.bfloat16)
    bias = torch.randn((256,), device="cuda", dtype=torch.bfloat16)
    ms = timed(lambda: F.linear(x, w, bias))
    out[f"K{K}"] = {"ms": round(ms, 4), "tflops": round(2 * B * K * 256 / ms / 1e9, 1)}
    del x
print(json.dumps(out))

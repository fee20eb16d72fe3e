"""ctf_policy_view_gemm against the library's GEMM on the rollout's shape (profiles/r04_policy_fact.md).

    python tools/view_gemm_bench.py [rows] [kv]
"""
import ctypes as C
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
abi = importlib.import_module("marl-ctf-development_amd._abi")


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    kv = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    lib = abi.load_library()
    a = (torch.rand((rows, kv), device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.randn((256, kv), device="cuda") * 0.5).to(torch.bfloat16)
    wt = w.t().contiguous()
    out = torch.empty((rows, 256), device="cuda")
    ref = torch.empty((rows, 256), device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    native = lambda: lib.ctf_policy_view_gemm(C.c_void_p(a.data_ptr()), C.c_void_p(w.data_ptr()), rows, kv, C.c_void_p(out.data_ptr()), 0, st)
    library = lambda: torch.mm(a, wt, out_dtype=torch.float32, out=ref)

    def timed(fn, n=30):
        for _ in range(5):
            fn()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(n):
            fn()
        t1.record()
        t1.synchronize()
        return t0.elapsed_time(t1) / n

    res = {"rows": rows, "kv": kv, "native_ms": round(timed(native), 4), "library_ms": round(timed(library), 4)}
    res["max_abs_diff"] = float((out - ref).abs().max())
    res["native_read_tbs"] = round(rows * kv * 2 / res["native_ms"] / 1e9, 2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

"""The factored path's view GEMM ([E][KV] x [KV][256], float32 out): which call form does the library run fastest?  (profiling only)

    python tools/view_gemm_forms_probe.py [rows] [kv]
"""
import json
import sys

import torch


def timed(fn, n=30):
    for _ in range(5):
        fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        fn()
    t1.record()
    t1.synchronize()
    return round(t0.elapsed_time(t1) / n, 4)


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    kv = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    a = (torch.rand((rows, kv), device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.randn((256, kv), device="cuda") * 0.5).to(torch.bfloat16)   # [256][KV]
    wt = w.t().contiguous()                                                  # [KV][256]
    out = torch.empty((rows, 256), device="cuda")
    res = {"rows": rows, "kv": kv}
    res["nn_f32"] = timed(lambda: torch.mm(a, wt, out_dtype=torch.float32, out=out))
    res["nt_f32"] = timed(lambda: torch.mm(a, w.t(), out_dtype=torch.float32, out=out))
    res["nn_bf16"] = timed(lambda: torch.mm(a, wt))
    res["nt_bf16"] = timed(lambda: torch.mm(a, w.t()))
    res["linear_bf16"] = timed(lambda: torch.nn.functional.linear(a, w))
    for s in (2, 4, 8, 16):
        ab = a.view(s, rows // s, kv)
        ob = out.view(s, rows // s, 256)
        res[f"bmm_nn_f32_s{s}"] = timed(lambda: torch.bmm(ab, wt.expand(s, kv, 256), out_dtype=torch.float32, out=ob))
        res[f"bmm_nt_f32_s{s}"] = timed(lambda: torch.bmm(ab, w.t().expand(s, kv, 256), out_dtype=torch.float32, out=ob))
    # split over K: two half products added
    h = kv // 2
    res["k_halves_f32"] = timed(lambda: torch.addmm(torch.mm(a[:, :h], wt[:h], out_dtype=torch.float32), a[:, h:].float()[:1] * 0, wt[h:].float()[:, :1].t() * 0) if False else
                                torch.mm(a[:, :h], wt[:h], out_dtype=torch.float32).add_(torch.mm(a[:, h:], wt[h:], out_dtype=torch.float32)))
    print(json.dumps(res))


if __name__ == "__main__":
    main()

"""fc1's weight gradient dW = dy^T act: the library's single GEMM (split-K + fix-up) against batched forms over sample ranges
(profiling only; profiles/r04_learner_roofline.md).

    python tools/fc1_wgrad_probe.py [samples] [kp]
"""
import json
import sys

import torch


def timed(fn, n=10):
    for _ in range(3):
        fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        fn()
    t1.record()
    t1.synchronize()
    return round(t0.elapsed_time(t1) / n, 4)


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    kp = int(sys.argv[2]) if len(sys.argv) > 2 else 4160
    dy = (torch.randn((m, 256), device="cuda") * 0.05).to(torch.bfloat16)
    act = torch.tanh(torch.randn((m, kp), device="cuda")).to(torch.bfloat16)
    ref = torch.mm(dy.t(), act).float()
    want = None
    out = {"samples": m, "kp": kp, "mm_bf16_out": timed(lambda: torch.mm(dy.t(), act))}
    try:
        out["mm_f32_out"] = timed(lambda: torch.mm(dy.t(), act, out_dtype=torch.float32))
    except Exception as e:  # noqa: BLE001
        out["mm_f32_out"] = repr(e)[:80]
    for s in (2, 4, 8, 16, 32, 64, 128, 256):
        if m % s or m // s < 2048:
            continue
        a = dy.view(s, m // s, 256).transpose(1, 2)
        b = act.view(s, m // s, kp)
        for name, fn in (("bmm_bf16", lambda: torch.bmm(a, b).float().sum(0)),
                         ("bmm_f32", lambda: torch.bmm(a, b, out_dtype=torch.float32).sum(0))):
            try:
                t = timed(fn)
                err = float((fn() - ref).abs().max())
                out[f"{name}_s{s}"] = [t, round(err, 4)]
            except Exception as e:  # noqa: BLE001
                out[f"{name}_s{s}"] = repr(e)[:80]
    print(json.dumps(out))


if __name__ == "__main__":
    main()

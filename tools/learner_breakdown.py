#!/usr/bin/env python3
"""Where the learner's time goes (configs[4]'s update, learner.PPOLearner.optimise on the 8_arena shapes): sample-passes per
second of one epoch over a synthetic batch, for the two ways of feeding the network (one-hot planes through the stock
convolutions / compact codes: native front as the forward + MIOpen gradients ("codes"), stock channels-last modules
("codes_stock")) and several piece sizes, plus a
forward / backward split of one piece by torch.cuda events.

    python tools/learner_breakdown.py [--samples 1048576]        (GPU box)
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("marl-ctf-development_amd")
learner = importlib.import_module("marl-ctf-development_amd.learner")
C, G, M = 14, 15, 22


def batch(n, dev):
    g = torch.Generator(device="cpu").manual_seed(1)
    codes = torch.randint(0, C, (n, G, G), dtype=torch.uint8, generator=g)
    cell = torch.randint(0, G * G, (n,), generator=g)
    codes.view(n, -1)[torch.arange(n), cell] |= 128
    return dict(grids=codes.to(dev), meta=torch.rand((n, M), generator=g).to(dev).half(), logp=(-torch.rand(n, generator=g) * 2).to(dev),
                act=torch.randint(0, 5, (n,), generator=g).to(dev).to(torch.uint8), mask=torch.randint(0, 2, (n,), generator=g).to(dev).to(torch.uint8),
                adv=torch.randn(n, generator=g).to(dev), ret=torch.randn(n, generator=g).to(dev), val=torch.randn(n, generator=g).to(dev))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=1 << 20)
    ap.add_argument("--pieces", default="16384,65536,262144")
    ap.add_argument("--variants", default="codes,planes")
    ap.add_argument("--benchmark", action="store_true", help="torch.backends.cudnn.benchmark: MIOpen searches its solvers per shape instead of its heuristic pick")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = bool(a.benchmark)
    b = batch(a.samples, dev)
    out = {}
    for variant in a.variants.split(","):
        for piece in [int(x) for x in a.pieces.split(",")]:
            if variant == "planes" and piece > 65536:
                continue  # MIOpen's solver for such batches runs at thousands of samples per second
            torch.manual_seed(0)
            net = pkg.policy_native.CtfPolicyNative(9, C, G, M).to(dev)
            lrn = learner.PPOLearner(net, C, update_epochs=1, num_minibatches=4)
            lrn.codes_direct = variant.startswith("codes")
            net.native_training = variant == "codes"  # codes_stock: the stock channels-last modules also for the forward
            n_warm = min(a.samples, 4 * piece)
            run = lambda n: lrn.optimise(b["grids"][:n], b["meta"][:n], b["logp"][:n], b["act"][:n], b["mask"][:n], b["adv"][:n], b["ret"][:n], b["val"][:n],
                                         micro_batch=piece)
            run(n_warm)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            losses = run(a.samples)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            key = f"{variant}_piece{piece}"
            out[key] = {"sample_passes_per_s": a.samples / dt, "s": dt, "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
                        "losses": [float(x) for x in losses]}
            print(key, json.dumps(out[key]), flush=True)
            # forward / backward of one piece
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            n = piece
            ev[0].record()
            _, lp, ent, v = net.get_action_and_value(lrn._planes(b["grids"][:n]), b["meta"][:n].float(), b["mask"][:n], b["act"][:n].long())
            ev[1].record()
            (lp.sum() + ent.sum() + v.sum()).backward()
            ev[2].record()
            torch.cuda.synchronize()
            out[key].update(forward_ms=ev[0].elapsed_time(ev[1]), backward_ms=ev[1].elapsed_time(ev[2]))
            print("   one piece: forward %.3f ms, backward %.3f ms" % (out[key]["forward_ms"], out[key]["backward_ms"]), flush=True)
            del net, lrn
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Kernel-level profile (torch.profiler) of one forward + backward piece of the learner, per way of feeding the network.
    python tools/learner_profile.py [piece] [variant: planes|codes|planes_cl]      (GPU box)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from torch.profiler import profile, ProfilerActivity
import learner_breakdown as lb

piece = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
variants = [v for v in sys.argv[2].split(",") if v != "none"] if len(sys.argv) > 2 else ["planes", "codes"]
dev = torch.device("cuda", 0)
b = lb.batch(piece, dev)
for variant in variants:
    torch.manual_seed(0)
    net = lb.pkg.policy_native.CtfPolicyNative(9, lb.C, lb.G, lb.M).to(dev)
    if variant == "planes_cl":
        net = net.to(memory_format=torch.channels_last)
    lrn = lb.learner.PPOLearner(net, lb.C)
    lrn.codes_direct = variant.startswith("codes")
    net.native_training = variant == "codes"

    def step():
        x = lrn._planes(b["grids"])
        if variant == "planes_cl":
            x = x.contiguous(memory_format=torch.channels_last)
        _, lp, ent, v = net.get_action_and_value(x, b["meta"].float(), b["mask"], b["act"].long())
        (lp.sum() + ent.sum() + v.sum()).backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            step()
        torch.cuda.synchronize()
    print(f"==== {variant}, piece {piece}: 5 forward + backward passes")
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=90))

if os.environ.get("PROFILE_OPTIMISE"):
    # the whole update (minibatch gathers, losses, clipping, Adam) around the network passes: one epoch over 4 x piece samples
    import time
    n = 4 * piece
    b = lb.batch(n, dev)
    torch.manual_seed(0)
    net = lb.pkg.policy_native.CtfPolicyNative(9, lb.C, lb.G, lb.M).to(dev)
    lrn = lb.learner.PPOLearner(net, lb.C, update_epochs=1, num_minibatches=4)
    run = lambda: lrn.optimise(b["grids"], b["meta"], b["logp"], b["act"], b["mask"], b["adv"], b["ret"], b["val"], micro_batch=piece)
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        run()
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"==== optimise: one epoch over {n} samples in pieces of {piece}: wall {wall * 1e3:.1f} ms")
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=90))

#!/usr/bin/env python3
"""Which GEMMs does one forward + backward piece of the learner launch, with which shapes?  (profiling only)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from torch.profiler import profile, ProfilerActivity
import learner_breakdown as lb
piece = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dev = torch.device("cuda", 0)
b = lb.batch(piece, dev)
net = lb.pkg.policy_native.CtfPolicyNative(9, lb.C, lb.G, lb.M).to(dev)
lrn = lb.learner.PPOLearner(net, lb.C)
def step():
    _, lp, ent, v = net.get_action_and_value(lrn._planes(b["grids"]), b["meta"].float(), b["mask"], b["act"].long())
    (lp.sum() + ent.sum() + v.sum()).backward()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.device_time_total > 0 and (len(sys.argv) > 2 or any(k in e.key for k in ("mm", "linear", "matmul", "sum", "index_select", "mul", "_to_copy")))]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:(60 if len(sys.argv) > 2 else 28)]:
    print(f"{e.key:28s} {e.device_time_total / 1e3:8.3f} ms  x{e.count:<3d} {str(e.input_shapes)[:150]}")

#!/usr/bin/env python3
"""Stage timings of the native policy path on the arena shape: observe_codes, ctf_policy_features, fc1, ctf_policy_head.

    python tools/policy_native_bench.py [envs]        (8 agents per env => B = 8 * envs samples)
"""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    pkg = importlib.import_module("marl-ctf-development_amd")
    native = pkg.policy_native
    kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
    vec = pkg.VecGridworldCtf(E, device=0, **kw)
    acts = torch.zeros((E, vec.N_AGENTS), dtype=torch.int8, device="cuda")
    for t in range(20):
        vec.random_actions(acts, 7, t)
        vec.step(acts)
    net = native.CtfPolicyNative(9, vec.N_CHANNELS, vec.GRID_SIZE, vec.META_LEN).cuda().prepare()
    p = net._prep
    sel = list(range(vec.N_AGENTS))
    B = E * len(sel)
    out = {"envs": E, "B": B}
    with torch.no_grad():
        out["observe_codes"] = timed(lambda: vec.observe_codes())
        out["observe_codes_only"] = timed(lambda: vec.observe_codes(meta=False))
        codes, meta = vec.observe_codes()
        feats = torch.zeros((B, p["kp"]), dtype=torch.bfloat16, device="cuda")
        out["features"] = timed(lambda: net.features_from_codes(codes, meta, sel, out=feats))
        teams = [[i for i in range(vec.N_AGENTS) if vec.AGENT_TEAMS[i] == t] for t in (0, 1)]
        halves = [feats[:len(teams[0]) * E], feats[len(teams[0]) * E:]]
        out["features_two_teams_per_agent"] = timed(lambda: [net.features_from_codes(codes, meta, teams[t], out=halves[t]) for t in (0, 1)])
        out["features_two_teams_shared_view"] = timed(
            lambda: [net.features_from_codes(codes, meta, teams[t], out=halves[t], shared_view=True, self_cells=vec.self_cells) for t in (0, 1)])
        F = torch.nn.functional
        out["fc1"] = timed(lambda: F.linear(feats, p["fc1_w"], p["fc1_b"]))
        y1 = F.linear(feats, p["fc1_w"], p["fc1_b"])
        mask1 = torch.ones(B, device="cuda")
        out["head"] = timed(lambda: net._head(y1, mask=mask1))
        out["trunk_from_codes"] = timed(lambda: net.trunk_from_codes(codes, meta, sel))
        mask = torch.ones(B, device="cuda")
        out["act_from_codes"] = timed(lambda: net.act_from_codes(codes, meta, sel, mask))
        # round 4: fc1 carried through the shared view, stage by stage for the two teams of a step (include/ctf_policy.h)
        import ctypes as C

        lib = p["lib"]
        ptr = lambda t: C.c_void_p(t.data_ptr())
        st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
        G, M, N, sc = vec.GRID_SIZE, vec.META_LEN, vec.N_AGENTS, vec.self_cells
        net.fc1_from_codes_factored(codes, meta, teams[0], sc)
        b = net._act_bufs[("fact", E, len(teams[0]), 0)]
        arr = [(C.c_int32 * len(t))(*t) for t in teams]
        A = len(teams[0])
        out["fact_bucket_x2"] = timed(lambda: [lib.ctf_policy_fact_bucket(ptr(sc), E, N, G, arr[t], A, ptr(b["work"]), ptr(b["slot_of"]),
                                                                           ptr(b["row_of_slot"]), 0, st()) for t in (0, 1)])
        out["fact_front_x2"] = timed(lambda: [lib.ctf_policy_features_fact(ptr(codes), ptr(meta), ptr(sc), E, N, G, M, arr[t], A, ptr(p["f1"]), ptr(p["b1"]),
                                                                            ptr(p["f2"]), ptr(p["b2"]), ptr(b["slot_of"]), ptr(b["view"]), ptr(b["prow"]), 0, st())
                                              for t in (0, 1)])
        out["fact_view_gemm_x2"] = timed(lambda: [torch.bmm(b["view"].view(4, E // 4, b["kv"]), p["fc1_view_w"].t().expand(4, b["kv"], 256), out_dtype=torch.float32,
                                                             out=b["yview"].view(4, E // 4, 256)) for t in (0, 1)])
        out["fact_view_gemm_plain_mm_x2"] = timed(lambda: [torch.mm(b["view"], p["fc1_view_w"].t(), out_dtype=torch.float32, out=b["yview"]) for t in (0, 1)])
        out["fact_view_gemm_native_x2"] = timed(lambda: [lib.ctf_policy_view_gemm(ptr(b["view"]), ptr(p["fc1_view_w"]), E, b["kv"], ptr(b["yview"]), 0, st()) for t in (0, 1)])
        out["fact_patch_x2"] = timed(lambda: [lib.ctf_policy_fc1_patch(ptr(b["prow"]), ptr(b["row_of_slot"]), ptr(b["work"]), ptr(b["yview"]), ptr(p["pf"]),
                                                                        ptr(p["fc1_b32"]), E, A, G, M, ptr(b["y1"]), 0, st()) for t in (0, 1)])
        out["fact_fc1_from_codes_x2"] = timed(lambda: [net.fc1_from_codes_factored(codes, meta, teams[t], sc) for t in (0, 1)])
        masks = [torch.ones(A * E, device="cuda") for _ in (0, 1)]
        out["act_two_teams_factored"] = timed(lambda: [net.act_from_codes(codes, meta, teams[t], masks[t], shared_view=True, self_cells=sc) for t in (0, 1)])
        net.native_view_gemm = False
        out["act_two_teams_factored_library_gemm"] = timed(lambda: [net.act_from_codes(codes, meta, teams[t], masks[t], shared_view=True, self_cells=sc) for t in (0, 1)])
        net.native_view_gemm = True
        net.factored_fc1 = False
        out["act_two_teams_unfactored"] = timed(lambda: [net.act_from_codes(codes, meta, teams[t], masks[t], shared_view=True, self_cells=sc) for t in (0, 1)])
        net.factored_fc1 = True
    out = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}
    out["samples_per_s_act"] = round(B / (out["act_from_codes"] * 1e-3))
    flop = 2 * (169 * 16 * 14 * 9 + 121 * 32 * 16 * 9)
    out["features_tflops_conv"] = round(B * flop / (out["features"] * 1e-3) / 1e12, 1)
    out["features_act_write_gbs"] = round(B * p["kp"] * 2 / (out["features"] * 1e-3) / 1e9, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

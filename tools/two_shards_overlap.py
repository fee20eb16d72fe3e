"""One process, S handles of 65 536 / S envs each on S streams (every handle's own step -> render chain in order on its stream, the
handles independent): does another shard's k_step fill the render's gaps now that the render's stores pass by the caches?  Four
RANKS on one device read 224 M env-steps/s aggregate against 215 M for one handle (profiles/r05_four_ranks_one_gpu_bench.json).
Usage: python tools/two_shards_overlap.py [total_envs] [all|free|graph]"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("marl-ctf-development_amd")

E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
dev = torch.device("cuda:0")


def build(shards):
    vecs, tabs = [], []
    per = E // shards
    for s in range(shards):
        seeds = np.arange(s * per, (s + 1) * per, dtype=np.uint64) + 11
        v = pkg.VecGridworldCtf(per, device=0, py_seeds=seeds, np_seeds=seeds, log_metrics=True, **kw)
        _ = v.obs
        t = torch.empty((8, per, v.N_AGENTS), dtype=torch.int8, device=dev)
        for k in range(8):
            v.random_actions(t[k], seed=7, step=k, env_offset=s * per)
        vecs.append(v)
        tabs.append(t)
    return vecs, tabs


def run(shards, join_every_step, reps=200):
    vecs, tabs = build(shards)
    streams = [torch.cuda.Stream(device=dev) for _ in range(shards)]
    print(f"  placements: {[round(v.placement['render_over_fill'], 3) if v.placement else None for v in vecs]}", flush=True)

    def staggered(n):
        """exact call semantics: every call forks from the caller's stream and joins it again; inside, shard s + 1's step waits for shard
        s's step, so that it runs beside shard s's render"""
        cur = torch.cuda.current_stream(dev)
        for k in range(n):
            fork = torch.cuda.Event()
            fork.record(cur)
            prev = None
            ends = []
            for v, t, st in zip(vecs, tabs, streams):
                st.wait_event(fork)
                if prev is not None:
                    st.wait_event(prev)
                with torch.cuda.stream(st):
                    v.step(t[k % 8], auto_reset=True)
                    prev = torch.cuda.Event()
                    prev.record(st)
                    v.observe()
                    e = torch.cuda.Event()
                    e.record(st)
                    ends.append(e)
            for e in ends:
                cur.wait_event(e)

    graphs = []

    def build_graphs():
        """one hipGraph per action slot: shard s + 1's step behind shard s's step, each shard's render behind its own step — the fork
        and the join are edges of the graph, not events between streams; a replay on the caller's stream is one call with exact semantics"""
        cap = torch.cuda.Stream(device=dev)
        for k in range(8):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cap):
                prev = None
                for v, t, st in zip(vecs, tabs, streams):
                    st.wait_stream(cap)
                    if prev is not None:
                        st.wait_event(prev)
                    with torch.cuda.stream(st):
                        v.step(t[k], auto_reset=True)
                        prev = torch.cuda.Event()
                        prev.record(st)
                        v.observe()
                for st in streams:
                    cap.wait_stream(st)
            graphs.append(g)

    def steps(n):
        if join_every_step == "graph":
            if not graphs:
                build_graphs()
            for k in range(n):
                graphs[k % 8].replay()
            return
        if join_every_step == "staggered":
            return staggered(n)
        for k in range(n):
            for v, t, st in zip(vecs, tabs, streams):
                with torch.cuda.stream(st):
                    v.step_observe(t[k % 8], auto_reset=True)
            if join_every_step and shards > 1:  # every shard's step k done before any shard's step k + 1 starts
                evs = []
                for st in streams:
                    e = torch.cuda.Event()
                    e.record(st)
                    evs.append(e)
                for st in streams:
                    for e in evs:
                        st.wait_event(e)

    steps(60)
    torch.cuda.synchronize(dev)
    out = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        a.record()
        for st in streams:
            st.wait_event(a)
        steps(reps)
        for st in streams:
            e = torch.cuda.Event()
            e.record(st)
            torch.cuda.current_stream(dev).wait_event(e)
        b.record()
        torch.cuda.synchronize(dev)
        out.append(a.elapsed_time(b) / reps)
    ms = sorted(out)[2]
    print(f"{shards} shard(s) x {E // shards} envs, join every step: {join_every_step}: {ms:.4f} ms per step of all envs = {E / ms / 1e3:.1f} M env-steps/s", flush=True)
    for v in vecs:
        v.close()
    del vecs, tabs
    torch.cuda.empty_cache()


if __name__ == "__main__":
    plan = sys.argv[2] if len(sys.argv) > 2 else "all"
    if plan == "graph":
        run(1, False)
        run(2, "graph")
        run(4, "graph")
        run(1, "graph")
        run(2, False)
        sys.exit(0)
    run(1, False)
    if plan == "all":
        run(2, "staggered")
        run(4, "staggered")
        run(2, True)
    run(2, False)
    run(4, False)
    if plan == "all":
        run(1, "staggered")
    run(1, False)

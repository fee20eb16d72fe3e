"""Where does one env step of the drop-in facade go?  cProfile over bench.facade_1env's call pattern (ppo.py:59-98)."""
import cProfile, importlib, os, pstats, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
pkg = importlib.import_module("marl-ctf-development_amd")
kw = bench.WORKLOADS["arena"][1](pkg)
print(bench.facade_1env(pkg, kw, 0, budget_s=3.0))
random.seed(1); np.random.seed(1)
env = pkg.GridworldCtf(**kw)
rng = np.random.default_rng(0)
def run(k):
    for _ in range(k):
        for i in range(env.N_AGENTS):
            env.standardise_state(i, reverse_grid=env.AGENT_TEAMS[i] == 1)
            env.get_env_metadata(i)
        env.step([int(a) for a in rng.integers(0, 9, env.N_AGENTS)])
run(50)
pr = cProfile.Profile(); pr.enable(); run(2000); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
# the C call alone
vec = env._vec
a = np.zeros(env.N_AGENTS, np.int8)
t0 = time.perf_counter()
for _ in range(2000):
    vec.host_step(a, env._rng_in[0], env._rng_in[1], rng_out=True, obs=env._obs_host, meta=env._meta_host)
print("host_step alone: %.1f us" % ((time.perf_counter() - t0) / 2000 * 1e6))

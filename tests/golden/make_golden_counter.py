"""Golden vectors for the counter-based RNG mode (include/ctf_env.h CTF_RNG_COUNTER), recorded from the REFERENCE itself.

Runs in the BUILD container only (imports /root/reference through tests/golden/_refimport.py):

    python tests/golden/make_golden_counter.py

SURVEY 8(a): "a counter-based fast RNG as an opt-in mode whose parity is checked by running the oracle with random.shuffle /
np.random.rand / np.random.randint monkey-patched in the test harness to read the same tape".  Here the reference env runs with
exactly those three functions patched: each keeps its own published algorithm (CPython's Fisher-Yates over
_randbelow_with_getrandbits, NumPy's 53-bit random_sample, NumPy's masked-rejection bounded integer) and only the source of
its 32-bit words changes — word n of a stream is Philox4x32-10(key = the stream's 64-bit seed, counter = (n / 4, stream,
0x43544631))[n % 4], stream 0 for `random`, 1 for `np.random`.  Output: counter_*.npz in this directory — data only.
"""
import importlib
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _refimport  # noqa: E402
import make_golden as mg  # noqa: E402  (the same env kwargs as the MT19937 cases)

M32 = 0xFFFFFFFF


def philox4x32_10(key, ctr):
    """Salmon et al. 2011, Random123's philox4x32-10: key (2 words), counter (4 words) -> 4 words."""
    k0, k1 = key
    c0, c1, c2, c3 = ctr
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0, k1 = (k0 + 0x9E3779B9) & M32, (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


class Tape:
    def __init__(self, seed, stream):
        self.key, self.stream, self.n, self.block, self.block_id = (seed & M32, (seed >> 32) & M32), stream, 0, None, -1

    def word(self):
        b = self.n >> 2
        if b != self.block_id:
            self.block, self.block_id = philox4x32_10(self.key, (b & M32, (b >> 32) & M32, self.stream, 0x43544631)), b
        w = self.block[self.n & 3]
        self.n += 1
        return w


def patched(py_tape, np_tape):
    def randbelow(n):  # CPython Lib/random.py _randbelow_with_getrandbits
        k = n.bit_length()
        r = py_tape.word() >> (32 - k)
        while r >= n:
            r = py_tape.word() >> (32 - k)
        return r

    def shuffle(x):  # CPython Lib/random.py shuffle (3.10, random=None)
        for i in reversed(range(1, len(x))):
            j = randbelow(i + 1)
            x[i], x[j] = x[j], x[i]

    def rand():  # NumPy legacy random_sample
        a, b = np_tape.word() >> 5, np_tape.word() >> 6
        return (a * 67108864.0 + b) / 9007199254740992.0

    def randint(k):  # NumPy legacy randint(high) for a Python int: masked rejection on 32-bit words; randint(1) draws nothing
        rng = int(k) - 1
        if rng < 0:
            raise ValueError("high <= 0")
        if rng == 0:
            return 0
        mask = rng
        for s in (1, 2, 4, 8, 16):
            mask |= mask >> s
        while True:
            v = np_tape.word() & mask
            if v <= rng:
                return v

    return shuffle, rand, randint


def run_case(Ref, scn, name, scenario, kwargs, py_seed, np_seed, aseed, T, p_high=None):
    ref_scenario = getattr(scn, scenario) if isinstance(scenario, str) else scenario
    n = len(kwargs["AGENT_CONFIG"])
    arng = np.random.default_rng(aseed)
    actions = mg.biased_actions(arng, T, n, p_high) if p_high else arng.integers(0, 9, (T, n)).astype(np.int8)
    py_tape, np_tape = Tape(py_seed, 0), Tape(np_seed, 1)
    saved = random.shuffle, np.random.rand, np.random.randint
    random.shuffle, np.random.rand, np.random.randint = patched(py_tape, np_tape)
    try:
        env = Ref(SCENARIO=ref_scenario, **kwargs)
        rec = {k: [] for k in ("grid", "pos", "hp", "has_flag", "inv", "perm", "rewards", "done", "py_n", "np_n")}
        for t in range(T):
            if env.done:
                env.reset()
            _, rewards, done = env.step([int(a) for a in actions[t]])
            rec["grid"].append(env.grid.copy())
            rec["pos"].append([env.agent_positions[i] for i in range(n)])
            rec["hp"].append([float(env.agent_hp[i]) for i in range(n)])
            rec["has_flag"].append(env.has_flag.copy())
            rec["inv"].append([env.block_inventory[i] for i in range(n)])
            rec["perm"].append(list(env._arr))
            rec["rewards"].append([float(r) for r in rewards])
            rec["done"].append(int(done))
            rec["py_n"].append(py_tape.n)
            rec["np_n"].append(np_tape.n)
    finally:
        random.shuffle, np.random.rand, np.random.randint = saved
    meta = dict(name=name, scenario=mg.jsonable_scenario(scenario), kwargs=mg.jsonable_kwargs(kwargs), py_seed=py_seed, np_seed=np_seed,
                aseed=aseed, T=T, n=n, g=int(env.GRID_SIZE))
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), case_json=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), actions=actions,
        grid=np.array(rec["grid"], np.uint8), pos=np.array(rec["pos"], np.int8), hp=np.array(rec["hp"], np.float64),
        has_flag=np.array(rec["has_flag"], np.uint8), inv=np.array(rec["inv"], np.int32), perm=np.array(rec["perm"], np.uint8),
        rewards=np.array(rec["rewards"], np.float64), done=np.array(rec["done"], np.uint8),
        py_n=np.array(rec["py_n"], np.int64), np_n=np.array(rec["np_n"], np.int64))
    tags = sum(env.metrics["agent_tag_count"].values())
    print(f"{name:24s} N={n} T={T} words consumed: random {py_tape.n}, np.random {np_tape.n}; tags in the last episode {tags}; "
          f"reward sum {np.array(rec['rewards']).sum():.2f}")


def main():
    import warnings

    warnings.filterwarnings("ignore")
    Ref, scn = _refimport.import_reference()
    stress = dict(mg.ARENA_KW, GAME_STEPS=150, TAG_PROBABILITY=1.0, AGENT_TYPE_HP={0: 2, 1: 2, 2: 4, 3: 1.5}, VAULT_HP_COST=0.5, VAULT_MIN_HP=1.0)
    run_case(Ref, scn, "counter_arena", "arena_iii", mg.ARENA_KW, 0x0123456789ABCDEF, 0xFEDCBA9876543210, 77, 560)
    run_case(Ref, scn, "counter_split", "arrow", mg.SPLIT_KW, 42, 43, 78, 540)
    run_case(Ref, scn, "counter_arena_stress", "arena_iii", stress, 7, 2 ** 63 + 11, 79, 420, p_high=0.35)


if __name__ == "__main__":
    main()

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the tests check results, not speed: the timed search for a fast observation buffer (VecGridworldCtf._tune_obs_placement: up to
    # 3 s per 65 536-env batch, 10 s on a box that hands out slow allocations only) is cut short for the suite
    os.environ.setdefault("CTF_PLACEMENT_SECONDS", "0.2")
    # The CPU suite's torch work is tiny tensors (2 envs x 10 steps of the reference's PPO): with the default of one intra-op thread
    # per core, seven of eight threads only spin — the same wall time at 7 x the CPU time (51 s against 345 s of user time for the
    # reference-callers tests), and under CPU contention the spinning turned a 4-minute suite into one that did not finish in 40.
    # One thread here and in the processes the tests start; the GPU box keeps torch's default (device_count() does not touch the GPU).
    import torch

    if torch.cuda.device_count() == 0:
        os.environ.setdefault("OMP_NUM_THREADS", "1")
        torch.set_num_threads(1)


def pytest_sessionstart(session):
    """The suites need the in-tree libraries (HIP ABI + CPU oracle).  They normally arrive built; build them when missing."""
    hip = os.path.join(ROOT, "marl-ctf-development_amd", "csrc", "libctf_hip.so")
    orc = os.path.join(ROOT, "oracle", "libctf_oracle.so")
    if not (os.path.exists(hip) and os.path.exists(orc)):
        import __graft_entry__

        __graft_entry__.build()

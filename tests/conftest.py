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


def pytest_sessionstart(session):
    """The suites need the in-tree libraries (HIP ABI + CPU oracle).  They normally arrive built; build them when missing."""
    hip = os.path.join(ROOT, "marl-ctf-development_amd", "csrc", "libctf_hip.so")
    orc = os.path.join(ROOT, "oracle", "libctf_oracle.so")
    if not (os.path.exists(hip) and os.path.exists(orc)):
        import __graft_entry__

        __graft_entry__.build()

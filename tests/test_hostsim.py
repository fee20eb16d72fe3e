"""The step kernel's LOGIC on the CPU: marl-ctf-development_amd/csrc/ctf_step_core.h (env_step, the run-ahead MT19937 streams with
their hit bits / shuffle ring / end-of-step production, the counter-mode streams) is compiled for the host with one lane per
env and compared with the oracle after every step — state, float64 rewards, done and both generators' states in the standard
form — under AddressSanitizer + UBSan.  Two builds: the shipped window sizes, and every window shrunk to its minimum so that
the rare paths (direct loads beyond the hit-bit window, ring reloads, multi-batch production) run on every step.

This is a unit test of device code, not a CPU path of the product and not the oracle (tests/hostsim/hostsim.cpp says why).
Cross-lane behaviour (W > 1) is covered by the -m gpu tests only."""
import glob
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SIM = os.path.join(HERE, "hostsim")


def _asan_runtime():
    hits = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not hits:
        pytest.skip("no clang AddressSanitizer runtime in this image")
    return hits[0]


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-C", SIM, "-s"])
    return _asan_runtime()


@pytest.mark.parametrize("lib,mode", [("hostsim.so", "full"), ("hostsim_tiny.so", "quick")])
def test_step_logic_matches_oracle_on_the_host(built, lib, mode):
    env = dict(os.environ, LD_PRELOAD=built, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([sys.executable, os.path.join(SIM, "run_hostsim.py"), os.path.join(SIM, "_build", lib), mode],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + "\n" + out.stderr[-3000:]
    assert "all hostsim cases passed" in out.stdout

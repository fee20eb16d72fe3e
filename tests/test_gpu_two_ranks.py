"""The N-rank entry points with TWO REAL ranks on the one GPU a test box has: `bench.py --gpus 2` and `bench_rollout.py --gpus 2` start
their own ranks (the launcher of tests/test_bench_launch.py), both ranks use device 0 (CTF_BENCH_ONE_DEVICE=1) and talk over gloo — RCCL
refuses two ranks on one device, so the collectives' transport is the one thing this rehearsal does not share with an 8-GPU run; the
launcher, the rendezvous, rank -> global env range, seeds and action streams by global index, the barrier / max-over-ranks timing, the
self-check, the parameter broadcast and the data-parallel PPO update over the global minibatches (learner.PPOLearner(world=2)) all run as
they would there.  The reference's counterpart: one Ray task per env, one update over all rollouts (ppo.py:264-266,349-376)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *argv, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CTF_BENCH_DRYRUN")}
    env.update(CTF_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, script)] + list(argv), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=env, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_with_two_ranks_on_one_gpu():
    line = _run("bench.py", "--gpus", "2", "--envs-per-gpu", "8192", "--steps", "40", "--warmup", "5", "--no-cpu-baseline")
    assert line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1] and line["ranks_ok"] is True
    assert line["config"]["global_envs"] == 2 * 8192 and line["config"]["ranks_share_one_device"] is True
    assert line["device_status_bits"] == 0 and len(line["per_rank_ms_per_step"]) == 2 and min(line["per_rank_ms_per_step"]) > 0
    assert line["ms_per_step"] >= max(line["per_rank_ms_per_step"]) * 0.999   # the MAX over ranks (the barrier adds to it, never subtracts)
    assert line["value"] == pytest.approx(2 * 8192 * 40 / (line["ms_per_step"] * 40e-3), rel=1e-6)
    assert "cpu_baseline" not in line and set(line["secondary"]) == {"configs3_262144"}   # the N = 1 blocks are absent
    assert len(line["windows_ms_per_step"]) == 5 and sorted(line["windows_ms_per_step"])[2] == pytest.approx(line["ms_per_step"])
    # BASELINE configs[3] at its own size: 262 144 envs global = 131 072 per rank here, sharded by global env index
    c3 = line["secondary"]["configs3_262144"]
    assert c3["global_envs"] == 262144 and c3["envs_per_gpu"] == [131072, 131072] and c3["scaling"] == "strong"
    assert c3["device_status_bits"] == 0 and c3["value"] == pytest.approx(262144 / (c3["ms_per_step"] * 1e-3), rel=1e-6)
    assert c3["roofline"]["kernel"] in ("k_observe_tiles", "k_observe") and 0 < c3["roofline"]["frac"] < 1


def test_selfplay_iteration_with_two_ranks_on_one_gpu():
    res = _run("bench_rollout.py", "--gpus", "2", "--envs", "2048", "--steps", "8", "--micro-batch", "8192")
    assert res["n_gpus"] == 2 and res["global_envs"] == 4096 and res["ranks_share_one_device"] is True
    assert res["update_samples"] == 2 * 2048 * 8 * 4 and res["minibatch_order"] == "device"
    assert res["value"] > 0 and res["rollout_env_steps_per_s"] > 0 and res["update_sample_passes_per_s"] > 0
    assert all(abs(x) < 1e3 and x == x for x in res["losses_v_pg_entropy"])     # finite losses of the GLOBAL last minibatch
    assert "data-parallel" in res["note"]

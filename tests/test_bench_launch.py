"""`bench.py --gpus N` / `bench_rollout.py --gpus N` must themselves start N ranks (the reference's N-worker launch is one Ray task
per env, ppo.py:264-266,349-376).  Driven here without a GPU: CTF_BENCH_DRYRUN=1 runs everything of the bench but the kernels — the
launcher, torchrun's rendezvous on 127.0.0.1, the process group (gloo), the barrier, the max-over-ranks rule, the self-check and the
JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *argv, env=None, timeout=600):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(CTF_BENCH_DRYRUN="1", **(env or {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, script)] + list(argv), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, env=e, timeout=timeout, cwd=ROOT)


def _one_json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_2_without_torchrun_starts_two_ranks():
    r = _run("bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1] and line["ranks_ok"] is True
    # five timed windows, per window the MAX over ranks (rank r 'takes' 1 + r ms per step, window w 1 + |w - 2| % more); the line
    # reports the MEDIAN window, not the best one and not the mean
    assert line["windows"] == 5 and line["windows_ms_per_step"] == pytest.approx([2.04, 2.02, 2.0, 2.02, 2.04])
    assert line["per_rank_ms_per_step"] == pytest.approx([1.01, 2.02]) and line["ms_per_step"] == pytest.approx(2.02)
    assert line["config"]["global_envs"] == 2 * line["config"]["envs_per_gpu"]
    assert line["value"] == pytest.approx(2 * 65536 / 2.02e-3)
    assert "DRY RUN" in line["data"]  # a dry-run line can never pass for a measurement
    assert line["kernels_ms"]["k_step"] + line["kernels_ms"]["none (dry run)"] <= line["ms_per_step"]


def test_gpus_2_line_carries_configs3_at_its_own_global_size():
    """BASELINE.json configs[3] is 262 144 envs GLOBAL: besides the weak-scaling headline (65 536 per GPU) an N-rank line holds
    `secondary.configs3_262144`, 262 144 / N envs per rank, sharded by global env index (sharding.shard_range)."""
    r = _run("bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    c3 = _one_json_line(r.stdout)["secondary"]["configs3_262144"]
    assert c3["global_envs"] == 262144 and c3["envs_per_gpu"] == [131072, 131072] and c3["shards"] == [[0, 131072], [131072, 262144]]
    assert c3["scaling"] == "strong" and c3["n_gpus"] == 2 and "configs[3]" in c3["workload"] and "262144 envs GLOBAL" in c3["workload"]
    assert c3["value"] == pytest.approx(262144 / (c3["ms_per_step"] * 1e-3)) and c3["ms_per_step"] == pytest.approx(2.02)
    assert c3["roofline"]["algorithmic_bytes_per_env"] == 25878
    # an odd global size is split by the same rule the learner and the tests use
    r = _run("bench.py", "--gpus", "2", "--steps", "2", "--warmup", "0", "--configs3-envs", "1001")
    c3 = _one_json_line(r.stdout)["secondary"]["configs3_1001"]
    assert c3["envs_per_gpu"] == [501, 500] and c3["global_envs"] == 1001
    # N = 1 is not configs[3]: no such block (its shard rides as secondary.arena_32768 on a GPU)
    r = _run("bench.py", "--gpus", "1", "--steps", "2", "--warmup", "0")
    assert "secondary" not in _one_json_line(r.stdout)


def test_gpus_1_is_one_rank_in_process():
    r = _run("bench.py", "--gpus", "1", "--steps", "3", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    assert line["n_gpus"] == 1 and line["ranks_seen"] == [0]
    assert "starting" not in r.stderr  # no launcher


def test_world_size_that_contradicts_gpus_is_refused():
    r = _run("bench.py", "--gpus", "2", env=dict(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr and r.stdout.strip() == ""
    r = _run("bench.py", "--gpus", "8", env=dict(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and r.stdout.strip() == ""  # the driver's N=1 form with a wrong --gpus: no mislabelled line


def test_more_gpus_than_the_box_has_is_refused():
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CTF_BENCH_DRYRUN")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, env=e, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "visible" in r.stderr and r.stdout.strip() == ""


def test_a_failing_rank_fails_the_job():
    r = _run("bench.py", "--gpus", "2", "--steps", "2", env=dict(CTF_BENCH_DRYRUN_FAIL_RANK="1"))
    assert r.returncode != 0
    assert r.stdout.strip() == ""  # rank 0 never got past the first collective: no line


def test_the_launcher_never_imports_torch():
    """The parent that starts the ranks must not have touched HIP: its code path imports nothing but the standard library."""
    code = ("import sys, os; sys.argv=['bench.py','--gpus','2','--steps','1']; os.environ['CTF_BENCH_DRYRUN']='1';\n"
            "import subprocess; subprocess.call=lambda cmd, env=None: (print('CMD', ' '.join(cmd)), 0)[1]\n"
            "import runpy\n"
            "try:\n    runpy.run_path('bench.py', run_name='__main__')\nexcept SystemExit as e:\n    print('RC', e.code)\n"
            "print('TORCH', 'torch' in sys.modules, 'PKG', any('marl' in m for m in sys.modules))")
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e, cwd=ROOT, timeout=120)
    assert "RC 0" in r.stdout and "TORCH False PKG False" in r.stdout, r.stdout + r.stderr
    cmd = [ln for ln in r.stdout.splitlines() if ln.startswith("CMD")][0]
    assert "torch.distributed.run" in cmd and "--nproc-per-node=2" in cmd and "--master-addr 127.0.0.1" in cmd and cmd.endswith("--gpus 2 --steps 1")


def test_bench_rollout_gpus_2_runs_the_two_rank_iteration():
    """configs[4] for N ranks: the launcher, a parameter broadcast, and the data-parallel update over gloo — every rank ends with
    the same parameters."""
    r = _run("bench_rollout.py", "--gpus", "2", "--steps", "3")
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1] and line["parameters_identical_on_all_ranks"] is True
    assert "DRY RUN" in line["metric"]
    r = _run("bench_rollout.py", "--gpus", "2", env=dict(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and r.stdout.strip() == ""


def test_cpu_baseline_block_reports_one_core_the_box_share_and_all_usable_cores():
    """the C oracle timed on one core and on all host cores the process may use (affinity mask capped by the cgroup's CPU
    quota), core counts and CPU model stated; the per-env Python / NumPy restatement on one core beside it."""
    import importlib

    sys.path.insert(0, ROOT)
    import bench

    pkg = importlib.import_module("marl-ctf-development_amd")
    kw = bench.WORKLOADS["arena"][1](pkg)
    out = bench.cpu_baseline(pkg, kw, budget_s=0.6)
    assert out["kind"] == "port" and out["unit"] == "env-steps/s" and out["value"] > 0 and out["single_core_value"] > 0
    usable = min(len(os.sched_getaffinity(0)), bench._cpu_quota() or 1 << 30)
    assert out["cores"] == min(len(os.sched_getaffinity(0)), 16) and out["all_cores"] == usable
    assert out["all_cores_value"] and out["all_cores_value"] > 0 and out["all_cores_sample"]
    assert out["host_cpu"]["logical_cpus"] == os.cpu_count() and out["host_cpu"]["model"]
    assert out["python_numpy_1core"]["cores"] == 1 and out["python_numpy_1core"]["value"] > 0
    assert bench._cpu_quota() is None or bench._cpu_quota() >= 1

"""bench.py's launch path: `--gpus N` decides the number of GPUs.  Without a launcher it starts N rank processes
itself; under torch.distributed.run the two must agree; a mismatch exits non-zero instead of running one GPU and
reporting N.  The CPU legs rehearse the control plane only (gloo rendezvous, no solver); the GPU leg runs the real
bench with two ranks on the one GPU of the test box through the test-only librccl stand-in."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def _json_line(stdout: str) -> dict:
    return json.loads([ln for ln in stdout.strip().splitlines() if ln.startswith("{")][-1])


def test_gpus_flag_spawns_that_many_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--control-plane-only"], capture_output=True, text=True,
                       env=_env(), timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_ranks"] == 2 and len(set(out["rank_pids"])) == 2


def test_launcher_world_size_must_match_gpus_flag():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--control-plane-only"], capture_output=True, text=True,
                       env=_env(WORLD_SIZE="4", RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533"), timeout=60)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--control-plane-only"], capture_output=True, text=True,
                       env=_env(WORLD_SIZE="2", RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533"), timeout=60)
    assert r.returncode != 0


def test_under_torch_distributed_run():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", BENCH, "--gpus", "2", "--control-plane-only"],
                       capture_output=True, text=True, env=_env(), timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_ranks"] == 2


@pytest.mark.gpu
def test_bench_gpus_2_runs_two_ranks_and_reports_what_the_communicator_says():
    """The whole bench with `--gpus 2` and no launcher: two rank processes (distinct pids, neither this one's child
    shell), landmark-sharded, all-reducing the reduced camera system through the stand-in for librccl (RCCL refuses two
    ranks on one device); `n_gpus` is the communicator's own count."""
    env = _env(BA_BENCH_DEVICE="0", BA_RCCL_LIBRARY=os.path.join(ROOT, "tests", "fake_rccl", "librccl.so"))
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--workload", "small", "--steps", "2", "--warmup", "1",
                        "--large", "none", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and len(set(out["rank_pids"])) == 2 and os.getpid() not in out["rank_pids"]
    assert out["status"] in ("gtol", "ftol", "xtol") and out["value"] > 0
    assert "x2" in out["config"]["parallelism"]

"""The landmark-sharded solve with MORE than one rank, on the one-GPU test box.

RCCL refuses two ranks on one device, so the rank processes are started with tests/fake_rccl (a test-only stand-in
for librccl.so that does each collective through host shared memory) named in BA_RCCL_LIBRARY.  What runs is the
library's real multi-rank code: the union of the ranks' block patterns, the all-reduce of [S | rhs | U, gc] in every
LM iteration, the folded statistics, the device-side LM control on every rank.  The sharded estimate must agree with
the single-GPU solve of the same scene."""

import os
import subprocess
import sys

import numpy as np
import pytest

import scenes
from src import _native

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM_DIR = os.path.join(HERE, "fake_rccl")


def run_ranks(world, scene_name, tmp_path, **options):
    if not os.path.exists(os.path.join(SHIM_DIR, "librccl.so")):
        pytest.skip("tests/fake_rccl/librccl.so is not built (test infrastructure; __graft_entry__.build() makes it)")
    env = dict(os.environ)
    env["BA_RCCL_LIBRARY"] = os.path.join(SHIM_DIR, "librccl.so")
    args = [f"{k}={v}" for k, v in options.items()]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_two_rank_worker.py"), str(r), str(world),
                               str(tmp_path), scene_name] + args, env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=100)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-2000:]}"
    return [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]


def single_gpu(sc, **options):
    with _native.Solver(**options) as s:
        s.set_problem(len(sc["poses_guess"]), len(sc["points_guess"]), sc["obs_cam"], sc["obs_pt"], sc["obs_uv"],
                      sc["intr"])
        return s.solve(sc["poses_guess"], sc["points_guess"])


def check_against_single(parts, sc, P0, X0, rep0, atol):
    for part in parts:
        # replicated quantities are identical on every rank
        assert float(part["cost"]) == float(parts[0]["cost"])
        assert int(part["iterations"]) == int(parts[0]["iterations"])
        np.testing.assert_array_equal(part["poses"], parts[0]["poses"])
        assert int(part["nobs"]) == len(sc["obs_cam"])
    assert abs(float(parts[0]["cost0"]) - rep0["cost0"]) <= 1e-12 * rep0["cost0"]
    assert abs(float(parts[0]["cost"]) - rep0["cost"]) <= 1e-8 * rep0["cost"]
    X = np.concatenate([part["points"] for part in parts])
    assert [int(p["lo"]) for p in parts[1:]] == [int(p["hi"]) for p in parts[:-1]]
    np.testing.assert_allclose(parts[0]["poses"], P0, rtol=0, atol=atol)
    np.testing.assert_allclose(X, X0, rtol=0, atol=atol)


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_ring_matches_single_gpu(world, tmp_path):
    """The bench workload's shape (ring, 50 cameras, one-workgroup PCG) at 1/10 of its landmarks."""
    sc = scenes.ring_scene(50, 2000, 10)
    P0, X0, rep0 = single_gpu(sc, ftol=1e-12)
    parts = run_ranks(world, "ring", tmp_path, ftol=1e-12)
    check_against_single(parts, sc, P0, X0, rep0, atol=1e-6)


def test_sharded_unstructured_scene_takes_the_union_of_block_patterns(tmp_path):
    """Random visibility: the ranks see different camera pairs, so the pattern union matters."""
    sc = scenes.noisy_scene(8, 400)
    P0, X0, rep0 = single_gpu(sc, ftol=1e-12)
    parts = run_ranks(2, "noisy", tmp_path, ftol=1e-12)
    check_against_single(parts, sc, P0, X0, rep0, atol=1e-6)


def test_sharded_direct_solver(tmp_path):
    """Two ranks with a direct solver (300 cameras in a ring: cyclic reduction): the factorisation runs replicated on the
    all-reduced system."""
    sc = scenes.ring_scene(300, 3000, 6)
    P0, X0, rep0 = single_gpu(sc, ftol=1e-12)
    parts = run_ranks(2, "ring_long", tmp_path, ftol=1e-12)
    assert str(parts[0]["solver"]) == "cyclic_reduction"
    check_against_single(parts, sc, P0, X0, rep0, atol=1e-6)


def test_sharded_multi_kernel_pcg(tmp_path, monkeypatch):
    """Two ranks on the multi-kernel PCG path (camera systems too large for one workgroup): the ranks follow rank
    0's convergence scalars and camera step (the two broadcasts of the path)."""
    monkeypatch.setenv("BA_SMALL_SOLVE", "0")
    sc = scenes.ring_scene(50, 2000, 10)
    P0, X0, rep0 = single_gpu(sc, ftol=1e-12, pcg_tol=1e-6, linear_solver=1)
    parts = run_ranks(2, "ring", tmp_path, ftol=1e-12, pcg_tol=1e-6, linear_solver=1)
    assert str(parts[0]["solver"]) == "pcg"
    check_against_single(parts, sc, P0, X0, rep0, atol=1e-6)


def test_sharded_robust_loss(tmp_path):
    """Two ranks with a Huber loss: the robust cost and the IRLS weights are per observation, hence per shard."""
    sc = scenes.ring_scene(50, 2000, 10)
    P0, X0, rep0 = single_gpu(sc, ftol=1e-12, loss=2, f_scale=1.0)
    parts = run_ranks(2, "ring", tmp_path, ftol=1e-12, loss=2, f_scale=1.0)
    check_against_single(parts, sc, P0, X0, rep0, atol=1e-6)


@pytest.mark.parametrize("fault,stage", [("empty", "none"), ("nan", "set_problem")])
def test_invalid_input_on_one_rank_stops_every_rank(fault, stage, tmp_path):
    """One rank hands the library a shard without observations / a guess with a NaN: every rank must get the
    ValidationError (the ranks agree on a status before their data collectives) instead of rank 0 waiting forever
    inside the all-reduce; afterwards the same handles go through a good set_problem + solve together."""
    sc = scenes.ring_scene(50, 2000, 10)
    _, _, rep0 = single_gpu(sc)
    parts = run_ranks(2, "ring:" + fault, tmp_path)
    for r, part in enumerate(parts):
        assert str(part["stage"]) == stage, (r, str(part["stage"]), str(part["err"]))
        assert str(part["err"]) != ""
        assert abs(float(part["cost"]) - rep0["cost"]) <= 1e-8 * rep0["cost"]
    assert "another rank" in str(parts[0]["err"])


def test_sharded_solve_with_several_iterations_per_host_poll(tmp_path):
    """lm_check_every = 3 with two ranks: three LM iterations (each with its two all-reduces) are enqueued per host poll;
    the iterations enqueued after the device-side termination exit at once on every rank, collectives included, so the
    ranks stay in step and the result is the single-GPU one."""
    sc = scenes.ring_scene(300, 3000, 6)
    P0, X0, rep0 = single_gpu(sc, ftol=1e-12)
    parts = run_ranks(2, "ring_long", tmp_path, ftol=1e-12, lm_check_every=3)
    check_against_single(parts, sc, P0, X0, rep0, atol=1e-6)


@pytest.mark.parametrize("world", [2, 4])
def test_reduced_solve_sharded_by_camera_segment(world, tmp_path, monkeypatch):
    """BA_SHARDED_SOLVE=1 (opt-in): the reduced camera system of a trajectory is not solved on every
    rank.  S is scattered per rank as a partial sum, the camera segments of the two-level banded solver travel to their
    owners by one reduce per rank, every rank eliminates the interiors of its own segments, the separator system is
    all-reduced and solved on every rank, the ranks back-substitute their own segments, and the camera step is summed
    (one writer per entry).  Against the single-GPU solve (cyclic reduction there): cost 1e-8, estimates 1e-6; the
    replicated poses bit-identical on all ranks."""
    monkeypatch.setenv("BA_SHARDED_SOLVE", "1")
    sc = scenes.ring_scene(300, 3000, 6)
    P0, X0, rep0 = single_gpu(sc, ftol=1e-12)
    parts = run_ranks(world, "ring_long", tmp_path, ftol=1e-12)
    for part in parts:
        assert str(part["solver"]) == "banded_cholesky" and int(part["sharded_solve"]) == 1, (str(part["solver"]), int(part["sharded_solve"]))
        assert int(part["segments"]) % world == 0 and int(part["bad_pivots"]) == 0
    check_against_single(parts, sc, P0, X0, rep0, atol=1e-6)

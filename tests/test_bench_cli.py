"""bench.py's launcher plumbing on a box without GPUs: `--gpus N` starts the N ranks itself (fresh child
before torch / HIP is touched), forwards one JSON line and reports the world size the process group has."""
import json
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, **env):
    e = dict(os.environ, OMP_NUM_THREADS="1", **env)
    e.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=e, timeout=600)


def test_gpus_2_self_spawns_two_ranks_and_reports_them():
    p = _run(["--gpus", "2", "--rendezvous-only"], HEATFLOW_BENCH_BACKEND="gloo")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1                                   # exactly one JSON line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["max_rank_plus_1"] == 2 and out["backend"] == "gloo"


def test_gpus_1_needs_no_launcher():
    p = _run(["--gpus", "1", "--rendezvous-only"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip())["n_gpus"] == 1


def test_more_ranks_than_gpus_fails_loudly_with_rccl():
    """RCCL needs one GPU per rank: without them the run must exit non-zero, not report n_gpus: 1."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs are present")
    p = _run(["--gpus", "2", "--rendezvous-only"])
    assert p.returncode != 0 and not p.stdout.strip()
    assert "GPU(s) visible" in p.stderr


def test_bad_arguments_are_rejected():
    p = _run(["--gpus", "0"])
    assert p.returncode != 0


def test_cpu_farm_worker_reports_a_pool_of_single_threaded_processes():
    """C5's CPU baseline: the farm body (a GPU-free child of bench.py) runs sweep points through the oracle on a process
    pool and prints one JSON object with the fields of `cpu_baseline`."""
    p = _run(["--cpu-farm-worker", "--cpu-farm-points", "2", "--steps", "3"])
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["kind"] == "port" and out["cores"] == 2 and out["points"] == 2 and out["unit"] == "DOF-updates/s"
    assert out["value"] > 0 and out["wall_s"] >= out["per_point_s_mean"] * 0.5


def test_side_guard_writes_the_headline_and_ends_the_process_when_a_side_measurement_hangs(tmp_path):
    """N > 1: a collective of the C5 side measurement that never completes must not take the measured headline with it.  The
    guard (bench.SideGuard) writes the line with the failure noted in the side measurement's place and exits 0; disarmed in
    time it does nothing."""
    script = tmp_path / "guard.py"
    script.write_text(
        "import json, os, sys, time\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "out = {'metric': 'm', 'value': 1.0, 'config': {}}\n"
        "emit = lambda o: os.write(1, (json.dumps(o) + '\\n').encode())\n"
        "mode = sys.argv[1]\n"
        "g = bench.SideGuard(out, emit, 0.2 if mode == 'hang' else 30.0)\n"
        "if mode == 'hang':\n"
        "    time.sleep(20)\n"                      # the stuck collective
        "    sys.exit(3)\n"                         # never reached
        "assert g.disarm() and not g.disarm()\n"
        "out['config']['sweep64'] = {'value': 2.0}\n"
        "emit(out)\n")
    hang = subprocess.run([sys.executable, str(script), "hang"], capture_output=True, text=True, timeout=60)
    assert hang.returncode == 0, hang.stderr[-1000:]
    lines = [ln for ln in hang.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["value"] == 1.0 and "not finished" in line["config"]["sweep64"]["error"] and line["cpu_baseline"] is None
    ok = subprocess.run([sys.executable, str(script), "ok"], capture_output=True, text=True, timeout=60)
    assert ok.returncode == 0, ok.stderr[-1000:]
    assert json.loads(ok.stdout.strip())["config"]["sweep64"] == {"value": 2.0}

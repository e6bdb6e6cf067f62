"""CPU: `python bench.py --gpus N` without a launcher starts its own ranks (bench.py launch_ranks), relays exactly one
JSON line on stdout and the job's return code.  The ranks run the --selftest-launch body (gloo, no GPU, no model)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], capture_output=True, text=True, env=env,
                          timeout=300)


def test_self_launch_two_ranks():
    r = _run("--gpus", "2", "--selftest-launch", "ok")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2
    assert "noise on stdout from rank" in r.stderr and "noise" not in r.stdout       # everything else goes to stderr


def test_self_launch_propagates_failure():
    r = _run("--gpus", "2", "--selftest-launch", "fail")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_one_rank_needs_no_launcher():
    r = _run("--gpus", "1", "--selftest-launch", "ok")
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_ranks_seen"] == 1

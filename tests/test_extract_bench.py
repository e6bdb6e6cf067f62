"""CPU: tools/extract_bench.py (BASELINE configs[2] as one command) -- the launcher, the per-rank shards, the per-step exchange and
rank 0's Arrow file with a stand-in model over gloo; one rank and two."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "extract_bench.py"), "--selftest", *extra], capture_output=True,
                          text=True, env=env, timeout=300)


@pytest.mark.parametrize("gpus", [1, 2])
def test_extract_bench_selftest(gpus):
    r = _run("--gpus", str(gpus), "--images", "23", "--batch", "4")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["n_ranks_seen"] == gpus and d["rows_written"] == 23 and d["images"] == 23
    assert d["value"] > 0

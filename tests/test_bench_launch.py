"""``python bench.py --gpus N`` with no launcher around it starts its own ranks: the parent spawns
``torch.distributed.run`` children before touching the GPU, rank 0's line is the parent's stdout, the children's exit code is the
parent's.  ``--launch-check`` exercises exactly that plumbing without the product path (gloo on the CPU here)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=300)


def test_bare_python_bench_gpus_2_launches_two_ranks():
    r = _run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 only
    out = json.loads(lines[0])
    assert out["launch_check"] is True and out["ranks_seen"] == 2 and out["n_gpus"] == 2


def test_launch_check_single_process():
    r = _run(["--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["ranks_seen"] == 1

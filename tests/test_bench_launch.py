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


def test_bench_timeline_and_distribution_helpers():
    """bench.py's helpers that turn the event recorder's arrays into the line's ``step_ms`` / ``timeline_ms`` / ``timed_gemm_us``:
    synthetic records of four steps (marks 1-6 on two streams, GEMM launches with flops > 0)."""
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    d = bench._dist([24.0, 23.0, 25.0, 60.0, 23.5])
    assert d["min"] == 23.0 and d["max"] == 60.0 and d["first"] == 24.0 and d["median"] == 24.0
    b, e, f = [], [], []
    for k in range(4):                      # a step every 24 ms: forward 0..6, backward ..19, optimizer ..19.2, encoder pass 0.3..23.8
        t0 = 24.0 * k
        for tag, t in ((1, 0.0), (5, 0.3), (2, 6.0), (3, 19.0), (4, 19.2), (6, 23.8)):
            b.append(t0 + t); e.append(t0 + t); f.append(-float(tag))
        for j in range(48):                 # 48 GEMMs: 0.44 ms while the trunk is there, 0.30 ms after it
            st = t0 + 0.4 + j * 0.48
            b.append(st); e.append(st + (0.44 if st < t0 + 19.2 else 0.30)); f.append(1e11)
    b, e, f = np.array(b), np.array(e), np.array(f)
    t = bench._timeline(b, e, f)
    assert t["steps"] == 2 and abs(t["step"] - 24.0) < 1e-6
    assert abs(t["trunk_forward_end"] - 6.0) < 1e-6 and abs(t["trunk_backward_end"] - 19.0) < 1e-6 and abs(t["optimizer_end"] - 19.2) < 1e-6
    assert abs(t["encoder_pass_start"] - 0.3) < 1e-6 and abs(t["encoder_pass_end"] - 23.8) < 1e-6
    g = bench._gemm_split(b, e, f)
    assert abs(g["beside_trunk"] - 440.0) < 0.5 and abs(g["after_trunk"] - 300.0) < 0.5 and g["launches_after_trunk"] == 8
    # the schedule since round 5: passes back to back on the encoder stream - the pass launched in a step starts 2.4 ms into it (where the
    # one before ended) and ends 2.4 ms into the NEXT step; no GEMM starts after the trunk's optimizer step any more
    b, e, f = [], [], []
    for k in range(5):
        t0 = 22.0 * k
        for tag, t in ((1, 0.0), (6, 2.4), (5, 2.4), (2, 8.0), (3, 21.5), (4, 21.7)):
            b.append(t0 + t); e.append(t0 + t); f.append(-float(tag))
        for j in range(48):
            st = t0 + 2.5 + j * 0.45
            b.append(st); e.append(st + 0.42); f.append(1e11)
    b, e, f = np.array(b), np.array(e), np.array(f)
    t = bench._timeline(b, e, f)
    assert t["steps"] == 3 and abs(t["step"] - 22.0) < 1e-6 and abs(t["optimizer_end"] - 21.7) < 1e-6
    assert abs(t["encoder_pass_start"] - 2.4) < 1e-6 and abs(t["encoder_pass_end"] - 24.4) < 1e-6      # ends inside the next step's window
    g = bench._gemm_split(b, e, f)
    assert abs(g["beside_trunk"] - 420.0) < 0.5 and g["launches_after_trunk"] in (0, 1)


import pytest


@pytest.mark.gpu
def test_two_rank_bench_rehearsal_on_one_gpu():
    """The N-rank control flow of bench.py on a one-GPU box (RUART_BENCH_REHEARSE_ONE_GPU=1: both ranks on device 0, gradients over
    gloo): the self-launch, the timed region's barriers, rank 0's parity check on a second trainer - built as a plain replica, no
    collective in its set-up - while rank 1 waits, one JSON line from rank 0.  The numbers mean nothing; the run must end."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["RUART_BENCH_REHEARSE_ONE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                        "--no-bert512"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["config"]["global_batch"] == 128
    assert out["parity"]["holds"] and out["parity"]["max_abs_err_vs_reference"] < 1e-3

#!/usr/bin/env python3
"""Two training sessions in one process with SDNetTrainer.close() between them (what two train() calls, or train() -> predict_for_test()
-> train(), do): close() destroys the CU-masked encoder stream, the second session creates a new one - does it still overlap with the
trunk's streams?  (A stream's hardware queue slot follows its creation order: profiles/HISTORY.md round 5 (9).)  Median step time per session."""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
A = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]


def session(name, n=40):
    ts = []
    with tr.step_stream():
        for i in range(n):
            t0 = time.perf_counter()
            float(tr.update(A[i % 2], i, next_batch=A[(i + 1) % 2]))
            ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    print("%-40s median %.2f ms (p90 %.2f)" % (name, statistics.median(ts[5:]), sorted(ts[5:])[int(0.9 * (n - 5))]), flush=True)


if os.environ.get("EVAL_FIRST"):            # an evaluation before the first training step: its streams are created first
    tr.predict(A[0], next_batch=A[1])
    tr.predict(A[1])
    tr.close()
session("session 1")
for k in range(int(os.environ.get("SESSIONS", 3))):
    tr.close()
    if os.environ.get("EVAL_BETWEEN"):
        tr.predict(A[0])
        tr.close()
    session("session %d (after close())" % (k + 2))
tr.close(final=True)

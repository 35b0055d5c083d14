#!/usr/bin/env python3
"""Soak: a few hundred pipelined training steps over changing batches (fresh synthetic batch every step, ragged sizes), checks
that the loss stays finite, decreases on average, and that device memory does not grow."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200
unlock, stage = "--unlock" in sys.argv, "--stage" in sys.argv      # trained encoder (deferred readback) / batches staged inside update()
dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
if unlock:
    opt.pop("LOCK_BERT")
    opt["bert_train_gemm"] = "16"
    opt["lr"] = 2e-4
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
rng = np.random.default_rng(0)
def make(i):
    b = synth.synthetic_batch(opt, 64, seed=1000 + i % 8, n_q=int(rng.integers(20, 31)), n_ocr=int(rng.integers(60, 101)), n_od=int(rng.integers(10, 37)), ragged=True)
    return tr.ToCUDA(b)
nxt = make(0)
losses, mem = [], []
t0 = time.time()
cur = nxt
nxt = make(1)
for i in range(steps):
    if stage:
        losses.append(tr.update(cur, i, next_batch=nxt, stage_next=lambda: make(i + 2)))
        cur, nxt = nxt, tr.staged
    else:
        losses.append(tr.update(cur, i, next_batch=nxt))
        cur, nxt = nxt, make(i + 2)
    if i % 20 == 0:
        torch.cuda.synchronize()
        mem.append(torch.cuda.memory_allocated() / 2**30)
        print("step %4d loss %.4f avg %.4f  mem %.2f GiB reserved %.2f GiB  %.1f s" % (i, float(losses[-1]), np.mean([float(v) for v in losses[-20:]]), mem[-1], torch.cuda.memory_reserved() / 2**30, time.time() - t0), flush=True)
tr.close(final=True)          # the CU-masked run-ahead stream must not outlive the process teardown
losses = [float(v) for v in losses]
assert all(np.isfinite(losses))
assert np.mean(losses[-20:]) < np.mean(losses[:20]), (np.mean(losses[:20]), np.mean(losses[-20:]))
assert mem[-1] < mem[1] * 1.15 + 0.5, mem
print("soak ok: %d steps, loss %.3f -> %.3f, memory %.2f -> %.2f GiB" % (steps, np.mean(losses[:20]), np.mean(losses[-20:]), mem[1], mem[-1]))

#!/usr/bin/env python3
"""SDNetTrainer.predict in a loop with one batch of lookahead (what evaluate() does): ms per batch, and the share of the answer decode."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
batches = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
for i in range(4):
    tr.predict(batches[i % 2], next_batch=batches[(i + 1) % 2])
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for i in range(N):
    tr.predict(batches[i % 2], next_batch=batches[(i + 1) % 2])
torch.cuda.synchronize()
print("predict(): %.2f ms per batch of 64 (%.0f samples/s)" % ((time.perf_counter() - t0) / N * 1e3, 64 * N / (time.perf_counter() - t0)))
if "--trace" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(N):
        tr.predict(batches[i % 2], next_batch=batches[(i + 1) % 2])
    torch.cuda.synchronize()
    pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(12)
tr.close(final=True)

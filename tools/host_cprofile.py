#!/usr/bin/env python3
"""Python-level host cost of a training step (cProfile, sorted by own time): where the ~15 ms of enqueue time go."""
import cProfile, os, pstats, sys, time
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
    tr.update(batches[i % 2], i)
torch.cuda.synchronize()
N = 8
t0 = time.perf_counter()
for i in range(N):
    tr.update(batches[i % 2], i, next_batch=batches[(i + 1) % 2])
torch.cuda.synchronize()
print("plain: %.2f ms per step" % ((time.perf_counter() - t0) * 1e3 / N))
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    tr.update(batches[i % 2], i, next_batch=batches[(i + 1) % 2])
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
tr.close()

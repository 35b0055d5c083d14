#!/usr/bin/env python3
"""Trained-encoder step: host time of every update() call (readback deferred) against the step time - where does the host block?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
opt.pop("LOCK_BERT")
opt["bert_train_gemm"] = "16"
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
batches = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
for i in range(3):
    tr.update(batches[i % 2], i)
torch.cuda.synchronize()
if "--trace" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
hs = []
for i in range(8):
    a = time.perf_counter()
    tr.update(batches[i % 2], i)
    hs.append((time.perf_counter() - a) * 1e3)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
if "--trace" in sys.argv:
    pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(25)
print("host time per update():", " ".join("%.1f" % h for h in hs))
print("8 steps: host done after %.1f ms, device after %.1f ms (%.2f ms per step)" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 1e3 / 8))
tr.close(final=True)

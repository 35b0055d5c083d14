#!/usr/bin/env python3
"""Per-kernel times of one frozen-encoder pass of the bench batch, alone on the device (ruart_prof-free: HIP events around whole
passes are too coarse, so this prints the pass time; use under rocprofv3 --kernel-trace --stats for the per-kernel table)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt
from ruart_amd.bert import bert_encode, _Buffers

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
b = tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7, n_q=30, n_ocr=100, n_od=36))
packed = b[0]["_ruart_index"].packed
W = tr.network.Bert.weights
bf = _Buffers()
for _ in range(3):
    bert_encode(W, packed, bf)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 10
for _ in range(N):
    bert_encode(W, packed, bf)
torch.cuda.synchronize()
print("encoder pass alone (T=%d rows=%d): %.3f ms" % (packed.T, packed.Tp, (time.perf_counter() - t0) / N * 1e3))
tr.close(final=True)

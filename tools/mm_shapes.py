#!/usr/bin/env python3
"""Shapes and device time of the library GEMMs (aten::mm / addmm / bmm) in one training step."""
import os, sys
from collections import defaultdict
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
batches = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
for i in range(4):
    tr.update(batches[i % 2], i)
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for i in range(N):
        tr.update(batches[i % 2], i)
    torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0])
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::mm", "aten::addmm", "aten::bmm"):
        k = (e.key, str(e.input_shapes))
        agg[k][0] += e.count
        agg[k][1] += e.device_time_total
tot = 0
for (name, shp), (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += t
    print("%8.1f us/step %5.1f calls/step  avg %7.1f us  %s %s" % (t / N, cnt / N, t / max(cnt, 1), name, shp))
print("total %.2f ms/step" % (tot / N / 1e3))

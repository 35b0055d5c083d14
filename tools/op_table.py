#!/usr/bin/env python3
"""Launch inventory of one training step: aten / autograd-function calls per step by name (torch.profiler, CPU side), i.e. what the
~1 000 launches of the SDNet trunk are made of.   python tools/op_table.py [--precision fp16c]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ruart_amd import synth  # noqa: E402
from ruart_amd.arguments import default_opt  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="fp16c")
ap.add_argument("--steps", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64, bert_precision=a.precision)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
batches = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
for i in range(4):
    tr.update(batches[i % 2], i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False) as prof:
    for i in range(a.steps):
        tr.update(batches[i % 2], i)
    torch.cuda.synchronize()
ev = prof.key_averages()
rows = []
for e in ev:
    dev_us = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
    rows.append((e.count / a.steps, e.self_cpu_time_total / a.steps, dev_us / a.steps, e.key))
rows.sort(key=lambda r: -r[0])
print("%8s %12s %12s  name" % ("calls", "self cpu us", "device us"))
for c, cpu, d, k in rows[:70]:
    print("%8.1f %12.1f %12.1f  %s" % (c, cpu, d, k[:100]))

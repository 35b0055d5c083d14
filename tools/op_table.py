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
ap.add_argument("--shapes", action="store_true", help="group by input shapes and sort by self device time: where the small launches come from")
a = ap.parse_args()
dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64, bert_precision=a.precision)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
batches = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
for i in range(4):
    tr.update(batches[i % 2], i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=a.shapes) as prof:
    for i in range(a.steps):
        tr.update(batches[i % 2], i)
    torch.cuda.synchronize()
if a.shapes:
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        d = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
        if d > 0 and not e.key.startswith(("void ", "_Z", "Memcpy", "Memset", "hip")) and "kernel" not in e.key:
            rows.append((d / a.steps, e.count / a.steps, e.key, str(e.input_shapes)[:110]))
    rows.sort(key=lambda r: -r[0])
    print("%10s %7s  op / input shapes" % ("device us", "calls"))
    for d, c, k, sh in rows[:90]:
        print("%10.1f %7.1f  %-40s %s" % (d, c, k[:40], sh))
    sys.exit(0)
ev = prof.key_averages()
rows = []
for e in ev:
    dev_us = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
    rows.append((e.count / a.steps, e.self_cpu_time_total / a.steps, dev_us / a.steps, e.key))
rows.sort(key=lambda r: -r[0])
print("%8s %12s %12s  name" % ("calls", "self cpu us", "device us"))
for c, cpu, d, k in rows[:70]:
    print("%8.1f %12.1f %12.1f  %s" % (c, cpu, d, k[:100]))

#!/usr/bin/env python3
"""Where does the host time of a training step go?  torch.profiler table by self CPU time."""
import os, sys
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
with profile(activities=[ProfilerActivity.CPU], record_shapes=False) as prof:
    for i in range(4):
        tr.update(batches[i % 2], i)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))

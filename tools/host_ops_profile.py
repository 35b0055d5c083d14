#!/usr/bin/env python3
"""Host-side cost of the step per autograd node / ATen op (torch.profiler, CPU activity only), forward and backward separately."""
import os, sys, time
import torch
from torch.profiler import profile, ProfilerActivity
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
net = tr.network
N = 4
for which in ("forward", "backward"):
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        for i in range(N):
            q, ocr, od, targets, _ = batches[i % 2]
            net.train(); net.drop_emb = True
            if which == "forward":
                scores = net(q, ocr, od)[0]
                torch.cuda.synchronize()
            else:
                prof_on = False
                scores = net(q, ocr, od)[0]
            loss = tr.loss_func(scores, targets)
            tr.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            torch.cuda.synchronize()
    print("==== profile over %d steps (%s shown; divide by %d) ====" % (N, "all", N))
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))
    break
tr.close(final=True)

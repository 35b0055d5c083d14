#!/usr/bin/env python3
"""Where does the data-parallel machinery cost time on ONE rank?  Times the segments of a training step (host clock, device
synchronised between segments) with and without a world-size-1 RCCL group.   python tools/dp_overhead.py"""
import os, sys, time
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import dp, synth
from ruart_amd.arguments import default_opt

dev = torch.device("cuda:0")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
dp.init_process_group(dev, "nccl", rank=0, world_size=1)
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)


def run(use_dp):
    o = dict(opt)
    o["ruart_dp"] = use_dp
    tr, _ = bench.build_trainer(o, synth.bert_config(), dev, process_group=dist.group.WORLD if use_dp else None)
    batches = [tr.ToCUDA(synth.synthetic_batch(o, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
    net = tr.network
    seg = []
    for i in range(12):
        b = batches[i % 2]
        net.train(); net.drop_emb = True
        torch.cuda.synchronize(); t = [time.perf_counter()]
        scores, _ = net(b[0], b[1], b[2]); loss = tr.loss_func(scores, b[3])
        torch.cuda.synchronize(); t.append(time.perf_counter())
        tr.optimizer.zero_grad(set_to_none=True); loss.backward()
        torch.cuda.synchronize(); t.append(time.perf_counter())
        if tr.grad_sync is not None:
            tr.grad_sync.average_gradients()
        torch.cuda.synchronize(); t.append(time.perf_counter())
        tr.optimizer.clip_and_step(10.0, extra_sq=tr.grad_sync.pinned_sq if tr.grad_sync is not None else None)
        torch.cuda.synchronize(); t.append(time.perf_counter())
        if i >= 4:
            seg.append(np.diff(t) * 1e3)
    s = np.array(seg).mean(0)
    print("%-6s fwd %.2f  bwd %.2f  average_gradients %.2f  clip+step %.2f  | total %.2f ms (encoder inline, segments synchronised)" %
          ("dp" if use_dp else "plain", s[0], s[1], s[2], s[3], s.sum()))
    tr.close()


for flag in (False, False, True, False, True):
    run(flag)
dist.destroy_process_group()

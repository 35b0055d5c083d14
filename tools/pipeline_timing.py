#!/usr/bin/env python3
"""Step timing with the encoder prefetched one step ahead: host enqueue time vs wait-for-GPU, and the trunk alone
(encoder result already there, nothing prefetched) for reference."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
cfg = synth.bert_config()
tr, _ = bench.build_trainer(opt, cfg, dev)
batches = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
net = tr.network
hi = torch.cuda.Stream(device=dev, priority=-1)


def step(i, mode):
    """mode: 'pipe' = prefetch next; 'inline' = encoder inline; 'trunk' = encoder output ready before timing, none launched"""
    b = batches[i % 2]
    nb = batches[(i + 1) % 2]
    net.train(); net.drop_emb = True
    if mode == "trunk":
        net.Bert.prefetch(b[0]["_ruart_index"].packed)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(hi):
        if mode == "pipe":
            net.prefetch_bert(nb[0], nb[1], nb[2])
        scores, _ = net(b[0], b[1], b[2])
        t1 = time.perf_counter()
        loss = tr.loss_func(scores, b[3])
        tr.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        t2 = time.perf_counter()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 10)
        tr.optimizer.step()
        t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    return [(t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3]


for mode in ("inline", "pipe", "trunk"):
    for i in range(3):
        step(i, mode)
    r = np.array([step(i, mode) for i in range(10)])
    print("%-7s host enqueue ms: fwd %.2f  bwd %.2f  opt %.2f | wait-for-GPU %.2f | total %.2f" % ((mode,) + tuple(r.mean(0))))

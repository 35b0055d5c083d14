#!/usr/bin/env python3
"""Where the --include-h2d step spends its extra time: host time of ToCUDA and of update() per step, batches shipped from pinned host
copies every step against batches resident on the device."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt
from ruart_amd.batch import BatchIndex
from torch.utils.data._utils.pin_memory import pin_memory

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
host = []
for i in range(3):
    hb = synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)
    hb[0]["_ruart_host_index"] = BatchIndex(hb[0], hb[1], hb[2], opt)
    host.append(pin_memory(hb))
resident = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(3)]


def fresh(i, h2d):
    if not h2d:
        return resident[i % 3]
    hb = host[i % 3]
    hb[0].pop("_ruart_index", None)
    hb[0]["_ruart_host_index"].device = None
    return tr.ToCUDA(hb)


import cProfile, pstats
for h2d in (False, True, False, True):
    pr = cProfile.Profile() if "--trace" in sys.argv else None
    staged = {0: fresh(0, h2d)}
    acc = [0.0, 0.0]
    torch.cuda.synchronize()
    for i in range(25):
        if i == 5:
            torch.cuda.synchronize()
            acc = [0.0, 0.0]
            t0 = time.perf_counter()
            if pr: pr.enable()
        a = time.perf_counter()
        staged[i + 1] = fresh(i + 1, h2d)
        b = time.perf_counter()
        tr.update(staged.pop(i), i, next_batch=staged[i + 1])
        c = time.perf_counter()
        acc[0] += b - a
        acc[1] += c - b
    torch.cuda.synchronize()
    if pr:
        pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(14)
    print("h2d=%s  %.2f ms per step: staging the lookahead batch %.2f ms, update() %.2f ms" %
          (h2d, (time.perf_counter() - t0) / 20 * 1e3, acc[0] / 20 * 1e3, acc[1] / 20 * 1e3))
tr.close(final=True)

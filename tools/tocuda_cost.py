#!/usr/bin/env python3
"""Host time of SDNetTrainer.ToCUDA on a collated bench batch (index prepared beforehand, as a loader worker would): pageable vs pinned."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt
from ruart_amd.batch import BatchIndex

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
hb = synth.synthetic_batch(opt, 64, seed=7, n_q=30, n_ocr=100, n_od=36)
hb[0]["_ruart_host_index"] = BatchIndex(hb[0], hb[1], hb[2], opt)


def pin(o):
    if torch.is_tensor(o):
        return o.pin_memory()
    if isinstance(o, dict):
        return {k: pin(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(pin(v) for v in o)
    if hasattr(o, "pin_memory"):
        return o.pin_memory()
    return o


for name, b in (("pageable", hb), ("pinned", pin(hb))):
    ts = []
    for _ in range(6):
        b[0].pop("_ruart_index", None)
        b[0]["_ruart_host_index"].device = None
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = tr.ToCUDA(b)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append(((t1 - t) * 1e3, (time.perf_counter() - t) * 1e3))
    n = sum(1 for _ in bench.__dict__) and 0
    print("%-9s ToCUDA host %.2f ms, until copies are done %.2f ms" % (name, min(a for a, _ in ts), min(b_ for _, b_ in ts)))
tr.close(final=True)

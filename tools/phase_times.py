#!/usr/bin/env python3
"""Host (enqueue) time against device completion time of a training step's phases, one phase at a time with a device sync between
them: is a phase waiting for the host or for the GPU?   python tools/phase_times.py [--unlock]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt

unlock = "--unlock" in sys.argv
dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
if unlock:
    opt.pop("LOCK_BERT", None)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
batches = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
for i in range(4):
    tr.update(batches[i % 2], i)
torch.cuda.synchronize()
net = tr.network
acc = {}


def phase(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    a = acc.setdefault(name, [0.0, 0.0])
    a[0] += t1 - t0
    a[1] += t2 - t0
    return out


N = 8
for i in range(N):
    q, ocr, od, targets, _ = batches[i % 2]
    net.train()
    net.drop_emb = True
    scores = phase("forward", lambda: net(q, ocr, od)[0])
    loss = phase("loss", lambda: tr.loss_func(scores, targets))
    tr.optimizer.zero_grad(set_to_none=True)
    phase("backward", loss.backward)
    phase("clip+step", lambda: tr.optimizer.clip_and_step(opt["grad_clipping"]))
print("phase          host enqueue   until device done   (ms, mean of %d steps, phases separated by device syncs)" % N)
for k, (h, d) in acc.items():
    print("%-12s %10.2f %16.2f" % (k, h / N * 1e3, d / N * 1e3))
tr.close(final=True)

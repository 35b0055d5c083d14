#!/usr/bin/env python3
"""Is the training step host-bound?  Measures host enqueue time vs total step time, and the BERT-only portion."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt
from ruart_amd.bert import bert_encode

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
cfg = synth.bert_config()
tr, _ = bench.build_trainer(opt, cfg, dev)
batches = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
net = tr.network
def step(i, sync_inside=True):
    b = batches[i % 2]
    net.train(); net.drop_emb = True
    t0 = time.perf_counter()
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
for i in range(3): step(i)
r = np.array([step(i) for i in range(10)])
print("host enqueue ms: fwd %.2f  bwd %.2f  opt %.2f | wait-for-GPU %.2f | total %.2f" % tuple(r.mean(0)))
# BERT alone
bi = batches[0][0]["_ruart_index"]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): bert_encode(net.Bert.weights, bi.packed)
torch.cuda.synchronize(); print("bert_encode alone: %.2f ms" % ((time.perf_counter() - t0) * 100))
# forward only, no grad
net.eval(); net.drop_emb = False
with torch.no_grad():
    for _ in range(2): net(*batches[0][:3])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10): net(*batches[i % 2][:3])
    torch.cuda.synchronize(); print("forward only (eval, no grad): %.2f ms" % ((time.perf_counter() - t0) * 100))
# host-side batch preparation cost
t0 = time.perf_counter()
b = synth.synthetic_batch(opt, 64, seed=99, n_q=30, n_ocr=100, n_od=36)
t1 = time.perf_counter()
tr.ToCUDA(b); torch.cuda.synchronize()
t2 = time.perf_counter()
print("synthetic batch gen %.1f ms; ToCUDA (index prep + packing + H2D) %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
# the same with the host index already built (what VQA_collate(opt, prepare_index=True) does inside a DataLoader worker)
from ruart_amd.batch import BatchIndex
b = synth.synthetic_batch(opt, 64, seed=98, n_q=30, n_ocr=100, n_od=36)
t0 = time.perf_counter()
b[0]["_ruart_host_index"] = BatchIndex(b[0], b[1], b[2], opt)
t1 = time.perf_counter()
tr.ToCUDA(b); torch.cuda.synchronize()
t2 = time.perf_counter()
print("host index in the collate worker %.1f ms; ToCUDA with it (H2D only) %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))

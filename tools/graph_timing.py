#!/usr/bin/env python3
"""Trunk as captured graphs vs eager: host time and total time of one training step (encoder output prefetched, so only the
front + trunk + optimizer are in the timed region)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt

dev = torch.device("cuda:0")
for graph, streams in ((False, True), (False, False), (True, True), (True, False)):
    opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64, ruart_graph_trunk=graph, ruart_streams=streams)
    cfg = synth.bert_config()
    tr, _ = bench.build_trainer(opt, cfg, dev)
    batches = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
    net = tr.network
    hi = torch.cuda.Stream(device=dev, priority=-1)

    def step(i):
        b = batches[i % 2]
        net.train(); net.drop_emb = True
        net.Bert.prefetch(b[0]["_ruart_index"].packed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(hi):
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
        return [(t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3], float(loss)

    for i in range(6):
        step(i)
    r = np.array([step(i)[0] for i in range(10)])
    print("graph=%s three-streams=%s  host ms: fwd %.2f  bwd %.2f  opt %.2f | wait-for-GPU %.2f | total %.2f   (loss %.5f)" %
          ((graph, streams) + tuple(r.mean(0)) + (step(0)[1],)), flush=True)
    del tr, net

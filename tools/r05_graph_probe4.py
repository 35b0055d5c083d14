#!/usr/bin/env python3
"""The full step (front + trunk + optimizer, encoder output prefetched and waited for) with the trunk as make_graphed_callables graphs
against eager, one stream: where do the extra ~10 ms of tools/graph_timing.py go?  torch.profiler kernel table of one step each."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt

dev = torch.device("cuda:0")
for graph in (False, True):
    opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64, ruart_graph_trunk=graph, ruart_streams=False)
    tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
    batches = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
    net = tr.network
    hi = torch.cuda.Stream(device=dev, priority=-1)

    def step(i, marks=None):
        b = batches[i % 2]
        net.train(); net.drop_emb = True
        net.Bert.prefetch(b[0]["_ruart_index"].packed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(hi):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
            scores, _ = net(b[0], b[1], b[2])
            ev[1].record()
            loss = tr.loss_func(scores, b[3])
            tr.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            ev[2].record()
            tr.optimizer.clip_and_step(10.0)
            ev[3].record()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3, [ev[k].elapsed_time(ev[k + 1]) for k in range(3)]

    for i in range(6):
        step(i)
    r = [step(i) for i in range(8)]
    print("graph=%s: total %.2f ms; device fwd %.2f  bwd %.2f  opt %.2f" % ((graph, np.median([x[0] for x in r])) + tuple(np.median([x[1][k] for x in r]) for k in range(3))), flush=True)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as p:
        step(0)
    ka = p.key_averages()
    tot = sum(k.self_device_time_total for k in ka) / 1e3
    print("   sum of kernel time in one profiled step: %.2f ms" % tot)
    print(ka.table(sort_by="self_cuda_time_total", row_limit=14, max_name_column_width=60))
    tr.close()
    del tr, net

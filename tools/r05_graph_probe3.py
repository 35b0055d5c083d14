#!/usr/bin/env python3
"""torch.cuda.make_graphed_callables on the trunk (one stream): device time of its forward replay and of its backward replay, each
between events, against the eager trunk and against ONE hand-made graph of forward + backward (tools/r05_graph_probe2.py: 10.2 ms)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ruart_amd import synth  # noqa: E402
from ruart_amd.arguments import default_opt  # noqa: E402

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64, ruart_streams=False)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
net = tr.network
b = tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7, n_q=30, n_ocr=100, n_od=36))
grabbed = {}
orig = net._trunk_callable


def grab(*args):
    grabbed["args"] = tuple(t.detach().clone() for t in args)
    return orig(*args)


net._trunk_callable = grab
net.train()
net.drop_emb = True
with torch.no_grad():
    net(b[0], b[1], b[2])
torch.cuda.synchronize()
args = tuple(a.clone().requires_grad_(a.dtype == torch.float32 and i in (0, 3, 4)) for i, a in enumerate(grabbed["args"]))
trunk = net._trunk_module()
trunk.train(True)


def ev():
    return torch.cuda.Event(enable_timing=True)


def measure(fn_fwd, reps=12):
    f, bw, tot = [], [], []
    for _ in range(reps):
        net.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        out = fn_fwd(*args)
        e1.record()
        out.backward(torch.ones_like(out))
        e2.record()
        e2.synchronize()
        f.append(e0.elapsed_time(e1)); bw.append(e1.elapsed_time(e2)); tot.append(e0.elapsed_time(e2))
    med = lambda v: sorted(v)[len(v) // 2]
    return med(f), med(bw), med(tot)


s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s):
    for _ in range(3):
        measure(trunk, 1)
    print("eager trunk (autograd)       : fwd %.3f  bwd %.3f  total %.3f ms" % measure(trunk), flush=True)
    torch.cuda.synchronize()
    sample = tuple(a.detach().clone().requires_grad_(a.requires_grad) for a in args)
    graphed = torch.cuda.make_graphed_callables(trunk, sample, num_warmup_iters=3, allow_unused_input=True)
    net.zero_grad(set_to_none=True)
    for _ in range(3):
        measure(graphed, 1)
    print("make_graphed_callables trunk : fwd %.3f  bwd %.3f  total %.3f ms" % measure(graphed), flush=True)
    # where does the backward's time go: the replay itself, or what autograd does with ~150 returned gradients?
    import torch.autograd.profiler as prof
    net.zero_grad(set_to_none=True)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as p:
        out = graphed(*args)
        out.backward(torch.ones_like(out))
        torch.cuda.synchronize()
    print(p.key_averages().table(sort_by="cuda_time_total", row_limit=25))
tr.close(final=True)

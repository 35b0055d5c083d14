#!/usr/bin/env python3
"""Which ingredient makes the trunk slow as a torch-captured graph?  tools/r05_graph_probe.py: the eval-mode forward replays at eager
speed (133 nodes).  Here, one stream, torch.cuda.graph, device time eager vs replay for
  1. eval forward, no autograd          2. train-mode forward (dropout masks drawn inside), no autograd
  3. train-mode forward under autograd  4. forward + backward (torch.autograd.grad inside the capture)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ruart_amd import synth  # noqa: E402
from ruart_amd.arguments import default_opt  # noqa: E402
import ruart_amd.layers as L  # noqa: E402

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
args = grabbed["args"]
trunk = net._trunk_module()
params = [p for p in trunk.parameters() if p.requires_grad]
p_drop = L.dropout_p


def timed(fn, reps=15):
    dv, hs = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        e1.record()
        e1.synchronize()
        dv.append(e0.elapsed_time(e1))
        hs.append((t1 - t0) * 1e3)
    dv.sort()
    hs.sort()
    return dv[len(dv) // 2], hs[len(hs) // 2]


def variant(name, train, grad, bwd):
    trunk.train(train)
    L.set_dropout_prob(p_drop if train else 0.0)
    fargs = tuple(a.clone().requires_grad_(grad and a.dtype == torch.float32 and i in (0, 3, 4)) for i, a in enumerate(args))
    keep = {}

    def run():
        with torch.enable_grad() if grad else torch.no_grad():
            out = trunk(*fargs)
            if bwd:
                ins = [a for a in fargs if a.requires_grad] + params
                keep["g"] = torch.autograd.grad(out, ins, torch.ones_like(out), allow_unused=True)
            keep["out"] = out
        return out

    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        for _ in range(4):
            run()
        torch.cuda.synchronize()
        de, he = timed(run)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            run()
        torch.cuda.synchronize()
        g.replay()
        dg, hg = timed(g.replay)
    print("%-52s eager device %7.3f ms (host %6.3f) | graph replay device %7.3f ms (host %6.3f)" % (name, de, he, dg, hg), flush=True)
    del g, keep


variant("1. eval forward, no autograd", False, False, False)
variant("2. train-mode forward, no autograd", True, False, False)
variant("3. train-mode forward under autograd", True, True, False)
variant("4. train-mode forward + backward", True, True, True)
tr.close(final=True)

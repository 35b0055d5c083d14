#!/usr/bin/env python3
"""Does a graph replay slow down when its stream (or the device) is busy at launch time?  The trunk as make_graphed_callables graphs and
eager (one stream), forward + backward device time between events:
   idle device | behind an encoder pass enqueued on the SAME stream | beside an encoder pass on ANOTHER stream (unmasked / 240 CUs)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ruart_amd import hip, synth  # noqa: E402
from ruart_amd.arguments import default_opt  # noqa: E402
from ruart_amd.bert import bert_encode, _Buffers  # noqa: E402

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64, ruart_streams=False)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
net = tr.network
b = tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7, n_q=30, n_ocr=100, n_od=36))
packed = b[0]["_ruart_index"].packed
W = net.Bert.weights
bufs = _Buffers()
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


def measure(fn, before=None, reps=8):
    tot = []
    for _ in range(reps):
        net.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        if before is not None:
            before()
        e0, e2 = ev(), ev()
        e0.record()
        out = fn(*args)
        out.backward(torch.ones_like(out))
        e2.record()
        e2.synchronize()
        torch.cuda.synchronize()
        tot.append(e0.elapsed_time(e2))
    return sorted(tot)[len(tot) // 2]


s = torch.cuda.Stream(device=dev)
other = torch.cuda.Stream(device=dev)
masked = hip.cu_masked_stream(240, dev)


def enc_same():
    bert_encode(W, packed, bufs)


def enc_on(st):
    def f():
        with torch.cuda.stream(st):
            bert_encode(W, packed, bufs)
    return f


with torch.cuda.stream(s):
    for _ in range(3):
        measure(trunk, reps=1)
    sample = tuple(a.detach().clone().requires_grad_(a.requires_grad) for a in args)
    graphed = torch.cuda.make_graphed_callables(trunk, sample, num_warmup_iters=3, allow_unused_input=True)
    net.zero_grad(set_to_none=True)
    for _ in range(3):
        measure(graphed, reps=1)
    for name, before in (("idle device", None), ("behind an encoder pass on the SAME stream", enc_same),
                         ("beside an encoder pass on another stream", enc_on(other)), ("beside an encoder pass on a 240-CU stream", enc_on(masked))):
        print("%-46s eager fwd+bwd %7.3f ms | graphs %7.3f ms" % (name, measure(trunk, before), measure(graphed, before)), flush=True)
torch.cuda.synchronize()
hip.destroy_stream(masked)
tr.close(final=True)

#!/usr/bin/env python3
"""The MFMA LSTM recurrence alone and beside the frozen encoder's pass on a second (CU-masked) stream: how much of its per-step
latency is memory latency under load?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth, ops, hip
from ruart_amd.arguments import default_opt
from ruart_amd.bert import bert_encode, _Buffers

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
b = tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7, n_q=30, n_ocr=100, n_od=36))
packed = b[0]["_ruart_index"].packed
W = tr.network.Bert.weights
bf = _Buffers()
enc = hip.cu_masked_stream(240, dev)
B, T, h = 64, 100, 125
xp = torch.randn(B, T, 8 * h, device=dev, requires_grad=True)
whh = (torch.randn(2, 4 * h, h, device=dev) * 0.05).requires_grad_(True)
y = ops._LstmRecurrence.apply(xp, whh, 2)
gy = torch.randn_like(y)


def fwd():
    ops._LstmRecurrence.apply(xp.detach(), whh.detach(), 2)


def fb():
    ops._LstmRecurrence.apply(xp, whh, 2).backward(gy)


def timeit(f, n, busy):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    if busy:
        with torch.cuda.stream(enc):
            for _ in range(3):
                bert_encode(W, packed, bf)           # ~50 ms of GEMM / attention / LN beside the timed calls
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    time.sleep(0.002)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for busy in (False, True, False, True):
    print("encoder beside it: %-5s  fwd %.1f us   fwd+bwd(+dW) %.1f us" % (busy, timeit(fwd, 40, busy), timeit(fb, 20, busy)))
# the trunk's wide-but-short kernels beside the same pass
xa = torch.randn(6400, 250, device=dev, requires_grad=True)
wa = torch.randn(1000, 250, device=dev, requires_grad=True)
ba = torch.randn(1000, device=dev, requires_grad=True)
ga = torch.randn(6400, 1000, device=dev)
e1 = torch.randn(64, 100, 250, device=dev)
e2 = torch.randn(64, 100, 250, device=dev)
from ruart_amd import layers as L
cases = {
    "linear 6400x250 -> 1000, forward": lambda: ops.linear(xa.detach(), wa.detach(), ba.detach()),
    "linear 6400x250 -> 1000, forward + backward": lambda: ops.linear(xa, wa, ba).backward(ga),
    "elementwise add (64,100,250)": lambda: e1 + e2,
    "cat 2 x (64,100,250)": lambda: torch.cat([e1, e2], 2),
}
for name, f in cases.items():
    for busy in (False, True):
        print("%-45s encoder beside it: %-5s %.1f us" % (name, busy, timeit(f, 60, busy)))
hip.destroy_stream(enc)
tr.close(final=True)

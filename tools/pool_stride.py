#!/usr/bin/env python3
"""Sub-word pooling + layer mix (ruart_bert_pool_mix) against the stride between consecutive layers' outputs.  Question: with
the layers packed back to back a token's 12 rows are multiples of 512 KB apart at 43 008 rows - do they collide on one HBM
channel?  Answer (MI355X): no - padding of 1 KB .. 132 KB per layer leaves the time unchanged (~2 TB/s on the real pieces)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip
lib = hip.load(); d = torch.device("cuda:0")
Tp, H, NL = 43008, 768, 12
g = np.random.default_rng(0)
# words: ~1.5 pieces each over the token stream, skipping 2 of every ~5 tokens (CLS / SEP)
starts, lens, t = [], [], 0
while t < Tp - 8:
    t += 1
    for _ in range(int(g.integers(1, 3))):
        n = int(g.integers(1, 3)); starts.append(t); lens.append(n); t += n
    t += 1
W = len(starts)
ss = torch.tensor(starts, dtype=torch.int32, device=d); ll = torch.tensor(lens, dtype=torch.int32, device=d)
dst = torch.arange(W, dtype=torch.int32, device=d)
lw = torch.rand(NL, device=d)
out = torch.zeros(W, H, device=d)
for pad in (0, 512, 2048, 2048 + 64, 8192 + 2048, 65536 + 2048):
    stride = Tp * H + pad
    buf = torch.randn(NL * stride, device=d).to(torch.float16)
    def run():
        rc = lib.ruart_bert_pool_mix(hip.ptr(buf), stride, H, hip.DT_F16, NL, hip.ptr(ss), None, hip.ptr(ll), hip.ptr(dst), hip.ptr(lw), hip.ptr(out), H, W, H, hip.stream_ptr())
        assert rc == 0
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    pieces = int(sum(lens))
    print("pad %6d halves (%7d B): %7.1f us  %.2f TB/s  (%d words, %d pieces)" % (pad, pad * 2, us, pieces * NL * H * 2 / us / 1e6, W, pieces), flush=True)

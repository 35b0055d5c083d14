#!/usr/bin/env python3
"""Device time of the SDNet attention kernels at the step's shapes: forward, backward-q and backward-kv, for both kernel forms
(ruart_attn_set_prefetch 0 / 1), each launch between two events through the C ABI (no autograd, no host gaps in the figure)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip
lib = hip.load()
d = torch.device("cuda:0")
SHAPES = [(3, 64, 36, 40, 250, 250, 250), (3, 64, 100, 40, 250, 250, 250), (2, 64, 100, 36, 125, 250, 0), (1, 64, 61, 40, 300, 300, 1),
          (1, 64, 224, 40, 300, 300, 1), (1, 64, 40, 40, 300, 250, 300), (1, 64, 36, 36, 250, 250, 250), (1, 64, 100, 100, 250, 250, 250)]
def ev(): return torch.cuda.Event(enable_timing=True)
def timed(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = ev(), ev(); e0.record(); f(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
tot = {0: [0.0, 0.0], 1: [0.0, 0.0]}
for cnt, B, L1, L2, h, D3, dl in SHAPES:
    a = torch.randn(B, L1, h, device=d); k = torch.randn(B, L2, h, device=d); v = torch.randn(B, L2, D3, device=d)
    m = torch.ones(B, L2, dtype=torch.uint8, device=d); diag = torch.randn(max(dl, 1), device=d) if dl else None
    out = torch.empty(B, L1, D3, device=d); probs = torch.empty(B, L1, L2, device=d); go = torch.randn(B, L1, D3, device=d)
    ga, gk, gv, ds = torch.empty_like(a), torch.empty_like(k), torch.empty_like(v), torch.empty_like(probs)
    gd = torch.empty(B * ((L1 + 15) // 16), h, device=d) if dl > 1 else None
    S = hip.stream_ptr()
    fwd = lambda: hip.check(lib.ruart_attn_fwd_pscale(hip.ptr(a), hip.ptr(k), hip.ptr(v), hip.ptr(m), hip.ptr(diag), dl, 1, None, hip.ptr(out), hip.ptr(probs), B, L1, L2, h, D3, hip.stream_ptr()), "fwd")
    bwd = lambda: hip.check(lib.ruart_attn_bwd_pscale(hip.ptr(a), hip.ptr(k), hip.ptr(v), hip.ptr(probs), hip.ptr(go), hip.ptr(diag), dl, 1, None, hip.ptr(ga), hip.ptr(gk), hip.ptr(gv), hip.ptr(gd), hip.ptr(ds), B, L1, L2, h, D3, hip.stream_ptr()), "bwd")
    line = "x%d (%d,%d)x(%d) h %d D3 %d diag %d:" % (cnt, B, L1, L2, h, D3, dl)
    for form in (0, 1):
        lib.ruart_attn_set_prefetch(form)
        tf, tb = timed(fwd), timed(bwd)
        tot[form][0] += cnt * tf; tot[form][1] += cnt * tb
        line += "   [form %d] fwd %5.1f  bwd (q + kv) %6.1f us" % (form, tf, tb)
    print(line)
for form in (0, 1):
    print("form %d: forward %.2f ms + backward %.2f ms per step" % (form, tot[form][0] / 1e3, tot[form][1] / 1e3))
lib.ruart_attn_set_prefetch(1)

#!/usr/bin/env python3
"""Every fused-attention call of one training step (shapes), timed standalone: forward and backward kernels."""
import os, sys
from collections import Counter
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth, ops
from ruart_amd.arguments import default_opt

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
batches = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
for i in range(3):
    tr.update(batches[i % 2], i)
calls = Counter()
orig = ops.fused_attention
def spy(a, k, v, mask, diag=None, relu=False):
    calls[(tuple(a.shape), tuple(k.shape), tuple(v.shape), diag is not None and diag.numel() > 1, bool(relu))] += 1
    return orig(a, k, v, mask, diag=diag, relu=relu)
ops.fused_attention = spy
import ruart_amd.layers as L
tr.update(batches[0], 3)
ops.fused_attention = orig
torch.cuda.synchronize()
def timeit(f, n=10):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
from ruart_amd import hip
FORMS = [int(x) for x in os.environ.get("ATTN_FORMS", "0,1").split(",")]      # ruart_attn_set_prefetch settings to time
tot = {f: 0.0 for f in FORMS}
for (sa, sk, sv, hasd, relu), cnt in sorted(calls.items(), key=lambda kv: -kv[1]):
    a = torch.randn(*sa, device=dev, requires_grad=True); k = torch.randn(*sk, device=dev, requires_grad=True)
    v = torch.randn(*sv, device=dev, requires_grad=True); m = torch.ones(sk[0], sk[1], dtype=torch.uint8, device=dev)
    d = torch.randn(sa[2], device=dev, requires_grad=True) if hasd else None
    out = orig(a, k, v, m, diag=d, relu=relu); g = torch.randn_like(out)
    def fb():
        o = orig(a, k, v, m, diag=d, relu=relu); o.backward(g)
    line = "x%d  a %s k %s v %s diag %s relu %s:" % (cnt, sa, sk, sv, hasd, relu)
    for f in FORMS:
        hip.check(hip.load().ruart_attn_set_prefetch(f), "set_prefetch")
        tf = timeit(lambda: orig(a.detach(), k.detach(), v.detach(), m, diag=None if d is None else d.detach(), relu=relu))
        tfb = timeit(fb)
        tot[f] += cnt * tfb
        line += "  [form %d] fwd %.1f us, fwd+bwd %.1f us" % (f, tf, tfb)
    print(line)
for f in FORMS:
    print("form %d: sum fwd+bwd over the step: %.2f ms (standalone, includes host launch gaps)" % (f, tot[f] / 1e3))
hip.check(hip.load().ruart_attn_set_prefetch(1), "set_prefetch")
tr.close(final=True)

#!/usr/bin/env python3
"""ruart_bert_attention_split on a bench-shaped packed stream (windows of whole short sequences, fp32 [Q | K | V] rows): time per call
for every heads-per-workgroup setting (0 = the one-head kernel of rounds 2-4), bitwise comparison with that kernel.  The Infinity
Cache is flushed between calls (a 512 MB fill), as the encoder's own traffic does between two attention launches of a pass.

    python tools/attn_split_bench.py [--heads 0,2,3,4,6,12] [--long]
"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip, synth
from ruart_amd.arguments import default_opt
from ruart_amd.batch import BatchIndex

ap = argparse.ArgumentParser()
ap.add_argument("--heads", default="0,2,3,4,6,12")
ap.add_argument("--long", action="store_true", help="add 64 sequences of 200 pieces (multi-tile windows)")
ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda:0")
lib = hip.load()
opt = default_opt(vocab_size=20000, cuda=True, device=dev, bert_precision="fp16c", max_od_num=36, batch_size=64)
opt["bert_config"] = synth.bert_config()
q, ocr, od, gt, _ = synth.synthetic_batch(opt, 64, seed=7, n_q=30, n_ocr=100, n_od=36)
if a.long:
    ids = torch.randint(1000, 30000, (64, 200))
    q = dict(q); q["bert"] = ids; q["bert_mask"] = torch.ones_like(ids, dtype=torch.bool)
    from ruart_amd.bert import PackedTokens
    p = PackedTokens([(q["bert"], q["bert_mask"]), (ocr["bert"], ocr["bert_mask"]), (od["bert"], od["bert_mask"])], dev, mfma_long=False)
else:
    p = BatchIndex(q, ocr, od, opt, dev, pack=True, mfma_long=True).packed
H, NH = 768, 12
T, Tp, nb = p.T, p.Tp, p.n_blocks
g = torch.Generator(device="cpu").manual_seed(3)
qkv = (torch.randn(Tp, 3 * H, generator=g) * 1.5).to(dev)
ctx16 = torch.zeros(Tp, H, dtype=torch.float16, device=dev)
ctx8 = torch.zeros(Tp, 2 * H, dtype=torch.uint8, device=dev)
flush = torch.empty(128 << 20, dtype=torch.float32, device=dev)
print("tokens %d, windows %d, bytes per call %.1f MB" % (T, nb, T * 12288 / 1e6))


def call():
    hip.check(lib.ruart_bert_attention_split(hip.ptr(qkv), 3 * H, hip.ptr(ctx16), hip.ptr(ctx8), H, H, NH, nb, hip.ptr(p.blk[0]), hip.ptr(p.blk[1]),
                                             hip.ptr(p.blk[2]), hip.ptr(p.blk[3]), hip.ptr(p.tok_lo), hip.ptr(p.tok_hi), None, hip.stream_ptr()), "attn")


ref = None
for hpg in [int(x) for x in a.heads.split(",")]:
    hip.check(lib.ruart_bert_attention_split_set_heads(hpg), "set_heads")
    ctx16.zero_(); ctx8.zero_()
    call()
    torch.cuda.synchronize()
    out = (ctx16[:T].clone(), ctx8[:T].clone())
    if ref is None:
        ref = out
    same = bool((out[0].view(torch.int16) == ref[0].view(torch.int16)).all() and (out[1] == ref[1]).all())
    import hashlib
    digest = hashlib.sha1(out[0].cpu().numpy().tobytes() + out[1].cpu().numpy().tobytes()).hexdigest()[:12]
    ts = []
    for _ in range(a.reps):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    us = ts[len(ts) // 2]
    print("heads/workgroup %2d: %7.1f us  = %.2f TB/s of algorithmic bytes   bit-identical to the first setting: %s  sha1 %s" % (hpg, us, T * 12288 / us / 1e6, same, digest), flush=True)
hip.check(lib.ruart_bert_attention_split_set_heads(4), "set_heads")

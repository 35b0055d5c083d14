#!/usr/bin/env python3
"""North-star shape (B, L) = (64, 512): the encoder forward as ONE pass over all sequences against the same sequences split into
P independent parts on P streams.  At 32 768 rows the N = 768 products have 384 tiles = 1.5 rounds of the 256 CUs; parts on
separate streams let one part's kernels fill the other's partial rounds.   python tools/bert512_split.py [--parts 1,2,4]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip, synth
from ruart_amd.bert import BertEncoderWeights, PackedTokens, bert_encode, _Buffers

ap = argparse.ArgumentParser()
ap.add_argument("--parts", default="1,2,4")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--seq-len", type=int, default=512)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--precision", default="fp16")
a = ap.parse_args()
d = torch.device("cuda:0")
cfg = synth.bert_config()
W = BertEncoderWeights(synth.make_bert_weights(cfg, seed=1033, w_std=0.02), cfg, d, a.precision)
ids = torch.randint(1000, cfg["vocab_size"], (a.batch, a.seq_len))
flops = a.batch * a.seq_len * (169869312 + 36864 * a.seq_len)
for P in [int(x) for x in a.parts.split(",")]:
    per = a.batch // P
    parts = [PackedTokens([(ids[i * per:(i + 1) * per], torch.ones(per, a.seq_len, dtype=torch.bool))], d) for i in range(P)]
    bufs = [_Buffers() for _ in range(P)]
    streams = [torch.cuda.Stream(device=d) for _ in range(P)]
    def step():
        for p, b, s in zip(parts, bufs, streams):
            with torch.cuda.stream(s):
                bert_encode(W, p, b)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print("parts %d: %.3f ms  %.1f TFLOP/s  %.1f %% of 2.5 PF" % (P, dt * 1e3, flops / dt / 1e12, flops / dt / 2.5e13), flush=True)

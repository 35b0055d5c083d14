#!/usr/bin/env python3
"""Tile-walk sweep of the fp16c encoder GEMM: for every GROUP_M of ruart_gemm_set_tile_order the four projection shapes are run
`iters` times; prints the time per (order, shape) and writes the launch sequence to --seq, so that a rocprofv3 --pmc FETCH_SIZE
pass over the same command can be joined launch by launch (tools/gemm_corr_order_join.py).
    python tools/gemm_corr_order_sweep.py --orders 0,2,3,4,6,8,12,16 --iters 3 --seq gpurun_out/order_seq.json"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip  # noqa: E402
from ruart_amd.bert import split_f16c  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=42880)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--orders", default="0,2,3,4,6,8,12,16")
ap.add_argument("--seq", default="")
ap.add_argument("--hidden", type=int, default=768, help="1024 with --inter 4096: the bert-large (stress) projections")
ap.add_argument("--inter", type=int, default=3072)
ap.add_argument("--auto", action="store_true", help="also time the library's own per-shape choice (no ruart_gemm_set_tile_order)")
a = ap.parse_args()
lib = hip.load()
d = torch.device("cuda:0")
M = (a.rows + 255) // 256 * 256
H, I = a.hidden, a.inter
shapes = [("qkv", 3 * H, H, hip.ACT_NONE, False), ("ao", H, H, hip.ACT_NONE, True), ("ff1", I, H, hip.ACT_GELU, False),
          ("ff2", H, I, hip.ACT_NONE, True)]
g = torch.Generator(device="cpu").manual_seed(0)
ops = {}
for name, N, K, act, res in shapes:
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.03
    A16, A8 = [t.to(d) for t in split_f16c(A)]
    hi = W.half().float()
    W16 = W.half().to(d)
    W8 = torch.cat([hi * 2.0 ** hip.f16c_shifts()[2], (W - hi) * 2.0 ** hip.f16c_shifts()[3]], 1).clamp_(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).to(d)
    bias = torch.randn(N, generator=g).to(d)
    R32 = torch.randn(M, N, generator=g).to(d) if res else None
    gelu = act == hip.ACT_GELU
    Cc = torch.empty(M, N, dtype=torch.float16 if gelu else torch.float32, device=d)
    C8 = torch.empty(M, 2 * N, dtype=torch.uint8, device=d) if gelu else None
    ops[name] = (A16, A8, W16, W8, bias, R32, Cc, C8, N, K, act)


def run(name):
    A16, A8, W16, W8, bias, R32, Cc, C8, N, K, act = ops[name]
    assert lib.ruart_gemm_16c_nt(hip.ptr(A16), hip.ptr(A8), K, hip.ptr(W16), hip.ptr(W8), K, hip.ptr(bias), hip.ptr(R32), N, hip.ptr(Cc), N,
                                 hip.ptr(C8), M, N, K, act, hip.stream_ptr()) == 0


seq = []
for order in [int(x) for x in a.orders.split(",")]:
    assert lib.ruart_gemm_set_tile_order(order) == 0
    line = []
    for name, N, K, _, _ in shapes:
        run(name)                                   # warm (counted in seq too)
        seq.append([order, name])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run(name)
            seq.append([order, name])
        e1.record()
        torch.cuda.synchronize()
        line.append("%s %6.1f us" % (name, e0.elapsed_time(e1) * 1e3 / a.iters))
    print("order %2d: %s" % (order, "  ".join(line)), flush=True)
if a.auto:
    assert lib.ruart_gemm_set_tile_order(-1) == 0   # back to the rule in (row tiles, column tiles, K)
    line = []
    for name, N, K, _, _ in shapes:
        run(name)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run(name)
        e1.record()
        torch.cuda.synchronize()
        line.append("%s %6.1f us" % (name, e0.elapsed_time(e1) * 1e3 / a.iters))
    print("rule    : %s" % "  ".join(line), flush=True)
else:
    lib.ruart_gemm_set_tile_order(8)
if a.seq:
    json.dump({"rows": M, "seq": seq}, open(a.seq, "w"))

#!/usr/bin/env python3
"""Does splitting the encoder pass into concurrent half packs make it less sensitive to a sparse stream of small foreign kernels (what the
trunk's host-paced forward is)?  The pass of the bench batch on a 240-CU stream: one pack, or two / three packs on as many masked streams,
alone and beside a LOW-priority stream on which the host launches a small elementwise kernel over and over (~200 workgroups, ~5 us)."""
import os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth, hip
from ruart_amd.arguments import default_opt
from ruart_amd.bert import PackedTokens, bert_encode, _Buffers

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
W = tr.network.Bert.weights
q, ocr, od, _, _ = synth.synthetic_batch(opt, 64, seed=7, n_q=30, n_ocr=100, n_od=36)
groups = [(q["bert"], q["bert_mask"]), (ocr["bert"], ocr["bert_mask"]), (od["bert"], od["bert_mask"])]


def packs(n_parts):
    if n_parts == 1:
        return [PackedTokens(groups, dev)]
    parts = [[] for _ in range(n_parts)]
    for ids, m in groups:
        n = ids.shape[0]
        cuts = [n * i // n_parts for i in range(n_parts + 1)]
        for p in range(n_parts):
            parts[p].append((ids[cuts[p]:cuts[p + 1]], m[cuts[p]:cuts[p + 1]]))
    return [PackedTokens(g, dev) for g in parts]


noise_on = threading.Event()
stop = threading.Event()
noise_stream = hip.priority_stream(1, dev)
xs = torch.zeros(6400 * 256 * 4, device=dev)
noise_n = [200 * 1024]


def noise():
    torch.cuda.set_device(dev)
    with torch.cuda.stream(noise_stream):
        while not stop.is_set():
            if noise_on.is_set():
                xs[:noise_n[0]].add_(1.0)
            else:
                time.sleep(0.0005)


th = threading.Thread(target=noise, daemon=True)
th.start()


def timeit(f, n=8):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ps = packs(1)
bufs = [_Buffers()]
st = hip.cu_masked_stream(240, dev)


def run():
    cur = torch.cuda.current_stream()
    st.wait_stream(cur)
    with torch.cuda.stream(st):
        bert_encode(W, ps[0], bufs[0])
    cur.wait_stream(st)


print("encoder pass alone on 240 CUs: %.2f ms" % timeit(run), flush=True)
for wgs in (16, 64, 200, 800, 3200, 6400):
    noise_n[0] = wgs * 1024
    noise_on.set()
    time.sleep(0.05)
    t = timeit(run)
    noise_on.clear()
    noise_stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        xs[:noise_n[0]].add_(1.0)
    e1.record()
    torch.cuda.synchronize()
    print("beside a stream of %5d-workgroup elementwise kernels (%.1f us each alone, back to back): %.2f ms" % (wgs, e0.elapsed_time(e1) * 5, t), flush=True)
hip.destroy_stream(st)
stop.set()
th.join()
tr.close(final=True)

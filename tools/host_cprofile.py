#!/usr/bin/env python3
"""Python-level host cost of the trunk's FORWARD (the part of a pipelined step that is paced by the host's enqueue rate,
profiles/r04_host_delay.log): cProfile over SDNet.forward of 8 steps, top functions by own time and by cumulative time."""
import cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
batches = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
for i in range(4):
    tr.update(batches[i % 2], i, next_batch=batches[(i + 1) % 2])
torch.cuda.synchronize()
net = tr.network
orig = net.forward
pr = cProfile.Profile()
acc = {"t": 0.0, "n": 0}


def timed(*a, **k):
    t0 = time.perf_counter()
    pr.enable()
    out = orig(*a, **k)
    pr.disable()
    acc["t"] += time.perf_counter() - t0
    acc["n"] += 1
    return out


net.forward = timed
t0 = time.perf_counter()
for i in range(8):
    tr.update(batches[i % 2], i, next_batch=batches[(i + 1) % 2])
torch.cuda.synchronize()
print("step %.2f ms (profiled), SDNet.forward host time %.2f ms per step" % ((time.perf_counter() - t0) / 8 * 1e3, acc["t"] / acc["n"] * 1e3))
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print("\n".join(l[:170] for l in s.getvalue().splitlines()[:48]))

#!/usr/bin/env python3
"""One time base for the pipelined step: when every encoder GEMM of the pass that runs ahead starts and ends, and when the step
stream passes (1) the start of update, (2) the end of forward, (3) the end of backward, (4) the end of the optimizer.
Answers: how far does the encoder pass get while the trunk is on the device, and how long does it run alone?"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth, hip
from ruart_amd.arguments import default_opt

dev = torch.device("cuda:0")
lib = hip.load()
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
batches = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
for i in range(6):
    tr.update(batches[i % 2], i, next_batch=batches[(i + 1) % 2])
torch.cuda.synchronize()
net = tr.network
orig_fwd, orig_cs = net.forward, tr.optimizer.clip_and_step


def mark(tag):
    lib.ruart_prof_mark(tag, hip.stream_ptr(dev))


def fwd(*a, **k):
    mark(1)
    out = orig_fwd(*a, **k)
    mark(2)
    return out


def cs(*a, **k):
    mark(3)
    out = orig_cs(*a, **k)
    mark(4)
    return out


net.forward, tr.optimizer.clip_and_step = fwd, cs
FWD = "--fwd" in sys.argv                         # forward-only steps (evaluation mode, one host wait per batch as predict() has)


def one(i):
    if not FWD:
        return tr.update(batches[i % 2], i, next_batch=batches[(i + 1) % 2])
    net.eval()
    net.drop_emb = False
    nb, bb = batches[(i + 1) % 2], batches[i % 2]
    net.prefetch_bert(nb[0], nb[1], nb[2])

    def f():
        with torch.no_grad():
            sc, _ = net(bb[0], bb[1], bb[2])
        sc.sum().item()
    f() if "--default-stream" in sys.argv else tr.on_step_stream(f)


for i in range(4):
    one(i)
torch.cuda.synchronize()
lib.ruart_prof_enable(1)
N = 6
for i in range(N):
    one(i)
torch.cuda.synchronize()
M = 8192
b, e, f, n = (ctypes.c_float * M)(), (ctypes.c_float * M)(), (ctypes.c_double * M)(), ctypes.c_int(0)
hip.check(lib.ruart_prof_timeline(b, e, f, M, ctypes.byref(n)), "ruart_prof_timeline")
lib.ruart_prof_enable(0)
b, e, f = np.array(b[:n.value]), np.array(e[:n.value]), np.array(f[:n.value])
marks = [(t, int(-fl)) for t, fl in zip(b, f) if fl < 0]
gemm = [(bb, ee) for bb, ee, fl in zip(b, e, f) if fl > 0]
gfl = [fl for fl in f if fl > 0]
starts = [t for t, tag in marks if tag == 1]
print("step starts (ms):", " ".join("%.2f" % t for t in starts))
for k in range(2, len(starts) - 1):
    s0, s1 = starts[k], starts[k + 1]
    m = {tag: t - s0 for t, tag in marks if s0 <= t < s1}
    g = [(bb - s0, ee - s0) for bb, ee in gemm if s0 <= bb < s1]
    print("step %d: %.2f ms;  forward done %.2f, backward done %.2f, optimizer done %.2f" % (k, s1 - s0, m.get(2, -1), m.get(3, -1), m.get(4, -1)))
    if g:
        dur = np.array([ee - bb for bb, ee in g])
        print("   encoder GEMMs in this step: %d, first starts %.2f, last ends %.2f, busy %.2f ms (sum of launch-to-finish)" % (len(g), g[0][0], g[-1][1], dur.sum()))
        for lo, hi in ((0, m.get(2, 0)), (m.get(2, 0), m.get(3, 0)), (m.get(3, 0), s1 - s0)):
            sel = [(bb, ee) for bb, ee in g if lo <= bb < hi]
            if sel:
                print("     GEMMs started in [%.2f, %.2f): %3d, mean %.0f us" % (lo, hi, len(sel), 1e3 * np.mean([ee - bb for bb, ee in sel])))
# per projection (identified by its flops): launch-to-finish while the trunk is on the device against after it has finished
import collections
by = collections.defaultdict(lambda: [[], []])
for k in range(2, len(starts) - 1):
    s0, s1 = starts[k], starts[k + 1]
    m = {tag: t for t, tag in marks if s0 <= t < s1}
    for (bb, ee), fl in zip(gemm, gfl):
        if s0 <= bb < s1:
            by[round(fl / 1e9)][0 if bb < m.get(3, s1) else 1].append((ee - bb) * 1e3)
print("GFLOP per launch: mean us while the trunk runs / after it (counts)")
for fl, (a, c) in sorted(by.items()):
    print("  %6d: %6.0f / %6.0f   (%d / %d)" % (fl, np.mean(a) if a else 0, np.mean(c) if c else 0, len(a), len(c)))
tr.close(final=True)

#!/usr/bin/env python3
"""Why did the trunk replay 12.5 us per node slower as a torch-made graph when a raw hipGraph of C-ABI
launches replays at eager speed (profiles/r04_graph_node_cost.log)?  The SAME trunk forward (eval mode, no autograd, one stream) is
  (a) run eagerly,
  (b) captured by torch.cuda.graph and replayed,
  (c) captured by hipStreamBeginCapture on a stream of our own (torch only launches into it) and replayed with hipGraphLaunch,
device time between two events and host time of the enqueue for each; node types of both graphs are counted and their DOT dumps
written (gpurun_out/r05/graph_*.dot).

    python tools/r05_graph_probe.py [--train] [--streams]
"""
import argparse
import collections
import ctypes
import os
import re
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ruart_amd import hip, synth  # noqa: E402
from ruart_amd.arguments import default_opt  # noqa: E402
import ruart_amd.layers as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--train", action="store_true", help="training-mode forward (dropout masks drawn inside)")
ap.add_argument("--streams", action="store_true", help="three-stream trunk (default: one stream)")
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r05")
os.makedirs(OUT, exist_ok=True)

rt = ctypes.CDLL("libamdhip64.so")
P = ctypes.c_void_p


def ck(rc, what):
    if rc != 0:
        raise RuntimeError("%s -> hip error %d" % (what, rc))


dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64, ruart_streams=bool(a.streams))
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
net = tr.network
b = tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7, n_q=30, n_ocr=100, n_od=36))
grabbed = {}
orig = net._trunk_callable


def grab(*args):
    grabbed["args"] = tuple(t.detach().clone() for t in args)
    return orig(*args)


net._trunk_callable = grab
net.train(a.train)
net.drop_emb = a.train
with torch.no_grad():
    net(b[0], b[1], b[2])
torch.cuda.synchronize()
args = grabbed["args"]
trunk = net._trunk_module()
trunk.train(a.train)
if not a.train:
    L.set_dropout_prob(0.0)

my = P()
ck(rt.hipStreamCreateWithFlags(ctypes.byref(my), 1), "hipStreamCreateWithFlags")
st = torch.cuda.ExternalStream(my.value, device=dev)
ev = [P(), P()]
for e in ev:
    ck(rt.hipEventCreate(ctypes.byref(e)), "hipEventCreate")


def run():
    with torch.no_grad():
        return trunk(*args)


def timed(fn, reps=a.reps):
    dv, hs = [], []
    for _ in range(reps):
        ck(rt.hipStreamSynchronize(my), "sync")
        torch.cuda.synchronize()
        ck(rt.hipEventRecord(ev[0], my), "rec")
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        ck(rt.hipEventRecord(ev[1], my), "rec")
        ck(rt.hipEventSynchronize(ev[1]), "evsync")
        torch.cuda.synchronize()
        ms = ctypes.c_float()
        ck(rt.hipEventElapsedTime(ctypes.byref(ms), ev[0], ev[1]), "elapsed")
        dv.append(ms.value)
        hs.append((t1 - t0) * 1e3)
    dv.sort()
    hs.sort()
    return dv[len(dv) // 2], hs[len(hs) // 2]


def node_types(graph):
    nn = ctypes.c_size_t()
    ck(rt.hipGraphGetNodes(graph, None, ctypes.byref(nn)), "hipGraphGetNodes")
    nodes = (P * nn.value)()
    ck(rt.hipGraphGetNodes(graph, nodes, ctypes.byref(nn)), "hipGraphGetNodes")
    names = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "waitEvent", 7: "eventRecord"}
    cnt = collections.Counter()
    for n in nodes:
        t = ctypes.c_int()
        ck(rt.hipGraphNodeGetType(P(n), ctypes.byref(t)), "hipGraphNodeGetType")
        cnt[names.get(t.value, str(t.value))] += 1
    ne = ctypes.c_size_t()
    ck(rt.hipGraphGetEdges(graph, None, None, ctypes.byref(ne)), "hipGraphGetEdges")
    return nn.value, ne.value, dict(cnt)


with torch.cuda.stream(st):
    for _ in range(5):
        out_eager = run()
    torch.cuda.synchronize()
    de, he = timed(run)
    print("eager on our stream          : device %7.3f ms   host %7.3f ms" % (de, he), flush=True)

    # (c) raw capture on our own stream: relaxed mode (torch's allocator may query events)
    graph, gexec = P(), P()
    ck(rt.hipStreamBeginCapture(my, 2), "hipStreamBeginCapture")
    out_raw = run()
    ck(rt.hipStreamEndCapture(my, ctypes.byref(graph)), "hipStreamEndCapture")
    n, ne, kinds = node_types(graph)
    print("raw capture                  : %d nodes, %d edges, %s" % (n, ne, kinds), flush=True)
    rt.hipGraphDebugDotPrint(graph, os.path.join(OUT, "graph_raw.dot").encode(), 0)
    ck(rt.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, 0), "hipGraphInstantiate")
    ck(rt.hipGraphLaunch(gexec, my), "hipGraphLaunch")
    ck(rt.hipStreamSynchronize(my), "sync")
    dg, hg = timed(lambda: ck(rt.hipGraphLaunch(gexec, my), "hipGraphLaunch"))
    print("raw hipGraph replay          : device %7.3f ms   host %7.3f ms   (%+.2f us per node vs eager)" % (dg, hg, (dg - de) * 1e3 / max(n, 1)), flush=True)
    # replay on a torch pool stream too: does the stream the graph is launched into matter?
torch.cuda.synchronize()

# (b) torch.cuda.graph
g = torch.cuda.CUDAGraph()
if hasattr(g, "enable_debug_mode"):
    g.enable_debug_mode()
s2 = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s2):
    for _ in range(3):
        run()
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s2):
    out_torch = run()
torch.cuda.synchronize()
try:
    g.debug_dump(os.path.join(OUT, "graph_torch.dot"))
    txt = open(os.path.join(OUT, "graph_torch.dot")).read()
    print("torch graph DOT              : %d bytes, %d '->' edges, labels: %s" % (
        len(txt), txt.count("->"), dict(collections.Counter(re.findall(r'label="?\s*(\w+)', txt)).most_common(8))), flush=True)
except Exception as e:      # noqa: BLE001
    print("debug_dump failed:", e)


def replay_torch():
    g.replay()


# device time of a torch replay: events on the CURRENT torch stream
def timed_torch(fn, reps=a.reps):
    dv, hs = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        e1.record()
        e1.synchronize()
        dv.append(e0.elapsed_time(e1))
        hs.append((t1 - t0) * 1e3)
    dv.sort()
    hs.sort()
    return dv[len(dv) // 2], hs[len(hs) // 2]


with torch.cuda.stream(s2):
    replay_torch()
    dt_, ht_ = timed_torch(replay_torch)
    print("torch.cuda.graph replay      : device %7.3f ms   host %7.3f ms" % (dt_, ht_), flush=True)
    de2, he2 = timed_torch(run)
    print("eager on a torch pool stream : device %7.3f ms   host %7.3f ms" % (de2, he2), flush=True)
torch.cuda.synchronize()
print("max |raw - eager| = %.3g, max |torch graph - eager| = %.3g" % (float((out_raw - out_eager).abs().max()), float((out_torch - out_eager).abs().max())))
tr.close(final=True)

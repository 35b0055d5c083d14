#!/usr/bin/env python3
"""What does one kernel node of a hipGraph cost on this stack - measured WITHOUT torch in the capture path?

A chain of dependent launches of this repo's own C-ABI kernels (the trunk's sizes: split-bf16 products of (6400 x 500) . (500 x 250),
whole-tensor layer norms over (64, 100, 250) - 3 kernels per call -, and a tiny product as the 'trivial kernel' case) is
  (a) launched eagerly on one non-blocking stream created with hipStreamCreateWithFlags,
  (b) captured with hipStreamBeginCapture / hipStreamEndCapture on that stream, instantiated and replayed with hipGraphLaunch.
Device time between two events around the chain, and host time of the enqueue, both per kernel node.  torch only allocates the
buffers.  MI355X_MICROARCH.md (price list, row 'boundary') has 1.45 us per dependent boundary for eager == hipGraph; round 3 measured
12.5 us per node for graphs made by torch.cuda.make_graphed_callables.

    python tools/graph_node_cost.py [--calls 100] [--reps 20]
"""
import argparse
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--calls", type=int, default=100)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()

rt = ctypes.CDLL("libamdhip64.so")
P = ctypes.c_void_p


def ck(rc, what):
    if rc != 0:
        raise RuntimeError("%s -> hip error %d" % (what, rc))


lib = hip.load()
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)                                   # context
stream = P()
ck(rt.hipStreamCreateWithFlags(ctypes.byref(stream), 1), "hipStreamCreateWithFlags")     # hipStreamNonBlocking
ev = [P(), P()]
for e in ev:
    ck(rt.hipEventCreate(ctypes.byref(e)), "hipEventCreate")

g = torch.Generator().manual_seed(0)
M, K, N = 6400, 500, 250
x = [torch.randn(M, K, generator=g).to(dev), torch.randn(M, K, generator=g).to(dev)]
w = (torch.randn(K, N, generator=g) * 0.05).to(dev)
w2 = (torch.randn(N, K, generator=g) * 0.05).to(dev)
y = torch.empty(M, N, device=dev)
ln_ws = torch.empty(4096, device=dev)
ln_stats = torch.empty(8, device=dev)
tiny_a, tiny_b, tiny_c = torch.randn(16, 16, device=dev), torch.randn(16, 16, device=dev), torch.empty(16, 16, device=dev)
nul = P(None)


def mm(A, B, C, m, n, k, s):
    # C (m, n) = A (m, k) . B (k, n), both row-major: sam = k, sak = 1, sbk = n, sbn = 1; no workspace -> unsplit, one kernel
    ck(lib.ruart_gemm_x3(hip.ptr(A), k, 1, hip.ptr(B), n, 1, nul, nul, 0, 0, hip.ptr(C), n, m, n, k, nul, 0, nul, nul, 1.0, nul, 1, s),
       "ruart_gemm_x3")


def chain_trunk(s, calls):
    """dependent chain: x0 -> y = x0 . w -> whole-tensor LN(y) in place -> x1 = y . w2 -> ...; returns the kernel count"""
    n = 0
    for i in range(calls // 2):
        mm(x[i & 1], w, y, M, N, K, s)
        ck(lib.ruart_whole_ln_fwd(hip.ptr(y), hip.ptr(y), hip.ptr(ln_stats), hip.ptr(ln_ws), M * N, 1e-5, s), "ruart_whole_ln_fwd")
        mm(y, w2, x[(i & 1) ^ 1], M, K, N, s)
        n += 5
    return n


def chain_tiny(s, calls):
    for i in range(calls):
        mm(tiny_a if i & 1 == 0 else tiny_c, tiny_b, tiny_c if i & 1 == 0 else tiny_a, 16, 16, 16, s)
    return calls


def timed(fn):
    """(device us, host us) of one enqueue of fn between two events on `stream`"""
    ck(rt.hipStreamSynchronize(stream), "sync")
    ck(rt.hipEventRecord(ev[0], stream), "rec")
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    ck(rt.hipEventRecord(ev[1], stream), "rec")
    ck(rt.hipEventSynchronize(ev[1]), "evsync")
    ms = ctypes.c_float()
    ck(rt.hipEventElapsedTime(ctypes.byref(ms), ev[0], ev[1]), "elapsed")
    return ms.value * 1e3, (t1 - t0) * 1e6


def median(v):
    v = sorted(v)
    return v[len(v) // 2]


for name, chain, calls in (("trunk-sized kernels", chain_trunk, a.calls), ("trivial kernels", chain_tiny, a.calls)):
    nk = chain(stream, calls)                                 # warm-up (code objects, LDS attributes)
    ck(rt.hipStreamSynchronize(stream), "sync")
    eager = [timed(lambda: chain(stream, calls)) for _ in range(a.reps)]
    # capture the same chain: hipStreamCaptureModeThreadLocal = 1 (other threads of the process - none here - stay free)
    graph, gexec = P(), P()
    ck(rt.hipStreamBeginCapture(stream, 1), "hipStreamBeginCapture")
    chain(stream, calls)
    ck(rt.hipStreamEndCapture(stream, ctypes.byref(graph)), "hipStreamEndCapture")
    nn = ctypes.c_size_t()
    ck(rt.hipGraphGetNodes(graph, None, ctypes.byref(nn)), "hipGraphGetNodes")
    ck(rt.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, 0), "hipGraphInstantiate")
    ck(rt.hipGraphLaunch(gexec, stream), "hipGraphLaunch")    # first replay: upload
    ck(rt.hipStreamSynchronize(stream), "sync")
    replay = [timed(lambda: ck(rt.hipGraphLaunch(gexec, stream), "hipGraphLaunch")) for _ in range(a.reps)]
    de, he = median([t[0] for t in eager]), median([t[1] for t in eager])
    dg, hg = median([t[0] for t in replay]), median([t[1] for t in replay])
    print("%-20s %4d kernels (%d graph nodes)" % (name, nk, nn.value))
    print("   eager, one stream : device %9.1f us (%.2f us / kernel)   host enqueue %8.1f us (%.2f us / kernel)" % (de, de / nk, he, he / nk))
    print("   hipGraph replay   : device %9.1f us (%.2f us / kernel)   host enqueue %8.1f us" % (dg, dg / nk, hg))
    print("   graph - eager     : %+.2f us per kernel node" % ((dg - de) / nk), flush=True)
    ck(rt.hipGraphExecDestroy(gexec), "hipGraphExecDestroy")
    ck(rt.hipGraphDestroy(graph), "hipGraphDestroy")
ck(rt.hipStreamDestroy(stream), "hipStreamDestroy")

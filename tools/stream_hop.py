#!/usr/bin/env python3
"""What a cross-stream event wait costs on this runtime.  (1) ping-pong of small kernels between two streams, (2) a busy stream
that a third, otherwise idle stream waits on every K kernels, (3) the same with a second busy stream running beside it.
python tools/stream_hop.py"""
import time
import torch

dev = torch.device("cuda:0")
x = torch.zeros(1 << 20, device=dev)
big = torch.zeros(64 << 20, device=dev)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best * 1e3


A, B, C = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1)


def one_stream(n=400):
    with torch.cuda.stream(A):
        for _ in range(n):
            x.add_(1.0)


def ping_pong(n=400):
    for i in range(n // 2):
        with torch.cuda.stream(A):
            x.add_(1.0)
        B.wait_stream(A)
        with torch.cuda.stream(B):
            x.add_(1.0)
        A.wait_stream(B)


def busy(n=1000, every=0, second=False, waiter=C, t=big):
    for i in range(n):
        with torch.cuda.stream(A):
            t.add_(1.0)
        if second:
            with torch.cuda.stream(B):
                x.add_(1.0)
        if every and i % every == every - 1:
            waiter.wait_stream(A)


print("400 small kernels, one stream      %.3f ms" % timed(one_stream))
print("400 small kernels, ping-pong       %.3f ms  (400 event hops)" % timed(ping_pong))
for second in (False, True):
    base = timed(lambda: busy(second=second))
    for every in (100, 20):
        for name, w in (("idle high-priority stream", C), ("the second busy stream", B)):
            t = timed(lambda: busy(every=every, second=second, waiter=w))
            print("1000 x 64 MiB add on A%s; %s waits on A every %3d: %.3f ms vs %.3f ms  (%.1f us per wait)"
                  % (" + small kernels on B" if second else "", name, every, t, base, (t - base) * 1e3 / (1000 // every)))


# (4) does a barrier packet that sits UNSATISFIED at the head of an otherwise idle queue slow the dispatch of small kernels elsewhere?
def pending(n_wait, prio, n=3000):
    D = torch.cuda.Stream(dev)
    waiters = [torch.cuda.Stream(dev, priority=prio) for _ in range(n_wait)]
    torch.cuda.synchronize()
    with torch.cuda.stream(D):
        for _ in range(400):
            big.add_(1.0)
        e = D.record_event()
    for w in waiters:
        w.wait_event(e)
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(A):
        s.record()
        for _ in range(n):
            x.add_(1.0)
        t.record()
    torch.cuda.synchronize()
    return s.elapsed_time(t)


for prio in (0, -1):
    for n_wait in (0, 1, 2, 4):
        pending(n_wait, prio)
        print("3000 small kernels on A beside a 30 ms stream, %d idle streams (priority %d) blocked on its end: %.3f ms"
              % (n_wait, prio, min(pending(n_wait, prio) for _ in range(3))))

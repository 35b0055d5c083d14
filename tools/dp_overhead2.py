#!/usr/bin/env python3
"""Pipelined training steps under a world-size-1 RCCL group with the host time of GradSync's parts accumulated, and the same
with parts of it switched off (MODE = full | nolaunch | noavg).   python tools/dp_overhead2.py MODE"""
import os, sys, time
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import dp, synth
from ruart_amd.arguments import default_opt

mode = sys.argv[1] if len(sys.argv) > 1 else "full"
dev = torch.device("cuda:0")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29542", HSA_ENABLE_IPC_MODE_LEGACY="0")
dp.init_process_group(dev, "nccl", rank=0, world_size=1)
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev, process_group=dist.group.WORLD)
gs = tr.grad_sync
acc = {"launch": 0.0, "avg": 0.0}
orig_launch, orig_avg = gs._launch, gs.average_gradients


def timed_launch(bi):
    t = time.perf_counter()
    if mode == "nolaunch":
        class W:
            def wait(self): pass
        gs._work[bi] = W()
    else:
        orig_launch(bi)
    acc["launch"] += time.perf_counter() - t


def timed_avg():
    t = time.perf_counter()
    if mode == "noavg":
        gs._reset()
    else:
        orig_avg()
    acc["avg"] += time.perf_counter() - t


flags = set(mode.split("+"))
import ctypes
_hip = ctypes.CDLL("libamdhip64.so")
_evs = []
_norm = {}


def raw_wait(dst, src, fl=0x2 | 0x20000000):        # hipEventDisableTiming | hipEventDisableSystemFence
    if len(_evs) < 64:
        ev = ctypes.c_void_p()
        assert _hip.hipEventCreateWithFlags(ctypes.byref(ev), ctypes.c_uint(0x2 if "rawplain" in flags else fl)) == 0
        _evs.append(ev)
    _evs.append(_evs.pop(0))
    ev = _evs[-1]
    assert _hip.hipEventRecord(ev, ctypes.c_void_p(src.cuda_stream)) == 0
    assert _hip.hipStreamWaitEvent(ctypes.c_void_p(dst.cuda_stream), ev, ctypes.c_uint(0)) == 0



def custom_launch(bi):
    """GradSync._launch with parts removable: nowait, norecord, nocopy, nocoll, samestream"""
    flat = gs._flat[bi]
    grads = gs._grads(bi)
    cur = torch.cuda.current_stream(flat.device)
    comm = cur if "samestream" in flags else gs._comm_stream(flat.device)
    if "commnormal" in flags:
        if "c" not in _norm:
            _norm["c"] = torch.cuda.Stream(device=flat.device)
        comm = _norm["c"]
    if "nowait" not in flags:
        for s in gs._streams[bi] | {cur}:
            if s != comm:
                acc["events"] = acc.get("events", 0) + 1
                if "recordonly" in flags:
                    s.record_event()
                elif "rawevent" in flags:
                    raw_wait(comm, s)
                elif "curonly" in flags:
                    if s == cur:
                        comm.wait_stream(s)
                else:
                    comm.wait_stream(s)
    if "norecord" not in flags and comm != cur:
        for g in grads:
            g.record_stream(comm)
    with torch.cuda.stream(comm):
        if "nocopy" not in flags:
            torch._foreach_copy_(gs._slices[bi], grads)
        if "nocoll" in flags:
            class W:
                def wait(self): pass
            gs._work[bi] = W()
        else:
            gs._work[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=gs.group, async_op=True)


def custom_avg():
    while gs._next < len(gs.buckets):
        gs._launch(gs._next)
        gs._next += 1
    for bi in range(len(gs.buckets)):
        if "nowork" not in flags:
            gs._work[bi].wait()
        if "norepoint" not in flags:
            for sl, (_, p, rows) in zip(gs._slices[bi], gs.buckets[bi]):
                if rows is None:
                    p.grad = sl
                else:
                    p.grad[:rows].copy_(sl)
    gs._reset()


def post_avg():
    """everything after backward(), on the step stream, synchronous collectives"""
    for bi in range(len(gs.buckets)):
        torch._foreach_copy_(gs._slices[bi], gs._grads(bi))
        if "onebucket" not in flags:
            dist.all_reduce(gs._flat[bi], op=dist.ReduceOp.SUM, group=gs.group, async_op="async" in flags)
        for sl, (_, p, rows) in zip(gs._slices[bi], gs.buckets[bi]):
            if rows is None:
                p.grad = sl
            else:
                p.grad[:rows].copy_(sl)
    gs._reset()


if "post" in flags:
    gs._launch_ready = lambda: None
    orig_avg = post_avg
if flags & {"nowait", "norecord", "samestream", "nowork", "norepoint", "custom"}:
    orig_launch, orig_avg = custom_launch, custom_avg
gs._launch, gs.average_gradients = timed_launch, timed_avg
if mode == "nocoll_":
    class _W:
        def wait(self): pass
    dist.all_reduce = lambda *a, **k: _W()
if mode == "nocopy_":
    torch._foreach_copy_ = lambda *a, **k: None
batches = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
for i in range(5):
    tr.update(batches[i % 2], i, next_batch=batches[(i + 1) % 2])
torch.cuda.synchronize()
acc = {"launch": 0.0, "avg": 0.0}
N = 20
t0 = time.perf_counter()
for i in range(N):
    tr.update(batches[i % 2], i, next_batch=batches[(i + 1) % 2])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N * 1e3
print("%-9s %.2f ms per step; host time per step: _launch %.2f ms (all buckets), average_gradients %.2f ms (incl. late launches)"
      % (mode, dt, acc["launch"] / N * 1e3, acc["avg"] / N * 1e3), "events/step", acc.get("events", 0) / N, "buckets", len(gs.buckets))
tr.close(final=True)
dist.destroy_process_group()

#!/usr/bin/env python3
"""Which source lines issue the trunk's small torch launches?  One training step under a TorchDispatchMode: every aten op that is
not a pure view is counted against the innermost ruart_amd frame on the Python stack (forward: the model's own line; backward of a
custom Function: its line in ops.py; native autograd nodes - fan-in adds, CatBackward, SliceBackward - have no Python frame and are
listed as "(autograd engine)" with the op and its shapes).
    python tools/op_sites2.py [--top 80]
"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ruart_amd import synth  # noqa: E402
from ruart_amd.arguments import default_opt  # noqa: E402

VIEWS = {"as_strided", "view", "_unsafe_view", "slice", "t", "transpose", "reshape", "expand", "unsqueeze", "squeeze", "select", "narrow",
         "detach", "alias", "permute", "empty", "empty_like", "empty_strided", "split", "split_with_sizes", "unbind", "_reshape_alias",
         "new_empty", "new_empty_strided", "lift_fresh", "record_stream", "is_pinned", "_local_scalar_dense", "sym_size", "sym_stride",
         "sym_numel", "unfold", "view_as", "chunk", "resize_", "set_", "_to_copy_view", "result_type", "stride", "size", "is_same_size"}

ap = argparse.ArgumentParser()
ap.add_argument("--top", type=int, default=80)
a = ap.parse_args()
dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
batches = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
for i in range(4):
    tr.update(batches[i % 2], i)
torch.cuda.synchronize()

sites = collections.Counter()
shapes = {}


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name not in VIEWS:
            fr = "(autograd engine)"
            for f in reversed(traceback.extract_stack(limit=40)):
                if "/ruart_amd/" in f.filename and not f.filename.endswith("hip.py"):
                    fr = "%s:%d" % (os.path.basename(f.filename), f.lineno)
                    break
            shp = tuple(tuple(x.shape) for x in args if isinstance(x, torch.Tensor))[:2]
            sites[(fr, name)] += 1
            shapes.setdefault((fr, name), collections.Counter())[shp] += 1
        return func(*args, **(kwargs or {}))


# the autograd engine's worker thread does not inherit a Python dispatch mode pushed on this thread: run backward on this thread
torch.autograd.set_multithreading_enabled(False)
with Spy():
    tr.update(batches[0], 4)
torch.cuda.synchronize()
print("%5s  %-28s %-28s %s" % ("calls", "site", "op", "most common shapes"))
for (fr, name), n in sites.most_common(a.top):
    sh = "; ".join("%s x%d" % (k, v) for k, v in shapes[(fr, name)].most_common(40 if n >= 20 else 2))
    print("%5d  %-28s %-28s %s" % (n, fr, name, sh if n >= 20 else sh[:150]))
print("total non-view aten calls:", sum(sites.values()))

#!/usr/bin/env python3
"""Where do the trunk's small launches come from?  One training step under torch.profiler with Python stacks:
  * forward: every aten op that launches a kernel, grouped by the innermost ruart_amd source line that called it;
  * backward: kernels launched under each autograd node type (``evaluate_function: XBackward``), i.e. what autograd itself adds
    (gradient fan-in adds, CatBackward slices made contiguous, views).
    python tools/op_sites.py [--top 60]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ruart_amd import synth  # noqa: E402
from ruart_amd.arguments import default_opt  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--top", type=int, default=60)
a = ap.parse_args()
dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
batches = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
for i in range(4):
    tr.update(batches[i % 2], i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.update(batches[0], 4)
    torch.cuda.synchronize()

events = prof.events()
# kernels launched per CPU op: walk the event tree; attribute each device kernel to its innermost CPU op and that op's ancestors
by_id = {e.id: e for e in events}
launch_parent = collections.Counter()
site = collections.Counter()
site_dev = collections.Counter()
bwd_node = collections.Counter()
bwd_dev = collections.Counter()


def user_frame(e):
    st = getattr(e, "stack", None) or []
    for fr in st:
        if "ruart_amd/" in fr and "hip.py" not in fr:
            return fr.split("ruart_amd/")[-1][:70]
    return None


for e in events:
    if e.device_type != torch.autograd.DeviceType.CPU:
        continue
    kern = [k for k in (e.kernels or [])]
    if not kern or (e.cpu_children and any(c.kernels for c in e.cpu_children)):
        continue                                  # only the innermost op that owns the launch
    dur = sum(k.duration for k in kern)
    # climb to find an autograd node or a user frame
    p, node, fr = e, None, user_frame(e)
    while p is not None:
        if p.name.startswith("autograd::engine::evaluate_function:"):
            node = p.name.split(":", 3)[-1].strip()
            break
        if fr is None:
            fr = user_frame(p)
        p = p.cpu_parent
    if node is not None:
        bwd_node[(node, e.name)] += len(kern)
        bwd_dev[(node, e.name)] += dur
    else:
        site[(fr or "?", e.name)] += len(kern)
        site_dev[(fr or "?", e.name)] += dur

print("== forward (and optimizer): launches by source line, op ==")
for (fr, name), n in site.most_common(a.top):
    print("%4d  %8.1f us  %-60s %s" % (n, site_dev[(fr, name)], fr, name[:40]))
print("\n== backward: launches by autograd node, op ==")
for (node, name), n in bwd_node.most_common(a.top):
    print("%4d  %8.1f us  %-44s %s" % (n, bwd_dev[(node, name)], node[:44], name[:50]))
print("\ntotal launches: forward-side %d, backward-side %d" % (sum(site.values()), sum(bwd_node.values())))

#!/usr/bin/env python3
"""Join the launch sequence of tools/gemm_corr_order_sweep.py with a rocprofv3 --pmc FETCH_SIZE counter_collection.csv of the same
command: L2-miss read bytes per launch (x2, the gfx950 correction of MI355X_MICROARCH.md) per (tile order, shape), beside the
operand bytes a launch needs once.   python tools/gemm_corr_order_join.py <seq.json> <counter_collection.csv>"""
import collections
import csv
import json
import sys

meta = json.load(open(sys.argv[1]))
M, seq = meta["rows"], meta["seq"]
rows = [r for r in csv.DictReader(open(sys.argv[2])) if r["Counter_Name"] == "FETCH_SIZE" and "gemm_16c_nt_256p8" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
assert len(rows) == len(seq), (len(rows), len(seq))
acc = collections.defaultdict(list)
for (order, name), r in zip(seq, rows):
    acc[(order, name)].append(2 * float(r["Counter_Value"]) * 1024 / 1e6)
NK = {"qkv": (2304, 768), "ao": (768, 768), "ff1": (3072, 768), "ff2": (768, 3072)}
need = {k: (M * K * 4 + N * K * 4 + (M * N * 4 if k in ("ao", "ff2") else 0)) / 1e6 for k, (N, K) in NK.items()}
orders = sorted({o for o, _ in acc})
print("L2-miss read MB per launch (FETCH_SIZE x2); operand bytes needed once: " + ", ".join("%s %.0f MB" % (k, v) for k, v in need.items()))
for o in orders:
    print("order %2d: " % o + "  ".join("%s %7.1f MB (%.2fx)" % (k, sum(acc[(o, k)][1:]) / max(1, len(acc[(o, k)]) - 1),
                                                                  sum(acc[(o, k)][1:]) / max(1, len(acc[(o, k)]) - 1) / need[k]) for k in NK))

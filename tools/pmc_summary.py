#!/usr/bin/env python3
"""Average the counters of a rocprofv3 --pmc run per kernel name: python tools/pmc_summary.py <dir> [name-substring]"""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        v = v[2:] if len(v) > 3 else v
        print("   %-36s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))

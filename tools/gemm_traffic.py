#!/usr/bin/env python3
"""HBM-side traffic of the encoder GEMM from two rocprofv3 PMC passes over tools/gemm_bench.py (FETCH_SIZE and WRITE_SIZE are
collected separately: they do not fit one pass).  Counter unit = KiB; FETCH_SIZE is doubled (gfx950 tallies a wide coalesced
128-B request as 64 B, MI355X_MICROARCH.md); WRITE_SIZE is exact.
   python tools/gemm_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel substring> <out.json>"""
import csv, json, sys

fetch_csv, write_csv, kname, out = sys.argv[1:5]
M = 43008
shapes = [("qkv_N2304_K768", 2304, 768, 2, 0), ("ao_N768_K768", 768, 768, 4, 2), ("ff1_N3072_K768", 3072, 768, 2, 0), ("ff2_N768_K3072", 768, 3072, 4, 2)]


def per_shape(path, counter):
    runs, last = [], None                      # consecutive launches of one (kernel, grid) = one shape of gemm_bench.py
    rows = [r for r in csv.DictReader(open(path)) if kname in r["Kernel_Name"] and r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        key = (r["Kernel_Name"], r["Grid_Size"])
        if key != last:
            runs.append([])
            last = key
        runs[-1].append(float(r["Counter_Value"]))
    return [sum(v[1:]) / max(len(v) - 1, 1) for v in runs]     # first launch of a shape: cold caches


f, w = per_shape(fetch_csv, "FETCH_SIZE"), per_shape(write_csv, "WRITE_SIZE")
assert len(f) == len(w) == 4, (len(f), len(w))
res, tot, alg = {}, 0.0, 0.0
for (name, N, K, out_b, res_b), fk, wk in zip(shapes, f, w):
    b = fk * 1024 * 2 + wk * 1024
    a = M * K * 2 + N * K * 2 + M * N * out_b + M * N * res_b + N * 4
    res[name] = {"FETCH_SIZE": round(fk, 1), "WRITE_SIZE": round(wk, 1), "hbm_bytes_corrected": int(b), "algorithmic_bytes": int(a)}
    tot += b
    alg += a
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/gemm_bench.py --variants 5 --orders 8, "
                     "M=43008 rows (bench token count padded to 256), f16, kernel " + kname,
           "unit_note": "counter unit = KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced stream); "
                        "WRITE_SIZE exact", "per_launch_KiB": res, "avg_bytes_per_launch": int(tot / 4),
           "algorithmic_bytes_per_launch_avg": int(alg / 4), "traffic_over_algorithmic": round(tot / alg, 2)}, open(out, "w"), indent=1)
print(open(out).read())

#!/usr/bin/env python3
"""`roofline.traffic` of bench.py: HBM-side bytes per launch of the encoder GEMM kernels, from the per-kernel table that
tools/pmc_hbm_table.py builds out of two rocprofv3 PMC passes over the whole bench step (FETCH_SIZE x2 - the gfx950 correction of
MI355X_MICROARCH.md -, WRITE_SIZE exact).   python tools/gemm_traffic2.py <pmc_hbm_per_kernel.csv> <out.json> [rows=43008]"""
import csv
import json
import sys

table, out = sys.argv[1:3]
M = int(sys.argv[3]) if len(sys.argv) > 3 else 43008
H, I = 768, 3072
# algorithmic bytes per launch (operands once, output once; fp16c: f16 + two e4m3 bytes per operand element, fp32 out / residual)
alg = {
    "gemm_16c_nt_256p8": {"<0>": M * H * 4 + 3 * H * H * 4 + M * 3 * H * 4,                      # QKV: fp32 out
                          "<1>": (M * H * 4 + H * H * 4 + 2 * M * H * 4 + M * I * 4 + H * I * 4 + 2 * M * H * 4) / 2,   # AO, FF2 (avg)
                          "<2>": M * H * 4 + I * H * 4 + M * I * 4},                             # FF1: split out (4 B per element)
}
res = {}
rows = list(csv.DictReader(open(table)))
for kern in ("gemm_16c_nt_256p8", "gemm_16_nt_256p8"):
    tot_b, tot_n, tot_alg, per = 0.0, 0, 0.0, {}
    for r in rows:
        if kern not in r["kernel"]:
            continue
        n = int(r["dispatches"])
        b = (float(r["fetch_MB_per_dispatch(x2)"]) + float(r["write_MB_per_dispatch"])) * 1e6
        per[r["kernel"][:60]] = {"dispatches": n, "hbm_bytes_per_launch": int(b), "avg_us": float(r["avg_us"] or 0)}
        tot_b += b * n
        tot_n += n
        for tag, a in alg.get(kern, {}).items():
            if tag in r["kernel"]:
                per[r["kernel"][:60]]["algorithmic_bytes_per_launch"] = int(a)
                tot_alg += a * n
    if tot_n:
        res[kern] = {"avg_bytes_per_launch": int(tot_b / tot_n), "launches_seen": tot_n, "per_instantiation": per}
        if tot_alg:
            res[kern]["algorithmic_bytes_per_launch_avg"] = int(tot_alg / tot_n)
            res[kern]["traffic_over_algorithmic"] = round(tot_b / tot_alg, 2)
res["source"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --no-prefetch, joined by "
                 "tools/pmc_hbm_table.py; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950), WRITE_SIZE exact; M = %d rows" % M)
json.dump(res, open(out, "w"), indent=1)
print(open(out).read())

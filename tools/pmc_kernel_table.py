#!/usr/bin/env python3
"""Join any number of rocprofv3 --pmc passes (counter_collection.csv files or directories holding them) into ONE per-kernel table:
mean of every counter per dispatch, kernels ranked by dispatch count x SQ_BUSY_CYCLES (or by count).
    python tools/pmc_kernel_table.py out.csv pass1_dir pass2_dir ... [--match substr,substr]
Derived columns (when their inputs were collected):
    clock_GHz        = GRBM_GUI_ACTIVE / 8 / duration                      (MI355X_MICROARCH.md, DVFS give-back)
    mfma_busy        = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)
    wait_any / wait_inst / active = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
    l2_hit           = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
"""
import collections, csv, glob, os, sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
match = None
for a in sys.argv[1:]:
    if a.startswith("--match"):
        match = a.split("=", 1)[1].split(",") if "=" in a else None
out_path, srcs = args[0], args[1:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for s in srcs:
    files = [s] if s.endswith(".csv") else glob.glob(os.path.join(s, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
counters = sorted({c for k in acc for c in acc[k]})
rows = []
for k, cs in acc.items():
    if match and not any(m in k for m in match):
        continue
    n = max(len(v) for v in cs.values())
    mean = {c: (sum(v) / len(v)) for c, v in cs.items()}
    d = sum(dur[k]) / len(dur[k]) if dur.get(k) else float("nan")
    der = {}
    g = mean.get("GRBM_GUI_ACTIVE")
    if g and d == d and d > 0:
        der["clock_GHz"] = g / 8.0 / (d * 1e3)
    if g and "SQ_VALU_MFMA_BUSY_CYCLES" in mean:
        der["mfma_busy"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (g / 8.0 * 256 * 4)
    wc = mean.get("SQ_WAVE_CYCLES")
    if wc:
        for nm, c in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst", "SQ_WAIT_INST_ANY"), ("active", "SQ_ACTIVE_INST_ANY"), ("wait_inst_lds", "SQ_WAIT_INST_LDS")):
            if c in mean:
                der[nm] = mean[c] / wc
    if "TCC_HIT_sum" in mean and "TCC_MISS_sum" in mean and mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"] > 0:
        der["l2_hit"] = mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])
    rows.append((-(n * (d if d == d else 1.0)), k, n, d, mean, der))
rows.sort()
dcols = ["clock_GHz", "mfma_busy", "wait_any", "wait_inst", "wait_inst_lds", "active", "l2_hit"]
with open(out_path, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches_per_pass", "avg_us(profiled)"] + dcols + counters)
    for _, k, n, d, mean, der in rows[:60]:
        w.writerow([k[:100], n, "%.1f" % d] + ["%.3f" % der[c] if c in der else "" for c in dcols] + ["%.0f" % mean[c] if c in mean else "" for c in counters])
print("wrote", out_path, len(rows), "kernels")

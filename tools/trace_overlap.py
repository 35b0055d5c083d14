"""Summarise a rocprofv3 kernel trace of bench.py: per training step, the span of encoder kernels vs trunk kernels and
how much they overlap.  usage: python tools/trace_overlap.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
t0 = rows[0][0]
enc = ("gemm_16_nt", "gemm_16c_nt", "attn_flash", "attn_varlen", "rows_layernorm", "embed_ln")
# steps are delimited by embed_ln launches (one per encoder pass)
marks = [s for s, e, n, q, st in rows if "embed_ln" in n]
print("encoder passes:", len(marks))
byq = defaultdict(lambda: [0, 0.0])
for s, e, n, q, st in rows:
    byq[(q, st)][0] += 1
    byq[(q, st)][1] += (e - s) / 1e6
for k, v in sorted(byq.items()):
    print("queue/stream", k, "kernels", v[0], "busy ms %.2f" % v[1])
for i in range(len(marks) - 1):
    lo, hi = marks[i], marks[i + 1]
    seg = [r for r in rows if lo <= r[0] < hi]
    e_busy = sum(e - s for s, e, n, q, st in seg if any(t in n for t in enc)) / 1e6
    t_busy = sum(e - s for s, e, n, q, st in seg if not any(t in n for t in enc)) / 1e6
    e_seg = [r for r in seg if any(t in r[2] for t in enc)]
    e_span = (max(r[1] for r in e_seg) - min(r[0] for r in e_seg)) / 1e6
    def union_of(ev):
        ev = sorted(ev)
        u, cur_s, cur_e = 0, ev[0][0], ev[0][1]
        for s, e in ev[1:]:
            if s > cur_e:
                u += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        return u + cur_e - cur_s
    union = union_of([(s, e) for s, e, *_ in seg])
    t_union = union_of([(s, e) for s, e, n, *_ in seg if not any(t in n for t in enc)])
    e0, e1 = min(r[0] for r in e_seg), max(r[1] for r in e_seg)
    t_in = [(s, e) for s, e, n, *_ in seg if not any(t in n for t in enc) and s >= e0 and e <= e1]
    print("   trunk union %.2f ms; trunk kernels inside the encoder span: %d, their union %.2f ms, mean duration %.1f us" %
          (t_union / 1e6, len(t_in), (union_of(t_in) if t_in else 0) / 1e6, (sum(e - s for s, e in t_in) / max(len(t_in), 1)) / 1e3))
    print("pass %d: period %.2f ms  encoder busy %.2f (span %.2f)  trunk busy %.2f  gpu-any-busy %.2f  idle %.2f" %
          (i, (hi - lo) / 1e6, e_busy, e_span, t_busy, union / 1e6, (hi - lo - union) / 1e6))

# which trunk kernels stretch while an encoder pass is running?
import bisect
e_int = sorted((s, e) for s, e, n, *_ in rows if any(t in n for t in enc))
e_starts = [s for s, _ in e_int]
def overlaps_encoder(s, e):
    i = bisect.bisect_right(e_starts, e) - 1
    while i >= 0 and e_int[i][0] > s - 2_000_000:
        if e_int[i][1] > s and e_int[i][0] < e:
            return True
        i -= 1
    return False
stat = defaultdict(lambda: [0, 0.0, 0, 0.0])
for s, e, n, q, st in rows[len(rows) // 3:]:
    if any(t in n for t in enc):
        continue
    k = n[:60]
    if overlaps_encoder(s, e):
        stat[k][0] += 1; stat[k][1] += (e - s) / 1e3
    else:
        stat[k][2] += 1; stat[k][3] += (e - s) / 1e3
print("\n%-60s %8s %10s %8s %10s %8s" % ("trunk kernel", "n_in", "us_in", "n_out", "us_out", "extra_ms"))
out = []
for k, (a, ta, b, tb) in stat.items():
    if a and b:
        out.append((a * (ta / a - tb / b) / 1e3, k, a, ta / a, b, tb / b))
for extra, k, a, ma, b, mb in sorted(out, reverse=True)[:18]:
    print("%-60s %8d %10.1f %8d %10.1f %8.2f" % (k, a, ma, b, mb, extra))
print("total extra ms over the analysed window: %.2f" % sum(o[0] for o in out))

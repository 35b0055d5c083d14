#!/usr/bin/env python3
"""Per-wave stamps of the two tile forms of the fp16c projections (diagnostic build: tools/build_variant.sh stamps -DRUART_P8_STAMPS, run with
RUART_HIP_LIB=build/libruart_hip_stamps.so): fill / K loop / epilogue per workgroup and, per CU, how many workgroups were resident over time.
    python tools/r06_dual_stamps.py [--rows 42752]"""
import argparse, ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip
from ruart_amd.bert import split_f16c
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=42752)
a = ap.parse_args()
lib = hip.load(); d = torch.device("cuda:0")
lib.ruart_gemm_set_stamps.argtypes = [ctypes.c_void_p]; lib.ruart_gemm_set_stamps.restype = ctypes.c_int
sa = hip.f16c_shifts()
M = (a.rows + 255) // 256 * 256
g = torch.Generator().manual_seed(0)
for name, N, K, act in [("qkv", 2304, 768, hip.ACT_NONE), ("ff1", 3072, 768, hip.ACT_GELU)]:
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.03
    A16, A8 = [t.to(d) for t in split_f16c(A)]
    hi = W.half().float()
    W16 = W.half().to(d)
    W8 = torch.cat([hi * 2.0 ** sa[2], (W - hi) * 2.0 ** sa[3]], 1).clamp_(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).to(d)
    bias = torch.randn(N, generator=g).to(d)
    gelu = act == hip.ACT_GELU
    C = torch.empty(M, N, dtype=torch.float16 if gelu else torch.float32, device=d)
    C8 = torch.empty(M, 2 * N, dtype=torch.uint8, device=d) if gelu else None
    for dual in (0, 1):
        nw = 4 if dual else 8
        nwg = (M // 256) * (N // (128 if dual else 256))
        st = torch.zeros(nwg * nw * 8, dtype=torch.int64, device=d)
        lib.ruart_gemm_16c_set_dual(dual)
        def run():
            assert lib.ruart_gemm_16c_nt(hip.ptr(A16), hip.ptr(A8), K, hip.ptr(W16), hip.ptr(W8), K, hip.ptr(bias), None, 0, hip.ptr(C), N,
                                         hip.ptr(C8), M, N, K, act, hip.stream_ptr()) == 0
        lib.ruart_gemm_set_stamps(None)
        for _ in range(3): run()
        lib.ruart_gemm_set_stamps(st.data_ptr()); run(); torch.cuda.synchronize(); lib.ruart_gemm_set_stamps(None)
        rw = st.cpu().numpy().reshape(nwg, nw, 8)
        tw = rw[:, :, :5].astype(np.float64) * 0.01
        t = np.concatenate([tw[:, :, :1].min(1), tw[:, :, 1:].max(1)], 1)
        fill, f16, f8, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]
        nt = K // 64
        tile = t[:, 4] - t[:, 0]
        print("%-4s %s: workgroups %4d | fill %5.2f | f16 run %6.2f us = %.3f/K-tile | fp8 run %6.2f = %.3f/K-tile | epilogue+drain %5.2f | workgroup %6.2f (p10 %.1f p90 %.1f) | kernel %7.1f us"
              % (name, "256x128 dual" if dual else "256x256     ", nwg, np.median(fill), np.median(f16), np.median(f16) / nt, np.median(f8), np.median(f8) / nt,
                 np.median(epi), np.median(tile), np.percentile(tile, 10), np.percentile(tile, 90), t[:, 4].max() - t[:, 0].min()))
        raw = rw[:, 0, :]
        xcc, hw = raw[:, 6] & 0xf, raw[:, 7]
        key = xcc * 10000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 20 + ((hw >> 8) & 0xf)
        # residency per CU over the kernel: time with 0 / 1 / 2 workgroups between the first start and the last end on that CU; and the same for
        # workgroups INSIDE their K loop (stamps 1..3)
        res, inl = np.zeros(4), np.zeros(4)
        for k in np.unique(key):
            sel = np.where(key == k)[0]
            for lo, hi_, acc in ((t[sel, 0], t[sel, 4], res), (t[sel, 1], t[sel, 3], inl)):
                ev = sorted([(x, 1) for x in lo] + [(x, -1) for x in hi_])
                n, prev = 0, t[sel, 0].min()
                for x, dlt in ev:
                    acc[min(n, 3)] += x - prev
                    prev, n = x, n + dlt
                acc[0] += t[sel, 4].max() - prev
        print("     %d CUs seen; CU time with 0 / 1 / 2 / 3+ workgroups resident: %s %% ; with 0 / 1 / 2 / 3+ inside their K loop: %s %%"
              % (len(np.unique(key)), " / ".join("%.1f" % (100 * x / res.sum()) for x in res), " / ".join("%.1f" % (100 * x / inl.sum()) for x in inl)))
        cyc = rw[:, 0, 5].astype(np.float64)
        loop_us = tw[:, 0, 3] - tw[:, 0, 1]
        print("     K loop of wave 0: %.0f shader cycles per K-tile, in-loop clock %.3f GHz" % (np.median(cyc) / (2 * nt), np.median(cyc / loop_us) * 1e-3))
        order = np.argsort(t[:, 0])
        for r0 in range(0, nwg, 512 if dual else 256):
            sel = order[r0:r0 + (512 if dual else 256)]
            print("     workgroups %4d-%4d by start: start spread %6.1f us | fill %5.2f | f16 %.3f/K-tile | fp8 %.3f/K-tile | epi %5.2f | workgroup %6.2f"
                  % (r0, r0 + len(sel) - 1, t[sel, 0].max() - t[sel, 0].min(), np.median(fill[sel]), np.median(f16[sel]) / nt, np.median(f8[sel]) / nt,
                     np.median(epi[sel]), np.median(tile[sel])))
lib.ruart_gemm_16c_set_dual(0)

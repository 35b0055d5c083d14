#!/usr/bin/env python3
"""In-kernel phase times of the fp16c encoder GEMM (diagnostic build: tools/build_variant.sh stamps -DRUART_P8_STAMPS, run with
RUART_HIP_LIB=build/libruart_hip_stamps.so): per workgroup s_memrealtime (100 MHz) at start / pipeline filled / f16 run done /
fp8 run done / stores drained, plus the XCC id - per shape: fill, per-K-tile time of both runs, epilogue, and how far the
workgroups of one XCD that share an operand panel drift apart.   python tools/r06_corr_stamps.py [--rows 42752] [--cus N]"""
import argparse, ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip
from ruart_amd.bert import split_f16c
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=42752)
ap.add_argument("--dump", default="")
a = ap.parse_args()
lib = hip.load(); d = torch.device("cuda:0")
lib.ruart_gemm_set_stamps.argtypes = [ctypes.c_void_p]; lib.ruart_gemm_set_stamps.restype = ctypes.c_int
sa = hip.f16c_shifts()
M = (a.rows + 255) // 256 * 256
g = torch.Generator().manual_seed(0)
for name, N, K, act, res in [("qkv", 2304, 768, hip.ACT_NONE, False), ("ao", 768, 768, hip.ACT_NONE, True), ("ff1", 3072, 768, hip.ACT_GELU, False),
                             ("ff2", 768, 3072, hip.ACT_NONE, True)]:
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.03
    A16, A8 = [t.to(d) for t in split_f16c(A)]
    hi = W.half().float()
    W16 = W.half().to(d)
    W8 = torch.cat([hi * 2.0 ** sa[2], (W - hi) * 2.0 ** sa[3]], 1).clamp_(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).to(d)
    bias = torch.randn(N, generator=g).to(d)
    R32 = torch.randn(M, N, generator=g).to(d) if res else None
    gelu = act == hip.ACT_GELU
    C = torch.empty(M, N, dtype=torch.float16 if gelu else torch.float32, device=d)
    C8 = torch.empty(M, 2 * N, dtype=torch.uint8, device=d) if gelu else None
    ntiles = (M // 256) * (N // 256)
    st = torch.zeros(ntiles * 64, dtype=torch.int64, device=d)
    def run():
        assert lib.ruart_gemm_16c_nt(hip.ptr(A16), hip.ptr(A8), K, hip.ptr(W16), hip.ptr(W8), K, hip.ptr(bias), hip.ptr(R32), N, hip.ptr(C), N,
                                     hip.ptr(C8), M, N, K, act, hip.stream_ptr()) == 0
    lib.ruart_gemm_set_stamps(None)
    for _ in range(3): run()
    lib.ruart_gemm_set_stamps(st.data_ptr()); run(); torch.cuda.synchronize(); lib.ruart_gemm_set_stamps(None)
    rw = st.cpu().numpy().reshape(ntiles, 8, 8)          # [workgroup][wave][stamp]
    tw = rw[:, :, :5].astype(np.float64) * 0.01          # us
    # workgroup view: first wave to start, last wave to reach every later mark
    t = np.concatenate([tw[:, :, :1].min(1), tw[:, :, 1:].max(1)], 1)
    raw = rw[:, 0, :]
    xcc = raw[:, 6] & 0xf
    fill, f16, f8, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]
    nt = K // 64
    t0 = t[:, 0].min()
    tile = t[:, 4] - t[:, 0]
    print("%-4s tiles %4d | fill %5.2f | f16 run %6.2f us = %.3f/K-tile | fp8 run %6.2f us = %.3f/K-tile | epilogue+drain %5.2f | tile %6.2f (p10 %.1f p90 %.1f) | kernel %7.1f us | sum of medians x rounds %.0f"
          % (name, ntiles, np.median(fill), np.median(f16), np.median(f16) / nt, np.median(f8), np.median(f8) / nt, np.median(epi), np.median(tile),
             np.percentile(tile, 10), np.percentile(tile, 90), t[:, 4].max() - t0, np.median(tile) * ntiles / 256))
    # by round of start time: how the per-K-tile time changes once the workgroups have de-phased
    order = np.argsort(t[:, 0])
    for r0 in range(0, ntiles, 256):
        sel = order[r0:r0 + 256]
        print("     workgroups %4d-%4d by start: start spread %6.1f us | fill %5.2f | f16 %.3f/K-tile | fp8 %.3f/K-tile | epi %5.2f | tile %6.2f"
              % (r0, r0 + len(sel) - 1, t[sel, 0].max() - t[sel, 0].min(), np.median(fill[sel]), np.median(f16[sel]) / nt, np.median(f8[sel]) / nt,
                 np.median(epi[sel]), np.median(tile[sel])))
    print("     workgroups per XCC id:", np.bincount(xcc.astype(int), minlength=8).tolist())
    # the waves of one workgroup: spread of their ends, and the CU's idle time between two workgroups (first wave of the next one
    # starts - last wave of the previous one has drained its stores)
    hw = raw[:, 7]
    key = xcc * 10000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 20 + ((hw >> 8) & 0xf)
    gaps = []
    for k in np.unique(key):
        sel = np.where(key == k)[0]
        o = sel[np.argsort(t[sel, 0])]
        gaps += [t[b, 0] - t[a_, 4] for a_, b in zip(o[:-1], o[1:])]
    gaps = np.array(gaps) if gaps else np.zeros(1)
    wend = tw[:, :, 4]
    wloop = tw[:, :, 3]
    print("     per workgroup: last - first wave end %.2f us (p90 %.2f), last - first wave out of the K loop %.2f | CU idle between workgroups: median %.2f us p10 %.2f p90 %.2f (%d CUs seen)"
          % (np.median(wend.max(1) - wend.min(1)), np.percentile(wend.max(1) - wend.min(1), 90), np.median(wloop.max(1) - wloop.min(1)),
             np.median(gaps), np.percentile(gaps, 10), np.percentile(gaps, 90), len(np.unique(key))))
    cyc = rw[:, 0, 5].astype(np.float64)
    loop_us = tw[:, 0, 3] - tw[:, 0, 1]
    print("     K loop of wave 0: %.0f shader cycles per K-tile (2 048 = the matrix pipe's own time: %.1f %%), in-loop clock %.3f GHz (p10 %.3f p90 %.3f)"
          % (np.median(cyc) / (2 * nt), 100.0 * 2048 * 2 * nt / np.median(cyc), np.median(cyc / loop_us) * 1e-3, np.percentile(cyc / loop_us, 10) * 1e-3,
             np.percentile(cyc / loop_us, 90) * 1e-3))
    if a.dump:
        np.save("%s_%s.npy" % (a.dump, name), rw)

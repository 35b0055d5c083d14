#!/usr/bin/env python3
"""Long race screen of the counted-vmcnt GEMM (variant 5) against the plain two-stage kernel (variant 3): many launches, many
shapes incl. every epilogue form, with a second stream hammering HBM at the same time to perturb DMA landing times."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip
lib = hip.load(); d = torch.device("cuda:0")
n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 150
noise = torch.empty(256 * 2**20 // 4, device=d)
side = torch.cuda.Stream(device=d)
bad_total = 0
for dt in (hip.DT_F16, hip.DT_BF16):
    td = hip.TORCH_DTYPE[dt]
    for (M, N, K, act, res) in [(43008, 768, 768, 0, True), (43008, 2304, 768, 0, False), (21504, 3072, 768, 1, False), (43008, 768, 3072, 0, True),
                                (1024, 1024, 128, 0, False), (2560, 512, 256, 1, False), (7680, 256, 1024, 0, True)]:
        g = torch.Generator().manual_seed(M + N + K)
        A = torch.randn(M, K, generator=g).to(td).to(d); W = (torch.randn(N, K, generator=g) * 0.05).to(td).to(d)
        bias = torch.randn(N, generator=g).to(d); R = torch.randn(M, N, generator=g).to(td).to(d) if res else None
        out_t, out_dt = (torch.float32, hip.DT_F32) if res else (td, dt)
        def run(variant, C):
            lib.ruart_gemm_set_variant(variant)
            rc = lib.ruart_gemm_16_nt(hip.ptr(A), K, hip.ptr(W), K, hip.ptr(bias), hip.ptr(R), N, dt, hip.ptr(C), N, out_dt, M, N, K, act, dt, hip.stream_ptr())
            assert rc == 0
        C3 = torch.empty(M, N, dtype=out_t, device=d); run(3, C3)
        C5 = torch.empty(M, N, dtype=out_t, device=d)
        bad = 0
        for it in range(n_launch):
            if it % 3 == 0:
                with torch.cuda.stream(side):
                    noise.mul_(1.0001)                       # 2 GB of HBM traffic beside the GEMM
            C5.fill_(float("nan"))
            run(5, C5)
            bad += int(not torch.equal(C5, C3))
        torch.cuda.synchronize()
        bad_total += bad
        print("%s dt %d M %5d N %5d K %5d act %d res %d: %d / %d launches differ" % ("OK " if bad == 0 else "BAD", dt, M, N, K, act, int(res), bad, n_launch), flush=True)
lib.ruart_gemm_set_variant(5)
sys.exit(1 if bad_total else 0)

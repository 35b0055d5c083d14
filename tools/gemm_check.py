#!/usr/bin/env python3
"""Cross-check every GEMM tile variant against variant 0 and a float64 reference on BERT-like shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip
lib = hip.load(); d = torch.device("cuda:0")
ok = True
for dt in (hip.DT_F16, hip.DT_BF16):
    td = hip.TORCH_DTYPE[dt]
    for (M, N, K, act, res) in [(512, 768, 768, 0, True), (768, 2304, 768, 0, False), (256, 3072, 768, 1, False), (1024, 768, 3072, 0, True), (256, 256, 64, 0, False)]:
        g = torch.Generator().manual_seed(M + N + K)
        A = torch.randn(M, K, generator=g).to(td); W = (torch.randn(N, K, generator=g) * 0.05).to(td)
        bias = torch.randn(N, generator=g); R = torch.randn(M, N, generator=g).to(td) if res else None
        ref = A.double() @ W.double().t() + bias.double()
        if act: ref = ref * 0.5 * (1 + torch.erf(ref / 2 ** 0.5))
        if res: ref = ref + R.double()
        Ad, Wd, bd = A.to(d), W.to(d), bias.to(d); Rd = R.to(d) if res else None
        outs = {}
        for v in (0, 3, 5):
            lib.ruart_gemm_set_variant(v)
            C = torch.full((M, N), float("nan"), dtype=torch.float32 if res else td, device=d)
            rc = lib.ruart_gemm_16_nt(hip.ptr(Ad), K, hip.ptr(Wd), K, hip.ptr(bd), hip.ptr(Rd), N, dt, hip.ptr(C), N, hip.DT_F32 if res else dt, M, N, K, act, dt, hip.stream_ptr())
            torch.cuda.synchronize()
            err = float((C.double().cpu() - ref).abs().max())
            outs[v] = C.float().cpu()
            tol = (2e-3 if res else (3e-2 if dt == hip.DT_BF16 else 4e-3)) * max(1.0, float(ref.abs().max()) / 4)
            same = bool(torch.equal(outs[v], outs[0]))
            flag = "OK " if (rc == 0 and err < tol) else "BAD"
            ok &= flag == "OK "
            print("%s dt %d M %5d N %5d K %5d act %d res %d variant %d: rc %d max err %.3e bitwise==v0 %s" % (flag, dt, M, N, K, act, int(res), v, rc, err, same), flush=True)
# race screen for the counted-vmcnt / staggered schedule (variant 5): many launches at several sizes, bitwise against variant 3
for dt in (hip.DT_F16, hip.DT_BF16):
    td = hip.TORCH_DTYPE[dt]
    for (M, N, K) in [(4096, 768, 768), (43008, 2304, 768), (8192, 3072, 768), (8192, 768, 3072), (2048, 256, 128), (512, 512, 256)]:
        g = torch.Generator().manual_seed(M + N + K + 1)
        Ad = torch.randn(M, K, generator=g).to(td).to(d); Wd = (torch.randn(N, K, generator=g) * 0.05).to(td).to(d)
        bd = torch.randn(N, generator=g).to(d)
        lib.ruart_gemm_set_variant(3)
        C3 = torch.empty((M, N), dtype=td, device=d)
        lib.ruart_gemm_16_nt(hip.ptr(Ad), K, hip.ptr(Wd), K, hip.ptr(bd), None, N, dt, hip.ptr(C3), N, dt, M, N, K, 0, dt, hip.stream_ptr())
        lib.ruart_gemm_set_variant(5)
        bad = 0
        C5 = torch.empty((M, N), dtype=td, device=d)
        for it in range(30):
            C5.fill_(float("nan"))
            lib.ruart_gemm_16_nt(hip.ptr(Ad), K, hip.ptr(Wd), K, hip.ptr(bd), None, N, dt, hip.ptr(C5), N, dt, M, N, K, 0, dt, hip.stream_ptr())
            bad += int(not torch.equal(C5, C3))
        torch.cuda.synchronize()
        ok &= bad == 0
        print("%s race screen dt %d M %5d N %5d K %5d: %d / 30 launches differ from variant 3" % ("OK " if bad == 0 else "BAD", dt, M, N, K, bad), flush=True)
lib.ruart_gemm_set_variant(5)
sys.exit(0 if ok else 1)

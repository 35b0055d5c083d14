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

# The weight-gradient kernel (gemm_16_tn_256p8: own read schedule and vmcnt accounting) and the corrected kernel (gemm_16c_nt_256p8):
# no second implementation with the same summation order exists, so every launch is compared bitwise with the first one, and the
# first one with a float64 product at the operands' own precision.
from ruart_amd.bert import split_f16c  # noqa: E402
for (M, N, T, tch) in [(768, 768, 45056, 1664), (2304, 768, 45056, 5120), (768, 3072, 45056, 6528), (256, 256, 256, 128)]:
    g = torch.Generator().manual_seed(M + N + T)
    P = (torch.randn(T, M, generator=g) * 1e-3).bfloat16().to(d)
    Q = torch.randn(T, N, generator=g).bfloat16().to(d)
    nz = (T + tch - 1) // tch
    first = torch.empty(nz, M, N, device=d)
    assert lib.ruart_gemm_16_tn_splitk(hip.ptr(P), M, hip.ptr(Q), N, hip.ptr(first), N, M, N, T, tch, hip.DT_BF16, hip.stream_ptr()) == 0
    ref = P[:, :256].double().t() @ Q[:, :256].double()
    err = float((first.sum(0)[:256, :256].double() - ref).abs().max() / ref.abs().max())
    out = torch.empty_like(first)
    bad = 0
    for it in range(n_launch):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                noise.mul_(1.0001)
        out.fill_(float("nan"))
        assert lib.ruart_gemm_16_tn_splitk(hip.ptr(P), M, hip.ptr(Q), N, hip.ptr(out), N, M, N, T, tch, hip.DT_BF16, hip.stream_ptr()) == 0
        bad += int(not torch.equal(out, first))
    torch.cuda.synchronize()
    bad_total += bad + int(err > 1e-5)
    print("%s TN M %5d N %5d T %5d chunk %4d: %d / %d launches differ from the first; first vs float64 %.1e" %
          ("OK " if bad == 0 and err <= 1e-5 else "BAD", M, N, T, tch, bad, n_launch, err), flush=True)
for (M, N, K, act, res) in [(43008, 768, 768, 0, True), (43008, 2304, 768, 0, False), (21504, 3072, 768, 1, False), (43008, 768, 3072, 0, True)]:
    g = torch.Generator().manual_seed(M + N + K + 1)
    A = torch.randn(M, K, generator=g).to(d)
    W = (torch.randn(N, K, generator=g) * 0.05).to(d)
    a16, a8 = split_f16c(A)
    whi = W.half().float()
    w16 = W.half()
    w8 = torch.cat([whi * 2.0 ** hip.f16c_shifts()[2], (W - whi) * 2.0 ** hip.f16c_shifts()[3]], 1).clamp_(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
    bias = torch.randn(N, generator=g).to(d)
    R = torch.randn(M, N, generator=g).to(d) if res else None
    if act:
        C = torch.empty(M, N, dtype=torch.float16, device=d)
        C8 = torch.empty(M, 2 * N, dtype=torch.uint8, device=d)
    else:
        C, C8 = torch.empty(M, N, device=d), None

    def run():
        rc = lib.ruart_gemm_16c_nt(hip.ptr(a16), hip.ptr(a8), K, hip.ptr(w16), hip.ptr(w8), K, hip.ptr(bias), hip.ptr(R), N, hip.ptr(C), N,
                                   hip.ptr(C8), M, N, K, act, hip.stream_ptr())
        assert rc == 0
    run()
    first = C.clone()
    bad = 0
    for it in range(n_launch):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                noise.mul_(1.0001)
        C.fill_(float("nan"))
        run()
        bad += int(not torch.equal(C, first))
    torch.cuda.synchronize()
    bad_total += bad
    print("%s 16c M %5d N %5d K %5d act %d res %d: %d / %d launches differ from the first" % ("OK " if bad == 0 else "BAD", M, N, K, act, int(res), bad,
                                                                                            n_launch), flush=True)
sys.exit(1 if bad_total else 0)

"""Per-kernel parity on a real MI355X: every HIP kernel, through the C ABI, against the CPU oracle
(oracle/ruart_oracle.py) and the reference-generated golden vectors (tests/golden/*.npz).

Tolerances: fp32 kernels 2e-5 abs (same fmaf-order noise as torch CPU vs the reference);
bf16 production path 4e-2 abs on O(1) layer-normed activations (bf16 has 8 mantissa bits; the bound that matters,
1e-3 on the final answer probabilities, is checked end-to-end in test_gpu_sdnet.py)."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ruart_oracle as O           # noqa: E402  (checker only)
from ruart_amd import hip, synth               # noqa: E402


def dev():
    assert torch.cuda.is_available(), "these tests need the GPU box"
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def maxerr(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 768), (384, 768, 3072), (128, 2304, 768)])
@pytest.mark.parametrize("mode", ["plain", "gelu", "res16_out_f32", "res_f32_out16"])
@pytest.mark.parametrize("dt", [hip.DT_BF16, hip.DT_F16])
def test_gemm_16(M, N, K, mode, dt):
    lib = hip.load()
    d = dev()
    td = hip.TORCH_DTYPE[dt]
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(td)
    W = (torch.randn(N, K, generator=g) * 0.05).to(td)
    bias = torch.randn(N, generator=g)
    ref = A.double() @ W.double().t() + bias.double()
    res = None
    res_dt, out_dt, act = dt, dt, hip.ACT_NONE
    if mode == "gelu":
        act = hip.ACT_GELU
        ref = O.gelu_erf(ref)
    elif mode == "res16_out_f32":
        res = torch.randn(M, N, generator=g).to(td)
        ref = ref + res.double()
        out_dt = hip.DT_F32
    elif mode == "res_f32_out16":
        res = torch.randn(M, N, generator=g)
        ref = ref + res.double()
        res_dt = hip.DT_F32
    Ad, Wd, bd = A.to(d), W.to(d), bias.to(d)
    Rd = res.to(d) if res is not None else None
    C = torch.empty(M, N, dtype=torch.float32 if out_dt == hip.DT_F32 else td, device=d)
    rc = lib.ruart_gemm_16_nt(hip.ptr(Ad), K, hip.ptr(Wd), K, hip.ptr(bd), hip.ptr(Rd), N, res_dt, hip.ptr(C), N, out_dt, M, N, K,
                              act, dt, hip.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    # inputs are exactly representable, accumulation is fp32: only the output rounding differs between the two storage types
    tol = 2e-3 if out_dt == hip.DT_F32 else (3e-2 if dt == hip.DT_BF16 else 4e-3)
    assert maxerr(C.float(), ref) < tol * max(1.0, float(ref.abs().max()) / 4)


def _w8(W):
    hi = W.to(torch.float16).float()
    _, _, sw_hi, sw_lo = hip.f16c_shifts()
    pair = torch.cat([hi * 2.0 ** sw_hi, (W - hi) * 2.0 ** sw_lo], 1).clamp_(-448.0, 448.0)
    return W.to(torch.float16).contiguous(), pair.to(torch.float8_e4m3fn).view(torch.uint8).contiguous()


@pytest.mark.parametrize("M,N,K,mode", [(512, 768, 768, "res32"), (768, 512, 128, "plain"), (1024, 2304, 768, "plain"), (512, 3072, 768, "gelu"),
                                        (512, 768, 3072, "res32"), (256, 256, 192, "res16")])
@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_gemm_16_one_wave_per_simd_variant(M, N, K, mode, dt):
    """Variant 7 of ruart_gemm_16_nt (round 4 experiment: 4 waves x 128x128 per 256x256 tile, accumulators in AGPRs, one barrier per
    K-tile) multiplies in the same k order as the shipped four-phase kernel (variant 5): the outputs are BIT-identical, every element
    written.  K = 192: an odd number of K-tiles (the four-phase kernel needs an even one and hands that shape to its two-stage sibling:
    compared to rounding there)."""
    lib = hip.load()
    d = dev()
    code = hip.PRECISION[dt]
    td = hip.TORCH_DTYPE[code]
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(td).to(d)
    W = (torch.randn(N, K, generator=g) * 0.05).to(td).to(d)
    bias = torch.randn(N, generator=g).to(d)
    R, rdt, of = None, code, False
    if mode == "res32":
        R, rdt, of = torch.randn(M, N, generator=g).to(d), hip.DT_F32, True
    elif mode == "res16":
        R = torch.randn(M, N, generator=g).to(td).to(d)
    act = hip.ACT_GELU if mode == "gelu" else hip.ACT_NONE
    outs = []
    try:
        for v in (5, 7):
            assert lib.ruart_gemm_set_variant(v) == 0
            C = torch.full((M, N), float("nan"), dtype=torch.float32 if of else td, device=d)
            rc = lib.ruart_gemm_16_nt(hip.ptr(A), K, hip.ptr(W), K, hip.ptr(bias), hip.ptr(R), N, rdt, hip.ptr(C), N, hip.DT_F32 if of else code, M, N, K,
                                      act, code, hip.stream_ptr())
            assert rc == 0
            torch.cuda.synchronize()
            outs.append(C)
    finally:
        lib.ruart_gemm_set_variant(5)
    assert not torch.isnan(outs[1].float()).any()
    if K % 128 == 0:
        assert torch.equal(outs[0], outs[1])
    else:
        assert float((outs[0].float() - outs[1].float()).abs().max()) <= 1e-2 * max(1.0, float(outs[0].float().abs().max()))


@pytest.mark.parametrize("M,N,K,cus", [(81 * 256, 768, 768, 240),        # 3 tail tiles, 6 slices of 2 K-tiles
                                       (128 * 256, 768, 3072, 256),      # the north-star halves: 384 tiles = 1.5 rounds, long K: 2 slices
                                       (128 * 256, 768, 768, 256)])      # ... short K: left alone
@pytest.mark.parametrize("mode", ["plain", "res32", "gelu"])
@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_gemm_16_tail_split(M, N, K, cus, mode, dt):
    """ruart_gemm_16_nt_ws (plain 16-bit product): tail tiles cut along K + fix-up launch == the single launch to fp32 summation order,
    deterministic, every element written."""
    lib = hip.load()
    d = dev()
    code = hip.PRECISION[dt]
    td = hip.TORCH_DTYPE[code]
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(td).to(d)
    W = (torch.randn(N, K, generator=g) * 0.03).to(td).to(d)
    bias = (torch.randn(N, generator=g) * 0.1).to(d)
    Rd = torch.randn(M, N, generator=g).to(d) if mode == "res32" else None
    act = hip.ACT_GELU if mode == "gelu" else hip.ACT_NONE
    out_dt = hip.DT_F32 if mode == "res32" else code
    nbytes = int(lib.ruart_gemm_16_tail_ws_bytes(M, N, K, cus))
    tiles, r = (M // 256) * (N // 256), ((M // 256) * (N // 256)) % cus
    assert (nbytes > 0) == (tiles > cus and r > 0 and (4 * r <= cus or (5 * r <= 3 * cus and K >= 2048)))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=d)

    def run(use_ws):
        C = torch.full((M, N), float("nan"), dtype=torch.float32 if mode == "res32" else td, device=d)
        rc = lib.ruart_gemm_16_nt_ws(hip.ptr(A), K, hip.ptr(W), K, hip.ptr(bias), hip.ptr(Rd), N, hip.DT_F32, hip.ptr(C), N, out_dt, M, N, K, act,
                                     code, hip.ptr(ws) if use_ws else None, nbytes if use_ws else 0, cus, hip.stream_ptr())
        assert rc == 0
        torch.cuda.synchronize()
        return C

    C0, C1, C2 = run(False), run(True), run(True)
    assert torch.equal(C1, C2)
    assert not torch.isnan(C1.float()).any()
    tol = 2e-6 if mode == "res32" else (1e-2 if dt == "bf16" else 2e-3)              # 16-bit outputs: one rounding step
    err = float((C1.float() - C0.float()).abs().max()) / max(1.0, float(C0.float().abs().max()))
    assert err < tol, err
    if nbytes == 0:
        assert torch.equal(C0, C1)


@pytest.mark.parametrize("M,N,K,cus", [(81 * 256, 768, 768, 240),      # 243 tiles: 3 tail tiles cut into 6 slices of 4 K-tiles
                                       (85 * 256, 1024, 3072, 240),    # 340 tiles: 100 tail tiles (long K), 2 slices = the f16 and the fp8 phase
                                       (85 * 256, 1024, 768, 240),     # the same at K = 768: a 42 %-full last round of short tiles is left alone
                                       (43 * 256, 768, 3072, 120),     # 129 tiles: 9 tail tiles, 8 slices of 12 K-tiles (K = 3072)
                                       (16 * 256, 768, 768, 240)])     # one partial round: left alone (no second launch)
@pytest.mark.parametrize("mode", ["plain", "res", "gelu"])
def test_gemm_16c_tail_split(M, N, K, cus, mode):
    """ruart_gemm_16c_nt_ws: the tiles of the last, partial round of `cus` CUs are cut along K, a second launch sums a tile's slices in
    slice order and runs its epilogue.  Same product as the single launch to fp32 summation order, bit-identical between two runs, and
    every output element written (the buffers start as NaN)."""
    from ruart_amd.bert import split_f16c
    lib = hip.load()
    d = dev()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.03
    bias = (torch.randn(N, generator=g) * 0.1).to(d)
    A16, A8 = [t.to(d) for t in split_f16c(A)]
    W16, W8 = [t.to(d) for t in _w8(W)]
    Rd = torch.randn(M, N, generator=g).to(d) if mode == "res" else None
    act = hip.ACT_GELU if mode == "gelu" else hip.ACT_NONE
    nbytes = int(lib.ruart_gemm_16c_tail_ws_bytes(M, N, K, cus))
    tiles = (M // 256) * (N // 256)
    r = tiles % cus
    assert (nbytes > 0) == (tiles > cus and r > 0 and (4 * r <= cus or (5 * r <= 3 * cus and K >= 2048)))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=d)

    def run(use_ws):
        if mode == "gelu":
            C = torch.full((M, N), float("nan"), dtype=torch.float16, device=d)
            C8 = torch.full((M, 2 * N), 0x7f, dtype=torch.uint8, device=d)          # 0x7f = e4m3 NaN
        else:
            C = torch.full((M, N), float("nan"), dtype=torch.float32, device=d)
            C8 = None
        rc = lib.ruart_gemm_16c_nt_ws(hip.ptr(A16), hip.ptr(A8), K, hip.ptr(W16), hip.ptr(W8), K, hip.ptr(bias), hip.ptr(Rd), N, hip.ptr(C), N,
                                      hip.ptr(C8), M, N, K, act, 3, hip.ptr(ws) if use_ws else None, nbytes if use_ws else 0, cus,
                                      hip.stream_ptr())
        assert rc == 0
        torch.cuda.synchronize()
        return C, C8

    C0, C80 = run(False)
    C1, C81 = run(True)
    C2, C82 = run(True)
    assert torch.equal(C1, C2) and (C81 is None or torch.equal(C81, C82))            # deterministic
    assert not torch.isnan(C1.float()).any()
    ref = C0.float()
    tol = 2e-3 if mode == "gelu" else 2e-6                                           # f16 output rounding / fp32 summation order
    err = float((C1.float() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    assert err < tol, err
    if nbytes == 0:
        assert torch.equal(C0, C1)
    if C81 is not None:
        assert int((C81 != C80).sum()) < 0.01 * C81.numel()                          # the fp8 companions move only where a value sat on a rounding edge


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 768), (256, 768, 3072), (256, 2304, 768)])
@pytest.mark.parametrize("mode", ["plain", "res", "gelu"])
def test_gemm_16c_fp8_correction(M, N, K, mode):
    """ruart_gemm_16c_nt: f16 MFMA product + block-scaled fp8 correction of both operands' rounding residuals, against an fp64
    product of the UNROUNDED fp32 operands.  The plain f16 kernel on the same data is measured beside it: the corrected product
    must be >= 8x closer (it is ~2^-16 against ~2^-12 relative)."""
    from ruart_amd.bert import split_f16c
    lib = hip.load()
    d = dev()
    g = torch.Generator().manual_seed(M + N + K + len(mode))
    A = torch.randn(M, K, generator=g) * (1.0 + 3.0 * (torch.rand(M, 1, generator=g) > 0.9))      # some rows of larger magnitude
    if mode == "plain":
        A[0, :8] = torch.tensor([0.0, 1e-6, -1e-4, 90.0, -440.0, 3e-3, 1.0, -1.0])                    # range edges of the fp8 halves
    W = torch.randn(N, K, generator=g) * 0.03
    bias = torch.randn(N, generator=g) * 0.1
    ref = A.double() @ W.double().t() + bias.double()
    A16, A8 = split_f16c(A)
    W16, W8 = _w8(W)
    A16d, A8d, W16d, W8d, bd = A16.to(d), A8.to(d), W16.to(d), W8.to(d), bias.to(d)
    res = Rd = None
    act = hip.ACT_NONE
    C8 = None
    if mode == "res":
        res = torch.randn(M, N, generator=g)
        Rd = res.to(d)
        ref = ref + res.double()
    if mode == "gelu":
        act = hip.ACT_GELU
        ref = O.gelu_erf(ref)
        C = torch.zeros(M, N, dtype=torch.float16, device=d)
        C8 = torch.zeros(M, 2 * N, dtype=torch.uint8, device=d)
    else:
        C = torch.zeros(M, N, dtype=torch.float32, device=d)
    rc = lib.ruart_gemm_16c_nt(hip.ptr(A16d), hip.ptr(A8d), K, hip.ptr(W16d), hip.ptr(W8d), K, hip.ptr(bd), hip.ptr(Rd), N, hip.ptr(C), N,
                               hip.ptr(C8), M, N, K, act, hip.stream_ptr())
    assert rc == 0
    # the plain f16 product of the same (rounded) operands
    P = torch.zeros(M, N, dtype=torch.float32, device=d)
    rc = lib.ruart_gemm_16_nt(hip.ptr(A16d), K, hip.ptr(W16d), K, hip.ptr(bd), hip.ptr(Rd), N, hip.DT_F32, hip.ptr(P), N, hip.DT_F32, M, N, K,
                              hip.ACT_NONE, hip.DT_F16, hip.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    scale = float(ref.abs().mean())
    if mode == "gelu":
        sa_lo, sa_hi, _, _ = hip.f16c_shifts()
        lo8 = C8[:, :N].view(torch.float8_e4m3fn).float().cpu() / 2.0 ** sa_lo
        hi8 = C8[:, N:].view(torch.float8_e4m3fn).float().cpu() / 2.0 ** sa_hi
        got = C.float().cpu() + lo8
        err = float((got.double() - ref).abs().max())
        print("gelu split out: max err %.2e (mean |ref| %.2e); f16 part alone %.2e" % (err, scale, maxerr(C.float(), ref)))
        assert err < 3e-5 * max(1.0, float(ref.abs().max()))
        assert float((hi8.double() - ref).abs().max()) < 0.07 * float(ref.abs().max()) + 2e-3      # e4m3: 3 mantissa bits
        return
    err_c, err_p = maxerr(C, ref), maxerr(P, ref)
    # (row 0 of the "plain" case carries the range-edge values - its products are two orders above the matrix's typical magnitude,
    # so it is held to its own scale below and left out of the matrix-wide rms)
    r0 = 1 if mode == "plain" else 0
    rms_c = float((C.double().cpu() - ref)[r0:].pow(2).mean().sqrt())
    rms_p = float((P.double().cpu() - ref)[r0:].pow(2).mean().sqrt())
    if mode == "plain":
        assert float((C.double().cpu() - ref)[0].abs().max()) < 1e-4 * float(ref[0].abs().max())
    print("K=%d %s: corrected max %.2e rms %.2e | plain f16 max %.2e rms %.2e | mean |ref| %.2e" % (K, mode, err_c, rms_c, err_p, rms_p, scale))
    assert rms_c < rms_p / 8 and err_c < err_p / 4
    assert rms_c < 2e-5 * scale * max(1.0, (K / 768) ** 0.5)


def test_gemm_16_rejects_bad_shapes():
    lib = hip.load()
    d = dev()
    x = torch.zeros(128, 64, dtype=torch.bfloat16, device=d)
    assert lib.ruart_gemm_16_nt(hip.ptr(x), 64, hip.ptr(x), 64, None, None, 0, 1, hip.ptr(x), 128, 1, 100, 128, 64, 0, 1,
                                hip.stream_ptr()) != 0


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (37, 53, 29), (64, 64, 16), (200, 250, 1388), (130, 125, 250), (100, 1200, 300)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_f32(M, N, K, act):
    lib = hip.load()
    d = dev()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = A.double() @ W.double().t() + bias.double()
    ref = O.gelu_erf(ref) if act == 1 else (torch.relu(ref) if act == 2 else ref)
    ref = ref + res.double()
    Ad, Wd, bd, Rd = A.to(d), W.to(d), bias.to(d), res.to(d)
    C = torch.empty(M, N, device=d)
    assert lib.ruart_gemm_f32_nt(hip.ptr(Ad), K, hip.ptr(Wd), K, hip.ptr(bd), hip.ptr(Rd), N, hip.ptr(C), N, M, N, K, act,
                                 hip.stream_ptr()) == 0
    assert maxerr(C, ref) < 1e-5 * max(1.0, float(ref.abs().max())) * max(1.0, K / 64)


@pytest.mark.parametrize("H", [128, 768, 1024])
@pytest.mark.parametrize("out_dt", [hip.DT_F32, hip.DT_BF16, hip.DT_F16])
def test_rows_layernorm_and_embed(H, out_dt):
    lib = hip.load()
    d = dev()
    g = torch.Generator().manual_seed(H)
    rows = 37
    x = torch.randn(rows, H, generator=g) * 2 + 0.3
    gamma, beta = 1 + 0.1 * torch.randn(H, generator=g), 0.1 * torch.randn(H, generator=g)
    ref = O.bert_layer_norm(x, gamma, beta)
    out = torch.empty(rows, H, dtype=hip.TORCH_DTYPE[out_dt], device=d)
    xd, gd, bd = x.to(d), gamma.to(d), beta.to(d)
    assert lib.ruart_rows_layernorm(hip.ptr(xd), H, hip.ptr(gd), hip.ptr(bd), 1e-12, hip.ptr(out), H, out_dt, rows, H,
                                    hip.stream_ptr()) == 0
    assert maxerr(out.float(), ref) < {hip.DT_F32: 3e-6, hip.DT_BF16: 2e-2, hip.DT_F16: 3e-3}[out_dt]
    V, P = 50, 40
    word, ptab, ttab = torch.randn(V, H, generator=g), torch.randn(P, H, generator=g), torch.randn(2, H, generator=g)
    ids = torch.randint(0, V, (rows,), generator=g)
    pos = torch.randint(0, P, (rows,), generator=g)
    ref = O.bert_layer_norm(word[ids] + ptab[pos] + ttab[0], gamma, beta)
    wd, pd, td = word.to(d), ptab.to(d), ttab.to(d)
    idd, posd = ids.int().to(d), pos.int().to(d)
    assert lib.ruart_bert_embed_ln(hip.ptr(idd), hip.ptr(posd), hip.ptr(wd), hip.ptr(pd), hip.ptr(td), hip.ptr(gd), hip.ptr(bd),
                                   1e-12, hip.ptr(out), H, out_dt, rows, H, hip.stream_ptr()) == 0
    assert maxerr(out.float(), ref) < {hip.DT_F32: 3e-6, hip.DT_BF16: 2e-2, hip.DT_F16: 3e-3}[out_dt]


# ---------------------------------------------------------------------------------------------------------
def _bert_case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    c = z["cfg"]
    cfg = synth.bert_config(vocab_size=int(c[0]), hidden_size=int(c[1]), num_hidden_layers=int(c[2]),
                            num_attention_heads=int(c[3]), intermediate_size=int(c[4]), max_position_embeddings=int(c[5]))
    return z, cfg, synth.make_bert_weights(cfg, seed=int(z["seed"]))


@pytest.mark.parametrize("name", ["bert_small", "bert_base"])
@pytest.mark.parametrize("precision,pack,tol", [("fp32", True, 5e-5), ("fp32", False, 5e-5), ("bf16", True, 6e-2), ("fp16", True, 1.5e-2), ("fp16", False, 1.5e-2),
                                                ("x3", True, 3e-4), ("fp16c", True, 6e-4), ("fp16c", False, 6e-4),
                                                ("fp16c-fold", True, 6e-4), ("fp16c-fold", False, 6e-4),
                                                ("fp16-fold", True, 1.5e-2), ("fp16-fold", False, 1.5e-2), ("bf16-fold", True, 6e-2)])
def test_bert_encoder_vs_reference_golden(golden_dir, name, precision, pack, tol):
    """Whole encoder through ruart_bert_forward (fp16c-fold: ruart_bert_forward_folded, the LayerNorms folded into the projections;
    its pre-LayerNorm rows are normalised by layer_outputs) vs the reference's own layer outputs (gen_golden.py)."""
    from ruart_amd.bert import BertEncoderWeights, PackedTokens, bert_encode, layer_outputs
    z, cfg, w = _bert_case(golden_dir, name)
    fold = precision.endswith("-fold")
    precision = precision.replace("-fold", "")
    if (precision == "fp16c" or fold) and cfg["hidden_size"] % 256:
        pytest.skip("the f16 + fp8-correction GEMM and the folded passes take hidden sizes that are multiples of 256 (bert-base / bert-large)")
    d = dev()
    W = BertEncoderWeights(w, cfg, d, precision, ln_fold=fold)
    assert W.ln_fold == fold
    ids, mask = T(z["ids"]), T(z["mask"])
    packed = PackedTokens([(ids, mask)], d, pack=pack, mfma_long=precision in ("fp16", "bf16"))
    layers = layer_outputs(bert_encode(W, packed)).float().cpu()
    gi = packed.group_index[0]
    sel = T(z["mask"]).bool()
    for k in z.files:
        if not k.startswith("layer"):
            continue
        ref = T(z[k])[sel]                                   # (valid tokens, H) in row-major order
        got = layers[int(k[5:])][T(gi)[sel]]
        err = maxerr(got, ref)
        assert err < tol, "%s %s %s: max abs err %.3e" % (name, precision, k, err)


@pytest.mark.parametrize("precision,tol", [("fp32", 5e-5), ("fp16", 1.5e-2), ("bf16", 8e-2), ("fp16c", 5e-4)])
def test_bert_long_sequences_and_split_groups(golden_dir, precision, tol):
    """Sequences longer than one 64-query block (the (B, 512)-style shape: MFMA flash kernel in the 16-bit modes, key-tiled
    VALU kernel in fp32) mixed with short ones, several groups in one pass, and the unpacked -10000 mode."""
    from ruart_amd.bert import BertEncoderWeights, PackedTokens, bert_encode
    hid = 256 if precision == "fp16c" else 128        # (the f16 + fp8-correction GEMM needs hidden % 256 == 0)
    cfg = synth.bert_config(vocab_size=300, hidden_size=hid, num_hidden_layers=2, num_attention_heads=hid // 64, intermediate_size=2 * hid,
                            max_position_embeddings=256)
    w = synth.make_bert_weights(cfg, seed=3, w_std=0.08)
    g = np.random.default_rng(0)
    lens_a, lens_b = [200, 65, 64, 1, 130], [7, 3, 256, 129]
    def mk(lens, L):
        ids = np.zeros((len(lens), L), dtype=np.int64)
        for i, l in enumerate(lens):
            ids[i, :l] = g.integers(1, 300, size=l)
        return T(ids), T(ids != 0)
    ga, gb = mk(lens_a, 200), mk(lens_b, 256)
    d = dev()
    W = BertEncoderWeights(w, cfg, d, precision)
    wt = {k: T(v) for k, v in w.items()}
    for pack in (True, False):
        packed = PackedTokens([ga, gb], d, pack=pack, mfma_long=precision in ("fp16", "bf16"))
        assert (packed.n_long_blocks > 0) == (precision in ("fp16", "bf16"))
        layers = bert_encode(W, packed).float().cpu()
        for gi, (ids, mask) in enumerate((ga, gb)):
            with torch.no_grad():
                ref = O.bert_forward(wt, cfg, ids, mask)
            idx = T(packed.group_index[gi])
            for l in range(2):
                err = maxerr(layers[l][idx[mask]], ref[l][mask])
                assert err < tol, (precision, pack, gi, l, err)


def test_bert_rows_longer_than_512_are_windowed(golden_dir):
    """``Bert.forward`` on rows of up to 600 word pieces against the oracle's restatement of the reference's 512-windowing
    (Models/Bert/Bert.py:133-138: independent windows, positions restart), incl. a word whose pieces straddle the boundary and
    the ``BERT_MAX_BatchSize`` row split (:65-85)."""
    from ruart_amd.bert import Bert
    d = dev()
    cfg = synth.bert_config(vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                            max_position_embeddings=512)
    w = synth.make_bert_weights(cfg, seed=9, w_std=0.05)
    g = np.random.default_rng(4)
    lens = [600, 40, 513]
    Lb = 600
    ids = np.zeros((3, Lb), dtype=np.int64)
    for i, l in enumerate(lens):
        ids[i, :l] = g.integers(1, 300, size=l)
    mask = ids != 0
    # words: spans of 1-3 pieces; row 0 has a word over pieces [510, 514) - across the window boundary
    offsets, Lw = [], 8
    offsets.append([[0, 1], [1, 3], [510, 514], [514, 515], [598, 600]])
    offsets.append([[0, 2], [2, 3], [39, 40]])
    offsets.append([[511, 513], [0, 1]])
    wmask = np.zeros((3, Lw), dtype=bool)
    for i, o in enumerate(offsets):
        wmask[i, :len(o)] = True
    from ruart_amd.bert import BertEncoderWeights

    def small_bert(**opt_extra):
        m = Bert.__new__(Bert)                                               # (the constructor insists on bert-base / -large dims)
        torch.nn.Module.__init__(m)
        m._device, m.pack, m.opt = d, True, dict(opt_extra)
        m.weights = BertEncoderWeights(w, cfg, d, "fp32")
        return m

    outs = small_bert()(T(ids), T(mask), offsets, T(wmask))
    bw = {k: T(v) for k, v in w.items()}
    per_window = [O.bert_forward(bw, cfg, T(ids[:, p0:p0 + 512]), T(mask[:, p0:p0 + 512])) for p0 in range(0, Lb, 512)]
    ref_layers = [torch.cat([pw[l] for pw in per_window], 1) for l in range(cfg["num_hidden_layers"])]     # the reference's window loop
    ref = O.pool_subwords(ref_layers, offsets, T(wmask))
    for l in range(cfg["num_hidden_layers"]):
        assert outs[l].shape == ref[l].shape and maxerr(outs[l], ref[l]) < 5e-5, l
    outs2 = small_bert(BERT_MAX_BatchSize=2)(T(ids), T(mask), offsets, T(wmask))
    for a, b in zip(outs, outs2):
        assert maxerr(a, b) < 1e-6


def test_pool_mix_fwd_bwd():
    from ruart_amd.bert import BertEncoderWeights, PackedTokens, bert_encode, Bert
    cfg = synth.bert_config(vocab_size=300, hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256,
                            max_position_embeddings=64)
    w = synth.make_bert_weights(cfg, seed=9)
    d = dev()
    opt = {"bert_state": w, "bert_config": cfg, "bert_precision": "fp32", "BERT_LINEAR_COMBINE": True}
    m = Bert.__new__(Bert)
    torch.nn.Module.__init__(m)
    m._device, m.pack = d, True
    m.weights = BertEncoderWeights(w, cfg, d, "fp32")
    g = np.random.default_rng(4)
    N, L, Lw = 6, 14, 5
    ids = np.zeros((N, L), dtype=np.int64)
    offsets, wmask = [], np.zeros((N, Lw), dtype=bool)
    for n in range(N):
        nw = int(g.integers(0, Lw + 1)) if n else Lw
        cur, offs = 1, []
        for j in range(nw):
            c = int(g.integers(1, 3))
            offs.append([cur, cur + c])
            cur += c
            wmask[n, j] = True
        if nw >= 2:
            offs[1] = [offs[1][0], offs[1][0]]            # an empty span -> zeros (Bert.py:160-165)
        ids[n, :cur + 1] = g.integers(1, 300, size=cur + 1)
        offsets.append(offs if nw else [1, 1])
    ids_t, mask_t = T(ids), T(ids != 0)
    packed, layers = m.encode([(ids_t, mask_t)])
    alpha = torch.tensor([0.3, -0.2, 1.1], device=d, requires_grad=True)
    gamma = torch.tensor(0.8, device=d, requires_grad=True)
    lw = torch.softmax(alpha, 0) * gamma
    out = m.pool_mix(packed, layers, 0, offsets, T(wmask), lw)
    gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(1))
    (out * gy.to(d)).sum().backward()
    wt = {k: T(v) for k, v in w.items()}
    a_c = alpha.detach().cpu().requires_grad_()
    g_c = gamma.detach().cpu().requires_grad_()
    with torch.no_grad():
        ref_layers = O.bert_forward(wt, cfg, ids_t, mask_t)
    pooled = O.pool_subwords(ref_layers, offsets, T(wmask))
    ref = O.linear_sum(pooled, a_c, g_c.view(1, 1))
    (ref * gy).sum().backward()
    assert maxerr(out, ref) < 3e-5
    assert maxerr(alpha.grad, a_c.grad) < 2e-4 * max(1.0, float(a_c.grad.abs().max()))
    assert maxerr(gamma.grad, g_c.grad) < 2e-4 * max(1.0, float(g_c.grad.abs().max()))
    # the reference-compatible API: list of per-layer pooled tensors
    outs = m.forward(ids_t, mask_t, offsets, T(wmask))
    for l in range(3):
        assert maxerr(outs[l], pooled[l]) < 3e-5


# ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def layers_golden(golden_dir):
    return np.load(os.path.join(golden_dir, "layers.npz"))


@pytest.mark.parametrize("tag", ["attn_a", "attn_b", "attn_c", "attn_d"])
def test_fused_attention_vs_reference(layers_golden, tag):
    from ruart_amd import ops
    z = layers_golden
    d = dev()
    x1 = T(z[tag + "_x1"]).to(d).requires_grad_()
    x2 = T(z[tag + "_x2"]).to(d).requires_grad_()
    W = T(z[tag + "_W"]).to(d).requires_grad_()
    diag = T(z[tag + "_diag"]).to(d).requires_grad_()
    x3 = T(z[tag + "_x3"]).to(d).requires_grad_() if tag + "_x3" in z.files else None
    # raw projections in; ReLU and the diagonal are applied inside the kernel (and chained in its backward)
    y = ops.fused_attention(x1 @ W.t(), x2 @ W.t(), x2 if x3 is None else x3, T(z[tag + "_mask"]).to(d), diag=diag, relu=True)
    assert maxerr(y, T(z[tag + "_y"])) < 1e-5 * max(1.0, float(np.abs(z[tag + "_y"]).max()))
    y.backward(T(z[tag + "_gy"]).to(d))
    ops.nan_flag.check_and_clear()
    for got, name in ((x1.grad, "_gx1"), (x2.grad, "_gx2"), (W.grad, "_gW")):
        ref = T(z[tag + name])
        assert maxerr(got, ref) < 3e-5 * max(1.0, float(ref.abs().max())), name
    if x3 is not None:
        assert maxerr(x3.grad, T(z[tag + "_gx3"])) < 3e-5
    if tag + "_gdiag" in z.files:
        ref = T(z[tag + "_gdiag"])
        assert maxerr(diag.grad, ref) < 3e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("B,L1,L2,h,D3", [(1, 1, 1, 1, 1), (2, 205, 40, 300, 300), (3, 100, 100, 250, 250), (2, 33, 17, 8, 250),
                                          (2, 40, 256, 37, 5), (2, 300, 300, 250, 250), (1, 20, 384, 16, 8),
                                          (64, 1, 100, 500, 500), (3, 1, 40, 250, 250), (2, 1, 384, 7, 3)])     # single-query fast path
def test_fused_attention_shapes(B, L1, L2, h, D3):
    from ruart_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(B * 1000 + L1 + L2)
    a, k, v = torch.randn(B, L1, h, generator=g), torch.randn(B, L2, h, generator=g), torch.randn(B, L2, D3, generator=g)
    mask = torch.ones(B, L2, dtype=torch.uint8)
    for b in range(B):
        mask[b, int(torch.randint(1, L2 + 1, (1,), generator=g)):] = 0
    ac, kc, vc = [t.clone().requires_grad_() for t in (a, k, v)]
    s = torch.bmm(ac, kc.transpose(1, 2)).masked_fill(mask.eq(0).unsqueeze(1), float("-inf"))
    ref = torch.bmm(torch.softmax(s, 2), vc)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    ad, kd, vd = [t.to(d).requires_grad_() for t in (a, k, v)]
    y = ops.fused_attention(ad, kd, vd, mask.to(d))
    y.backward(gy.to(d))
    scale = max(1.0, float(ref.abs().max()))
    assert maxerr(y, ref) < 2e-5 * scale
    for got, want in ((ad.grad, ac.grad), (kd.grad, kc.grad), (vd.grad, vc.grad)):
        assert maxerr(got, want) < 5e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("B,L1,L2,h,D3,relu", [(5, 9, 9, 64, 64, False), (2, 50, 50, 64, 64, False), (3, 1, 12, 64, 64, False),
                                               (2, 37, 21, 30, 17, True), (2, 20, 300, 16, 40, False)])
def test_fused_attention_probability_dropout(B, L1, L2, h, D3, relu):
    """ruart_attn_fwd/bwd_pscale: out = (softmax(mask(a.k^T)) * prob_scale) . v with the multiplier of BERT's attention-probability
    dropout (modeling.py:244-246); output and all three gradients against torch autograd on the CPU."""
    from ruart_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(B * 100 + L1 + L2)
    a, k, v = torch.randn(B, L1, h, generator=g), torch.randn(B, L2, h, generator=g), torch.randn(B, L2, D3, generator=g)
    mask = torch.ones(B, L2, dtype=torch.uint8)
    for b in range(B):
        mask[b, int(torch.randint(1, L2 + 1, (1,), generator=g)):] = 0
    ps = (torch.rand(B, L1, L2, generator=g) >= 0.3).float() / 0.7
    diag = torch.rand(h, generator=g) + 0.5 if relu else None
    ac, kc, vc = [t.clone().requires_grad_() for t in (a, k, v)]
    aa, kk = (torch.relu(ac) * diag, torch.relu(kc)) if relu else (ac, kc)
    s = torch.bmm(aa, kk.transpose(1, 2)).masked_fill(mask.eq(0).unsqueeze(1), float("-inf"))
    ref = torch.bmm(torch.softmax(s, 2) * ps, vc)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    ad, kd, vd = [t.to(d).requires_grad_() for t in (a, k, v)]
    y = ops.fused_attention(ad, kd, vd, mask.to(d), diag=None if diag is None else diag.to(d), relu=relu, prob_scale=ps.to(d))
    y.backward(gy.to(d))
    assert maxerr(y, ref) < 2e-5 * max(1.0, float(ref.abs().max()))
    for got, want in ((ad.grad, ac.grad), (kd.grad, kc.grad), (vd.grad, vc.grad)):
        assert maxerr(got, want) < 5e-5 * max(1.0, float(want.abs().max()))
    # no multiplier == the plain entry point, bit for bit (L1 == 1 without a multiplier takes the single-query kernel instead)
    if L1 == 1:
        return
    y0 = ops.fused_attention(ad.detach(), kd.detach(), vd.detach(), mask.to(d), diag=None if diag is None else diag.to(d), relu=relu)
    y1 = ops.fused_attention(ad.detach(), kd.detach(), vd.detach(), mask.to(d), diag=None if diag is None else diag.to(d), relu=relu,
                             prob_scale=torch.ones(B, L1, L2, device=d))
    assert torch.equal(y0, y1)


def test_whole_layer_norm(layers_golden):
    from ruart_amd import ops
    z = layers_golden
    d = dev()
    x = T(z["wln_x"]).to(d).requires_grad_()
    y = ops.whole_layer_norm(x)
    assert maxerr(y, T(z["wln_y"])) < 3e-6
    y.backward(T(z["wln_gy"]).to(d))
    assert maxerr(x.grad, T(z["wln_gx"])) < 3e-6
    # conf-sized tensor vs torch CPU
    g = torch.Generator().manual_seed(5)
    big = torch.randn(64, 100, 250, generator=g) * 1.7 - 0.4
    bc = big.clone().requires_grad_()
    ref = torch.nn.functional.layer_norm(bc, bc.size())
    gy = torch.randn(big.shape, generator=g)
    ref.backward(gy)
    bd = big.to(d).requires_grad_()
    yd = ops.whole_layer_norm(bd)
    yd.backward(gy.to(d))
    assert maxerr(yd, ref) < 5e-6 and maxerr(bd.grad, bc.grad) < 5e-6


@pytest.fixture(params=[1, 0], ids=["mfma16", "valu"])
def lstm_variant(request):
    """both kernel forms of the persistent recurrence: 16 batch rows per workgroup on the matrix cores (default) / one row, fp32 FMAs"""
    lib = hip.load()
    assert lib.ruart_lstm_set_variant(request.param) == 0
    yield request.param
    assert lib.ruart_lstm_set_variant(1) == 0


@pytest.mark.parametrize("B,Tn,Din,Hh,bid", [(3, 11, 10, 6, True), (5, 4, 14, 9, False), (64, 100, 300, 125, True), (2, 1, 5, 128, True),
                                             (19, 7, 12, 125, True), (33, 40, 20, 124, False)])
def test_lstm_layer(B, Tn, Din, Hh, bid, lstm_variant):
    from ruart_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(B + Tn + Hh)
    k = 1.0 / np.sqrt(Hh)
    nd = 2 if bid else 1
    names = ["w_ih", "w_hh", "b_ih", "b_hh"]
    shapes = [(4 * Hh, Din), (4 * Hh, Hh), (4 * Hh,), (4 * Hh,)]
    P = [[(torch.rand(s, generator=g) * 2 - 1) * k for s in shapes] for _ in range(nd)]
    x = torch.randn(B, Tn, Din, generator=g)
    xc = x.clone().requires_grad_()
    Pc = [[t.clone().requires_grad_() for t in p] for p in P]
    ys = [O.lstm_direction(xc, *Pc[0])]
    if bid:
        ys.append(O.lstm_direction(xc, *Pc[1], reverse=True))
    ref = torch.cat(ys, 2)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    xd = x.to(d).requires_grad_()
    Pd = [[t.to(d).requires_grad_() for t in p] for p in P]
    args = list(Pd[0]) + (list(Pd[1]) if bid else [])
    y = ops.lstm_layer(xd, Pd[0][0], Pd[0][1], Pd[0][2], Pd[0][3], *(Pd[1] if bid else []))
    y.backward(gy.to(d))
    ops.nan_flag.check_and_clear()
    assert maxerr(y, ref) < 2e-5
    assert maxerr(xd.grad, xc.grad) < 5e-5 * max(1.0, float(xc.grad.abs().max()))
    for dd in range(nd):
        for i, n in enumerate(names):
            want = Pc[dd][i].grad
            assert maxerr(Pd[dd][i].grad, want) < 1e-4 * max(1.0, float(want.abs().max())), (dd, n)


def test_fused_lstm_cell_ragged():
    """ruart_lstm_cell_fwd/bwd against the same step written with torch ops, rows >= n_active passing through."""
    from ruart_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(3)
    N, h, n = 37, 300, 21
    pre, hp, cp = torch.randn(n, 4 * h, generator=g), torch.randn(N, h, generator=g), torch.randn(N, h, generator=g)
    gh, gc = torch.randn(N, h, generator=g), torch.randn(N, h, generator=g)
    a, b, c = [t.clone().requires_grad_() for t in (pre, hp, cp)]
    i, f, gg, o = a[:, :h], a[:, h:2 * h], a[:, 2 * h:3 * h], a[:, 3 * h:]
    cn = torch.sigmoid(f) * c[:n] + torch.sigmoid(i) * torch.tanh(gg)
    hn = torch.sigmoid(o) * torch.tanh(cn)
    h_ref, c_ref = torch.cat([hn, b[n:]], 0), torch.cat([cn, c[n:]], 0)
    ((h_ref * gh).sum() + (c_ref * gc).sum()).backward()
    ad, bd, cd = [t.to(d).requires_grad_() for t in (pre, hp, cp)]
    h_out, c_out = ops.lstm_cell(ad, bd, cd, n)
    ((h_out * gh.to(d)).sum() + (c_out * gc.to(d)).sum()).backward()
    assert maxerr(h_out, h_ref) < 2e-6 and maxerr(c_out, c_ref) < 2e-6
    assert maxerr(ad.grad, a.grad) < 1e-5 and maxerr(cd.grad, c.grad) < 1e-5
    assert maxerr(bd.grad[n:], b.grad[n:]) < 1e-6 and float(bd.grad[:n].abs().max()) == 0.0


def test_wide_lstm_module_path():
    """StackedBRNN with hidden > 128 (the generic multi2one module API) on the fused cell + library GEMMs."""
    import ruart_amd.layers as L
    d = dev()
    L.set_dropout_prob(0.0)
    torch.manual_seed(0)
    m = L.StackedBRNN(20, 150, 1, bidirectional=True).to(d)
    x = torch.randn(4, 5, 20, device=d, requires_grad=True)
    y = m(x, None)
    ref_m = torch.nn.LSTM(20, 150, batch_first=True, bidirectional=True)
    ref_m.load_state_dict({k.replace("rnns.0.", ""): v.cpu() for k, v in m.state_dict().items()})
    xc = x.detach().cpu().requires_grad_()
    yr = ref_m(xc)[0]
    gy = torch.randn(yr.shape)
    yr.backward(gy)
    y.backward(gy.to(d))
    assert maxerr(y, yr) < 1e-5 and maxerr(x.grad, xc.grad) < 5e-5


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(6400, 250, 300), (500, 125, 6400), (300, 300, 14336), (1107, 1200, 300), (129, 131, 33),
                                   (64, 500, 250), (2304, 1000, 1250), (17, 5, 4097), (12800, 1200, 1388), (4096, 1100, 2048)])
def test_gemm_x3_all_layouts(M, N, K):
    """ruart_gemm_x3 (fp32 GEMM as three bf16 MFMA products) against float64, for the four stride combinations the trunk uses
    (x W^T, dY W, dY^T X and the remaining one), odd sizes (scalar load path, partial tiles, K tail), split-K shapes, both tile
    sizes (the last two shapes take the 256x256 tile, the last one with a K split) and a bias.
    Error bound: each product carries <= 2^-16 relative error -> |err| <= ~2e-5 * sum_k |a||b|; repeated launches are
    bit-identical (split-K sums its slices in a fixed order)."""
    from ruart_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(K, N, generator=g) * 0.1
    bias = torch.randn(N, generator=g)
    ref = a.double() @ b.double() + bias.double()
    bound = 2.5e-5 * (a.abs().double() @ b.abs().double()) + 1e-6
    ad, bd, biasd = a.cuda(), b.cuda(), bias.cuda()
    assert ops.trunk_gemm == "x3"
    outs = []
    for a_t in (False, True):
        for b_t in (False, True):
            av = ad.t().contiguous().t() if a_t else ad              # same values, column-major memory
            bv = bd.t().contiguous().t() if b_t else bd
            c = ops.mm(av, bv, biasd)
            c2 = ops.mm(av, bv, biasd)
            assert torch.equal(c, c2)
            err = (c.double().cpu() - ref).abs()
            assert bool((err <= bound).all()), (a_t, b_t, float((err / bound).max()))
            outs.append(c)
    # (the tile size and the K split are chosen per layout, so different layouts may sum in a different order: each is held to
    #  the bound above and to run-to-run bit equality, not to equality with the others)
    # views with a row stride larger than the row (slices of a wider matrix) and an unaligned base
    wide = torch.randn(M, K + 7, generator=g).cuda()
    av = wide[:, 3:3 + K]
    c = ops.mm(av, bd)
    err = (c.double().cpu() - av.double().cpu() @ b.double()).abs()
    assert bool((err <= 2.5e-5 * (av.abs().double().cpu() @ b.abs().double()) + 1e-6).all())


def test_linear_x3_autograd_matches_fp32():
    from ruart_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(64, 100, 300, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(1000, 300, generator=g) * 0.05).cuda().requires_grad_(True)
    b = torch.randn(1000, generator=g).cuda().requires_grad_(True)
    gy = torch.randn(64, 100, 1000, generator=g).cuda()
    y = ops.linear(x, w, b)
    y.backward(gy)
    got = (y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    y2 = torch.nn.functional.linear(x, w, b)
    y2.backward(gy)
    for name, a_, r_ in zip(("y", "gx", "gw", "gb"), got, (y2.detach(), x.grad, w.grad, b.grad)):
        scale = float(r_.abs().max())
        assert float((a_ - r_).abs().max()) <= 3e-5 * scale + 1e-6, name


@pytest.mark.parametrize("B,T,K,N", [(64, 37, 250, 500), (64, 100, 1250, 1000), (64, 100, 800, 250), (64, 36, 1800, 250), (8, 5, 253, 130),
                                     (3, 7, 36, 20), (64, 40, 1088, 1000)])
def test_linear_x3_fused_dropout_mask(B, T, K, N):
    """ops.linear(x, w, b, mask=) folds the variational-dropout multiply (one mask row per batch row, shared over time) into
    the forward GEMM's operand loads (a_scale), the weight gradient's (b_scale) and the input gradient's epilogue (c_scale): same
    values and gradients as multiplying first - at the trunk's shapes (16- and 8-byte mask loads, both tile sizes, split K) and at
    an odd width (scalar loads)."""
    from ruart_amd import ops
    assert ops.fuse_operand_masks
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, T, K, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(N, K, generator=g) * 0.05).cuda().requires_grad_(True)
    b = torch.randn(N, generator=g).cuda().requires_grad_(True)
    mask = (torch.bernoulli(torch.full((B, K), 0.7), generator=g) * (1.0 / 0.7)).cuda()
    mask.keep, mask.keep_scale = (mask != 0).view(torch.uint8), 1.0 / 0.7         # the byte form layers.MaskBank attaches
    gy = torch.randn(B, T, N, generator=g).cuda()
    y = ops.linear(x, w, b, mask=mask)
    y.backward(gy)
    got = (y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    y2 = torch.nn.functional.linear(x * mask.unsqueeze(1), w, b)
    y2.backward(gy)
    for name, a_, r_ in zip(("y", "gx", "gw", "gb"), got, (y2.detach(), x.grad, w.grad, b.grad)):
        scale = float(r_.abs().max())
        assert float((a_ - r_).abs().max()) <= 3e-5 * scale + 1e-6, name
    assert bool((got[1][mask.unsqueeze(1).expand_as(x) == 0] == 0).all())      # dropped features get exactly zero gradient
    # a mask without the byte form takes the multiply-first path: same numbers
    x.grad = w.grad = b.grad = None
    plain = mask.clone()
    y3 = ops.linear(x, w, b, mask=plain)
    y3.backward(gy)
    assert float((y3.detach() - got[0]).abs().max()) <= 1e-5 * float(got[0].abs().max())
    assert float((w.grad - got[2]).abs().max()) <= 1e-5 * float(got[2].abs().max())


@pytest.mark.parametrize("V,D,n,pad", [(2000, 300, 17000, 1), (51, 12, 17000, None), (75, 8, 2560, None), (300, 300, 40, 1), (10, 1000, 500, None),
                                       (64, 768, 43000, None), (30522, 768, 43000, None)])
def test_embedding_backward_from_host_sort(V, D, n, pad):
    """ops.embedding: forward = table lookup; backward = ruart_embedding_bwd_sorted driven by the host-side sort of the ids
    (batch._sort_ids): must equal nn.Embedding's gradient (padding row zero), for wide word tables, narrow POS / entity tables
    with thousands of hits per row (the two-level form: ruart_embedding_bwd_split), the trainable encoder's position table (64 rows,
    ~670 hits each) and word-piece table ([CLS] / [SEP]: 8 600 hits each), and rows that are never hit."""
    from ruart_amd import ops
    from ruart_amd.batch import _sort_ids
    g = torch.Generator().manual_seed(V + D + n)
    emb = torch.nn.Embedding(V, D, padding_idx=pad).cuda()
    ids = torch.randint(0, V, (n,), generator=g)
    if pad is not None:
        ids[::7] = pad
    if V > 20000:
        ids[::5], ids[1::5] = 101, 102
    sort = tuple(torch.from_numpy(a).cuda() for a in _sort_ids(ids.numpy(), pad))
    assert len(sort) == (4 if n // V > 64 or V > 20000 else 3)
    gy = torch.randn(n, D, generator=g).cuda()
    out = ops.embedding(emb, ids.cuda(), sort)
    out.backward(gy)
    got = emb.weight.grad.clone()
    emb.weight.grad = None
    ref_out = emb(ids.cuda())
    ref_out.backward(gy)
    assert torch.equal(out, ref_out)
    ref = emb.weight.grad
    assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    if pad is not None:
        assert float(got[pad].abs().max()) == 0.0
    out2 = ops.embedding(emb, ids.cuda(), sort)                      # deterministic: same sum order every time
    emb.weight.grad = None
    out2.backward(gy)
    assert torch.equal(emb.weight.grad, got)


@pytest.mark.parametrize("rows,cols", [(6400, 1000), (2304, 250), (65, 7), (12841, 1200), (40, 300)])
def test_colsum_f32(rows, cols):
    """ops.colsum: bias gradients of the trunk's projections (ruart_colsum_f32; small inputs stay on torch)."""
    from ruart_amd import ops
    g = torch.Generator().manual_seed(rows + cols)
    x = torch.randn(rows, cols + 3, generator=g).cuda()[:, :cols]                  # a row stride that is not the width
    got = ops.colsum(x)
    ref = x.double().sum(0)
    assert float((got.double() - ref).abs().max()) < 1e-5 * float(x.abs().double().sum(0).max())
    assert torch.equal(got, ops.colsum(x))                                          # ordered: the same bits again


def test_phoc_table_matches_reference(golden_dir):
    """ruart_phoc_table (one launch for the whole word list) bit-exact against the reference's build_phoc rows, against the
    oracle on fresh random words, and the error contract for bytes outside [a-z0-9]."""
    from ruart_amd.phoc import PHOC_DIM, build_phoc, normalize, phoc_table
    z = np.load(os.path.join(golden_dir, "phoc.npz"))
    words = str(z["words"]).split("\n")
    ref = np.unpackbits(z["bits"], axis=1)[:, :PHOC_DIM].astype(np.float32)
    got = phoc_table(words, dev()).cpu().numpy()
    assert got.shape == ref.shape and np.array_equal(got, ref), [w for w, a, b in zip(words, got, ref) if not np.array_equal(a, b)][:5]
    g = np.random.default_rng(77)
    alpha = "abcdefghijklmnopqrstuvwxyz0123456789"
    fresh = ["".join(alpha[int(k)] for k in g.integers(0, 36, size=int(g.integers(1, 90)))) for _ in range(3000)]
    got = phoc_table(fresh, dev(), normalized=True).cpu().numpy()
    for w, row in zip(fresh[:400], got):
        assert np.array_equal(row, O.build_phoc_raw(w)), w
    assert set(np.unique(got)) <= {0.0, 1.0}
    assert build_phoc("Hello-42") == O.build_phoc("Hello-42") and normalize("  Café-7 ") == "caf7"
    assert phoc_table([], dev()).shape == (0, PHOC_DIM)
    with pytest.raises(RuntimeError):
        phoc_table(["ok", "not-ok"], dev(), normalized=True)          # the reference's raw builder raises on '-'
    with pytest.raises(Exception):
        phoc_table(["a"], "cpu")


@pytest.mark.parametrize("rows,N,K", [(43008, 768, 768), (5000, 2304, 768), (777, 256, 3072), (64, 768, 768)])
def test_weight_gradient_bf16_tn(rows, N, K):
    """ruart_gemm_bf16_tn: dW = dY^T . X straight from row-major dY and X, one bf16 product, fp32 accumulation, split along the
    row dimension; against the same product of the bf16-rounded operands in float64 (tight) and of the fp32 operands (bf16-level)."""
    from ruart_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(rows + N)
    gy, x = torch.randn(rows, N, generator=g) * 0.01, torch.randn(rows, K, generator=g)
    got = ops._dw_bf16(gy.to(d), x.to(d)).cpu().double()
    ref16 = gy.bfloat16().double().t() @ x.bfloat16().double()
    ref = gy.double().t() @ x.double()
    scale = float(ref.abs().max())
    assert float((got - ref16).abs().max()) < 2e-5 * scale + 1e-7 * rows ** 0.5
    assert float((got - ref).abs().max()) < 2e-2 * scale


@pytest.mark.parametrize("layout", ["nt", "nn", "tn", "tt"])
def test_gemm_x1_layouts(layout):
    """ops.mm(mode='x1') -> ruart_gemm_x1: one bf16 product per term with fp32 accumulation, in all four operand layouts, against
    the product of the bf16-rounded operands in float64."""
    from ruart_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(11)
    M, N, K = 700, 500, 1100
    a = torch.randn(K, M, generator=g).t() if layout[0] == "t" else torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g).t() if layout[1] == "t" else torch.randn(K, N, generator=g)
    got = ops.mm(a.to(d), b.to(d), mode="x1").cpu().double()
    ref = a.bfloat16().double() @ b.bfloat16().double()
    assert float((got - ref).abs().max()) < 3e-5 * float(ref.abs().max()) + 1e-4
    full = ops.mm(a.to(d), b.to(d), mode="x3").cpu().double()
    assert float((full - a.double() @ b.double()).abs().max()) < 2e-4 * float(ref.abs().max())      # the three-product form stays fp32-class


def test_priority_streams():
    """ruart_stream_create_priority: the LOW level torch's stream pool lacks (the trunk's streams beside the CU-masked encoder pass,
    SDNet.trunk_stream_priority) - granted levels are ordered high < normal < low, work enqueued on such a stream runs and joins."""
    from ruart_amd import hip
    d = torch.device("cuda:0")
    hi, no, lo = (hip.priority_stream(p, d) for p in (-1, 0, 1))
    assert hi._ruart_level <= no._ruart_level <= lo._ruart_level and lo._ruart_level > hi._ruart_level
    x = torch.arange(1 << 20, device=d, dtype=torch.float32)
    lo.wait_stream(torch.cuda.current_stream(d))
    with torch.cuda.stream(lo):
        y = (x * 2).sum()
    torch.cuda.current_stream(d).wait_stream(lo)
    assert float(y) == float((1 << 20) * ((1 << 20) - 1))
    for st in (hi, no, lo):
        hip.destroy_stream(st)


@pytest.mark.gpu
@pytest.mark.parametrize("long_rows", [False, True])
def test_attention_split_heads_per_workgroup_bit_identical(long_rows):
    """ruart_bert_attention_split (fp16c mode: fp32 Q/K/V rows in, f16 + 2 x e4m3 context rows out): the multi-head workgroup forms
    (next head's loads in flight, context rows written one step late) produce the one-head kernel's bits for every heads-per-workgroup
    setting - ragged windows of whole short sequences, and 64-query blocks of long sequences (several key tiles) - and the one-head
    kernel itself agrees with a float64 softmax(QK^T)V on the window's keys."""
    from ruart_amd.bert import PackedTokens
    lib = hip.load()
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    lens = [3, 5, 8, 1, 64, 17, 2, 9, 33, 4, 6, 7, 12, 30, 5, 5] * 6
    if long_rows:
        lens = lens[:20] + [200, 130, 65, 512]
    L = max(lens)
    ids = torch.zeros(len(lens), L, dtype=torch.int64)
    mask = torch.zeros(len(lens), L, dtype=torch.bool)
    for i, n in enumerate(lens):
        ids[i, :n] = torch.randint(5, 1000, (n,), generator=g)
        mask[i, :n] = True
    p = PackedTokens([(ids, mask)], d, mfma_long=False)          # long rows as 64-query blocks of the short-window kernel
    H, NH = 768, 12
    T, Tp, nb = p.T, p.Tp, p.n_blocks
    qkv = (torch.randn(Tp, 3 * H, generator=g) * 1.2).to(d)
    outs = {}
    try:
        for hpg in (0, 2, 3, 4, 6, 12):
            hip.check(lib.ruart_bert_attention_split_set_heads(hpg), "set_heads")
            c16 = torch.zeros(Tp, H, dtype=torch.float16, device=d)
            c8 = torch.zeros(Tp, 2 * H, dtype=torch.uint8, device=d)
            hip.check(lib.ruart_bert_attention_split(hip.ptr(qkv), 3 * H, hip.ptr(c16), hip.ptr(c8), H, H, NH, nb, hip.ptr(p.blk[0]), hip.ptr(p.blk[1]),
                                                     hip.ptr(p.blk[2]), hip.ptr(p.blk[3]), hip.ptr(p.tok_lo), hip.ptr(p.tok_hi), None, hip.stream_ptr()),
                      "ruart_bert_attention_split")
            torch.cuda.synchronize()
            outs[hpg] = (c16[:T].cpu(), c8[:T].cpu())
    finally:
        hip.check(lib.ruart_bert_attention_split_set_heads(2), "set_heads")
    for hpg, (a16, a8) in outs.items():
        assert torch.equal(a16.view(torch.int16), outs[0][0].view(torch.int16)), hpg
        assert torch.equal(a8, outs[0][1]), hpg
    # the one-head kernel against float64 (Q is stored pre-scaled: the kernel applies no 1/sqrt(d))
    x = qkv[:T].double().cpu()
    lo, hi = p.tok_lo.cpu().numpy(), p.tok_hi.cpu().numpy()
    ref = torch.zeros(T, H, dtype=torch.float64)
    for s0 in sorted(set(lo.tolist())):
        s1 = int(hi[s0])
        for h in range(NH):
            q, k, v = (x[s0:s1, i * H + h * 64:i * H + (h + 1) * 64] for i in range(3))
            ref[s0:s1, h * 64:(h + 1) * 64] = torch.softmax(q @ k.t(), 1) @ v
    err = (outs[0][0].double() - ref).abs().max().item()
    assert err < 2e-3 * max(1.0, ref.abs().max().item()), err          # f16 storage of the context rows


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(64, 100, 40, 250, 250, "vec"), (64, 36, 40, 250, 250, "vec"), (8, 100, 100, 250, 250, "vec"),
                                   (4, 224, 40, 300, 300, "one"), (64, 100, 36, 125, 250, "one"), (3, 37, 53, 70, 33, None),
                                   (2, 17, 128, 64, 64, "vec"), (2, 5, 7, 3, 5, None)])
def test_sdnet_attention_prefetch_forms_bit_identical(shape):
    """ruart_attn_fwd / _bwd: the register-prefetched kernels (round 5; next chunk's loads under this chunk's MFMAs) against the staged
    forms of rounds 1-4 - same bits in the context rows, the probabilities and every gradient, over the step's shapes (deep attention,
    self attention, pre-align with its scalar diagonal, the 125-wide object / position attention) and ragged odd ones."""
    from ruart_amd import ops
    B, L1, L2, h, D3, diag_kind = shape
    lib = hip.load()
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    a = torch.randn(B, L1, h, generator=g).to(d).requires_grad_(True)
    k = torch.randn(B, L2, h, generator=g).to(d).requires_grad_(True)
    v = torch.randn(B, L2, D3, generator=g).to(d).requires_grad_(True)
    mask = (torch.rand(B, L2, generator=g) > 0.2).to(torch.uint8)
    mask[:, 0] = 1
    mask = mask.to(d)
    diag = None
    if diag_kind == "vec":
        diag = torch.randn(1, 1, h, generator=g).to(d).requires_grad_(True)
    elif diag_kind == "one":
        diag = torch.full((1, 1, 1), 0.06, device=d)
    gout = torch.randn(B, L1, D3, generator=g).to(d)
    res = {}
    try:
        for on in (0, 1):
            hip.check(lib.ruart_attn_set_prefetch(on), "ruart_attn_set_prefetch")
            for t in (a, k, v, diag):
                if t is not None:
                    t.grad = None
            y = ops.fused_attention(a, k, v, mask, diag=diag, relu=diag is not None)
            y.backward(gout)
            torch.cuda.synchronize()
            res[on] = [y.detach().clone(), a.grad.clone(), k.grad.clone(), v.grad.clone()] + (
                [diag.grad.clone()] if diag is not None and diag.requires_grad else [])
    finally:
        hip.check(lib.ruart_attn_set_prefetch(1), "ruart_attn_set_prefetch")
    for t0, t1 in zip(res[0], res[1]):
        assert torch.equal(t0.view(torch.int32), t1.view(torch.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("W,N,D", [(12841, 6400, 1388), (3421, 2304, 300), (5, 3, 768), (1, 1, 4)])
def test_rows_scale_equals_gather_and_multiply(W, N, D):
    """ruart_rows_scale (the packed variational dropout, layers.row_dropout): x * mask[row_of] in one pass, forward and backward, bit
    for bit what torch's gather + multiply computes; a column-slice view of a wider gradient goes in without a copy."""
    from ruart_amd import ops
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(W, D, generator=g).to(d).requires_grad_(True)
    mask = ((torch.rand(N, D, generator=g) > 0.3).float() / 0.7).to(d)
    row_of = torch.sort(torch.randint(0, N, (W,), generator=g)).values.to(d)
    y = ops.rows_scale(x, mask, row_of)
    ref = x.detach() * mask[row_of]
    assert torch.equal(y.detach().view(torch.int32), ref.view(torch.int32))
    wide = torch.randn(W, D + 8, generator=g).to(d)
    gy = wide[:, 4:4 + D]                                   # strided rows, 16-byte aligned start
    y.backward(gy)
    assert torch.equal(x.grad.view(torch.int32), (gy * mask[row_of]).view(torch.int32))


def _row_partials(y):
    """[M][4][2] (sum, sumsq) over 256-column tiles of y (fp64 sums), the layout of ruart_gemm_16c_nt_fold"""
    M, H = y.shape
    p = torch.zeros(M, 4, 2, dtype=torch.float64)
    for t in range(H // 256):
        blk = y[:, t * 256:(t + 1) * 256].double()
        p[:, t, 0] = blk.sum(1)
        p[:, t, 1] = (blk * blk).sum(1)
    return p.float()


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(256, 768, 768), (512, 1024, 1024), (256, 768, 3072)])
@pytest.mark.parametrize("res_ln", [False, True])
def test_gemm_16c_fold_producer(M, N, K, res_ln):
    """kind 3 of ruart_gemm_16c_nt_fold: y = A W^T + b + residual (the residual rows normalised from their partials when res_ln),
    written fp32, in the split operand form and as row partials - against fp64."""
    from ruart_amd.bert import split_f16c
    lib = hip.load()
    d = dev()
    g = torch.Generator().manual_seed(M + N + K + int(res_ln))
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.03
    bias = torch.randn(N, generator=g) * 0.1
    R = torch.randn(M, N, generator=g) * 1.5 + 0.2
    gam, bet = 1.0 + 0.3 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g)
    ref = A.double() @ W.double().t() + bias.double()
    if res_ln:
        mu = R.double().mean(1, keepdim=True)
        var = ((R.double() - mu) ** 2).mean(1, keepdim=True)
        ref = ref + (R.double() - mu) / torch.sqrt(var + 1e-12) * gam.double() + bet.double()
    else:
        ref = ref + R.double()
    A16, A8 = split_f16c(A)
    W16, W8 = _w8(W)
    dv = lambda t: t.to(d)
    A16d, A8d, W16d, W8d, bd, Rd, gd, bed = map(dv, (A16, A8, W16, W8, bias, R, gam, bet))
    rp = dv(_row_partials(R)) if res_ln else None
    C = torch.zeros(M, N, dtype=torch.float32, device=d)
    C16 = torch.zeros(M, N, dtype=torch.float16, device=d)
    C8 = torch.zeros(M, 2 * N, dtype=torch.uint8, device=d)
    part = torch.full((M, 4, 2), -7.0, dtype=torch.float32, device=d)
    rc = lib.ruart_gemm_16c_nt_fold(hip.ptr(A16d), hip.ptr(A8d), K, hip.ptr(W16d), hip.ptr(W8d), K, hip.ptr(bd), 3, None, 0, None, 1.0,
                                    hip.ptr(Rd), N, hip.ptr(rp), N // 256, hip.ptr(gd), hip.ptr(bed), hip.ptr(C), N, hip.ptr(C16), hip.ptr(C8),
                                    hip.ptr(part), M, N, K, N, 1e-12, hip.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    scale = float(ref.abs().mean())
    assert maxerr(C, ref) < 6e-5 * scale * max(1.0, (K / 768) ** 0.5) * 4
    h16, h8 = split_f16c(C.cpu())
    assert torch.equal(h16, C16.cpu()) and torch.equal(h8, C8.cpu())            # the split rows are the split of the fp32 rows written
    want = _row_partials(C.cpu())
    got = part.cpu()
    nt = N // 256
    assert maxerr(got[:, :nt, 0], want[:, :nt, 0]) < 2e-4 * max(1.0, float(want[:, :nt, 0].abs().max()))
    assert maxerr(got[:, :nt, 1], want[:, :nt, 1]) < 2e-5 * float(want[:, :nt, 1].abs().max())
    if nt < 4:
        assert float((got[:, nt:] + 7.0).abs().max()) == 0.0                   # unused slots untouched


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1024, 2304, 768), (768, 3072, 768), (512, 768, 3072)])
@pytest.mark.parametrize("gelu", [False, True])
@pytest.mark.parametrize("fold", [False, True])
def test_gemm_16c_dual_form_is_bitwise_the_256x256_form(M, N, K, gelu, fold):
    """ruart_gemm_16c_set_dual(1): the QKV / intermediate products on 256 x 128 tiles with two resident workgroups per CU
    (gemm_16c_nt_256x128d) accumulate every output element's products in the order of the 256 x 256 kernel: outputs bit for bit equal,
    plain and LayerNorm-folded, fp32 and GELU + split epilogues; K = 128 is the shortest K loop (two K-tiles per run: prologue and the two
    closing tiles only)."""
    from ruart_amd.bert import split_f16c
    if fold and not 256 <= K <= 1024:
        pytest.skip("row partials: one to four 256-column tiles")
    lib = hip.load()
    d = dev()
    g = torch.Generator().manual_seed(M + N + K + int(gelu) + 2 * int(fold))
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.03
    bias, colc = torch.randn(N, generator=g), torch.randn(N, generator=g)
    A16, A8 = split_f16c(A)
    W16, W8 = _w8(W)
    A16d, A8d, W16d, W8d, bd, cd = [t.to(d) for t in (A16, A8, W16, W8, bias, colc)]
    pd = _row_partials(A).to(d) if fold else None
    outs = []
    try:
        for dual in (0, 1):
            lib.ruart_gemm_16c_set_dual(dual)
            C = torch.full((M, N), 3.0, dtype=torch.float16 if gelu else torch.float32, device=d)
            C8 = torch.full((M, 2 * N), 1, dtype=torch.uint8, device=d) if gelu else None
            if fold:
                rc = lib.ruart_gemm_16c_nt_fold(hip.ptr(A16d), hip.ptr(A8d), K, hip.ptr(W16d), hip.ptr(W8d), K, hip.ptr(bd), 2 if gelu else 0, hip.ptr(pd),
                                                max(K // 256, 1), hip.ptr(cd), 2.0, None, 0, None, 0, None, None, hip.ptr(C), N, None, hip.ptr(C8), None,
                                                M, N, K, K, 1e-12, hip.stream_ptr())
            else:
                rc = lib.ruart_gemm_16c_nt(hip.ptr(A16d), hip.ptr(A8d), K, hip.ptr(W16d), hip.ptr(W8d), K, hip.ptr(bd), None, 0, hip.ptr(C), N,
                                           hip.ptr(C8), M, N, K, hip.ACT_GELU if gelu else hip.ACT_NONE, hip.stream_ptr())
            assert rc == 0, rc
            torch.cuda.synchronize()
            outs.append((C, C8))
    finally:
        lib.ruart_gemm_16c_set_dual(0)
    assert bool(torch.isfinite(outs[1][0].float()).all())
    assert torch.equal(outs[0][0], outs[1][0])
    assert not gelu or torch.equal(outs[0][1], outs[1][1])


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(256, 2304, 768), (512, 3072, 768), (256, 4096, 1024)])
@pytest.mark.parametrize("gelu", [False, True])
def test_gemm_16c_fold_consumer(M, N, K, gelu):
    """kind 0 / 2 of ruart_gemm_16c_nt_fold on pre-LayerNorm rows y with folded weights: rstd 2^s (y W'^T - mu c) + d against
    LayerNorm(y) W^T + b in fp64 (gamma with a few large gains, so that the fold needs s > 0), and against the unfolded fp16c product
    of the materialised LayerNorm rows: the folded form must be about as accurate."""
    from ruart_amd.bert import split_f16c
    lib = hip.load()
    d = dev()
    g = torch.Generator().manual_seed(M + N + K + int(gelu))
    y = torch.randn(M, K, generator=g) * (0.5 + 2.0 * torch.rand(M, 1, generator=g)) + 0.3 * torch.randn(M, 1, generator=g)
    y[:, 5] += 20.0                                                               # an outlier dimension, as a pretrained encoder has
    W = torch.randn(N, K, generator=g) * 0.03
    bias = torch.randn(N, generator=g) * 0.1
    gam, bet = 1.0 + 0.2 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    gam[7] = 40.0                                                                 # |W'| reaches ~4: s = 1
    mu = y.double().mean(1, keepdim=True)
    var = ((y.double() - mu) ** 2).mean(1, keepdim=True)
    x = (y.double() - mu) / torch.sqrt(var + 1e-12) * gam.double() + bet.double()
    ref = x @ W.double().t() + bias.double()
    if gelu:
        ref = O.gelu_erf(ref)
    wf = W * gam[None, :]
    sh = 0
    while float(wf.abs().max()) * 2.0 ** -sh >= 3.4:
        sh += 1
    assert sh >= 1
    wf = wf * 2.0 ** -sh
    dvec = (bias.double() + W.double() @ bet.double()).float()
    cvec = wf.double().sum(1).float()
    A16, A8 = split_f16c(y)
    W16, W8 = _w8(wf)
    dv = lambda t: t.to(d)
    A16d, A8d, W16d, W8d, dd, cd, pd = map(dv, (A16, A8, W16, W8, dvec, cvec, _row_partials(y)))
    if gelu:
        C = torch.zeros(M, N, dtype=torch.float16, device=d)
        C8 = torch.zeros(M, 2 * N, dtype=torch.uint8, device=d)
    else:
        C = torch.zeros(M, N, dtype=torch.float32, device=d)
        C8 = None
    rc = lib.ruart_gemm_16c_nt_fold(hip.ptr(A16d), hip.ptr(A8d), K, hip.ptr(W16d), hip.ptr(W8d), K, hip.ptr(dd), 2 if gelu else 0, hip.ptr(pd),
                                    K // 256, hip.ptr(cd), float(2.0 ** sh), None, 0, None, 0, None, None, hip.ptr(C), N, None, hip.ptr(C8), None,
                                    M, N, K, K, 1e-12, hip.stream_ptr())
    assert rc == 0
    # the unfolded product of the materialised rows
    X16, X8 = split_f16c(x.float())
    V16, V8 = _w8(W)
    U = torch.zeros(M, N, dtype=torch.float32, device=d)
    # (named device copies: a temporary inside hip.ptr(...) is freed as soon as its address is taken, and the caching allocator may hand
    # the block to the next temporary of the same call - the product then reads another operand's bytes)
    X16d, X8d, V16d, V8d, biasd = map(dv, (X16, X8, V16, V8, bias))
    rc = lib.ruart_gemm_16c_nt(hip.ptr(X16d), hip.ptr(X8d), K, hip.ptr(V16d), hip.ptr(V8d), K, hip.ptr(biasd), None, 0,
                               hip.ptr(U), N, None, M, N, K, hip.ACT_NONE, hip.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    if gelu:
        sa_lo, _, _, _ = hip.f16c_shifts()
        got = C.float().cpu() + C8[:, :N].view(torch.float8_e4m3fn).float().cpu() / 2.0 ** sa_lo
        ref_u = O.gelu_erf(U.double().cpu())
    else:
        got, ref_u = C.cpu(), U.double().cpu()
    e_f = float((got.double() - ref).pow(2).mean().sqrt())
    e_u = float((ref_u - ref).pow(2).mean().sqrt())
    print("fold rms err %.2e (max %.2e) | unfolded fp16c rms %.2e | mean |ref| %.2e" % (e_f, maxerr(got, ref), e_u, float(ref.abs().mean())))
    assert e_f < 4 * e_u + 1e-6 and maxerr(got, ref) < 2e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K", [(256, 768, 768), (512, 1024, 1024), (256, 768, 3072)])
@pytest.mark.parametrize("res_ln", [False, True])
def test_gemm_16_fold_producer(M, N, K, res_ln, dt):
    """kind 3 of ruart_gemm_16_nt_fold (round 6: the plain 16-bit pass's attention-output / output dense with the LayerNorm behind it
    folded away): y = A W^T + b + residual (16-bit rows, normalised from their partials when res_ln) written in 16 bits, and the row
    partials of the UNROUNDED y - against fp64 on the operands as the kernel reads them."""
    lib = hip.load()
    d = dev()
    code, td = (hip.DT_F16, torch.float16) if dt == "fp16" else (hip.DT_BF16, torch.bfloat16)
    g = torch.Generator().manual_seed(M + N + K + int(res_ln))
    A = torch.randn(M, K, generator=g).to(td)
    W = (torch.randn(N, K, generator=g) * 0.03).to(td)
    bias = torch.randn(N, generator=g) * 0.1
    R = (torch.randn(M, N, generator=g) * 1.5 + 0.2).to(td)
    gam, bet = 1.0 + 0.3 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g)
    ref = A.double() @ W.double().t() + bias.double()
    if res_ln:
        mu = R.double().mean(1, keepdim=True)
        var = ((R.double() - mu) ** 2).mean(1, keepdim=True)
        ref = ref + (R.double() - mu) / torch.sqrt(var + 1e-12) * gam.double() + bet.double()
    else:
        ref = ref + R.double()
    dv = lambda t: t.to(d)
    Ad, Wd, bd, Rd, gd, bed = map(dv, (A, W, bias, R, gam, bet))
    rp = dv(_row_partials(R.float())) if res_ln else None
    C = torch.zeros(M, N, dtype=td, device=d)
    part = torch.full((M, 4, 2), -7.0, dtype=torch.float32, device=d)
    rc = lib.ruart_gemm_16_nt_fold(hip.ptr(Ad), K, hip.ptr(Wd), K, hip.ptr(bd), 3, None, 0, None, 1.0, hip.ptr(Rd), N, hip.ptr(rp), N // 256,
                                   hip.ptr(gd), hip.ptr(bed), hip.ptr(C), N, hip.ptr(part), M, N, K, N, 1e-12, code, hip.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    ulp = 2.0 ** -11 if dt == "fp16" else 2.0 ** -8
    assert maxerr(C.float(), ref) < 1.5 * ulp * float(ref.abs().max())              # y rounded once to 16 bits
    want = _row_partials(ref.float())
    got = part.cpu()
    nt = N // 256
    assert maxerr(got[:, :nt, 0], want[:, :nt, 0]) < 5e-4 * max(1.0, float(want[:, :nt, 0].abs().max()))       # the partials come from the fp32 y
    assert maxerr(got[:, :nt, 1], want[:, :nt, 1]) < 5e-5 * float(want[:, :nt, 1].abs().max())
    if nt < 4:
        assert float((got[:, nt:] + 7.0).abs().max()) == 0.0                       # unused slots untouched


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K", [(256, 2304, 768), (512, 3072, 768), (256, 4096, 1024)])
@pytest.mark.parametrize("gelu", [False, True])
def test_gemm_16_fold_consumer(M, N, K, gelu, dt):
    """kind 0 / 2 of ruart_gemm_16_nt_fold on 16-bit pre-LayerNorm rows y with folded 16-bit weights: rstd (y W'^T - mu c) + d against
    LayerNorm(y) W^T + b in fp64, and against the plain 16-bit product of the materialised LayerNorm rows: about as accurate."""
    lib = hip.load()
    d = dev()
    code, td = (hip.DT_F16, torch.float16) if dt == "fp16" else (hip.DT_BF16, torch.bfloat16)
    g = torch.Generator().manual_seed(M + N + K + int(gelu))
    y = (torch.randn(M, K, generator=g) * (0.5 + 2.0 * torch.rand(M, 1, generator=g)) + 0.3 * torch.randn(M, 1, generator=g)).to(td)
    W = torch.randn(N, K, generator=g) * 0.03
    bias = torch.randn(N, generator=g) * 0.1
    gam, bet = 1.0 + 0.2 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    mu = y.double().mean(1, keepdim=True)
    var = ((y.double() - mu) ** 2).mean(1, keepdim=True)
    x = (y.double() - mu) / torch.sqrt(var + 1e-12) * gam.double() + bet.double()
    ref = x @ W.double().t() + bias.double()
    if gelu:
        ref = O.gelu_erf(ref)
    wf = (W * gam[None, :]).to(td)                                                  # W' as the matrix cores multiply it
    dvec = (bias.double() + W.double() @ bet.double()).float()
    cvec = wf.double().sum(1).float()
    dv = lambda t: t.to(d)
    yd, wfd, dd, cd, pd = map(dv, (y, wf, dvec, cvec, _row_partials(y.float())))
    C = torch.zeros(M, N, dtype=td, device=d)
    rc = lib.ruart_gemm_16_nt_fold(hip.ptr(yd), K, hip.ptr(wfd), K, hip.ptr(dd), 2 if gelu else 0, hip.ptr(pd), K // 256, hip.ptr(cd), 1.0, None, 0,
                                   None, 0, None, None, hip.ptr(C), N, None, M, N, K, K, 1e-12, code, hip.stream_ptr())
    assert rc == 0
    # the unfolded product of the materialised rows
    xd, Wd, biasd = dv(x.to(td)), dv(W.to(td)), dv(bias)
    U = torch.zeros(M, N, dtype=td, device=d)
    rc = lib.ruart_gemm_16_nt(hip.ptr(xd), K, hip.ptr(Wd), K, hip.ptr(biasd), None, 0, code, hip.ptr(U), N, code, M, N, K,
                              hip.ACT_GELU if gelu else hip.ACT_NONE, code, hip.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    e_f = float((C.double().cpu() - ref).pow(2).mean().sqrt())
    e_u = float((U.double().cpu() - ref).pow(2).mean().sqrt())
    print("fold rms err %.2e | unfolded %s rms %.2e | mean |ref| %.2e" % (e_f, dt, e_u, float(ref.abs().mean())))
    assert e_f < 2.5 * e_u + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("NL,reg", [(5, 1), (12, 1), (12, 0)])
def test_pool_mix_over_prelayernorm_rows(NL, reg):
    """ruart_bert_pool_mix_ln / _bwd (the folded pass's pooling: rows normalised on the fly) == ruart_bert_pool_mix / _bwd over the
    materialised LayerNorm rows, last layer compacted.  NL = 12 takes the kernels that keep gamma / beta in registers over four words
    per workgroup (reg = 0: switched off)."""
    lib = hip.load()
    lib.ruart_bert_pool_ln_set_variant(reg)
    d = dev()
    g = torch.Generator().manual_seed(11)
    Tp, H, W = 512, 768, 301
    y = (torch.randn(NL, Tp, H, generator=g) * 1.3 + 0.2).to(d)
    gam, bet = (1.0 + 0.2 * torch.randn(NL, H, generator=g)).to(d), (0.1 * torch.randn(NL, H, generator=g)).to(d)
    mu = y.mean(2, keepdim=True)
    rstd = 1.0 / torch.sqrt(((y - mu) ** 2).mean(2, keepdim=True) + 1e-12)
    stats = torch.cat([mu, rstd], 2).contiguous()
    x = ((y - mu) * rstd * gam[:, None, :] + bet[:, None, :]).contiguous()
    n = torch.randint(1, 4, (W,), generator=g).int()
    n[:7] = torch.tensor([1, 2, 3, 4, 1, 2, 5]).int()
    st = torch.randint(0, Tp - 8, (W,), generator=g).int()
    st_last = torch.randint(0, 200, (W,), generator=g).int()
    dst = torch.randperm(W + 20, generator=g)[:W].int()
    lw = torch.randn(NL, generator=g)
    gy = torch.randn(W + 20, H, generator=g)
    std, nd, stl, dstd, lwd, gyd = st.to(d), n.to(d), st_last.to(d), dst.to(d), lw.to(d), gy.to(d)
    o1 = torch.zeros(W + 20, H, device=d)
    o2 = torch.zeros(W + 20, H, device=d)
    assert lib.ruart_bert_pool_mix(hip.ptr(x), Tp * H, H, hip.DT_F32, NL, hip.ptr(std), hip.ptr(stl), hip.ptr(nd), hip.ptr(dstd), hip.ptr(lwd),
                                   hip.ptr(o1), H, W, H, hip.stream_ptr()) == 0
    assert lib.ruart_bert_pool_mix_ln(hip.ptr(y), Tp * H, H, NL, hip.ptr(stats), Tp, hip.ptr(gam), hip.ptr(bet), hip.ptr(std), hip.ptr(stl),
                                      hip.ptr(nd), hip.ptr(dstd), hip.ptr(lwd), hip.ptr(o2), H, W, H, hip.stream_ptr()) == 0
    p1, p2 = torch.empty(W * NL, device=d), torch.empty(W * NL, device=d)
    g1, g2 = torch.zeros(NL, device=d), torch.zeros(NL, device=d)
    assert lib.ruart_bert_pool_mix_bwd(hip.ptr(x), Tp * H, H, hip.DT_F32, NL, hip.ptr(std), hip.ptr(stl), hip.ptr(nd), hip.ptr(dstd), hip.ptr(gyd),
                                       H, hip.ptr(p1), hip.ptr(g1), W, H, hip.stream_ptr()) == 0
    assert lib.ruart_bert_pool_mix_ln_bwd(hip.ptr(y), Tp * H, H, NL, hip.ptr(stats), Tp, hip.ptr(gam), hip.ptr(bet), hip.ptr(std), hip.ptr(stl),
                                          hip.ptr(nd), hip.ptr(dstd), hip.ptr(gyd), H, hip.ptr(p2), hip.ptr(g2), W, H, hip.stream_ptr()) == 0
    torch.cuda.synchronize()
    lib.ruart_bert_pool_ln_set_variant(1)
    assert maxerr(o2, o1) < 2e-5 * max(1.0, float(o1.abs().max()))
    assert maxerr(g2, g1) < 1e-4 * max(1.0, float(g1.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("M,N", [(256, 3072), (128, 256), (43008, 3072)])
def test_gelu_bwd_rows(M, N):
    """ruart_gelu_bwd_rows (the trainable encoder's GELU backward as an elementwise pass): d *= gelu'(h) in place, g = gelu(h), 128-row
    strip column sums of the unrounded d - against torch in fp64 (erf GELU; the kernel's logistic-polynomial form is 3.4e-6 off)."""
    lib = hip.load()
    d = dev()
    g = torch.Generator().manual_seed(M + N)
    h = (torch.randn(M, N, generator=g) * 1.5).to(torch.float16)
    acc = (torch.randn(M, N, generator=g) * 0.01).to(torch.bfloat16)
    hd, dd = h.to(d), acc.to(d).clone()
    gb = torch.zeros(M, N, dtype=torch.bfloat16, device=d)
    part = torch.zeros(M // 128, N, dtype=torch.float32, device=d)
    assert lib.ruart_gelu_bwd_rows(hip.ptr(dd), hip.ptr(hd), N, hip.ptr(gb), hip.ptr(part), M, N, hip.stream_ptr()) == 0
    torch.cuda.synchronize()
    x = h.double()
    cdf = 0.5 * (1.0 + torch.erf(x / 2.0 ** 0.5))
    gref = x * cdf
    dref = acc.double() * (cdf + x * torch.exp(-0.5 * x * x) / (2.0 * np.pi) ** 0.5)
    assert maxerr(gb.float(), gref) < 2.0 ** -8 * float(gref.abs().max())                 # bf16 rounding of the output
    assert maxerr(dd.float(), dref) < 2.0 ** -8 * float(dref.abs().max()) + 1e-6
    pref = dref.view(M // 128, 128, N).sum(1)
    assert maxerr(part, pref) < 2e-5 * 128 ** 0.5 + 1e-4 * float(pref.abs().max())

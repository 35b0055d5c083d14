"""Kernels of the trainable encoder's 16-bit path (csrc/bert_train_kernels.hip, bert_train_attn.hip, the split-K form of the
encoder GEMM) through the C ABI against plain torch fp32 / autograd references of the same ops (Models/Bert/modeling.py:155-168
LayerNorm, :52-57 GELU, :224-250 self-attention, :260-264 / :299-303 dense -> dropout -> residual -> LayerNorm).

Tolerances: activations travel in f16 (11 significant bits) and GEMM-bound gradients in bf16 (8 bits); every comparison states its
bound next to the assertion."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ruart_amd import hip                      # noqa: E402

DEV = "cuda:0"


def _st():
    return hip.stream_ptr()


def _ln_fwd(lib, x, res, gamma, beta, p, seed, post):
    R, H = x.shape
    y = torch.empty(R, H, dtype=torch.float16, device=DEV)
    pre = torch.empty(R, H, dtype=torch.float16, device=DEV)
    stats = torch.empty(R, 2, dtype=torch.float32, device=DEV)
    rc = lib.ruart_ln_train_fwd(hip.ptr(x), H, hip.ptr(res), H, hip.ptr(gamma), hip.ptr(beta), 1e-12, float(p), int(seed), int(post),
                                hip.ptr(y), hip.ptr(pre), hip.ptr(stats), H, R, H, _st())
    assert rc == 0
    return y, pre, stats


def _ln_bwd(lib, dy, pre, stats, gamma, p, seed, post, add=None, add_scale=None):
    R, H = dy.shape
    d_res = torch.empty(R, H, dtype=torch.float32, device=DEV)
    d_gemm = torch.empty(R, H, dtype=torch.bfloat16, device=DEV)
    dg, db, dbias = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    ws = torch.empty(int(lib.ruart_ln_train_bwd_ws_floats(H)), device=DEV)
    rc = lib.ruart_ln_train_bwd(hip.ptr(dy), H, hip.ptr(add), hip.ptr(add_scale), hip.ptr(pre), H, hip.ptr(stats), hip.ptr(gamma), float(p),
                                int(seed), int(post), hip.ptr(d_res), H, hip.ptr(d_gemm), H, hip.ptr(dg), hip.ptr(db), hip.ptr(dbias), 0,
                                hip.ptr(ws), R, H, _st())
    assert rc == 0
    if not post:                                # the dense layer's bias gradient = column sums of what goes to d_gemm, before its rounding
        cs = d_gemm.float().sum(0)
        assert float((dbias - cs).abs().max()) < 4e-3 * float(d_gemm.float().abs().sum(0).max()) + 1e-6
    return d_res, d_gemm, dg, db


@pytest.mark.parametrize("R,H", [(37, 768), (1030, 1024), (5, 128)])
def test_ln_train_fwd_bwd_vs_autograd(R, H):
    lib = hip.load()
    g = torch.Generator().manual_seed(R + H)
    x = torch.randn(R, H, generator=g).to(DEV)
    res = torch.randn(R, H, generator=g).half().to(DEV)
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(H, generator=g)).to(DEV)
    dy = torch.randn(R, H, generator=g).to(DEV)
    add = torch.randn(R, H, generator=g).to(DEV)
    a = torch.tensor([0.37], device=DEV)
    y, pre, stats = _ln_fwd(lib, x, res, gamma, beta, 0.0, 1, 0)
    xr, gr, br = x.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    ref = torch.nn.functional.layer_norm(xr + res.float(), (H,), gr, br, 1e-12)
    assert float((y.float() - ref).abs().max()) < 4e-3                          # f16 output of O(1) values
    ref.backward(dy + 0.37 * add)
    d_res, d_gemm, dg, db = _ln_bwd(lib, dy, pre, stats, gamma, 0.0, 1, 0, add=add, add_scale=a)
    scale = float(xr.grad.abs().max())
    assert float((d_res - xr.grad).abs().max()) < 3e-3 * scale                  # the LayerNorm input is re-read from its f16 copy
    assert float((d_gemm.float() - xr.grad).abs().max()) < 1e-2 * scale         # bf16 copy
    assert float((dg - gr.grad).abs().max()) < 3e-3 * float(gr.grad.abs().max()) + 1e-3
    assert float((db - br.grad).abs().max()) < 1e-4 * float(br.grad.abs().max()) + 1e-4


def test_ln_train_dropout_masks_regenerate():
    """pre-LN dropout (post = 0) and the embeddings' post-LN dropout (post = 1): the backward regenerates the forward's mask from the
    seed; a different seed gives a different mask; the kept fraction is 1 - p."""
    lib = hip.load()
    R, H, p = 300, 768, 0.1
    gamma, beta = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
    x = torch.full((R, H), 2.0, device=DEV)
    _, pre, stats = _ln_fwd(lib, x, None, gamma, beta, p, 77, 0)
    mask = pre.float() / 2.0                                                     # 0 or 1 / (1 - p)
    kept = float((mask > 0).float().mean())
    assert abs(kept - (1 - p)) < 0.01 and float((mask[mask > 0] - 1 / (1 - p)).abs().max()) < 2e-3
    x2 = torch.randn(R, H, device=DEV)
    _, pre2, stats2 = _ln_fwd(lib, x2, None, gamma, beta, p, 77, 0)
    dy = torch.randn(R, H, device=DEV)
    d_res, d_gemm, _, _ = _ln_bwd(lib, dy, pre2, stats2, gamma, p, 77, 0)
    assert float((d_gemm.float() - d_res * mask).abs().max()) < 1e-2 * float(d_res.abs().max()) * 1.2     # same mask, bf16 rounding
    _, pre3, _ = _ln_fwd(lib, x, None, gamma, beta, p, 78, 0)
    assert float(((pre3 > 0) != (pre > 0)).float().mean()) > 0.1                # another stream
    # post = 1: y = dropout(LN(x)); with gamma = 1, beta = 0 the kept entries equal the normalised input * 1/(1-p)
    y, pre4, stats4 = _ln_fwd(lib, x2, None, gamma, beta, p, 5, 1)
    ln = torch.nn.functional.layer_norm(x2, (H,), None, None, 1e-12)
    m = (y != 0)
    assert abs(float(m.float().mean()) - (1 - p)) < 0.01
    assert float((y.float()[m] - ln[m] / (1 - p)).abs().max()) < 5e-3
    d_in, _, _, _ = _ln_bwd(lib, dy, pre4, stats4, gamma, p, 5, 1)
    xr = x2.clone().requires_grad_()
    (torch.nn.functional.layer_norm(xr, (H,), None, None, 1e-12) * m.float() / (1 - p) * dy).sum().backward()
    assert float((d_in - xr.grad).abs().max()) < 3e-3 * float(xr.grad.abs().max())


def test_colsum_transpose_cast_mix():
    lib = hip.load()
    g = torch.Generator().manual_seed(3)
    R, N = 1000, 3072
    h = (2 * torch.randn(R, N, generator=g)).half().to(DEV)
    dg = torch.randn(R, N, generator=g).bfloat16().to(DEV)
    hb = torch.empty(R, N, dtype=torch.bfloat16, device=DEV)
    assert lib.ruart_f16_to_bf16(hip.ptr(h), hip.ptr(hb), R * N, _st()) == 0
    assert torch.equal(hb, h.float().bfloat16())
    # column sums (bias gradients)
    ws = torch.empty(((R + 255) // 256) * N, device=DEV)
    cs = torch.zeros(N, device=DEV)
    assert lib.ruart_colsum_bf16(hip.ptr(dg), N, R, N, hip.ptr(cs), 0, hip.ptr(ws), _st()) == 0
    assert float((cs - dg.float().sum(0)).abs().max()) < 1e-3 * float(dg.float().sum(0).abs().max()) + 1e-3
    assert lib.ruart_colsum_bf16(hip.ptr(dg), N, R, N, hip.ptr(cs), 1, hip.ptr(ws), _st()) == 0
    assert float((cs - 2 * dg.float().sum(0)).abs().max()) < 2e-3 * float(dg.float().sum(0).abs().max()) + 2e-3
    # transpose
    t = torch.empty(N, 1024, dtype=torch.float16, device=DEV).zero_()
    assert lib.ruart_transpose16(hip.ptr(h), N, hip.ptr(t), 1024, R, N, 0, _st()) == 0
    assert torch.equal(t[:, :R], h.t()) and float(t[:, R:].abs().max()) == 0.0
    tb = torch.empty(N, 1024, dtype=torch.bfloat16, device=DEV).zero_()
    assert lib.ruart_transpose16(hip.ptr(h), N, hip.ptr(tb), 1024, R, N, 1, _st()) == 0         # f16 -> bf16 on the way
    assert torch.equal(tb[:, :R], h.float().bfloat16().t())
    # layer mix and its weight gradient
    NL, H = 12, 768
    layers = torch.randn(NL, R, H, generator=g).half().to(DEV)
    w = torch.randn(NL, generator=g).to(DEV)
    mixed = torch.empty(R, H, device=DEV)
    assert lib.ruart_mix_rows(hip.ptr(layers), R * H, H, NL, hip.ptr(w), hip.ptr(mixed), H, R, H, _st()) == 0
    ref = (layers.float() * w.view(-1, 1, 1)).sum(0)
    assert float((mixed - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    gm = torch.randn(R, H, generator=g).to(DEV)
    dw = torch.empty(NL, device=DEV)
    ws2 = torch.empty(512 * NL, device=DEV)
    assert lib.ruart_mix_rows_bwd(hip.ptr(layers), R * H, H, NL, hip.ptr(gm), H, hip.ptr(dw), hip.ptr(ws2), R, H, _st()) == 0
    refw = (layers.float() * gm.unsqueeze(0)).sum((1, 2))
    assert float((dw - refw).abs().max()) < 1e-4 * float(refw.abs().max()) + 1e-2


def test_gelu_backward_in_the_gemm_epilogue():
    """ruart_gemm_16_nt_gelu_bwd: dH = (dY . W2) * gelu'(H), G = gelu(H) and the per-strip column sums of dH, one kernel."""
    lib = hip.load()
    g = torch.Generator().manual_seed(21)
    M, N, K = 512, 768, 256
    dY = (torch.randn(M, K, generator=g) * 1e-3).bfloat16().to(DEV)
    Wt = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(DEV)                    # W2^T: (intermediate, hidden)
    Hh = (2 * torch.randn(M, N, generator=g)).half().to(DEV)
    dH = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    G = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    part = torch.empty(int(lib.ruart_gemm_16_nt_gelu_bwd_ws_floats(M, N)), device=DEV)
    assert part.numel() == (M // 128) * N
    assert lib.ruart_gemm_16_nt_gelu_bwd(hip.ptr(dY), K, hip.ptr(Wt), K, hip.ptr(Hh), N, hip.ptr(dH), hip.ptr(G), N, hip.ptr(part), M, N, K, _st()) == 0
    hr = Hh.double().cpu().requires_grad_()
    ref_g = torch.nn.functional.gelu(hr)
    acc = dY.double().cpu() @ Wt.double().cpu().t()
    ref_g.backward(acc)
    ref_dh = hr.grad
    assert float((G.double().cpu() - ref_g.detach()).abs().max()) < 1e-2 * float(ref_g.abs().max())            # bf16 output
    assert float((dH.double().cpu() - ref_dh).abs().max()) < 1e-2 * float(ref_dh.abs().max())
    db = torch.empty(N, device=DEV)
    assert lib.ruart_colsum_f32_rows(hip.ptr(part), M // 128, N, N, hip.ptr(db), 0, _st()) == 0
    assert float((db.double().cpu() - ref_dh.sum(0)).abs().max()) < 2e-4 * float(ref_dh.abs().sum(0).max())     # unrounded fp32 sums
    assert lib.ruart_gemm_16_nt_gelu_bwd(hip.ptr(dY), K, hip.ptr(Wt), K, hip.ptr(Hh), N, hip.ptr(dH), hip.ptr(G), N, None, M, N, K, _st()) == 0
    assert lib.ruart_gemm_16_nt_gelu_bwd(hip.ptr(dY), K, hip.ptr(Wt), K, None, N, hip.ptr(dH), hip.ptr(G), N, None, M, N, K, _st()) != 0


def test_weight_prep_f16_and_transposed_bf16():
    lib = hip.load()
    g = torch.Generator().manual_seed(5)
    H = 200
    W = torch.randn(3 * H, H + 8, generator=g).to(DEV)                                  # three (H x H) weights inside wider storage
    w16 = torch.zeros(3 * H, H, dtype=torch.float16, device=DEV)
    wT = torch.zeros(H, 3 * H, dtype=torch.bfloat16, device=DEV)
    for i, sc in enumerate((0.125, 1.0, 1.0)):
        assert lib.ruart_weight_prep(hip.ptr(W[i * H:]), H + 8, sc, hip.ptr(w16[i * H:]), H, hip.ptr(wT[:, i * H:]), 3 * H, H, H, _st()) == 0
    ref = W[:, :H].clone()
    ref[:H] *= 0.125
    assert torch.equal(w16, ref.half()) and torch.equal(wT, ref.bfloat16().t())
    assert lib.ruart_weight_prep(hip.ptr(W), H + 8, 1.0, None, 0, None, 0, H, H, _st()) != 0
    # the batch form: the three pieces and a fourth, non-square weight in one launch
    W2 = torch.randn(70, 333, generator=g).to(DEV)
    b16 = torch.zeros(3 * H, H, dtype=torch.float16, device=DEV)
    bT = torch.zeros(H, 3 * H, dtype=torch.bfloat16, device=DEV)
    o16, oT = torch.zeros(70, 333, dtype=torch.float16, device=DEV), torch.zeros(333, 70, dtype=torch.bfloat16, device=DEV)
    items = (hip.WPrepItemC * 4)()
    for i, sc in enumerate((0.125, 1.0, 1.0)):
        items[i] = hip.WPrepItemC(W[i * H:].data_ptr(), b16[i * H:].data_ptr(), bT[:, i * H:].data_ptr(), H + 8, H, 3 * H, H, H, sc)
    items[3] = hip.WPrepItemC(W2.data_ptr(), o16.data_ptr(), oT.data_ptr(), 333, 333, 70, 70, 333, 2.0)
    assert lib.ruart_weight_prep_batch(items, 4, _st()) == 0
    assert torch.equal(b16, w16) and torch.equal(bT, wT)
    assert torch.equal(o16, (2 * W2).half()) and torch.equal(oT, (2 * W2).bfloat16().t())
    assert lib.ruart_weight_prep_batch(items, 9, _st()) != 0


def test_intermediate_dense_keeps_preactivation():
    """ruart_gemm_16_nt_gelu2: one product, two outputs - H = A . W^T + b and G = gelu(H) (Models/Bert/modeling.py:287-288)."""
    lib = hip.load()
    g = torch.Generator().manual_seed(12)
    M, N, K = 512, 768, 256
    A = torch.randn(M, K, generator=g).half().to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.1).half().to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    Hh = torch.empty(M, N, dtype=torch.float16, device=DEV)
    G = torch.empty(M, N, dtype=torch.float16, device=DEV)
    assert lib.ruart_gemm_16_nt_gelu2(hip.ptr(A), K, hip.ptr(W), K, hip.ptr(b), hip.ptr(Hh), hip.ptr(G), N, M, N, K, hip.DT_F16, _st()) == 0
    ref = A.double().cpu() @ W.double().cpu().t() + b.double().cpu()
    assert float((Hh.double().cpu() - ref).abs().max()) < 2e-3 * float(ref.abs().max())
    refg = torch.nn.functional.gelu(ref)
    assert float((G.double().cpu() - refg).abs().max()) < 2e-3 * float(refg.abs().max())
    assert lib.ruart_gemm_16_nt_gelu2(hip.ptr(A), K, hip.ptr(W), K, hip.ptr(b), None, hip.ptr(G), N, M, N, K, hip.DT_F16, _st()) != 0


@pytest.mark.parametrize("M,N,K,kchunk", [(768, 768, 43008, 1536), (2304, 768, 5120, 2048), (256, 3072, 1280, 512)])
def test_weight_gradient_splitk(M, N, K, kchunk):
    """dW = dY^T . X as a split-K NT product of the transposed 16-bit operands, slabs summed by ruart_splitk_reduce."""
    lib = hip.load()
    g = torch.Generator().manual_seed(M + K)
    rows = K - 100                                                                # K = token rows padded to 256; the pad rows are zero
    dY = torch.zeros(K, M).bfloat16()
    X = torch.zeros(K, N).bfloat16()
    dY[:rows] = (torch.randn(rows, M, generator=g) * 1e-3).bfloat16()
    X[:rows] = torch.randn(rows, N, generator=g).bfloat16()
    dYd, Xd = dY.to(DEV), X.to(DEV)
    dYt = torch.empty(M, K, dtype=torch.bfloat16, device=DEV)
    Xt = torch.empty(N, K, dtype=torch.bfloat16, device=DEV)
    assert lib.ruart_transpose16(hip.ptr(dYd), M, hip.ptr(dYt), K, K, M, 0, _st()) == 0
    assert lib.ruart_transpose16(hip.ptr(Xd), N, hip.ptr(Xt), K, K, N, 0, _st()) == 0
    nz = (K + kchunk - 1) // kchunk
    part = torch.empty(nz, M, N, device=DEV)
    assert lib.ruart_gemm_16_nt_splitk(hip.ptr(dYt), K, hip.ptr(Xt), K, hip.ptr(part), N, M, N, K, kchunk, hip.DT_BF16, _st()) == 0
    dW = torch.full((M, N), 7.0, device=DEV)
    assert lib.ruart_splitk_reduce(hip.ptr(part), M * N, nz, hip.ptr(dW), M * N, 0.5, 0, _st()) == 0
    ref = 0.5 * (dY.double().t() @ X.double())
    assert float((dW.double().cpu() - ref).abs().max()) < 2e-5 * float(ref.abs().max()) + 1e-7       # exact bf16 operands, fp32 sums
    assert lib.ruart_splitk_reduce(hip.ptr(part), M * N, nz, hip.ptr(dW), M * N, 0.5, 1, _st()) == 0  # accumulate form
    assert float((dW.double().cpu() - 2 * ref).abs().max()) < 4e-5 * float(ref.abs().max()) + 1e-7
    # the TN kernel takes the same operands as they lie (no transposes) and must agree with the slabs of the NT form
    part2 = torch.empty(nz, M, N, device=DEV)
    assert lib.ruart_gemm_16_tn_splitk(hip.ptr(dYd), M, hip.ptr(Xd), N, hip.ptr(part2), N, M, N, K, kchunk, hip.DT_BF16, _st()) == 0
    err = float((part2.double() - part.double()).abs().max())
    assert err < 2e-5 * float(part.abs().max()), err                                # same products, another order within a K = 32 step
    dW2 = torch.empty(M, N, device=DEV)
    assert lib.ruart_splitk_reduce(hip.ptr(part2), M * N, nz, hip.ptr(dW2), M * N, 0.5, 0, _st()) == 0
    assert float((dW2.double().cpu() - ref).abs().max()) < 2e-5 * float(ref.abs().max()) + 1e-7
    # f16 operands, strided rows (a column block of a wider matrix)
    Xw = torch.zeros(K, N + 256).half()
    Xw[:rows] = torch.randn(rows, N + 256, generator=g).half()
    dYh = (dY.float() * 1024).half().to(DEV)
    Xwd = Xw.to(DEV)
    assert lib.ruart_gemm_16_tn_splitk(hip.ptr(dYh), M, hip.ptr(Xwd[:, 256:]), N + 256, hip.ptr(part2), N, M, N, K, kchunk, hip.DT_F16, _st()) == 0
    ref2 = (dYh.double().cpu().t() @ Xw[:, 256:].double())
    got2 = part2.double().sum(0).cpu()
    assert float((got2 - ref2).abs().max()) < 2e-5 * float(ref2.abs().max()) + 1e-7


def _attn_case(g, lens, heads):
    """packed tokens of the given sequence lengths, windows of whole sequences (<= 64 tokens)"""
    H = heads * 64
    T = sum(lens)
    qkv = torch.randn(T, 3 * H, generator=g)
    qkv[:, :H] *= 0.125 * 3                                                        # queries arrive pre-scaled; *3: peaked rows too
    cu = np.concatenate([[0], np.cumsum(lens)])
    q0, q1, s = [], [], 0
    while s < len(lens):
        e = s
        while e < len(lens) and cu[e + 1] - cu[s] <= 64:
            e += 1
        q0.append(cu[s]); q1.append(cu[e]); s = e
    tok_lo = np.repeat(cu[:-1], lens)
    return qkv, torch.tensor(q0, dtype=torch.int32), torch.tensor(q1, dtype=torch.int32), torch.tensor(tok_lo, dtype=torch.int32), cu


def _attn_ref(qkv, cu, heads):
    H = heads * 64
    T = qkv.shape[0]
    out = torch.zeros(T, H, dtype=qkv.dtype)
    for a, b in zip(cu[:-1], cu[1:]):
        q, k, v = [qkv[a:b, i * H:(i + 1) * H].view(b - a, heads, 64).transpose(0, 1) for i in range(3)]
        p = torch.softmax(q @ k.transpose(1, 2), -1)
        out[a:b] = (p @ v).transpose(0, 1).reshape(b - a, H)
    return out


def test_attention_train_fwd_bwd_vs_autograd():
    lib = hip.load()
    g = torch.Generator().manual_seed(11)
    heads = 3
    H = heads * 64
    lens = [5, 1, 64, 3, 8, 30, 30, 7, 50, 2, 2, 2, 63]
    qkv, q0, q1, tok_lo, cu = _attn_case(g, lens, heads)
    T = qkv.shape[0]
    q16 = qkv.half()
    qd, q0d, q1d, lod = q16.to(DEV), q0.to(DEV), q1.to(DEV), tok_lo.to(DEV)
    ctx = torch.zeros(T, H, dtype=torch.float16, device=DEV)
    assert lib.ruart_attn_train_fwd(hip.ptr(qd), 3 * H, hip.ptr(ctx), H, H, heads, len(q0), hip.ptr(q0d), hip.ptr(q1d), hip.ptr(lod), 0.0, 0, _st()) == 0
    xr = q16.double().requires_grad_()
    ref = _attn_ref(xr, cu, heads)
    assert float((ctx.double().cpu() - ref).abs().max()) < 4e-3                    # f16 operands / output, O(1) values
    dO = torch.randn(T, H, generator=g) * 1e-3
    dOb = dO.bfloat16()
    ref.backward(dOb.double())
    dqkv = torch.zeros(T, 3 * H, dtype=torch.bfloat16, device=DEV)
    bpart = torch.full((len(q0), 2 * H), float("nan"), device=DEV)
    assert lib.ruart_attn_train_bwd(hip.ptr(qd), 3 * H, hip.ptr(dOb.to(DEV)), H, hip.ptr(dqkv), 3 * H, H, heads, len(q0), hip.ptr(q0d), hip.ptr(q1d),
                                    hip.ptr(lod), 0.0, 0, hip.ptr(bpart), _st()) == 0
    got, want = dqkv.double().cpu(), xr.grad
    # the windows' column sums of dQ and dV (bias gradients), taken before the bf16 rounding
    bsum = bpart.double().sum(0).cpu()
    for name, sl, bs in (("dQ", slice(0, H), bsum[:H]), ("dV", slice(2 * H, 3 * H), bsum[H:])):
        ref_b = want[:, sl].sum(0)
        assert float((bs - ref_b).abs().max()) < 2e-3 * float(want[:, sl].abs().sum(0).max()), name
    for name, sl in (("dQ", slice(0, H)), ("dK", slice(H, 2 * H)), ("dV", slice(2 * H, 3 * H))):
        e = float((got[:, sl] - want[:, sl]).abs().max()) / float(want[:, sl].abs().max())
        rel = float((got[:, sl] - want[:, sl]).norm() / want[:, sl].norm())
        assert e < 4e-2 and rel < 1.5e-2, (name, e, rel)                           # bf16 operands: 8 significant bits


def test_attention_train_dropout_is_consistent():
    """With probability dropout the output is linear in V for a fixed mask: <dO, O> == <dV, V> holds exactly when the backward
    regenerates the forward's mask; the share of dropped probabilities is p."""
    lib = hip.load()
    g = torch.Generator().manual_seed(12)
    heads, p = 2, 0.1
    H = heads * 64
    qkv, q0, q1, tok_lo, cu = _attn_case(g, [40, 24, 64, 9, 33, 31], heads)
    T = qkv.shape[0]
    # values that are exact in bf16 and f16, so that the two kernels see the same V
    qkv[:, 2 * H:] = torch.randint(-4, 5, (T, H), generator=g).float() / 4
    qd, q0d, q1d, lod = qkv.half().to(DEV), q0.to(DEV), q1.to(DEV), tok_lo.to(DEV)
    ctx = torch.zeros(T, H, dtype=torch.float16, device=DEV)
    ctx0 = torch.zeros(T, H, dtype=torch.float16, device=DEV)
    args = (H, heads, len(q0), hip.ptr(q0d), hip.ptr(q1d), hip.ptr(lod))
    assert lib.ruart_attn_train_fwd(hip.ptr(qd), 3 * H, hip.ptr(ctx), H, *args, p, 1234, _st()) == 0
    assert lib.ruart_attn_train_fwd(hip.ptr(qd), 3 * H, hip.ptr(ctx0), H, *args, 0.0, 1234, _st()) == 0
    assert float((ctx.float() - ctx0.float()).abs().max()) > 1e-2                 # the mask does something
    dO = (torch.randint(-4, 5, (T, H), generator=g).float() / 64).bfloat16()
    dqkv = torch.zeros(T, 3 * H, dtype=torch.bfloat16, device=DEV)
    assert lib.ruart_attn_train_bwd(hip.ptr(qd), 3 * H, hip.ptr(dO.to(DEV)), H, hip.ptr(dqkv), 3 * H, *args, p, 1234, None, _st()) == 0
    lhs = float((dO.double() * ctx.double().cpu()).sum())
    rhs = float((dqkv[:, 2 * H:].double().cpu() * qkv[:, 2 * H:].double()).sum())
    assert abs(lhs - rhs) < 2e-2 * max(abs(lhs), 1e-3), (lhs, rhs)                 # same mask on both sides (f16 / bf16 rounding only)
    dqkv2 = torch.zeros_like(dqkv)
    assert lib.ruart_attn_train_bwd(hip.ptr(qd), 3 * H, hip.ptr(dO.to(DEV)), H, hip.ptr(dqkv2), 3 * H, *args, p, 99, None, _st()) == 0
    rhs2 = float((dqkv2[:, 2 * H:].double().cpu() * qkv[:, 2 * H:].double()).sum())
    assert abs(lhs - rhs2) > 5 * abs(lhs - rhs)                                    # another seed: another mask


def _long_plan(lens):
    """(windows, chunks) of a packed stream as bert.PackedTokens.train_plan cuts it: whole short sequences in <= 64-token windows,
    every longer sequence in chunks of <= 64 tokens that attend to the whole sequence"""
    from ruart_amd.bert import PackedTokens
    lens = np.asarray(lens, dtype=np.int64)
    cu = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=cu[1:])
    blk, _ = PackedTokens._plan_blocks(lens, cu, mfma_long=False)
    own = (blk[0] == blk[2]) & (blk[1] == blk[3])
    win = blk[:2, own]
    ch = blk[:, ~own]
    first = np.array([int(np.nonzero(ch[0] == k0)[0][0]) for k0 in ch[2]], dtype=np.int32)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.int32)).to(DEV)
    return cu, [to(win[0]), to(win[1])], [to(ch[0]), to(ch[1]), to(ch[2]), to(ch[3]), to(first)], torch.from_numpy(np.repeat(cu[:-1], lens).astype(np.int32)).to(DEV)


def _attn_long_run(lib, qd, dOd, heads, cu, win, ch, tok_lo, p, seed):
    """forward + backward of a packed stream through the window kernels (short sequences) and the *_long kernels (the others)"""
    T, H = qd.shape[0], heads * 64
    ctx = torch.zeros(T, H, dtype=torch.float16, device=DEV)
    lse = torch.full((T, heads), float("nan"), device=DEV)
    nw, nc = win[0].numel(), ch[0].numel()
    if nw:
        assert lib.ruart_attn_train_fwd(hip.ptr(qd), 3 * H, hip.ptr(ctx), H, H, heads, nw, hip.ptr(win[0]), hip.ptr(win[1]), hip.ptr(tok_lo), p, seed, _st()) == 0
    assert lib.ruart_attn_train_fwd_long(hip.ptr(qd), 3 * H, hip.ptr(ctx), H, H, heads, nc, hip.ptr(ch[0]), hip.ptr(ch[1]), hip.ptr(ch[2]), hip.ptr(ch[3]),
                                         p, seed, hip.ptr(lse), _st()) == 0
    dqkv = torch.zeros(T, 3 * H, dtype=torch.bfloat16, device=DEV)
    bw = torch.zeros(max(nw, 1), 2 * H, device=DEV)
    bl = torch.full((nc, 2 * H), float("nan"), device=DEV)
    if nw:
        assert lib.ruart_attn_train_bwd(hip.ptr(qd), 3 * H, hip.ptr(dOd), H, hip.ptr(dqkv), 3 * H, H, heads, nw, hip.ptr(win[0]), hip.ptr(win[1]),
                                        hip.ptr(tok_lo), p, seed, hip.ptr(bw), _st()) == 0
    delta = torch.empty(T, heads, device=DEV)
    scale = torch.empty(nc, heads, device=DEV)
    assert lib.ruart_attn_train_bwd_long(hip.ptr(qd), 3 * H, hip.ptr(dOd), H, hip.ptr(dqkv), 3 * H, H, heads, nc, hip.ptr(ch[0]),
                                         hip.ptr(ch[1]), hip.ptr(ch[2]), hip.ptr(ch[3]), hip.ptr(ch[4]), p, seed, hip.ptr(lse), hip.ptr(delta),
                                         hip.ptr(scale), hip.ptr(bl), _st()) == 0
    torch.cuda.synchronize()
    return ctx, dqkv, bw.double().sum(0) + bl.double().sum(0)


def test_attention_train_long_sequences_vs_autograd():
    """Sequences of 65 .. 512 word pieces (ruart_attn_train_fwd_long / _bwd_long: 64-token chunks against the whole sequence, online
    softmax, dQ and dK / dV in two launches) beside short ones (the window kernels), against a float64 autograd attention per
    sequence: context rows, [dQ | dK | dV] rows and the query / value bias sums.  Lengths cover a ragged last chunk (65, 130, 200), a
    chunk-aligned one (128) and the maximum (512)."""
    lib = hip.load()
    g = torch.Generator().manual_seed(21)
    heads = 2
    H = heads * 64
    lens = [5, 65, 64, 130, 3, 512, 30, 128, 200, 40]
    cu, win, ch, tok_lo = _long_plan(lens)
    assert win[0].numel() >= 2 and ch[0].numel() == 2 + 3 + 8 + 2 + 4
    T = int(cu[-1])
    qkv = torch.randn(T, 3 * H, generator=g)
    qkv[:, :H] *= 0.125 * 3
    q16 = qkv.half()
    dO = torch.randn(T, H, generator=g) * 1e-3
    dO[int(cu[5]):int(cu[5]) + 64] *= 37.0                       # chunks of one sequence with different dO scales
    dOb = dO.bfloat16()
    ctx, dqkv, bsum = _attn_long_run(lib, q16.to(DEV), dOb.to(DEV), heads, cu, win, ch, tok_lo, 0.0, 0)
    xr = q16.double().requires_grad_()
    ref = _attn_ref(xr, cu, heads)
    assert float((ctx.double().cpu() - ref).abs().max()) < 4e-3                    # f16 operands / output, O(1) values
    ref.backward(dOb.double())
    got, want = dqkv.double().cpu(), xr.grad
    bsum = bsum.cpu()
    for name, sl, bs in (("dQ", slice(0, H), bsum[:H]), ("dV", slice(2 * H, 3 * H), bsum[H:])):
        assert float((bs - want[:, sl].sum(0)).abs().max()) < 2e-3 * float(want[:, sl].abs().sum(0).max()), name
    for name, sl in (("dQ", slice(0, H)), ("dK", slice(H, 2 * H)), ("dV", slice(2 * H, 3 * H))):
        for a, b in zip(cu[:-1], cu[1:]):                                          # per sequence: the long ones must hold the same bound
            w_, g_ = want[a:b, sl], got[a:b, sl]
            e = float((g_ - w_).abs().max()) / float(w_.abs().max())
            rel = float((g_ - w_).norm() / w_.norm())
            assert e < 4e-2 and rel < 1.5e-2, (name, int(b - a), e, rel)           # bf16 results, f16 operands


def test_attention_train_long_matches_the_window_kernels_and_their_dropout_stream():
    """A 64-token sequence pushed through the *_long kernels as ONE chunk must agree with the window kernels - same probabilities, the
    SAME dropout mask (the hash is indexed by query token and key offset in both) - and with dropout <dO, O> == <dV, V> holds across
    the chunks of a long sequence (the backward regenerates the forward's mask in both of its launches)."""
    lib = hip.load()
    g = torch.Generator().manual_seed(22)
    heads, p, seed = 2, 0.1, 4321
    H = heads * 64
    lens = [64, 50, 300]
    cu = np.concatenate([[0], np.cumsum(lens)])
    T = int(cu[-1])
    qkv = torch.randn(T, 3 * H, generator=g)
    qkv[:, :H] *= 0.125 * 2
    qkv[:, 2 * H:] = torch.randint(-4, 5, (T, H), generator=g).float() / 4          # exact in f16 and bf16
    qd = qkv.half().to(DEV)
    dO = (torch.randint(-4, 5, (T, H), generator=g).float() / 64).bfloat16()
    dOd = dO.to(DEV)
    to = lambda a: torch.tensor(a, dtype=torch.int32, device=DEV)
    tok_lo = torch.from_numpy(np.repeat(cu[:-1], lens).astype(np.int32)).to(DEV)
    # (a) the first two sequences as windows, the third as chunks; (b) all three as chunks of themselves
    win = [to([0, 64]), to([64, 114])]
    q0 = list(range(114, T, 64))
    ch_a = [to(q0), to([min(a + 64, T) for a in q0]), to([114] * len(q0)), to([T] * len(q0)), to([0] * len(q0))]
    ctx_a, dq_a, _ = _attn_long_run(lib, qd, dOd, heads, cu, win, ch_a, tok_lo, p, seed)
    allq = [0, 64] + q0
    ch_b = [to(allq), to([64, 114] + [min(a + 64, T) for a in q0]), to([0, 64] + [114] * len(q0)), to([64, 114] + [T] * len(q0)),
            to([0, 1] + [2] * len(q0))]
    ctx_b, dq_b, _ = _attn_long_run(lib, qd, dOd, heads, cu, [to([]), to([])], ch_b, tok_lo, p, seed)
    assert float((ctx_a.float() - ctx_b.float()).abs().max()) < 4e-3              # same mask: only the rounding point of P differs
    assert float((dq_a.float() - dq_b.float()).abs().max()) < 3e-2 * float(dq_a.float().abs().max())
    ctx0, _, _ = _attn_long_run(lib, qd, dOd, heads, cu, win, ch_a, tok_lo, 0.0, seed)
    assert float((ctx_a.float() - ctx0.float()).abs().max()) > 1e-2               # the mask does something
    lhs = float((dO.double() * ctx_a.double().cpu()).sum())
    rhs = float((dq_a[:, 2 * H:].double().cpu() * qkv[:, 2 * H:].double()).sum())
    assert abs(lhs - rhs) < 2e-2 * max(abs(lhs), 1e-3), (lhs, rhs)


def test_trainable_encoder_long_sequence_stays_off_the_vendor_gemm():
    """A sequence of 385..512 word pieces is beyond the fused attention kernel's key panel: the fp32-class trainable encoder serves
    it slice by slice on ruart_gemm_x3 (bert_train.py) - never torch.bmm - and agrees with a plain torch attention, values and
    gradients."""
    import torch.nn.functional as F
    from ruart_amd import synth
    from ruart_amd.bert import PackedTokens
    from ruart_amd.bert_train import BertModelTrainable
    dev = torch.device("cuda:0")
    cfg = synth.bert_config(vocab_size=300, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256,
                            hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = BertModelTrainable(synth.make_bert_weights(cfg, seed=3, w_std=0.05), cfg, dev)
    g = torch.Generator().manual_seed(0)
    L = 400
    ids = torch.randint(5, 300, (2, L), generator=g)
    mask = torch.ones(2, L, dtype=torch.bool)
    mask[1, 390:] = False
    ids[1, 390:] = 0
    packed = PackedTokens([(ids, mask)], dev, mfma_long=False)
    calls = {"bmm": 0}
    orig = torch.bmm
    torch.bmm = lambda *a, **k: (calls.__setitem__("bmm", calls["bmm"] + 1), orig(*a, **k))[1]
    try:
        out = m(packed, training=False)                        # (1, T, H)
        out.sum().backward()
    finally:
        torch.bmm = orig
    assert calls["bmm"] == 0
    assert out.shape == (1, packed.T, 128) and torch.isfinite(out).all()
    # reference: the same layer with torch ops on the padded layout
    P = {n: p.detach().double().cpu() for n, p in m._p.items()}
    x = (P["embeddings.word_embeddings.weight"][ids] + P["embeddings.position_embeddings.weight"][torch.arange(L)].unsqueeze(0)
         + P["embeddings.token_type_embeddings.weight"][0])
    ln = lambda t, pre: F.layer_norm(t, (128,), P[pre + ".gamma"], P[pre + ".beta"], 1e-12)
    x = ln(x, "embeddings.LayerNorm")
    pre = "encoder.layer.0."
    lin = lambda t, n: t @ P[pre + n + ".weight"].t() + P[pre + n + ".bias"]
    q, k, v = (lin(x, "attention.self." + n).view(2, L, 2, 64).transpose(1, 2) for n in ("query", "key", "value"))
    s = (q @ k.transpose(-1, -2)) / 8.0 + (~mask).double().view(2, 1, 1, L) * -1e9
    ctx = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(2, L, 128)
    x = ln(lin(ctx, "attention.output.dense") + x, pre + "attention.output.LayerNorm")
    h = F.gelu(lin(x, "intermediate.dense"))
    x = ln(lin(h, "output.dense") + x, pre + "output.LayerNorm")
    ref = torch.cat([x[0], x[1, :390]], 0)
    assert float((out[0].double().cpu() - ref).abs().max()) < 2e-4

"""autograd wrappers around the SDNet kernels of libruart_hip.so (fp32, device tensors only).

Each Function's forward AND backward launch hand-written HIP kernels through the C ABI - the dense projections included
(``mm`` / ``linear`` / ``addmm`` on ruart_gemm_x3).  No library GEMM runs inside a training or inference step of the default
modes: the vendor library's solutions for these shapes are all stream-K kernels (Tensile ``SK3``: a fixed grid of workgroups that
exchange partial tiles through flags in memory and spin on each other), which need every workgroup of the launch resident at the
same time; three of them launched on three streams beside the encoder's 256-workgroup GEMMs do not get that, and the device stops
making progress (the hang of round 1).  Only the exact-fp32 validation mode (``trunk_gemm == "fp32"``) still calls
torch.mm, and SDNet runs that mode on ONE stream.
There is no CPU path: a CPU tensor raises ``hip.HipError``.
"""
import ctypes
import os

import torch

from . import hip

_WS = {}
_PLAN_BYTES = {}      # (M, N, K, a K-contiguous, b K-contiguous) -> split-K workspace bytes of ruart_gemm_x3_plan
# Timing diagnostics only (tools/r04_abl.sh, r04_abl2.sh): RUART_ABL_SKIP=lstm,x3,attn (here), pool (bert.py), trunk (sdnet.py) leaves out the launches of a kernel class - outputs are
# zero-filled, results are WRONG - to read a class's marginal cost in the pipelined step.  Empty in every product run.
_ABL_SKIP = frozenset(x for x in os.environ.get("RUART_ABL_SKIP", "").split(",") if x)
_HOST_DELAY_US = float(os.environ.get("RUART_ABL_HOST_DELAY_US", 0) or 0)
if _ABL_SKIP or _HOST_DELAY_US:
    # a stray variable must not corrupt a real run silently: the knobs work only next to an explicit RUART_DIAGNOSTICS=1, and say so
    if os.environ.get("RUART_DIAGNOSTICS") != "1":
        raise RuntimeError("RUART_ABL_SKIP / RUART_ABL_HOST_DELAY_US are timing-only ablations that produce WRONG results; "
                           "set RUART_DIAGNOSTICS=1 beside them to confirm, or unset them")
    import sys as _sys
    print("ruart_amd: TIMING DIAGNOSTICS ACTIVE (RUART_ABL_SKIP=%s, RUART_ABL_HOST_DELAY_US=%g): outputs of this process are wrong"
          % (",".join(sorted(_ABL_SKIP)), _HOST_DELAY_US), file=_sys.stderr, flush=True)


def _scratch(device, n, tag=""):
    key = (device, hip.stream_handle(device), tag)      # per stream: branches run concurrently
    t = _WS.get(key)
    if t is None or t.numel() < n:
        t = torch.empty(max(n, 4096), dtype=torch.float32, device=device)
        _WS[key] = t
    return t


class _NanFlag:
    """Device-side flag honouring the reference's NaN asserts (Models/Layers.py:169,290,430,462,467) with ONE
    host sync per step instead of one per op."""

    def __init__(self):
        self.flag = None

    def ensure(self, device):
        if self.flag is None or self.flag.device != device:
            self.flag = torch.zeros(1, dtype=torch.int32, device=device)
            hip.check(hip.load().ruart_set_nan_flag(hip.ptr(self.flag)), "ruart_set_nan_flag")
        return self.flag

    def check_and_clear(self):
        if self.flag is None:
            return
        bad = int(self.flag.item())
        self.flag.zero_()
        assert bad == 0, "NaN produced inside the SDNet kernels (reference: assert torch.sum(torch.isnan(x)) == 0)"


nan_flag = _NanFlag()


# ---------------------------------------------------------------------------------------------------------
class _FusedAttention(torch.autograd.Function):
    """out = softmax_j(mask(a . k^T)) . v with a = act(pa) * diag, k = act(pk)   (Models/Layers.py:228-231, 244, 275-288)."""

    @staticmethod
    def forward(ctx, pa, pk, v, mask, diag, relu, pscale=None):
        lib = hip.load()
        for t in (pa, pk, v):
            hip.require_gpu(t, torch.float32)
        nan_flag.ensure(pa.device)
        B, L1, h = pa.shape
        L2, D3 = pk.shape[1], v.shape[2]
        dl = 0 if diag is None else diag.numel()
        out = torch.empty(B, L1, D3, dtype=torch.float32, device=pa.device)
        probs = torch.empty(B, L1, L2, dtype=torch.float32, device=pa.device)
        if "attn" in _ABL_SKIP:
            out.zero_()
            probs.zero_()
        rc = 0 if "attn" in _ABL_SKIP else lib.ruart_attn_fwd_pscale(hip.ptr(pa), hip.ptr(pk), hip.ptr(v), hip.ptr(mask), hip.ptr(diag), dl, int(relu),
                                       hip.ptr(pscale), hip.ptr(out), hip.ptr(probs), B, L1, L2, h, D3, hip.stream_ptr())
        hip.check(rc, "ruart_attn_fwd")
        ctx.save_for_backward(pa, pk, v, probs, diag, pscale)
        ctx.relu = int(relu)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = hip.load()
        pa, pk, v, probs, diag, pscale = ctx.saved_tensors
        B, L1, h = pa.shape
        L2, D3 = pk.shape[1], v.shape[2]
        dl = 0 if diag is None else diag.numel()
        gout = gout.contiguous()
        ga, gk, gv = torch.empty_like(pa), torch.empty_like(pk), torch.empty_like(v)
        ds = torch.empty_like(probs)
        gdiag = (torch.empty(B * ((L1 + 15) // 16), h, dtype=torch.float32, device=pa.device)
                 if (dl > 1 and ctx.needs_input_grad[4]) else None)
        if "attn" in _ABL_SKIP:
            for t in (ga, gk, gv, gdiag):
                if t is not None:
                    t.zero_()
        rc = 0 if "attn" in _ABL_SKIP else lib.ruart_attn_bwd_pscale(hip.ptr(pa), hip.ptr(pk), hip.ptr(v), hip.ptr(probs), hip.ptr(gout), hip.ptr(diag), dl, ctx.relu,
                                       hip.ptr(pscale), hip.ptr(ga), hip.ptr(gk), hip.ptr(gv), hip.ptr(gdiag), hip.ptr(ds), B, L1, L2,
                                       h, D3, hip.stream_ptr())
        hip.check(rc, "ruart_attn_bwd")
        return ga, gk, gv, None, (colsum(gdiag).view_as(diag) if gdiag is not None else None), None, None


def fused_attention(a, k, v, mask, diag=None, relu=False, prob_scale=None):
    """a (B,L1,h), k (B,L2,h), v (B,L2,D3) fp32; mask (B,L2) uint8/bool (0 = masked key).  With ``relu`` / ``diag`` the
    activation of the reference's AttentionScore is applied inside the kernel: a <- ReLU(a) * diag, k <- ReLU(k).
    ``prob_scale`` (B,L1,L2) fp32 of 0 / 1/(1-p): dropout on the attention probabilities (BERT)."""
    m = mask.to(torch.uint8).contiguous()
    d = None if diag is None else diag.contiguous().view(-1)
    ps = None if prob_scale is None else prob_scale.contiguous()
    return _FusedAttention.apply(a.contiguous(), k.contiguous(), v.contiguous(), m, d, relu, ps)


# ---------------------------------------------------------------------------------------------------------
class _FusedScorer(torch.autograd.Function):
    """probs (B, L + 1) = softmax([x_i . u(i), w . (softmax(mask(x . uh)) . x) + b])  - Models/Layers.py:352-432 in one launch per
    direction (csrc/sdnet_scorer.hip).  u1 scores the slots i >= ES, u2 the first ES; all three (B, D)."""

    @staticmethod
    def forward(ctx, x, u1, u2, uh, w, b, mask, ES, mask_flag):
        lib = hip.load()
        for t in (x, u1, u2, uh, w, b):
            hip.require_gpu(t, torch.float32)
        nan_flag.ensure(x.device)
        B, L, D = x.shape
        probs = torch.empty(B, L + 1, dtype=torch.float32, device=x.device)
        a = torch.empty(B, L, dtype=torch.float32, device=x.device)
        hip.check(lib.ruart_scorer_fwd(hip.ptr(x), hip.ptr(u1), hip.ptr(u2), hip.ptr(uh), hip.ptr(w), hip.ptr(b), hip.ptr(mask), hip.ptr(probs),
                                       hip.ptr(a), B, L, D, int(ES), int(bool(mask_flag)), hip.stream_ptr()), "ruart_scorer_fwd")
        ctx.save_for_backward(x, u1, u2, uh, w, probs, a)
        ctx.ES = int(ES)
        return probs

    @staticmethod
    def backward(ctx, gp):
        lib = hip.load()
        x, u1, u2, uh, w, probs, a = ctx.saved_tensors
        B, L, D = x.shape
        gp = gp.contiguous()
        gx = torch.empty_like(x)
        gu1, gu2, guh, gwp = (torch.empty(B, D, dtype=torch.float32, device=x.device) for _ in range(4))
        gbp = torch.empty(B, dtype=torch.float32, device=x.device)
        hip.check(lib.ruart_scorer_bwd(hip.ptr(x), hip.ptr(u1), hip.ptr(u2), hip.ptr(uh), hip.ptr(w), hip.ptr(probs), hip.ptr(a), hip.ptr(gp),
                                       hip.ptr(gx), hip.ptr(gu1), hip.ptr(gu2), hip.ptr(guh), hip.ptr(gwp), hip.ptr(gbp), B, L, D, ctx.ES,
                                       hip.stream_ptr()), "ruart_scorer_bwd")
        return gx, gu1, gu2, guh, colsum(gwp).view_as(w), gbp.sum().view(1), None, None, None


def fused_scorer(x, u1, u2, uh, w, b, mask, ES, mask_flag):
    """x (B, L, D), u1 / u2 / uh (B, D), w (D,) or (1, D), b (1,), mask (B, L): see _FusedScorer.  D % 4 == 0, L <= 1024."""
    return _FusedScorer.apply(x.contiguous(), u1.contiguous(), u2.contiguous(), uh.contiguous(), w.contiguous().view(-1), b.contiguous().view(1),
                              mask.to(torch.uint8).contiguous(), ES, mask_flag)


# ---------------------------------------------------------------------------------------------------------
class _WholeLayerNorm(torch.autograd.Function):
    """F.layer_norm(x, x.size()) - normalisation over the WHOLE tensor, no affine (Models/Layers.py:167-168)."""

    @staticmethod
    def forward(ctx, x, eps):
        lib = hip.load()
        hip.require_gpu(x, torch.float32)
        nan_flag.ensure(x.device)
        y = torch.empty_like(x)
        stats = torch.empty(2, dtype=torch.float32, device=x.device)
        ws = _scratch(x.device, 2048)
        hip.check(lib.ruart_whole_ln_fwd(hip.ptr(x), hip.ptr(y), hip.ptr(stats), hip.ptr(ws), x.numel(), eps, hip.stream_ptr()),
                  "ruart_whole_ln_fwd")
        ctx.save_for_backward(y, stats)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = hip.load()
        y, stats = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(y)
        ws = _scratch(y.device, 2048)
        hip.check(lib.ruart_whole_ln_bwd(hip.ptr(y), hip.ptr(gy), hip.ptr(stats), hip.ptr(gx), hip.ptr(ws), y.numel(),
                                         hip.stream_ptr()), "ruart_whole_ln_bwd")
        return gx, None


def whole_layer_norm(x, eps=1e-5):
    return _WholeLayerNorm.apply(x.contiguous(), eps)


# ---------------------------------------------------------------------------------------------------------
# Dense projections of the trunk.  "x3": fp32 operands split into bf16 hi + lo inside the kernel, three 16-bit MFMA products
# (ruart_gemm_x3, relative error ~2^-16 per product, ~5x the fp32-MFMA rate); "fp32": torch.mm / addmm (rocBLAS, exact fp32
# MFMA) - the validation mode (SDNet sets it when opt['bert_precision'] == 'fp32') and the fallback for tiny products.
trunk_gemm = "x3"
# GEMM form of the trunk's BACKWARD products (dX = dY W, dW = dY^T X, dW_hh): "x3" (three bf16 products, fp32-class) or "x1" (one
# bf16 product with fp32 accumulation - what mixed-precision training uses for gradients).  SDNet.forward sets it from
# opt['ruart_trunk_grad_gemm'] (default "x3").  Forward products always use trunk_gemm.
trunk_grad_gemm = "x3"
# variational-dropout masks of the projections' inputs read inside the GEMM's operand loads (see _Linear)
fuse_operand_masks = os.environ.get("RUART_FUSE_MASK", "1") != "0"


def _one_unit_stride(t):
    """A 2-D view whose memory is row- or column-contiguous (what ruart_gemm_x3 addresses); copies only when it is neither."""
    r, c = t.shape
    if t.stride(1) == 1 and (t.stride(0) >= c or r == 1):
        return t
    if t.stride(0) == 1 and (t.stride(1) >= r or c == 1):
        return t
    return t.contiguous()


_X3_SPAN_LIMIT = 1 << 30


def _span(t):
    """elements between the first and the last element of a 2-D view, inclusive"""
    return (t.shape[0] - 1) * abs(t.stride(0)) + (t.shape[1] - 1) * abs(t.stride(1)) + 1 if t.numel() else 0


def mm(a, b, bias=None, mode=None, out=None, a_keep=None, b_keep=None, keep_scale=1.0, c_scale=None, rpm=1, residual=None):
    """a (M,K) . b (K,N) (+ bias (N,)) (+ residual (M,N)) -> (M,N) fp32 on the split-bf16 MFMA kernel - every size, down to a
    single row: no product of a step goes to the vendor library (see the module docstring).
    ``mode``: 'x3' | 'x1' | 'fp32' (torch.mm: the validation mode); default = the module-level ``trunk_gemm`` (autograd Functions
    pass the mode of their forward).  ``out``: optional contiguous (M, N) destination.
    Variational-dropout masks fused into the product (each mask row is shared by ``rpm`` consecutive rows): ``a_keep`` (M/rpm, K) /
    ``b_keep`` (K/rpm, N) uint8 - the operand element is multiplied by ``keep_scale`` where the byte is non-zero and dropped where
    it is zero, inside the operand loads (one of the two per call, operand in its natural orientation); ``c_scale`` (M/rpm, N) fp32
    multiplies the output in the epilogue."""
    M, K = a.shape
    N = b.shape[1]
    mode = mode or trunk_gemm
    if mode in ("x3", "x1"):           # the product modes have no CPU path (torch.mm below is the fp32 VALIDATION mode only)
        for t in (a, b):
            if not (t.is_cuda and t.dtype == torch.float32):
                raise hip.HipError("ops.mm: fp32 device tensors only (there is no CPU path); got %s on %s" % (t.dtype, t.device))
    use_x3 = mode in ("x3", "x1")

    def apply_keep(t, keep):
        return t * (keep.repeat_interleave(rpm, 0) != 0).to(t.dtype) * keep_scale

    if use_x3:
        a, b = _one_unit_stride(a), _one_unit_stride(b)
        # the kernels address an operand as base + 32-bit element offset (< 2^30 elements from the first to the last one, include/ruart_hip.h):
        # a view with a huge pitch is compacted here; an operand that is itself that large is split over its rows
        if _span(a) >= _X3_SPAN_LIMIT and a.numel() < _X3_SPAN_LIMIT:
            a = a.contiguous()
        if _span(b) >= _X3_SPAN_LIMIT and b.numel() < _X3_SPAN_LIMIT:
            b = b.contiguous()
        if _span(a) >= _X3_SPAN_LIMIT and a_keep is None and c_scale is None and M > 1:
            # (rows of the product are independent: a b-side mask applies to both halves unchanged - it masks b's K x N elements)
            out = torch.empty(M, N, dtype=torch.float32, device=a.device) if out is None else out
            h = M // 2
            mm(a[:h], b, bias, mode, out[:h], residual=None if residual is None else residual[:h], b_keep=b_keep, keep_scale=keep_scale, rpm=rpm)
            mm(a[h:], b, bias, mode, out[h:], residual=None if residual is None else residual[h:], b_keep=b_keep, keep_scale=keep_scale, rpm=rpm)
            return out
        if max(_span(a), _span(b)) >= _X3_SPAN_LIMIT:
            raise ValueError("ops.mm: an operand spans >= 2^30 elements in a layout the split-bf16 kernel cannot address; split the product")
        # the fused operand masks follow the operand's natural orientation only (and one operand at a time): otherwise multiply first
        if a_keep is not None and (a.stride(1) != 1 or b_keep is not None or max(M, K) >= (1 << 20)):
            a, a_keep = _one_unit_stride(apply_keep(a, a_keep)), None
        if b_keep is not None and (b.stride(1) != 1 or max(N, K) >= (1 << 20)):
            b, b_keep = _one_unit_stride(apply_keep(b, b_keep)), None
    if not use_x3:
        if a_keep is not None:
            a = apply_keep(a, a_keep)
        if b_keep is not None:
            b = apply_keep(b, b_keep)
        r = torch.mm(a, b) if bias is None else torch.addmm(bias, a, b)
        if c_scale is not None:
            r = r * c_scale.repeat_interleave(rpm, 0)
        if residual is not None:
            r = r + residual
        return r if out is None else out.copy_(r)
    lib = hip.load()
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    elif not (out.is_contiguous() and out.shape == (M, N) and out.dtype == torch.float32):
        raise ValueError("mm(out=): a contiguous fp32 (M, N) tensor is required")
    sam, sak = (a.stride(0), 1) if a.stride(1) == 1 else (1, a.stride(1))
    sbk, sbn = (1, b.stride(1)) if b.stride(0) == 1 and b.stride(1) != 1 else (b.stride(0), 1)
    if b.stride(0) == 1 and b.stride(1) == 1:            # K == 1 or N == 1: either description is valid
        sbk, sbn = b.stride(0), 1
    for sc, shape, dt in ((a_keep, (M // rpm, K), torch.uint8), (b_keep, (K // rpm, N), torch.uint8), (c_scale, (M // rpm, N), torch.float32)):
        if sc is not None and not (sc.is_contiguous() and tuple(sc.shape) == shape and sc.dtype == dt):
            raise ValueError("mm: a fused mask must be a contiguous %s %s tensor" % (dt, shape))
    pkey = (M, N, K, sak == 1, sbk == 1)
    ws_bytes = _PLAN_BYTES.get(pkey)
    if ws_bytes is None:                                   # (the plan is a pure function of the shape and the two layouts: asked once)
        nbytes = ctypes.c_size_t(0)
        hip.check(lib.ruart_gemm_x3_plan(M, N, K, int(sak == 1), int(sbk == 1), None, ctypes.byref(nbytes)), "ruart_gemm_x3_plan")
        ws_bytes = _PLAN_BYTES[pkey] = int(nbytes.value)
    ws = _scratch(a.device, ws_bytes // 4, "x3") if ws_bytes else None
    fn = lib.ruart_gemm_x1 if mode == "x1" else lib.ruart_gemm_x3
    ldr = 0
    if residual is not None:
        if not (residual.shape == (M, N) and residual.dtype == torch.float32 and residual.stride(1) == 1):
            residual = residual.contiguous()
        ldr = residual.stride(0)
    if "x3" in _ABL_SKIP:
        return out.zero_()
    hip.check(fn(hip.ptr(a), sam, sak, hip.ptr(b), sbk, sbn, hip.ptr(bias), hip.ptr(residual), ldr, hip.ACT_NONE, hip.ptr(out), N,
                 M, N, K, hip.ptr(ws), ws_bytes, hip.ptr(a_keep), hip.ptr(b_keep), float(keep_scale), hip.ptr(c_scale), int(rpm),
                 hip.stream_ptr()), "ruart_gemm_x3")
    return out


def colsum(x):
    """x.sum(0) of a 2-D fp32 device matrix in two ordered stages (ruart_colsum_f32): the bias gradients of the trunk's projections -
    torch's generic reduction takes 33 us for a (6400, 1000) gradient, this 8."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and x.shape[0] > 64):
        return x.sum(0)
    rows, cols = x.shape
    lib = hip.load()
    out = torch.empty(cols, dtype=torch.float32, device=x.device)
    ws = _scratch(x.device, int(lib.ruart_colsum_f32_ws_floats(rows, cols)), "colsum")
    hip.check(lib.ruart_colsum_f32(hip.ptr(x), x.stride(0), rows, cols, hip.ptr(out), 0, hip.ptr(ws), hip.stream_ptr(x.device)), "ruart_colsum_f32")
    return out


# ---- deferred weight gradients ------------------------------------------------------------------------------------------------
# A training step has ~45 nn.Linear sites in the trunk.  Alone, each dW = dY^T X is a small output with a long reduction: 10-30 % of
# the chip for 20-50 us plus a split-K reduction launch.  With ``defer_weight_grads`` on (SDNet sets it for training passes), the
# backward of ``_Linear`` only RECORDS (dY, X, W) and queues one end-of-backward callback with the autograd engine; the callback runs
# all recorded products as ruart_gemm_x3_tn_grouped launches (one per alignment class and wave), writes / accumulates ``W.grad`` and
# fires the parameters' post-accumulate hooks (dp.GradSync) - all still inside ``loss.backward()``, so callers see nothing new.
# Every product is tiled, split and summed exactly as the single launch would: bitwise the same gradients.
defer_weight_grads = False
_deferred = []


def _flush_weight_grads():
    global _deferred
    items, _deferred = _deferred, []
    if not items:
        return
    lib = hip.load()
    dev = items[0][0].device
    with torch.cuda.device(dev):
        # the operands were produced on the trunk's three streams: this (the caller's) stream waits for each of them once, and the
        # caching allocator learns that their memory is read here
        cur = torch.cuda.current_stream(dev)
        for st in {it[3] for it in items}:
            if st != cur:
                cur.wait_stream(st)
        for gy, xm, _, st in items:
            if st != cur:
                gy.record_stream(cur)
                xm.record_stream(cur)
        # waves: a weight's k-th contribution goes to wave k (problems of one call must write distinct outputs)
        seen, waves = {}, []
        for gy, xm, w, _ in items:
            k = seen.get(id(w), 0)
            seen[id(w)] = k + 1
            while len(waves) <= k:
                waves.append([])
            waves[k].append((gy, xm, w))
        out = {}
        for k, wave in enumerate(waves):
            arr = (hip.X3TnProblemC * len(wave))()
            for i, (gy, xm, w) in enumerate(wave):
                if k == 0:
                    out[id(w)] = (w, torch.empty(w.shape, dtype=torch.float32, device=dev))
                c = out[id(w)][1]
                q = arr[i]
                q.A, q.B, q.C = gy.data_ptr(), xm.data_ptr(), c.data_ptr()
                q.lda, q.ldb, q.ldc = gy.stride(0), xm.stride(0), c.stride(0)
                q.M, q.N, q.K = gy.shape[1], xm.shape[1], gy.shape[0]
                q.accumulate = 1 if k > 0 else 0
            nbytes = int(lib.ruart_gemm_x3_tn_grouped_ws(arr, len(wave)))
            ws = _scratch(dev, nbytes // 4, "x3g") if nbytes else None
            hip.check(lib.ruart_gemm_x3_tn_grouped(arr, len(wave), hip.ptr(ws), nbytes, hip.stream_ptr(dev)), "ruart_gemm_x3_tn_grouped")
        for w, g in out.values():
            if w.grad is None:
                w.grad = g
            else:
                w.grad = w.grad + g
            for hook in (getattr(w, "_post_accumulate_grad_hooks", None) or {}).values():
                hook(w)


def _defer_weight_grad(gy, xm, w):
    """Record dW += gy^T xm for the end of this backward pass; False when the operands do not fit the grouped kernel."""
    if not (gy.dim() == 2 and xm.dim() == 2 and gy.stride(1) == 1 and xm.stride(1) == 1 and w.is_leaf and w.dim() == 2
            and w.shape == (gy.shape[1], xm.shape[1]) and gy.dtype == torch.float32 and xm.dtype == torch.float32):
        return False
    if torch.cuda.is_current_stream_capturing():
        return False
    if not _deferred:
        torch.autograd.Variable._execution_engine.queue_callback(_flush_weight_grads)
    _deferred.append((gy, xm, w, torch.cuda.current_stream(gy.device)))
    return True


class _Linear(torch.autograd.Function):
    """y = (x * mask) W^T (+ b) with x (rows, K), W (N, K), mask (rows / rpm, K) or None, on ruart_gemm_x3.
    With the mask's one-byte-per-element form at hand (``keep``, ``keep_scale``: layers.MaskBank) no pass of its own applies it: the
    forward reads the bytes beside x in the operand loads (a_keep), dW = dY^T (x * mask) reads them beside x again (b_keep), and
    dX = (dY W) * mask comes out of the GEMM epilogue (c_scale) - x * mask is never materialised.  Without it (a mask drawn outside the
    bank; ``fuse_operand_masks = False`` / RUART_FUSE_MASK=0: the form of rounds 2-3) the forward multiplies once and keeps the product."""

    @staticmethod
    def forward(ctx, x, w, b, mask, rpm, wparts=None, keep=None, keep_scale=1.0):
        """``wparts``: [(Parameter, row0, row1), ...] when ``w`` is a concatenation of parameters along its rows (the two directions of a
        BiLSTM input projection): deferred weight gradients are then taken per part, straight into the parameters."""
        ctx.mode = trunk_grad_gemm if trunk_gemm == "x3" else trunk_gemm       # form of the two backward products
        fused = (mask is not None and keep is not None and fuse_operand_masks and trunk_gemm == "x3" and not defer_weight_grads
                 and x.stride(1) == 1 and x.shape[0] < (1 << 20) and x.shape[1] < (1 << 20))
        ctx.fused, ctx.keep_scale = fused, float(keep_scale)
        xm = x if (mask is None or fused) else (x.view(-1, rpm, x.shape[1]) * mask.unsqueeze(1)).view(x.shape)
        ctx.save_for_backward(xm, w, mask, keep if fused else None)
        # the Parameter behind w, when w IS one (a grouped end-of-backward product writes its .grad directly, see above)
        ctx.wparam = w if (isinstance(w, torch.nn.Parameter) and w.requires_grad) else None
        ctx.wparts = wparts
        ctx.has_bias = b is not None
        ctx.rpm = rpm
        return mm(xm, w.t(), b, a_keep=keep if fused else None, keep_scale=keep_scale, rpm=rpm)

    @staticmethod
    def backward(ctx, gy):
        xm, w, mask, keep = ctx.saved_tensors
        gx = mm(gy, w, mode=ctx.mode, c_scale=mask, rpm=ctx.rpm) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            wp, parts = ctx.wparam, ctx.wparts
            done = False
            if defer_weight_grads and ctx.mode == "x3" and gy.stride(1) == 1 and not ctx.fused:
                if wp is not None:
                    done = _defer_weight_grad(gy, xm, wp)
                elif parts and all(p.requires_grad and p.is_leaf for p, _, _ in parts):
                    done = all([_defer_weight_grad(gy[:, r0:r1], xm, p) for p, r0, r1 in parts])
            if not done:
                gw = mm(gy.t(), xm, mode=ctx.mode, b_keep=keep, keep_scale=ctx.keep_scale, rpm=ctx.rpm)
        gb = colsum(gy) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb, None, None, None, None, None


class _AddMM(torch.autograd.Function):
    """base + x W^T in one launch (the residual rides in the GEMM epilogue): the recurrent product of the wide `multi2one` LSTM
    step added to the step's input projection (Models/SDNet.py:269-271 via nn.LSTM)."""

    @staticmethod
    def forward(ctx, base, x, w):
        ctx.save_for_backward(x, w)
        ctx.mode = trunk_grad_gemm if trunk_gemm == "x3" else trunk_gemm
        return mm(x, w.t(), residual=base)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = mm(gy, w, mode=ctx.mode) if ctx.needs_input_grad[1] else None
        gw = mm(gy.t(), x, mode=ctx.mode) if ctx.needs_input_grad[2] else None
        return (gy if ctx.needs_input_grad[0] else None), gx, gw


class _MatMul2D(torch.autograd.Function):
    """a (M, K) . b (K, N) for two fp32 device matrices, both differentiable, every product on ruart_gemm_x3."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        ctx.mode = trunk_grad_gemm if trunk_gemm == "x3" else trunk_gemm
        return mm(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        ga = mm(g, b.t(), mode=ctx.mode) if ctx.needs_input_grad[0] else None
        gb = mm(a.t(), g, mode=ctx.mode) if ctx.needs_input_grad[1] else None
        return ga, gb


def matmul2d(a, b):
    return _MatMul2D.apply(a, b)


def addmm(base, x, w):
    """base (M, N) + x (M, K) . w (N, K)^T."""
    if trunk_gemm != "x3":             # exact-fp32 validation mode
        return torch.addmm(base, x, w.t())
    if not x.is_cuda:
        raise hip.HipError("ops.addmm: device tensors only (there is no CPU path)")
    return _AddMM.apply(base, x.contiguous(), w)


def linear(x, w, b=None, mask=None, wparts=None):
    """F.linear for fp32 device tensors of any leading shape, through ``mm``.  ``mask`` (B, K) with x (B, T, K): variational
    dropout (x * mask[:, None, :]) applied inside the op.  A mask that carries its byte form (``mask.keep`` uint8 (B, K), ``mask.keep_scale``
    = 1/(1-p): what layers.MaskBank hands out) is read inside the three products; any other mask costs one multiply in the forward."""
    if trunk_gemm != "x3":             # exact-fp32 validation mode
        if mask is not None:
            x = x * mask.unsqueeze(1)
        return torch.nn.functional.linear(x, w, b)
    if not x.is_cuda:
        raise hip.HipError("ops.linear: device tensors only (there is no CPU path)")
    lead = x.shape[:-1]
    rpm = 1
    keep, keep_scale = None, 1.0
    if mask is not None:
        if x.dim() != 3 or mask.shape != (x.shape[0], x.shape[2]):
            raise ValueError("linear(mask=): x (B, T, K) with mask (B, K)")
        rpm = x.shape[1]
        keep, keep_scale = getattr(mask, "keep", None), getattr(mask, "keep_scale", 1.0)      # layers.MaskBank attaches them
        mask = mask.contiguous()
    x2 = x.reshape(-1, x.shape[-1])
    y = _Linear.apply(x2, w, b, mask, rpm, wparts, keep, keep_scale)
    return y.view(*lead, w.shape[0])


# ---------------------------------------------------------------------------------------------------------
def _padded16(x, dtype, mult=256):
    """fp32 (M, K) -> 16-bit (ceil(M / mult) * mult, K), zero rows at the end (the MFMA GEMM's row granularity)."""
    M, K = x.shape
    Mp = (M + mult - 1) // mult * mult
    y = torch.empty(Mp, K, dtype=dtype, device=x.device)
    y[:M].copy_(x)
    if Mp > M:
        y[M:].zero_()
    return y


def _gemm16(a16, w16, bias, M_out, dt):
    """a16 (Mp, K) . w16 (N, K)^T (+ bias) -> fp32 (M_out, N) on ruart_gemm_16_nt (the encoder's 256x256 MFMA kernel)."""
    lib = hip.load()
    Mp, K = a16.shape
    N = w16.shape[0]
    out = torch.empty(Mp, N, dtype=torch.float32, device=a16.device)
    hip.check(lib.ruart_gemm_16_nt(hip.ptr(a16), K, hip.ptr(w16), K, hip.ptr(bias), None, 0, dt, hip.ptr(out), N, hip.DT_F32, Mp, N, K,
                                   hip.ACT_NONE, dt, hip.stream_ptr()), "ruart_gemm_16_nt")
    return out[:M_out]


class _Linear16(torch.autograd.Function):
    """y = x W^T + b for fp32 x (M, K), W (N, K) with 16-bit MFMA operands (N, K multiples of 256 / 128): forward in f16 (11
    significand bits; activations are O(1)-O(100)), dX = dY W in bf16 (gradients need the exponent range), both on the encoder's
    GEMM kernel with on-the-fly casts; dW = dY^T X stays on the split-bf16 kernel (its reduction runs over all M rows)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return _gemm16(_padded16(x, torch.float16), w.to(torch.float16), b, x.shape[0], hip.DT_F16)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _gemm16(_padded16(gy, torch.bfloat16), w.t().contiguous().to(torch.bfloat16), None, gy.shape[0], hip.DT_BF16)
        if ctx.needs_input_grad[1]:
            gw = _dw_bf16(gy, x)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum(0)
        return gx, gw, gb


def _dw_bf16(gy, x):
    """dW (N, K) = gy^T . x for gy (rows, N), x (rows, K) fp32: one bf16 MFMA product with fp32 accumulation
    (ruart_gemm_bf16_tn), or the split-bf16 kernel when the operands are not laid out for it."""
    lib = hip.load()
    rows, N = gy.shape
    K = x.shape[1]
    if not (gy.is_contiguous() and x.is_contiguous() and N % 4 == 0 and K % 4 == 0 and gy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0):
        return mm(gy.t(), x, mode="x3")
    nbytes = ctypes.c_size_t(0)
    hip.check(lib.ruart_gemm_x3_plan(N, K, rows, 0, 0, None, ctypes.byref(nbytes)), "ruart_gemm_x3_plan")
    ws = _scratch(gy.device, nbytes.value // 4, "x3") if nbytes.value else None
    out = torch.empty(N, K, dtype=torch.float32, device=gy.device)
    rc = lib.ruart_gemm_bf16_tn(hip.ptr(gy), N, hip.ptr(x), K, hip.ptr(out), K, N, K, rows, hip.ptr(ws), nbytes.value, hip.stream_ptr())
    if rc != 0:
        return mm(gy.t(), x, mode="x3")
    return out


def linear16(x, w, b=None):
    """``linear`` with 16-bit MFMA operands for the two row-parallel products (see _Linear16); falls back to ``linear`` for shapes
    the encoder's GEMM does not take."""
    N, K = w.shape
    if not x.is_cuda or x.dim() != 2 or N % 256 or K % 256:
        return linear(x, w, b)
    return _Linear16.apply(x.contiguous(), w, b)


# ---------------------------------------------------------------------------------------------------------
class _LstmRecurrence(torch.autograd.Function):
    """Sequential part of one (Bi)LSTM layer.  xproj (B,T,ndir*4h) already holds x W_ih^T + b_ih + b_hh."""

    @staticmethod
    def forward(ctx, xproj, w_hh, ndir):
        lib = hip.load()
        hip.require_gpu(xproj, torch.float32)
        hip.require_gpu(w_hh, torch.float32)
        nan_flag.ensure(xproj.device)
        B, T, G = xproj.shape
        h = G // (4 * ndir)
        y = torch.empty(B, T, ndir * h, dtype=torch.float32, device=xproj.device)
        need = xproj.requires_grad or w_hh.requires_grad
        gates = torch.empty_like(xproj) if need else None
        cells = torch.empty_like(y) if need else None
        hprev = torch.empty_like(y) if w_hh.requires_grad else None
        if "lstm" in _ABL_SKIP:
            y.zero_()
            for t in (gates, cells, hprev):
                if t is not None:
                    t.zero_()
        else:
            hip.check(lib.ruart_lstm_fwd(hip.ptr(xproj), hip.ptr(w_hh), hip.ptr(y), hip.ptr(gates), hip.ptr(cells), hip.ptr(hprev), B, T, h,
                                         ndir, hip.stream_ptr()), "ruart_lstm_fwd")
        ctx.ndir, ctx.h = ndir, h
        ctx.mode = trunk_grad_gemm if trunk_gemm == "x3" else trunk_gemm
        ctx.save_for_backward(w_hh, gates, cells, hprev)
        ctx.shape = y.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = hip.load()
        w_hh, gates, cells, hprev = ctx.saved_tensors
        ndir, h = ctx.ndir, ctx.h
        B, T, _ = ctx.shape
        gy = gy.contiguous()
        gx = torch.empty_like(gates)
        if "lstm" in _ABL_SKIP:
            gx.zero_()
        else:
            hip.check(lib.ruart_lstm_bwd(hip.ptr(gy), hip.ptr(w_hh), hip.ptr(gates), hip.ptr(cells), hip.ptr(gx), B, T, h, ndir,
                                         hip.stream_ptr()), "ruart_lstm_bwd")
        # grad_W_hh[d] = sum_{b,t} da[b,t,d] (x) h_prev[b,t,d]: one GEMM per direction on column slices (strided views, no copies)
        gw = None
        if ctx.needs_input_grad[1]:
            gx2, hp2 = gx.view(B * T, -1), hprev.view(B * T, -1)
            if ndir == 1:
                gw = mm(gx2.t(), hp2, mode=ctx.mode).unsqueeze(0)
            else:
                # both directions in ONE product (ndir 4h x ndir h over the B T rows) whose diagonal blocks are the two gradients: the
                # off-diagonal half of the work is wasted, but each of these products is a 30 us launch-and-latency affair plus its
                # own split-K reduction - 22 of them per step before, 11 now
                full = mm(gx2.t(), hp2, mode=ctx.mode)
                gw = torch.stack([full[d * 4 * h:(d + 1) * 4 * h, d * h:(d + 1) * h] for d in range(ndir)], 0)
        return gx, gw, None


class _LstmPackParams(torch.autograd.Function):
    """(w_ih, w_ih_r, b_ih, b_hh, b_ih_r, b_hh_r, w_hh, w_hh_r) -> (w, b, whh) as lstm_layer's kernels take them
    (ruart_lstm_pack_params); the backward hands out slices of the incoming gradients - no launch."""

    @staticmethod
    def forward(ctx, w_ih, w_ih_r, b_ih, b_hh, b_ih_r, b_hh_r, w_hh, w_hh_r):
        G, K = w_ih.shape
        h = w_hh.shape[1]
        if not (w_ih_r.shape == (G, K) and w_hh.shape == w_hh_r.shape == (G, h)
                and b_ih.shape == b_hh.shape == b_ih_r.shape == b_hh_r.shape == (G,)):
            raise ValueError("lstm_layer: the two directions must have the same shapes")
        w = torch.empty(2 * G, K, dtype=torch.float32, device=w_ih.device)
        b = torch.empty(2 * G, dtype=torch.float32, device=w_ih.device)
        whh = torch.empty(2, G, h, dtype=torch.float32, device=w_ih.device)
        hip.check(hip.load().ruart_lstm_pack_params(hip.ptr(w_ih), hip.ptr(w_ih_r), hip.ptr(b_ih), hip.ptr(b_hh), hip.ptr(b_ih_r), hip.ptr(b_hh_r),
                                                    hip.ptr(w_hh), hip.ptr(w_hh_r), hip.ptr(w), hip.ptr(b), hip.ptr(whh), G, K, h,
                                                    hip.stream_ptr(w_ih.device)), "ruart_lstm_pack_params")
        ctx.G = G
        return w, b, whh

    @staticmethod
    def backward(ctx, gw, gb, gwhh):
        G = ctx.G
        gw0 = gw1 = gb0 = gb1 = gh0 = gh1 = None
        if gw is not None:
            gw0, gw1 = gw[:G], gw[G:]
        if gb is not None:
            gb0, gb1 = gb[:G], gb[G:]
        if gwhh is not None:
            gh0, gh1 = gwhh[0], gwhh[1]
        return gw0, gw1, gb0, gb0, gb1, gb1, gh0, gh1


def lstm_layer(x, w_ih, w_hh, b_ih, b_hh, w_ih_r=None, w_hh_r=None, b_ih_r=None, b_hh_r=None, mask=None):
    """One nn.LSTM layer (batch_first, zero state), uni- or bidirectional, on the persistent HIP recurrence.
    Parameters use torch's nn.LSTM layout so checkpoints load unchanged."""
    bidir = w_ih_r is not None
    wparts = None
    if bidir:
        params = (w_ih, w_ih_r, b_ih, b_hh, b_ih_r, b_hh_r, w_hh, w_hh_r)
        if trunk_gemm == "x3" and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in params):
            w, b, whh = _LstmPackParams.apply(*params)          # one launch instead of two cats, two adds and a stack
        else:
            w = torch.cat([w_ih, w_ih_r], 0)
            b = torch.cat([b_ih + b_hh, b_ih_r + b_hh_r], 0)
            whh = torch.stack([w_hh, w_hh_r], 0)
        if isinstance(w_ih, torch.nn.Parameter) and isinstance(w_ih_r, torch.nn.Parameter):
            wparts = [(w_ih, 0, w_ih.shape[0]), (w_ih_r, w_ih.shape[0], w.shape[0])]
    else:
        w, b, whh = w_ih, b_ih + b_hh, w_hh.unsqueeze(0)
    xproj = linear(x, w, b, mask=mask, wparts=wparts)   # mask: the input's variational-dropout mask, fused into the projection
    return _LstmRecurrence.apply(xproj.contiguous(), whh.contiguous(), 2 if bidir else 1)


# ---------------------------------------------------------------------------------------------------------
class _LstmCell(torch.autograd.Function):
    """Pointwise part of one LSTM step over a ragged, length-sorted batch (see ruart_lstm_cell_fwd)."""

    @staticmethod
    def forward(ctx, pre, h_prev, c_prev, n_active):
        lib = hip.load()
        for t in (pre, h_prev, c_prev):
            hip.require_gpu(t, torch.float32)
        N, h = h_prev.shape
        h_out, c_out = torch.empty_like(h_prev), torch.empty_like(c_prev)
        acts = torch.empty_like(pre)
        hip.check(lib.ruart_lstm_cell_fwd(hip.ptr(pre), hip.ptr(h_prev), hip.ptr(c_prev), hip.ptr(h_out), hip.ptr(c_out),
                                          hip.ptr(acts), n_active, N, h, hip.stream_ptr()), "ruart_lstm_cell_fwd")
        ctx.save_for_backward(acts, c_prev, c_out)
        ctx.n_active = n_active
        return h_out, c_out

    @staticmethod
    def backward(ctx, gh, gc):
        lib = hip.load()
        acts, c_prev, c_out = ctx.saved_tensors
        N, h = c_prev.shape
        gh = gh.contiguous() if gh is not None else None
        gc = gc.contiguous() if gc is not None else None
        g_pre = torch.empty_like(acts)
        g_h, g_c = torch.empty_like(c_prev), torch.empty_like(c_prev)
        hip.check(lib.ruart_lstm_cell_bwd(hip.ptr(gh), hip.ptr(gc), hip.ptr(acts), hip.ptr(c_prev), hip.ptr(c_out), hip.ptr(g_pre),
                                          hip.ptr(g_h), hip.ptr(g_c), ctx.n_active, N, h, hip.stream_ptr()), "ruart_lstm_cell_bwd")
        return g_pre, g_h, g_c, None


def lstm_cell(pre, h_prev, c_prev, n_active):
    """pre (n_active, 4h), h_prev / c_prev (N, h) -> (h, c) of all N rows; rows >= n_active pass through."""
    return _LstmCell.apply(pre.contiguous(), h_prev.contiguous(), c_prev.contiguous(), int(n_active))


# ---------------------------------------------------------------------------------------------------------
class _RowScale(torch.autograd.Function):
    """y[w] = x[w] * mask[row_of[w]] for a packed (words, D) fp32 matrix and one mask row per item (ruart_rows_scale): the mask rows are
    read through the index inside the kernel, in the forward and in the backward - no (words, D) copy of the mask exists."""

    @staticmethod
    def forward(ctx, x, mask, row_of):
        y = torch.empty_like(x)
        hip.check(hip.load().ruart_rows_scale(hip.ptr(x), x.stride(0), hip.ptr(mask), mask.stride(0), hip.ptr(row_of), hip.ptr(y), y.stride(0),
                                              x.shape[0], x.shape[1], hip.stream_ptr(x.device)), "ruart_rows_scale")
        ctx.save_for_backward(mask, row_of)
        return y

    @staticmethod
    def backward(ctx, g):
        mask, row_of = ctx.saved_tensors
        if g.stride(1) != 1 or g.stride(0) % 4 or g.data_ptr() % 16:
            g = g.contiguous()
        gx = torch.empty(g.shape, dtype=torch.float32, device=g.device)
        hip.check(hip.load().ruart_rows_scale(hip.ptr(g), g.stride(0), hip.ptr(mask), mask.stride(0), hip.ptr(row_of), hip.ptr(gx), gx.stride(0),
                                              g.shape[0], g.shape[1], hip.stream_ptr(g.device)), "ruart_rows_scale")
        return gx, None, None


def rows_scale(x, mask, row_of):
    """x (W, D) * mask (N, D)[row_of (W,)] on the fused kernel when the layout allows it (fp32 device tensors, D % 4 == 0, aligned rows),
    else the gather-and-multiply of torch."""
    if (x.is_cuda and x.dtype == torch.float32 and mask.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 0 and x.shape[1] % 4 == 0
            and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and mask.stride(1) == 1 and mask.stride(0) % 4 == 0
            and mask.data_ptr() % 16 == 0 and row_of.dtype == torch.int64 and row_of.is_contiguous()):
        return _RowScale.apply(x, mask, row_of)
    return x * mask[row_of]


# ---------------------------------------------------------------------------------------------------------
def embedding_grad(gy, sort, shape):
    """Gradient of an embedding table (``shape`` = (V, D)) from the (n, D) gradient rows of its lookups and the host-prepared sort of
    their ids (batch._sort_ids): 3 tensors = one workgroup per looked-up row; 4 tensors = the two-level form for rows with very many
    occurrences.  Ordered sums, no atomics: the same bits every run."""
    V, D = shape
    lib = hip.load()
    gw = torch.zeros(V, D, dtype=torch.float32, device=gy.device)
    gy = gy.reshape(-1, D).contiguous()
    if len(sort) == 3:
        order, seg_start, seg_row = sort
        hip.check(lib.ruart_embedding_bwd_sorted(hip.ptr(gy), hip.ptr(order), hip.ptr(seg_start), hip.ptr(seg_row), seg_row.numel(), D, hip.ptr(gw),
                                                 hip.stream_ptr(gy.device)), "ruart_embedding_bwd_sorted")
    else:
        order, sub_start, row_first, row_id = sort
        n_sub = sub_start.numel() - 1
        ws = torch.empty(max(n_sub, 1), D, dtype=torch.float32, device=gy.device)
        hip.check(lib.ruart_embedding_bwd_split(hip.ptr(gy), hip.ptr(order), hip.ptr(sub_start), n_sub, hip.ptr(row_first), hip.ptr(row_id),
                                                row_id.numel(), D, hip.ptr(ws), hip.ptr(gw), hip.stream_ptr(gy.device)), "ruart_embedding_bwd_split")
    return gw


class _Embedding(torch.autograd.Function):
    """weight[ids] whose backward uses a host-prepared sort of the ids (``embedding_grad``) instead of sorting on the device: ``sort`` =
    int32 device tensors from batch.BatchIndex (padding_idx already left out)."""

    @staticmethod
    def forward(ctx, weight, ids, *sort):
        ctx.save_for_backward(*sort)
        ctx.wshape = weight.shape
        return weight.index_select(0, ids.reshape(-1)).view(*ids.shape, weight.shape[1])

    @staticmethod
    def backward(ctx, gy):
        return (embedding_grad(gy, ctx.saved_tensors, ctx.wshape), None) + (None,) * len(ctx.saved_tensors)


def embedding(module, ids, sort=None):
    """``module(ids)`` for an nn.Embedding; with ``sort`` (and a trainable fp32 table on the device) the gradient takes the
    host-sorted path."""
    w = module.weight
    if sort is None or not w.requires_grad or not torch.is_grad_enabled() or not w.is_cuda or w.dtype != torch.float32:
        return module(ids)
    return _Embedding.apply(w, ids, *sort)

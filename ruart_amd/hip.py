"""ctypes binding of libruart_hip.so (the C ABI declared in include/ruart_hip.h).

The library is plain HIP (no torch types in any signature): tensors cross the boundary as
``tensor.data_ptr()`` plus explicit sizes, and launches go to ``torch.cuda.current_stream()``.
There is NO fallback: if the shared object cannot be loaded or built, importing a product module
that needs it raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_longlong, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libruart_hip.so")

DT_F32, DT_BF16, DT_F16 = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
TORCH_DTYPE = {DT_F32: torch.float32, DT_BF16: torch.bfloat16, DT_F16: torch.float16}
# storage type of the encoder's layer outputs per precision mode.  x3: fp32 storage, split-bf16 MFMA products; fp16c: fp32 residual
# stream / layer outputs, GEMM operands as f16 + two e4m3 bytes (f16 MFMA + block-scaled fp8 correction, csrc/gemm_corr.hip)
PRECISION = {"fp32": DT_F32, "bf16": DT_BF16, "fp16": DT_F16, "x3": DT_F32, "fp16c": DT_F32}


class BertModelC(Structure):
    _fields_ = [("hidden", c_int), ("n_heads", c_int), ("n_layers", c_int), ("intermediate", c_int), ("dtype", c_int),
                ("ln_eps", c_float),
                ("word_emb", c_void_p), ("pos_emb", c_void_p), ("type_emb", c_void_p), ("emb_ln_g", c_void_p), ("emb_ln_b", c_void_p),
                ("w_qkv", POINTER(c_void_p)), ("b_qkv", POINTER(c_void_p)),
                ("w_ao", POINTER(c_void_p)), ("b_ao", POINTER(c_void_p)),
                ("ln1_g", POINTER(c_void_p)), ("ln1_b", POINTER(c_void_p)),
                ("w_ff1", POINTER(c_void_p)), ("b_ff1", POINTER(c_void_p)),
                ("w_ff2", POINTER(c_void_p)), ("b_ff2", POINTER(c_void_p)),
                ("ln2_g", POINTER(c_void_p)), ("ln2_b", POINTER(c_void_p)), ("f32_gemm", c_int), ("corr8", c_int),
                ("w8_qkv", POINTER(c_void_p)), ("w8_ao", POINTER(c_void_p)), ("w8_ff1", POINTER(c_void_p)), ("w8_ff2", POINTER(c_void_p)),
                ("tail_cus", c_int), ("ln_fold", c_int), ("fold_c_qkv", POINTER(c_void_p)), ("fold_c_ff1", POINTER(c_void_p)),
                ("fold_s_qkv", POINTER(c_float)), ("fold_s_ff1", POINTER(c_float))]


class X3TnProblemC(Structure):
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("lda", c_int), ("ldb", c_int), ("ldc", c_int), ("M", c_int), ("N", c_int),
                ("K", c_int), ("accumulate", c_int)]


class WPrepItemC(Structure):
    _fields_ = [("w", c_void_p), ("out16", c_void_p), ("outT_bf16", c_void_p), ("ldw", c_int), ("ld16", c_int), ("ldT", c_int), ("rows", c_int),
                ("cols", c_int), ("scale", c_float)]


class BertBatchC(Structure):
    _fields_ = [("n_tokens", c_int), ("n_rows", c_int), ("ids", c_void_p), ("pos_ids", c_void_p), ("n_blocks", c_int),
                ("blk_q0", c_void_p), ("blk_q1", c_void_p), ("blk_k0", c_void_p), ("blk_k1", c_void_p),
                ("tok_lo", c_void_p), ("tok_hi", c_void_p), ("key_bias", c_void_p),
                ("n_long_blocks", c_int), ("lblk_q0", c_void_p), ("lblk_q1", c_void_p), ("lblk_k0", c_void_p), ("lblk_k1", c_void_p),
                ("n_last_rows", c_int), ("last_rows", c_void_p)]


_P, _I, _F, _LL = c_void_p, c_int, c_float, c_longlong
_SIGNATURES = {
    "ruart_version": (c_char_p, []),
    "ruart_gemm_16_nt": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ruart_gemm_16_tail_ws_bytes": (c_size_t, [_I, _I, _I, _I]),
    "ruart_gemm_16_nt_ws": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, c_size_t, _I, _P]),
    "ruart_gemm_16c_nt": (_I, [_P, _P, _I, _P, _P, _I, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P]),
    "ruart_gemm_16c_nt_sel": (_I, [_P, _P, _I, _P, _P, _I, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "ruart_gemm_16c_tail_ws_bytes": (c_size_t, [_I, _I, _I, _I]),
    "ruart_gemm_16c_nt_ws": (_I, [_P, _P, _I, _P, _P, _I, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _P, c_size_t, _I, _P]),
    "ruart_f16c_shifts": (_I, [POINTER(c_int)]),
    "ruart_gemm_16c_set_dual": (_I, [_I]),
    "ruart_bert_set_correction": (_I, [_I, _I, _I, _I, ctypes.c_ulonglong]),
    "ruart_rows_layernorm_split": (_I, [_P, _I, _P, _P, _F, _P, _P, _P, _I, _I, _I, _P]),
    "ruart_bert_embed_ln_split": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _I, _I, _I, _P]),
    "ruart_bert_attention_split": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ruart_bert_attention_split_set_heads": (_I, [_I]),
    "ruart_gemm_16_nt_splitk": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ruart_gemm_16_nt_gelu2": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ruart_gemm_16_nt_gelu_bwd_ws_floats": (c_size_t, [_I, _I]),
    "ruart_gemm_16_nt_gelu_bwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _P]),
    "ruart_colsum_f32_ws_floats": (c_size_t, [_I, _I]),
    "ruart_colsum_f32": (_I, [_P, _I, _I, _I, _P, _I, _P, _P]),
    "ruart_colsum_f32_rows": (_I, [_P, _I, _I, _I, _P, _I, _P]),
    "ruart_gemm_16_tn_splitk": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ruart_ln_train_fwd": (_I, [_P, _I, _P, _I, _P, _P, _F, _F, ctypes.c_uint, _I, _P, _P, _P, _I, _I, _I, _P]),
    "ruart_ln_train_bwd_ws_floats": (c_size_t, [_I]),
    "ruart_ln_train_bwd": (_I, [_P, _I, _P, _P, _P, _I, _P, _P, _F, ctypes.c_uint, _I, _P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _I, _P]),
    "ruart_gelu_bwd_rows": (_I, [_P, _P, _I, _P, _P, _I, _I, _P]),
    "ruart_f16_to_bf16": (_I, [_P, _P, _LL, _P]),
    "ruart_weight_prep": (_I, [_P, _I, _F, _P, _I, _P, _I, _I, _I, _P]),
    "ruart_weight_prep_batch": (_I, [POINTER(WPrepItemC), _I, _P]),
    "ruart_colsum_bf16": (_I, [_P, _I, _I, _I, _P, _I, _P, _P]),
    "ruart_transpose16": (_I, [_P, _I, _P, _I, _I, _I, _I, _P]),
    "ruart_splitk_reduce": (_I, [_P, _LL, _I, _P, _LL, _F, _I, _P]),
    "ruart_mix_rows": (_I, [_P, _LL, _I, _I, _P, _P, _I, _I, _I, _P]),
    "ruart_mix_rows_bwd": (_I, [_P, _LL, _I, _I, _P, _I, _P, _P, _I, _I, _P]),
    "ruart_attn_train_fwd": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _F, ctypes.c_uint, _P]),
    "ruart_attn_train_bwd": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _F, ctypes.c_uint, _P, _P]),
    "ruart_attn_train_fwd_long": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _F, ctypes.c_uint, _P, _P]),
    "ruart_attn_train_bwd_long": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _F, ctypes.c_uint, _P, _P, _P, _P, _P]),
    "ruart_gemm_f32_nt": (_I, [_P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "ruart_bert_embed_ln": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _P, _I, _I, _I, _I, _P]),
    "ruart_rows_layernorm": (_I, [_P, _I, _P, _P, _F, _P, _I, _I, _I, _I, _P]),
    "ruart_bert_attention": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "ruart_bert_pool_set_variant": (_I, [_I]),
    "ruart_bert_pool_mix": (_I, [_P, _LL, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ruart_bert_pool_mix_bwd": (_I, [_P, _LL, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _P]),
    "ruart_rows_gather": (_I, [_P, _I, _P, _LL, _P, _LL, _I, _P, _LL, _P, _LL, _I, _P, _LL, _P, _LL, _I, _P]),
    "ruart_cast_f32_to_16": (_I, [_P, _P, _I, _LL, _F, _P]),
    "ruart_bert_workspace_bytes": (c_size_t, [POINTER(BertModelC), _I]),
    "ruart_bert_forward": (_I, [POINTER(BertModelC), POINTER(BertBatchC), _P, _P, c_size_t, _P]),
    "ruart_bert_workspace_bytes_folded": (c_size_t, [POINTER(BertModelC), _I]),
    "ruart_bert_forward_folded": (_I, [POINTER(BertModelC), POINTER(BertBatchC), _P, _P, _P, c_size_t, _P]),
    "ruart_gemm_16c_nt_fold": (_I, [_P, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P, _F, _P, _I, _P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "ruart_gemm_16_nt_fold": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _F, _P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _F, _I, _P]),
    "ruart_rows_stats_finish": (_I, [_P, _I, _I, _F, _F, _P, _P]),
    "ruart_bert_pool_mix_ln": (_I, [_P, _LL, _I, _I, _P, _LL, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ruart_bert_pool_ln_set_variant": (_I, [_I]),
    "ruart_bert_pool_mix_ln_bwd": (_I, [_P, _LL, _I, _I, _P, _LL, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _P]),
    "ruart_attn_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ruart_attn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ruart_attn_set_prefetch": (_I, [_I]),
    "ruart_attn_fwd_pscale": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ruart_attn_bwd_pscale": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ruart_whole_ln_fwd": (_I, [_P, _P, _P, _P, _LL, _F, _P]),
    "ruart_whole_ln_bwd": (_I, [_P, _P, _P, _P, _P, _LL, _P]),
    "ruart_scorer_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ruart_scorer_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ruart_lstm_set_variant": (_I, [_I]),
    "ruart_lstm_pack_params": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ruart_lstm_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ruart_lstm_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ruart_rows_scale": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _P]),
    "ruart_lstm_cell_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ruart_lstm_cell_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ruart_set_nan_flag": (_I, [_P]),
    "ruart_grad_norm_clip": (_I, [_P, _P, _P, _P, _I, _F, _P, _P, _P, _P]),
    "ruart_adamax_step": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _F, _F, _F, _P]),
    "ruart_embedding_bwd_sorted": (_I, [_P, _P, _P, _P, _I, _I, _P, _P]),
    "ruart_embedding_bwd_split": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _P, _P, _P]),
    "ruart_phoc_table": (_I, [_P, _P, _I, _P, _I, _P, _P]),
    "ruart_gemm_bf16_tn": (_I, [_P, ctypes.c_longlong, _P, ctypes.c_longlong, _P, _I, _I, _I, _I, _P, ctypes.c_size_t, _P]),
    "ruart_gemm_x3_tn_grouped_ws": (c_size_t, [POINTER(X3TnProblemC), _I]),
    "ruart_gemm_x3_tn_grouped": (_I, [POINTER(X3TnProblemC), _I, _P, c_size_t, _P]),
    "ruart_gemm_x3_plan": (_I, [_I, _I, _I, _I, _I, POINTER(ctypes.c_int), POINTER(ctypes.c_size_t)]),
    "ruart_gemm_x3": (_I, [_P, ctypes.c_longlong, ctypes.c_longlong, _P, ctypes.c_longlong, ctypes.c_longlong, _P, _P, _I, _I, _P, _I, _I,
                           _I, _I, _P, ctypes.c_size_t, _P, _P, ctypes.c_float, _P, _I, _P]),
    "ruart_gemm_x1": (_I, [_P, ctypes.c_longlong, ctypes.c_longlong, _P, ctypes.c_longlong, ctypes.c_longlong, _P, _P, _I, _I, _P, _I, _I,
                           _I, _I, _P, ctypes.c_size_t, _P, _P, ctypes.c_float, _P, _I, _P]),
    "ruart_stream_create_cu_masked": (_I, [_I, POINTER(ctypes.c_void_p)]),
    "ruart_stream_create_priority": (_I, [_I, POINTER(ctypes.c_void_p)]),
    "ruart_stream_destroy": (_I, [_P]),
    "ruart_gemm_set_tile_order": (_I, [_I]),
    "ruart_gemm_set_variant": (_I, [_I]),
    "ruart_prof_enable": (_I, [_I]),
    "ruart_prof_read": (_I, [POINTER(ctypes.c_double), POINTER(c_longlong), POINTER(ctypes.c_double)]),
    "ruart_prof_mark": (_I, [_I, _P]),
    "ruart_prof_timeline": (_I, [_P, _P, _P, _I, _P]),
}

_lib = None


def load(build_if_missing=True):
    """Load (building first if the .so is absent) and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if not build_if_missing:
            raise OSError("libruart_hip.so not built: run `python -m ruart_amd.build`")
        from . import build as _build
        _build.build(verbose=False)
    # RUART_HIP_LIB: load an experimental build of the same ABI instead (kernel A/B runs, tools/build_variant.sh)
    lib = ctypes.CDLL(os.environ.get("RUART_HIP_LIB") or LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if os.environ.get("RUART_GEMM_VARIANT"):             # experiments only: tile variant of the encoder GEMM (default 5)
        lib.ruart_gemm_set_variant(int(os.environ["RUART_GEMM_VARIANT"]))
    if os.environ.get("RUART_CORR_DUAL"):                # experiments only: the 256 x 128 two-workgroups-per-CU form of the fp16c QKV / intermediate
        lib.ruart_gemm_16c_set_dual(int(os.environ["RUART_CORR_DUAL"]))        # products (measured 5-8 % slower than the 256 x 256 form)
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGNATURES)


class HipError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise HipError("%s failed with hipError_t %d" % (what, rc))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_handle(device=None):
    """Raw handle (int) of the current stream of ``device`` (a tensor's device; None: the current device).  Every kernel launch of a step
    asks for it - ~200 times per step -, so it goes through torch's C entry point where there is one (0.3 us) instead of building a
    torch.cuda.Stream object through three layers of device-index helpers (8 us: 1 ms of a step's enqueue time, which is partly on
    the pipelined step's critical path - profiles/r04_host_delay.log)."""
    if _raw_stream is not None:
        if device is None:
            idx = _cur_device()
        elif isinstance(device, int):
            idx = device
        else:
            idx = device.index if getattr(device, "index", None) is not None else _cur_device()
        return _raw_stream(idx)
    return torch.cuda.current_stream(device).cuda_stream


def stream_ptr(device=None):
    """The current stream of ``device`` (a tensor's device), not of whichever device happens to be current: a rank whose GPU is
    not the current device must not launch on device 0's stream with pointers into its own memory."""
    return c_void_p(stream_handle(device))


_DTYPE_CODE = {torch.float32: DT_F32, torch.bfloat16: DT_BF16, torch.float16: DT_F16}


def dtype_code(t):
    """RUART_DT_* of a tensor's storage type."""
    return _DTYPE_CODE[t.dtype]


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def f16c_shifts():
    """(SA_LO, SA_HI, SW_HI, SW_LO): the exponents of the fp16c mode's e4m3 companions as compiled into the library (csrc/common.h)."""
    out = (c_int * 4)()
    check(load().ruart_f16c_shifts(out), "ruart_f16c_shifts")
    return tuple(int(v) for v in out)


def require_gpu(t, dtype=None):
    if not t.is_cuda:
        raise HipError("ruart_amd kernels need device tensors; got a CPU tensor (there is no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise HipError("expected %s, got %s" % (dtype, t.dtype))
    if not t.is_contiguous():
        raise HipError("tensor must be contiguous")
    return t


def cu_masked_stream(n_cus, device):
    """torch view of a HIP stream limited to ``n_cus`` compute units (ruart_stream_create_cu_masked)."""
    import torch
    lib = load()
    out = ctypes.c_void_p()
    with torch.cuda.device(device):
        check(lib.ruart_stream_create_cu_masked(int(n_cus), ctypes.byref(out)), "ruart_stream_create_cu_masked")
    # No atexit destroy: at interpreter exit the HIP runtime, RCCL and - under rocprofv3 - the profiler's tool library are torn
    # down in an order this module does not control; destroying the stream from an atexit hook after the profiler had finalised
    # is what ended a round-1 profiling run with SIGSEGV in __cxa_finalize (gpurun_out/pftrace.log: the masked stream was the
    # default prefetch stream then).  The process exit releases the stream; call ruart_stream_destroy yourself to drop one earlier.
    st = torch.cuda.ExternalStream(out.value, device=device)
    st._ruart_handle = out.value            # for destroy_stream()
    st._ruart_masked = True
    return st


def priority_stream(priority, device):
    """torch view of a non-blocking HIP stream of HIP priority ``priority`` (-1 high, 0 normal, 1 LOW - a level torch's stream pool
    does not offer; ruart_stream_create_priority)."""
    import torch
    lib = load()
    out = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = lib.ruart_stream_create_priority(int(priority), ctypes.byref(out))
    if rc < 0:
        raise HipError("ruart_stream_create_priority failed (%d)" % rc)
    st = torch.cuda.ExternalStream(out.value, device=device)
    st._ruart_handle = out.value
    st._ruart_level = rc
    return st


def destroy_stream(st):
    """Destroy a stream made by ``cu_masked_stream`` (after synchronising it).  Call it before the process ends when a profiler is
    attached: on ROCm 7.2 a CU-masked queue that is still alive when the runtime's static destructors run takes rocprofv3's
    teardown down with SIGSEGV in __cxa_finalize (after the tool has written its output) - seen in round 1 and again in round 2
    with no atexit hook of ours involved."""
    handle = getattr(st, "_ruart_handle", None)
    if handle is None:
        return
    st.synchronize()
    load().ruart_stream_destroy(handle)
    st._ruart_handle = None

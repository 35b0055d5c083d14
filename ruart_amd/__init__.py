"""ruart_amd: MI355X-native hot path of RUArt (BERT encoder -> SDNet trunk -> answer scores)."""

# Encoder precision modes (opt['bert_precision']) whose answer probabilities stay within 1e-3 of the reference's fp32 CPU path on
# EVERY output at BASELINE.json's full size (tests/test_gpu_parity_full.py holds them to that bound against the reference's own
# outputs): the only modes bench.py may quote as its headline.  'fp16' / 'bf16' are throughput modes.
PASSING_PRECISIONS = ("fp32", "x3", "fp16c")

# The mode a user gets when opt['bert_precision'] is absent: the fastest one that holds the bound.  Its GEMM needs hidden and
# intermediate sizes that are multiples of 256 (bert-base / bert-large are); a toy encoder that is not gets the next passing mode.
DEFAULT_PRECISION = "fp16c"


def precision_of(opt, hidden=None):
    """opt['bert_precision'], or the default for an encoder of width ``hidden`` (taken from opt['bert_config'] / BERT_LARGE when None)."""
    p = opt.get("bert_precision")
    if p:
        return p
    if hidden is None:
        hidden = (opt.get("bert_config") or {}).get("hidden_size", 1024 if "BERT_LARGE" in opt else 768)
    return DEFAULT_PRECISION if hidden % 256 == 0 else "x3"

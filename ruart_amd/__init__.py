"""ruart_amd: MI355X-native hot path of RUArt (BERT encoder -> SDNet trunk -> answer scores)."""

# Encoder precision modes (opt['bert_precision']) whose answer probabilities stay within 1e-3 of the reference's fp32 CPU path on
# EVERY output at BASELINE.json's full size (tests/test_gpu_parity_full.py holds them to that bound against the reference's own
# outputs): the only modes bench.py may quote as its headline.  'fp16' / 'bf16' are throughput modes.
PASSING_PRECISIONS = ("fp32", "x3", "fp16c")

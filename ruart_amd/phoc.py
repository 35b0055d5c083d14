"""PHOC word descriptors (Utils/phoc.py:8-12 + Utils/cphoc.c): ``build_phoc(token) -> list[604]`` and the batched
``phoc_table(words) -> FloatTensor(len(words), 604)`` that fills the ``phoc_embedding`` table of a vocabulary in one launch of
``ruart_phoc_table`` (the reference's preprocessing calls ``build_phoc`` once per vocabulary word, Utils/CoQAUtils.py:75-87).
Both run on the GPU; there is no host fallback."""
import numpy as np
import torch

from . import hip

PHOC_DIM = 604
_ALPHABET = frozenset("abcdefghijklmnopqrstuvwxyz0123456789")


def normalize(token):
    """Lower-case, strip, keep [a-z0-9] only (Utils/phoc.py:8-10)."""
    return "".join(c for c in token.lower().strip() if c in _ALPHABET)


def phoc_table(words, device="cuda", normalized=False):
    """Rows of 0/1 fp32, one per word, on ``device``."""
    lib = hip.load()
    device = torch.device(device)
    if device.type != "cuda":
        raise hip.HipError("phoc_table runs on the GPU (got device %s)" % device)
    words = [w if normalized else normalize(w) for w in words]
    out = torch.empty(len(words), PHOC_DIM, dtype=torch.float32, device=device)
    if not words:
        return out
    enc = [w.encode("ascii") for w in words]
    offsets = np.zeros(len(enc) + 1, dtype=np.int32)
    np.cumsum([len(e) for e in enc], out=offsets[1:])
    chars = np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8)
    with torch.cuda.device(device):
        d_chars = torch.from_numpy(chars.copy()).to(device)
        d_off = torch.from_numpy(offsets).to(device)
        status = torch.zeros(1, dtype=torch.int32, device=device)
        rc = lib.ruart_phoc_table(hip.ptr(d_chars), hip.ptr(d_off), len(words), hip.ptr(out), PHOC_DIM, hip.ptr(status), hip.stream_ptr())
        hip.check(rc, "ruart_phoc_table")
        bad = int(status.item())
    if bad:
        raise RuntimeError("Error: unigram outside [a-z0-9] in word %r" % words[bad - 1])
    return out


def build_phoc(token):
    """The reference's per-token call: a list of 604 floats."""
    return phoc_table([token])[0].tolist()

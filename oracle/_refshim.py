"""Harness-side shims that let the *unmodified* reference be imported on CPU.

TEST INFRASTRUCTURE ONLY.  Used by oracle/gen_golden.py and oracle/time_reference.py
in the build container, where /root/reference is mounted.  Nothing of the reference
is copied: it is imported from where it lies.  The GPU box has no /root/reference
and never runs this file.

What is shimmed (all ordinary missing-module / no-device conditions):
  * ``spacy`` / ``fasttext`` are not installed: stub modules; only ``len(POS)`` and
    ``len(ENT)`` reach the model (Models/SDNet.py:123,128), so the stub exposes
    50 tagger labels and 74 entity moves  => table sizes 51 / 75.
  * ``h5py`` (Models/SDNetTrainer.py:19, image-feature loading only) is not installed: empty stub module.
  * ``.cuda()`` is hard-coded in forward (Models/SDNet.py:281-299 ...): patched to
    identity so the reference runs as a PyTorch-CPU program.
"""
import os
import sys
import types
import json
import tempfile

import torch

REF = "/root/reference"
N_POS_LABELS = 50
N_ENT_MOVES = 74


def install():
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not mounted at " + REF)
    if "spacy" not in sys.modules:
        spacy = types.ModuleType("spacy")

        class _NLP:
            class tagger:
                labels = ["T%d" % i for i in range(N_POS_LABELS)]

            class entity:
                move_names = ["M%d" % i for i in range(N_ENT_MOVES)]

        spacy.load = lambda *a, **k: _NLP()
        sys.modules["spacy"] = spacy
    if "fasttext" not in sys.modules:
        ft = types.ModuleType("fasttext")
        ft.load_model = lambda *a, **k: None
        sys.modules["fasttext"] = ft
    if "h5py" not in sys.modules:                 # imported at the top of Models/SDNetTrainer.py:19, used only for image features
        sys.modules["h5py"] = types.ModuleType("h5py")
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    if REF not in sys.path:
        sys.path.insert(0, REF)


def write_bert_dir(cfg, weights):
    """Materialise a HF-0.x style model directory (bert_config.json + pytorch_model.bin
    with the 'bert.' prefix, Models/Bert/modeling.py:497-521) from synthetic weights."""
    d = tempfile.mkdtemp(prefix="ruart_bert_")
    with open(os.path.join(d, "bert_config.json"), "w") as f:
        json.dump(cfg, f)
    sd = {k: torch.from_numpy(v.copy()) for k, v in weights.items()}
    torch.save(sd, os.path.join(d, "pytorch_model.bin"))
    return d

"""Config-file reader with the reference's semantics.

Mirrors ``Utils/Arguments.py:41-66`` of the reference: one ``KEY [VALUE]`` per
line, tabs are blanks, a line starting with ``#`` is a comment, a bare key
means ``True``, the first occurrence of a key wins, values are coerced
int -> float -> bool -> str.  Lines with three or more fields are ignored,
exactly like the reference.  Feature gating everywhere else is by *key
presence* (``'LN' in opt``), so the result is a plain ``dict``.
"""
import os


def _coerce(text):
    for cast in (int, float):
        try:
            return cast(text)
        except ValueError:
            pass
    if text.lower() in ("true", "false"):
        return text.lower() == "true"
    return text


class Arguments:
    def __init__(self, confFile):
        if not os.path.exists(confFile):
            raise Exception("The argument file does not exist: " + confFile)
        self.confFile = confFile

    def readArguments(self):
        opt = {}
        with open(self.confFile, encoding="utf-8") as f:
            for raw in f:
                line = raw.replace("\t", " ").strip()
                if line.startswith("#"):
                    continue
                fields = line.split()
                if len(fields) not in (1, 2) or fields[0] in opt:
                    continue
                opt[fields[0]] = True if len(fields) == 1 else _coerce(fields[1])
        return opt


def default_opt(**overrides):
    """The shipped ST-VQA base configuration as an ``opt`` dict, plus the keys the
    reference injects at run time (``main.py:28-30``, ``CoQAPreprocess.py:486-497``)."""
    here = os.path.dirname(os.path.abspath(__file__))
    opt = Arguments(os.path.join(here, "..", "configs", "stvqa_base.conf")).readArguments()
    opt.update({"cuda": False, "datadir": ".", "vocab_size": 2000,
                # len(POS) / len(ENT): the reference takes them from spaCy at import
                # (Utils/CoQAUtils.py:31-32); here they are explicit config.
                "pos_vocab_size": 51, "ent_vocab_size": 75})
    opt.update(overrides)
    return opt

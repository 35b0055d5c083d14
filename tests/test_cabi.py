"""The C-ABI library builds for gfx950, loads, and exports every symbol include/ruart_hip.h declares.
No compute calls here (no GPU in the build container)."""
import os
import re

from ruart_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "ruart_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ruart_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_functions() == hip.exported_symbols()


def test_library_exports_every_declared_symbol():
    lib = hip.load()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.ruart_version().decode().startswith("ruart_hip")


def test_product_modules_do_not_import_the_oracle():
    """oracle/ is test infrastructure: no product module may import or execute it."""
    pkg = os.path.join(ROOT, "ruart_amd")
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|import_module\(.oracle|__import__\(.oracle", re.M)
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            assert not pat.search(open(os.path.join(pkg, fn)).read()), fn

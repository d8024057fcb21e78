"""CPU: the C-ABI library loads here and exports every symbol include/idxtts.h declares
(no compute calls: there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from indextts_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            text = open(os.path.join(ROOT, "include", fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names |= set(re.findall(r"\b(idxtts_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_is_built():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"


def test_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared()
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert declared == set(_lib.SYMBOLS), f"binding list out of date: {declared ^ set(_lib.SYMBOLS)}"


def test_version_and_error_string():
    lib = _lib.load()
    assert lib.idxtts_version() >= 100
    assert lib.idxtts_last_error() is not None


def test_product_path_has_no_cpu_fallback():
    import torch
    from indextts_amd import vocoder
    x = torch.zeros(1, 2, 8)
    with pytest.raises(RuntimeError):
        vocoder.anti_alias_activation_forward(x, torch.zeros(12), torch.zeros(12), torch.zeros(2), torch.zeros(2))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "index-tts_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{fn} imports the oracle"

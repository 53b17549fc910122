"""CPU-side checks of the boundary: the library loads, exports every symbol
include/fqdedup_hip.h declares, and refuses to work without a device."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from fastqdedup_amd import _lib
    L = _lib.load()
    with open(os.path.join(ROOT, "include", "fqdedup_hip.h")) as fh:
        header = fh.read()
    declared = sorted(set(re.findall(r"\b(fqd_[a-z_0-9]+)\s*\(", header)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/fqdedup_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == declared


def test_no_cpu_fallback_without_device():
    import fastqdedup_amd as F
    from fastqdedup_amd import _lib
    if _lib.load().fqd_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        F.Context(0)
    with pytest.raises(RuntimeError):
        F.Trie()
    with pytest.raises(RuntimeError):
        F.within_distance("AAAA", "AAAC", 1)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under fastqdedup_amd/ may mention it."""
    pkg = os.path.join(ROOT, "fastqdedup_amd")
    for base, _, files in os.walk(pkg):
        if os.sep + "build" in base:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle" not in text.lower(), (base, f)

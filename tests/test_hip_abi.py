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


def test_hot_kernels_do_not_spill_to_scratch():
    """The build records what the compiler reports per kernel (-Rpass-analysis=kernel-resource-usage).
    Every hand-written kernel of the hot path must keep its state in registers/LDS: scratch memory
    costs an HBM round trip per access. Allowed: the three kernels that hold the banded-DP rows of
    the edit predicate in per-thread arrays (single calls / edit verification), and rocPRIM's own."""
    from fastqdedup_amd.build import build, kernel_resources
    build()
    res = kernel_resources()
    assert len(res) > 50
    allowed = ("pairs_within_kernel", "contains_kernel", "edit_verify_kernel", "rocprim")
    ours = {k: v for k, v in res.items() if "rocprim" not in k}
    assert any("grouped_candidates_kernel" in k for k in ours) and any("pack_kernel" in k for k in ours)
    spilling = [k for k, v in ours.items()
                if v.get("ScratchSize [bytes/lane]", 0) and not any(a in k for a in allowed)]
    assert not spilling, spilling

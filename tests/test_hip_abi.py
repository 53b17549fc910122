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


def test_python_surface_has_the_references_names():
    """Every name a user of the reference imports exists under both package names (reference
    src/fastqdedup/__init__.py, _trie.pyi:20-44, _distance.pyi, _fastq.pyi): attribute presence
    and signatures only -- no device needed."""
    import inspect
    import fastqdedup
    import fastqdedup_amd as F
    for name in ("Trie", "within_distance", "cluster_dissection_directional", "cluster_dissection_adjacency",
                 "cluster_dissection_highest_count", "CLUSTER_DISSECTION_METHODS", "deduplicate_cluster", "main",
                 "argument_parser", "length_string_to_slices", "DEFAULT_MAX_DISTANCE", "DEFAULT_PREFIX",
                 "DEFAULT_CLUSTER_DISSECTION", "DEFAULT_MAX_AVERAGE_ERROR_RATE"):
        assert hasattr(F, name) and hasattr(fastqdedup, name), name
    for name in ("trie_stats", "Timer", "initiate_logger", "fastq_average_error_rate"):
        assert hasattr(fastqdedup, name), name
    from fastqdedup._trie import Trie
    from fastqdedup._distance import within_distance  # noqa: F401
    from fastqdedup._fastq import average_error_rate  # noqa: F401
    assert Trie is F.Trie
    for member in ("add_sequence", "contains_sequence", "pop_cluster", "memory_size", "raw_stats", "alphabet",
                   "number_of_sequences"):
        assert hasattr(Trie, member), member
    assert isinstance(inspect.getattr_static(Trie, "alphabet"), property)
    assert isinstance(inspect.getattr_static(Trie, "number_of_sequences"), property)
    assert list(inspect.signature(Trie.pop_cluster).parameters) == ["self", "max_distance", "use_edit_distance"]
    assert list(inspect.signature(Trie.contains_sequence).parameters) == ["self", "sequence", "max_distance",
                                                                          "use_edit_distance"]
    flags = {a.dest for a in F.argument_parser()._actions}
    assert {"fastq", "check_lengths", "output", "prefix", "max_distance", "max_average_error_rate", "edit",
            "cluster_dissection_method", "verbose", "quiet"} <= flags

"""Host-side mirror of the reference's Python surface for the clustering hot path
(reference: src/fastqdedup/__init__.py:60-130, _trie.pyi:20-44, _distance.pyi:19-21),
running on the MI355X through ``libfqdedup_hip.so``.

Same names, argument meaning and exceptions as the reference:
``Trie``, ``within_distance``, ``cluster_dissection_{directional,adjacency,highest_count}``,
``CLUSTER_DISSECTION_METHODS`` -- plus the batch entry ``cluster_keys`` that the
CLI uses so that 50 M reads never become 50 M Python objects.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from ._lib import METHODS, METRIC_EDIT, METRIC_HAMMING, Context

DEFAULT_MAX_DISTANCE = 1

_default_ctx: Optional[Context] = None


def default_context() -> Context:
    """Process-wide context on device ``FQD_DEVICE`` (else ``LOCAL_RANK``, else 0)."""
    global _default_ctx
    if _default_ctx is None:
        dev = int(os.environ.get("FQD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _default_ctx = Context(dev)
    return _default_ctx


def _metric(use_edit_distance: bool) -> int:
    return METRIC_EDIT if use_edit_distance else METRIC_HAMMING


def _method_id(method) -> int:
    if isinstance(method, str):
        try:
            return METHODS[method]
        except KeyError:
            raise ValueError(f"unknown cluster dissection method {method!r}") from None
    return int(method)


def pack_strings(strings: Sequence[str], what: str = "Sequence") -> Tuple[np.ndarray, np.ndarray]:
    """list[str] -> (uint8 bytes, uint64 offsets[n+1]); ASCII only (_triemodule.c:684-688)."""
    enc = []
    for s in strings:
        if type(s) is not str:
            raise TypeError(f"{what} must be a str, got {type(s).__name__}")
        try:
            enc.append(s.encode("ascii"))
        except UnicodeEncodeError:
            raise ValueError(f"{what} must consist only of ASCII characters") from None
    off = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        off[1:] = np.cumsum(np.fromiter((len(e) for e in enc), dtype=np.uint64, count=len(enc)))
    raw = np.frombuffer(b"".join(enc), dtype=np.uint8) if enc else np.zeros(0, dtype=np.uint8)
    return raw, off


@dataclass
class ClusterResult:
    kept_read_ids: np.ndarray          # ascending first-holder ids of the kept keys
    n_reads: int
    n_counted: int                     # == Trie.number_of_sequences after pass 1
    n_unique: int
    n_edges: int
    n_clusters: int                    # == pop_cluster calls in the reference loop
    n_kept: int                        # == len(deduplicated_set)
    stage_ms: Dict[str, float]


def cluster_keys(keys, offsets=None, key_len: int = 0, weights=None, read_ids=None, *,
                 max_distance: int = DEFAULT_MAX_DISTANCE, use_edit_distance: bool = False,
                 method="directional", context: Optional[Context] = None,
                 kept_out=None, stage_times: bool = True) -> ClusterResult:
    """The whole hot path in one call (reference __init__.py:240-281 + the pass-2
    selection rule :201-206). ``keys`` is the concatenation of the key bytes --
    a numpy uint8 array (host) or a torch uint8 tensor already in HBM --
    with either ``offsets`` (n+1, uint64) or a fixed ``key_len``. ``weights`` is
    1 for reads that passed the quality filter and 0 for reads that did not
    (None = all passed). Returns the ids of the reads the reference would write."""
    if max_distance < 0:
        raise ValueError("max_distance should be non-negative")
    ctx = context or default_context()
    direct = kept_out is not None and hasattr(kept_out, "is_cuda") and kept_out.is_cuda
    if direct:
        ctx.set_kept_output(kept_out)      # the list is written into the caller's buffer, no copy
    try:
        s = ctx.cluster_keys(keys, offsets, key_len, weights, read_ids, max_distance=max_distance,
                             metric=_metric(use_edit_distance), method=_method_id(method))
        kept = ctx.kept_read_ids(s["n_kept"], kept_out)
    finally:
        if direct:
            ctx.set_kept_output(None)
    ms = ctx.stage_times()[0] if stage_times else {}     # (resolving the event pairs costs ~20 us of host time)
    return ClusterResult(kept, s["n_reads"], s["n_counted"], s["n_unique"], s["n_edges"],
                         s["n_clusters"], s["n_kept"], ms)


# ---------------------------------------------------------------------------
# within_distance  (reference _distancemodule.c:46-93)
# ---------------------------------------------------------------------------

def within_distance(string1, string2, /, max_distance, use_edit_distance=False) -> bool:
    if not isinstance(string1, str):
        raise TypeError(f"within_distance() argument 1 must be str, not {type(string1).__name__}")
    if not isinstance(string2, str):
        raise TypeError(f"within_distance() argument 2 must be str, not {type(string2).__name__}")
    if not isinstance(max_distance, int):
        raise TypeError(f"'{type(max_distance).__name__}' object cannot be interpreted as an integer")
    try:
        b1 = string1.encode("latin-1")
    except UnicodeEncodeError:
        raise ValueError("string1 must be ASCII or latin-1 encoded.") from None
    try:
        b2 = string2.encode("latin-1")
    except UnicodeEncodeError:
        raise ValueError("string2 must be ASCII or latin-1 encoded.") from None
    a = np.frombuffer(b1 or b"\0", dtype=np.uint8)
    b = np.frombuffer(b2 or b"\0", dtype=np.uint8)
    ao = np.array([0, len(b1)], dtype=np.uint64)
    bo = np.array([0, len(b2)], dtype=np.uint64)
    out = default_context().within_distance(a, ao, b, bo, int(max_distance), _metric(use_edit_distance))
    return bool(out[0])


# ---------------------------------------------------------------------------
# cluster dissection  (reference __init__.py:60-130)
# ---------------------------------------------------------------------------

def _dissect_on_device(cluster, method: int, max_distance: int, use_edit_distance: bool) -> List[str]:
    items = list(cluster)
    if not items:
        if method == METHODS["highest_count"]:
            raise IndexError("list index out of range")  # cluster[0] in the reference (:101)
        return []
    counts = np.fromiter((int(c) for c, _ in items), dtype=np.int64, count=len(items))
    if (counts < 0).any() or (counts > 0xFFFFFFFF).any():
        raise ValueError("counts must fit 32 bits")
    raw, off = pack_strings([s for _, s in items], "cluster key")
    ctx = default_context()
    n = ctx.pack_keys(raw, off)
    shape = ctx.shape()
    recs = np.empty(n * shape.stride_words, dtype=np.uint32)
    lens = np.empty(n, dtype=np.uint32)
    hashes = np.empty(n, dtype=np.uint32)
    ctx.export_packed(recs, lens, hashes)
    # every list item is its own node, even if a key repeats (the reference
    # compares items pairwise and never merges them)
    ctx.import_unique(recs, lens, counts.astype(np.uint32), np.arange(n, dtype=np.uint64), n)
    if method == METHODS["highest_count"]:
        # the reference takes the maximum of the LIST it is given, connected or not (:99-102):
        # a star over all items makes the list one component
        star = np.zeros((max(n - 1, 0), 2), dtype=np.uint32)
        star[:, 1] = np.arange(1, n, dtype=np.uint32)
        ctx.import_edges(star.reshape(-1) if n > 1 else np.zeros(2, dtype=np.uint32), n - 1)
    else:
        ctx.find_edges(max_distance, _metric(use_edit_distance))
    ctx.components()
    n_kept = ctx.dissect(method)
    kept_idx = ctx.kept_read_ids(n_kept)
    picked = [items[int(i)] for i in kept_idx]
    # the reference yields roots in descending (count, key) order
    picked.sort(reverse=True)
    return [s for _, s in picked]


def cluster_dissection_directional(cluster: List[Tuple[int, str]],
                                   max_distance: int = DEFAULT_MAX_DISTANCE,
                                   use_edit_distance: bool = False) -> Iterator[str]:
    """reference __init__.py:60-91"""
    yield from _dissect_on_device(cluster, METHODS["directional"], max_distance, use_edit_distance)


def cluster_dissection_highest_count(cluster: List[Tuple[int, str]],
                                     max_distance: int = DEFAULT_MAX_DISTANCE,
                                     use_edit_distance: bool = False) -> Iterator[str]:
    """reference __init__.py:94-102"""
    yield from _dissect_on_device(cluster, METHODS["highest_count"], max_distance, use_edit_distance)


def cluster_dissection_adjacency(cluster: List[Tuple[int, str]],
                                 max_distance: int = DEFAULT_MAX_DISTANCE,
                                 use_edit_distance: bool = False) -> Iterator[str]:
    """reference __init__.py:105-122"""
    yield from _dissect_on_device(cluster, METHODS["adjacency"], max_distance, use_edit_distance)


CLUSTER_DISSECTION_METHODS = {
    "highest_count": cluster_dissection_highest_count,
    "adjacency": cluster_dissection_adjacency,
    "directional": cluster_dissection_directional,
}


# ---------------------------------------------------------------------------
# Trie  (reference _triemodule.c:596-1009, _trie.pyi:20-44)
# ---------------------------------------------------------------------------

class Trie:
    """Drop-in for ``fastqdedup.Trie`` backed by the device key store.

    ``add_sequence`` only appends to a host-side staging list; the first
    ``pop_cluster`` / ``contains_sequence`` after an add moves the keys to HBM and
    clusters them there (pack -> collapse -> bucket pair search -> components).
    Popped clusters are the connected components the reference's BFS extracts
    (_triemodule.c:865-895), emitted in the reference's order: ascending seed key
    in trie-alphabet order, a longer key before its own prefix (:510-551).

    Representation-specific introspection differs by design:
    ``alphabet`` lists the constructor alphabet followed by new symbols in order
    of first appearance (the reference registers a symbol only when an inner
    node looks it up, :266-273); ``memory_size()`` reports the HBM footprint of
    the packed store; ``raw_stats()`` (per-layer node census of a pointer trie)
    does not exist for a bit-plane store.
    """

    def __init__(self, alphabet: str = ""):
        if not isinstance(alphabet, str):
            raise TypeError(f"Trie.__new__() argument 'alphabet' must be str, not {type(alphabet).__name__}")
        try:
            alphabet.encode("ascii")
        except UnicodeEncodeError:
            raise ValueError("Alphabet should be an ASCII string.") from None
        if len(alphabet) > 254:
            raise ValueError("Maximum alphabet length exceeded")
        seen = set()
        for ch in alphabet:
            if ch in seen:
                raise ValueError("Alphabet should consist of unique characters."
                                 f"Character {ch} was repeated. ")
            seen.add(ch)
        self._alphabet: List[str] = list(alphabet)
        self._seen = seen
        self._keys: List[str] = []      # unique keys in the store + staged adds
        self._weights: List[int] = []
        self._nseq = 0
        self._clusters: Optional[List[List[Tuple[int, str]]]] = None   # remaining, emission order
        self._params: Optional[Tuple[int, bool]] = None
        default_context()  # fail now, loudly, when there is no device

    # -- properties ---------------------------------------------------------
    @property
    def alphabet(self) -> str:
        return "".join(self._alphabet)

    @property
    def number_of_sequences(self) -> int:
        return self._nseq

    # -- mutation -----------------------------------------------------------
    def add_sequence(self, sequence, /) -> None:
        if type(sequence) is not str:
            raise TypeError(f"Sequence must be a str, got {type(sequence).__name__}")
        if not sequence.isascii():
            raise ValueError("Sequence must consist only of ASCII characters")
        if len(sequence) > 0xFFFFFFFF:
            raise ValueError("Sequences larger than 4294967295 can not be stored in the Trie")
        self._unpop()
        self._keys.append(sequence)
        self._weights.append(1)
        self._nseq += 1
        fresh = set(sequence) - self._seen
        if fresh:
            for ch in sequence:
                if ch in fresh and ch not in self._seen:
                    self._seen.add(ch)
                    self._alphabet.append(ch)

    def _unpop(self):
        """Adds after pops: the not-yet-popped clusters go back into the store."""
        if self._clusters is not None:
            self._keys = [k for cl in self._clusters for _, k in cl]
            self._weights = [c for cl in self._clusters for c, _ in cl]
            self._clusters = None
            self._params = None

    def _order_key(self):
        rank = {ch: i for i, ch in enumerate(self._alphabet)}
        end = len(rank) + 1

        def key(s: str):
            return tuple(rank[ch] for ch in s) + (end,)   # longer key before its prefix
        return key

    def _cluster(self, max_distance: int, use_edit_distance: bool):
        self._unpop()
        ctx = default_context()
        raw, off = pack_strings(self._keys)
        w = np.asarray(self._weights, dtype=np.uint32)
        ctx.pack_keys(raw, off)
        nu = ctx.collapse(w, None)
        ctx.find_edges(max_distance, _metric(use_edit_distance))
        ctx.components()
        first, counts, labels, _ = ctx.unique_table(nu, labels=True, kept=False)
        keys = [self._keys[int(i)] for i in first]
        order = np.argsort(labels, kind="stable")
        groups: List[List[Tuple[int, str]]] = []
        okey = self._order_key()
        start = 0
        lab_sorted = labels[order]
        for end in range(1, nu + 1):
            if end == nu or lab_sorted[end] != lab_sorted[start]:
                members = [(int(counts[j]), keys[int(j)]) for j in order[start:end]]
                members.sort(key=lambda m: okey(m[1]))
                groups.append(members)
                start = end
        groups.sort(key=lambda g: okey(g[0][1]))
        self._clusters = groups
        self._params = (max_distance, bool(use_edit_distance))

    def pop_cluster(self, max_distance, use_edit_distance=False) -> List[Tuple[int, str]]:
        if not isinstance(max_distance, int):
            raise TypeError(f"'{type(max_distance).__name__}' object cannot be interpreted as an integer")
        if max_distance < 0:
            raise ValueError("max_distance should be non-negative")
        if self._nseq == 0:
            raise LookupError("No sequences left in Trie.")
        if self._clusters is None or self._params != (max_distance, bool(use_edit_distance)):
            self._cluster(max_distance, bool(use_edit_distance))
        cluster = self._clusters.pop(0)
        self._nseq -= sum(c for c, _ in cluster)
        return cluster

    def contains_sequence(self, sequence, /, max_distance: int = 0, use_edit_distance: bool = False) -> bool:
        if not isinstance(sequence, str):
            raise TypeError(f"contains_sequence() argument 1 must be str, not {type(sequence).__name__}")
        if not sequence.isascii():
            raise ValueError("sequence must contain only ASCII characters")
        if self._nseq == 0:
            return False   # the reference dereferences a NULL root here (_triemodule.c:755)
        self._unpop()
        ctx = default_context()
        raw, off = pack_strings(self._keys)
        ctx.pack_keys(raw, off)
        ctx.collapse(np.asarray(self._weights, dtype=np.uint32), None)
        q, qo = pack_strings([sequence])
        q = q if q.size else np.zeros(1, dtype=np.uint8)
        return bool(ctx.contains(q, qo, int(max_distance), _metric(use_edit_distance))[0])

    # -- introspection ------------------------------------------------------
    def memory_size(self) -> int:
        """HBM bytes of the packed key store for the current keys (not comparable
        with the reference's node bytes, _triemodule.c:553-570)."""
        if not self._keys:
            return 0
        self._unpop()
        ctx = default_context()
        raw, off = pack_strings(self._keys)
        ctx.pack_keys(raw, off)
        nu = ctx.collapse(np.asarray(self._weights, dtype=np.uint32), None)
        sh = ctx.shape()
        return nu * (sh.stride_words * 4 + 4 + 8 + (4 if sh.ragged else 0))

    def raw_stats(self):
        raise NotImplementedError(
            "raw_stats() is a per-layer node census of the reference's pointer trie "
            "(_triemodule.c:572-594); the device store is a flat bit-plane table and has no layers")

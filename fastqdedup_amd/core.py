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
    route: Dict[str, bool] = None      # which way the job took through the library (Context.route)


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
                         s["n_clusters"], s["n_kept"], ms, ctx.route())


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

def lazily_registered(ctx: Context, registered: str, unregistered) -> List[str]:
    """The symbols of ``unregistered`` that an inner node of the reference's trie would have
    looked up by now (TrieNode_AddSequence, _triemodule.c:266-273), in the order it would have met
    them, for the keys stored in ``ctx``: per symbol the earliest (time, depth,
    new-key-after-old-key) over the keys that hold it; one device round
    (``fqd_store_symbol_events``) per candidate key -- usually one or two."""
    listing = registered + "".join(sorted(unregistered))
    best = {c: None for c in unregistered}
    after = {c: None for c in unregistered}
    searching = sorted(unregistered)
    reuse = False
    for _ in range(256):                       # (a symbol that needs more candidates keeps its best so far)
        if not searching:
            break
        rows = ctx.store_symbol_events(listing, "".join(searching), [after[c] for c in searching], reuse)
        reuse = True
        still = []
        for c, (fid, depth, partner) in zip(searching, rows):
            if fid is None or (best[c] is not None and fid >= best[c][0]):
                continue                       # no further key can be looked up earlier
            if partner is not None:
                event = (max(fid, partner), depth, 1 if fid > partner else 0)
                if best[c] is None or event < best[c]:
                    best[c] = event
            after[c] = fid
            still.append(c)
        searching = still
    return [c for _, c in sorted((e, c) for c, e in best.items() if e is not None)]


class TableCensus:
    """What ``trie_stats`` needs (``alphabet``, ``raw_stats()``, ``memory_size()``; reference
    __init__.py:133-157) for the unique table a context holds after ``cluster_keys``: the trie
    ``deduplicate_cluster`` would have built from the same reads (``Trie(alphabet="ACGTN")``,
    __init__.py:240)."""

    def __init__(self, ctx: Context, alphabet: str = "ACGTN"):
        sh = ctx.shape()
        present = {chr(b) for b in bytes(sh.alphabet)[: int(sh.alphabet_size)]}
        others = present - set(alphabet)
        if others:
            alphabet += "".join(lazily_registered(ctx, alphabet, others))
            others -= set(alphabet)
        self.alphabet = alphabet
        self._memory, stats = ctx.trie_stats(alphabet + "".join(sorted(others)), int(sh.max_len) + 1)
        self._stats = [row[: len(alphabet) + 1] for row in stats]

    def raw_stats(self) -> List[List[int]]:
        return self._stats

    def memory_size(self) -> int:
        return self._memory


class Trie:
    """Drop-in for ``fastqdedup.Trie`` (reference _triemodule.c:596-1009, _trie.pyi:20-44) over a
    device-resident store: its own context holds the unique-key table; nothing is re-packed.

    * ``add_sequence`` appends to a host-side list of pending adds; the next query packs ONLY those
      and merges them into the resident table on the device (``fqd_store_add_keys``).
    * ``pop_cluster`` clusters the table on the device (bucket pair search -> components) and
      asks for the clusters in the reference's order (``fqd_get_clusters``: ascending seed key in
      alphabet order, a longer key before its own prefix, _triemodule.c:510-551); Python objects
      are made for the popped cluster only. Popped rows are marked removed on the device
      (``fqd_store_remove``); an add between pops merges and clusters again.
    * ``contains_sequence`` searches the resident table (removed rows are skipped).
    * ``memory_size`` / ``raw_stats`` are the node census of the trie the reference would hold
      (``fqd_trie_stats``): exact after adds and after adds followed by pops -- what
      ``deduplicate_cluster`` does; an add into a partly popped trie is merged as if the popped keys
      had never been stored.

    * ``alphabet`` grows the way the reference's does: a symbol outside the constructor alphabet is
      registered when an INNER node first looks it up (_triemodule.c:266-273) -- not when it first
      appears, never for bases that only ever sat in a leaf's suffix. The moments are worked out on
      the device from the stored keys (``fqd_store_symbol_events``) whenever the alphabet, the pop
      order or the census is asked for while unregistered symbols exist.
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
        self._alphabet: List[str] = list(alphabet)    # registered symbols, index order
        self._seen = seen
        self._unregistered: set = set()               # symbols met in keys, not (yet) looked up by an inner node
        self._strings: List[str] = []   # every sequence ever added; a read id is its index here
        self._merged = 0                # strings[:_merged] are in the device store
        self._nseq = 0
        self._max_len = 0               # max_sequence_size of the reference: never shrinks (:700-702)
        self._n_unique = 0              # rows of the resident table
        self._offsets: Optional[np.ndarray] = None   # remaining clusters (CSR over table rows), pop order
        self._members: Optional[np.ndarray] = None
        self._cursor = 0
        self._params: Optional[Tuple[int, bool]] = None
        self._first: Optional[np.ndarray] = None     # per row: first holder id, count
        self._counts: Optional[np.ndarray] = None
        default_context()               # fail now, loudly, when there is no device
        self._ctx: Optional[Context] = None

    # -- properties ---------------------------------------------------------
    @property
    def alphabet(self) -> str:
        self._resolve_alphabet()
        return "".join(self._alphabet)

    def _listing(self) -> str:
        """Every symbol of the stored keys: the registered ones in index order, then the others."""
        return "".join(self._alphabet) + "".join(sorted(self._unregistered))

    def _resolve_alphabet(self) -> None:
        """Register the symbols an inner node of the reference's trie has looked up by now."""
        if not self._unregistered or self._nseq == 0:
            return
        self._flush()
        for c in lazily_registered(self._store(), "".join(self._alphabet), self._unregistered):
            self._alphabet.append(c)
            self._unregistered.discard(c)

    @property
    def number_of_sequences(self) -> int:
        return self._nseq

    # -- the device store ---------------------------------------------------
    def _store(self) -> Context:
        if self._ctx is None:
            self._ctx = Context(default_context().device)
        return self._ctx

    def _flush(self) -> None:
        """Pending adds -> the resident table (only they are packed)."""
        if self._merged == len(self._strings):
            return
        pending = self._strings[self._merged:]
        raw, off = pack_strings(pending)
        ids = np.arange(self._merged, len(self._strings), dtype=np.uint64)
        data = raw if raw.size else np.zeros(1, dtype=np.uint8)
        self._n_unique = self._store().store_add_keys(data, off, 0, None, ids)
        self._merged = len(self._strings)
        self._offsets = self._members = self._first = self._counts = None    # rows were renumbered

    # -- mutation -----------------------------------------------------------
    def add_sequence(self, sequence, /) -> None:
        if type(sequence) is not str:
            raise TypeError(f"Sequence must be a str, got {type(sequence).__name__}")
        if not sequence.isascii():
            raise ValueError("Sequence must consist only of ASCII characters")
        if len(sequence) > 0xFFFFFFFF:
            raise ValueError("Sequences larger than 4294967295 can not be stored in the Trie")
        self._strings.append(sequence)
        self._nseq += 1
        self._max_len = max(self._max_len, len(sequence))
        fresh = set(sequence) - self._seen
        if fresh:
            self._seen |= fresh
            self._unregistered |= fresh

    def _cluster(self, max_distance: int, use_edit_distance: bool) -> None:
        ctx = self._store()
        self._resolve_alphabet()
        ctx.find_edges(max_distance, _metric(use_edit_distance))
        ctx.components()
        self._offsets, self._members = ctx.clusters(self._listing())
        self._first, self._counts, _, _ = ctx.unique_table(self._n_unique, labels=False, kept=False)
        self._cursor = 0
        self._params = (max_distance, bool(use_edit_distance))

    def pop_cluster(self, max_distance, use_edit_distance=False) -> List[Tuple[int, str]]:
        if not isinstance(max_distance, int):
            raise TypeError(f"'{type(max_distance).__name__}' object cannot be interpreted as an integer")
        if max_distance < 0:
            raise ValueError("max_distance should be non-negative")
        if self._nseq == 0:
            raise LookupError("No sequences left in Trie.")
        self._flush()
        if self._offsets is None or self._params != (max_distance, bool(use_edit_distance)):
            self._cluster(max_distance, bool(use_edit_distance))
        lo, hi = int(self._offsets[self._cursor]), int(self._offsets[self._cursor + 1])
        self._cursor += 1
        rows = self._members[lo:hi]
        cluster = [(int(self._counts[r]), self._strings[int(self._first[r])]) for r in rows]
        self._store().store_remove(np.ascontiguousarray(rows))
        self._nseq -= sum(c for c, _ in cluster)
        return cluster

    def contains_sequence(self, sequence, /, max_distance: int = 0, use_edit_distance: bool = False) -> bool:
        if not isinstance(sequence, str):
            raise TypeError(f"contains_sequence() argument 1 must be str, not {type(sequence).__name__}")
        if not sequence.isascii():
            raise ValueError("sequence must contain only ASCII characters")
        if self._nseq == 0:
            return False   # the reference dereferences a NULL root here (_triemodule.c:755)
        self._flush()
        q, qo = pack_strings([sequence])
        q = q if q.size else np.zeros(1, dtype=np.uint8)
        return bool(self._store().contains(q, qo, int(max_distance), _metric(use_edit_distance))[0])

    # -- introspection ------------------------------------------------------
    def _census(self):
        layers = self._max_len + 1
        if self._nseq == 0:
            return 0, [[0] * (len(self.alphabet) + 1) for _ in range(layers)]
        self._flush()
        alphabet = self.alphabet                   # (registers what has been looked up by now)
        memory, stats = self._store().trie_stats(self._listing(), layers)
        return memory, [row[: len(alphabet) + 1] for row in stats]

    def memory_size(self) -> int:
        """Bytes of the reference's trie nodes for the stored keys (_triemodule.c:553-570, :909-913)."""
        return self._census()[0]

    def raw_stats(self) -> List[List[int]]:
        """Per trie layer: leaves, then inner nodes by child-array width (_triemodule.c:572-594, :929-964)."""
        return self._census()[1]

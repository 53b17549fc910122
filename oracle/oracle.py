"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes face of ``oracle/libfqd_oracle.so`` (plain-C restatement of the
reference hot path, see ``fqd_oracle.h``) with the same Python surface as the
reference (``Trie``, ``within_distance``, ``cluster_dissection_*``), plus
``load_reference()`` which imports the reference's own C extensions from
``oracle/_ref`` when that directory has been built (``make -C oracle ref``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module. ``fastqdedup_amd`` never does.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import subprocess
import sys
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfqd_oracle.so")

METHODS = {"highest_count": 0, "adjacency": 1, "directional": 2}

_E_NOMEM, _E_VALUE, _E_LOOKUP, _E_RUNTIME = -1, -2, -3, -4


def build(force: bool = False) -> None:
    """Compile the C restatement (and oracle/_ref when the reference is here)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "fqd_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "libfqd_oracle.so"],
                              stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.fqo_within_hamming.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
    L.fqo_within_edit.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
    L.fqo_trie_new.restype = C.c_void_p
    L.fqo_trie_new.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), u8p]
    L.fqo_trie_free.argtypes = [C.c_void_p]
    L.fqo_trie_add.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32]
    L.fqo_trie_contains.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_int, C.c_int]
    L.fqo_trie_number_of_sequences.restype = C.c_int64
    L.fqo_trie_number_of_sequences.argtypes = [C.c_void_p]
    L.fqo_trie_max_sequence_size.restype = C.c_uint32
    L.fqo_trie_max_sequence_size.argtypes = [C.c_void_p]
    L.fqo_trie_alphabet.restype = C.c_size_t
    L.fqo_trie_alphabet.argtypes = [C.c_void_p, C.c_char_p]
    L.fqo_trie_memory_size.restype = C.c_size_t
    L.fqo_trie_memory_size.argtypes = [C.c_void_p]
    L.fqo_trie_raw_stats.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    L.fqo_trie_pop_cluster.restype = C.c_int64
    L.fqo_trie_pop_cluster.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.fqo_cluster_bytes.restype = u8p
    L.fqo_cluster_bytes.argtypes = [C.c_void_p]
    L.fqo_cluster_offsets.restype = u64p
    L.fqo_cluster_offsets.argtypes = [C.c_void_p]
    L.fqo_cluster_counts.restype = u32p
    L.fqo_cluster_counts.argtypes = [C.c_void_p]
    L.fqo_dissect.restype = C.c_int64
    L.fqo_dissect.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                              C.c_int, C.c_int, C.c_void_p]
    L.fqo_dedup.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_int,
                            C.c_int, C.c_void_p, u64p, u64p, u64p, C.POINTER(C.c_double)]
    L.fqo_average_error_rate.argtypes = [C.c_char_p, C.c_size_t, C.c_uint8, C.POINTER(C.c_double), u8p]
    _lib = L
    return L


def _raise(code: int, what: str = ""):
    if code == _E_NOMEM:
        raise MemoryError(what)
    if code == _E_VALUE:
        raise ValueError(what or "invalid value")
    if code == _E_LOOKUP:
        raise LookupError("No sequences left in Trie.")
    raise RuntimeError(what or f"oracle error {code}")


def _ascii(s: str, what: str) -> bytes:
    if not isinstance(s, str):
        raise TypeError(f"{what} must be a str, got {type(s).__name__}")
    try:
        return s.encode("ascii")
    except UnicodeEncodeError:
        raise ValueError(f"{what} must consist only of ASCII characters") from None


def within_distance(s1: str, s2: str, /, max_distance: int, use_edit_distance: bool = False) -> bool:
    """_distancemodule.c:46-93 (1-byte-kind strings only)."""
    if not isinstance(s1, str) or not isinstance(s2, str):
        raise TypeError("within_distance() arguments 1 and 2 must be str")
    try:
        b1, b2 = s1.encode("latin-1"), s2.encode("latin-1")
    except UnicodeEncodeError:
        raise ValueError("strings must be ASCII or latin-1 encoded.") from None
    f = lib().fqo_within_edit if use_edit_distance else lib().fqo_within_hamming
    return bool(f(b1, len(b1), b2, len(b2), int(max_distance)))


def average_error_rate(phred_scores: str, *, phred_offset: int = 33) -> float:
    """_fastqmodule.c:38-76"""
    if not isinstance(phred_scores, str):
        raise TypeError("phred_scores must be str")
    if not phred_scores.isascii():
        raise ValueError("phred_scores must be ASCII encoded.")
    b = phred_scores.encode("ascii")
    out, bad = C.c_double(0.0), C.c_uint8(0)
    rc = lib().fqo_average_error_rate(b, len(b), phred_offset, C.byref(out), C.byref(bad))
    if rc:
        raise ValueError(f"Character {chr(bad.value)} outside of valid phred range "
                         f"('{chr(phred_offset)}' to '{chr(126)}')")
    return out.value


class Trie:
    """Same surface as the reference's ``fastqdedup._trie.Trie`` (_trie.pyi:20-44)."""

    def __init__(self, alphabet: str = ""):
        if not isinstance(alphabet, str):
            raise TypeError("alphabet must be str")
        try:
            ab = alphabet.encode("ascii")
        except UnicodeEncodeError:
            raise ValueError("Alphabet should be an ASCII string.") from None
        err = C.c_int(0)
        rep = C.c_uint8(0)
        self._h = lib().fqo_trie_new(ab, len(ab), C.byref(err), C.byref(rep))
        if not self._h:
            if err.value == _E_VALUE and rep.value:
                raise ValueError("Alphabet should consist of unique characters."
                                 f"Character {chr(rep.value)} was repeated. ")
            if err.value == _E_VALUE:
                raise ValueError("Maximum alphabet length exceeded")
            _raise(err.value)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:
            _lib.fqo_trie_free(h)
            self._h = None

    def add_sequence(self, sequence: str, /, count: int = 1) -> None:
        b = _ascii(sequence, "Sequence")
        rc = lib().fqo_trie_add(self._h, b, len(b), count)
        if rc:
            _raise(rc)

    def contains_sequence(self, sequence: str, /, max_distance: int = 0,
                          use_edit_distance: bool = False) -> bool:
        b = _ascii(sequence, "sequence")
        return bool(lib().fqo_trie_contains(self._h, b, len(b), int(max_distance),
                                            int(bool(use_edit_distance))))

    def pop_cluster(self, max_distance: int, use_edit_distance: bool = False
                    ) -> List[Tuple[int, str]]:
        L = lib()
        if max_distance < 0:
            raise ValueError("max_distance should be non-negative")
        n = L.fqo_trie_pop_cluster(self._h, int(max_distance), int(bool(use_edit_distance)))
        if n < 0:
            _raise(int(n))
        off = np.ctypeslib.as_array(L.fqo_cluster_offsets(self._h), shape=(n + 1,))
        cnt = np.ctypeslib.as_array(L.fqo_cluster_counts(self._h), shape=(n,))
        total = int(off[n])
        raw = bytes(np.ctypeslib.as_array(L.fqo_cluster_bytes(self._h), shape=(max(total, 1),))[:total])
        return [(int(cnt[i]), raw[int(off[i]):int(off[i + 1])].decode("ascii")) for i in range(n)]

    def memory_size(self) -> int:
        return int(lib().fqo_trie_memory_size(self._h))

    def raw_stats(self) -> List[List[int]]:
        L = lib()
        cols = len(self.alphabet) + 1
        layers = int(L.fqo_trie_max_sequence_size(self._h)) + 1
        buf = (C.c_size_t * (cols * layers))()
        L.fqo_trie_raw_stats(self._h, buf)
        return [[int(buf[r * cols + c]) for c in range(cols)] for r in range(layers)]

    @property
    def alphabet(self) -> str:
        buf = C.create_string_buffer(256)
        n = lib().fqo_trie_alphabet(self._h, buf)
        return buf.raw[:n].decode("latin-1")

    @property
    def number_of_sequences(self) -> int:
        return int(lib().fqo_trie_number_of_sequences(self._h))


def _pack_cluster(cluster: Sequence[Tuple[int, str]]):
    counts = np.fromiter((c for c, _ in cluster), dtype=np.uint32, count=len(cluster))
    enc = [s.encode("latin-1") for _, s in cluster]
    off = np.zeros(len(cluster) + 1, dtype=np.uint64)
    if enc:
        off[1:] = np.cumsum([len(e) for e in enc], dtype=np.uint64)
    raw = np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8)
    return counts, raw, off


def _dissect(method: int, cluster, max_distance: int, use_edit_distance: bool) -> Iterator[str]:
    cluster = list(cluster)
    if not cluster:
        if method == 0:
            raise IndexError("list index out of range")  # cluster[0] in the reference
        return
    counts, raw, off = _pack_cluster(cluster)
    out = np.zeros(len(cluster), dtype=np.uint64)
    k = lib().fqo_dissect(method, counts.ctypes.data, raw.ctypes.data, off.ctypes.data,
                          len(cluster), int(max_distance), int(bool(use_edit_distance)),
                          out.ctypes.data)
    if k < 0:
        _raise(int(k))
    for i in out[:k]:
        yield cluster[int(i)][1]


def cluster_dissection_directional(cluster, max_distance: int = 1, use_edit_distance: bool = False):
    """__init__.py:60-91"""
    return _dissect(2, cluster, max_distance, use_edit_distance)


def cluster_dissection_highest_count(cluster, max_distance: int = 1, use_edit_distance: bool = False):
    """__init__.py:94-102"""
    return _dissect(0, cluster, max_distance, use_edit_distance)


def cluster_dissection_adjacency(cluster, max_distance: int = 1, use_edit_distance: bool = False):
    """__init__.py:105-122"""
    return _dissect(1, cluster, max_distance, use_edit_distance)


CLUSTER_DISSECTION_METHODS = {
    "highest_count": cluster_dissection_highest_count,
    "adjacency": cluster_dissection_adjacency,
    "directional": cluster_dissection_directional,
}


def dedup(keys: np.ndarray, offsets: np.ndarray, weights: Optional[np.ndarray] = None, *,
          max_distance: int = 1, use_edit_distance: bool = False, method: str = "directional"):
    """Whole hot path on the CPU (``fqo_dedup``). Returns a dict with the sorted
    first-holder read ids of the kept keys and the counters the reference logs."""
    keys = np.ascontiguousarray(keys, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.uint32)
    out = np.zeros(max(n, 1), dtype=np.uint64)
    nk, ncl, nu = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    st = (C.c_double * 3)()
    kp = keys.ctypes.data if keys.size else np.zeros(1, np.uint8).ctypes.data
    rc = lib().fqo_dedup(kp, offsets.ctypes.data, n, None if w is None else w.ctypes.data,
                         int(max_distance), int(bool(use_edit_distance)), METHODS[method],
                         out.ctypes.data, C.byref(nk), C.byref(ncl), C.byref(nu), st)
    if rc:
        _raise(rc)
    return {"kept_read_ids": out[:nk.value].copy(), "n_clusters": ncl.value,
            "n_unique": nu.value, "stage_seconds": {"insert": st[0], "pop_cluster": st[1],
                                                    "dissect": st[2]}}


# ---------------------------------------------------------------------------
# The real reference (oracle/_ref): its own C extensions compiled from
# /root/reference where they lie. Present in the build container and -- as
# prebuilt .so files -- on the GPU box; never in git.
# ---------------------------------------------------------------------------

def reference_available() -> bool:
    d = os.path.join(_HERE, "_ref")
    return os.path.isdir(d) and any(f.startswith("_trie") for f in os.listdir(d))


def load_reference():
    """Returns (ref_trie_module, ref_distance_module) or raises ImportError."""
    d = os.path.join(_HERE, "_ref")
    mods = []
    for name in ("_trie", "_distance"):
        cand = [f for f in (os.listdir(d) if os.path.isdir(d) else []) if f.startswith(name + ".")]
        if not cand:
            raise ImportError(f"oracle/_ref/{name}*.so not built (make -C oracle ref)")
        spec = importlib.util.spec_from_file_location(name, os.path.join(d, cand[0]))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mods.append(mod)
    return tuple(mods)


if __name__ == "__main__":
    build(force=True)
    print("oracle built; reference available:", reference_available(), file=sys.stderr)

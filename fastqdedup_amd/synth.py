"""Deterministic synthetic read keys.

Counter-based: every base is a pure function of (seed, read index, position),
so the numpy generator here and the HIP generator in ``csrc/synth.hip``
(``fqd_synth_keys``) produce byte-identical keys; tests check that on the GPU.

Model: ``M = max(1, n // copies)`` molecules; ``F = max(1, M // 4)`` inserts.
Molecule ``m`` = ``umi`` random bases followed by the ``L - umi`` bases of
insert ``h(seed, 1, m) mod F`` (so different molecules can share an insert --
the same-fragment/different-UMI case that creates false bucket candidates).
Read ``r`` copies molecule ``h(seed, 0, r) mod M`` (about Poisson(copies)
copies each) and then, per base, becomes ``N`` with probability ``n_rate`` or
one of the three other bases with probability ``sub_rate``
(reference README.rst:120-122 motivates 1e-3).
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_MUL1 = np.uint64(0xBF58476D1CE4E5B9)
_MUL2 = np.uint64(0x94D049BB133111EB)
_STREAM = np.uint64(0xD1B54A32D192ED03)

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + _GOLD
        z = (z ^ (z >> np.uint64(30))) * _MUL1
        z = (z ^ (z >> np.uint64(27))) * _MUL2
        return z ^ (z >> np.uint64(31))


def stream_hash(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """h(seed, stream, idx) = splitmix64(splitmix64(seed + stream * C) ^ idx)."""
    with np.errstate(over="ignore"):
        base = splitmix64(np.array([np.uint64(seed) + np.uint64(stream) * _STREAM],
                                   dtype=np.uint64))[0]
    return splitmix64(base ^ idx.astype(np.uint64))


def rate_threshold(p: float) -> int:
    """Probability -> threshold on the top 53 bits of a hash (exact on host and device)."""
    return int(p * float(1 << 53))


# The SKEWED model (``skew=SKEW`` below; what real libraries look like and the uniform model does not):
#   * molecule abundance is heavy-tailed: read r copies molecule floor(M * x^2), x uniform in [0, 1) -- molecule m
#     draws mass ~ 1/sqrt(m) -- and a share ``hot`` of all reads copies molecule 0 (one key with n * hot copies);
#   * every ``lowc_every``-th molecule is low-complexity: the first half of its key is poly-A, so all of them share
#     segment 0 of a two-segment split (a crowded bucket of the neighbour search);
#   * a share ``ladder`` of the reads draws from a LADDER of 4^8 keys: one random prefix followed by every value of
#     the last eight bases -- 65 536 keys, each with 24 neighbours at Hamming distance 1, ONE connected component.
SKEW = {"hot": 0.02, "ladder": 0.01, "lowc_every": 100}
if __import__("os").environ.get("FQD_SKEW"):       # experiments: FQD_SKEW="hot=0.02,ladder=0,lowc_every=0"
    SKEW = {k: (int(v) if k == "lowc_every" else float(v))
            for k, v in (kv.split("=") for kv in __import__("os").environ["FQD_SKEW"].split(","))}


def synth_keys(n: int, length: int, umi: int, seed: int, *, copies: int = 4,
               sub_rate: float = 1e-3, n_rate: float = 1e-4, skew=None) -> np.ndarray:
    """Returns an ``(n, length)`` uint8 array of ASCII keys (reads ``0 .. n`` of
    an ``n``-read job). ``umi >= length`` makes every molecule fully random."""
    return _synth(n, length, umi, seed, copies, sub_rate, n_rate, 0, n, skew)


def synth_keys_range(n_total: int, start: int, count: int, length: int, umi: int, seed: int, *,
                     copies: int = 4, sub_rate: float = 1e-3, n_rate: float = 1e-4, skew=None) -> np.ndarray:
    """Reads ``start .. start+count`` of an ``n_total``-read job (one rank's shard)."""
    return _synth(count, length, umi, seed, copies, sub_rate, n_rate, start, n_total, skew)


def _synth(n, length, umi, seed, copies, sub_rate, n_rate, start, n_total, skew=None):
    if n == 0:
        return np.zeros((0, length), dtype=np.uint8)
    umi = min(umi, length)
    M = max(1, n_total // copies)
    F = max(1, M // 4)
    r = np.arange(start, start + n, dtype=np.uint64)
    lowc = ladder = None
    if skew:
        x = (stream_hash(seed, 0, r) >> np.uint64(11)).astype(np.float64) * (1.0 / float(1 << 53))
        mol = np.minimum((np.float64(M) * x * x).astype(np.uint64), np.uint64(M - 1))
        pick = stream_hash(seed, 8, r) >> np.uint64(11)
        thr_hot, thr_lad = rate_threshold(skew.get("hot", 0.0)), rate_threshold(skew.get("ladder", 0.0))
        mol = np.where(pick < np.uint64(thr_hot), np.uint64(0), mol)
        ladder = (pick >= np.uint64(thr_hot)) & (pick < np.uint64(thr_hot + thr_lad))
        rung = stream_hash(seed, 9, r) & np.uint64(0xFFFF)
        every = int(skew.get("lowc_every", 0))
        if every:
            lowc = (mol % np.uint64(every)) == np.uint64(every // 2)
    else:
        mol = stream_hash(seed, 0, r) % np.uint64(M)                  # (n,)
    ins = stream_hash(seed, 1, mol) % np.uint64(F)                    # (n,)
    out = np.empty((n, length), dtype=np.uint8)
    thr_n = np.uint64(rate_threshold(n_rate))
    thr_s = np.uint64(rate_threshold(n_rate) + rate_threshold(sub_rate))
    rest = length - umi
    for b in range(length):
        if b < umi:
            true = stream_hash(seed, 2, mol * np.uint64(umi) + np.uint64(b)) & np.uint64(3)
        else:
            true = stream_hash(seed, 3, ins * np.uint64(rest) + np.uint64(b - umi)) & np.uint64(3)
        if lowc is not None and b < length // 2:
            true = np.where(lowc, np.uint64(0), true)
        if ladder is not None:
            if b + 8 >= length:
                rung_base = (rung >> np.uint64(2 * (b + 8 - length))) & np.uint64(3)
            else:
                rung_base = np.full(n, stream_hash(seed, 7, np.array([b], dtype=np.uint64))[0] & np.uint64(3), dtype=np.uint64)
            true = np.where(ladder, rung_base, true)
        e = stream_hash(seed, 4, r * np.uint64(length) + np.uint64(b))
        u = e >> np.uint64(11)
        shift = np.uint64(1) + (e & np.uint64(0x7FF)) % np.uint64(3)
        sub = (true + shift) & np.uint64(3)
        code = np.where(u < thr_s, sub, true).astype(np.int64)
        col = BASES[code]
        col = np.where(u < thr_n, np.uint8(ord("N")), col)
        out[:, b] = col
    return out


def fixed_offsets(n: int, length: int) -> np.ndarray:
    return np.arange(n + 1, dtype=np.uint64) * np.uint64(length)


def indel_variant(keys: np.ndarray, seed: int, *, indel_rate: float = 0.01, start: int = 0):
    """The indel tail of SURVEY.md 8d (a share ``indel_rate`` of the reads is one base short or one
    base long): ``keys`` is the ``(n, L)`` array of ``synth_keys`` / ``synth_keys_range`` (reads
    ``start ..``); returns ``(bytes, offsets)`` of the ragged job. Device twin: ``fqd_synth_indel_keys``."""
    n, L = keys.shape
    r = np.arange(start, start + n, dtype=np.uint64)
    e = stream_hash(seed, 5, r)
    hit = (e >> np.uint64(11)) < np.uint64(rate_threshold(indel_rate))
    h = stream_hash(seed, 6, r)
    lens = np.full(n, L, dtype=np.int64)
    ins = hit & ((e & np.uint64(1)) == np.uint64(1))
    dele = hit & ~ins
    lens[ins] += 1
    lens[dele] -= 1
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens).astype(np.uint64)
    out = np.empty(int(offsets[-1]), dtype=np.uint8)
    plain = ~hit
    # unaffected reads: one vectorised copy
    idx = (offsets[:-1][plain].astype(np.int64)[:, None] + np.arange(L, dtype=np.int64)[None, :]).reshape(-1)
    out[idx] = keys[plain].reshape(-1)
    low = h & np.uint64(0xFFFFFFFF)
    for i in np.flatnonzero(hit):
        row = keys[i]
        o = int(offsets[i])
        if ins[i]:
            pos = int(low[i] % np.uint64(L + 1))
            base = BASES[int((h[i] >> np.uint64(32)) & np.uint64(3))]
            out[o:o + L + 1] = np.concatenate([row[:pos], [base], row[pos:]])
        else:
            pos = int(low[i] % np.uint64(L))
            out[o:o + L - 1] = np.concatenate([row[:pos], row[pos + 1:]])
    return out, offsets

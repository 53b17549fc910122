"""The ``fastqdedup`` command line on the MI355X path: same flags, defaults, log lines
and output naming as the reference (``src/fastqdedup/__init__.py:209-412``,
README.rst:42-89). What changes is where the work happens: the per-read loop of
``deduplicate_cluster`` (:242-276) becomes one quality-gate kernel launch plus one
``cluster_keys`` call, and pass 2 (:189-206) writes the records whose numbers the GPU
returned instead of re-hashing every key.
"""
from __future__ import annotations

import argparse
import datetime
import logging
import os
import resource
import time
from typing import Callable, Iterator, List, Optional, Tuple

import numpy as np

from . import fastq
from .core import (CLUSTER_DISSECTION_METHODS, DEFAULT_MAX_DISTANCE, TableCensus, cluster_dissection_directional,
                   cluster_keys, default_context)

DEFAULT_PREFIX = "fastqdedup_R"
DEFAULT_CLUSTER_DISSECTION = "directional"
DEFAULT_MAX_AVERAGE_ERROR_RATE = 0.001

ClusterDissectionFunc = Callable[[List[Tuple[int, str]], int, bool], Iterator[str]]

# reference score_to_error_rate.py: 10 ** -(i / 10) for i in 0..127
SCORE_TO_ERROR_RATE = np.array([10 ** -(i / 10) for i in range(128)], dtype=np.float64)


class Timer:
    """reference __init__.py:42-51"""

    def __init__(self):
        self.start_time = time.time()

    def get_difference(self) -> datetime.timedelta:
        now = time.time()
        delta = datetime.timedelta(seconds=round(now - self.start_time))
        self.start_time = now
        return delta


def average_error_rate(phred_scores: str, *, phred_offset: int = 33) -> float:
    """``fastqdedup._fastq.average_error_rate`` (_fastqmodule.c:38-76) on the device."""
    if not isinstance(phred_scores, str):
        raise TypeError(f"average_error_rate() argument 1 must be str, not {type(phred_scores).__name__}")
    if not phred_scores.isascii():
        raise ValueError("phred_scores must be ASCII encoded.")
    raw = np.frombuffer(phred_scores.encode("ascii"), dtype=np.uint8)
    hi = 126
    for ch in raw:
        if not (phred_offset <= ch <= hi):
            raise ValueError(f"Character {chr(ch)} outside of valid phred range "
                             f"('{chr(phred_offset)}' to '{chr(hi)}')")
    off = np.array([0, len(raw)], dtype=np.uint64)
    data = raw if len(raw) else np.zeros(1, dtype=np.uint8)
    _, means, _ = default_context().quality_filter(data, off, threshold=1.0, phred_offset=phred_offset,
                                                   want_means=True, table=SCORE_TO_ERROR_RATE)
    return float(means[0])


def trie_stats(trie) -> str:
    """The DEBUG-level census the reference prints after pass 1 (__init__.py:133-157): one row per
    trie layer -- leaves, then inner nodes by child-array width -- a totals row, and the node /
    suffix / total byte split. Same columns and number formats, so logs stay comparable."""
    stats = trie.raw_stats()
    cols = len(trie.alphabet) + 1                       # column 0 = leaves, column k = inner nodes of width k
    sums = [0] * (cols + 1)
    cell = "{:10}".format
    lines = ["layer     terminal  " + "".join(cell(k) for k in range(1, cols)) + "     total"]
    for depth, row in enumerate(stats):
        row_total = sum(row)
        for k in range(cols):
            sums[k] += row[k]
        sums[cols] += row_total
        lines.append("".join(cell(v) for v in [str(depth), *row, row_total]))
    lines.append("".join(cell(v) for v in ["total", *sums]))
    node_bytes = sum((8 + 8 * k) * sums[k] for k in range(cols))
    total_bytes = trie.memory_size()
    gib = float(1 << 30)
    lines.append(f"Node memory usage: {node_bytes / gib:.2} GiB")
    lines.append(f"Suffix memory usage: {(total_bytes - node_bytes) / gib:.2} GiB")
    lines.append(f"Total memory usage: {total_bytes / gib:.2} GiB")
    return "\n".join(lines) + "\n"


def length_string_to_slices(length_string: str) -> List[slice]:
    """'8,8,8' or '8:16,8,24:8:-1' -> slices (reference __init__.py:364-375)."""
    out = []
    for part in length_string.split(","):
        vals = [None if x in ("None", "") else int(x) for x in part.split(":")]
        out.append(slice(*vals))
    return out


def _method_name(func_or_name) -> str:
    if isinstance(func_or_name, str):
        return func_or_name
    for name, fn in CLUSTER_DISSECTION_METHODS.items():
        if fn is func_or_name:
            return name
    name = getattr(func_or_name, "__name__", "")
    for known in CLUSTER_DISSECTION_METHODS:
        if name.endswith(known):
            return known
    raise ValueError("unknown cluster dissection function")


def deduplicate_cluster(input_files: List[str], output_files: List[str], check_slices: Optional[List[slice]],
                        max_distance: int = DEFAULT_MAX_DISTANCE,
                        max_average_error_rate: float = DEFAULT_MAX_AVERAGE_ERROR_RATE,
                        cluster_dissection_func=cluster_dissection_directional,
                        use_edit_distance: bool = False):
    """Same signature as the reference (__init__.py:209-217)."""
    if len(input_files) != len(output_files):
        raise ValueError(f"Amount of output files ({len(output_files)}) "
                         f"must be equal to the amount of input files "
                         f"({len(input_files)}). ")
    if check_slices and len(input_files) != len(check_slices):
        raise ValueError(f"Amount of check lengths ({len(check_slices)}) "
                         f"must be equal to the amount of input files "
                         f"({len(input_files)}). ")
    method = _method_name(cluster_dissection_func)
    logger = logging.getLogger("fastqdedup")
    timer = Timer()
    ctx = default_context()
    slices = check_slices if check_slices else None
    filter_on_quality = max_average_error_rate < 1.0
    chunk_records = int(os.environ.get("FQD_FASTQ_CHUNK_RECORDS", 2_000_000))

    # ---- pass 1 (reference :242-252): the files are streamed in chunks of records; what stays in host
    # memory is the KEYS (and one weight per record), never a whole file
    # The keys of all chunks go into ONE buffer that grows geometrically (joining per-chunk arrays at the end held
    # every key twice for a moment: ~30 GB for 50 M reads of 300 nt); offsets are kept only from the chunk on that
    # proves the keys ragged (fixed-length keys need none: 8 bytes per read saved).
    class _Grow:
        def __init__(self, dtype):
            self.a, self.n = np.empty(1 << 20, dtype=dtype), 0

        def add(self, part):
            need = self.n + len(part)
            if need > len(self.a):
                bigger = np.empty(max(need, len(self.a) * 2), dtype=self.a.dtype)
                bigger[: self.n] = self.a[: self.n]
                self.a = bigger
            self.a[self.n:need] = part
            self.n = need

        def view(self):
            return self.a[: self.n]

    key_buf, weight_buf, off_buf = _Grow(np.uint8), _Grow(np.uint32), None
    n, discarded, key_len, key_bytes = 0, 0, None, 0
    for tables, m, _first in fastq.zip_chunks(input_files, chunk_records):
        fastq.check_mates(tables, m)
        keys, key_off, fixed = fastq.build_strings(tables, slices, m, "sequence")
        if filter_on_quality:
            quals, qual_off, qual_len = fastq.build_strings(tables, slices, m, "qualities")
            data = quals if len(quals) else np.zeros(1, dtype=np.uint8)
            w, _, d = ctx.quality_filter(data, None if qual_len else qual_off, qual_len,
                                         threshold=max_average_error_rate, table=SCORE_TO_ERROR_RATE)
            weight_buf.add(np.asarray(w, dtype=np.uint32))
            discarded += d
        was_fixed = key_len
        key_len = fixed if key_len is None else (key_len if key_len == fixed else 0)
        if not key_len:
            if off_buf is None:
                # ragged from here on: the offsets of the fixed-length chunks before this one are arithmetic
                off_buf = _Grow(np.uint64)
                off_buf.add(np.zeros(1, dtype=np.uint64))
                if n and was_fixed:
                    off_buf.add(np.arange(1, n + 1, dtype=np.uint64) * np.uint64(was_fixed))
            off_buf.add(key_off[1:].astype(np.uint64) + np.uint64(key_bytes))
        key_buf.add(keys)
        key_bytes += len(keys)
        n += m
    if filter_on_quality:
        logger.info(f"{discarded} records out of {n} "
                    f"records had an error rate higher than {max_average_error_rate} "
                    f"and were discarded.")

    res = None
    if n:
        keys = key_buf.view()
        key_off = None if key_len else off_buf.view()
        weights = weight_buf.view() if filter_on_quality else None
        res = cluster_keys(keys, key_off, key_len or 0, weights, max_distance=max_distance,
                           use_edit_distance=use_edit_distance, method=method, context=ctx)
        del keys, key_off, weights
    del key_buf, weight_buf, off_buf
    n_counted = res.n_counted if res else 0
    logger.info(f"Processed {n_counted} sequences. ({timer.get_difference()})")
    if logger.level <= logging.DEBUG and res is not None and res.n_unique:
        # the node census of the trie the reference would hold at this point (__init__.py:260-264),
        # from the unique table the context still holds
        stats = trie_stats(TableCensus(ctx))
        logger.debug(f"Calculated stats. ({timer.get_difference()})")
        logger.debug("\n" + stats)
    n_kept = res.n_kept if res else 0
    n_clusters = res.n_clusters if res else 0
    logger.info(f"Found {n_kept} distinct reads in {n_clusters} clusters."
                f"({timer.get_difference()})")

    # ---- pass 2 (reference :189-206): the files are streamed again; the records whose numbers the
    # GPU returned are written (the first holder of every kept key, filtered or not)
    keep = res.kept_read_ids.astype(np.int64) if res else np.zeros(0, dtype=np.int64)
    outs = [fastq.open_write(path) for path in output_files]
    try:
        for tables, m, first in fastq.zip_chunks(input_files, chunk_records):
            lo, hi = np.searchsorted(keep, [first, first + m])
            local = keep[lo:hi] - first
            for table, out in zip(tables, outs):
                fastq.write_records_to(table, local, out)
    finally:
        for out in outs:
            out.close()
    logger.info(f"Filtered FASTQ files based on distinct reads from each cluster. "
                f"({timer.get_difference()}) ")


def initiate_logger(verbose: int = 0, quiet: int = 0):
    """reference __init__.py:291-302"""
    level = logging.INFO - 10 * (verbose - quiet)
    logger = logging.getLogger("fastqdedup")
    logger.setLevel(level)
    handler = logging.StreamHandler()
    handler.setLevel(level)
    handler.setFormatter(logging.Formatter("{asctime}:{levelname}:{name}: {message}",
                                           datefmt="%m/%d/%Y %I:%M:%S", style="{"))
    logger.addHandler(handler)


def argument_parser() -> argparse.ArgumentParser:
    """The reference's flag surface (__init__.py:305-361, README.rst:42-89): same flags, dests,
    types and defaults; the help texts are this package's own."""
    p = argparse.ArgumentParser(
        prog="fastqdedup",
        description="Remove duplicate reads from FASTQ files without alignment: read keys are clustered by "
                    "Hamming or Levenshtein distance on an MI355X and one or more representatives per "
                    "cluster are written out.")
    p.add_argument("fastq", metavar="FASTQ", nargs="+",
                   help="Input FASTQ file(s): R1, then optionally R2 and/or a UMI file; records are taken "
                        "in lockstep and their sequences concatenated into one key.")
    p.add_argument("-l", "--check-lengths",
                   help="One entry per input file, separated by commas, limiting which bases of that file "
                        "enter the key: a plain number N means the first N bases, and Python slice syntax "
                        "(start:stop[:step], e.g. '4:8' or '::8') is accepted. Example: '--check-lengths "
                        "16,8' keys on R1[:16] + R2[:8].")
    p.add_argument("-o", "--output", action="append", required=False,
                   help="Where to write a deduplicated file; repeat the flag once per input file, in the "
                        "same order (e.g. '-o out_R1.fastq.gz -o out_R2.fastq.gz'). Without it the names "
                        "are derived from --prefix.")
    p.add_argument("-p", "--prefix", default=DEFAULT_PREFIX,
                   help=f"Stem of the automatic output names <prefix><file number>.fastq.gz "
                        f"(default '{DEFAULT_PREFIX}').")
    p.add_argument("-d", "--max-distance", type=int, default=DEFAULT_MAX_DISTANCE,
                   help=f"Keys at most this far apart end up in the same cluster (default {DEFAULT_MAX_DISTANCE}).")
    p.add_argument("-e", "--max-average-error-rate", type=float, default=DEFAULT_MAX_AVERAGE_ERROR_RATE,
                   help="Records whose mean per-base error probability (from the phred qualities of the "
                        "bases selected by --check-lengths) exceeds this value are not counted "
                        f"(default {DEFAULT_MAX_AVERAGE_ERROR_RATE}).")
    p.add_argument("-E", "--no-average-error-rate-filter", action="store_const",
                   dest="max_average_error_rate", const=1.0,
                   help="Switch the quality filter off.")
    p.add_argument("--edit", action="store_true",
                   help="Measure distance as Levenshtein (substitutions, insertions, deletions) rather than "
                        "Hamming (substitutions only).")
    p.add_argument("-c", "--cluster-dissection-method", choices=CLUSTER_DISSECTION_METHODS.keys(),
                   default=DEFAULT_CLUSTER_DISSECTION,
                   help="Which members of a cluster survive. highest_count: only the most frequent key. "
                        "adjacency: the most frequent key, then again the most frequent of those not within "
                        "the distance of a survivor, and so on. directional (default): like adjacency, but "
                        "a key is only absorbed by a neighbour that is at least about twice as frequent, "
                        "so genuine variants with comparable support are both kept.")
    p.add_argument("-v", "--verbose", action="count", default=0, help="More log output (repeatable).")
    p.add_argument("-q", "--quiet", action="count", default=0, help="Less log output (repeatable).")
    return p


def main(argv=None):
    args = argument_parser().parse_args(argv)
    initiate_logger(args.verbose, args.quiet)
    logger = logging.getLogger("fastqdedup")
    input_files: List[str] = args.fastq
    check_slices = length_string_to_slices(args.check_lengths) if args.check_lengths else None
    if args.output:
        output_files = args.output
    else:
        output_files = [args.prefix + str(x) + ".fastq.gz" for x in range(1, len(input_files) + 1)]
    distance_name = "Levenshtein" if args.edit else "Hamming"
    timer = Timer()
    logger.info(f"Input files: {', '.join(input_files)}")
    logger.info(f"Output files: {', '.join(output_files)}")
    logger.info(f"Check lengths: {args.check_lengths}")
    logger.info(f"Maximum {distance_name} distance: {args.max_distance}")
    logger.info(f"Maximum average error rate: {args.max_average_error_rate}")
    logger.info(f"Cluster dissection method: {args.cluster_dissection_method}")
    deduplicate_cluster(input_files, output_files, check_slices, args.max_distance,
                        args.max_average_error_rate,
                        CLUSTER_DISSECTION_METHODS[args.cluster_dissection_method], args.edit)
    usage = resource.getrusage(resource.RUSAGE_SELF)
    logger.info(f"Finished. Total time: {timer.get_difference()}. "
                f"Memory usage: {usage.ru_maxrss / (1024 ** 2):.2} GiB")


if __name__ == "__main__":
    main()

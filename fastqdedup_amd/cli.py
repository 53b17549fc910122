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
import resource
import time
from typing import Callable, Iterator, List, Optional, Tuple

import numpy as np

from . import fastq
from .core import (CLUSTER_DISSECTION_METHODS, DEFAULT_MAX_DISTANCE, cluster_dissection_directional,
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

    tables, n = fastq.read_all(input_files)
    fastq.check_mates(tables, n)
    slices = check_slices if check_slices else None
    keys, key_off, key_len = fastq.build_strings(tables, slices, n, "sequence")

    filter_on_quality = max_average_error_rate < 1.0
    weights = None
    discarded = 0
    if filter_on_quality and n:
        quals, qual_off, qual_len = fastq.build_strings(tables, slices, n, "qualities")
        data = quals if len(quals) else np.zeros(1, dtype=np.uint8)
        weights, _, discarded = ctx.quality_filter(data, None if qual_len else qual_off, qual_len,
                                                   threshold=max_average_error_rate,
                                                   table=SCORE_TO_ERROR_RATE)
        del quals, qual_off
        logger.info(f"{discarded} records out of {n} "
                    f"records had an error rate higher than {max_average_error_rate} "
                    f"and were discarded.")
    elif filter_on_quality:
        logger.info(f"0 records out of 0 records had an error rate higher than "
                    f"{max_average_error_rate} and were discarded.")

    res = None
    if n:
        res = cluster_keys(keys, None if key_len else key_off, key_len, weights, max_distance=max_distance,
                           use_edit_distance=use_edit_distance, method=method, context=ctx)
    n_counted = res.n_counted if res else 0
    logger.info(f"Processed {n_counted} sequences. ({timer.get_difference()})")
    n_kept = res.n_kept if res else 0
    n_clusters = res.n_clusters if res else 0
    logger.info(f"Found {n_kept} distinct reads in {n_clusters} clusters."
                f"({timer.get_difference()})")

    keep = res.kept_read_ids.astype(np.int64) if res else np.zeros(0, dtype=np.int64)
    for table, path in zip(tables, output_files):
        fastq.write_records(table, keep, path)
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
    """Flag for flag the reference's parser (__init__.py:305-361, README.rst:42-89)."""
    p = argparse.ArgumentParser(prog="fastqdedup")
    p.add_argument("fastq", metavar="FASTQ", nargs="+",
                   help="Forward FASTQ and optional reverse and UMI FASTQ files.")
    p.add_argument("-l", "--check-lengths",
                   help="Comma-separated string with the maximum string check length of each file. "
                        "For example 'fastqdedup --check-lengths 16,8 R1.fastq R2.fastq' only checks the "
                        "first 16 bases of R1 and the first 8 bases of R2 for duplication. Supports slice "
                        "notation such as '4:8' or '::8'.")
    p.add_argument("-o", "--output", action="append", required=False,
                   help="Output file (optional), must be specified multiple times for multiple input "
                        "files. For example ``fastqdedup -o dedupR1.fastq -o dedupR2.fastq R1.fastq "
                        "R2.fastq``.")
    p.add_argument("-p", "--prefix", default=DEFAULT_PREFIX,
                   help=f"Prefix for the output files. Default: '{DEFAULT_PREFIX}'")
    p.add_argument("-d", "--max-distance", type=int, default=DEFAULT_MAX_DISTANCE,
                   help="The Hamming distance at which inputs are considered different. "
                        f"Default: {DEFAULT_MAX_DISTANCE}.")
    p.add_argument("-e", "--max-average-error-rate", type=float, default=DEFAULT_MAX_AVERAGE_ERROR_RATE,
                   help="The maximum average per base error rate for each FASTQ record. Average is "
                        "evaluated over bases taken into account by --check-lengths."
                        f"Default: {DEFAULT_MAX_AVERAGE_ERROR_RATE}")
    p.add_argument("-E", "--no-average-error-rate-filter", action="store_const",
                   dest="max_average_error_rate", const=1.0,
                   help="Do not filter on average per base error rate.")
    p.add_argument("--edit", action="store_true",
                   help="Use edit (Levenshtein) distance instead of Hamming distance.")
    p.add_argument("-c", "--cluster-dissection-method", choices=CLUSTER_DISSECTION_METHODS.keys(),
                   default=DEFAULT_CLUSTER_DISSECTION,
                   help="How to approach clusters with multiple reads. 'highest_count' selects only one "
                        "read, the one with the highest count. 'adjacency' starts from the read with the "
                        "highest count and selects all reads that are within the specified distance. The "
                        "process is repeated for the remaining reads. 'directional' is similar to "
                        "adjacency but uses counts to determine if an error is a PCR/sequencing artifact "
                        "or derived from a difference in the molecule (default).")
    p.add_argument("-v", "--verbose", action="count", default=0, help="Increase log verbosity.")
    p.add_argument("-q", "--quiet", action="count", default=0, help="Reduce log verbosity.")
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

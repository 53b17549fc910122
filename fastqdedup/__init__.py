"""``fastqdedup`` -- the reference's package name, served by the MI355X implementation.

A user of rhpvorderman/fastqdedup imports ``fastqdedup`` and runs the ``fastqdedup`` console
script (reference setup.py:57-58, __main__.py:17-20). This shim gives both names to
``fastqdedup_amd``: the same module-level API (reference src/fastqdedup/__init__.py) and the same
private extension-module names (``fastqdedup._trie.Trie``, ``fastqdedup._distance.within_distance``,
``fastqdedup._fastq.average_error_rate``; setup.py:52-56), every one of them running on the GPU
through ``libfqdedup_hip.so``. Nothing is implemented here.
"""
from fastqdedup_amd import (CLUSTER_DISSECTION_METHODS, DEFAULT_CLUSTER_DISSECTION,  # noqa: F401
                            DEFAULT_MAX_AVERAGE_ERROR_RATE, DEFAULT_MAX_DISTANCE, DEFAULT_PREFIX, Trie,
                            argument_parser, cluster_dissection_adjacency, cluster_dissection_directional,
                            cluster_dissection_highest_count, deduplicate_cluster, length_string_to_slices,
                            main, within_distance)
from fastqdedup_amd import average_error_rate as fastq_average_error_rate  # noqa: F401
from fastqdedup_amd.cli import Timer, initiate_logger, trie_stats  # noqa: F401

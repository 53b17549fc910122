"""fastqdedup_amd -- the clustering hot path of fastqdedup on MI355X (gfx950).

Drop-in for the reference's ``Trie`` / ``within_distance`` /
``cluster_dissection_*`` surface (reference src/fastqdedup/__init__.py:32-34,
60-130), implemented as hand-written HIP kernels behind a C ABI
(include/fqdedup_hip.h). No CPU fallback: without ``libfqdedup_hip.so`` and a
gfx950 device every entry point raises.
"""
from .core import (CLUSTER_DISSECTION_METHODS, DEFAULT_MAX_DISTANCE, ClusterResult, Trie,
                   cluster_dissection_adjacency, cluster_dissection_directional,
                   cluster_dissection_highest_count, cluster_keys, default_context,
                   pack_strings, within_distance)
from ._lib import Context
from .cli import (DEFAULT_CLUSTER_DISSECTION, DEFAULT_MAX_AVERAGE_ERROR_RATE, DEFAULT_PREFIX, argument_parser,
                  average_error_rate, deduplicate_cluster, length_string_to_slices, main)

__all__ = [
    "CLUSTER_DISSECTION_METHODS", "ClusterResult", "Context", "DEFAULT_MAX_DISTANCE", "Trie",
    "cluster_dissection_adjacency", "cluster_dissection_directional",
    "cluster_dissection_highest_count", "cluster_keys", "default_context", "pack_strings",
    "within_distance", "argument_parser", "average_error_rate", "deduplicate_cluster",
    "length_string_to_slices", "main",
]

"""CPU ORACLE -- TEST INFRASTRUCTURE. Drives the REAL reference extensions
(oracle/_ref, compiled from /root/reference where it lies) the way
``deduplicate_cluster`` does (reference __init__.py:240-276): ``Trie("ACGTN")``,
one ``add_sequence`` per read, ``pop_cluster`` until empty, dissect every cluster.

The reference's three dissection functions are Python code inside its
``__init__.py`` (which cannot be imported: dnaio/xopen are absent, and the file
does not exist on the GPU box). They are restated below in Python with the same
loops, calling the reference's own C ``within_distance``, so the baseline pays the
same interpreter costs the reference pays. tests/test_oracle_ref_driver.py checks
them against the golden vectors minted from the reference's own functions.

Only bench.py's cpu_baseline leg and tests/ may import this.
"""
from __future__ import annotations

import time
from typing import Dict, Iterable, Iterator, List


def make_dissectors(within_distance):
    def directional(cluster, max_distance=1, use_edit_distance=False) -> Iterator[str]:
        """__init__.py:60-91"""
        pool = sorted(cluster)
        while pool:
            root = pool.pop()
            chain = [root]
            for t_count, t_key in chain:
                if not pool:
                    break
                rest = []
                for item in pool:
                    c, k = item
                    if 2 * c - 1 <= t_count and within_distance(t_key, k, max_distance, use_edit_distance):
                        chain.append(item)
                    else:
                        rest.append(item)
                pool = rest
            yield root[1]

    def highest_count(cluster, max_distance=1, use_edit_distance=False) -> Iterator[str]:
        """__init__.py:94-102"""
        yield sorted(cluster, reverse=True)[0][1]

    def adjacency(cluster, max_distance=1, use_edit_distance=False) -> Iterator[str]:
        """__init__.py:105-122"""
        pool = sorted(cluster, reverse=True)
        while pool:
            root = pool[0][1]
            pool = [it for it in pool[1:]
                    if not within_distance(root, it[1], max_distance, use_edit_distance)]
            yield root

    return {"directional": directional, "highest_count": highest_count, "adjacency": adjacency}


def run_reference_path(keys: Iterable[str], max_distance: int = 1, use_edit_distance: bool = False,
                       method: str = "directional") -> Dict:
    """Returns kept keys, counters and the stage split of the reference CPU path."""
    from . import oracle as O
    ref_trie, ref_dist = O.load_reference()
    dissect = make_dissectors(ref_dist.within_distance)[method]
    t0 = time.perf_counter()
    trie = ref_trie.Trie(alphabet="ACGTN")
    add = trie.add_sequence
    n = 0
    for k in keys:
        add(k)
        n += 1
    t1 = time.perf_counter()
    kept: List[str] = []
    clusters = 0
    unique = 0
    t_pop = t_dis = 0.0
    while trie.number_of_sequences:
        a = time.perf_counter()
        cl = trie.pop_cluster(max_distance, use_edit_distance)
        b = time.perf_counter()
        kept.extend(dissect(cl, max_distance, use_edit_distance))
        c = time.perf_counter()
        t_pop += b - a
        t_dis += c - b
        clusters += 1
        unique += len(cl)
    return {"n": n, "kept_keys": kept, "n_clusters": clusters, "n_unique": unique,
            "seconds": {"insert": t1 - t0, "pop_cluster": t_pop, "dissect": t_dis,
                        "total": (t1 - t0) + t_pop + t_dis}}

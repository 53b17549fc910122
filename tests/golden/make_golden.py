#!/usr/bin/env python3
"""Mint golden vectors with the REFERENCE ITSELF. Run in the build container only.

    python3 tests/golden/make_golden.py

What runs here:
  * the reference's C extensions, compiled from /root/reference where they lie
    into oracle/_ref (``make -C oracle ref``): ``Trie`` and ``within_distance``;
  * the reference's three ``cluster_dissection_*`` functions, compiled in this
    process from /root/reference/src/fastqdedup/__init__.py (the module itself
    cannot be imported: it imports dnaio/xopen, which this image lacks and which
    the hot path does not use). Only those function definitions are executed,
    with ``within_distance`` bound to the reference's own C implementation.
    Nothing of the reference's text is written anywhere.

What is written: DATA only -- input keys and the reference's outputs -- as
``tests/golden/ref_vectors.json.gz`` (short keys), ``ref_vectors_long.json.gz``
(300-nt keys with N at d <= 2; 149/150/151- and 299/300/301-nt mixed-length sets
for the edit metric) and ``fastq_error_rates.json`` (the reference's
``_fastq.average_error_rate`` on every valid phred character and on random
strings, as exact hex floats). ``/root/reference`` does not exist on the GPU
box; the committed fixtures are what travels.
"""
from __future__ import annotations

import ast
import gzip
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

REF_INIT = "/root/reference/src/fastqdedup/__init__.py"
WANTED = ("cluster_dissection_directional", "cluster_dissection_highest_count",
          "cluster_dissection_adjacency")


def load_reference_dissection(within_distance):
    """Compile the reference's dissection functions from its own source file."""
    with open(REF_INIT) as fh:
        tree = ast.parse(fh.read(), REF_INIT)
    keep = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in WANTED:
            keep.append(node)
        elif isinstance(node, ast.Assign) and any(
                isinstance(t, ast.Name) and t.id.startswith("DEFAULT_") for t in node.targets):
            keep.append(node)
    mod = ast.Module(body=keep, type_ignores=[])
    from typing import Iterator, List, Tuple
    ns = {"within_distance": within_distance, "List": List, "Tuple": Tuple, "Iterator": Iterator}
    exec(compile(mod, REF_INIT, "exec"), ns)
    return {"directional": ns["cluster_dissection_directional"],
            "highest_count": ns["cluster_dissection_highest_count"],
            "adjacency": ns["cluster_dissection_adjacency"]}


def run_reference(ref_trie, dissect, keys, weights, d, edit):
    """Drive the reference exactly as deduplicate_cluster does (__init__.py:240-276)."""
    trie = ref_trie.Trie(alphabet="ACGTN")
    for k, w in zip(keys, weights):
        for _ in range(w):
            trie.add_sequence(k)
    clusters = []
    while trie.number_of_sequences:
        clusters.append(trie.pop_cluster(d, edit))
    kept = {}
    for name, fn in dissect.items():
        out = []
        for cl in clusters:
            out.extend(fn(cl, d, edit))
        kept[name] = out
    return clusters, kept


def mutate(rng, s, alphabet, sub, indel):
    out = []
    for ch in s:
        x = rng.random()
        if x < sub:
            out.append(rng.choice([c for c in alphabet if c != ch] or [ch]))
        elif x < sub + indel / 2:
            continue
        elif x < sub + indel:
            out.append(ch)
            out.append(rng.choice(alphabet))
        else:
            out.append(ch)
    return "".join(out)


def make_inputs():
    """name -> (keys, weights). Small enough that the JSON stays small."""
    from fastqdedup_amd.synth import synth_keys
    cases = {}
    for seed, n, L, umi, sub, nr in [(11, 3000, 20, 6, 0.01, 0.004), (12, 3000, 33, 8, 0.006, 0.002),
                                     (13, 2000, 64, 12, 0.004, 0.001), (14, 1500, 100, 12, 0.003, 0.001)]:
        k = synth_keys(n, L, umi, seed, sub_rate=sub, n_rate=nr)
        cases[f"synth_s{seed}_n{n}_L{L}"] = ([bytes(r).decode() for r in k], [1] * n)
    rng = random.Random(7)
    # mixed lengths + indels (exercises the edit metric across length classes)
    mols = ["".join(rng.choice("ACGT") for _ in range(rng.randint(10, 14))) for _ in range(250)]
    keys = [mutate(rng, rng.choice(mols), "ACGT", 0.02, 0.02) for _ in range(1500)]
    cases["mixed_len_indel"] = (keys, [1] * len(keys))
    # low-complexity: giant components, ties at equal count, N/T/G ordering
    keys = ["".join(rng.choice("ACGTN" if rng.random() < 0.3 else "ACGT") for _ in range(6))
            for _ in range(2500)]
    cases["giant_L6"] = (keys, [1] * len(keys))
    keys = ["".join(rng.choice("ACGT") for _ in range(8)) for _ in range(6000)]
    cases["giant_L8"] = (keys, [1] * len(keys))
    # prefix keys, empty key, quality-failed holders (weight 0), heavy counts
    keys = ["", "A", "AC", "ACG", "ACGT", "ACGTA", "ACGTN", "ACGTT", "ACGTG", "T", "TT", "TTT",
            "TTTT", "TTTN", "NTTT", "ATTT", "TTTTA", "GGGGGGGG", "GGGGGGGT", "GGGGGGTT", "AAAAAAAA",
            "AAAAAAAC", "AAAAAAAA", "TTTTTTTT", "TTTTTTTN"]
    weights = [1, 2, 1, 3, 1, 1, 1, 1, 1, 5, 1, 1, 1, 1, 1, 1, 1, 0, 4, 1, 0, 1, 7, 1, 1]
    cases["prefix_weights"] = (keys, weights)
    # the reference's own dissection fixture as a trie input (tests/test_fastqdedup.py:37-45)
    keys = ["AAAGT", "AAAAT", "AACAA", "AAAAA", "CAAAA", "CTAAA"]
    cases["ref_test_cluster"] = (keys, [3, 10, 50, 60, 10, 30])
    return cases


def indel_reads(rng, n_mol, length, n_reads, sub, p_del, p_ins, alphabet="ACGT"):
    """Reads of `length`-nt molecules: substitutions at rate `sub`; a share p_del loses one base,
    a share p_ins gains one (the 149/151-nt sub-population next to 150-nt reads)."""
    mols = ["".join(rng.choice(alphabet) for _ in range(length)) for _ in range(n_mol)]
    out = []
    for _ in range(n_reads):
        s = list(rng.choice(mols))
        for i in range(len(s)):
            if rng.random() < sub:
                s[i] = rng.choice([c for c in alphabet + "N" if c != s[i]])
        x = rng.random()
        if x < p_del:
            del s[rng.randrange(len(s))]
        elif x < p_del + p_ins:
            s.insert(rng.randrange(len(s) + 1), rng.choice(alphabet))
        out.append("".join(s))
    return out


def make_long_inputs():
    """name -> (keys, weights, [(edit, d), ...]): the shapes of configs 4 and 5."""
    from fastqdedup_amd.synth import synth_keys
    cases = {}
    k = synth_keys(900, 300, 300, 15, sub_rate=2.5e-3, n_rate=1.5e-3)
    cases["synth_s15_n900_L300_N"] = ([bytes(r).decode() for r in k], [1] * 900,
                                      [(False, 0), (False, 1), (False, 2), (True, 1), (True, 2)])
    rng = random.Random(150)
    keys = indel_reads(rng, 260, 150, 1300, 2e-3, 0.08, 0.08)
    cases["mixed_149_150_151"] = (keys, [1] * len(keys), [(False, 1), (True, 1), (True, 2)])
    rng = random.Random(300)
    keys = indel_reads(rng, 200, 300, 1000, 1.5e-3, 0.06, 0.06)
    cases["mixed_299_300_301"] = (keys, [1] * len(keys), [(False, 2), (True, 1), (True, 2)])
    return cases


def make_fastq_fixture():
    """The reference's _fastq.average_error_rate (_fastqmodule.c:38-76) on every valid phred
    character, on random strings and with other offsets; values as exact hex floats."""
    import importlib.util
    d = os.path.join(ROOT, "oracle", "_ref")
    so = [f for f in os.listdir(d) if f.startswith("_fastq.")][0]
    spec = importlib.util.spec_from_file_location("_fastq", os.path.join(d, so))
    fq = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fq)
    rng = random.Random(94)
    doc = {"generator": "tests/golden/make_golden.py", "single": {}, "strings": [], "errors": []}
    for c in range(33, 127):
        doc["single"][chr(c)] = fq.average_error_rate(chr(c)).hex()
    for _ in range(120):
        n = rng.choice([1, 2, 3, 7, 16, 50, 100, 151, 300])
        lo, hi = rng.choice([(33, 126), (33, 75), (60, 75), (35, 45)])
        s = "".join(chr(rng.randint(lo, hi)) for _ in range(n))
        doc["strings"].append({"phred": s, "offset": 33, "value": fq.average_error_rate(s).hex()})
    for off in (0, 64):
        for _ in range(20):
            n = rng.randint(1, 60)
            s = "".join(chr(rng.randint(off, min(126, off + 62))) for _ in range(n))
            doc["strings"].append({"phred": s, "offset": off,
                                   "value": fq.average_error_rate(s, phred_offset=off).hex()})
    doc["strings"].append({"phred": "", "offset": 33, "value": repr(fq.average_error_rate(""))})   # nan
    for bad, off in [(" ", 33), ("I\x7f", 33), ("?", 64), ("\x1f", 0 + 33)]:
        try:
            fq.average_error_rate(bad, phred_offset=off)
            raise AssertionError("reference accepted " + repr(bad))
        except ValueError as exc:
            doc["errors"].append({"phred": bad, "offset": off, "message": str(exc)})
    path = os.path.join(HERE, "fastq_error_rates.json")
    with open(path, "w") as fh:
        json.dump(doc, fh, indent=0, sort_keys=True)
        fh.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes", file=sys.stderr)


def _run_record(clusters, kept):
    uniq = sorted({k for cl in clusters for _, k in cl})
    index = {k: i for i, k in enumerate(uniq)}
    return {"n_clusters": len(clusters), "labels": _labels(clusters, index), "counts": _counts(clusters, index),
            "kept": {m: sorted(index[k] for k in v) for m, v in kept.items()},
            "pop_order_seed": [index[cl[0][1]] for cl in clusters]}


def main():
    from oracle import oracle as O
    O.build()
    ref_trie, ref_dist = O.load_reference()
    dissect = load_reference_dissection(ref_dist.within_distance)
    make_fastq_fixture()
    long_out = {"generator": "tests/golden/make_golden.py", "cases": {}}
    for name, (keys, weights, runs_wanted) in make_long_inputs().items():
        runs = {}
        for edit, d in runs_wanted:
            clusters, kept = run_reference(ref_trie, dissect, keys, weights, d, edit)
            runs[f"{'L' if edit else 'H'}{d}"] = _run_record(clusters, kept)
        long_out["cases"][name] = {"keys": keys, "weights": weights, "runs": runs}
        print(name, len(keys), {k: (v["n_clusters"], len(v["kept"]["directional"]))
                                for k, v in runs.items()}, file=sys.stderr)
    path = os.path.join(HERE, "ref_vectors_long.json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as fh:
        fh.write(json.dumps(long_out, separators=(",", ":")).encode())
    print("wrote", path, os.path.getsize(path), "bytes", file=sys.stderr)
    out = {"generator": "tests/golden/make_golden.py", "cases": {}}
    for name, (keys, weights) in make_inputs().items():
        runs = {}
        ds = (0, 1, 2) if len(keys) <= 3000 else (0, 1)
        for edit in (False, True):
            for d in ds:
                if edit and d == 2 and name.startswith("giant"):
                    continue  # minutes of reference time for no extra coverage
                clusters, kept = run_reference(ref_trie, dissect, keys, weights, d, edit)
                uniq = sorted({k for cl in clusters for _, k in cl})
                index = {k: i for i, k in enumerate(uniq)}
                runs[f"{'L' if edit else 'H'}{d}"] = {
                    "n_clusters": len(clusters),
                    # cluster id per unique key (unique keys in sorted order)
                    "labels": _labels(clusters, index),
                    "counts": _counts(clusters, index),
                    "kept": {m: sorted(index[k] for k in v) for m, v in kept.items()},
                    # order in which the reference pops clusters: min index of each cluster's seed
                    "pop_order_seed": [index[cl[0][1]] for cl in clusters],
                }
        out["cases"][name] = {"keys": keys, "weights": weights, "runs": runs}
        print(name, len(keys), {k: (v["n_clusters"], len(v["kept"]["directional"]))
                                for k, v in runs.items()}, file=sys.stderr)
    path = os.path.join(HERE, "ref_vectors.json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as fh:
        fh.write(json.dumps(out, separators=(",", ":")).encode())
    print("wrote", path, os.path.getsize(path), "bytes", file=sys.stderr)


def _labels(clusters, index):
    lab = [0] * len(index)
    for ci, cl in enumerate(clusters):
        for _, k in cl:
            lab[index[k]] = ci
    return lab


def _counts(clusters, index):
    cnt = [0] * len(index)
    for cl in clusters:
        for c, k in cl:
            cnt[index[k]] = c
    return cnt


if __name__ == "__main__":
    main()

"""The Trie OBJECT on the device store (8f-3), its node census (a13) and the cluster iterator of the
C ABI, against the CPU oracle's trie -- itself equal to the reference's on 1 500 random tries
(tests/test_oracle_vs_reference_live.py) -- and the reference's known answers. GPU only."""
import random

import numpy as np
import pytest

import surface_checks as sc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    import fastqdedup_amd
    return fastqdedup_amd


def _rand(rng, syms, lo, hi):
    return "".join(rng.choice(syms) for _ in range(rng.randint(lo, hi)))


def test_known_answer_memory_size_and_raw_stats(F, known_answers):
    sc.check_trie_stats(F, known_answers, lazy_alphabet=True)


def test_alphabet_grows_like_the_references(F, oracle, known_answers):
    """reference tests/test_trie.py:139-158 (a symbol is registered when an inner node first looks
    it up) and a fuzz against the oracle's trie with alphabets that do NOT cover the keys: the
    alphabet after every few adds, the census, and the pop order that follows from it."""
    trie = F.Trie()
    for add, want in known_answers["trie_alphabet"]["growth"]:
        trie.add_sequence(add)
        assert trie.alphabet == want
    rng = random.Random(11)
    for trial in range(80):
        alpha = rng.choice(["", "", "AC", "TG", "N", "ACGTN"])
        syms = rng.choice(["ACGT", "ACGTN", "abcXN", "ab"])
        a, b = F.Trie(alpha), oracle.Trie(alpha)
        for i in range(rng.randint(1, 40)):
            s = _rand(rng, syms, 0, 7)
            a.add_sequence(s)
            b.add_sequence(s)
            if rng.random() < 0.3:
                assert a.alphabet == b.alphabet, (trial, i)
        assert a.alphabet == b.alphabet, trial
        assert a.memory_size() == b.memory_size(), trial
        assert a.raw_stats() == b.raw_stats(), trial
        d, edit = rng.randint(0, 2), rng.random() < 0.5
        while b.number_of_sequences:
            ca, cb = a.pop_cluster(d, edit), b.pop_cluster(d, edit)
            assert sorted(ca) == sorted(cb) and ca[0] == cb[0], trial
            assert a.memory_size() == b.memory_size()


def test_trie_census_fuzz_matches_oracle_trie(F, oracle):
    """As tests/test_oracle_vs_reference_live.py::test_trie_fuzz drives the reference: random adds,
    then memory_size / raw_stats / contains; then pops, with the census compared after EVERY pop
    (the oracle really deletes and prunes, _triemodule.c:301-363; the device marks rows removed)."""
    rng = random.Random(20261004)
    for trial in range(120):
        alpha, syms = rng.choice([("ACGTN", "ACGTN"), ("ACGTN", "ACGT"), ("AC", "AC"), ("TGCA", "ACGT"),
                                  ("ACGTN", "AC"), ("NTGCA", "ACGTN"), ("abcXN", "abcXN")])
        a, b = F.Trie(alpha), oracle.Trie(alpha)
        for _ in range(rng.randint(1, 40)):
            s = _rand(rng, syms, 0, 7)
            a.add_sequence(s)
            b.add_sequence(s)
        assert a.alphabet == b.alphabet
        assert a.memory_size() == b.memory_size(), trial
        assert a.raw_stats() == b.raw_stats(), trial
        d, edit = rng.randint(0, 3), rng.random() < 0.5
        for _ in range(4):
            q = _rand(rng, syms, 0, 7)
            assert a.contains_sequence(q, d, edit) == b.contains_sequence(q, d, edit)
        while b.number_of_sequences:
            ca, cb = a.pop_cluster(d, edit), b.pop_cluster(d, edit)
            assert sorted(ca) == sorted(cb) and ca[0] == cb[0], trial
            assert a.number_of_sequences == b.number_of_sequences
            assert a.memory_size() == b.memory_size(), (trial, "after pop")
            assert a.raw_stats() == b.raw_stats(), (trial, "after pop")
            q = _rand(rng, syms, 0, 7)
            assert a.contains_sequence(q, d, edit) == b.contains_sequence(q, d, edit)
        assert a.memory_size() == 0 and a.number_of_sequences == 0


def test_pops_with_changing_parameters_never_see_popped_keys(F, oracle):
    """pop_cluster(d) then pop_cluster(d') on the same object: the second clustering runs over the
    resident table, whose popped rows must link nothing (the reference deleted them from its trie,
    _triemodule.c:830-831,875-876). AAAA / AACC / CCAA: once AAAA is gone, d=2 leaves AACC and CCAA apart."""
    a, b = F.Trie("ACGTN"), oracle.Trie("ACGTN")
    for s in ("AAAA", "AACC", "CCAA"):
        a.add_sequence(s)
        b.add_sequence(s)
    assert sorted(a.pop_cluster(1)) == sorted(b.pop_cluster(1)) == [(1, "AAAA")]
    while b.number_of_sequences:
        ca, cb = a.pop_cluster(2), b.pop_cluster(2)
        assert sorted(ca) == sorted(cb) and ca[0] == cb[0]
    assert a.number_of_sequences == 0
    rng = random.Random(5)
    for trial in range(60):
        syms = rng.choice(["ACGT", "ACGTN", "AC"])
        a, b = F.Trie("ACGTN"), oracle.Trie("ACGTN")
        for _ in range(rng.randint(2, 50)):
            s = _rand(rng, syms, 3, 6)
            a.add_sequence(s)
            b.add_sequence(s)
        while b.number_of_sequences:
            d, edit = rng.randint(0, 3), rng.random() < 0.5     # new parameters at every pop
            ca, cb = a.pop_cluster(d, edit), b.pop_cluster(d, edit)
            assert sorted(ca) == sorted(cb) and ca[0] == cb[0], (trial, d, edit)
            assert a.number_of_sequences == b.number_of_sequences


def test_census_at_scale_matches_oracle_trie(F, oracle):
    """30 k reads of 40 nt (config-like synthetic keys with N): the census after pass 1 -- what the
    reference's DEBUG log prints (__init__.py:260-264) -- and after popping half of the clusters."""
    from fastqdedup_amd.synth import synth_keys
    keys = [bytes(r).decode() for r in synth_keys(30000, 40, 8, 77, sub_rate=4e-3, n_rate=2e-3)]
    a, b = F.Trie("ACGTN"), oracle.Trie("ACGTN")
    for k in keys:
        a.add_sequence(k)
        b.add_sequence(k)
    assert a.memory_size() == b.memory_size()
    assert a.raw_stats() == b.raw_stats()
    n = 0
    while b.number_of_sequences and n < 4000:
        ca, cb = a.pop_cluster(1), b.pop_cluster(1)
        assert sorted(ca) == sorted(cb) and ca[0] == cb[0]
        n += 1
    assert a.number_of_sequences == b.number_of_sequences
    assert a.memory_size() == b.memory_size()
    assert a.raw_stats() == b.raw_stats()


def test_trie_stats_log_table(F, known_answers):
    """The DEBUG table of the reference's -v run (__init__.py:133-157) from the device census."""
    from fastqdedup_amd.cli import trie_stats
    case = known_answers["trie_stats_known"]
    trie = F.Trie(case["alphabet"])
    for k in case["adds"]:
        trie.add_sequence(k)
    text = trie_stats(trie)
    lines = text.splitlines()
    assert lines[0].split() == ["layer", "terminal", "1", "2", "3", "4", "5", "total"]
    assert lines[1].split() == ["0", "0", "0", "0", "0", "0", "1", "1"]
    assert lines[7].split() == ["total", "4", "1", "0", "0", "2", "2", "9"]
    assert "Total memory usage" in lines[-1]


def test_store_grows_its_geometry(F, oracle):
    """Batches that bring new symbols, longer keys and other lengths: the resident records are
    re-encoded on the device (fqd_store_add_keys), nothing is packed twice."""
    rng = random.Random(5)
    a, b = F.Trie("ACGT"), oracle.Trie("ACGT")
    batches = [("ACGT", 12, 12), ("ACGT", 12, 12), ("ACGTN", 12, 12), ("ACGTNRY", 12, 12), ("ACGT", 5, 20),
               ("acgt", 33, 40), ("ACGT", 12, 12)]
    for syms, lo, hi in batches:
        pool = [_rand(rng, syms, lo, hi) for _ in range(30)]
        for _ in range(200):
            s = list(rng.choice(pool))
            if rng.random() < 0.3 and s:
                s[rng.randrange(len(s))] = rng.choice(syms)
            s = "".join(s)
            a.add_sequence(s)
            b.add_sequence(s)
        q = rng.choice(pool)
        assert a.contains_sequence(q, 1) == b.contains_sequence(q, 1)      # flushes the batch
        assert a.number_of_sequences == b.number_of_sequences
    assert a.alphabet == b.alphabet       # symbols registered as the reference's trie meets them
    for _ in range(40):
        ca, cb = a.pop_cluster(2, True), b.pop_cluster(2, True)
        assert sorted(ca) == sorted(cb)
    a.add_sequence("ACGTACGTACGT")                                          # an add between pops: merge
    b.add_sequence("ACGTACGTACGT")
    while b.number_of_sequences:
        ca, cb = a.pop_cluster(2, True), b.pop_cluster(2, True)
        assert sorted(ca) == sorted(cb)
    assert a.number_of_sequences == 0


def test_cluster_iterator_of_the_abi(F, oracle):
    """fqd_get_clusters / fqd_read_clusters alone reproduce the pop_cluster sequence: same clusters,
    same order, seed first (reference _triemodule.c:778-897, seeds :510-551)."""
    from fastqdedup_amd.synth import synth_keys
    n, L = 20000, 24
    keys = synth_keys(n, L, 6, 99, sub_rate=6e-3, n_rate=2e-3)
    strs = [bytes(r).decode() for r in keys]
    ctx = F.Context(0)
    ctx.pack_keys(keys.reshape(-1), None, L)
    nu = ctx.collapse()
    ctx.find_edges(1)
    nc = ctx.components()
    offsets, members = ctx.clusters("ACGTN")
    first, counts, _, _ = ctx.unique_table(nu, labels=False, kept=False)
    assert len(offsets) == nc + 1 and offsets[-1] == nu == len(members)
    trie = oracle.Trie("ACGTN")
    for s in strs:
        trie.add_sequence(s)
    for c in range(nc):
        rows = members[int(offsets[c]):int(offsets[c + 1])]
        got = [(int(counts[r]), strs[int(first[r])]) for r in rows]
        want = trie.pop_cluster(1)
        assert got[0] == want[0] and sorted(got) == sorted(want), c
    assert trie.number_of_sequences == 0
    # the key order itself: alphabet order, a longer key before its prefix -- other alphabets too
    order = ctx.trie_order("TGCAN", nu)
    rank = {ch: i for i, ch in enumerate("TGCAN")}
    listed = [strs[int(first[r])] for r in order]
    assert listed == sorted(listed, key=lambda s: [rank[ch] for ch in s])

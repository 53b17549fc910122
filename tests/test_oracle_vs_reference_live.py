"""Live fuzz of the CPU oracle against the real reference extensions in
oracle/_ref (built from /root/reference where it lies). Skipped where _ref has
not been built. CPU only."""
import random

import pytest


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.reference_available():
        pytest.skip("oracle/_ref not built")
    return oracle.load_reference()


def _rand(rng, syms, lo, hi):
    return "".join(rng.choice(syms) for _ in range(rng.randint(lo, hi)))


def test_trie_fuzz(oracle, ref):
    ref_trie, _ = ref
    rng = random.Random(20261003)
    for _ in range(1500):
        alpha = rng.choice(["", "ACGTN", "AC", "TGCA"])
        syms = rng.choice(["AC", "ACG", "ACGT", "ACGTN", "abcXN"])
        a, b = oracle.Trie(alpha), ref_trie.Trie(alpha)
        for _ in range(rng.randint(1, 40)):
            s = _rand(rng, syms, 0, 7)
            a.add_sequence(s)
            b.add_sequence(s)
        assert a.alphabet == b.alphabet
        assert a.memory_size() == b.memory_size()
        assert a.raw_stats() == b.raw_stats()
        d, edit = rng.randint(0, 3), rng.random() < 0.5
        for _ in range(5):
            q = _rand(rng, syms, 0, 7)
            assert a.contains_sequence(q, d, edit) == b.contains_sequence(q, d, edit)
        while b.number_of_sequences:
            assert a.pop_cluster(d, edit) == b.pop_cluster(d, edit)
            assert a.number_of_sequences == b.number_of_sequences
            if rng.random() < 0.2:
                s = _rand(rng, syms, 0, 7)
                a.add_sequence(s)
                b.add_sequence(s)
        assert a.number_of_sequences == 0


def test_distance_fuzz(oracle, ref):
    _, ref_dist = ref
    rng = random.Random(5)
    for _ in range(20000):
        s1, s2 = _rand(rng, "ACG", 0, 9), _rand(rng, "ACG", 0, 9)
        d = rng.randint(0, 4)
        for edit in (False, True):
            assert oracle.within_distance(s1, s2, d, edit) == ref_dist.within_distance(s1, s2, d, edit)

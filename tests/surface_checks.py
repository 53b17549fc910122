"""Known-answer checks written against the reference's Python surface
(``Trie`` / ``within_distance`` / ``cluster_dissection_*``), so the same body
pins the CPU oracle (``-m "not gpu"``) and the HIP product (``-m gpu``).
Data: tests/golden/reference_known_answers.json (the reference's own tests as data).
"""
import pytest


def check_within_distance(impl, ka):
    for s1, s2, d, want in ka["within_distance_hamming"]["cases"]:
        assert impl.within_distance(s1, s2, d) is want, (s1, s2, d)
    for s1, s2, d, want in ka["within_distance_levenshtein"]["cases"]:
        assert impl.within_distance(s1, s2, d, use_edit_distance=True) is want, (s1, s2, d)


def check_trie_contains(impl, ka):
    for case in ka["trie_contains"]:
        trie = impl.Trie()
        for k in case["keys"]:
            trie.add_sequence(k)
        for q, d, edit, want in case["queries"]:
            got = trie.contains_sequence(q, d, use_edit_distance=edit)
            assert got is want, (case["cite"], q, d, edit)


def check_trie_pop_cluster(impl, ka):
    for case in ka["trie_pop_cluster"]:
        trie = impl.Trie()
        for k in case["adds"]:
            trie.add_sequence(k)
        got = []
        while True:
            try:
                got.append(trie.pop_cluster(case["d"], use_edit_distance=case["edit"]))
            except LookupError:
                break
        got_sets = [frozenset(map(tuple, c)) for c in got]
        want_sets = [frozenset((n, s) for n, s in c) for c in case["clusters"]]
        assert sorted(got_sets, key=sorted) == sorted(want_sets, key=sorted), case["cite"]
        assert trie.number_of_sequences == 0


def check_trie_bookkeeping(impl, ka):
    case = ka["trie_number_of_sequences"]
    trie = impl.Trie()
    for k in case["adds"]:
        trie.add_sequence(k)
    assert trie.number_of_sequences == case["after_add"]
    while True:
        try:
            trie.pop_cluster(0)
        except LookupError:
            break
    assert trie.number_of_sequences == case["after_pop_all_d0"]
    with pytest.raises(ValueError):
        trie.add_sequence("ok")
        trie.pop_cluster(-1)
    with pytest.raises(TypeError):
        trie.add_sequence(b"bytes")
    with pytest.raises(ValueError):
        trie.add_sequence("café")
    ctor, want = ka["trie_alphabet"]["ctor"]
    assert impl.Trie(alphabet=ctor).alphabet == want
    bad, msg = ka["trie_alphabet"]["repeated"]
    with pytest.raises(ValueError) as err:
        impl.Trie(alphabet=bad)
    err.match(msg)


def check_dissection(impl, ka):
    fns = impl.CLUSTER_DISSECTION_METHODS
    for name, case in ka["dissection"].items():
        cluster = [(n, s) for n, s in case["cluster"]]
        for method in ("highest_count", "adjacency", "directional"):
            if method not in case:
                continue
            before = cluster[:]
            got = list(fns[method](cluster))
            assert cluster == before, "input list must not be mutated"
            assert len(got) == len(set(got))
            assert set(got) == set(case[method]), (name, method)


def check_trie_stats(impl, ka, lazy_alphabet=True):
    """Trie.memory_size / raw_stats / pop order on the survey's measured known answer
    (reference _triemodule.c:553-594, :909-964). lazy_alphabet: the implementation also reproduces
    the order in which the reference's trie registers symbols outside the constructor alphabet."""
    case = ka["trie_stats_known"]
    trie = impl.Trie(case["alphabet"])
    for k in case["adds"]:
        trie.add_sequence(k)
    assert trie.memory_size() == case["memory_size"]
    assert trie.raw_stats() == case["raw_stats"]
    assert trie.alphabet == case["alphabet"]
    for want in case["pop_d1"]:
        got = trie.pop_cluster(1)
        assert sorted(k for _, k in got) == sorted(want)
        assert got[0][1] == want[0]
    assert trie.number_of_sequences == 0
    assert trie.memory_size() == 0
    for k in case["then_add"]:
        trie.add_sequence(k)
    if lazy_alphabet:
        assert trie.alphabet == case["alphabet_after"]
    else:
        assert trie.alphabet.startswith(case["alphabet"]) and set(trie.alphabet) >= set("".join(case["then_add"]))

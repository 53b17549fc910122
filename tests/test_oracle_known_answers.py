"""The CPU oracle against every known answer the reference's own tests hold
(SURVEY.md section 4 / 8c). CPU only."""
import surface_checks as sc


def test_within_distance(oracle, known_answers):
    sc.check_within_distance(oracle, known_answers)


def test_trie_contains(oracle, known_answers):
    sc.check_trie_contains(oracle, known_answers)


def test_trie_pop_cluster(oracle, known_answers):
    sc.check_trie_pop_cluster(oracle, known_answers)


def test_trie_bookkeeping(oracle, known_answers):
    sc.check_trie_bookkeeping(oracle, known_answers)


def test_dissection(oracle, known_answers):
    sc.check_dissection(oracle, known_answers)


def test_alphabet_growth(oracle, known_answers):
    trie = oracle.Trie()
    for add, want in known_answers["trie_alphabet"]["growth"]:
        trie.add_sequence(add)
        assert trie.alphabet == want


def test_trie_stats_known_answer(oracle, known_answers):
    ka = known_answers["trie_stats_known"]
    trie = oracle.Trie(ka["alphabet"])
    for k in ka["adds"]:
        trie.add_sequence(k)
    assert trie.memory_size() == ka["memory_size"]
    assert trie.raw_stats() == ka["raw_stats"]
    popped = []
    while trie.number_of_sequences:
        popped.append([s for _, s in trie.pop_cluster(1)])
    assert popped == ka["pop_d1"]          # exact emission order, members included
    for k in ka["then_add"]:
        trie.add_sequence(k)
    assert trie.alphabet == ka["alphabet_after"]


def test_pass2_rule(oracle, known_answers):
    import numpy as np
    ka = known_answers["pass2_rule"]
    keys = [k.encode() for k in ka["keys"]]
    raw = np.frombuffer(b"".join(keys), dtype=np.uint8)
    off = np.concatenate([[0], np.cumsum([len(k) for k in keys])]).astype(np.uint64)
    w = np.array(ka["passes_quality"], dtype=np.uint32)
    out = oracle.dedup(raw, off, w, max_distance=ka["d"], method=ka["method"])
    assert out["kept_read_ids"].tolist() == ka["kept_read_ids"]
    assert out["n_clusters"] == ka["n_clusters"]


def test_trie_stats_known_answer(oracle, known_answers):
    sc.check_trie_stats(oracle, known_answers, lazy_alphabet=True)

"""The CPU oracle against golden vectors minted by the reference itself
(tests/golden/make_golden.py -> ref_vectors.json.gz). CPU only."""
import numpy as np
import pytest

METHODS = ("highest_count", "adjacency", "directional")


def _case_names():
    from conftest import load_ref_vectors
    return sorted(load_ref_vectors()["cases"])


@pytest.mark.parametrize("name", _case_names())
def test_trie_clusters_and_dissection(oracle, ref_vectors, name):
    case = ref_vectors["cases"][name]
    keys, weights = case["keys"], case["weights"]
    for tag, run in case["runs"].items():
        edit, d = tag[0] == "L", int(tag[1])
        trie = oracle.Trie("ACGTN")
        for k, w in zip(keys, weights):
            if w:
                trie.add_sequence(k, w)
        uniq = sorted({k for k, w in zip(keys, weights) if w})
        index = {k: i for i, k in enumerate(uniq)}
        labels = [-1] * len(uniq)
        counts = [0] * len(uniq)
        kept = {m: [] for m in METHODS}
        seeds = []
        ci = 0
        while trie.number_of_sequences:
            cl = trie.pop_cluster(d, edit)
            seeds.append(index[cl[0][1]])
            for c, k in cl:
                labels[index[k]] = ci
                counts[index[k]] = c
            for m in METHODS:
                kept[m].extend(index[k] for k in oracle.CLUSTER_DISSECTION_METHODS[m](cl, d, edit))
            ci += 1
        assert ci == run["n_clusters"], (name, tag)
        assert labels == run["labels"], (name, tag)
        assert counts == run["counts"], (name, tag)
        assert seeds == run["pop_order_seed"], (name, tag)
        for m in METHODS:
            assert sorted(kept[m]) == run["kept"][m], (name, tag, m)


@pytest.mark.parametrize("name", _case_names())
def test_whole_path_kept_read_ids(oracle, ref_vectors, name):
    """fqo_dedup (the batch form used as the checker for the HIP path) against the
    reference's kept-key sets mapped through the first-holder rule."""
    case = ref_vectors["cases"][name]
    keys, weights = case["keys"], case["weights"]
    enc = [k.encode() for k in keys]
    raw = np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8)
    off = np.concatenate([[0], np.cumsum([len(e) for e in enc])]).astype(np.uint64)
    w = np.array(weights, dtype=np.uint32)
    uniq = sorted({k for k, ww in zip(keys, weights) if ww})
    first = {}
    for i, k in enumerate(keys):
        first.setdefault(k, i)
    for tag, run in case["runs"].items():
        edit, d = tag[0] == "L", int(tag[1])
        for m in METHODS:
            out = oracle.dedup(raw, off, w, max_distance=d, use_edit_distance=edit, method=m)
            want = sorted(first[uniq[i]] for i in run["kept"][m])
            assert out["kept_read_ids"].tolist() == want, (name, tag, m)
            assert out["n_clusters"] == run["n_clusters"]
            assert out["n_unique"] == len(uniq)

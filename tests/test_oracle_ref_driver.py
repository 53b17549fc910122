"""oracle/ref_driver.py (the CPU-baseline driver around the real reference
extensions) against the golden vectors and against the C oracle. CPU only."""
import pytest


@pytest.fixture(scope="module")
def driver(oracle):
    if not oracle.reference_available():
        pytest.skip("oracle/_ref not built")
    from oracle import ref_driver
    return ref_driver


def test_driver_matches_golden(driver, ref_vectors):
    for name in ("synth_s11_n3000_L20", "giant_L6", "ref_test_cluster"):
        case = ref_vectors["cases"][name]
        keys = [k for k, w in zip(case["keys"], case["weights"]) for _ in range(w)]
        uniq = sorted(set(keys))
        for tag in ("H1", "L1"):
            run = case["runs"][tag]
            for m in ("directional", "adjacency", "highest_count"):
                out = driver.run_reference_path(keys, int(tag[1]), tag[0] == "L", m)
                assert sorted(out["kept_keys"]) == [uniq[i] for i in run["kept"][m]], (name, tag, m)
                assert out["n_clusters"] == run["n_clusters"]


def test_c_oracle_matches_reference_at_scale(driver, oracle):
    """200 k reads through the real reference trie vs fqo_dedup (the checker the GPU tests use)."""
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    n, L = 200_000, 50
    keys = synth_keys(n, L, 8, 31, sub_rate=2e-3, n_rate=2e-4)
    strs = [bytes(r).decode() for r in keys]
    first = {}
    for i, s in enumerate(strs):
        first.setdefault(s, i)
    for d, m in ((1, "directional"), (2, "adjacency")):
        ref = driver.run_reference_path(strs, d, False, m)
        mine = oracle.dedup(keys.reshape(-1), fixed_offsets(n, L), max_distance=d, method=m)
        assert sorted(first[k] for k in ref["kept_keys"]) == mine["kept_read_ids"].tolist()
        assert ref["n_clusters"] == mine["n_clusters"] and ref["n_unique"] == mine["n_unique"]

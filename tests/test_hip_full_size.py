"""Parity at the sizes bench.py times (BASELINE.json configs 2 and 3 in full): the HIP path with
its DEFAULT switches -- so the fused pack -> collapse route, the 8192 level-1 slabs, the 2^16
level-2 cursors and the id-bin kept list engage on their own -- against the CPU oracle on the very
same bytes. The oracle needs about as long as the reference (~2 min for
50 M x 32 nt), so both oracle runs start in threads (ctypes drops the GIL) before any GPU work.
GPU only."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (name, reads, key length, umi, seed, d, method): bench.py's WORKLOADS, unchanged
CONFIGS = {
    "config3": (50_000_000, 32, 32, 1003, 1, "directional"),
    "config2": (10_000_000, 100, 12, 1002, 1, "directional"),
}


class _OracleRun:
    def __init__(self, oracle, host, n, L, d, method):
        from fastqdedup_amd.synth import fixed_offsets
        self.out, self.err = None, None

        def work():
            try:
                self.out = oracle.dedup(host, fixed_offsets(n, L), max_distance=d, method=method)
            except BaseException as exc:  # surfaced by result()
                self.err = exc
        self.thread = threading.Thread(target=work, daemon=True)
        self.thread.start()

    def result(self):
        self.thread.join()
        if self.err is not None:
            raise self.err
        return self.out


@pytest.fixture(scope="module")
def full_size(oracle):
    """Device keys of both configurations + their oracle answers, running in the background."""
    import torch
    import fastqdedup_amd as F
    ctx = F.Context(0)
    jobs = {}
    for name, (n, L, umi, seed, d, method) in CONFIGS.items():
        dev = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
        ctx.synth_keys(dev, n, 0, n, L, umi, seed)
        host = dev.cpu().numpy()
        jobs[name] = (dev, _OracleRun(oracle, host, n, L, d, method))
    yield F, ctx, jobs
    jobs.clear()


@pytest.mark.parametrize("name", ["config2", "config3"])
def test_timed_configuration_matches_oracle_at_full_size(full_size, name, monkeypatch):
    F, ctx, jobs = full_size
    for var in ("FQD_COLLAPSE", "FQD_NO_FUSED_PACK", "FQD_FUSED_MIN_READS", "FQD_EDGES", "FQD_LDS_NO_SLABS",
                "FQD_GROUP_NO_SLABS", "FQD_KEPT_BY_MAP", "FQD_KEPT_BY_SORT", "FQD_DIRECTIONAL_ROUNDS",
                "FQD_NO_COMPACT_RECORDS"):
        monkeypatch.delenv(var, raising=False)       # the switches bench.py runs with: none
    n, L, umi, seed, d, method = CONFIGS[name]
    dev, run = jobs[name]
    got = F.cluster_keys(dev, key_len=L, max_distance=d, method=method, context=ctx)
    again = F.cluster_keys(dev, key_len=L, max_distance=d, method=method, context=ctx)
    assert np.array_equal(got.kept_read_ids, again.kept_read_ids)      # a warm context answers the same
    times = ctx.kernel_times(reset=True)
    if name == "config3":
        # the route bench.py times: pack fused with level 1 (no level-1 scatter launch), 12-byte records through
        # level 2 and the LDS dedupe (the keys with an N -- 160 K of 50 M reads -- through the side path), the reads
        # binned by segment 0 and search pass 0 done by the compaction
        assert got.route["fused_pack"] and got.route["compact_records"] and got.route["pass0_in_collapse"], got.route
        assert got.route["pass0_continued"] and not got.route["restarted"] and not got.route["search_retried"], got.route
        assert times["pack_kernel"][1] and not times["part_scatter_kernel<1>"][1], times
        assert times["part_scatter12_kernel"][1] and times["bucket_dedupe12_kernel"][1], times
        assert not times["part_scatter_kernel<2>"][1] and not times["bucket_dedupe_kernel"][1], times
    want = run.result()
    assert got.n_reads == n
    assert got.n_unique == want["n_unique"]
    assert got.n_clusters == want["n_clusters"]
    assert got.n_kept == len(want["kept_read_ids"])
    assert np.array_equal(got.kept_read_ids, want["kept_read_ids"])


def test_skewed_workload_matches_oracle(oracle):
    """SURVEY.md 7.4 "Skew": 4 M reads of config 3's shape under the skewed model of fastqdedup_amd/synth.py -- a key
    with 80 000 copies, heavy-tailed abundance, 1 % of the molecules poly-A in segment 0 (10 000 keys in one bucket of
    the search), a ladder of keys that is ONE component of tens of thousands of members -- against the oracle's trie
    (which takes any distribution, _triemodule.c:380-495). A warm context (it has met the crowded buckets and the
    overfull slabs on the first job) answers the same, and never with a device-wide sort."""
    import torch
    import fastqdedup_amd as F
    from fastqdedup_amd.synth import SKEW, fixed_offsets
    n, L = 4_000_000, 32
    ctx = F.Context(0)
    dev = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
    ctx.synth_keys(dev, n, 0, n, L, L, 1003, skew=SKEW)
    host = dev.cpu().numpy()
    run = _OracleRun(oracle, host, n, L, 1, "directional")
    first = F.cluster_keys(dev, key_len=L, max_distance=1, method="directional", context=ctx)
    warm = F.cluster_keys(dev, key_len=L, max_distance=1, method="directional", context=ctx)
    want = run.result()
    for got in (first, warm):
        assert (got.n_unique, got.n_clusters) == (want["n_unique"], want["n_clusters"])
        assert np.array_equal(got.kept_read_ids, want["kept_read_ids"])
    assert warm.route["search_grouped"] and not warm.route["search_sort"] and not warm.route["collapse_sort"], warm.route
    # the same under the adjacency method and at distance 0 (no search at all)
    for method, d in (("adjacency", 1), ("highest_count", 1), ("directional", 0)):
        want2 = oracle.dedup(host, fixed_offsets(n, L), max_distance=d, method=method)
        got2 = F.cluster_keys(dev, key_len=L, max_distance=d, method=method, context=ctx)
        assert (got2.n_unique, got2.n_clusters) == (want2["n_unique"], want2["n_clusters"]), (method, d)
        assert np.array_equal(got2.kept_read_ids, want2["kept_read_ids"]), (method, d)


@pytest.mark.parametrize("n,L,ladder", [(2_000_000, 32, True), (1_000_000, 300, False), (1_000_000, 300, True)])
def test_skewed_workload_at_distance_2_matches_oracle(oracle, n, L, ladder):
    """The skewed model at Hamming d = 2 (BASELINE config 4's distance; 2 M reads of 32 nt, and config 4's own key length:
    1 M reads of 300 nt -- the (hash, position) collapse, 128-byte records, the cooperative verification): crowded segment values are
    matched on finer pieces -- every set of 2 of 8 pieces masked out, a pair reported under the smallest set that holds
    its differences (group.hip "crowded buckets") -- instead of pairwise on the sort path, which rounds 2-3 fell back to
    from d = 2 on. Against the oracle's trie (`TrieNode_FindNearest`, `_triemodule.c:380-495`, takes any distribution);
    the warm context must not sort."""
    import torch
    import fastqdedup_amd as F
    from fastqdedup_amd.synth import SKEW
    d = 2
    ctx = F.Context(0)
    dev = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
    # (300-nt keys without the model's ladder: the refinement alone; with it: the ladder's 4^8 keys vary inside ONE of the
    # 16 fine pieces of a 300-nt key and stay one group of the refinement -- the context then compares its crowded buckets
    # all pairs in tiles (group.hip "the last resort"; until round 4 this case fell to the sort path, quadratic in one
    # wave); at 32 nt a piece is two bases and the ladder splits into groups of 256)
    skew = SKEW if ladder else {"hot": 0.02, "ladder": 0.0, "lowc_every": 100}
    ctx.synth_keys(dev, n, 0, n, L, L, 1004, skew=skew)
    host = dev.cpu().numpy()
    runs = {m: _OracleRun(oracle, host, n, L, d, m) for m in ("directional", "adjacency")}
    first = F.cluster_keys(dev, key_len=L, max_distance=d, method="directional", context=ctx)
    warm = F.cluster_keys(dev, key_len=L, max_distance=d, method="directional", context=ctx)
    want = runs["directional"].result()
    for got in (first, warm):
        assert (got.n_unique, got.n_clusters) == (want["n_unique"], want["n_clusters"])
        assert np.array_equal(got.kept_read_ids, want["kept_read_ids"])
    # (32-nt keys at d = 2: a third of a key is 10-11 bases -- even uniform keys share segment values by the dozen, the
    # candidate lists outgrow their budget and the search takes the sort path, skew or not; 300-nt keys must not sort)
    assert warm.route["search_refined"] or warm.route["search_tiles"], warm.route
    if L == 300:
        assert warm.route["search_grouped"] and not warm.route["search_sort"], warm.route
        assert warm.route["search_tiles"], warm.route       # (few enough crowded pairs to take them all, with or without the ladder)
    want2 = runs["adjacency"].result()
    got2 = F.cluster_keys(dev, key_len=L, max_distance=d, method="adjacency", context=ctx)
    assert (got2.n_unique, got2.n_clusters) == (want2["n_unique"], want2["n_clusters"])
    assert np.array_equal(got2.kept_read_ids, want2["kept_read_ids"])


def test_long_keys_match_oracle_at_5m(oracle):
    """The shapes of BASELINE.json's configs 4 and 5 at 5 M reads of 300 nt, the largest the oracle's trie answers in
    about two minutes (96 s and 58 s for the reference): Hamming d = 2 directional -- three
    routed passes' worth of search on 128-byte records, the (hash, position) collapse -- and Levenshtein d = 1
    adjacency over keys of 299 / 300 / 301 nt (a 1 % indel tail: the ragged pack and collapse, the Hamming passes for
    pairs of one length and the edit search proper for pairs of different lengths). Both oracle runs start in threads
    before any GPU work."""
    import torch
    import fastqdedup_amd as F
    n, L = 5_000_000, 300
    ctx = F.Context(0)
    dev4 = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
    ctx.synth_keys(dev4, n, 0, n, L, L, 1004)
    host4 = dev4.cpu().numpy()
    run4 = _OracleRun(oracle, host4, n, L, 2, "directional")
    dev5, off5 = ctx.synth_indel_keys(n, 0, n, L, L, 1005, indel_rate=0.01)
    host5, hoff5 = dev5.cpu().numpy(), off5.cpu().numpy().astype(np.uint64)
    out5, err5 = {}, []

    def work5():
        try:
            out5.update(oracle.dedup(host5, hoff5, max_distance=1, use_edit_distance=True, method="adjacency"))
        except BaseException as exc:
            err5.append(exc)
    t5 = threading.Thread(target=work5, daemon=True)
    t5.start()
    got4 = F.cluster_keys(dev4, key_len=L, max_distance=2, method="directional", context=ctx)
    got5 = F.cluster_keys(dev5, off5, 0, max_distance=1, use_edit_distance=True, method="adjacency", context=ctx)
    assert got4.route["collapse_pairs"] and got4.route["search_grouped"] and not got4.route["search_sort"], got4.route
    assert got5.route["search_edit"] and not got5.route["collapse_sort"], got5.route
    want4 = run4.result()
    assert (got4.n_unique, got4.n_clusters) == (want4["n_unique"], want4["n_clusters"])
    assert np.array_equal(got4.kept_read_ids, want4["kept_read_ids"])
    t5.join()
    if err5:
        raise err5[0]
    assert (got5.n_unique, got5.n_clusters) == (out5["n_unique"], out5["n_clusters"])
    assert np.array_equal(got5.kept_read_ids, out5["kept_read_ids"])

"""Parity of the HIP path (through the C ABI) with the reference: golden vectors
minted by the reference, the reference's own known answers, and the CPU oracle
on seeded inputs. GPU only."""
import os

import numpy as np
import pytest

import surface_checks as sc

pytestmark = pytest.mark.gpu

METHODS = ("highest_count", "adjacency", "directional")


@pytest.fixture(scope="module")
def F():
    import fastqdedup_amd
    return fastqdedup_amd


@pytest.fixture(scope="module")
def ctx(F):
    return F.Context(0)


def _pack(keys):
    enc = [k.encode() for k in keys]
    raw = np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8)[: sum(len(e) for e in enc)]
    off = np.concatenate([[0], np.cumsum([len(e) for e in enc])]).astype(np.uint64)
    return raw, off


def _golden_cases():
    from conftest import load_ref_vectors
    return sorted(load_ref_vectors()["cases"])


@pytest.mark.parametrize("name", _golden_cases())
def test_golden_vectors(F, ctx, ref_vectors, name):
    case = ref_vectors["cases"][name]
    keys, weights = case["keys"], case["weights"]
    raw, off = _pack(keys)
    w = np.array(weights, dtype=np.uint32)
    uniq = sorted({k for k, ww in zip(keys, weights) if ww})
    first = {}
    for i, k in enumerate(keys):
        first.setdefault(k, i)
    for tag, run in case["runs"].items():
        edit, d = tag[0] == "L", int(tag[1])
        for m in METHODS:
            got = F.cluster_keys(raw, off, weights=w, max_distance=d, use_edit_distance=edit,
                                 method=m, context=ctx)
            want = sorted(first[uniq[i]] for i in run["kept"][m])
            assert got.n_unique == len(uniq), (name, tag)
            assert got.n_clusters == run["n_clusters"], (name, tag)
            assert got.kept_read_ids.tolist() == want, (name, tag, m)
        # component partition, key for key
        fid, cnt, lab, _ = ctx.unique_table(got.n_unique, labels=True, kept=False)
        dev_keys = [keys[int(i)] for i in fid]
        assert sorted(dev_keys) == uniq
        ref_label = dict(zip(uniq, run["labels"]))
        ref_count = dict(zip(uniq, run["counts"]))
        seen = {}
        for k, c, l in zip(dev_keys, cnt, lab):
            assert int(c) == ref_count[k]
            assert seen.setdefault(int(l), ref_label[k]) == ref_label[k], "component split/merged"
        assert len(seen) == run["n_clusters"]


def test_known_answers_within_distance(F, known_answers):
    sc.check_within_distance(F, known_answers)


def test_known_answers_dissection(F, known_answers):
    sc.check_dissection(F, known_answers)


def test_known_answers_pop_cluster(F, known_answers):
    sc.check_trie_pop_cluster(F, known_answers)


def test_known_answers_contains(F, known_answers):
    sc.check_trie_contains(F, known_answers)


def test_known_answers_trie_bookkeeping(F, known_answers):
    sc.check_trie_bookkeeping(F, known_answers)


def test_pass2_rule(F, ctx, known_answers):
    ka = known_answers["pass2_rule"]
    raw, off = _pack(ka["keys"])
    w = np.array(ka["passes_quality"], dtype=np.uint32)
    got = F.cluster_keys(raw, off, weights=w, max_distance=ka["d"], method=ka["method"], context=ctx)
    assert got.kept_read_ids.tolist() == ka["kept_read_ids"]
    assert got.n_clusters == ka["n_clusters"]
    assert got.n_counted == ka["processed"]


@pytest.mark.parametrize("n,L,umi,d,sub,nr", [
    (10000, 50, 8, 1, 1e-3, 1e-4),      # BASELINE config 1
    (200000, 100, 12, 1, 2e-3, 2e-4),   # config 2 shape
    (300000, 32, 32, 1, 2e-3, 2e-4),    # config 3 shape (key = R1[:16] + R2[:16])
    (60000, 300, 300, 2, 1e-3, 1e-4),   # config 4 shape, d=2
    (50000, 24, 6, 2, 5e-3, 1e-3),
    (40000, 12, 4, 3, 5e-3, 1e-3),      # low complexity, larger buckets, d=3
])
def test_against_oracle_synthetic(F, ctx, oracle, n, L, umi, d, sub, nr):
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    keys = synth_keys(n, L, umi, 4242 + L, sub_rate=sub, n_rate=nr).reshape(-1)
    off = fixed_offsets(n, L)
    for m in METHODS:
        got = F.cluster_keys(keys, key_len=L, max_distance=d, method=m, context=ctx)
        want = oracle.dedup(keys, off, max_distance=d, method=m)
        assert got.n_unique == want["n_unique"]
        assert got.n_clusters == want["n_clusters"]
        assert np.array_equal(got.kept_read_ids, want["kept_read_ids"]), (m, n, L, d)


def test_large_path_sweep_against_oracle(F, oracle):
    """Random jobs big enough for the paths large inputs take -- LDS collapse (n >= 32768, keys of
    <= 32 nt), partitioned search passes (>= 65536 unique keys), kept list by window compaction,
    closed-form directional -- over key lengths, alphabets, distances 0..3, the three dissection
    rules and quality weights. Every job must give the oracle's kept ids and counters."""
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    rng = np.random.default_rng(20261004)
    methods = list(METHODS)
    for job in range(14):
        L = int(rng.choice([12, 16, 20, 27, 32, 33, 48, 64, 90]))
        n = int(rng.integers(90_000, 320_000))
        umi = int(rng.choice([L, L, max(4, L // 3)]))
        d = int(rng.choice([0, 1, 1, 2, 3]))
        sub = float(rng.choice([1e-3, 4e-3, 1e-2]))
        nr = float(rng.choice([0.0, 3e-4]))
        copies = int(rng.choice([1, 2, 4, 9]))
        keys = synth_keys(n, L, umi, 9000 + job, copies=copies, sub_rate=sub, n_rate=nr)
        if job % 4 == 3:                       # an alphabet that needs a scan: lower-case symbols too
            low = rng.random(n) < 0.01
            keys[low] = np.char.lower(keys[low].view("S1")).view(np.uint8)
        raw = np.ascontiguousarray(keys).reshape(-1)
        w = None if job % 3 else rng.choice(np.array([0, 1, 1, 1, 3], dtype=np.uint32), size=n)
        m = methods[job % 3]
        got = F.cluster_keys(raw, key_len=L, weights=w, max_distance=d, method=m, context=F.Context(0))
        want = oracle.dedup(raw, fixed_offsets(n, L), w, max_distance=d, method=m)
        tag = (job, n, L, umi, d, m, copies)
        assert (got.n_unique, got.n_clusters) == (want["n_unique"], want["n_clusters"]), tag
        assert np.array_equal(got.kept_read_ids, want["kept_read_ids"]), tag


def test_lds_collapse_slab_overflow_retries_exactly(F, oracle):
    """Level 2 of the LDS collapse gives every bucket a fixed slab (no histogram pass). A key with
    thousands of copies overfills its slab: the level must be redone with exact bucket sizes, and
    the answer must be the oracle's -- on this job and on the next one of the same context."""
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    n, L = 300_000, 32
    keys = synth_keys(n, L, L, 81, sub_rate=3e-3, n_rate=2e-4)
    rng = np.random.default_rng(8)
    heavy = rng.choice(n, size=4000, replace=False)
    keys[heavy] = keys[heavy[0]]                     # one key, 4000 copies
    keys[heavy[::9], 5] = ord("N")                   # and a few hundred near copies of it
    raw = np.ascontiguousarray(keys).reshape(-1)
    ctx = F.Context(0)
    want = oracle.dedup(raw, fixed_offsets(n, L), max_distance=1, method="directional")
    for _ in range(2):
        got = F.cluster_keys(raw, key_len=L, max_distance=1, method="directional", context=ctx)
        assert (got.n_unique, got.n_clusters) == (want["n_unique"], want["n_clusters"])
        assert np.array_equal(got.kept_read_ids, want["kept_read_ids"])
    plain = F.cluster_keys(np.ascontiguousarray(synth_keys(n, L, L, 82)).reshape(-1), key_len=L, context=ctx)
    assert plain.n_unique > 0


@pytest.mark.parametrize("case", ["plain", "weights", "foreign_byte", "crowded_part", "len20", "uint4_records",
                                  "many_n", "n_copies", "acgt_two_planes", "acgt_weights"])
def test_fused_pack_collapse_matches_oracle(F, oracle, monkeypatch, case):
    """cluster_keys is ONE C call (fqd_cluster_keys); for short fixed-length keys the pack kernel
    then partitions its records straight into the collapse (no packed reads in read order). Same
    answer as the oracle and as the two-call way -- also when the attempt has to be abandoned: a byte
    outside "ACGNT" (the plain pack learns the alphabet) or a key with so many copies that a
    level-1 slab overflows (plain pack from then on, for this context)."""
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    monkeypatch.setenv("FQD_FUSED_MIN_READS", "100000")
    n, L = 300_000, (20 if case == "len20" else 32)
    # the records travel as 12 bytes (two key words + read index), keys with an N as uint4 through side slabs
    # and a hash table of their own: "uint4_records" pins the older format, "many_n" overfills the side slabs
    # (uint4 records from then on), "n_copies" gives the side path duplicates to count, "acgt_*": a two-plane
    # alphabet needs no side path
    if case == "uint4_records":
        monkeypatch.setenv("FQD_NO_COMPACT_RECORDS", "1")
    two_planes = case.startswith("acgt")
    keys = synth_keys(n, L, L, 91, sub_rate=3e-3, n_rate=0 if two_planes else 2e-2 if case == "many_n" else 2e-4)
    rng = np.random.default_rng(4)
    weights = None
    if case in ("weights", "acgt_weights"):
        weights = rng.choice(np.array([0, 1, 1, 1], dtype=np.uint32), size=n)
    if case == "n_copies":
        src = rng.choice(n, size=400, replace=False)
        keys[src, rng.integers(0, L, size=400)] = ord("N")
        for k in range(5):                      # five more holders of each of those keys
            keys[rng.choice(n, size=400, replace=False)] = keys[src]
    if case == "foreign_byte":
        keys[rng.choice(n, size=50, replace=False), 7] = ord("R")
    if case == "crowded_part":
        heavy = rng.choice(n, size=40_000, replace=False)
        keys[heavy] = keys[heavy[0]]
        keys[heavy[::11], 3] = ord("N")
    raw = np.ascontiguousarray(keys).reshape(-1)
    want = oracle.dedup(raw, fixed_offsets(n, L), weights=weights, max_distance=1, method="directional")
    ctx = F.Context(0)
    if two_planes:
        present = np.zeros(128, dtype=np.uint8)
        present[[ord(ch) for ch in "ACGT"]] = 1
        ctx.configure(present, L, False)
    for job in range(2):        # the second job runs on a context that has learnt (fused_off / slab_off / compact_off)
        ctx.kernel_times(reset=True)
        got = F.cluster_keys(raw, key_len=L, weights=weights, max_distance=1, method="directional", context=ctx)
        kt = ctx.kernel_times(reset=True)
        # which way in: the plain level-1 scatter runs only when the fused attempt was given up
        fused_only = case != "foreign_byte"
        assert kt["part_scatter_kernel<1>"][1] == (0 if fused_only else 1)
        # (crowded_part: the key with 40 000 copies fills a level-1 slab -- once more, and from then on, with the spill
        # list. Until round 4 its 3 600 copies with an N then overfilled the side slabs of the few level-2 tiles they sat
        # in and the fused way in was given up for the context; the pack kernel now takes keys with an N out itself, a
        # workgroup's share into its own slab, and the attempt with the spill list goes through.)
        assert kt["pack_kernel"][1] == (2 if case in ("many_n", "crowded_part") and job == 0 else
                                        1 if fused_only else 3)
        if case == "crowded_part":
            assert got.route["fused_pack"] and got.route["spill_list"] and got.route["compact_records"], got.route
        # ... and which records: 12-byte ones unless pinned or given up
        if fused_only and case != "crowded_part":
            compact = case != "uint4_records" and not (case == "many_n" and job == 1)
            assert kt["part_scatter12_kernel"][1] == (1 if compact else 0)
            assert kt["part_scatter_kernel<2>"][1] == (0 if compact and case != "many_n" else 1)
        assert got.n_counted == (n if weights is None else int(weights.sum()))
        assert (got.n_unique, got.n_clusters) == (want["n_unique"], want["n_clusters"])
        assert np.array_equal(got.kept_read_ids, want["kept_read_ids"])
    # the two-call way on the same context, and what the one-call way leaves behind
    ctx.pack_keys(raw, None, L)
    two = ctx.cluster(weights, None, max_distance=1, method=2)
    assert (two["n_unique"], two["n_edges"], two["n_kept"]) == (got.n_unique, got.n_edges, got.n_kept)
    if case == "plain":
        F.cluster_keys(raw, key_len=L, context=ctx)
        with pytest.raises(Exception, match="fqd_collapse"):
            ctx.collapse()       # the packed reads were never written: needs fqd_pack_keys first


@pytest.mark.parametrize("switch", ["FQD_NODE_RECORDS", "FQD_NO_NODE_RECORDS"])
@pytest.mark.parametrize("d", [1, 2])
def test_components_and_pass_1_on_node_records(F, oracle, monkeypatch, switch, d):
    """The closed-form directional dissection behind fqd_cluster_keys: components and pass 1 in ONE sweep over the edges
    on node records (parent, state byte) -- the default from distance 2 on -- or as the union-find beside pass 1 on
    their own arrays; both pinned at both distances, with keys of 15 and more copies (the state byte's count nibble
    saturates) and a giant component (the union-find's two phases on the second job). Against the oracle
    (`cluster_dissection_directional`, `__init__.py:60-91`; `Trie.pop_cluster`, `_triemodule.c:778-897`)."""
    from fastqdedup_amd.synth import SKEW, fixed_offsets, synth_keys
    monkeypatch.setenv(switch, "1")
    n, L = 400_000, 32
    keys = synth_keys(n, L, 12, 77 + d, sub_rate=4e-3, n_rate=1e-4, skew=SKEW)
    raw = np.ascontiguousarray(keys).reshape(-1)
    want = oracle.dedup(raw, fixed_offsets(n, L), max_distance=d, method="directional")
    ctx = F.Context(0)
    for job in range(2):
        got = F.cluster_keys(raw, key_len=L, max_distance=d, method="directional", context=ctx)
        assert (got.n_unique, got.n_clusters, got.n_kept) == (want["n_unique"], want["n_clusters"],
                                                              len(want["kept_read_ids"])), (switch, d, job)
        assert np.array_equal(got.kept_read_ids, want["kept_read_ids"]), (switch, d, job)
    # another method on the same context afterwards (the state bytes of the sweep must not leak into it)
    want2 = oracle.dedup(raw, fixed_offsets(n, L), max_distance=d, method="adjacency")
    got2 = F.cluster_keys(raw, key_len=L, max_distance=d, method="adjacency", context=ctx)
    assert np.array_equal(got2.kept_read_ids, want2["kept_read_ids"]), (switch, d)


@pytest.mark.parametrize("d", [1, 2, 3])
def test_crowded_buckets_are_matched_on_finer_pieces(F, oracle, monkeypatch, d):
    """Crowded segment values at distances 1, 2 and 3 (group.hip "crowded buckets": sets of d pieces masked out, a pair
    reported under the smallest set that holds its differing pieces, and only if the first main segment it agrees on is
    a crowded one): the skewed model at 300 K reads with the crowding limit lowered to 64 items, so that hundreds of
    buckets take that way. Against the oracle's trie (`TrieNode_FindNearest`, `_triemodule.c:380-495`), two methods."""
    from fastqdedup_amd.synth import SKEW, fixed_offsets, synth_keys
    monkeypatch.setenv("FQD_GROUP_CROWDED_LIMIT", "64")
    monkeypatch.setenv("FQD_GROUP_TILE_BUDGET", "0")          # (so few crowded keys would go all pairs in tiles by default)
    n, L = 300_000, 32
    keys = synth_keys(n, L, 12, 900 + d, sub_rate=4e-3, n_rate=1e-4, skew=SKEW)
    raw = np.ascontiguousarray(keys).reshape(-1)
    ctx = F.Context(0)
    for method in ("directional", "adjacency"):
        want = oracle.dedup(raw, fixed_offsets(n, L), max_distance=d, method=method)
        for job in range(2):        # (the first job meets the overfull slabs; from the second on: exact sizes, refinement)
            got = F.cluster_keys(raw, key_len=L, max_distance=d, method=method, context=ctx)
            assert (got.n_unique, got.n_clusters, got.n_kept) == (want["n_unique"], want["n_clusters"],
                                                                  len(want["kept_read_ids"])), (d, method, job)
            assert np.array_equal(got.kept_read_ids, want["kept_read_ids"]), (d, method, job)
    # (d = 3 on 32-nt keys: the candidates of the fine groups outgrow the budget and the context turns to all pairs in tiles)
    assert (got.route["search_refined"] or got.route["search_tiles"]) and got.route["search_grouped"], got.route
    assert not got.route["search_sort"], got.route


def test_crowded_buckets_beyond_the_fine_items_go_all_pairs(F, oracle, monkeypatch):
    """A segment value shared by thousands of keys is matched on finer segments (group.hip "crowded buckets"); when the
    crowded keys are more than the fine items can address, the crowded buckets are compared ALL PAIRS in tiles (group.hip
    "the last resort"), and where that is switched off (or the records are too long for its LDS tiles) the search runs
    again on the sort path -- never an error: the reference's trie takes any distribution (`_triemodule.c:380-495`).
    FQD_GROUP_FINE_LIMIT makes "too many" small."""
    from fastqdedup_amd.synth import SKEW, fixed_offsets, synth_keys
    monkeypatch.setenv("FQD_FUSED_MIN_READS", "100000")
    n, L = 300_000, 32
    keys = synth_keys(n, L, 12, 171, sub_rate=3e-3, n_rate=1e-4, skew=SKEW)
    raw = np.ascontiguousarray(keys).reshape(-1)
    want = oracle.dedup(raw, fixed_offsets(n, L), max_distance=1, method="directional")
    ctx = F.Context(0)
    monkeypatch.setenv("FQD_GROUP_TILE_BUDGET", "0")          # (so few crowded keys would go all pairs in tiles by default)
    got = F.cluster_keys(raw, key_len=L, max_distance=1, method="directional", context=ctx)      # (the context meets the skew)
    assert got.route["search_refined"] and not got.route["search_tiles"], got.route
    monkeypatch.setenv("FQD_GROUP_FINE_LIMIT", "1000")
    got = F.cluster_keys(raw, key_len=L, max_distance=1, method="directional", context=ctx)
    assert got.route["search_tiles"] and got.route["search_grouped"] and not got.route["search_sort"], got.route
    assert (got.n_unique, got.n_clusters, got.n_kept) == (want["n_unique"], want["n_clusters"], len(want["kept_read_ids"]))
    assert np.array_equal(got.kept_read_ids, want["kept_read_ids"])
    monkeypatch.setenv("FQD_GROUP_NO_TILES", "1")
    got = F.cluster_keys(raw, key_len=L, max_distance=1, method="directional", context=ctx)
    assert not got.route["search_tiles"] and got.route["search_sort"] and got.route["search_retried"], got.route
    assert (got.n_unique, got.n_clusters, got.n_kept) == (want["n_unique"], want["n_clusters"], len(want["kept_read_ids"]))
    assert np.array_equal(got.kept_read_ids, want["kept_read_ids"])


@pytest.mark.parametrize("L,d,budget", [(300, 1, "0"), (100, 2, "0"), (300, 2, None), (100, 4, None)])
def test_a_family_inside_one_piece_goes_all_pairs(F, oracle, monkeypatch, L, d, budget):
    """The skewed model's LADDER -- every value of eight adjacent bases behind one prefix -- varies inside ONE fine piece
    of a long key: the refinement cannot split it, its candidate pairs outgrow the budget, and the context then compares
    crowded buckets all pairs in tiles (`gp_crowded_tiles_kernel`) instead of walking them on the sort path. d = 4 has no
    fine pieces at all and goes to the tiles at once, and so do crowded buckets of few enough pairs (budget None: the
    default; "0": the fine pieces first, as for a job with millions of crowded keys). Against the oracle's trie
    (`TrieNode_FindNearest`, `_triemodule.c:380-495`)."""
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    monkeypatch.setenv("FQD_EDGES", "grouped")
    if budget is not None:
        monkeypatch.setenv("FQD_GROUP_TILE_BUDGET", budget)
        monkeypatch.setenv("FQD_GROUP_CAND_BUDGET", "300000")      # (so that the ladder keys left in one fine group are too many)
    n = 80_000                # (the oracle's trie walks the ladder too: its time grows with the square of the ladder)
    keys = synth_keys(n, L, 12, 177, sub_rate=3e-3, n_rate=1e-4, skew={"hot": 0.02, "ladder": 0.08, "lowc_every": 100})
    raw = np.ascontiguousarray(keys).reshape(-1)
    ctx = F.Context(0)
    for method in ("directional", "adjacency"):
        want = oracle.dedup(raw, fixed_offsets(n, L), max_distance=d, method=method)
        got = F.cluster_keys(raw, key_len=L, max_distance=d, method=method, context=ctx)
        assert (got.n_unique, got.n_clusters, got.n_kept) == (want["n_unique"], want["n_clusters"],
                                                              len(want["kept_read_ids"])), (method, got.route)
        assert np.array_equal(got.kept_read_ids, want["kept_read_ids"]), method
        assert got.route["search_tiles"] and not got.route["search_sort"], (method, got.route)


@pytest.mark.parametrize("d", [1, 2])
@pytest.mark.parametrize("route", ["tiles", "fine pieces"])
def test_crowded_buckets_of_ragged_keys(F, oracle, monkeypatch, d, route):
    """Keys of SEVERAL lengths (58-62 nt) that share their first 40 bases by the thousand: crowded buckets of a ragged
    search -- equal records of different lengths are different keys, the segments of a pair are those of ITS length --
    through the tiles and through the fine pieces. Against the oracle's trie (`_triemodule.c:380-495`)."""
    import random
    rng = random.Random(40 + d)
    monkeypatch.setenv("FQD_EDGES", "grouped")
    monkeypatch.setenv("FQD_GROUP_TILE_BUDGET", "0" if route == "fine pieces" else "1000000000000")
    stem = "".join(rng.choice("ACGT") for _ in range(40))
    mols = [stem + "".join(rng.choice("ACGT") for _ in range(rng.randint(18, 22))) for _ in range(6000)]
    mols += ["".join(rng.choice("ACGT") for _ in range(rng.randint(58, 62))) for _ in range(30_000)]
    strs = []
    for _ in range(150_000):
        m = list(rng.choice(mols))
        for _ in range(rng.choice((0, 0, 0, 1, 1, 2))):
            m[rng.randrange(len(m))] = rng.choice("ACGT")
        strs.append("".join(m))
    strs += [stem[:k] for k in range(30, 41)] * 2 + [stem + "A" * k for k in range(0, 20)]
    raw, off = _pack(strs)
    ctx = F.Context(0)
    for method in ("directional", "adjacency"):
        want = oracle.dedup(raw, off, max_distance=d, method=method)
        for job in range(2):
            got = F.cluster_keys(raw, off, max_distance=d, method=method, context=ctx)
            assert (got.n_unique, got.n_clusters, got.n_kept) == (want["n_unique"], want["n_clusters"],
                                                                  len(want["kept_read_ids"])), (method, job, got.route)
            assert np.array_equal(got.kept_read_ids, want["kept_read_ids"]), (method, job)
    assert got.route["search_tiles" if route == "tiles" else "search_refined"] and not got.route["search_sort"], got.route


def test_crowded_buckets_on_a_two_plane_alphabet_go_all_pairs(F, oracle, monkeypatch):
    """ACGT only (two bit planes, `gp_crowded_tiles_kernel<2, 1>`): the skewed model without its N, crowded buckets all
    pairs in tiles."""
    from fastqdedup_amd.synth import SKEW, fixed_offsets, synth_keys
    monkeypatch.setenv("FQD_EDGES", "grouped")
    monkeypatch.setenv("FQD_GROUP_TILE_BUDGET", "1000000000000")
    n, L = 200_000, 32
    keys = synth_keys(n, L, 12, 178, sub_rate=3e-3, n_rate=0, skew=SKEW)
    raw = np.ascontiguousarray(keys).reshape(-1)
    present = np.zeros(128, dtype=np.uint8)
    present[[ord(ch) for ch in "ACGT"]] = 1
    for d in (1, 2):
        ctx = F.Context(0)
        ctx.configure(present, L, False)
        want = oracle.dedup(raw, fixed_offsets(n, L), max_distance=d, method="directional")
        for job in range(2):
            got = F.cluster_keys(raw, key_len=L, max_distance=d, method="directional", context=ctx)
            assert (got.n_unique, got.n_clusters, got.n_kept) == (want["n_unique"], want["n_clusters"],
                                                                  len(want["kept_read_ids"])), (d, job, got.route)
            assert np.array_equal(got.kept_read_ids, want["kept_read_ids"]), (d, job)
        assert got.route["search_tiles"], got.route


@pytest.mark.parametrize("case", ["plain", "n_keys", "two_planes", "rows_over_150", "probe_overflow", "edge_overflow",
                                  "no_patience", "small_grid"])
def test_one_kernel_collapse_equals_the_two_kernels(F, oracle, monkeypatch, case):
    """FQD_ONE_KERNEL_COLLAPSE=1: dedupe + compaction + search pass 0 of the compact records as ONE persistent kernel
    (collapse_lds.hip bucket_collapse12_kernel; off by default -- it is slower than the two kernels, DESIGN "the one-kernel
    collapse") must give what the oracle gives, also where pass 0 gives way (a bucket with more rows than pass 0 takes, a
    full probe list, an edge list too short) and when its waits run into their limit at once (the host then runs the two
    kernels). What it replaces: `TrieNode_AddSequence` counting (`_triemodule.c:222-288`)."""
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    monkeypatch.setenv("FQD_FUSED_MIN_READS", "100000")
    monkeypatch.setenv("FQD_ONE_KERNEL_COLLAPSE", "1")
    n, L, d = 400_000, 32, 1
    rng = np.random.default_rng(23)
    # (n_keys: 1 % of the keys with an N -- the side slabs hold 1.5 %; more and the job takes uint4 records)
    keys = synth_keys(n, L, 12, 231, sub_rate=3e-3, n_rate=0 if case == "two_planes" else 3e-4 if case == "n_keys" else 1e-4)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    if case == "rows_over_150":
        monkeypatch.setenv("FQD_P0_MAX_ROWS", "150")
    if case == "probe_overflow":
        rows = rng.choice(n, size=90, replace=False)
        keys[rows] = keys[rows[0]]
        keys[rows, 16 + (np.arange(90) % 16)] = ord("N")
        keys[rows, 31 - (np.arange(90) % 15)] = acgt[(np.arange(90) // 15) % 4]
    if case == "edge_overflow":
        monkeypatch.setenv("FQD_P0_EDGE_CAP", "2000")
    if case == "no_patience":
        monkeypatch.setenv("FQD_ONE_KERNEL_WAIT_TICKS", "0")
    if case == "small_grid":            # (many rounds per workgroup, teams of 4)
        monkeypatch.setenv("FQD_ONE_KERNEL_GRID", "64")
        monkeypatch.setenv("FQD_ONE_KERNEL_TEAM", "4")
    raw = np.ascontiguousarray(keys).reshape(-1)
    want = oracle.dedup(raw, fixed_offsets(n, L), max_distance=d, method="directional")
    ctx = F.Context(0)
    if case == "two_planes":
        present = np.zeros(128, dtype=np.uint8)
        present[[ord(ch) for ch in "ACGT"]] = 1
        ctx.configure(present, L, False)
    for job in range(2):
        got = F.cluster_keys(raw, key_len=L, max_distance=d, method="directional", context=ctx)
        assert (got.n_unique, got.n_clusters, got.n_kept) == (want["n_unique"], want["n_clusters"],
                                                              len(want["kept_read_ids"])), (case, job)
        assert np.array_equal(got.kept_read_ids, want["kept_read_ids"]), (case, job)
        r = got.route
        assert r["fused_pack"] and r["compact_records"], (case, r)
        if case != "no_patience":       # (there the kernel may or may not have had to wait)
            assert r["one_kernel_collapse"], (case, r)
        if case in ("plain", "n_keys", "two_planes", "small_grid"):
            assert r["pass0_in_collapse"] and r["pass0_continued"] and not r["restarted"], (case, r)


@pytest.mark.parametrize("case", ["plain", "n_keys", "d2", "two_planes", "rows_over_512", "table_overflow",
                                  "probe_overflow", "edge_overflow", "ladder"])
def test_routed_collapse_does_search_pass_0(F, oracle, monkeypatch, case):
    """The routed collapse (fqd::Pass0): for a Hamming search behind fqd_cluster_keys the reads are binned by segment 0
    of the key, and the compaction of a bucket reports the pairs of search pass 0 itself. Against the oracle, with the
    route the job took (Context.route) asserted -- also where the shortcut must give way: a bucket with more rows than
    a wave's LDS holds or a full probe list of keys with an N (the search then runs pass 0 too), a bucket with more
    keys than the dedupe's LDS table (the job starts over with whole-key hashing, and the context stays with it), an
    edge list too short for pass 0's pairs (the search grows it and runs every pass)."""
    from fastqdedup_amd.synth import SKEW, fixed_offsets, synth_keys
    monkeypatch.setenv("FQD_FUSED_MIN_READS", "100000")
    n, L, d = 300_000, 32, 2 if case == "d2" else 1
    rng = np.random.default_rng(17)
    keys = synth_keys(n, L, 12, 171, sub_rate=3e-3, n_rate=0 if case == "two_planes" else 1e-3 if case == "n_keys" else 1e-4,
                      skew=SKEW if case == "ladder" else None)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    if case == "rows_over_512":
        monkeypatch.setenv("FQD_P0_MAX_ROWS", "150")      # (as if a wave held 150 rows: most buckets have more)
    if case == "table_overflow":
        # 5 000 keys sharing segment 0 (their first 16 bases) and nothing else: more than the dedupe's LDS table holds
        m = 5000
        rows = rng.choice(n, size=m, replace=False)
        keys[rows, :16] = keys[rows[0], :16]
        keys[rows, 16:] = acgt[rng.integers(0, 4, size=(m, 16))]
    if case == "probe_overflow":
        # 90 keys with an N in segment 1 that share segment 0: more than a bucket's probe list holds
        rows = rng.choice(n, size=90, replace=False)
        keys[rows] = keys[rows[0]]
        keys[rows, 16 + (np.arange(90) % 16)] = ord("N")
        keys[rows, 31 - (np.arange(90) % 15)] = acgt[(np.arange(90) // 15) % 4]
    if case == "edge_overflow":
        monkeypatch.setenv("FQD_P0_EDGE_CAP", "2000")
    raw = np.ascontiguousarray(keys).reshape(-1)
    want = oracle.dedup(raw, fixed_offsets(n, L), max_distance=d, method="directional")
    ctx = F.Context(0)
    if case == "two_planes":
        present = np.zeros(128, dtype=np.uint8)
        present[[ord(ch) for ch in "ACGT"]] = 1
        ctx.configure(present, L, False)
    for job in range(2):
        got = F.cluster_keys(raw, key_len=L, max_distance=d, method="directional", context=ctx)
        assert (got.n_unique, got.n_clusters, got.n_kept) == (want["n_unique"], want["n_clusters"],
                                                              len(want["kept_read_ids"])), (case, job)
        assert np.array_equal(got.kept_read_ids, want["kept_read_ids"]), (case, job)
        r = got.route
        if case == "table_overflow":
            # job 0: the routed attempt overflows a bucket's table and slab -- once more with whole-key hashing and the
            # spill list (a full slab looks like a key with very many copies); job 1 starts that way
            assert not r["restarted"] and not r["pass0_in_collapse"] and not r["pass0_continued"]
            assert r["fused_pack"] and r["spill_list"]
        elif case == "probe_overflow":
            assert r["fused_pack"] and r["compact_records"] and not r["pass0_in_collapse"] and not r["pass0_continued"]
        elif case in ("edge_overflow", "rows_over_512"):     # (found out by the search: it runs again, every pass)
            assert r["pass0_in_collapse"] and r["search_retried"] and not r["pass0_continued"]
        elif case == "ladder":          # (the skewed model's hot key overfills a slab: with the spill list from then on)
            assert r["fused_pack"] and r["compact_records"] and r["spill_list"] and not r["restarted"], r
            assert r["search_refined"] or r["search_tiles"], r       # (the ladder shares one segment-0 value: matched on finer segments, or all pairs in tiles)
        else:
            assert r["fused_pack"] and r["compact_records"] and r["pass0_in_collapse"] and r["pass0_continued"], (case, r)
            assert r["search_grouped"] and not r["restarted"]
    # the same keys the stage-by-stage way, on the same context
    ctx.pack_keys(raw, None, L)
    two = ctx.cluster(None, None, max_distance=d, method=2)
    assert (two["n_unique"], two["n_edges"], two["n_kept"]) == (got.n_unique, got.n_edges, got.n_kept)


@pytest.mark.parametrize("case", ["one_hot_key", "zipf", "hot_all_t", "hot_with_n", "weights", "spill_full"])
def test_spill_list_takes_keys_with_many_copies(F, oracle, monkeypatch, case):
    """Keys with hundreds of copies -- or a share of ALL reads -- overfill the slabs of the fused collapse. The first
    such job of a context runs again with the SPILL LIST (records that find a slab full are collapsed through the side
    path's table, and the dedupe merges its rows into that), later jobs start that way. Against the oracle, with the
    route asserted; a full spill list sends the job the stage-by-stage way."""
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    monkeypatch.setenv("FQD_FUSED_MIN_READS", "100000")
    n, L = 400_000, 32
    rng = np.random.default_rng(23)
    keys = synth_keys(n, L, 12, 99, sub_rate=3e-3, n_rate=1e-3)
    weights = None
    if case in ("one_hot_key", "weights", "spill_full"):
        share = 0.6 if case == "spill_full" else 0.1            # (the spill list holds an eighth of the reads + 65 536)
        rows = rng.choice(n, size=int(n * share), replace=False)
        keys[rows] = keys[rows[0]]
        near = rows[: len(rows) // 50]                          # ... and its one-error cloud
        keys[near, rng.integers(0, L, size=len(near))] = ord("A")
    if case == "zipf":
        # copies ~ 1 / rank: a few keys with thousands of copies, hundreds with more than a slab holds
        # (sources without an N: thousands of copies of a key with an N fill the side slabs -- case hot_with_n)
        src = rng.choice(np.flatnonzero(~(keys == ord("N")).any(axis=1)), size=4000, replace=False)
        p = 1.0 / np.arange(1, 4001)
        pick = rng.choice(4000, size=n // 4, p=p / p.sum())
        rows = rng.choice(n, size=n // 4, replace=False)
        keys[rows] = keys[src[pick]]
    if case == "hot_all_t":
        rows = rng.choice(n, size=n // 10, replace=False)
        keys[rows] = ord("T")                                   # (the all-ones pattern of the LDS tables)
    if case == "hot_with_n":
        rows = rng.choice(n, size=n // 10, replace=False)
        keys[rows] = keys[rows[0]]
        keys[rows, 5] = ord("N")
    if case == "weights":
        weights = rng.integers(0, 3, size=n).astype(np.uint32)
    raw = np.ascontiguousarray(keys).reshape(-1)
    want = oracle.dedup(raw, fixed_offsets(n, L), max_distance=1, method="directional", weights=weights)
    ctx = F.Context(0)
    for job in range(2):
        got = F.cluster_keys(raw, key_len=L, max_distance=1, method="directional", weights=weights, context=ctx)
        assert (got.n_unique, got.n_clusters, got.n_kept) == (want["n_unique"], want["n_clusters"],
                                                              len(want["kept_read_ids"])), (case, job)
        assert np.array_equal(got.kept_read_ids, want["kept_read_ids"]), (case, job)
        r = got.route
        if case == "spill_full":
            assert r["restarted"] == (job == 0) and not r["spill_list"] and r["fused_pack"] == False, r
        elif case == "hot_with_n":
            pass        # (keys with an N fill the side slabs: uint4 records, stage by stage -- just the result counts)
        else:
            assert r["fused_pack"] and r["compact_records"] and r["spill_list"] and not r["restarted"], (case, job, r)


@pytest.mark.parametrize("L,n_rate,d", [(1, 0.0, 0), (7, 1e-3, 1), (16, 0.05, 1), (17, 1e-4, 2), (31, 1e-3, 1),
                                        (32, 0.003, 2), (24, 0.0, 1)])
def test_compact_records_over_lengths_and_n_rates(F, oracle, monkeypatch, L, n_rate, d):
    """The 12-byte records of the fused collapse (two key words; keys with an N through the side path) against the
    oracle and against the uint4 records, over key lengths 1..32, N rates from none to so many that the side slabs
    overflow (uint4 records then), and distances 0..2."""
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    monkeypatch.setenv("FQD_FUSED_MIN_READS", "50000")
    n = 260_000                   # (the fused way needs two partition levels: more than 204 800 reads)
    keys = synth_keys(n, L, min(L, 12), 300 + L, sub_rate=4e-3, n_rate=n_rate)
    raw = np.ascontiguousarray(keys).reshape(-1)
    want = oracle.dedup(raw, fixed_offsets(n, L), max_distance=d, method="directional")
    ctx = F.Context(0)
    got = F.cluster_keys(raw, key_len=L, max_distance=d, method="directional", context=ctx)
    kt = ctx.kernel_times(reset=True)
    assert kt["pack_kernel"][1] >= 1
    assert kt["part_scatter12_kernel"][1] >= 1                                        # 12-byte records (tried first)
    if L >= 16:    # (very short keys -- 5 or 16 384 distinct ones -- overfill their slabs: once more with the spill
                   # list, and if that is full as well the plain way takes over)
        assert kt["part_scatter12_kernel"][1] == 1
        assert kt["part_scatter_kernel<1>"][1] == 0                                   # ... the fused way in
    assert (got.n_unique, got.n_clusters) == (want["n_unique"], want["n_clusters"])
    assert np.array_equal(got.kept_read_ids, want["kept_read_ids"])
    monkeypatch.setenv("FQD_NO_COMPACT_RECORDS", "1")
    other = F.cluster_keys(raw, key_len=L, max_distance=d, method="directional", context=F.Context(0))
    assert (other.n_unique, other.n_edges, other.n_kept) == (got.n_unique, got.n_edges, got.n_kept)
    assert np.array_equal(other.kept_read_ids, got.kept_read_ids)


@pytest.mark.parametrize("case", ["plain", "weights", "tag_collisions", "heavy_key", "few_buckets", "len300_d2",
                                  "slices", "slices_weights", "slices_tag_collisions", "slices_heavy_key"])
def test_long_record_collapse_without_sort_matches_oracle(F, oracle, monkeypatch, case):
    """Keys above 32 nt (records longer than one uint4) collapse through (hash, position) pairs: the
    pairs are partitioned, an LDS table per bucket matches them, records are compared where they
    lie (collapse_pairs.hip). Same answer as the oracle and as the sort-based collapse -- with
    weights, with different keys sharing a tag (masked tags), with a key whose copies overfill a
    slab (exact bucket sizes then), and when a bucket overflows the table (sort-based path then). `slices*`: every
    bucket cut into slices of 64 pairs, a workgroup each, their rows joined by `pairs_merge_kernel` (what a bucket of
    more than 4096 pairs gets by default: `heavy_key`'s 5000 copies are two slices)."""
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    n, L, d = (120_000, 100, 1) if case != "len300_d2" else (70_000, 300, 2)
    if case.startswith("slices"):
        monkeypatch.setenv("FQD_PAIRS_SLICE", "64")
        case = case[7:] or "plain"
    if case == "heavy_key":
        n = 300_000                      # two partition levels: the second one works on slabs
    keys = synth_keys(n, L, 12, 61, sub_rate=2e-3, n_rate=2e-4)
    rng = np.random.default_rng(6)
    weights = None
    if case == "weights":
        weights = rng.choice(np.array([0, 1, 1, 1], dtype=np.uint32), size=n)
    if case == "heavy_key":
        heavy = rng.choice(n, size=5000, replace=False)
        keys[heavy] = keys[heavy[0]]
        keys[heavy[::13], 40] = ord("N")
    if case == "tag_collisions":
        monkeypatch.setenv("FQD_PAIRS_TAG_MASK", "0x7")
    if case == "few_buckets":
        monkeypatch.setenv("FQD_LDS_BUCKET_BITS", "2")
    raw = np.ascontiguousarray(keys).reshape(-1)
    want = oracle.dedup(raw, fixed_offsets(n, L), weights=weights, max_distance=d, method="directional")
    ctx = F.Context(0)
    monkeypatch.setenv("FQD_COLLAPSE", "pairs")
    for _ in range(2):
        ctx.kernel_times(reset=True)
        got = F.cluster_keys(raw, key_len=L, weights=weights, max_distance=d, method="directional", context=ctx)
        kt = ctx.kernel_times(reset=True)
        assert kt["bucket_dedupe_kernel"][1] >= 1                       # the pairs path ran ...
        assert (kt["head_flags_kernel"][1] > 0) == (case == "few_buckets")   # ... and only then the sort path
        assert got.n_counted == (n if weights is None else int(weights.sum()))
        assert (got.n_unique, got.n_clusters) == (want["n_unique"], want["n_clusters"])
        assert np.array_equal(got.kept_read_ids, want["kept_read_ids"])
    monkeypatch.setenv("FQD_COLLAPSE", "sort")
    by_sort = F.cluster_keys(raw, key_len=L, weights=weights, max_distance=d, method="directional", context=F.Context(0))
    assert (by_sort.n_unique, by_sort.n_edges) == (got.n_unique, got.n_edges)
    assert np.array_equal(by_sort.kept_read_ids, got.kept_read_ids)


def test_long_record_compaction_queued_before_the_count_is_known(F, oracle, monkeypatch):
    """collapse_pairs queues the compaction behind its read-back WITHOUT waiting for the number of unique keys when the
    unique table of the context's last job has room for as many keys as that job had; a job with MORE unique keys than
    fit must come out right all the same (the queued launch writes nothing, the host launches again), and so must the
    jobs after it. Three jobs on one context: few unique keys, many, few again; and once with the wait as before."""
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    monkeypatch.setenv("FQD_COLLAPSE", "pairs")
    L = 100
    jobs = [synth_keys(150_000, L, 12, 301, copies=8, sub_rate=1e-3, n_rate=1e-4),       # ~19 K molecules
            synth_keys(150_000, L, L, 302, copies=1, sub_rate=2e-3, n_rate=1e-4),        # every read its own key
            synth_keys(120_000, L, 12, 303, copies=4, sub_rate=2e-3, n_rate=1e-4)]
    want = [oracle.dedup(np.ascontiguousarray(k).reshape(-1), fixed_offsets(k.shape[0], L), max_distance=1,
                         method="directional") for k in jobs]
    for wait_first in (False, True):
        if wait_first:
            monkeypatch.setenv("FQD_NO_OPTIMISTIC_COMPACT", "1")
        ctx = F.Context(0)
        for rounds in range(2):
            for k, w in zip(jobs, want):
                got = F.cluster_keys(np.ascontiguousarray(k).reshape(-1), key_len=L, max_distance=1, method="directional",
                                     context=ctx)
                assert (got.n_unique, got.n_clusters, got.n_kept) == (w["n_unique"], w["n_clusters"], len(w["kept_read_ids"]))
                assert np.array_equal(got.kept_read_ids, w["kept_read_ids"]), (wait_first, rounds)
                assert got.route["collapse_pairs"], got.route


def test_every_fast_path_agrees_with_the_plain_paths(F, monkeypatch):
    """2 M reads through the default route (pack fused with level 1, slabs at every partition level,
    segment hashes written by the compaction, kept ids through id bins) and through every
    alternative the switches select: same counters, same kept ids. (The oracle pins the default
    route at sizes it finishes in seconds; this ties the other routes to it at a size where every
    one of them really runs.)"""
    import torch
    from fastqdedup_amd.synth import synth_keys
    monkeypatch.setenv("FQD_FUSED_MIN_READS", "1000000")
    n, L = 2_000_000, 32
    keys = torch.from_numpy(np.ascontiguousarray(synth_keys(n, L, L, 2024, sub_rate=2e-3, n_rate=2e-4)).reshape(-1)).cuda()

    def run(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        try:
            ctx = F.Context(0)
            ctx.kernel_times(reset=True)
            r = F.cluster_keys(keys, key_len=L, max_distance=1, method="directional", context=ctx)
            return r, ctx.kernel_times(reset=True)
        finally:
            for k in env:
                monkeypatch.delenv(k)

    base, kt = run()
    assert kt["part_scatter_kernel<1>"][1] == 0 and kt["segment_hashes_kernel"][1] == 0   # fused pack, early hashes
    assert base.n_kept == len(base.kept_read_ids) and np.all(np.diff(base.kept_read_ids.astype(np.int64)) > 0)
    for env in ({"FQD_NO_FUSED_PACK": "1"}, {"FQD_NO_EARLY_SEG_HASHES": "1"}, {"FQD_KEPT_BY_MAP": "1"},
                {"FQD_KEPT_BY_SORT": "1"}, {"FQD_LDS_NO_SLABS": "1", "FQD_GROUP_NO_SLABS": "1", "FQD_NO_FUSED_PACK": "1"},
                {"FQD_COLLAPSE": "sort", "FQD_EDGES": "sort"}, {"FQD_DIRECTIONAL_ROUNDS": "1"},
                {"FQD_NO_COMPACT_RECORDS": "1"}, {"FQD_GROUP_L1_SLABS_MIN_TILES": "1"}):
        other, okt = run(**env)
        if "FQD_GROUP_L1_SLABS_MIN_TILES" in env:      # level 1 of the search partition in slab mode: no histogram pass
            assert okt["gp_hist_kernel"][1] == 0 and kt["gp_hist_kernel"][1] >= 1
        assert (other.n_unique, other.n_edges, other.n_clusters, other.n_kept) == \
               (base.n_unique, base.n_edges, base.n_clusters, base.n_kept), env
        assert np.array_equal(other.kept_read_ids, base.kept_read_ids), env


def test_edit_d1_equal_length_matches_oracle(F, ctx, oracle):
    """Levenshtein <= 1 on equal-length keys == Hamming <= 1 (BASELINE config 5 shape)."""
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    n, L = 40000, 300
    keys = synth_keys(n, L, L, 77, sub_rate=1e-3).reshape(-1)
    got = F.cluster_keys(keys, key_len=L, max_distance=1, use_edit_distance=True, method="adjacency",
                         context=ctx)
    want = oracle.dedup(keys, fixed_offsets(n, L), max_distance=1, use_edit_distance=True,
                        method="adjacency")
    assert np.array_equal(got.kept_read_ids, want["kept_read_ids"])


def test_edit_metric_mixed_lengths_matches_oracle(F, ctx, oracle):
    """The bucketed Levenshtein search: indels, several length classes, d up to 3."""
    import random
    rng = random.Random(11)
    mols = ["".join(rng.choice("ACGT") for _ in range(rng.randint(14, 40))) for _ in range(600)]
    keys = []
    for _ in range(6000):
        s = list(rng.choice(mols))
        for _ in range(rng.choice([0, 0, 1, 1, 2])):
            op = rng.random()
            pos = rng.randrange(len(s) + 1)
            if op < 0.4 and s:
                s[min(pos, len(s) - 1)] = rng.choice("ACGTN")
            elif op < 0.7 and len(s) > 1:
                del s[min(pos, len(s) - 1)]
            else:
                s.insert(pos, rng.choice("ACGT"))
        keys.append("".join(s))
    raw, off = _pack(keys)
    for d in (1, 2, 3):
        for m in METHODS:
            got = F.cluster_keys(raw, off, max_distance=d, use_edit_distance=True, method=m, context=ctx)
            want = oracle.dedup(raw, off, max_distance=d, use_edit_distance=True, method=m)
            assert got.n_unique == want["n_unique"]
            assert got.n_clusters == want["n_clusters"], (d, m)
            assert np.array_equal(got.kept_read_ids, want["kept_read_ids"]), (d, m)


def test_edit_d2_equal_length_matches_oracle(F, ctx, oracle):
    """Equal lengths but d=2: one insertion + one deletion is not a Hamming neighbour."""
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    n, L = 20000, 40
    keys = synth_keys(n, L, 8, 5, sub_rate=4e-3, n_rate=1e-3)
    shifted = np.roll(keys[:2000], 1, axis=1)          # rotate: 1 ins + 1 del away from the original
    keys = np.concatenate([keys, shifted]).reshape(-1)
    n += 2000
    got = F.cluster_keys(keys, key_len=L, max_distance=2, use_edit_distance=True, method="directional",
                         context=ctx)
    want = oracle.dedup(keys, fixed_offsets(n, L), max_distance=2, use_edit_distance=True,
                        method="directional")
    ham = F.cluster_keys(keys, key_len=L, max_distance=2, method="directional", context=ctx)
    assert got.n_clusters == want["n_clusters"] < ham.n_clusters
    assert np.array_equal(got.kept_read_ids, want["kept_read_ids"])


def test_contains_sequence_fuzz(F, oracle):
    import random
    rng = random.Random(2)
    for _ in range(40):
        syms = rng.choice(["AC", "ACGT", "ACGTN", "abcXN"])
        a, b = F.Trie(), oracle.Trie()
        for _ in range(rng.randint(1, 30)):
            s = "".join(rng.choice(syms) for _ in range(rng.randint(0, 7)))
            a.add_sequence(s)
            b.add_sequence(s)
        for _ in range(8):
            q = "".join(rng.choice(syms + "U") for _ in range(rng.randint(0, 8)))
            d, edit = rng.randint(0, 3), rng.random() < 0.5
            assert a.contains_sequence(q, d, edit) == b.contains_sequence(q, d, edit), (q, d, edit)


def test_ragged_and_foreign_alphabet(F, ctx, oracle):
    import random
    rng = random.Random(3)
    mols = ["".join(rng.choice("acgtXN-") for _ in range(rng.randint(0, 40))) for _ in range(400)]
    keys = []
    for _ in range(5000):
        s = list(rng.choice(mols))
        if s and rng.random() < 0.3:
            s[rng.randrange(len(s))] = rng.choice("acgtXN-")
        keys.append("".join(s))
    raw, off = _pack(keys)
    w = np.array([rng.choice([0, 1, 1, 1, 3]) for _ in keys], dtype=np.uint32)
    for d in (0, 1, 2):
        for m in METHODS:
            got = F.cluster_keys(raw, off, weights=w, max_distance=d, method=m, context=ctx)
            want = oracle.dedup(raw, off, w, max_distance=d, method=m)
            assert got.n_unique == want["n_unique"]
            assert got.n_clusters == want["n_clusters"]
            assert np.array_equal(got.kept_read_ids, want["kept_read_ids"]), (d, m)
    sh = ctx.shape()
    assert bytes(sh.alphabet[: sh.alphabet_size]).decode() == "".join(sorted(set("".join(keys))))


def test_hash_collisions_do_not_merge_keys(F, oracle, monkeypatch):
    """Narrow the collapse hash to 5 bits: thousands of distinct keys share a hash
    and must still come out as distinct keys."""
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    monkeypatch.setenv("FQD_HASH_BITS", "5")
    n, L = 6000, 20
    keys = synth_keys(n, L, 6, 9, sub_rate=0.01, n_rate=0.002).reshape(-1)
    ctx = F.Context(0)
    got = F.cluster_keys(keys, key_len=L, max_distance=1, method="directional", context=ctx)
    want = oracle.dedup(keys, fixed_offsets(n, L), max_distance=1, method="directional")
    assert got.n_unique == want["n_unique"]
    assert np.array_equal(got.kept_read_ids, want["kept_read_ids"])


@pytest.mark.parametrize("path,bucket_bits", [("lds", None), ("sort", None), ("lds", "1")])
def test_collapse_paths_agree_with_oracle(F, oracle, monkeypatch, path, bucket_bits):
    """Both collapse implementations (LDS bucket dedupe / sort + verify), and the overflow
    fallback from the first to the second, give the oracle's answer: weights incl. 0, caller
    read ids, one key with 150 k copies, many singletons."""
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    monkeypatch.setenv("FQD_COLLAPSE", path)
    if bucket_bits:
        monkeypatch.setenv("FQD_LDS_BUCKET_BITS", bucket_bits)
    rng = np.random.default_rng(12)
    L = 32
    a = synth_keys(120_000, L, L, 77, sub_rate=2e-3, n_rate=2e-4)
    heavy = np.tile(np.frombuffer(b"ACGTTGCAACGTTGCAACGTTGCAACGTTGCA", dtype=np.uint8), (150_000, 1))
    singles = np.frombuffer(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=80_000 * L).tobytes(),
                            dtype=np.uint8).reshape(-1, L)
    keys = np.concatenate([a, heavy, singles])
    keys = keys[rng.permutation(len(keys))].reshape(-1)
    n = len(keys) // L
    w = rng.choice(np.array([0, 1, 1, 1, 3], dtype=np.uint32), size=n)
    ctx = F.Context(0)
    got = F.cluster_keys(keys, key_len=L, weights=w, max_distance=1, method="directional", context=ctx)
    want = oracle.dedup(keys, fixed_offsets(n, L), w, max_distance=1, method="directional")
    assert got.n_unique == want["n_unique"] and got.n_clusters == want["n_clusters"]
    assert got.n_counted == int(w.sum())
    assert np.array_equal(got.kept_read_ids, want["kept_read_ids"])
    ids = (np.arange(n, dtype=np.uint64) * 3 + 11)
    got2 = F.cluster_keys(keys, key_len=L, weights=w, read_ids=ids, max_distance=1, method="adjacency",
                          context=ctx)
    want2 = oracle.dedup(keys, fixed_offsets(n, L), w, max_distance=1, method="adjacency")
    assert np.array_equal(got2.kept_read_ids, want2["kept_read_ids"] * 3 + 11)


def test_random_small_inputs_sweep(F, ctx, oracle):
    """600 tiny random jobs: alphabets of 1-6 symbols, key lengths 0-12 (ragged or fixed),
    weights incl. 0, d 0-3, both metrics, all methods -- kept ids, cluster and unique counts."""
    import random
    rng = random.Random(2026)
    for trial in range(600):
        syms = rng.choice(["A", "AC", "ACG", "ACGT", "ACGTN", "ACGTNX", "acgt#-"])
        lo, hi = rng.choice([(0, 4), (3, 3), (5, 9), (12, 12), (0, 12)])
        n = rng.randint(1, 60)
        pool = ["".join(rng.choice(syms) for _ in range(rng.randint(lo, hi))) for _ in range(rng.randint(1, 12))]
        keys = []
        for _ in range(n):
            s = list(rng.choice(pool))
            if s and rng.random() < 0.5:
                s[rng.randrange(len(s))] = rng.choice(syms)
            if rng.random() < 0.15 and lo != hi:
                s = s[:-1] if s and rng.random() < 0.5 else s + [rng.choice(syms)]
            keys.append("".join(s))
        raw, off = _pack(keys)
        w = np.array([rng.choice([0, 1, 1, 2, 5]) for _ in keys], dtype=np.uint32)
        d, edit, m = rng.randint(0, 3), rng.random() < 0.5, rng.choice(METHODS)
        got = F.cluster_keys(raw, off, weights=w, max_distance=d, use_edit_distance=edit, method=m, context=ctx)
        want = oracle.dedup(raw, off, w, max_distance=d, use_edit_distance=edit, method=m)
        ctxt = (trial, keys, w.tolist(), d, edit, m)
        assert got.n_unique == want["n_unique"], ctxt
        assert got.n_clusters == want["n_clusters"], ctxt
        assert got.kept_read_ids.tolist() == want["kept_read_ids"].tolist(), ctxt


def test_trie_interleaved_add_pop_matches_oracle_trie(F, oracle):
    """The drop-in Trie against the oracle's trie (which equals the reference's call for call):
    interleaved add_sequence / pop_cluster, both metrics; clusters compared as sets AND in
    emission order (ascending seed in trie-alphabet order, longer key before its prefix)."""
    import random
    rng = random.Random(77)
    for trial in range(60):
        a, b = F.Trie("ACGTN"), oracle.Trie("ACGTN")
        d, edit = rng.randint(0, 2), rng.random() < 0.5
        pool = ["".join(rng.choice("ACGTN") for _ in range(rng.randint(2, 7))) for _ in range(10)]
        for _ in range(rng.randint(1, 40)):
            s = list(rng.choice(pool))
            if rng.random() < 0.4:
                s[rng.randrange(len(s))] = rng.choice("ACGT")
            s = "".join(s)
            a.add_sequence(s)
            b.add_sequence(s)
        assert a.number_of_sequences == b.number_of_sequences
        while b.number_of_sequences:
            ca, cb = a.pop_cluster(d, edit), b.pop_cluster(d, edit)
            assert sorted(ca) == sorted(cb), (trial, d, edit)
            assert ca[0] == cb[0]                       # same seed first
            assert a.number_of_sequences == b.number_of_sequences
            if rng.random() < 0.25:                     # add between pops: re-clustered on the device
                s = "".join(rng.choice("ACGT") for _ in range(rng.randint(2, 7)))
                a.add_sequence(s)
                b.add_sequence(s)
        with pytest.raises(LookupError):
            a.pop_cluster(d, edit)


def test_edge_cases(F, ctx):
    empty = F.cluster_keys(np.zeros(0, np.uint8), np.zeros(1, np.uint64), context=ctx)
    assert (empty.n_reads, empty.n_unique, empty.n_clusters, empty.n_kept) == (0, 0, 0, 0)
    one = F.cluster_keys(np.frombuffer(b"ACGT", np.uint8), key_len=4, context=ctx)
    assert one.kept_read_ids.tolist() == [0] and one.n_clusters == 1
    same = F.cluster_keys(np.frombuffer(b"ACGT" * 1000, np.uint8), key_len=4, context=ctx)
    assert same.kept_read_ids.tolist() == [0] and same.n_unique == 1 and same.n_counted == 1000
    raw, off = _pack(["", "", "A", ""])
    e = F.cluster_keys(raw, off, max_distance=1, context=ctx)
    assert e.n_unique == 2 and e.n_clusters == 2 and e.kept_read_ids.tolist() == [0, 2]
    with pytest.raises(ValueError):
        F.cluster_keys(np.array([65, 200, 67], np.uint8), key_len=3, context=ctx)
    with pytest.raises(ValueError):
        F.cluster_keys(np.frombuffer(b"ACGT", np.uint8), key_len=4, max_distance=-1, context=ctx)
    # d larger than the key: every pair of equal-length keys is adjacent
    raw, off = _pack(["AC", "GT", "TT", "ACG"])
    big = F.cluster_keys(raw, off, max_distance=5, method="highest_count", context=ctx)
    assert big.n_clusters == 2


def test_device_resident_input_and_synth_twin(F, ctx):
    import torch
    from fastqdedup_amd.synth import synth_keys
    n, L, umi, seed = 50000, 100, 12, 1002
    host = synth_keys(n, L, umi, seed)
    dev = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
    ctx.synth_keys(dev, n, 0, n, L, umi, seed)
    assert np.array_equal(dev.cpu().numpy().reshape(n, L), host), "HIP and numpy generators differ"
    a = F.cluster_keys(dev, key_len=L, context=ctx)
    b = F.cluster_keys(host.reshape(-1), key_len=L, context=ctx)
    assert np.array_equal(a.kept_read_ids, b.kept_read_ids)


@pytest.mark.parametrize("edit,d", [(False, 1), (False, 2), (True, 2)])
def test_bucket_shards_union_to_the_whole_search(F, oracle, edit, d):
    """What rank r of a G-rank job does in stage 3 (search only the buckets with hash % G == r),
    done here for r = 0..2 on one GPU: the union of the edge shards must give the oracle's result."""
    import random
    rng = random.Random(31)
    if edit:
        mols = ["".join(rng.choice("ACGT") for _ in range(rng.randint(18, 30))) for _ in range(500)]
        keys = []
        for _ in range(5000):
            s = list(rng.choice(mols))
            for _ in range(rng.choice([0, 1, 1, 2])):
                pos = rng.randrange(len(s))
                r = rng.random()
                if r < 0.4:
                    s[pos] = rng.choice("ACGTN")
                elif r < 0.7 and len(s) > 2:
                    del s[pos]
                else:
                    s.insert(pos, rng.choice("ACGT"))
            keys.append("".join(s))
        raw, off = _pack(keys)
        key_len = 0
    else:
        from fastqdedup_amd.synth import fixed_offsets, synth_keys
        n, key_len = 150_000, 36
        raw = synth_keys(n, key_len, 8, 17, sub_rate=4e-3, n_rate=5e-4).reshape(-1)
        off = fixed_offsets(n, key_len)
    ctx = F.Context(0)
    metric = 1 if edit else 0
    ctx.pack_keys(raw, None if key_len else off, key_len)
    nu = ctx.collapse()
    shards = []
    for r in range(3):
        ne = ctx.find_edges(d, metric, r, 3)
        e = np.empty((ne, 2), dtype=np.uint32)
        ctx.export_edges(e)
        shards.append(e)
    whole = ctx.find_edges(d, metric, 0, 1)
    union = np.concatenate(shards)
    uniq = {(int(a), int(b)) for a, b in union}
    assert len(uniq) == whole                      # nothing lost; Hamming shards are disjoint
    if not edit:
        assert len(union) == whole
    ctx.import_edges(np.ascontiguousarray(union), len(union))
    n_clusters = ctx.components()
    n_kept = ctx.dissect(2)
    kept = ctx.kept_read_ids(n_kept)
    want = oracle.dedup(raw, off, max_distance=d, use_edit_distance=edit, method="directional")
    assert nu == want["n_unique"] and n_clusters == want["n_clusters"]
    assert np.array_equal(kept, want["kept_read_ids"])


@pytest.mark.parametrize("bits,d,L", [(None, 1, 32), ("4", 2, 36), ("11", 1, 100), ("1", 3, 48), (None, 2, 160),
                                      ("6", 2, 300), (None, 3, 200)])
def test_grouped_search_equals_sorted_search(F, oracle, monkeypatch, bits, d, L):
    """The sort-free search pass (group.hip: partition + one wave per bucket) finds exactly the
    edges of the radix-sort pass -- with roomy buckets, with buckets far larger than the LDS slice
    (few bucket bits), single-level and two-level partitions, candidates verified by one thread or
    (records of 64 bytes and more) by several lanes each -- and the oracle's clusters."""
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    n = 60_000
    raw = synth_keys(n, L, 8, 41, sub_rate=6e-3, n_rate=5e-4).reshape(-1)
    found = {}
    ctx = F.Context(0)
    ctx.pack_keys(raw, None, L)
    ctx.collapse()             # ONE unique table (its row order is not reproducible across collapses)
    for mode in ("sort", "grouped"):
        monkeypatch.setenv("FQD_EDGES", mode)
        if bits and mode == "grouped":
            monkeypatch.setenv("FQD_GROUP_BUCKET_BITS", bits)
        ne = ctx.find_edges(d, 0, 0, 1)
        e = np.empty((ne, 2), dtype=np.uint32)
        ctx.export_edges(e)
        found[mode] = sorted(map(tuple, e.tolist()))
        if mode == "grouped":
            n_clusters = ctx.components()
            kept = ctx.kept_read_ids(ctx.dissect(2))
    assert found["sort"] == found["grouped"] and len(set(found["grouped"])) == len(found["grouped"])
    want = oracle.dedup(raw, fixed_offsets(n, L), max_distance=d, method="directional")
    assert n_clusters == want["n_clusters"] and np.array_equal(kept, want["kept_read_ids"])


@pytest.mark.parametrize("budget", [None, "100000"])
def test_grouped_search_with_a_crowded_segment(F, oracle, monkeypatch, budget):
    """3000 keys share their first half: one bucket of the pass over segment 0 holds them all and
    every pair of them is a candidate (4.5 M). The candidate lists grow to hold them; with a small
    budget the search falls back to the sort path instead. Same answer as the oracle either way."""
    from fastqdedup_amd.synth import synth_keys
    rng = np.random.default_rng(5)
    L = 32
    base = synth_keys(40_000, L, L, 43, sub_rate=5e-3, n_rate=3e-4)
    crowd = np.tile(base[0], (3000, 1))
    crowd[:, L // 2:] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(3000, L // 2))
    crowd[1::50, L // 2:] = crowd[0, L // 2:]             # some exact copies and near copies too
    crowd[2::50, L - 1] = ord("N")
    keys = np.concatenate([base, crowd])
    rng.shuffle(keys)
    raw = np.ascontiguousarray(keys).reshape(-1)
    monkeypatch.setenv("FQD_EDGES", "grouped")
    if budget:
        monkeypatch.setenv("FQD_GROUP_CAND_BUDGET", budget)
    res = F.cluster_keys(raw, key_len=L, max_distance=1, method="directional", context=F.Context(0))
    off = np.arange(len(keys) + 1, dtype=np.uint64) * L
    want = oracle.dedup(raw, off, max_distance=1, method="directional")
    assert res.n_unique == want["n_unique"] and res.n_clusters == want["n_clusters"]
    assert np.array_equal(res.kept_read_ids, want["kept_read_ids"])


@pytest.mark.parametrize("level1_slabs", [False, True])
def test_grouped_search_slab_overflow_searches_again(F, oracle, monkeypatch, level1_slabs):
    """Level 2 of the search's (hash, uid) partition also uses fixed slabs (and, for millions of items, level 1 --
    here by a lowered threshold). 3000 keys sharing their
    first half overfill one: the search must run again with exact bucket sizes (and keep doing so on
    this context), with the oracle's answer and the same edges as a context that never used slabs."""
    if level1_slabs:
        monkeypatch.setenv("FQD_GROUP_L1_SLABS_MIN_TILES", "1")
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    rng = np.random.default_rng(15)
    n, L = 300_000, 32
    keys = synth_keys(n, L, L, 47, sub_rate=4e-3, n_rate=2e-4)
    crowd = rng.choice(n, size=3000, replace=False)
    keys[crowd, :L // 2] = keys[crowd[0], :L // 2]
    raw = np.ascontiguousarray(keys).reshape(-1)
    monkeypatch.setenv("FQD_EDGES", "grouped")
    want = oracle.dedup(raw, fixed_offsets(n, L), max_distance=1, method="directional")
    ctx = F.Context(0)
    for _ in range(2):
        got = F.cluster_keys(raw, key_len=L, max_distance=1, method="directional", context=ctx)
        assert (got.n_unique, got.n_clusters) == (want["n_unique"], want["n_clusters"])
        assert np.array_equal(got.kept_read_ids, want["kept_read_ids"])
    monkeypatch.setenv("FQD_GROUP_NO_SLABS", "1")
    exact = F.cluster_keys(raw, key_len=L, max_distance=1, method="directional", context=F.Context(0))
    assert exact.n_edges == got.n_edges
    assert np.array_equal(exact.kept_read_ids, got.kept_read_ids)


@pytest.mark.parametrize("edit", [False, True])
def test_directional_closed_form_equals_rounds(F, oracle, monkeypatch, edit):
    """The directional dissection has a closed form on collapsed tables (two passes over the edges)
    and a relaxation-round form (lists with repeated keys): same verdicts, and the oracle's."""
    import random
    rng = random.Random(77)
    mols = ["".join(rng.choice("ACGT") for _ in range(rng.choice([14, 15, 16]) if edit else 16)) for _ in range(700)]
    keys, weights = [], []
    for _ in range(30_000):
        s = list(rng.choice(mols))
        for _ in range(rng.choice([0, 0, 0, 1, 1, 2])):
            pos = rng.randrange(len(s))
            s[pos] = rng.choice("ACGTN")
        keys.append("".join(s))
        weights.append(rng.choice([0, 1, 1, 1, 2, 5]))
    raw, off = _pack(keys)
    w = np.array(weights, dtype=np.uint32)
    want = oracle.dedup(raw, off, w, max_distance=2 if not edit else 1, use_edit_distance=edit, method="directional")
    got = {}
    for mode in ("closed", "rounds"):
        if mode == "rounds":
            monkeypatch.setenv("FQD_DIRECTIONAL_ROUNDS", "1")
        res = F.cluster_keys(raw, off, weights=w, max_distance=2 if not edit else 1, use_edit_distance=edit,
                             method="directional", context=F.Context(0))
        got[mode] = res.kept_read_ids
        assert res.n_clusters == want["n_clusters"]
    assert np.array_equal(got["closed"], got["rounds"])
    assert np.array_equal(got["closed"], want["kept_read_ids"])


def test_segment_passes_partition_the_search(F, oracle):
    """fqd_find_edges_segments: disjoint pass ranges give every edge of the whole search exactly
    once (a pair is reported in the first segment it agrees on)."""
    from fastqdedup_amd.synth import synth_keys
    n, L, d = 120_000, 40, 2
    raw = synth_keys(n, L, 8, 23, sub_rate=5e-3, n_rate=5e-4).reshape(-1)
    ctx = F.Context(0)
    ctx.pack_keys(raw, None, L)
    ctx.collapse()
    whole_n = ctx.find_edges(d, 0, 0, 1)
    whole = np.empty((whole_n, 2), dtype=np.uint32)
    ctx.export_edges(whole)
    parts = []
    for lo, hi in ((0, 1), (1, 3)):
        ne = ctx.find_edges_segments(d, lo, hi)
        e = np.empty((ne, 2), dtype=np.uint32)
        ctx.export_edges(e)
        parts.append(e)
    union = np.concatenate(parts)
    assert len(union) == whole_n
    assert {(int(a), int(b)) for a, b in union} == {(int(a), int(b)) for a, b in whole}
    with pytest.raises(ValueError):
        ctx.find_edges_segments(d, 2, 4)


def test_grouping_by_segment_owner(F):
    """fqd_export_packed_by_segment / fqd_export_unique_by_segment: a stable split into parts; keys
    that agree on the segment (and all copies of a key) land in the same part; the owners worked
    out inside the pack pass (fqd_set_owner_rule) equal the ones worked out afterwards."""
    import torch
    from fastqdedup_amd.synth import synth_keys
    dev = torch.device("cuda", 0)
    n, L, parts, nseg = 70_001, 32, 5, 2
    host = synth_keys(n, L, 8, 29, sub_rate=5e-3, n_rate=5e-4)
    keys = torch.from_numpy(host.reshape(-1).copy()).to(dev)
    ctx = F.Context(0)

    def grouped(rule_in_pack):
        ctx.set_owner_rule(parts if rule_in_pack else 0, nseg, 0)
        ctx.pack_keys(keys, None, L)
        ctx.set_owner_rule(0)
        stride = ctx.shape().stride_words
        recs = torch.empty((n, stride), dtype=torch.int32, device=dev)
        ids = torch.empty(n, dtype=torch.int64, device=dev)
        counts = ctx.export_packed_by_segment(parts, nseg, 0, 1000, None, recs, None, ids, None)
        return recs.cpu().numpy(), ids.cpu().numpy(), [int(c) for c in counts]

    recs_a, ids_a, counts_a = grouped(True)
    recs_b, ids_b, counts_b = grouped(False)
    assert counts_a == counts_b and sum(counts_a) == n
    assert np.array_equal(ids_a, ids_b) and np.array_equal(recs_a, recs_b)
    bounds = np.concatenate([[0], np.cumsum(counts_a)])
    seen = {}
    for p in range(parts):
        part_ids = ids_a[bounds[p]:bounds[p + 1]] - 1000
        assert np.all(np.diff(part_ids) > 0)                     # stable: read order kept
        for half in {bytes(host[i, :L // nseg]) for i in part_ids[:2000]}:
            assert seen.setdefault(half, p) == p                 # one owner per segment-0 content
    assert sorted(ids_a - 1000) == list(range(n))

    # the unique table, by segment 1, with job-wide ids
    nu = ctx.collapse()
    stride = ctx.shape().stride_words
    urecs = torch.empty((nu, stride), dtype=torch.int32, device=dev)
    uids = torch.empty(nu, dtype=torch.int32, device=dev)
    ucounts = ctx.export_unique_by_segment(parts, nseg, 1, 500, urecs, None, uids)
    assert sum(int(c) for c in ucounts) == nu
    assert sorted(uids.cpu().tolist()) == list(range(500, 500 + nu))
    # rows fetched back by index are the rows that were exported
    rows = (uids - 500).contiguous()
    g_recs = torch.empty_like(urecs)
    g_counts = torch.empty(nu, dtype=torch.int32, device=dev)
    ctx.gather_unique(rows, nu, g_recs, None, g_counts)
    assert torch.equal(g_recs, urecs)
    _first, counts_tbl, _l, _k = ctx.unique_table(nu, labels=False, kept=False)
    assert np.array_equal(g_counts.cpu().numpy().astype(np.uint32), counts_tbl[rows.cpu().numpy()])
    with pytest.raises(ValueError):
        ctx.gather_unique(torch.tensor([nu], dtype=torch.int32, device=dev), 1, g_recs, None, g_counts)


@pytest.mark.parametrize("L,alphabet,path", [(32, "ACGTN", None), (32, "ACGTN", "sort"), (72, "ACGTN", None),
                                             (20, "ACGT", "lds")])
def test_collapse_of_received_reads(F, monkeypatch, L, alphabet, path):
    """fqd_export_packed_by_segment(ids=NULL) + fqd_collapse_received: the sender's read index rides
    in the record's padding word and the receiver adds the sender's id base. Same unique table
    (first ids, counts) as a plain collapse with an explicit id array -- through the LDS collapse,
    the sort-based one (ids extracted, padding cleared), long records, and a geometry whose padding
    starts before the fourth word."""
    import torch
    from fastqdedup_amd.synth import synth_keys
    dev = torch.device("cuda", 0)
    if path:
        monkeypatch.setenv("FQD_COLLAPSE", path)
    nx, ny, base = 45_000, 30_000, 1000
    allk = synth_keys(nx + ny, L, 8, 51, sub_rate=4e-3, n_rate=3e-4 if "N" in alphabet else 0.0)
    ctx = F.Context(0)
    sent = []
    for part in (allk[:nx], allk[nx:]):
        ctx.pack_keys(np.ascontiguousarray(part).reshape(-1), None, L)
        sh = ctx.shape()
        assert sh.stride_words > sh.planes * sh.words
        recs = torch.empty((part.shape[0], sh.stride_words), dtype=torch.int32, device=dev)
        counts = ctx.export_packed_by_segment(1, 2, 0, 0, None, recs, None, None, None)
        assert [int(c) for c in counts] == [part.shape[0]]
        sent.append(recs)
    got_ctx = F.Context(0)
    got_ctx.pack_keys(np.ascontiguousarray(allk[:1]).reshape(-1), None, L)      # same geometry
    received = torch.cat(sent)
    got_ctx.import_packed(received, None, nx + ny, borrow=True)
    nu = got_ctx.collapse_received(None, [0, nx, nx + ny], [base, base + nx], base + nx + ny)
    first, counts, _l, _k = got_ctx.unique_table(nu, labels=False, kept=False)

    want_ctx = F.Context(0)
    want_ctx.pack_keys(np.ascontiguousarray(allk).reshape(-1), None, L)
    nu2 = want_ctx.collapse(None, np.arange(base, base + nx + ny, dtype=np.uint64))
    first2, counts2, _l2, _k2 = want_ctx.unique_table(nu2, labels=False, kept=False)
    assert nu == nu2
    order, order2 = np.argsort(first), np.argsort(first2)
    assert np.array_equal(first[order], first2[order2]) and np.array_equal(counts[order], counts2[order2])
    # and the search still works on that table (padding never takes part in a key)
    assert got_ctx.find_edges(1, 0, 0, 1) == want_ctx.find_edges(1, 0, 0, 1)


def test_fallback_paths_kept_by_sort_and_grouping_by_sort(F, monkeypatch):
    """Two fallbacks that large jobs never take: the kept-id list by scan + gather + radix sort
    (id ranges far larger than the table) and the owner grouping by radix sort (more than 256
    parts). Both must give what the default paths give."""
    import torch
    from fastqdedup_amd.synth import synth_keys
    dev = torch.device("cuda", 0)
    n, L = 80_000, 32
    host = synth_keys(n, L, 8, 61, sub_rate=5e-3, n_rate=3e-4).reshape(-1)
    a = F.cluster_keys(host, key_len=L, context=F.Context(0))
    monkeypatch.setenv("FQD_KEPT_BY_SORT", "1")
    b = F.cluster_keys(host, key_len=L, context=F.Context(0))
    monkeypatch.delenv("FQD_KEPT_BY_SORT")
    assert np.array_equal(a.kept_read_ids, b.kept_read_ids) and a.n_clusters == b.n_clusters

    ctx = F.Context(0)
    ctx.pack_keys(torch.from_numpy(host.copy()).to(dev), None, L)
    stride = ctx.shape().stride_words
    outs = []
    for by_sort in (False, True):
        if by_sort:
            monkeypatch.setenv("FQD_GROUP_BY_SORT", "1")
        recs = torch.empty((n, stride), dtype=torch.int32, device=dev)
        ids = torch.empty(n, dtype=torch.int64, device=dev)
        counts = ctx.export_packed_by_segment(7, 2, 0, 5, None, recs, None, ids, None)
        outs.append((recs.cpu(), ids.cpu(), [int(c) for c in counts]))
    assert outs[0][2] == outs[1][2] and torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_kept_list_written_into_the_callers_buffer(F):
    """cluster_keys(kept_out=device tensor): the ascending kept-id list is produced in the caller's
    buffer (fqd_set_kept_output), same ids as the default; once the buffer is withdrawn the context
    says so instead of handing out a stale list."""
    import torch
    from fastqdedup_amd.synth import synth_keys
    n, L = 100_000, 32
    host = synth_keys(n, L, 8, 71, sub_rate=5e-3, n_rate=3e-4).reshape(-1)
    want = F.cluster_keys(host, key_len=L, context=F.Context(0))
    ctx = F.Context(0)
    dev = torch.from_numpy(host.copy()).to("cuda:0")
    buf = torch.full((n,), -1, dtype=torch.int64, device="cuda:0")
    got = F.cluster_keys(dev, key_len=L, context=ctx, kept_out=buf)
    assert got.n_kept == want.n_kept
    assert np.array_equal(buf[: got.n_kept].cpu().numpy().astype(np.uint64), want.kept_read_ids)
    assert int(buf[got.n_kept].item()) == -1                      # nothing written past the list
    with pytest.raises(RuntimeError):
        ctx.kept_read_ids(got.n_kept)
    small = torch.empty(16, dtype=torch.int64, device="cuda:0")   # too small to be used directly
    ctx.set_kept_output(small)
    try:
        ctx.pack_keys(dev, None, L)
        s = ctx.cluster(None, None, max_distance=1, metric=0, method=2)
        assert np.array_equal(ctx.kept_read_ids(s["n_kept"]), want.kept_read_ids)
    finally:
        ctx.set_kept_output(None)


def test_edge_labels_and_kept_except(F, oracle):
    """fqd_edge_labels (components of a caller's edge list) and fqd_list_kept_except (verdicts
    computed elsewhere) against the plain single-context path."""
    import torch
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    dev = torch.device("cuda", 0)
    n, L, d = 90_000, 32, 1
    raw = synth_keys(n, L, 8, 31, sub_rate=5e-3, n_rate=5e-4).reshape(-1)
    ctx = F.Context(0)
    ctx.pack_keys(raw, None, L)
    nu = ctx.collapse()
    ne = ctx.find_edges(d, 0, 0, 1)
    edges = torch.empty((ne, 2), dtype=torch.int32, device=dev)
    ctx.export_edges(edges)
    n_clusters = ctx.components()
    n_kept = ctx.dissect(2)
    kept_ids = ctx.kept_read_ids(n_kept)
    _first, _counts, labels, kept = ctx.unique_table(nu)

    other = F.Context(0)
    roots = torch.empty(ne, dtype=torch.int32, device=dev)
    assert other.edge_labels(edges, ne, nu, roots) == n_clusters
    e = edges.cpu().numpy()
    assert np.array_equal(roots.cpu().numpy().astype(np.uint32), labels[e[:, 0]])
    with pytest.raises(ValueError):
        other.edge_labels(edges, ne, int(e.max()), roots)          # an end outside [0, n_nodes)

    dropped = torch.from_numpy(np.flatnonzero(kept == 0).astype(np.int32)).to(dev)
    assert ctx.list_kept_except(dropped, dropped.shape[0]) == n_kept
    assert np.array_equal(ctx.kept_read_ids(n_kept), kept_ids)
    flags = torch.empty(nu, dtype=torch.uint8, device=dev)
    ctx.kept_flags_into(flags)
    assert np.array_equal(flags.cpu().numpy(), kept)
    want = oracle.dedup(raw, fixed_offsets(n, L), max_distance=d, method="directional")
    assert np.array_equal(kept_ids, want["kept_read_ids"])


@pytest.mark.parametrize("shortcuts", [True, False])
def test_sharded_path_on_rccl_world1(F, oracle, monkeypatch, shortcuts):
    """The production multi-GPU code path (HipBackend + torch.distributed 'nccl' = RCCL) with a
    one-rank group: every collective, every export/import of the C ABI, against the oracle. With ONE rank
    the collectives are the identity; FQD_COMM_NO_SHORTCUT=1 sends them through RCCL all the same (all-to-all(v),
    all-gather, all-reduce of one rank with itself, ordered against the library's stream by stream waits)."""
    if not shortcuts:
        monkeypatch.setenv("FQD_COMM_NO_SHORTCUT", "1")
    import torch
    import torch.distributed as dist
    from fastqdedup_amd.sharded import HipBackend, cluster_keys_sharded
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29611" if shortcuts else "29612"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ctx = F.Context(0)
        backend = HipBackend(ctx, dev)
        n, L = 120000, 50
        host = synth_keys(n, L, 8, 21, sub_rate=3e-3, n_rate=5e-4).reshape(-1)
        keys = torch.from_numpy(host).to(dev)
        w = np.ones(n, dtype=np.int32)
        w[::7] = 0
        for d, m, plan in ((1, "directional", "segment-routed"), (2, "adjacency", "segment-routed"),
                           (1, "highest_count", "segment-routed"), (0, "directional", "segment-routed"),
                           (1, "directional", "gathered"), (2, "adjacency", "gathered")):
            got = cluster_keys_sharded(backend, keys, None, L, torch.from_numpy(w).to(dev),
                                       max_distance=d, method=m, plan=plan)
            want = oracle.dedup(host, fixed_offsets(n, L), w.astype(np.uint32), max_distance=d, method=m)
            assert got.plan == plan
            assert got.n_unique == want["n_unique"] and got.n_clusters == want["n_clusters"], (d, m, plan)
            assert np.array_equal(got.kept_read_ids.cpu().numpy().astype(np.uint64), want["kept_read_ids"]), (d, m, plan)
            assert got.n_kept == len(want["kept_read_ids"])
        # short keys: the fused way in (owner-major slabs, exchanged as they are -- in three chunks when the
        # collectives are real: the exchange of a chunk under the pack of the next)
        monkeypatch.setenv("FQD_OWNER_SLABS_MIN_READS", "1000")
        if not shortcuts:
            monkeypatch.setenv("FQD_SHARD_CHUNKS", "3")
        n, L = 200000, 32
        host = synth_keys(n, L, L, 23, sub_rate=3e-3, n_rate=3e-4).reshape(-1)
        got = cluster_keys_sharded(backend, torch.from_numpy(host).to(dev), None, L, max_distance=1, method="directional")
        want = oracle.dedup(host, fixed_offsets(n, L), max_distance=1, method="directional")
        assert (got.n_unique, got.n_clusters) == (want["n_unique"], want["n_clusters"])
        assert np.array_equal(got.kept_read_ids.cpu().numpy().astype(np.uint64), want["kept_read_ids"])
        # ragged + edit metric through the same path
        rag = ["ACGTACGTAC", "ACGTACGTA", "ACGTACGTACG", "TTTTTTTTTT", "TTTTTTTTT", "GGGGG"] * 50
        raw, off = _pack(rag)
        got = cluster_keys_sharded(backend, torch.from_numpy(raw.copy()).to(dev),
                                   torch.from_numpy(off.astype(np.int64)).to(dev), 0,
                                   max_distance=1, use_edit_distance=True, method="directional")
        want = oracle.dedup(raw, off, max_distance=1, use_edit_distance=True, method="directional")
        assert np.array_equal(got.kept_read_ids.cpu().numpy().astype(np.uint64), want["kept_read_ids"])
    finally:
        dist.destroy_process_group()


def test_synth_twin_beyond_4gib(ctx):
    """A launch is capped at 2^32 threads: the generator must still fill a 4.5 GB buffer."""
    import torch
    from fastqdedup_amd.synth import synth_keys_range
    n, L, umi, seed = 15_000_000, 300, 300, 1005
    dev = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
    ctx.synth_keys(dev, n, 0, n, L, umi, seed)
    for start in (0, (1 << 32) // L - 3, n - 50):
        want = synth_keys_range(n, start, 50, L, umi, seed)
        got = dev[start * L:(start + 50) * L].cpu().numpy().reshape(50, L)
        assert np.array_equal(got, want), start
    del dev


def test_full_size_properties(F, ctx):
    """BASELINE config 2 at full size (10 M x 100 nt, UMI 12, d=1): properties that
    need no CPU-side answer."""
    import torch
    n, L, umi, seed = 10_000_000, 100, 12, 1002
    dev = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
    ctx.synth_keys(dev, n, 0, n, L, umi, seed)
    d0 = F.cluster_keys(dev, key_len=L, max_distance=0, context=ctx)
    assert d0.n_kept == d0.n_unique == d0.n_clusters           # d=0: one cluster per distinct key
    ids0 = d0.kept_read_ids
    assert (np.diff(ids0.astype(np.int64)) > 0).all()          # sorted, no read twice
    hc = F.cluster_keys(dev, key_len=L, max_distance=1, method="highest_count", context=ctx)
    dr = F.cluster_keys(dev, key_len=L, max_distance=1, method="directional", context=ctx)
    ad = F.cluster_keys(dev, key_len=L, max_distance=1, method="adjacency", context=ctx)
    assert hc.n_unique == d0.n_unique and hc.n_kept == hc.n_clusters
    assert hc.n_clusters == dr.n_clusters == ad.n_clusters < d0.n_clusters
    assert hc.n_kept <= dr.n_kept <= d0.n_kept and hc.n_kept <= ad.n_kept <= d0.n_kept
    # every kept read is the first holder of a distinct key: subset of the d=0 answer
    assert np.isin(dr.kept_read_ids, ids0).all() and np.isin(ad.kept_read_ids, ids0).all()
    # idempotence: clustering only the kept reads again keeps them all under highest_count's
    # complement -- no two kept keys of `adjacency` are adjacent
    sel = torch.from_numpy(ad.kept_read_ids.astype(np.int64)).to("cuda:0")
    sub = dev.view(n, L)[sel].contiguous().view(-1)
    again = F.cluster_keys(sub, key_len=L, max_distance=1, method="adjacency", context=ctx)
    assert again.n_edges == 0 and again.n_kept == ad.n_kept


def test_synth_indel_twin(ctx):
    """The device generator of the indel tail (fqd_synth_indel_keys) equals synth.indel_variant byte for byte."""
    from fastqdedup_amd.synth import indel_variant, synth_keys
    n, L, umi, seed = 30_000, 150, 150, 1005
    got_bytes, got_off = ctx.synth_indel_keys(n, 0, n, L, umi, seed, indel_rate=0.03)
    want_bytes, want_off = indel_variant(synth_keys(n, L, umi, seed), seed, indel_rate=0.03)
    assert np.array_equal(got_off.cpu().numpy().astype(np.uint64), want_off)
    assert np.array_equal(got_bytes.cpu().numpy(), want_bytes)
    lens = np.diff(want_off.astype(np.int64))
    assert set(lens.tolist()) == {L - 1, L, L + 1}


@pytest.mark.parametrize("path", ["grouped", "sort", "auto", "auto_own_index_hashes"])
def test_edit_search_with_an_indel_tail_matches_oracle(F, oracle, monkeypatch, path):
    """SURVEY.md 8d's config-5 variant: paired 2x150 keys (300 nt) of which 1 % are 299 or 301 nt
    long, Levenshtein d = 1, adjacency -- 250 k reads against the oracle, through the sort-free
    search (items partitioned and matched in LDS) and through the sorted one; both must also
    report every edge exactly once (same edge count)."""
    from fastqdedup_amd.synth import indel_variant, synth_keys
    if path.startswith("auto"):   # d = 1: Hamming passes for pairs of one length, the edit search for the others
        monkeypatch.delenv("FQD_EDIT", raising=False)
        if path == "auto_own_index_hashes":     # ... hashing its index items itself instead of taking the passes' hashes
            monkeypatch.setenv("FQD_EDIT_OWN_INDEX_HASHES", "1")
    else:
        monkeypatch.setenv("FQD_EDIT", path)
    n, L, seed = 250_000, 300, 1005
    raw, off = indel_variant(synth_keys(n, L, L, seed), seed, indel_rate=0.01)
    ctx = F.Context(0)
    got = F.cluster_keys(raw, off, max_distance=1, use_edit_distance=True, method="adjacency", context=ctx)
    want = oracle.dedup(raw, off, max_distance=1, use_edit_distance=True, method="adjacency")
    assert got.n_unique == want["n_unique"]
    assert got.n_clusters == want["n_clusters"]
    assert np.array_equal(got.kept_read_ids, want["kept_read_ids"])
    times = ctx.kernel_times(reset=True)
    if path != "sort":
        assert times["gp_scatter_kernel"][1] and times["verify_candidates_kernel"][1], times   # the sort-free way ran
    test_edit_search_with_an_indel_tail_matches_oracle.edges = getattr(
        test_edit_search_with_an_indel_tail_matches_oracle, "edges", {})
    test_edit_search_with_an_indel_tail_matches_oracle.edges[path] = got.n_edges
    e = test_edit_search_with_an_indel_tail_matches_oracle.edges
    assert len(set(e.values())) == 1, e


@pytest.mark.parametrize("d,method", [(2, "directional"), (3, "adjacency"), (1, "highest_count")])
def test_grouped_edit_search_mixed_lengths(F, oracle, monkeypatch, d, method):
    """Several length classes of comparable size, indels and substitutions, d up to 3: the probing
    rule (the smaller class probes the larger, the own class with shifts) and the
    first-matching-configuration rule against the oracle and against the sorted search."""
    import random
    rng = random.Random(100 + d)
    mols = ["".join(rng.choice("ACGT") for _ in range(rng.choice([38, 39, 40, 40, 41]))) for _ in range(2500)]
    strs = []
    for _ in range(40_000):
        s = list(rng.choice(mols))
        for _ in range(rng.choice([0, 0, 1, 1, 2, 3])):
            pos = rng.randrange(len(s))
            s[pos:pos + 1] = rng.choice([[], [rng.choice("ACGTN")], [s[pos], rng.choice("ACGT")]])
        strs.append("".join(s))
    raw, off = _pack(strs)
    ctx = F.Context(0)
    out = {}
    for path in ("grouped", "sort"):
        monkeypatch.setenv("FQD_EDIT", path)
        out[path] = F.cluster_keys(raw, off, max_distance=d, use_edit_distance=True, method=method, context=ctx)
    want = oracle.dedup(raw, off, max_distance=d, use_edit_distance=True, method=method)
    for path, got in out.items():
        assert got.n_unique == want["n_unique"] and got.n_clusters == want["n_clusters"], path
        assert np.array_equal(got.kept_read_ids, want["kept_read_ids"]), path
    assert out["grouped"].n_edges == out["sort"].n_edges


def test_ragged_keys_collapse_without_a_sort(F, oracle, monkeypatch):
    """Ragged keys (trimmed reads) through the (hash, position) pairs collapse: lengths are compared
    with the records -- 'AC' and 'ACA' pack to the same words when A is code 0."""
    import random
    rng = random.Random(6)
    mols = ["".join(rng.choice("ACGT") for _ in range(rng.randint(20, 60))) for _ in range(20_000)]
    strs = [rng.choice(mols) for _ in range(150_000)]
    strs += ["A" * k for k in range(1, 70)] * 3 + ["AC", "ACA", "ACAA", "AC"]
    raw, off = _pack(strs)
    want = oracle.dedup(raw, off, max_distance=1, method="directional")
    for path in ("pairs", "pairs with lengths looked up", "pairs in slices of 64", "sort"):
        monkeypatch.setenv("FQD_COLLAPSE", path.split()[0])
        if "looked up" in path:
            monkeypatch.setenv("FQD_NO_LEN_IN_RECORD", "1")   # (by default a ragged record carries its key's length, pack.hip)
        else:
            monkeypatch.delenv("FQD_NO_LEN_IN_RECORD", raising=False)
        if "slices" in path:
            monkeypatch.setenv("FQD_PAIRS_SLICE", "64")       # (the rows of a bucket's slices joined by pairs_merge_kernel)
        ctx = F.Context(0)
        got = F.cluster_keys(raw, off, max_distance=1, method="directional", context=ctx)
        assert got.n_unique == want["n_unique"], path
        assert np.array_equal(got.kept_read_ids, want["kept_read_ids"]), path
        times = ctx.kernel_times(reset=True)
        if path != "sort":
            assert times["bucket_dedupe_kernel"][1] and not times["head_flags_kernel"][1], times


def test_page_locked_id_arrays_are_capped(F, monkeypatch):
    """Large kept-id lists come back in page-locked host memory (the copy runs at the link's rate); a caller that keeps
    many results must not keep unbounded memory locked: beyond FQD_PINNED_IDS_CAP bytes of live results the arrays are
    pageable, and a result that is dropped gives its share back."""
    import gc
    from fastqdedup_amd import _lib
    gc.collect()
    live0 = _lib._pinned_live[0]
    n = 1 << 18
    monkeypatch.setenv("FQD_PINNED_IDS_CAP", str(live0 + 8 * n + 8 * n // 2))      # (room for one and a half such arrays)
    a = _lib._host_ids(n)
    assert _lib._pinned_live[0] == live0 + 8 * n
    b = _lib._host_ids(n)                   # over the cap: pageable, not counted
    assert _lib._pinned_live[0] == live0 + 8 * n and a.shape == b.shape == (n,) and b.dtype == np.uint64
    a[:] = 7
    del a
    gc.collect()
    assert _lib._pinned_live[0] == live0
    c = _lib._host_ids(n)                   # room again
    assert _lib._pinned_live[0] == live0 + 8 * n
    del c, b


def test_a_table_with_lengths_in_its_rows_becomes_a_store(F, monkeypatch):
    """Ragged long keys clustered in one call leave a unique table whose rows hold their key's length in their last
    padding word (pack.hip); keys ADDED to that table through the store are packed without -- the store clears the word
    of the resident rows first, or equal keys of the two kinds would not meet. 'AC' + 'A' * k against 'AC' + 'A' * (k+1):
    the same key words when A is code 0."""
    import random
    monkeypatch.setenv("FQD_COLLAPSE", "pairs")
    rng = random.Random(11)
    mols = ["".join(rng.choice("ACGT") for _ in range(rng.randint(40, 70))) for _ in range(5_000)]
    mols += ["AC" + "A" * k for k in range(30, 60)]
    first = [rng.choice(mols) for _ in range(70_000)]
    later = [rng.choice(mols) for _ in range(70_000)] + ["".join(rng.choice("ACGT") for _ in range(50)) for _ in range(500)]
    ctx = F.Context(0)
    raw, off = _pack(first)
    got = F.cluster_keys(raw, off, max_distance=0, method="highest_count", context=ctx)
    assert got.n_unique == len(set(first)) and got.route["collapse_pairs"], got.route
    raw2, off2 = _pack(later)
    assert ctx.store_add_keys(raw2, off2) == len(set(first) | set(later))
    # ... and the other way round: a store first, then a one-call job on the same context
    got = F.cluster_keys(raw, off, max_distance=0, method="highest_count", context=ctx)
    assert got.n_unique == len(set(first))


def test_host_keys_uploaded_in_pieces_under_the_pack(F, oracle, monkeypatch):
    """Keys in host memory, the fused way in: the bytes travel in pieces on the second stream and the pack kernel of a
    piece runs under the copy of the next (large jobs by default; FQD_UPLOAD_PIECES pins the number). The read indices
    of the later pieces start at their first read (PackScatter::id_base)."""
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    monkeypatch.setenv("FQD_FUSED_MIN_READS", "100000")
    n, L = 300_001, 32                      # (not a multiple of anything: the last piece is short)
    keys = synth_keys(n, L, 12, 4711, sub_rate=3e-3, n_rate=1e-3)
    raw = np.ascontiguousarray(keys).reshape(-1)
    want = oracle.dedup(raw, fixed_offsets(n, L), max_distance=1, method="directional")
    for pieces in ("1", "3", "16"):
        monkeypatch.setenv("FQD_UPLOAD_PIECES", pieces)
        ctx = F.Context(0)
        got = F.cluster_keys(raw, key_len=L, max_distance=1, method="directional", context=ctx)
        assert got.route["fused_pack"], got.route
        per = (-(-n // int(pieces)) + 8191) // 8192 * 8192          # (pieces end on multiples of 8192 reads)
        assert ctx.kernel_times(reset=True)["pack_kernel"][1] == (1 if pieces == "1" else -(-n // per))
        assert (got.n_unique, got.n_clusters) == (want["n_unique"], want["n_clusters"]), pieces
        assert np.array_equal(got.kept_read_ids, want["kept_read_ids"]), pieces


def test_a_context_tries_its_fast_path_again(F, oracle, monkeypatch):
    """A file with a jackpot key makes the context run the fused collapse with its spill list and without routing.
    After eight jobs that spilled nothing it tries the routed collapse again (and keeps it when the data allows);
    a retry that fails doubles the wait."""
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    monkeypatch.setenv("FQD_FUSED_MIN_READS", "100000")
    n, L = 300_000, 32
    plain = synth_keys(n, L, 12, 77, sub_rate=3e-3, n_rate=1e-4)
    hot = plain.copy()
    rows = np.random.default_rng(5).choice(n, size=n // 8, replace=False)
    hot[rows] = hot[rows[0]]
    raws = {"plain": np.ascontiguousarray(plain).reshape(-1), "hot": np.ascontiguousarray(hot).reshape(-1)}
    want = {k: oracle.dedup(v, fixed_offsets(n, L), max_distance=1, method="directional") for k, v in raws.items()}
    ctx = F.Context(0)

    def job(kind):
        got = F.cluster_keys(raws[kind], key_len=L, max_distance=1, method="directional", context=ctx)
        assert np.array_equal(got.kept_read_ids, want[kind]["kept_read_ids"]), kind
        return got.route
    assert job("plain")["pass0_in_collapse"]
    r = job("hot")
    assert r["spill_list"] and not r["pass0_in_collapse"]
    for i in range(8):                       # eight jobs that spill nothing: still the careful way ...
        r = job("plain")
        assert r["spill_list"] and not r["pass0_in_collapse"], i
    r = job("plain")                         # ... then the fast one again
    assert r["pass0_in_collapse"] and not r["spill_list"] and not r["restarted"], r
    r = job("hot")                           # (and back, as before)
    assert r["spill_list"] and not r["restarted"], r
    monkeypatch.setenv("FQD_NO_FAST_PATH_RETRY", "1")
    for i in range(10):
        r = job("plain")
        assert r["spill_list"], i


@pytest.mark.parametrize("d,switch", [(1, "FQD_DIRECTIONAL_SPLIT_UNIONS"), (2, "FQD_DIRECTIONAL_NO_SPLIT_UNIONS"), (2, None),
                                      (1, None)])
def test_directional_unions_in_pass_1_or_1b(F, oracle, monkeypatch, d, switch):
    """The closed-form directional dissection unites the edges between count-1 keys either in its first pass over the
    edges or (distance >= 2 by default) in a pass of its own that also drops the edges between two tainted keys from
    the list of pass 2: same verdicts either way, against the oracle -- many count-1 variants next to each other
    (sub_rate 1 %), keys with several copies, weights that make count-1 keys out of popular ones."""
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    if switch:
        monkeypatch.setenv(switch, "1")
    n, L = 120_000, 32
    keys = synth_keys(n, L, 10, 99 + d, copies=6, sub_rate=1e-2, n_rate=1e-4)
    weights = (np.random.default_rng(d).random(n) < 0.6).astype(np.uint32)
    raw = np.ascontiguousarray(keys).reshape(-1)
    for w in (None, weights):
        want = oracle.dedup(raw, fixed_offsets(n, L), w, max_distance=d, method="directional")
        got = F.cluster_keys(raw, key_len=L, weights=w, max_distance=d, method="directional", context=F.Context(0))
        assert (got.n_unique, got.n_clusters, got.n_kept) == (want["n_unique"], want["n_clusters"], len(want["kept_read_ids"]))
        assert np.array_equal(got.kept_read_ids, want["kept_read_ids"])

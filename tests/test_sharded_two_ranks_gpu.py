"""Two and three ranks of the production multi-GPU path on ONE MI355X, both plans: every
process runs the arithmetic on GPU 0 through HipBackend (every C-ABI export/import, owner
grouping, segment-routed search passes, clusters dissected away from their owners, id
windows) while the collectives go over gloo on host copies (RCCL refuses two ranks on one
device; ``comm_via_host``). Checked against the CPU oracle. GPU only."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, plan, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if plan.endswith("+padded"):
        # whole slabs on the wire, slack included (equal-split all-to-all) instead of their filled prefixes (all-to-all-v)
        plan = plan[: -len("+padded")]
        os.environ["FQD_NO_DENSE_SLABS"] = "1"
    if plan.endswith("+chunks3"):
        # ... leaving in three chunks per rank (the exchange of a chunk runs under the pack of the next)
        plan = plan[: -len("+chunks3")]
        os.environ["FQD_SHARD_CHUNKS"] = "3"
    if plan.endswith("+slabs"):
        # the fused way in (owner-major slabs) also for a job this small
        plan = plan[: -len("+slabs")]
        os.environ["FQD_OWNER_SLABS_MIN_READS"] = "1000"
        if world == 5:
            os.environ["FQD_LDS_BUCKET_BITS"] = "18"
    else:
        os.environ["FQD_NO_OWNER_SLABS"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fastqdedup_amd as F
        from fastqdedup_amd.sharded import HipBackend, cluster_keys_sharded

        gpu = torch.device("cuda", 0)
        keys, offsets, key_len, weights, d, edit, method = case[rank]
        k = torch.from_numpy(keys).to(gpu)
        o = None if offsets is None else torch.from_numpy(offsets.astype(np.int64)).to(gpu)
        w = None if weights is None else torch.from_numpy(weights.astype(np.int32)).to(gpu)
        # the production backend on the GPU; only the collectives detour through host memory
        res = cluster_keys_sharded(HipBackend(F.Context(0), gpu), k, o, key_len, w, max_distance=d,
                                   use_edit_distance=edit, method=method, plan=plan, comm_via_host=True)
        assert res.plan == ("gathered" if edit else plan)
        q.put((rank, res.kept_read_ids.tolist(), res.n_clusters, res.n_unique, res.n_kept, res.n_reads))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape,plan", [("fixed32", "segment-routed"), ("fixed32", "gathered"),
                                        ("fixed32", "segment-routed+slabs"), ("fixed32_3ranks", "segment-routed+slabs"),
                                        ("fixed32_5ranks", "segment-routed+slabs"),
                                        ("fixed32", "segment-routed+slabs+padded"),
                                        ("fixed32_3ranks", "segment-routed+slabs+chunks3+padded"),
                                        ("fixed32", "segment-routed+slabs+chunks3"),
                                        ("fixed32_3ranks", "segment-routed+slabs+chunks3"),
                                        ("fixed32_foreign", "segment-routed+slabs"),
                                        ("fixed32_hot", "segment-routed+slabs"), ("fixed32_hot", "segment-routed"),
                                        ("fixed32_foreign", "segment-routed"),
                                        ("fixed100_weights", "segment-routed"), ("fixed100_weights", "gathered"),
                                        ("fixed300_d2", "segment-routed"),
                                        ("ragged_hamming3", "segment-routed"), ("ragged_edit", "gathered")])
def test_ranks_on_one_gpu(oracle, shape, plan):
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    world = 2
    if shape == "fixed32_5ranks":
        # five ranks: 32 hash bins per owner, and (FQD_LDS_BUCKET_BITS, set in the worker) the widest
        # level 2 the receiving collapse can have (1024 bins) -- the geometry of an 8-GPU job
        world = 5
        n, L, d, edit, method = 200_000, 32, 1, False, "directional"
        allk = synth_keys(n, L, L, 25, sub_rate=3e-3, n_rate=3e-4)
        cuts = [0, 50_000, 90_000, 90_500, 150_000, n]
        case = [(allk[cuts[r]:cuts[r + 1]].reshape(-1), None, L, None, d, edit, method) for r in range(5)]
        raw, off, w = allk.reshape(-1), fixed_offsets(n, L), None
    elif shape == "fixed32_3ranks":
        # three ranks (3 x 64 owner-major bins), one of them with few reads, d = 2, adjacency
        world = 3
        n, L, d, edit, method = 150_000, 32, 2, False, "adjacency"
        allk = synth_keys(n, L, L, 15, sub_rate=3e-3, n_rate=3e-4)
        cuts = [0, 80_000, 81_000, n]
        case = [(allk[cuts[r]:cuts[r + 1]].reshape(-1), None, L, None, d, edit, method) for r in range(3)]
        raw, off, w = allk.reshape(-1), fixed_offsets(n, L), None
    elif shape in ("fixed32", "fixed32_foreign", "fixed32_hot"):
        n, L, d, edit, method = 120_000, 32, 1, False, "directional"
        allk = synth_keys(n, L, L, 5, sub_rate=3e-3, n_rate=3e-4)
        cut = 70_000
        if shape == "fixed32_hot":
            # a key with a sixth of all reads (and its one-error cloud), spread over both ranks: the owner's slabs of
            # the fused way in overflow -- every rank repeats the way in the general way
            rows = np.random.default_rng(11).choice(n, size=n // 6, replace=False)
            allk[rows] = allk[rows[0]]
            near = rows[: len(rows) // 30]
            allk[near, np.random.default_rng(12).integers(0, L, size=len(near))] = ord("G")
        if shape == "fixed32_foreign":
            # a symbol outside "ACGNT" on rank 1 only: the optimistic pack fails there, every
            # rank must fall back to the scanned, merged symbol table
            allk[cut + 11] = np.frombuffer(bytes(allk[cut + 11]).lower(), dtype=np.uint8)
            allk[cut + 12] = allk[cut + 11]
        case = [(allk[:cut].reshape(-1), None, L, None, d, edit, method),
                (allk[cut:].reshape(-1), None, L, None, d, edit, method)]
        raw, off, w = allk.reshape(-1), fixed_offsets(n, L), None
        cuts = [0, cut, n]
    elif shape == "fixed300_d2":
        # BASELINE config 4's actual plan: key = R1 + R2 (300 nt, 128-byte records), Hamming d = 2,
        # directional, segment-routed over the ranks (three search passes, two of them routed)
        n, L, d, edit, method = 80_000, 300, 2, False, "directional"
        allk = synth_keys(n, L, L, 1004, sub_rate=1e-3, n_rate=1e-4)
        cut = 45_000
        case = [(allk[:cut].reshape(-1), None, L, None, d, edit, method),
                (allk[cut:].reshape(-1), None, L, None, d, edit, method)]
        raw, off, w = allk.reshape(-1), fixed_offsets(n, L), None
        cuts = [0, cut, n]
    elif shape == "fixed100_weights":
        n, L, d, edit, method = 60_000, 100, 2, False, "adjacency"
        allk = synth_keys(n, L, 12, 6, sub_rate=3e-3, n_rate=3e-4)
        w = np.random.default_rng(1).choice(np.array([0, 1, 1, 2], dtype=np.uint32), size=n)
        cut = 25_000
        case = [(allk[:cut].reshape(-1), None, L, w[:cut], d, edit, method),
                (allk[cut:].reshape(-1), None, L, w[cut:], d, edit, method)]
        raw, off = allk.reshape(-1), fixed_offsets(n, L)
        cuts = [0, cut, n]
    elif shape == "ragged_hamming3":
        # three ranks (one of them empty), two key lengths, weights with zeros, d = 2
        world = 3
        a = synth_keys(30_000, 24, 8, 8, sub_rate=4e-3, n_rate=3e-4)
        b = synth_keys(20_000, 20, 8, 9, sub_rate=4e-3, n_rate=3e-4)
        enc = [bytes(r) for r in a] + [bytes(r) for r in b]
        np.random.default_rng(3).shuffle(enc)
        n, d, edit, method = len(enc), 2, False, "directional"
        w = np.random.default_rng(2).choice(np.array([0, 1, 1, 3], dtype=np.uint32), size=n)
        cuts = [0, 21_000, 21_000, n]

        def pk(part):
            r = np.frombuffer(b"".join(part), dtype=np.uint8).copy()
            o = np.concatenate([[0], np.cumsum([len(e) for e in part])]).astype(np.uint64)
            return r, o
        case = []
        for r in range(world):
            rr, oo = pk(enc[cuts[r]:cuts[r + 1]])
            case.append((rr, oo, 0, w[cuts[r]:cuts[r + 1]], d, edit, method))
        raw, off = pk(enc)
    else:
        import random
        rng = random.Random(4)
        mols = ["".join(rng.choice("ACGT") for _ in range(rng.randint(16, 26))) for _ in range(300)]
        strs = []
        for _ in range(4000):
            s = list(rng.choice(mols))
            if rng.random() < 0.4:
                pos = rng.randrange(len(s))
                s[pos:pos + 1] = rng.choice([[], [rng.choice("ACGT")], [s[pos], rng.choice("ACGTN")]])
            strs.append("".join(s))
        d, edit, method, w = 1, True, "directional", None
        enc = [s.encode() for s in strs]
        cut = 1500

        def pk(part):
            r = np.frombuffer(b"".join(part), dtype=np.uint8).copy()
            o = np.concatenate([[0], np.cumsum([len(e) for e in part])]).astype(np.uint64)
            return r, o
        r0, o0 = pk(enc[:cut])
        r1, o1 = pk(enc[cut:])
        case = [(r0, o0, 0, None, d, edit, method), (r1, o1, 0, None, d, edit, method)]
        raw, off = pk(enc)
        cuts = [0, cut, len(enc)]
        n = len(enc)

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, plan, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = oracle.dedup(raw, off, w, max_distance=d, use_edit_distance=edit, method=method)
    want_ids = want["kept_read_ids"].tolist()
    union = []
    for rank, kept, n_clusters, n_unique, n_kept, n_reads in got:
        lo, hi = cuts[rank], cuts[rank + 1]
        assert kept == [i for i in want_ids if lo <= i < hi], (shape, rank)
        union += kept
        assert (n_clusters, n_unique, n_kept, n_reads) == (want["n_clusters"], want["n_unique"],
                                                           len(want_ids), n)
    assert sorted(union) == want_ids


def _big_worker(rank, world, port, n_per_rank, L, seed, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fastqdedup_amd as F
        from fastqdedup_amd.sharded import HipBackend, cluster_keys_sharded

        gpu = torch.device("cuda", 0)
        ctx = F.Context(0)
        keys = torch.empty(n_per_rank * L, dtype=torch.uint8, device=gpu)
        ctx.synth_keys(keys, world * n_per_rank, rank * n_per_rank, n_per_rank, L, L, seed)
        res = cluster_keys_sharded(HipBackend(ctx, gpu), keys, None, L, max_distance=1, method="directional",
                                   comm_via_host=True)
        kept = res.kept_read_ids
        # a digest instead of millions of ids through the queue
        q.put((rank, int(kept.shape[0]), int(kept.sum().item()), int((kept * (kept % 1009 + 1)).sum().item()),
               res.n_clusters, res.n_unique, res.n_kept, bool((kept[1:] > kept[:-1]).all().item())))
    finally:
        dist.destroy_process_group()


def test_two_ranks_at_scale_equal_the_single_gpu_job():
    """2 x 3 M reads through the segment-routed plan (LDS collapse of received reads with carried
    ids, partitioned search passes, binned candidate lists -- the paths large jobs take) against
    the plain single-GPU path on the same 6 M reads."""
    import fastqdedup_amd as F
    world, n_per_rank, L, seed = 2, 3_000_000, 32, 1777
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_big_worker, args=(r, world, port, n_per_rank, L, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0

    gpu = torch.device("cuda", 0)
    one = F.Context(0)
    n = world * n_per_rank
    keys = torch.empty(n * L, dtype=torch.uint8, device=gpu)
    one.synth_keys(keys, n, 0, n, L, L, seed)
    want = F.cluster_keys(keys, key_len=L, max_distance=1, method="directional", context=one)
    ids = torch.from_numpy(want.kept_read_ids.astype(np.int64))
    for rank, count, total, mixed, n_clusters, n_unique, n_kept, ascending in got:
        mine = ids[(ids >= rank * n_per_rank) & (ids < (rank + 1) * n_per_rank)]
        assert ascending
        assert (count, total, mixed) == (int(mine.shape[0]), int(mine.sum().item()),
                                         int((mine * (mine % 1009 + 1)).sum().item())), rank
        assert (n_clusters, n_unique, n_kept) == (want.n_clusters, want.n_unique, want.n_kept)


@pytest.mark.parametrize("routed,mostly_unique", [(False, False), (True, False), (False, True)])
def test_eight_owners_geometry_in_one_process(routed, mostly_unique, monkeypatch):
    """The way in of an 8-rank job -- 8 senders x 8 owners, 32 hash bins per owner (config 4's geometry), ids stamped
    with 8 sender ranks -- on ONE process (a GPU box admits 6 processes): every "rank" packs its shard into owner-major
    slabs, every "owner" gets the slab ranges an all-to-all would deliver and collapses them. The union of the owners'
    unique tables must be the single-GPU job's table; with owner routing (fqd_set_owner_routing) the owners' collapse
    reports search pass 0, whose pairs must be the ones the search finds without it. mostly_unique: ~1560 reads per
    bucket (what 50 M received reads leave in the 2^15 buckets an owner of an 8-rank job can make), nearly all of them
    distinct keys -- the dedupe then runs with its 2048-slot table (1024 slots overflowed there)."""
    import fastqdedup_amd as F
    world, n_per_rank, L, seed, d = 8, 400_000, 32, 4242, 1
    if mostly_unique:
        monkeypatch.setenv("FQD_LDS_BUCKET_BITS", "8")
    n = world * n_per_rank
    gpu = torch.device("cuda", 0)
    ctx = F.Context(0)
    dna = np.zeros(128, dtype=np.uint8)
    dna[[ord(ch) for ch in "ACGNT"]] = 1
    ctx.configure(dna, L, False)
    assert ctx.owner_routing_possible(L, d + 1)
    ctx.set_owner_routing(routed)
    geometry = F.Context.owner_slab_geometry(n_per_rank, world)
    hb, subs, cap = geometry
    assert hb == 32                                              # 256 level-1 bins over 8 owners
    parts = world * hb * subs
    keys = torch.empty(n * L, dtype=torch.uint8, device=gpu)
    ctx.synth_keys(keys, n, 0, n, L, 12 if not mostly_unique else L, seed, copies=1 if mostly_unique else 4,
                   sub_rate=3e-3, n_rate=1e-3)
    send = torch.empty((world, parts * cap, 4), dtype=torch.int32, device=gpu)
    scur = torch.empty((world, parts), dtype=torch.int32, device=gpu)
    per_owner = np.zeros(world, dtype=np.int64)
    for s in range(world):
        counts = ctx.pack_to_owner_slabs(keys[s * n_per_rank * L:(s + 1) * n_per_rank * L], L, world, d + 1, 0, geometry,
                                         send[s], scur[s])
        assert counts is not None and sum(counts) == n_per_rank
        per_owner += np.array(counts)
    ppo = hb * subs                                              # slabs per (sender, owner)
    first_all, count_all, pass0_pairs = [], [], set()
    for p in range(world):
        recv = torch.stack([send[s].reshape(world, ppo * cap, 4)[p] for s in range(world)]).reshape(-1, 4).contiguous()
        rcur = torch.stack([scur[s].reshape(world, ppo)[p] for s in range(world)]).reshape(-1).contiguous()
        nu = ctx.collapse_owner_slabs(recv, rcur, world, p, geometry, [s * n_per_rank for s in range(world)], n,
                                      int(per_owner[p]), d + 1)
        assert nu is not None, p
        ne = ctx.find_edges_segments(d, 0, 1)
        assert ctx.route()["pass0_continued"] == routed, (p, ctx.route())
        first, counts, _, _ = ctx.unique_table(nu, labels=False, kept=False)
        edges = np.empty((ne, 2), dtype=np.int32)
        ctx.export_edges(edges)
        a, b = first[edges[:, 0]], first[edges[:, 1]]
        pass0_pairs |= set(zip(np.minimum(a, b).tolist(), np.maximum(a, b).tolist()))
        first_all.append(first)
        count_all.append(counts)
    first_all, count_all = np.concatenate(first_all), np.concatenate(count_all)
    order = np.argsort(first_all)
    one = F.Context(0)
    want = F.cluster_keys(keys, key_len=L, max_distance=d, method="directional", context=one)
    w_first, w_counts, _, _ = one.unique_table(want.n_unique, labels=False, kept=False)
    w_order = np.argsort(w_first)
    assert np.array_equal(first_all[order], w_first[w_order])
    assert np.array_equal(count_all[order], w_counts[w_order])
    # pass 0 = the pairs within distance d that agree on segment 0: the same set either way (kept for the other run)
    if mostly_unique:
        return
    seen = test_eight_owners_geometry_in_one_process.__dict__.setdefault("pairs", {})
    seen[routed] = pass0_pairs
    if len(seen) == 2:
        assert seen[False] == seen[True] and len(seen[True]) > 1000


def test_dense_owner_slabs_never_write_past_the_rows_they_were_given():
    """fqd_dense_owner_slabs derives the fills from the cursors, not from the caller's count: rows behind the capacity
    it was told must be dropped, never written (round 3's advisor: after a pack that gave up, cursors of whatever value
    drove the copy past a 16-byte buffer). A canary behind a deliberately short buffer stays intact; with room for
    every row the rows are the slabs' filled prefixes, back to back."""
    import fastqdedup_amd as F
    world, n, L, d = 2, 300_000, 32, 1
    gpu = torch.device("cuda", 0)
    ctx = F.Context(0)
    dna = np.zeros(128, dtype=np.uint8)
    dna[[ord(ch) for ch in "ACGNT"]] = 1
    ctx.configure(dna, L, False)
    geometry = F.Context.owner_slab_geometry(n, world)
    hb, subs, cap = geometry
    parts = world * hb * subs
    keys = torch.empty(n * L, dtype=torch.uint8, device=gpu)
    ctx.synth_keys(keys, n, 0, n, L, 12, 99)
    slabs = torch.empty((parts * cap, 4), dtype=torch.int32, device=gpu)
    cursors = torch.empty(parts, dtype=torch.int32, device=gpu)
    counts = ctx.pack_to_owner_slabs(keys, L, world, d + 1, 0, geometry, slabs, cursors)
    assert counts is not None and sum(counts) == n
    fills = torch.empty(parts, dtype=torch.int32, device=gpu)
    # room for every row: the filled prefixes, back to back
    rows = torch.empty((n, 4), dtype=torch.int32, device=gpu)
    ctx.dense_owner_slabs(slabs, cursors, world, geometry, rows, fills)
    ctx.synchronize()
    f = fills.cpu().numpy().astype(np.int64)
    assert int(f.sum()) == n
    starts = np.concatenate([[0], np.cumsum(f)])
    sl = slabs.cpu().numpy().reshape(parts, cap, 4)
    got = rows.cpu().numpy()
    for p in (0, 1, parts // 2, parts - 1):
        assert np.array_equal(got[starts[p]:starts[p + 1]], sl[p, :f[p]]), p
    # room for a tenth: nothing behind it may change
    short = n // 10
    big = torch.full((n + 64, 4), 0x5A5A5A5A, dtype=torch.int32, device=gpu)
    ctx.dense_owner_slabs(slabs, cursors, world, geometry, big[:short], fills)
    ctx.synchronize()
    assert bool((big[short:] == 0x5A5A5A5A).all()), "rows were written behind the capacity"
    assert np.array_equal(big[:short].cpu().numpy(), got[:short])

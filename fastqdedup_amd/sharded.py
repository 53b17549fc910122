"""Multi-GPU clustering: one process per MI355X, ``torch.distributed`` (backend
"nccl" == RCCL over xGMI) for the exchange steps, the HIP library for all the
arithmetic. The reference has no counterpart (it is single-threaded,
SURVEY.md section 2); this is how the hot path of ONE job spreads over the 8
GPUs of a node (DESIGN.md "multi-GPU").

Exchange steps (everything else is rank-local):

  1. all-gather of 3 small words/rank + all-reduce(MAX) of the 128-entry
     symbol table  -> every rank packs with the SAME record geometry.
  2. all-to-all(v) of packed reads by ``owner = key_hash mod G``: all copies of a
     key meet on one rank, which collapses them (count, first holder).
  3. all-gather of the per-rank unique tables: with 288 GB of HBM every rank
     holds the whole unique-key table (config 4: 1e8 keys x 128 B = 12.8 GB),
     so the bucket pair search needs no further key movement -- rank r searches
     the buckets with ``bucket_hash mod G == r`` (the k-mer-bucket shard).
  4. all-gather of the edge shards; components and dissection then run on the
     full edge list on every rank (they are a few percent of the job).

The ``backend`` object does the arithmetic: ``HipBackend`` in production; tests
inject a numpy stand-in to exercise the exchange logic on CPU with gloo.
"""
from __future__ import annotations

from dataclasses import dataclass

import os
import time

import numpy as np
import torch
import torch.distributed as dist

from ._lib import METHODS, METRIC_EDIT, METRIC_HAMMING, Context


@dataclass
class ShardedResult:
    kept_read_ids: torch.Tensor   # int64, ascending: the kept global read ids among THIS rank's reads
    n_reads: int                  # whole job
    n_unique: int
    n_edges: int
    n_clusters: int
    n_kept: int


class HipBackend:
    """Arithmetic of the sharded job on one MI355X through libfqdedup_hip.so."""

    def __init__(self, ctx: Context, device: torch.device):
        self.ctx = ctx
        self.device = device

    def scan(self, keys, offsets, key_len):
        present, max_len, ragged = self.ctx.scan_keys(keys, offsets, key_len)
        return present, max_len, ragged

    def configure(self, present, max_len, ragged):
        self.ctx.configure(present, max_len, ragged)

    def pack_by_owner(self, keys, offsets, key_len, n_parts, id0, weights):
        """Pack this rank's keys and return them grouped by owner rank (part 0 first, every
        part in read order): rows, lengths (None unless ragged), global ids, weights (None if
        not given), rows per part."""
        n = self.ctx.pack_keys(keys, offsets, key_len)
        sh = self.ctx.shape()
        self.stride = int(sh.stride_words)
        self.ragged = bool(sh.ragged)
        recs = torch.empty((n, self.stride), dtype=torch.int32, device=self.device)
        lens = torch.empty(n, dtype=torch.int32, device=self.device) if self.ragged else None
        ids = torch.empty(n, dtype=torch.int64, device=self.device)
        w_in = None if weights is None else torch.as_tensor(weights).to(self.device).to(torch.int32).contiguous()
        w_out = None if weights is None else torch.empty(n, dtype=torch.int32, device=self.device)
        counts = self.ctx.export_packed_by_owner(n_parts, id0, w_in, recs, lens, ids, w_out)
        return recs, lens, ids, w_out, [int(c) for c in counts]

    def collapse_packed(self, recs, lens, weights, read_ids):
        n = recs.shape[0]
        self.ctx.import_packed(recs, lens if self.ragged else None, n)
        nu = self.ctx.collapse(weights, read_ids)
        urecs = torch.empty((nu, self.stride), dtype=torch.int32, device=self.device)
        ulens = torch.empty(nu, dtype=torch.int32, device=self.device) if self.ragged else None
        ucounts = torch.empty(nu, dtype=torch.int32, device=self.device)
        ufirst = torch.empty(nu, dtype=torch.int64, device=self.device)
        self.ctx.export_unique(urecs, ulens, ucounts, ufirst)
        return urecs, ulens, ucounts, ufirst

    def find_edges(self, urecs, ulens, ucounts, ufirst, max_distance, metric, shard, n_shards):
        self.ctx.import_unique(urecs, ulens, ucounts, ufirst, urecs.shape[0])
        ne = self.ctx.find_edges(max_distance, metric, shard, n_shards)
        edges = torch.empty((ne, 2), dtype=torch.int32, device=self.device)
        self.ctx.export_edges(edges)
        return edges

    def finish(self, edges, method, id_lo, id_hi):
        self.ctx.import_edges(edges, edges.shape[0])
        n_clusters = self.ctx.components()
        self.ctx.set_id_window(id_lo, id_hi)     # list only the kept ids among this rank's own reads
        try:
            self.ctx.dissect(method)
        finally:
            self.ctx.set_id_window()
        n_kept, n_listed = self.ctx.kept_count()
        kept = torch.empty(n_listed, dtype=torch.int64, device=self.device)
        self.ctx.kept_read_ids(n_listed, kept)
        return kept, n_clusters, n_kept


class _PhaseTimer:
    """FQD_SHARD_TIMING=1: wall time per phase (with a device sync at each mark) on stderr."""

    def __init__(self, dev):
        self.dev, self.t, self.out = dev, time.perf_counter(), []

    def mark(self, name):
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)
        now = time.perf_counter()
        self.out.append(f"{name}={1e3 * (now - self.t):.2f}")
        self.t = now

    def done(self):
        import sys
        print("[fqd shard ms] " + " ".join(self.out), file=sys.stderr, flush=True)


def _all_gather_rows(x: torch.Tensor, group) -> torch.Tensor:
    """Concatenation over ranks (rank-major) of tensors that differ in dim 0."""
    world = dist.get_world_size(group)
    n = torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    pad = torch.zeros((cap,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    pad[: x.shape[0]] = x
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def _all_to_all_rows(x: torch.Tensor, send_counts: torch.Tensor, recv_counts, group) -> torch.Tensor:
    """all-to-all(v) of the rows of x, already grouped by destination rank."""
    out = torch.empty((int(sum(recv_counts)),) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_to_all_single(out, x.contiguous(), output_split_sizes=list(recv_counts),
                           input_split_sizes=[int(c) for c in send_counts.tolist()], group=group)
    return out


def cluster_keys_sharded(backend, keys, offsets=None, key_len: int = 0, weights=None, *,
                         max_distance: int = 1, use_edit_distance: bool = False,
                         method="directional", group=None) -> ShardedResult:
    """Cluster the union of every rank's keys as ONE job. ``keys`` is this rank's
    shard (device tensor or numpy array, as ``cluster_keys``). Read ids are global:
    rank r's reads follow rank r-1's. Counters are global; the id list is this rank's share."""
    if max_distance < 0:
        raise ValueError("max_distance should be non-negative")
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = backend.device
    tick = _PhaseTimer(dev) if os.environ.get("FQD_SHARD_TIMING") else None
    method_id = METHODS[method] if isinstance(method, str) else int(method)
    metric = METRIC_EDIT if use_edit_distance else METRIC_HAMMING

    # ---- 1. common geometry --------------------------------------------------
    present, max_len, ragged = backend.scan(keys, offsets, key_len)
    if offsets is None:
        nbytes = keys.numel() if hasattr(keys, "numel") else keys.size
        n_local = nbytes // key_len if key_len else 0
    else:
        n_local = (offsets.numel() if hasattr(offsets, "numel") else offsets.size) - 1
    mine = torch.tensor([n_local, max_len, int(ragged)], dtype=torch.int64, device=dev)
    everyone = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine, group=group)
    everyone = torch.stack(everyone).cpu().numpy()
    n_per_rank = everyone[:, 0]
    lens_seen = {int(m) for n, m, _ in everyone if n > 0}
    g_ragged = bool(everyone[:, 2].any()) or len(lens_seen) > 1
    g_max_len = int(everyone[:, 1].max()) if len(everyone) else 0
    p = torch.from_numpy(np.ascontiguousarray(present, dtype=np.uint8)).to(dev).to(torch.int32)
    dist.all_reduce(p, op=dist.ReduceOp.MAX, group=group)
    g_present = p.to(torch.uint8).cpu().numpy()
    backend.configure(g_present, g_max_len, g_ragged)
    if tick:
        tick.mark("geometry")
    id0 = int(n_per_rank[:rank].sum())
    n_total = int(n_per_rank.sum())

    # ---- 2. all copies of a key to its owner rank ------------------------------
    s_recs, s_lens, s_ids, s_w, send_counts = backend.pack_by_owner(keys, offsets, key_len, world, id0, weights)
    if tick:
        tick.mark("pack+group-by-owner")
    send_counts = torch.tensor(send_counts, dtype=torch.int64, device=dev)
    counts_in = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(counts_in, send_counts, group=group)
    recv_counts = [int(c) for c in counts_in.tolist()]
    r_recs = _all_to_all_rows(s_recs, send_counts, recv_counts, group)
    r_ids = _all_to_all_rows(s_ids, send_counts, recv_counts, group)
    r_lens = _all_to_all_rows(s_lens, send_counts, recv_counts, group) if g_ragged else None
    r_w = None if s_w is None else _all_to_all_rows(s_w, send_counts, recv_counts, group)
    # The received rows are ALREADY in global id order: all-to-all delivers source ranks in rank
    # order, rank r's ids precede rank r+1's, and every source sent its rows in id order. The
    # collapse's stable sort therefore makes the smallest global id the head of each run.
    if tick:
        tick.mark("all-to-all")
    urecs, ulens, ucounts, ufirst = backend.collapse_packed(r_recs, r_lens, r_w, r_ids)
    if tick:
        tick.mark("collapse")

    # ---- 3. whole unique table on every rank; search this rank's bucket shard ---
    g_recs = _all_gather_rows(urecs, group)
    g_lens = _all_gather_rows(ulens, group) if g_ragged else None
    g_counts = _all_gather_rows(ucounts, group)
    g_first = _all_gather_rows(ufirst, group)
    if tick:
        tick.mark("all-gather-unique")
    edges = backend.find_edges(g_recs, g_lens, g_counts, g_first, max_distance, metric, rank, world)
    if tick:
        tick.mark("find-edges")

    # ---- 4. all edges everywhere; components + dissection ------------------------
    # Every rank labels and dissects the whole graph (a few percent of the job) but LISTS only
    # the kept ids among its own reads [id0, id0 + n_local): what its pass 2 would need.
    g_edges = _all_gather_rows(edges, group)
    kept, n_clusters, n_kept = backend.finish(g_edges.contiguous(), method_id, id0, id0 + n_local)
    if tick:
        tick.mark("gather-edges+finish")
        tick.done()
    return ShardedResult(kept, n_total, int(g_recs.shape[0]), int(g_edges.shape[0]), int(n_clusters),
                         int(n_kept))

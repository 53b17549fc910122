"""Multi-GPU clustering: one process per MI355X, ``torch.distributed`` (backend
"nccl" == RCCL over xGMI) for the exchange steps, the HIP library for all the
arithmetic. The reference has no counterpart (it is single-threaded,
SURVEY.md section 2); this is how the hot path of ONE job spreads over the 8
GPUs of a node (DESIGN.md "multi-GPU").

Two plans:

``segment-routed`` (Hamming; the default). Per-rank work and traffic stay constant as
ranks are added (weak scaling), except one all-gather of the edge list (8 B/edge):

  1. geometry: 2 words/rank all-gather; fixed-length keys are packed at once with the DNA
     alphabet and a one-word all-reduce says whether anybody met another byte (only then are
     the keys scanned and the symbol tables merged with a 128-byte all-reduce(MAX)).
  2. all-to-all(v) of packed reads by ``owner = hash(segment 0 of the key) mod G``
     (pigeonhole segments of the d+1 split). All copies of a key meet on one rank, which
     collapses them into ITS rows of the job-wide unique table (uid = rank base + row) --
     and every pair of keys agreeing on segment 0 is already together, so search pass 0
     is rank-local. 16 bytes per read on the wire for keys of <= 32 nt: the read's index on
     its rank rides in the record's padding word.
  3. for each further segment s: all-to-all(v) of (record, uid) by
     ``hash(segment s) mod G``; search pass s runs on the received rows. A pair is
     emitted in the first segment it agrees on, so every edge appears once, somewhere.
  4. all-gather of the edges (uid pairs); a union-find over them on every rank labels
     each edge with its component; rank r takes the clusters with ``label mod G == r``.
  5. the rank fetches (record, count) of its clusters' keys from their owners
     (request/response all-to-all), dissects, and returns the DROPPED uids to their
     owners; an owner keeps every other row. Keys without neighbours never move.
  6. kept first-holder ids go to the rank that read them (all-to-all by id window).

``gathered`` (edit metric, or FQD_SHARD_PLAN=gathered): all-to-all by key hash,
all-gather of the whole unique table (it fits: 288 GB/GPU), bucket-sharded search,
all-gather of edges, components + dissection replicated. Simple, but per-rank work
grows with G.

The ``backend`` object does the arithmetic: ``HipBackend`` in production; tests
inject a numpy stand-in to exercise the exchange logic on CPU with gloo.
"""
from __future__ import annotations

from dataclasses import dataclass

import os
import sys
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
    plan: str = "segment-routed"
    phases_ms: dict | None = None


class HipBackend:
    """Arithmetic of the sharded job on one MI355X through libfqdedup_hip.so."""

    def __init__(self, ctx: Context, device: torch.device):
        self.ctx = ctx
        self.device = device
        self._aux = None      # second context: routed search passes and the clusters dissected here
        self._geometry = None
        self._ext = {}        # the contexts' HIP streams as torch sees them (ordering by stream waits)

    # ---- geometry -------------------------------------------------------------------
    def scan(self, keys, offsets, key_len):
        present, max_len, ragged = self.ctx.scan_keys(keys, offsets, key_len)
        return present, max_len, ragged

    def configure(self, present, max_len, ragged):
        self._geometry = (np.ascontiguousarray(present, dtype=np.uint8), int(max_len), bool(ragged))
        self.ctx.configure(*self._geometry)
        if self._aux is not None:
            self._aux.configure(*self._geometry)
        sh = self.ctx.shape()
        self.stride = int(sh.stride_words)
        self.ragged = bool(sh.ragged)

    def lib_streams(self):
        """The streams the library works on (one per context in use), as torch stream objects."""
        if self.device.type != "cuda":
            return []
        out = []
        for c in (self.ctx, self._aux):
            if c is not None and c.stream_handle():
                if id(c) not in self._ext:
                    self._ext[id(c)] = torch.cuda.ExternalStream(c.stream_handle(), device=self.device)
                out.append(self._ext[id(c)])
        return out

    @property
    def aux(self) -> Context:
        if self._aux is None:
            self._aux = Context(self.ctx.device)
            self._aux.configure(*self._geometry)
        return self._aux

    def _rows(self, n):
        recs = torch.empty((n, self.stride), dtype=torch.int32, device=self.device)
        lens = torch.empty(n, dtype=torch.int32, device=self.device) if self.ragged else None
        return recs, lens

    # ---- reads -> owner ----------------------------------------------------------------
    def pack_by_owner(self, keys, offsets, key_len, n_parts, id0, weights, n_segments=0):
        """Pack this rank's keys and return them grouped by owner rank (part 0 first, every
        part in read order): rows, lengths (None unless ragged), global ids, weights (None if
        not given), rows per part. n_segments == 0: owner = key hash; else owner = hash of
        segment 0 of the n_segments-way split."""
        self.ctx.set_owner_rule(n_parts if n_segments else 0, max(n_segments, 1), 0)   # owners in the pack pass
        try:
            n = self.ctx.pack_keys(keys, offsets, key_len)
        finally:
            self.ctx.set_owner_rule(0)
        sh = self.ctx.shape()
        self.stride = int(sh.stride_words)
        self.ragged = bool(sh.ragged)
        recs, lens = self._rows(n)
        # Records with a padding word carry the read's local index in it: no id array on the wire
        # (16 instead of 24 bytes per read for keys of <= 32 nt); the receiver adds the sender's id base.
        carried = bool(n_segments) and int(sh.stride_words) > int(sh.planes) * int(sh.words)
        ids = None if carried else torch.empty(n, dtype=torch.int64, device=self.device)
        w_in = None if weights is None else torch.as_tensor(weights).to(self.device).to(torch.int32).contiguous()
        w_out = None if weights is None else torch.empty(n, dtype=torch.int32, device=self.device)
        if n_segments:
            counts = self.ctx.export_packed_by_segment(n_parts, n_segments, 0, id0, w_in, recs, lens, ids, w_out)
        else:
            counts = self.ctx.export_packed_by_owner(n_parts, id0, w_in, recs, lens, ids, w_out)
        return recs, lens, ids, w_out, [int(c) for c in counts]

    # ---- reads -> owner, the fused way (short fixed-length keys, no weights) -------------------
    def owner_routing_possible(self, key_len, n_segments):
        return self.ctx.owner_routing_possible(key_len, n_segments)

    def set_owner_routing(self, enable):
        self.ctx.set_owner_routing(enable)

    def pack_into_owner_slabs(self, keys, key_len, n_parts, n_segments, geometry, slabs_out, cursors_out):
        """One chunk of this rank's reads into the caller's slab buffers (rows of 4 words; cursors): reads per
        owner, or None when the general way must be taken."""
        counts = self.ctx.pack_to_owner_slabs(keys, key_len, n_parts, n_segments, 0, geometry, slabs_out, cursors_out)
        if counts is None:
            return None
        sh = self.ctx.shape()
        self.stride = int(sh.stride_words)
        self.ragged = bool(sh.ragged)
        return counts

    def dense_owner_slabs(self, slabs, cursors, n_parts, geometry, n_rows):
        """The filled prefixes of the slabs back to back (n_rows x 4 words) and every slab's fill (fqd_dense_owner_slabs)."""
        hb, subs, _cap = geometry
        rows = torch.empty((max(int(n_rows), 1), 4), dtype=torch.int32, device=self.device)
        fills = torch.empty(n_parts * hb * subs, dtype=torch.int32, device=self.device)
        self.ctx.dense_owner_slabs(slabs, cursors, n_parts, geometry, rows, fills)
        return rows[: int(n_rows)], fills

    def collapse_owner_slabs(self, slabs, cursors, n_senders, my_part, geometry, sender_id0, id_limit, n_reads,
                             search_segments):
        """The receiving side (fqd_collapse_owner_slabs): unique keys of this rank, or None."""
        nu = self.ctx.collapse_owner_slabs(slabs, cursors, n_senders, my_part, geometry, sender_id0, id_limit, n_reads,
                                           search_segments)
        if nu is not None:
            self.n_unique_local = nu
        return nu

    def collapse_resident(self, recs, lens, weights, read_ids, seg_rows=None, seg_id0=None, id_limit=None) -> int:
        """Collapse the received reads; the unique table stays in the context. read_ids None: the
        records carry the sender's local index; rows seg_rows[s]..seg_rows[s+1] came from the rank
        whose reads start at id seg_id0[s]."""
        self.ctx.import_packed(recs, lens if self.ragged else None, recs.shape[0], borrow=True)
        if read_ids is None:
            self.n_unique_local = self.ctx.collapse_received(weights, seg_rows, seg_id0,
                                                             2**64 - 1 if id_limit is None else id_limit)
        else:
            self.n_unique_local = self.ctx.collapse(weights, read_ids)
        return self.n_unique_local

    def collapse_packed(self, recs, lens, weights, read_ids):
        nu = self.collapse_resident(recs, lens, weights, read_ids)
        urecs, ulens = self._rows(nu)
        ucounts = torch.empty(nu, dtype=torch.int32, device=self.device)
        ufirst = torch.empty(nu, dtype=torch.int64, device=self.device)
        self.ctx.export_unique(urecs, ulens, ucounts, ufirst)
        return urecs, ulens, ucounts, ufirst

    # ---- segment-routed plan --------------------------------------------------------------
    def local_edges(self, max_distance, seg_lo, seg_hi):
        """Edges (row pairs of this rank's unique table) of search passes [seg_lo, seg_hi)."""
        ne = self.ctx.find_edges_segments(max_distance, seg_lo, seg_hi)
        edges = torch.empty((ne, 2), dtype=torch.int32, device=self.device)
        self.ctx.export_edges(edges)
        return edges

    def unique_by_segment(self, n_parts, n_segments, segment, uid_base):
        nu = self.n_unique_local
        recs, lens = self._rows(nu)
        uids = torch.empty(nu, dtype=torch.int32, device=self.device)
        counts = self.ctx.export_unique_by_segment(n_parts, n_segments, segment, uid_base, recs, lens, uids)
        return recs, lens, uids, [int(c) for c in counts]

    def routed_edges(self, recs, lens, uids, max_distance, segment):
        """Search pass `segment` over rows received from every rank; edges as uid pairs."""
        n = recs.shape[0]
        self.aux.import_unique(recs, lens if self.ragged else None, None, None, n, borrow=True)
        self.aux.declare_distinct_keys()
        ne = self.aux.find_edges_segments(max_distance, segment, segment + 1)
        edges = torch.empty((ne, 2), dtype=torch.int32, device=self.device)
        self.aux.export_edges(edges)
        return uids[edges.long()] if ne else edges

    def edge_labels(self, edges, n_nodes):
        """(component label of every edge, number of components over all n_nodes nodes)."""
        roots = torch.empty(edges.shape[0], dtype=torch.int32, device=self.device)
        n_components = self.ctx.edge_labels(edges, edges.shape[0], n_nodes, roots)
        return roots, n_components

    def cluster_subgraph(self, edges, roots, n_nodes, n_parts, part):
        """This rank's clusters out of the job-wide edge list: (touched nodes ascending, their
        edges renumbered to positions in that list) -- fqd_cluster_subgraph."""
        n_edges = int(edges.shape[0])
        touched = torch.empty(min(2 * n_edges, n_nodes), dtype=torch.int32, device=self.device)
        sub = torch.empty((n_edges, 2), dtype=torch.int32, device=self.device)
        if not n_edges:
            return touched, sub
        nt, ns = self.ctx.cluster_subgraph(edges, roots, n_edges, n_nodes, n_parts, part, touched, sub)
        return touched[:nt], sub[:ns]

    MAX_HOME_RANKS = 16

    def local_labels(self, edges):
        """Union-find over the rows of THIS rank's table and edges between them (row pairs): (label of every row -- the
        smallest row of its component --, number of components)."""
        nu = self.n_unique_local
        self.ctx.import_edges(edges.contiguous(), edges.shape[0])
        n_components = self.ctx.components()
        labels = torch.empty(max(nu, 1), dtype=torch.int32, device=self.device)
        self.ctx.labels_into(labels)
        return labels[:nu], n_components

    def cluster_subgraph_home(self, edges, roots, n_nodes, n_parts, part, uid_bounds):
        """... with the clusters that live on this rank alone apart (fqd_cluster_subgraph_home): (touched nodes of
        this rank's share of the SPANNING clusters, their edges renumbered, the edges of its HOME clusters as rows of
        its own table, the number of edges in spanning clusters job-wide -- every rank computes the same number)."""
        n_edges = int(edges.shape[0])
        touched = torch.empty(min(2 * n_edges, n_nodes), dtype=torch.int32, device=self.device)
        sub = torch.empty((n_edges, 2), dtype=torch.int32, device=self.device)
        home = torch.empty((n_edges, 2), dtype=torch.int32, device=self.device)
        if not n_edges:
            return touched, sub, home, 0
        nt, ns, nh, n_span = self.ctx.cluster_subgraph_home(edges, roots, n_edges, n_nodes, n_parts, part, uid_bounds,
                                                            touched, sub, home)
        return touched[:nt], sub[:ns], home[:nh], n_span

    def finish_owner_home(self, home_edges, method, dropped_rows, id_hi):
        """The home clusters dissected on the table this rank holds, the rows dropped elsewhere dropped as well ->
        (ascending kept first-holder ids, count)."""
        self.ctx.import_edges(home_edges.contiguous(), home_edges.shape[0])
        self.home_components = self.ctx.components()      # (components of the table under the home edges)
        self.ctx.set_id_window(0, id_hi)
        try:
            self.ctx.dissect_except(method, dropped_rows.to(torch.int32).contiguous(), dropped_rows.shape[0])
        finally:
            self.ctx.set_id_window()
        n_kept, n_listed = self.ctx.kept_count()
        kept = torch.empty(n_listed, dtype=torch.int64, device=self.device)
        self.ctx.kept_read_ids(n_listed, kept)
        return kept, n_kept

    def gather_unique(self, rows):
        n = rows.shape[0]
        recs, lens = self._rows(n)
        counts = torch.empty(n, dtype=torch.int32, device=self.device)
        self.ctx.gather_unique(rows.to(torch.int32).contiguous(), n, recs, lens, counts)
        return recs, lens, counts

    def dissect_subgraph(self, recs, lens, counts, edges, method):
        """Verdict (uint8, 1 = kept) for every row of a table of whole clusters."""
        n = recs.shape[0]
        self.aux.import_unique(recs, lens if self.ragged else None, counts, None, n, borrow=True)
        self.aux.declare_distinct_keys()      # rows of collapsed tables: no key twice
        self.aux.import_edges(edges.contiguous(), edges.shape[0])
        self.aux.components()
        self.aux.set_id_window(0, 0)          # only the verdicts are wanted here: no id is listed
        try:
            self.aux.dissect(method)
        finally:
            self.aux.set_id_window()
        kept = torch.empty(n, dtype=torch.uint8, device=self.device)
        self.aux.kept_flags_into(kept)
        return kept

    def finish_owner(self, dropped_rows, id_hi):
        """Every row is kept except the dropped ones -> (ascending kept first-holder ids, count)."""
        self.ctx.set_id_window(0, id_hi)
        try:
            self.ctx.list_kept_except(dropped_rows.to(torch.int32).contiguous(), dropped_rows.shape[0])
        finally:
            self.ctx.set_id_window()
        n_kept, n_listed = self.ctx.kept_count()
        kept = torch.empty(n_listed, dtype=torch.int64, device=self.device)
        self.ctx.kept_read_ids(n_listed, kept)
        return kept, n_kept

    # ---- gathered plan ----------------------------------------------------------------------
    def find_edges(self, urecs, ulens, ucounts, ufirst, max_distance, metric, shard, n_shards):
        self.ctx.import_unique(urecs, ulens, ucounts, ufirst, urecs.shape[0])
        self.ctx.declare_distinct_keys()      # every key was collapsed on its owner rank
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
    """Wall time per phase, with a device sync at each mark (FQD_SHARD_TIMING=1 prints it;
    bench.py runs one extra, untimed step with it for the per-phase table)."""

    def __init__(self, dev):
        self.dev, self.t, self.out = dev, time.perf_counter(), {}

    def mark(self, name):
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)
        now = time.perf_counter()
        self.out[name] = self.out.get(name, 0.0) + 1e3 * (now - self.t)
        self.t = now

    def done(self):
        if os.environ.get("FQD_SHARD_TIMING"):
            import sys
            print("[fqd shard ms] " + " ".join(f"{k}={v:.2f}" for k, v in self.out.items()), file=sys.stderr,
                  flush=True)
        return {k: round(v, 3) for k, v in self.out.items()}


class _Comm:
    """The collectives of the job. ``via_host`` stages every buffer through host memory: the
    two-ranks-on-one-GPU test runs the production arithmetic with gloo in between (RCCL refuses
    two ranks on one device).

    Ordering between the HIP library's stream and torch's / RCCL's is by stream waits, never by a
    host synchronisation: ``lib_stream`` (the context's stream as a ``torch.cuda.ExternalStream``)
    waits for torch's current stream after a collective, torch's current stream waits for it
    before one. With ONE rank every collective is the identity and nothing is copied."""

    def __init__(self, group, device, via_host=False, lib_streams=None):
        self.group, self.device, self.via_host = group, device, via_host
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        # one rank: every collective is the identity -- unless FQD_COMM_NO_SHORTCUT=1 sends it through the backend
        # anyway (a rehearsal of the RCCL calls and of their stream ordering on a one-GPU box)
        self.alone = self.world == 1 and not os.environ.get("FQD_COMM_NO_SHORTCUT")
        # callable -> the library contexts' streams (the second context appears on first use)
        self.lib_streams = lib_streams if (device.type == "cuda" and not via_host) else None

    def _wire(self, x):
        if self.via_host:
            return x.cpu()
        if self.lib_streams is not None and x.is_cuda:
            for st in self.lib_streams():      # the library produced x: torch's stream waits for it
                torch.cuda.current_stream(x.device).wait_stream(st)
        return x

    def _back(self, x):
        if self.via_host:
            return x.to(self.device)
        # what a collective produced on torch's stream must be complete before the next library call
        # reads it: the library's streams wait for torch's (no host round trip)
        if self.lib_streams is not None and x.is_cuda:
            for st in self.lib_streams():
                st.wait_stream(torch.cuda.current_stream(x.device))
        elif x.is_cuda:
            torch.cuda.current_stream(x.device).synchronize()
        return x

    def all_gather_ints(self, values) -> np.ndarray:
        """(world, len(values)) int64 on the host."""
        if self.alone:
            return np.array([[int(v) for v in values]], dtype=np.int64)
        wire_dev = torch.device("cpu") if self.via_host else self.device
        mine = torch.tensor(list(values), dtype=torch.int64, device=wire_dev)
        everyone = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(everyone, mine, group=self.group)
        return torch.stack(everyone).cpu().numpy()

    def any_flag(self, flag: bool) -> bool:
        """True when any rank raises the flag (a one-word all-reduce)."""
        if self.alone:
            return bool(flag)
        wire_dev = torch.device("cpu") if self.via_host else self.device
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=wire_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(int(t.item()))

    def all_reduce_max(self, x: torch.Tensor) -> torch.Tensor:
        if self.alone:
            return x
        w = self._wire(x)
        dist.all_reduce(w, op=dist.ReduceOp.MAX, group=self.group)
        return self._back(w)

    def exchange_counts(self, send_counts) -> list:
        if self.alone:
            return [int(c) for c in send_counts]
        wire_dev = torch.device("cpu") if self.via_host else self.device
        s = torch.tensor([int(c) for c in send_counts], dtype=torch.int64, device=wire_dev)
        r = torch.empty(self.world, dtype=torch.int64, device=wire_dev)
        dist.all_to_all_single(r, s, group=self.group)
        return [int(c) for c in r.tolist()]

    def all_to_all_rows(self, x, send_counts, recv_counts):
        """all-to-all(v) of the rows of x, already grouped by destination rank."""
        if x is None:
            return None
        if self.alone:
            return x                       # a rank's share of its own rows: no copy
        w = self._wire(x.contiguous())
        out = torch.empty((int(sum(recv_counts)),) + tuple(x.shape[1:]), dtype=x.dtype, device=w.device)
        dist.all_to_all_single(out, w, output_split_sizes=[int(c) for c in recv_counts],
                               input_split_sizes=[int(c) for c in send_counts], group=self.group)
        return self._back(out)

    def all_to_all_into(self, out, x):
        """Equal-split all-to-all of the rows of x into `out` (same shape), WITHOUT making the library's streams wait
        for it: the caller packs its next chunk meanwhile and calls settle() before the library reads `out`."""
        if self.alone:
            if out.data_ptr() != x.data_ptr():
                out.copy_(x)
            return
        w = self._wire(x)
        if self.via_host:
            o = torch.empty_like(w)
            dist.all_to_all_single(o, w, group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, w, group=self.group)

    def settle(self):
        """What the collectives queued so far produce is complete before the library's next call."""
        if self.alone:
            return
        if self.lib_streams is not None:
            # Ordering by stream waits alone relies on torch's work being queued on ITS current stream being the
            # legacy default stream, against which the library's (blocking) stream is ordered as well: temporaries
            # such as edges.contiguous() handed to an import are then not reused by torch's allocator before the
            # library's copy has run. A caller inside `with torch.cuda.stream(side)` would break that silently.
            cur = torch.cuda.current_stream(self.device)
            if cur.cuda_stream != torch.cuda.default_stream(self.device).cuda_stream:
                raise RuntimeError("fastqdedup_amd.sharded: run the plan on torch's default stream (the library's "
                                   "stream is ordered against it; another current stream is not)")
            for st in self.lib_streams():
                st.wait_stream(cur)
        elif self.device.type == "cuda" and not self.via_host:
            torch.cuda.current_stream(self.device).synchronize()

    def all_gather_rows(self, x):
        """Concatenation over ranks (rank-major) of tensors that differ in dim 0."""
        if x is None:
            return None
        if self.alone:
            return x
        sizes = [int(s) for s in self.all_gather_ints([x.shape[0]])[:, 0]]
        cap = max(max(sizes), 1)
        w = self._wire(x)
        pad = torch.zeros((cap,) + tuple(x.shape[1:]), dtype=x.dtype, device=w.device)
        pad[: x.shape[0]] = w
        parts = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(parts, pad, group=self.group)
        return self._back(torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0))


def _split_by_bounds(sorted_values: torch.Tensor, bounds) -> list:
    """Rows per part of an ascending tensor cut at bounds[1:-1] (part p = [bounds[p], bounds[p+1]))."""
    if len(bounds) <= 2:
        return [int(sorted_values.shape[0])]
    cuts = torch.searchsorted(sorted_values, torch.tensor(list(bounds[1:-1]), dtype=sorted_values.dtype,
                                                          device=sorted_values.device)).tolist()
    edges = [0] + [int(c) for c in cuts] + [int(sorted_values.shape[0])]
    return [edges[i + 1] - edges[i] for i in range(len(edges) - 1)]


def cluster_keys_sharded(backend, keys, offsets=None, key_len: int = 0, weights=None, *,
                         max_distance: int = 1, use_edit_distance: bool = False,
                         method="directional", group=None, plan: str | None = None,
                         comm_via_host: bool = False, timing: bool = False) -> ShardedResult:
    """Cluster the union of every rank's keys as ONE job. ``keys`` is this rank's
    shard (device tensor or numpy array, as ``cluster_keys``). Read ids are global:
    rank r's reads follow rank r-1's. Counters are global; the id list is this rank's share."""
    if max_distance < 0:
        raise ValueError("max_distance should be non-negative")
    plan = plan or os.environ.get("FQD_SHARD_PLAN") or "segment-routed"
    if plan not in ("segment-routed", "gathered"):
        raise ValueError(f"unknown plan {plan!r}")
    if use_edit_distance:
        plan = "gathered"      # the edit search buckets keys across lengths and shifts: no per-segment routing
    comm = _Comm(group, backend.device, comm_via_host, getattr(backend, "lib_streams", None))
    rank, world = comm.rank, comm.world
    dev = backend.device
    tick = _PhaseTimer(dev) if (timing or os.environ.get("FQD_SHARD_TIMING")) else None
    method_id = METHODS[method] if isinstance(method, str) else int(method)
    metric = METRIC_EDIT if use_edit_distance else METRIC_HAMMING

    # ---- 1. common geometry --------------------------------------------------
    if offsets is None:
        nbytes = keys.numel() if hasattr(keys, "numel") else keys.size
        n_local = nbytes // key_len if key_len else 0
    else:
        n_local = (offsets.numel() if hasattr(offsets, "numel") else offsets.size) - 1
    n_seg = max_distance + 1
    owner_segments = n_seg if plan == "segment-routed" else 0
    packed = None
    # Fixed-length keys: pack at once with the DNA alphabet "ACGNT" (no pass over the bytes just
    # to learn the symbols); one flag tells every rank whether somebody met another byte, and
    # only then are the keys scanned and the symbol tables merged.
    everyone = comm.all_gather_ints([n_local, key_len if offsets is None else -1])
    n_per_rank = everyone[:, 0]
    id_bounds = [0] + [int(x) for x in np.cumsum(n_per_rank)]
    id0, n_total = id_bounds[rank], id_bounds[-1]
    lens_seen = {int(m) for n, m in everyone if n > 0}
    n_unique_local = None
    if len(lens_seen) == 1 and min(lens_seen) > 0 and not os.environ.get("FQD_SHARD_SCAN"):
        dna = np.zeros(128, dtype=np.uint8)
        dna[[ord(ch) for ch in "ACGNT"]] = 1
        fixed = lens_seen.pop()
        backend.configure(dna, fixed, False)
        g_ragged = False
        # ---- the fused way in: packed straight into owner-major slabs, the all-to-all moves the
        # slabs, the owner's collapse starts at level 2 (short keys, no weights, jobs worth it)
        n_max = int(n_per_rank.max())
        can_slabs = (plan == "segment-routed" and weights is None and hasattr(backend, "pack_into_owner_slabs")
                     and n_max >= int(os.environ.get("FQD_OWNER_SLABS_MIN_READS", 1 << 20))
                     and not os.environ.get("FQD_NO_OWNER_SLABS"))
        want_slabs = not comm.any_flag(not can_slabs)      # (weights, switches: every rank must agree)
        if want_slabs and hasattr(backend, "set_owner_routing"):
            # the slabs binned by segment 0 on every rank (search pass 0 then happens in the owner's collapse,
            # fqd_set_owner_routing) -- or on none
            backend.set_owner_routing(not comm.any_flag(not backend.owner_routing_possible(fixed, n_seg)))
        if want_slabs:
            # FQD_SHARD_CHUNKS=n: the reads leave in n pieces -- while chunk k travels (the all-to-all runs on
            # torch's / RCCL's stream, ordered behind the pack of chunk k only), the library packs chunk k + 1 on
            # its own stream. A rank's chunks arrive as n senders of their own (chunk-major), each with its id
            # base. One chunk by default: every chunk needs slabs of its own, sized for the largest chunk of the
            # job, and the relative slack of a slab grows with their number (1.16 x the reads on the wire at one
            # chunk of 50 M, 1.24 x at two, 1.35 x at four) -- on one rank two chunks cost 0.18 ms (4.65 -> 4.83 ms:
            # two packs, twice the segments at the owner's level 2) against the <= 0.35 ms of pack time they can
            # hide, and on a link-bound exchange (6 ms at two ranks) the extra bytes cost more than that.
            chunks = 1
            if os.environ.get("FQD_SHARD_CHUNKS"):
                chunks = max(1, min(int(os.environ["FQD_SHARD_CHUNKS"]), 16))

            def chunk_bounds(n):          # chunk starts on multiples of 16 reads (16-byte aligned key bytes)
                return [min(n, (n * k // chunks + 15) // 16 * 16) for k in range(chunks)] + [n]
            all_bounds = [chunk_bounds(int(n)) for n in n_per_rank]
            n_max_chunk = max(max(b[k + 1] - b[k] for k in range(chunks)) for b in all_bounds)
            geometry = Context.owner_slab_geometry(max(n_max_chunk, 1), world)
            hb, subs, cap = geometry
            parts = world * hb * subs
            n_unique_local = None
            if chunks * parts * cap * 16 <= (64 << 30):
                # Slabs WITHOUT their slack on the wire (more than one rank): every chunk's filled prefixes back to back
                # (fqd_dense_owner_slabs), moved by an all-to-all-v, the fills by an equal-split one; the owner finds the
                # slabs in the rows by the fills. With one rank nothing travels and the slabs are read where they lie.
                # (every rank must take the same exchange -- all-to-all-v + counts against equal splits: one vote,
                # like want_slabs; ranks that disagreed would issue different collectives and hang)
                dense = not comm.alone and not comm.any_flag(not (hasattr(backend, "dense_owner_slabs")
                                                                  and not os.environ.get("FQD_NO_DENSE_SLABS")))
                send = torch.empty((1 if dense else chunks, parts * cap, 4), dtype=torch.int32, device=dev)
                scur = torch.empty((chunks, parts), dtype=torch.int32, device=dev)
                recv, rcur = (send, scur) if comm.alone else (None if dense else torch.empty_like(send),
                                                              torch.empty_like(scur))
                got_rows, got_counts = [], []
                ok, totals = True, [0] * world
                mine = all_bounds[rank]
                for k in range(chunks):
                    lo, hi = mine[k] * key_len, mine[k + 1] * key_len
                    slab_k = send[0] if dense else send[k]
                    try:
                        counts = backend.pack_into_owner_slabs(keys if chunks == 1 else keys[lo:hi], key_len, world, n_seg,
                                                               geometry, slab_k, scur[k])
                    except ValueError:
                        counts = None
                    if counts is None:
                        ok = False                 # (the buffers still travel: every rank takes part in every collective)
                    else:
                        totals = [a + b for a, b in zip(totals, counts)]
                    if dense:
                        out_counts = counts if counts is not None else [0] * world
                        if counts is None:
                            # the pack gave up (a full slab, a foreign byte): its cursors say nothing about what lies in
                            # the slabs -- no dense copy; zero rows and zero fills travel so that the collectives match
                            rows_k = torch.empty((0, 4), dtype=torch.int32, device=dev)
                            fills_k = torch.zeros(parts, dtype=torch.int32, device=dev)
                        else:
                            rows_k, fills_k = backend.dense_owner_slabs(slab_k, scur[k], world, geometry, sum(out_counts))
                        in_counts = comm.exchange_counts(out_counts)
                        got_rows.append(comm.all_to_all_rows(rows_k, out_counts, in_counts))
                        got_counts.append(in_counts)
                        comm.all_to_all_into(rcur[k], fills_k)
                        del rows_k, fills_k
                    else:
                        comm.all_to_all_into(recv[k], send[k])
                        comm.all_to_all_into(rcur[k], scur[k])
                comm.settle()
                if tick:
                    tick.mark("pack-to-owner-slabs")        # (with the exchange of all chunks but the last under it)
                if not comm.any_flag(not ok):               # somebody cannot: everybody takes the general way
                    if dense:
                        recv_counts = [sum(c[s_] for c in got_counts) for s_ in range(world)]
                        recv_rows = got_rows[0] if chunks == 1 else torch.cat(got_rows, dim=0)
                        recv_geometry = (hb, subs, 0)       # (cap 0: dense rows, the "cursors" are fills)
                    else:
                        recv_counts = comm.exchange_counts(totals)
                        recv_rows, recv_geometry = recv.reshape(-1, 4), geometry
                    if tick:
                        tick.mark("all-to-all-slabs")
                    sender_id0 = [id_bounds[s_] + all_bounds[s_][k] for k in range(chunks) for s_ in range(world)]
                    # a failure on ONE rank (fills that do not add up, a rank that received nothing) must not raise
                    # here: the others would wait in the vote below for ever. It counts as "this rank cannot".
                    n_recv = int(sum(recv_counts))
                    try:
                        if dense and n_recv == 0:
                            recv_rows = torch.zeros((1, 4), dtype=torch.int32, device=dev)     # (never a null buffer)
                        n_unique_local = backend.collapse_owner_slabs(recv_rows.contiguous(), rcur.reshape(-1),
                                                                      world * chunks, rank, recv_geometry, sender_id0,
                                                                      max(n_total, 1), n_recv, n_seg)
                    except (ValueError, RuntimeError) as e:
                        n_unique_local = None
                        if os.environ.get("FQD_DEBUG") or os.environ.get("FQD_SHARD_TIMING"):
                            print(f"[fqd] rank {rank}: the owner-slab collapse failed ({e}); general way", file=sys.stderr)
                    del recv_rows
                    if comm.any_flag(n_unique_local is None):
                        n_unique_local = None      # a bucket overflowed somewhere: once more, the general way
                        if os.environ.get("FQD_DEBUG") or os.environ.get("FQD_SHARD_TIMING"):
                            # (at 5-8 ranks an owner's level 2 has 2^15 buckets at most -- ~1500 reads each at 50 M
                            # received reads: mostly-unique data overfills the dedupe's 1024-slot table)
                            print(f"[fqd] rank {rank}: the owner-slab collapse gave up (a bucket's table or a slab was "
                                  f"full somewhere); all ranks repeat the way in the general way", file=sys.stderr)
                    if tick:
                        tick.mark("collapse")
                del send, scur, recv, rcur, got_rows
        if n_unique_local is None:
            foreign = 0
            try:
                packed = backend.pack_by_owner(keys, offsets, key_len, world, id0, weights, n_segments=owner_segments)
            except ValueError:
                foreign = 1
            if comm.any_flag(bool(foreign)):
                packed = None
    if packed is None and n_unique_local is None:
        present, max_len, ragged = backend.scan(keys, offsets, key_len)
        everyone = comm.all_gather_ints([n_local, max_len, int(ragged)])
        lens_seen = {int(m) for n, m, _ in everyone if n > 0}
        g_ragged = bool(everyone[:, 2].any()) or len(lens_seen) > 1
        g_max_len = int(everyone[:, 1].max()) if len(everyone) else 0
        p = torch.from_numpy(np.ascontiguousarray(present, dtype=np.uint8)).to(dev).to(torch.int32)
        g_present = comm.all_reduce_max(p).to(torch.uint8).cpu().numpy()
        backend.configure(g_present, g_max_len, g_ragged)
    if tick:
        tick.mark("geometry")

    if plan == "gathered":
        return _gathered(backend, comm, tick, packed, keys, offsets, key_len, weights, max_distance, metric,
                         method_id, g_ragged, id0, n_local, n_total)

    # ---- 2. reads to the owner of their segment 0; collapse -------------------------
    if n_unique_local is None:
        if packed is None:
            packed = backend.pack_by_owner(keys, offsets, key_len, world, id0, weights, n_segments=n_seg)
        s_recs, s_lens, s_ids, s_w, send_counts = packed
        del packed
        if tick:
            tick.mark("pack+group-by-owner")
        recv_counts = comm.exchange_counts(send_counts)
        r_recs = comm.all_to_all_rows(s_recs, send_counts, recv_counts)
        r_ids = comm.all_to_all_rows(s_ids, send_counts, recv_counts)
        r_lens = comm.all_to_all_rows(s_lens, send_counts, recv_counts) if g_ragged else None
        r_w = comm.all_to_all_rows(s_w, send_counts, recv_counts)
        del s_recs, s_ids, s_lens, s_w
        # (s_ids / r_ids are None when the records carry the read index in their padding word.)
        # The received rows are ALREADY in global id order: all-to-all delivers source ranks in rank
        # order, rank r's ids precede rank r+1's, and every source sent its rows in id order.
        if tick:
            tick.mark("all-to-all-reads")
        seg_rows = [0] + [int(x) for x in np.cumsum(recv_counts)]
        n_unique_local = backend.collapse_resident(r_recs, r_lens, r_w, r_ids, seg_rows=seg_rows,
                                                   seg_id0=id_bounds[:-1], id_limit=max(n_total, 1))
        del r_recs, r_ids, r_lens, r_w
        if tick:
            tick.mark("collapse")
    uid_bounds = [0] + [int(x) for x in np.cumsum(comm.all_gather_ints([n_unique_local])[:, 0])]
    uid0, n_unique = uid_bounds[rank], uid_bounds[-1]
    if n_unique >= 2**31 - 16:
        raise ValueError("more than 2^31 unique keys in one job")

    # ---- 3. search: pass 0 at home, pass s on the owner of segment s -----------------
    found = []
    if n_seg >= 1 and max_distance > 0:
        e0 = backend.local_edges(max_distance, 0, 1)
        found.append(e0 + uid0 if e0.shape[0] else e0)
        if tick:
            tick.mark("search-pass-0")
        for s in range(1, n_seg):
            q_recs, q_lens, q_uids, q_counts = backend.unique_by_segment(world, n_seg, s, uid0)
            back_counts = comm.exchange_counts(q_counts)
            x_recs = comm.all_to_all_rows(q_recs, q_counts, back_counts)
            x_uids = comm.all_to_all_rows(q_uids, q_counts, back_counts)
            x_lens = comm.all_to_all_rows(q_lens, q_counts, back_counts) if g_ragged else None
            del q_recs, q_lens, q_uids
            if tick:
                tick.mark("route-unique")
            found.append(backend.routed_edges(x_recs, x_lens, x_uids, max_distance, s))
            del x_recs, x_lens, x_uids
            if tick:
                tick.mark("search-routed")
    mine = torch.cat(found, dim=0) if found else torch.empty((0, 2), dtype=torch.int32, device=dev)

    # ---- 4. every edge to the owner of its ends; union-find at home; labels across ranks --------------
    # No rank ever holds the job's edge list, and no rank runs a union-find over the job's keys (rounds 1-3 all-gathered
    # every edge and labelled all U keys on EVERY rank: per-rank work that grew with the number of ranks -- 600 MB and
    # ~75 M edges per rank at 8 x 25 M reads of 300 nt). Per rank and step now: its own share of the edges.
    #   4a  an edge (u, v) goes to the owner of u (uids are contiguous per rank); there it is a HOME edge (v is its
    #       own as well) or a CROSS edge
    #   4b  union-find over the rank's own rows and home edges: local components, named by their smallest row
    #   4c  a cross edge becomes an edge between LOCAL COMPONENTS: u -> its component at home, then on to the owner of v,
    #       v -> its component there; both owners keep a copy (mine, theirs)
    #   4d  the smallest uid of a cluster reaches all of its local components by min-label rounds over those copies
    #       (a message per cross edge and round, until no label moves anywhere: clusters are a handful of keys, the
    #       rounds as many as a cluster has ranks in a row)
    #   4e  clusters = local components whose label is their own uid, summed over the ranks
    n_edges = int(comm.all_gather_ints([int(mine.shape[0])])[:, 0].sum())
    no_home = bool(os.environ.get("FQD_NO_HOME_CLUSTERS"))
    dev_t = mine.device
    if comm.alone and not no_home:
        # one rank: every edge is a home edge, every cluster a home cluster
        home_edges = mine
        dropped_here = torch.empty(0, dtype=torch.int64, device=dev_t)
        kept_owned, n_kept_owned = backend.finish_owner_home(home_edges, method_id, dropped_here, max(n_total, 1))
        n_clusters = int(backend.home_components)
        if tick:
            for name in ("gather-edges+label", "fetch-cluster-keys", "dissect", "verdicts-home"):
                tick.mark(name)
    else:
        inner = torch.tensor(uid_bounds[1:-1], dtype=torch.int64, device=dev_t)

        def owner_of(uids):
            return torch.bucketize(uids.to(torch.int64), inner, right=True)

        def to_owners(rows, dest):
            """rows grouped by destination rank (stable) and moved: what this rank receives."""
            order = torch.sort(dest, stable=True)[1]
            counts = torch.bincount(dest, minlength=world).tolist()
            got_counts = comm.exchange_counts(counts)
            return comm.all_to_all_rows(rows[order].contiguous(), counts, got_counts)

        # 4a
        at_u = to_owners(mine.to(torch.int32), owner_of(mine[:, 0]))
        v_owner = owner_of(at_u[:, 1])
        is_home = v_owner == rank
        home_local = (at_u[is_home] - uid0).to(torch.int32)
        cross = at_u[~is_home]                                        # u is mine, v is not
        cross_dest = v_owner[~is_home]
        del at_u
        # 4b
        lab, _n_local = backend.local_labels(home_local)
        lab = lab.to(torch.int64)
        n_rows = int(lab.shape[0])
        # 4c: (component of u at home as a uid, v) -> owner of v -> (component of u, component of v)
        cu = lab[(cross[:, 0] - uid0).long()] + uid0
        there = to_owners(torch.stack([cu.to(torch.int32), cross[:, 1]], dim=1), cross_dest)
        cv = lab[(there[:, 1] - uid0).long()] + uid0                 # (there: [component elsewhere, v mine])
        mine_theirs = torch.stack([cv, there[:, 0].to(torch.int64)], dim=1)
        back = to_owners(torch.stack([there[:, 0], cv.to(torch.int32)], dim=1), owner_of(there[:, 0]))
        pairs = torch.cat([mine_theirs, back.to(torch.int64)], dim=0)   # (my component, a component elsewhere) per cross edge end
        del there, back, mine_theirs
        touched = torch.zeros(max(n_rows, 1), dtype=torch.bool, device=dev_t)
        touched[(pairs[:, 0] - uid0).long()] = True
        if no_home:                    # (tests: every cluster with an edge is dealt out, none dissected in place)
            touched[lab[home_local.reshape(-1).long()]] = True
        # 4d
        label = torch.arange(n_rows, dtype=torch.int64, device=dev_t) + uid0     # (of component roots; other rows unused)
        pair_dest = owner_of(pairs[:, 1])
        pair_order = torch.sort(pair_dest, stable=True)[1]
        pair_counts = torch.bincount(pair_dest, minlength=world).tolist()
        pair_got = comm.exchange_counts(pair_counts)
        theirs_sorted = pairs[pair_order, 1]
        mine_sorted = (pairs[pair_order, 0] - uid0).long()
        # (three rounds between two looks at "did anything move": a look is an all-reduce and a host round trip, a round
        # costs little -- and a cluster of a handful of keys spans two or three ranks in a row)
        while True:
            for k in range(3):
                if k == 2:
                    before = label.clone()
                msg = torch.stack([theirs_sorted, label[mine_sorted]], dim=1)     # (their component, my label)
                got = comm.all_to_all_rows(msg, pair_counts, pair_got)
                if got.shape[0]:
                    label.scatter_reduce_(0, (got[:, 0] - uid0).long(), got[:, 1], reduce="amin", include_self=True)
            # converged when the LAST round of a batch moved nothing anywhere (a round that moves nothing on any rank
            # sends the same messages as the round before it: every later round is idle too)
            if not comm.any_flag(bool((label != before).any())):
                break
        del before
        # 4e
        is_root = lab == torch.arange(n_rows, dtype=torch.int64, device=dev_t)
        mine_clusters = int((is_root & (label == torch.arange(n_rows, dtype=torch.int64, device=dev_t) + uid0)).sum())
        n_clusters = int(comm.all_gather_ints([mine_clusters])[:, 0].sum())
        if tick:
            tick.mark("gather-edges+label")

        # ---- 5. clusters with keys on several ranks: keys and edges to the cluster's host; dissect; verdicts back ----
        # host = label mod ranks. Every rank PUSHES what it holds of such clusters (rows and edges; no request, no
        # list of clusters anywhere); clusters that live on one rank (no cross edge) are dissected there, in place.
        row_label = label[lab]                                          # every row's cluster
        row_spans = touched[lab]
        span_rows = torch.nonzero(row_spans).reshape(-1)               # ascending rows
        any_spanning = comm.any_flag(bool(span_rows.shape[0]))
        home_keep = ~row_spans[home_local[:, 0].long()] if home_local.shape[0] else torch.zeros(0, dtype=torch.bool, device=dev_t)
        home_edges = home_local[home_keep]
        if any_spanning:
            row_host = (row_label[span_rows] % world)
            a_recs, a_lens, a_counts = backend.gather_unique(span_rows.to(torch.int32))
            order = torch.sort(row_host, stable=True)[1]
            r_counts = torch.bincount(row_host, minlength=world).tolist()
            r_got = comm.exchange_counts(r_counts)
            t_uids = comm.all_to_all_rows((span_rows[order] + uid0).to(torch.int64), r_counts, r_got)
            t_recs = comm.all_to_all_rows(a_recs[order], r_counts, r_got)
            t_counts = comm.all_to_all_rows(a_counts[order], r_counts, r_got)
            t_lens = comm.all_to_all_rows(a_lens[order], r_counts, r_got) if g_ragged else None
            del a_recs, a_lens, a_counts
            # the clusters' edges: home edges of spanning clusters and the cross edges (each held once, at the owner of u)
            e_home = home_local[~home_keep].to(torch.int64) + uid0
            e_all = torch.cat([e_home, cross.to(torch.int64)], dim=0)
            e_host = row_label[(e_all[:, 0] - uid0).long()] % world
            t_edges = to_owners(e_all, e_host)
            if tick:
                tick.mark("fetch-cluster-keys")
            # rows by uid; the edges' ends become positions in that order
            t_sorted, t_perm = torch.sort(t_uids)
            sub_edges = torch.searchsorted(t_sorted, t_edges.reshape(-1)).reshape(-1, 2).to(torch.int32)
            verdict = backend.dissect_subgraph(t_recs[t_perm].contiguous(), t_lens[t_perm].contiguous() if t_lens is not None else None,
                                               t_counts[t_perm].contiguous(), sub_edges, method_id)
            dropped = t_sorted[verdict == 0]
            del t_recs, t_lens, t_counts, sub_edges, t_edges
            if tick:
                tick.mark("dissect")
            drop_counts = _split_by_bounds(dropped, uid_bounds)
            dropped_counts = comm.exchange_counts(drop_counts)
            dropped_here = comm.all_to_all_rows(dropped, drop_counts, dropped_counts)
        else:
            dropped_here = torch.empty(0, dtype=torch.int64, device=dev_t)
            if tick:
                for name in ("fetch-cluster-keys", "dissect"):
                    tick.mark(name)
        kept_owned, n_kept_owned = backend.finish_owner_home(home_edges, method_id, dropped_here - uid0, max(n_total, 1))
        if tick:
            tick.mark("verdicts-home")

    # ---- 6. kept ids to the rank that read them ------------------------------------------
    out_counts = _split_by_bounds(kept_owned, id_bounds)
    in_counts = comm.exchange_counts(out_counts)
    kept_mine = comm.all_to_all_rows(kept_owned, out_counts, in_counts)
    if world > 1:
        # G ascending runs -> one ascending list: mark the window's reads, list the marked ones
        window = torch.zeros(max(n_local, 1), dtype=torch.bool, device=kept_mine.device)
        window[kept_mine - id0] = True
        kept_mine = window.nonzero().reshape(-1)[: kept_mine.shape[0]] + id0
    n_kept = int(comm.all_gather_ints([n_kept_owned])[:, 0].sum())
    phases = None
    if tick:
        tick.mark("kept-ids-home")
        phases = tick.done()
    return ShardedResult(kept_mine, n_total, n_unique, n_edges, int(n_clusters), n_kept, "segment-routed", phases)


def _gathered(backend, comm, tick, packed, keys, offsets, key_len, weights, max_distance, metric, method_id,
              g_ragged, id0, n_local, n_total) -> ShardedResult:
    rank, world = comm.rank, comm.world
    # ---- 2. all copies of a key to its owner rank ------------------------------
    if packed is None:
        packed = backend.pack_by_owner(keys, offsets, key_len, world, id0, weights)
    s_recs, s_lens, s_ids, s_w, send_counts = packed
    del packed
    if tick:
        tick.mark("pack+group-by-owner")
    recv_counts = comm.exchange_counts(send_counts)
    r_recs = comm.all_to_all_rows(s_recs, send_counts, recv_counts)
    r_ids = comm.all_to_all_rows(s_ids, send_counts, recv_counts)
    r_lens = comm.all_to_all_rows(s_lens, send_counts, recv_counts) if g_ragged else None
    r_w = comm.all_to_all_rows(s_w, send_counts, recv_counts)
    # Received rows are in global id order (see the segment-routed plan): the collapse's stable
    # grouping makes the smallest global id the head of each run.
    if tick:
        tick.mark("all-to-all-reads")
    urecs, ulens, ucounts, ufirst = backend.collapse_packed(r_recs, r_lens, r_w, r_ids)
    if tick:
        tick.mark("collapse")

    # ---- 3. whole unique table on every rank; search this rank's bucket shard ---
    g_recs = comm.all_gather_rows(urecs)
    g_lens = comm.all_gather_rows(ulens) if g_ragged else None
    g_counts = comm.all_gather_rows(ucounts)
    g_first = comm.all_gather_rows(ufirst)
    if tick:
        tick.mark("all-gather-unique")
    edges = backend.find_edges(g_recs, g_lens, g_counts, g_first, max_distance, metric, rank, world)
    if tick:
        tick.mark("find-edges")

    # ---- 4. all edges everywhere; components + dissection ------------------------
    # Every rank labels and dissects the whole graph but LISTS only the kept ids among its own
    # reads [id0, id0 + n_local): what its pass 2 would need.
    g_edges = comm.all_gather_rows(edges)
    kept, n_clusters, n_kept = backend.finish(g_edges.contiguous(), method_id, id0, id0 + n_local)
    phases = None
    if tick:
        tick.mark("gather-edges+finish")
        phases = tick.done()
    return ShardedResult(kept, n_total, int(g_recs.shape[0]), int(g_edges.shape[0]), int(n_clusters),
                         int(n_kept), "gathered", phases)

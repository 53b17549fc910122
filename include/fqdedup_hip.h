/*
 * fqdedup_hip.h -- C ABI of libfqdedup_hip.so: the MI355X (gfx950) clustering
 * hot path of fastqdedup. Plain pointers and sizes only; no CPython, no torch.
 *
 * What each entry point replaces in the reference (/root/reference):
 *
 *   fqd_pack_keys        storage half of TrieNode_AddSequence / Trie.add_sequence
 *                        (src/fastqdedup/_triemodule.c:222-288, :677-706), driven by
 *                        the insert loop of deduplicate_cluster (__init__.py:242-252)
 *   fqd_collapse         duplicate-count half of TrieNode_AddSequence
 *                        (_triemodule.c:235-239, :261-264) + Trie.number_of_sequences
 *   fqd_find_edges       TrieNode_FindNearest + within_hamming_distance /
 *                        within_edit_distance (_triemodule.c:380-495, distances.h:8-88)
 *   fqd_components       the BFS of Trie.pop_cluster (_triemodule.c:778-897):
 *                        one popped cluster == one connected component
 *   fqd_dissect          cluster_dissection_{highest_count,adjacency,directional}
 *                        (__init__.py:60-130) and deduplicated_set.add (:276)
 *   fqd_cluster          the caller loop __init__.py:266-281 in one call
 *   fqd_get_kept_read_ids  first holder of every kept key in input order, i.e. what
 *                        filter_fastq_files_on_set selects (__init__.py:189-206)
 *   fqd_within_distance  _distance.within_distance (_distancemodule.c:46-93)
 *   fqd_contains         Trie.contains_sequence (_triemodule.c:730-758)
 *
 * Threading: one fqd_ctx = one device + one HIP stream; calls on one context
 * must be serialised (the reference holds the GIL for every call,
 * SURVEY.md section 8b). Every function returns FQD_OK or a negative code;
 * fqd_last_error() gives the message. There is NO CPU fallback: without a
 * usable gfx950 device fqd_create fails with FQD_E_DEVICE.
 *
 * Memory: `mem` says where caller buffers live (FQD_HOST or FQD_DEVICE).
 * Device buffers must be 16-byte aligned and stay valid until the call returns; device SOURCES of
 * the fqd_import_* calls are copied on the context's stream without a host wait, so they must stay
 * unchanged until the context has been synchronised (any call that returns a count does that) or
 * until work of the legacy default stream, which orders itself behind the context's stream, reuses
 * them -- what a torch caller does.
 * Results are copied into caller-provided buffers.
 */
#ifndef FQDEDUP_HIP_H
#define FQDEDUP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FQD_OK             0
#define FQD_E_NOMEM       -1  /* MemoryError  */
#define FQD_E_VALUE       -2  /* ValueError   */
#define FQD_E_LOOKUP      -3  /* LookupError  */
#define FQD_E_RUNTIME     -4  /* RuntimeError */
#define FQD_E_DEVICE      -5  /* RuntimeError: HIP error / no device */
#define FQD_E_STATE       -6  /* RuntimeError: stage called out of order */

#define FQD_HOST   0
#define FQD_DEVICE 1
/* fqd_import_packed / fqd_import_unique only: a device buffer the context READS IN PLACE (records
 * and lengths are not copied). The caller keeps it alive and unchanged until the stage that
 * consumes it has returned (fqd_collapse; fqd_find_edges* / fqd_dissect); the context lets go of
 * it the next time it needs that buffer. Records must be 16-byte aligned. */
#define FQD_DEVICE_BORROW 2

#define FQD_METRIC_HAMMING 0
#define FQD_METRIC_EDIT    1

#define FQD_METHOD_HIGHEST_COUNT 0
#define FQD_METHOD_ADJACENCY     1
#define FQD_METHOD_DIRECTIONAL   2

typedef struct fqd_ctx fqd_ctx;

typedef struct fqd_summary {
    uint64_t n_reads;      /* keys handed in                                  */
    uint64_t n_counted;    /* sum of weights == Trie.number_of_sequences      */
    uint64_t n_unique;     /* distinct keys with weight > 0                   */
    uint64_t n_edges;      /* unordered key pairs within distance             */
    uint64_t n_clusters;   /* == number of pop_cluster calls in the reference */
    uint64_t n_kept;       /* == len(deduplicated_set)                        */
} fqd_summary;

/* Which way the last job took through the library -- so that a caller can SEE a fast path that stopped engaging
 * (unusual data sends a call back to the stage-by-stage way: correct, but up to twice the time). Bits of
 * fqd_get_route(): */
#define FQD_ROUTE_FUSED_PACK      0x0001u  /* fqd_cluster_keys: the pack kernel wrote level 1 of the collapse itself     */
#define FQD_ROUTE_COMPACT_RECORDS 0x0002u  /* ... with 12-byte records behind it (keys with an N on the side path)       */
#define FQD_ROUTE_PASS0_IN_COLLAPSE 0x0004u /* ... reads binned by segment 0, search pass 0 done by the compaction       */
#define FQD_ROUTE_RESTARTED       0x0008u  /* a fused attempt was abandoned (foreign byte, full slab or LDS table) and the
                                            * job started over the stage-by-stage way                                    */
#define FQD_ROUTE_COLLAPSE_LDS    0x0010u  /* collapse: records of one uint4 partitioned and deduplicated in LDS          */
#define FQD_ROUTE_COLLAPSE_PAIRS  0x0020u  /* collapse: (hash, position) pairs for longer fixed-length records            */
#define FQD_ROUTE_COLLAPSE_SORT   0x0040u  /* collapse: radix sort + verification (ragged keys, small inputs, overflow)    */
#define FQD_ROUTE_SEARCH_GROUPED  0x0100u  /* neighbour search: partition + candidates + verification                     */
#define FQD_ROUTE_SEARCH_SORT     0x0200u  /* neighbour search: radix sort + pair kernel                                   */
#define FQD_ROUTE_SEARCH_EDIT     0x0400u  /* the Levenshtein search proper ran (grouped or sorted)                        */
#define FQD_ROUTE_SEARCH_RETRIED  0x0800u  /* the search ran again: an edge / candidate buffer or a slab was too small     */
#define FQD_ROUTE_PASS0_CONTINUED 0x1000u  /* the search took pass 0 from the collapse and ran the other passes only       */
#define FQD_ROUTE_SPILL_LIST      0x2000u  /* the fused collapse ran with its spill list: the context has met keys with
                                            * hundreds of copies (full slabs); no search pass 0 in the compaction then     */
#define FQD_ROUTE_SEARCH_REFINED  0x4000u  /* crowded segment values (thousands of keys sharing one) were matched on finer
                                            * segments instead of pairwise                                                 */
#define FQD_ROUTE_ONE_KERNEL_COLLAPSE 0x8000u /* dedupe + compaction (+ search pass 0) of the compact records ran as ONE
                                            * persistent kernel: no tmp rows between the LDS table and the unique table   */
#define FQD_ROUTE_SEARCH_TILES    0x10000u /* crowded segment values that the finer segments could not split (a family
                                            * varying inside one piece) were compared ALL PAIRS, tiled over the GPU         */
int fqd_get_route(const fqd_ctx *ctx, uint32_t *route);

/* Packed-key geometry chosen by fqd_pack_keys (DESIGN.md "data layout"). */
typedef struct fqd_shape {
    uint32_t planes;       /* bit planes per base: ceil(log2(alphabet size))  */
    uint32_t words;        /* 32-base words per plane: ceil(max_len/32)       */
    uint32_t stride_words; /* u32 words per record (multiple of 4)            */
    uint32_t max_len;      /* longest key, bases                              */
    uint32_t ragged;       /* 1 when key lengths differ                       */
    uint32_t alphabet_size;
    uint8_t  alphabet[128];/* symbols in ASCII order; code = index            */
} fqd_shape;

int         fqd_device_count(void);
const char *fqd_global_error(void);             /* message of a failed fqd_create */
int         fqd_create(int device, fqd_ctx **out);
void        fqd_destroy(fqd_ctx *ctx);
const char *fqd_last_error(const fqd_ctx *ctx);
int         fqd_synchronize(fqd_ctx *ctx);
/* The context's HIP stream (a hipStream_t), so that a caller working on other streams -- torch,
 * RCCL -- can order its work against the context's with events instead of host synchronisation. */
void       *fqd_get_stream(fqd_ctx *ctx);

/* ---- stage 1: keys -> bit-plane records + 32-bit hashes --------------------
 * bytes: concatenated ASCII keys. offsets: n+1 byte offsets, or NULL for n keys
 * of fixed_len bytes each. Bytes >= 128 are a FQD_E_VALUE (the reference
 * refuses non-ASCII keys, _triemodule.c:684-688). */
int fqd_pack_keys(fqd_ctx *ctx, const uint8_t *bytes, const uint64_t *offsets, uint64_t n,
                  uint32_t fixed_len, int mem);
/* Force the alphabet (symbols present, as a 128-entry 0/1 table), longest key
 * and raggedness before fqd_pack_keys, so that several contexts (ranks) share
 * one record geometry. present == NULL restores auto-detection. */
int fqd_configure(fqd_ctx *ctx, const uint8_t *present128, uint32_t max_len, int ragged);
/* What a buffer of keys needs: symbols present, longest key, raggedness. */
int fqd_scan_keys(fqd_ctx *ctx, const uint8_t *bytes, const uint64_t *offsets, uint64_t n,
                  uint32_t fixed_len, int mem, uint8_t *present128, uint32_t *max_len, int *ragged);
int fqd_get_shape(const fqd_ctx *ctx, fqd_shape *out);

/* ---- stage 2: exact duplicates -> unique keys, counts, first holders -------
 * weights: per key multiplicity (0 = present but not counted: a read that
 * failed the quality filter still is a "first holder", __init__.py:201-206),
 * NULL = 1 each. read_ids: caller's ids for the keys (NULL = 0..n-1). */
int fqd_collapse(fqd_ctx *ctx, const uint32_t *weights, const uint64_t *read_ids, int mem,
                 uint64_t *n_unique);

/* ---- stage 3: all pairs of unique keys within max_distance -----------------
 * Only buckets with (bucket_hash % n_shards) == shard are searched, so that the
 * ranks of a multi-GPU job split the search; (0, 1) searches everything. */
int fqd_find_edges(fqd_ctx *ctx, int max_distance, int metric, uint32_t shard, uint32_t n_shards,
                   uint64_t *n_edges);

/* ---- stage 4 / 5 ----------------------------------------------------------- */
int fqd_components(fqd_ctx *ctx, uint64_t *n_clusters);
int fqd_dissect(fqd_ctx *ctx, int method, uint64_t *n_kept);

/* ---- all of 2..5 ----------------------------------------------------------- */
int fqd_cluster(fqd_ctx *ctx, const uint32_t *weights, const uint64_t *read_ids, int mem,
                int max_distance, int metric, int method, fqd_summary *out);

/* ---- 1 + 2 in one call: the insert loop __init__.py:242-252 for one batch of keys. Short
 * fixed-length keys take the fused way in (see fqd_cluster_keys); search_segments > 0 announces a
 * Hamming search with that many pigeonhole segments, whose segment hashes the collapse then writes
 * on its way out. Afterwards the context holds the unique table (not the packed reads). */
int fqd_pack_collapse(fqd_ctx *ctx, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, uint32_t fixed_len,
                      int mem, const uint32_t *weights, const uint64_t *read_ids, int aux_mem,
                      uint32_t search_segments, uint64_t *n_unique);

/* ---- all of 1..5: the whole hot path in one call ------------------------------
 * Replaces the reference's per-read Trie.add_sequence loop plus the cluster loop
 * (src/fastqdedup/__init__.py:240-281) for one batch of keys: fqd_pack_keys + fqd_cluster, with
 * the same arguments and errors (`mem` places bytes/offsets, `aux_mem` weights/read_ids). Because
 * the key bytes are at hand for the whole call, short fixed-length keys (records of one uint4)
 * take a faster way in: the pack kernel partitions its records straight into the collapse, and
 * the packed reads are never written in read order. After this call the context holds the unique
 * table and everything behind it, but NOT the packed reads: the fqd_export_packed_* entry points and
 * fqd_collapse need a fqd_pack_keys first (FQD_E_STATE otherwise). */
int fqd_cluster_keys(fqd_ctx *ctx, const uint8_t *bytes, const uint64_t *offsets, uint64_t n,
                     uint32_t fixed_len, int mem, const uint32_t *weights, const uint64_t *read_ids,
                     int aux_mem, int max_distance, int metric, int method, fqd_summary *out);

/* ---- results --------------------------------------------------------------- */
/* The ids this context LISTS are the first holders that fall into [lo, hi) (default: all).
 * A rank of a multi-GPU job lists the ids of its own reads -- what its pass 2 needs.
 * n_kept counts every kept key, n_listed those inside the window. */
int fqd_set_id_window(fqd_ctx *ctx, uint64_t lo, uint64_t hi);
int fqd_get_kept_count(fqd_ctx *ctx, uint64_t *n_kept, uint64_t *n_listed);
/* n_listed ids, ascending: the first holder of every kept key (inside the id window). */
int fqd_get_kept_read_ids(fqd_ctx *ctx, uint64_t *out, int mem);
/* Optional: announce a DEVICE buffer of `capacity` ids BEFORE fqd_dissect / fqd_cluster; when it can
 * hold any outcome (capacity >= min(id range, unique keys)) the ascending list is written there
 * directly and fqd_get_kept_read_ids(ctx, that same pointer, FQD_DEVICE) has nothing left to
 * copy. NULL switches it off (do that before the buffer goes away). */
int fqd_set_kept_output(fqd_ctx *ctx, uint64_t *out_device, uint64_t capacity);
/* Per unique key u in [0, n_unique): first holder, count, component label
 * (= smallest u of the component), kept flag. Any pointer may be NULL. */
int fqd_get_unique_table(fqd_ctx *ctx, uint64_t *first_ids, uint32_t *counts, uint32_t *labels,
                         uint8_t *kept, int mem);

/* ---- exchange (multi-GPU: the caller moves these buffers with RCCL) -------- */
/* Packed reads of stage 1: recs n*stride_words u32, lens n u32, hashes n u32. Words of a record past
 * planes * words are padding: zero, except that a ragged key's record with such a word may hold the key's length in its
 * LAST one (nothing that reads key words looks there; records that hold the same key hold the same length). */
int fqd_export_packed(fqd_ctx *ctx, uint32_t *recs, uint32_t *lens, uint32_t *hashes, int mem);
int fqd_import_packed(fqd_ctx *ctx, const uint32_t *recs, const uint32_t *lens, uint64_t n, int mem);
/* The same reads grouped by owner = key_hash % n_parts (part 0 first), each part in read order:
 * the send buffers of the all-to-all. ids[i] = id0 + read index. recs/lens/ids/weights_out are
 * DEVICE buffers of n rows (lens, weights_out may be NULL); weights (device, NULL = 1 each) are
 * carried along; counts is a HOST array of n_parts row counts. */
int fqd_export_packed_by_owner(fqd_ctx *ctx, uint32_t n_parts, uint64_t id0, const uint32_t *weights,
                               uint32_t *recs, uint32_t *lens, uint64_t *ids, uint32_t *weights_out,
                               uint64_t *counts, int mem);
/* Same, with owner = hash(segment `segment` of the n_segments-way pigeonhole split, key length)
 * % n_parts. Every copy of a key and every pair of keys that agree on that segment meet on one
 * rank: the receiving rank collapses its reads AND runs search pass `segment` with no further key
 * movement (sharded.py, "segment-routed" plan). */
int fqd_export_packed_by_segment(fqd_ctx *ctx, uint32_t n_parts, uint32_t n_segments, uint32_t segment,
                                 uint64_t id0, const uint32_t *weights, uint32_t *recs, uint32_t *lens,
                                 uint64_t *ids, uint32_t *weights_out, uint64_t *counts, int mem);
/* ids == NULL above: no id array is produced; instead every exported record carries the read's
 * index on THIS rank in its first padding word (geometries with stride_words > planes * words:
 * FQD_E_VALUE otherwise). The receiver collapses such reads with fqd_collapse_received, naming the
 * row range each sender's reads occupy in its packed buffer (seg_rows, n_seg + 1 offsets, HOST)
 * and the sender's id base (seg_id0, HOST): id = seg_id0[s] + carried index; id_limit bounds all
 * ids (~0 = unknown). 16 instead of 24 bytes per read on the wire for keys of <= 32 nt. A buffer
 * lent with FQD_DEVICE_BORROW may get its padding words cleared. Otherwise as fqd_collapse. */
int fqd_collapse_received(fqd_ctx *ctx, const uint32_t *weights, const uint64_t *seg_rows,
                          const uint64_t *seg_id0, uint32_t n_seg, uint64_t id_limit, int mem,
                          uint64_t *n_unique);
/* The fused way in across ranks, for fixed-length keys whose record is one uint4 (<= 32 nt over
 * "ACGNT") and reads without weights: fqd_pack_to_owner_slabs packs this rank's keys straight into
 * owner-major slabs -- bin = owner * hash_bins + top hash bits of the record, `subs` slabs of `cap`
 * records per bin, owner by the rule of fqd_set_owner_rule (n_parts, n_segments, segment) -- so an
 * owner's share is ONE contiguous range of hash_bins * subs slabs (capacity included) of
 * slabs_out, with cursors_out[p] = first free slot of slab p (slot numbers count from the start of
 * slabs_out); every record carries its read's index on this rank. counts (HOST, n_parts) = reads
 * per owner. fqd_collapse_owner_slabs is the receiving side: `slabs` holds, sender by sender, the
 * ranges the n_senders ranks addressed to owner my_part (read in place), `cursors` their cursor
 * tables; sender_id0 (HOST) the id base of each sender's reads, id_limit a bound on all ids,
 * n_reads the reads received in all. A "sender" is one fqd_pack_to_owner_slabs call: a rank that
 * packs its reads in several pieces (so that a piece travels while the next is packed) sends
 * several, in any order -- a sender's reads lie below the next larger id base. The collapse starts
 * at level 2: the senders' bins are its
 * level 1. *done = 0 (either call): not applicable or a slab overflowed -- nothing was produced,
 * take the general way (fqd_pack_keys + fqd_export_packed_by_segment / fqd_collapse_received).
 * fqd_owner_slab_geometry gives (hash_bins, subs, cap) for ranks that pack at most n_max reads each;
 * all ranks of a job must use the same values. Replaces, per batch and rank, the reference's insert
 * loop __init__.py:242-252 like fqd_pack_keys + fqd_collapse do. */
/* Owner slabs binned by SEGMENT 0 of the key (what the owner rule looks at) instead of the whole key: the owner's
 * compaction then reports the pairs of search pass 0 itself, as fqd_cluster_keys does on one GPU, and
 * fqd_find_edges_segments(0, 1) has nothing left to search. Every rank of the job must bin alike: ask every rank
 * (possible: keys of <= 32 symbols with a segment 0 of >= 8, compact records, no crowded segment value met before)
 * and switch it on everywhere or nowhere; the setting holds for the fqd_pack_to_owner_slabs /
 * fqd_collapse_owner_slabs calls that follow. */
int fqd_owner_routing_possible(const fqd_ctx *ctx, uint32_t key_len, uint32_t n_segments, int *possible);
int fqd_set_owner_routing(fqd_ctx *ctx, int enable);
int fqd_owner_slab_geometry(uint64_t n_max, uint32_t n_parts, uint32_t *hash_bins, uint32_t *subs, uint32_t *cap);
int fqd_pack_to_owner_slabs(fqd_ctx *ctx, const uint8_t *bytes, uint64_t n, uint32_t fixed_len, int mem, uint32_t n_parts,
                            uint32_t n_segments, uint32_t segment, uint32_t hash_bins, uint32_t subs, uint32_t cap,
                            uint32_t *slabs_out, uint32_t *cursors_out, uint64_t *counts, int *done);
/* Owner slabs WITHOUT their slack on the wire: the filled prefixes of a sender's slabs back to back (rows_out, device,
 * room for the reads packed) and every slab's fill (fills_out, device, n_parts * hash_bins * subs words). The caller
 * moves rows by an all-to-all-v (reads per owner: fqd_pack_to_owner_slabs' counts) and the fills by an equal-split
 * all-to-all; the owner then calls fqd_collapse_owner_slabs with cap = 0, `slabs` = the received rows (sender by
 * sender), `cursors` = the received fills. rows_capacity: rows_out's room in rows -- the fills come from the cursors, and
 * rows behind the capacity are dropped, never written (a caller whose pack gave up passes cursors it cannot trust). */
int fqd_dense_owner_slabs(fqd_ctx *ctx, const uint32_t *slabs, const uint32_t *cursors, uint32_t n_parts, uint32_t hash_bins,
                          uint32_t subs, uint32_t cap, uint32_t *rows_out, uint64_t rows_capacity, uint32_t *fills_out);
int fqd_collapse_owner_slabs(fqd_ctx *ctx, const uint32_t *slabs, const uint32_t *cursors, uint32_t n_senders,
                             uint32_t my_part, uint32_t hash_bins, uint32_t subs, uint32_t cap,
                             const uint64_t *sender_id0, uint64_t id_limit, uint64_t n_reads, uint32_t search_segments,
                             uint64_t *n_unique, int *done);
/* Optional: announce the owner rule BEFORE fqd_pack_keys, which then works out every read's owner
 * in the same pass; a matching fqd_export_packed_by_segment skips its own pass over the records.
 * n_parts = 0 switches it off. */
int fqd_set_owner_rule(fqd_ctx *ctx, uint32_t n_parts, uint32_t n_segments, uint32_t segment);
/* The UNIQUE table grouped the same way for a later pass: uids[i] = uid_base + row (the job-wide
 * id of a unique key: owner rank's base + its row). recs/lens/uids DEVICE, counts HOST. */
int fqd_export_unique_by_segment(fqd_ctx *ctx, uint32_t n_parts, uint32_t n_segments, uint32_t segment,
                                 uint32_t uid_base, uint32_t *recs, uint32_t *lens, uint32_t *uids,
                                 uint64_t *counts, int mem);
/* Rows idx[0..n) of the unique table: records, lengths (may be NULL), counts. DEVICE buffers.
 * FQD_E_VALUE when an index lies outside the table. */
int fqd_gather_unique(fqd_ctx *ctx, const uint32_t *idx, uint64_t n, uint32_t *recs, uint32_t *lens,
                      uint32_t *counts, int mem);
/* Unique table of stage 2. */
int fqd_export_unique(fqd_ctx *ctx, uint32_t *recs, uint32_t *lens, uint32_t *counts,
                      uint64_t *first_ids, int mem);
/* counts NULL = 1 each, first_ids NULL = 0 each (a table that is only searched). first_ids, when
 * given, are pairwise distinct (they are read ids). */
int fqd_import_unique(fqd_ctx *ctx, const uint32_t *recs, const uint32_t *lens,
                      const uint32_t *counts, const uint64_t *first_ids, uint64_t n_unique, int mem);
/* The caller vouches that the imported rows hold pairwise DISTINCT keys (rows of collapsed tables
 * of other ranks): the table is then treated like one fqd_collapse made (no distance-0 search,
 * closed-form directional dissection). */
int fqd_declare_distinct_keys(fqd_ctx *ctx);
/* Edge list of stage 3: n_edges pairs (u, v), u < v. */
int fqd_export_edges(fqd_ctx *ctx, uint32_t *uv, int mem);
int fqd_import_edges(fqd_ctx *ctx, const uint32_t *uv, uint64_t n_edges, int mem);

/* Search passes [seg_lo, seg_hi) only of the Hamming search with max_distance (a pair is still
 * reported in the FIRST segment it agrees on, so disjoint ranges over any placement of the keys
 * give every edge exactly once). Same role as fqd_find_edges: Trie.pop_cluster's neighbour
 * search, reference _triemodule.c:807-895. */
int fqd_find_edges_segments(fqd_ctx *ctx, int max_distance, uint32_t seg_lo, uint32_t seg_hi,
                            uint64_t *n_edges);
/* Components of a caller's edge list over n_nodes nodes (DEVICE buffers): roots[e] = smallest
 * node of edge e's component; *n_components = n_nodes - merges. The job-wide pop_cluster
 * partition when the nodes are global unique ids. uv: pairs of node ids, 8-byte aligned. */
int fqd_edge_labels(fqd_ctx *ctx, const uint32_t *uv, uint64_t n_edges, uint64_t n_nodes, uint32_t *roots,
                    uint64_t *n_components, int mem);
/* The clusters one rank dissects, cut out of the job-wide edge list (DEVICE buffers): the edges whose
 * root (fqd_edge_labels) is congruent to `part` modulo n_parts, their distinct ends in ascending
 * order (touched_out, up to min(2 E, n_nodes) words) and the edges again with every end replaced by
 * its position in touched_out (sub_edges_out, 2 E words, any edge order). */
int fqd_cluster_subgraph(fqd_ctx *ctx, const uint32_t *uv, const uint32_t *roots, uint64_t n_edges, uint64_t n_nodes,
                         uint32_t n_parts, uint32_t part, uint32_t *touched_out, uint32_t *sub_edges_out,
                         uint64_t *n_touched, uint64_t *n_sub, int mem);
/* ... with HOME clusters apart: a cluster all of whose keys lie in ONE rank's range of the job-wide key numbering
 * (uid_bounds[r] .. uid_bounds[r + 1], n_parts + 1 host values, n_parts <= 16) is dissected by that rank on the
 * unique table it holds (fqd_import_edges(home edges) + fqd_components + fqd_dissect_except). home_edges_out (2 E
 * words): the edges of this rank's home clusters, ends as rows of its own table; touched_out / sub_edges_out as
 * above, over the clusters that span ranks only. */
int fqd_cluster_subgraph_home(fqd_ctx *ctx, const uint32_t *uv, const uint32_t *roots, uint64_t n_edges, uint64_t n_nodes,
                              uint32_t n_parts, uint32_t part, const uint64_t *uid_bounds, uint32_t *touched_out,
                              uint32_t *sub_edges_out, uint32_t *home_edges_out, uint64_t *n_touched, uint64_t *n_sub,
                              uint64_t *n_home, uint64_t *n_spanning_edges /* of ALL ranks' shares: the same everywhere */,
                              int mem);
/* fqd_dissect over the edges in the context (the home clusters), then rows dropped[0..n_dropped) (DEVICE; keys of
 * clusters dissected on other ranks, which have no edge here) are dropped as well. */
int fqd_dissect_except(fqd_ctx *ctx, int method, const uint32_t *dropped, uint64_t n_dropped, int mem, uint64_t *n_kept);
/* Kept list when the verdicts were computed on other ranks: every key of the unique table is
 * kept except rows dropped[0..n_dropped) (DEVICE). Afterwards fqd_get_kept_count /
 * fqd_get_kept_read_ids / fqd_get_unique_table(kept) answer as after fqd_dissect. */
int fqd_list_kept_except(fqd_ctx *ctx, const uint32_t *dropped, uint64_t n_dropped, int mem,
                         uint64_t *n_kept);

/* ---- single calls of the reference surface --------------------------------- */
/* out[i] = within_distance(a_i, b_i) for n pairs (_distancemodule.c:46-93). */
int fqd_within_distance(fqd_ctx *ctx, const uint8_t *a_bytes, const uint64_t *a_offsets,
                        const uint8_t *b_bytes, const uint64_t *b_offsets, uint64_t n,
                        int max_distance, int metric, uint8_t *out, int mem);
/* out[i] = 1 when some unique key (after fqd_collapse) is within max_distance of
 * query i (Trie.contains_sequence, _triemodule.c:730-758). */
int fqd_contains(fqd_ctx *ctx, const uint8_t *q_bytes, const uint64_t *q_offsets, uint64_t n,
                 int max_distance, int metric, uint8_t *out, int mem);

/* ---- the Trie OBJECT as a device-resident store ------------------------------------
 * The reference's Trie is filled one add_sequence at a time, emptied one pop_cluster at a time, and
 * can be asked about its nodes in between (_triemodule.c:596-1009). These entry points give a host
 * in any language the same object over the unique table of a context -- no host-side key list, no
 * re-packing of what is already stored.
 *
 * `alphabet` (n_alpha symbols) is Trie.alphabet: the order of the child slots. It decides the order
 * in which clusters are popped and the width of every node's child array; it must list every
 * symbol that occurs in the keys. */

/* Trie.add_sequence for a batch (_triemodule.c:677-706 -> TrieNode_AddSequence :222-288): the n new
 * keys are merged into the context's unique table -- counts add up, the first holder of a key that
 * was already stored stays -- and rows taken out with fqd_store_remove leave the table. On an empty
 * context this is fqd_pack_keys + fqd_collapse. The geometry (alphabet, longest key, raggedness)
 * grows as the new keys require; the stored records are re-encoded on the device. read_ids (NULL:
 * consecutive numbers continuing the earlier batches) must ascend and exceed every id stored so
 * far. Arguments otherwise as fqd_cluster_keys. uids of the table change with every call. */
int fqd_store_add_keys(fqd_ctx *ctx, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, uint32_t fixed_len,
                       int mem, const uint32_t *weights, const uint64_t *read_ids, int aux_mem, uint64_t *n_unique);
/* The removal half of Trie.pop_cluster (TrieNode_DeleteSequence, _triemodule.c:301-363, called at
 * :830 and :875): rows uids[0..n) of the unique table are no longer in the trie. fqd_contains,
 * fqd_get_clusters and fqd_trie_stats skip them from now on; the rows themselves (and everybody's
 * uid) stay until the next fqd_store_add_keys. */
int fqd_store_remove(fqd_ctx *ctx, const uint32_t *uids, uint64_t n, int mem);
int fqd_store_removed_count(const fqd_ctx *ctx, uint64_t *n_removed);
/* After fqd_components: the remaining clusters in the order Trie.pop_cluster returns them
 * (_triemodule.c:778-897: every cluster is seeded with the leftmost key left, TrieNode_GetSequence
 * :510-551 -- child slots in alphabet order, a longer key before its own prefix), the members of a
 * cluster in that same key order (the seed first). fqd_get_clusters computes, fqd_read_clusters
 * copies out: offsets[n_clusters + 1] into member_uids[n_members] (rows of the unique table). */
int fqd_get_clusters(fqd_ctx *ctx, const uint8_t *alphabet, uint32_t n_alpha, uint64_t *n_clusters,
                     uint64_t *n_members);
int fqd_read_clusters(fqd_ctx *ctx, uint64_t *offsets, uint32_t *member_uids, int mem);
/* One round of the search for WHEN the reference's trie registers symbols that are not in its
 * constructor alphabet (Trie.alphabet grows lazily: TrieNode_AddSequence registers a byte when an
 * inner node first looks it up, _triemodule.c:266-273 -- base j of key k at the moment another key
 * sharing k's first j bases is stored next to it). For each of symbols[0..n_symbols): among the
 * stored keys added later than after[s] (first-holder id; ~0 = no bound) whose FIRST occurrence of
 * the symbol lies at or above their longest common prefix with another stored key, the one added
 * first: cand_first[s] (its first-holder id, ~0 = none), cand_depth[s] (position of the symbol),
 * partner_first[s] (the earliest-added other key sharing that prefix). The symbol is looked up at
 * time max(cand_first, partner_first); the caller repeats with after[s] = cand_first[s] while an
 * earlier time is still possible (fastqdedup_amd/core.py). `alphabet` may be any listing of all
 * symbols in the keys; reuse_order != 0 skips the sort when table and alphabet are unchanged since
 * the previous round. HOST arrays; first-holder ids must lie below 2^32. */
int fqd_store_symbol_events(fqd_ctx *ctx, const uint8_t *alphabet, uint32_t n_alpha, int reuse_order,
                            const uint8_t *symbols, uint32_t n_symbols, const uint64_t *after, uint64_t *cand_first,
                            uint32_t *cand_depth, uint64_t *partner_first);
/* order_out[r] = row of the r-th key in that key order (all rows, removed or not). */
int fqd_trie_order(fqd_ctx *ctx, const uint8_t *alphabet, uint32_t n_alpha, uint32_t *order_out, int mem);
/* Trie.memory_size and Trie.raw_stats (_triemodule.c:909-964 over TrieNode_GetMemorySize /
 * TrieNode_GetStats :553-594) of the trie the reference would hold after adding the table's keys
 * and then deleting the removed ones: *memory_size = sum over leaves of 8 + suffix bytes and over
 * inner nodes of 8 + 8 * child slots; stats[layer * (n_alpha + 1) + 0] = leaves in that layer,
 * [.. + k] = inner nodes with k child slots (HOST array of n_layers * (n_alpha + 1) counters;
 * n_layers = longest key ever added + 1). Exact for "adds, then pops" histories -- the ones
 * deduplicate_cluster produces (__init__.py:240-281); an add into a partly popped trie is merged as
 * if the popped keys had never been stored. */
int fqd_trie_stats(fqd_ctx *ctx, const uint8_t *alphabet, uint32_t n_alpha, uint32_t n_layers, uint64_t *memory_size,
                   uint64_t *stats);

/* ---- the quality gate in front of the path ---------------------------------
 * _fastq.average_error_rate (reference _fastqmodule.c:38-76) over n phred strings as
 * deduplicate_cluster uses it (__init__.py:243-250): pass_out[i] = 0 when the mean of
 * 10**-(q/10) over string i exceeds `threshold`, else 1 (an empty string gives NaN and
 * passes). The sum runs in string order in double precision, like the reference, so the
 * means are bit-identical. table128: the 128 error rates (NULL: computed as the
 * reference's generator does). means_out may be NULL. A character outside
 * ['!' + 0 .. '~'] relative to phred_offset is a FQD_E_VALUE. */
int fqd_quality_filter(fqd_ctx *ctx, const uint8_t *bytes, const uint64_t *offsets, uint64_t n,
                       uint32_t fixed_len, uint32_t phred_offset, double threshold,
                       const double *table128, uint32_t *pass_out, double *means_out,
                       uint64_t *n_discarded, int mem);

/* ---- measurement ----------------------------------------------------------- */
#define FQD_T_PACK       0
#define FQD_T_COLLAPSE   1
#define FQD_T_EDGES      2
#define FQD_T_COMPONENTS 3
#define FQD_T_DISSECT    4
#define FQD_T_PAIRS_KERNEL 5   /* sum over launches of the bucket pair-compare kernel */
#define FQD_T_PACK_KERNEL  6   /* the pack kernel alone (stage FQD_T_PACK also holds copies/scans) */
#define FQD_T_COUNT      8
/* HIP-event milliseconds of the last run of each stage, measured on the
 * context's own stream; launches[k] = kernel launches summed into ms[k]. */
int fqd_stage_times(fqd_ctx *ctx, float *ms /* FQD_T_COUNT */, uint32_t *launches /* FQD_T_COUNT */);
/* Per-kernel HIP-event time (ms, summed over launches) and launch counts of the hand-written
 * kernels since the last reset, measured on the context's own stream. */
#define FQD_K_PACK           0
#define FQD_K_PART_HIST1     1
#define FQD_K_PART_SCATTER1  2
#define FQD_K_PART_HIST2     3
#define FQD_K_PART_SCATTER2  4
#define FQD_K_DEDUPE         5
#define FQD_K_COMPACT        6
#define FQD_K_HEAD_FLAGS     7
#define FQD_K_WRITE_UNIQUE   8
#define FQD_K_SEG_HASH       9
#define FQD_K_PAIRS         10
#define FQD_K_UF_UNION      11
#define FQD_K_UF_FLATTEN    12
#define FQD_K_DISSECT_ROUND 13
#define FQD_K_GROUP_HIST    14   /* both levels of the (hash, uid) partition of a search pass */
#define FQD_K_GROUP_SCATTER 15
#define FQD_K_VERIFY        16   /* verification of the candidate pairs of a grouped search pass */
#define FQD_K_KEPT_FLAGS    17   /* verdict per unique key + the byte map of kept first-holder ids */
#define FQD_K_PART_SCATTER12 18  /* level 2 and the dedupe of the compact (12-byte) records of fqd_cluster_keys */
#define FQD_K_DEDUPE12      19
#define FQD_K_COUNT         20
int fqd_kernel_times(fqd_ctx *ctx, float *ms /* FQD_K_COUNT */, uint32_t *launches /* FQD_K_COUNT */,
                     int reset);
/* What is timed. Timers never synchronise the host: they are event pairs recorded while the work
 * is queued and resolved when the getters above are called. stage_timers = 0 records none of the
 * stage events; only the kernels whose bit (1u << FQD_K_...) is set in kernel_mask get event
 * pairs. Default: the stage timers on, the kernel timers off unless FQD_KERNEL_TIMERS=1 is in the environment
 * when the context is created -- event pairs around every launch stop the stream at every record: 2.44 against
 * 2.31 ms per job at config 3 (50 M reads). */
int fqd_set_timing(fqd_ctx *ctx, int stage_timers, uint32_t kernel_mask);
/* Bucket statistics of the last fqd_find_edges (for the roofline's unit count):
 * keys gathered by the pair kernel, pairs compared, edges emitted. */
int fqd_edge_stats(fqd_ctx *ctx, uint64_t *keys_gathered, uint64_t *pairs_compared,
                   uint64_t *edges_emitted);

/* Synthetic keys (fastqdedup_amd/synth.py, byte-identical): writes count*length
 * ASCII bytes for reads start..start+count of an n_total-read job into `out`
 * (device memory). Bench/test utility. */
int fqd_synth_keys(fqd_ctx *ctx, uint8_t *out_device, uint64_t n_total, uint64_t start,
                   uint64_t count, uint32_t length, uint32_t umi, uint64_t seed, uint32_t copies,
                   uint64_t thr_n, uint64_t thr_sub);

/* The same generator with the SKEWED model of fastqdedup_amd/synth.py (SKEW): heavy-tailed molecule abundance
 * (molecule floor(M x^2)), a share thr_hot / 2^53 of the reads copying ONE molecule, every lowc_every-th molecule
 * poly-A in the first half of its key, a share thr_ladder / 2^53 of the reads drawn from a ladder of 4^8 keys that
 * form one connected component -- crowded buckets, a giant component, a key with a million copies (SURVEY.md 7.4
 * "Skew"; the reference's trie takes any distribution, _triemodule.c:380-495). Bench/test utility. */
int fqd_synth_keys_skewed(fqd_ctx *ctx, uint8_t *out_device, uint64_t n_total, uint64_t start, uint64_t count,
                          uint32_t length, uint32_t umi, uint64_t seed, uint32_t copies, uint64_t thr_n,
                          uint64_t thr_sub, uint64_t thr_hot, uint64_t thr_ladder, uint32_t lowc_every);

/* Achievable HBM bandwidth on this GPU, now: the best of `reps` device-to-device copies of `bytes`
 * bytes by a 16-byte-per-lane copy kernel on the context's stream, read + write counted, in GB/s
 * (MI355X_MICROARCH.md: 6.29 TB/s for a float4 copy). bench.py's `achievable_peak_gbs`. */
int fqd_copy_bandwidth(fqd_ctx *ctx, const void *src_device, void *dst_device, uint64_t bytes, uint32_t reps,
                       double *gb_per_s);
/* The same job with an indel tail (fastqdedup_amd/synth.py indel_variant, byte-identical): a share
 * thr_indel / 2^53 of the reads loses one base or gains one, so the keys have three lengths -- the
 * shape SURVEY.md 8d asks for to exercise the Levenshtein search. Two calls: lens_out_device != NULL
 * writes the `count` key lengths (uint64, device); with their exclusive scan in offsets_device
 * (count + 1 entries) the second call writes the key bytes. Bench/test utility. */
int fqd_synth_indel_keys(fqd_ctx *ctx, uint64_t n_total, uint64_t start, uint64_t count, uint32_t length,
                         uint32_t umi, uint64_t seed, uint32_t copies, uint64_t thr_n, uint64_t thr_sub,
                         uint64_t thr_indel, uint64_t *lens_out_device, const uint64_t *offsets_device,
                         uint8_t *out_device);

#ifdef __cplusplus
}
#endif
#endif

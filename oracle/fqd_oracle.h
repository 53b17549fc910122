/*
 * fqd_oracle.h -- CPU ORACLE. TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the clustering hot path of rhpvorderman/fastqdedup
 * (reference @ /root/reference, v0.1.0-dev):
 *     src/fastqdedup/distances.h        within_hamming_distance / within_edit_distance
 *     src/fastqdedup/_triemodule.c      Alphabet, TrieNode, AddSequence, DeleteSequence,
 *                                       FindNearest, GetSequence, stats, Trie.pop_cluster
 *     src/fastqdedup/__init__.py:60-130 cluster_dissection_{directional,highest_count,adjacency}
 *     src/fastqdedup/__init__.py:240-276 the caller loop of deduplicate_cluster
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library. The product package (fastqdedup_amd/) never does: it fails
 * loudly when the HIP library is missing.
 *
 * Pinning: tests/test_oracle_*.py check this restatement against (a) every known
 * answer in the reference's own tests (tests/golden/reference_known_answers.json),
 * (b) golden vectors produced by the reference itself (oracle/_ref, built from
 * the reference sources where they lie) and committed under tests/golden/, and
 * (c) live against oracle/_ref when that is present.
 */
#ifndef FQD_ORACLE_H
#define FQD_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes shared by all entry points */
#define FQO_OK            0
#define FQO_E_NOMEM      -1
#define FQO_E_VALUE      -2   /* maps to ValueError */
#define FQO_E_LOOKUP     -3   /* maps to LookupError: "No sequences left in Trie." */
#define FQO_E_RUNTIME    -4

#define FQO_METHOD_HIGHEST_COUNT 0
#define FQO_METHOD_ADJACENCY     1
#define FQO_METHOD_DIRECTIONAL   2

/* distances.h:8-31 and :33-88 */
int fqo_within_hamming(const uint8_t *a, size_t la, const uint8_t *b, size_t lb, int max_distance);
int fqo_within_edit(const uint8_t *a, size_t la, const uint8_t *b, size_t lb, int max_distance);

/* _triemodule.c: type Trie */
typedef struct fqo_trie fqo_trie;
fqo_trie *fqo_trie_new(const uint8_t *alphabet, size_t alphabet_len, int *err, uint8_t *repeated_char);
void      fqo_trie_free(fqo_trie *t);
int       fqo_trie_add(fqo_trie *t, const uint8_t *seq, uint32_t len, uint32_t count);
/* 1 found, 0 not found. (The reference dereferences a NULL root on an empty
 * trie, _triemodule.c:755; the restatement answers 0 there.) */
int       fqo_trie_contains(fqo_trie *t, const uint8_t *seq, uint32_t len, int max_distance, int use_edit);
int64_t   fqo_trie_number_of_sequences(const fqo_trie *t);
uint32_t  fqo_trie_max_sequence_size(const fqo_trie *t);
size_t    fqo_trie_alphabet(const fqo_trie *t, uint8_t *out /* >=256 bytes */);
size_t    fqo_trie_memory_size(const fqo_trie *t);
/* out must hold (max_sequence_size+1) * (alphabet_size+1) size_t values */
int       fqo_trie_raw_stats(const fqo_trie *t, size_t *out);
/* Pops one cluster. Returns member count (>=1) or FQO_E_LOOKUP / FQO_E_VALUE.
 * Members are then readable through the three accessors until the next call. */
int64_t          fqo_trie_pop_cluster(fqo_trie *t, int max_distance, int use_edit);
const uint8_t   *fqo_cluster_bytes(const fqo_trie *t);
const uint64_t  *fqo_cluster_offsets(const fqo_trie *t);  /* n+1 */
const uint32_t  *fqo_cluster_counts(const fqo_trie *t);   /* n   */

/* __init__.py:60-122. kept_idx_out (capacity n) receives indices into the
 * input cluster in the reference's yield order. Returns the number yielded. */
int64_t fqo_dissect(int method, const uint32_t *counts, const uint8_t *bytes,
                    const uint64_t *offsets, uint64_t n, int max_distance, int use_edit,
                    uint64_t *kept_idx_out);

/* The whole hot path as deduplicate_cluster drives it (__init__.py:240-276):
 * Trie("ACGTN"); add every key with weight>0 (weights NULL => 1 each);
 * pop clusters until empty; dissect; then map each kept key to the FIRST input
 * index holding that key, counted or not (pass-2 rule, __init__.py:201-206).
 * kept_first_ids (capacity n) is returned sorted ascending.
 * stage_seconds[3] = insert, pop_cluster, dissect (may be NULL). */
int fqo_dedup(const uint8_t *bytes, const uint64_t *offsets, uint64_t n,
              const uint32_t *weights, int max_distance, int use_edit, int method,
              uint64_t *kept_first_ids, uint64_t *n_kept, uint64_t *n_clusters,
              uint64_t *n_unique, double *stage_seconds);

/* _fastqmodule.c:38-76: mean of 10**-(q/10) over the phred string, summed in order in a
 * double; *bad_char receives the first character outside [phred_offset, 126] (then
 * FQO_E_VALUE). An empty string gives NaN (0.0/0). */
int fqo_average_error_rate(const uint8_t *phred, size_t len, uint8_t phred_offset, double *out,
                           uint8_t *bad_char);

#ifdef __cplusplus
}
#endif
#endif

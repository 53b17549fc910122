/*
 * fqd_oracle.c -- CPU ORACLE. TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * See fqd_oracle.h for scope, pinning and who may load this.
 *
 * Every function names the reference lines (under /root/reference/) whose
 * behaviour it restates. Written from the behaviour, not from the text: the
 * node walk is iterative where the reference recurses, deletion keeps an
 * explicit slot stack, and the undefined behaviour of the reference
 * (uint32 underflow at _triemodule.c:458-460,477-479 when the query is
 * exhausted in edit mode; NULL root in contains_sequence, :755) is replaced
 * by its always-observed outcome ("no match").
 */
#define _POSIX_C_SOURCE 200809L
#include "fqd_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------ */
/* distances.h                                                              */
/* ------------------------------------------------------------------------ */

/* distances.h:8-31 -- equal length required, at most max_distance byte
 * mismatches, early exit. */
int fqo_within_hamming(const uint8_t *a, size_t la, const uint8_t *b, size_t lb, int max_distance)
{
    if (la != lb)
        return 0;
    int budget = max_distance;
    for (size_t i = 0; i < la; i++) {
        if (a[i] != b[i] && --budget < 0)
            return 0;
    }
    return 1;
}

/* distances.h:33-88 -- bounded Levenshtein by budgeted search: skip equal
 * heads; at a mismatch spend one unit and try "drop a's head", "drop b's
 * head", else treat it as a substitution and go on. Length pre-check (:42-48)
 * and leftover check (:80-86). */
int fqo_within_edit(const uint8_t *a, size_t la, const uint8_t *b, size_t lb, int max_distance)
{
    size_t gap = la > lb ? la - lb : lb - la;
    if ((int64_t)gap > (int64_t)max_distance)
        return 0;
    int budget = max_distance;
    while (la && lb) {
        if (*a != *b) {
            if (--budget < 0)
                return 0;
            if (fqo_within_edit(a + 1, la - 1, b, lb, budget))
                return 1;
            if (fqo_within_edit(a, la, b + 1, lb - 1, budget))
                return 1;
        }
        a++, la--, b++, lb--;
    }
    gap = la > lb ? la - lb : lb - la;
    return (int64_t)gap <= (int64_t)budget;
}

/* ------------------------------------------------------------------------ */
/* Alphabet (_triemodule.c:32-67)                                           */
/* ------------------------------------------------------------------------ */

#define AB_UNKNOWN 255
#define AB_MAX 254

typedef struct {
    uint8_t symbol_of[256]; /* slot -> byte */
    uint8_t slot_of[256];   /* byte -> slot, AB_UNKNOWN when absent */
    uint32_t n;
} alphabet_t;

static int alphabet_init(alphabet_t *ab, const uint8_t *chars, size_t len, uint8_t *repeated)
{
    memset(ab->symbol_of, 0, sizeof ab->symbol_of);
    memset(ab->slot_of, AB_UNKNOWN, sizeof ab->slot_of);
    ab->n = 0;
    if (len > AB_MAX)
        return FQO_E_VALUE; /* :46-49 */
    for (size_t i = 0; i < len; i++) {
        uint8_t c = chars[i];
        if (ab->slot_of[c] != AB_UNKNOWN) { /* :55-61 */
            if (repeated)
                *repeated = c;
            return FQO_E_VALUE;
        }
        ab->slot_of[c] = (uint8_t)ab->n;
        ab->symbol_of[ab->n] = c;
        ab->n++;
    }
    return FQO_OK;
}

static uint8_t alphabet_slot_or_add(alphabet_t *ab, uint8_t c)
{ /* :266-273 lazily registers a byte the first time an INNER node looks it up */
    uint8_t s = ab->slot_of[c];
    if (s == AB_UNKNOWN) {
        s = (uint8_t)ab->n;
        ab->slot_of[c] = s;
        ab->symbol_of[s] = c;
        ab->n = (uint32_t)s + 1;
    }
    return s;
}

/* ------------------------------------------------------------------------ */
/* Node (_triemodule.c:102-113): 8-byte header, then either `span` child    */
/* pointers (inner node; slots >= span are implicitly empty) or `span`      */
/* bytes of key tail (leaf).                                                */
/* ------------------------------------------------------------------------ */

typedef struct tnode {
    uint32_t span;
    uint32_t hits : 31;
    uint32_t leaf : 1;
    union {
        struct tnode *kid[1];
        uint8_t tail[1];
    } u;
} tnode;

#define NODE_HEADER 8u
_Static_assert(offsetof(tnode, u) == NODE_HEADER, "node header must be 8 bytes");

static tnode *leaf_new(const uint8_t *tail, uint32_t len, uint32_t hits)
{ /* :191-208 */
    size_t bytes = NODE_HEADER + (size_t)len;
    tnode *nd = malloc(bytes < sizeof(tnode) ? sizeof(tnode) : bytes);
    if (!nd)
        return NULL;
    nd->span = len;
    nd->hits = hits;
    nd->leaf = 1;
    if (len)
        memcpy(nd->u.tail, tail, len);
    return nd;
}

static tnode *inner_widen(tnode *nd, uint32_t want)
{ /* :136-161 */
    uint32_t have = nd->span;
    if (want <= have)
        return nd;
    tnode *wider = realloc(nd, NODE_HEADER + sizeof(tnode *) * (size_t)want);
    if (!wider)
        return NULL;
    for (uint32_t i = have; i < want; i++)
        wider->u.kid[i] = NULL;
    wider->span = want;
    return wider;
}

static void node_destroy(tnode *nd)
{ /* :166-181 */
    if (!nd)
        return;
    if (!nd->leaf)
        for (uint32_t i = 0; i < nd->span; i++)
            node_destroy(nd->u.kid[i]);
    free(nd);
}

static inline tnode *inner_child(const tnode *nd, uint32_t slot)
{ /* :122-129 */
    return slot < nd->span ? nd->u.kid[slot] : NULL;
}

/* :222-288 TrieNode_AddSequence. Iterative walk; the only nested call is the
 * one-level push-down of a split leaf's old tail (:241-260). */
static int node_insert(tnode **slot, const uint8_t *seq, uint32_t len, uint32_t mult, alphabet_t *ab)
{
    for (;;) {
        tnode *nd = *slot;
        if (!nd) {
            *slot = leaf_new(seq, len, mult);
            return *slot ? FQO_OK : FQO_E_NOMEM;
        }
        if (nd->leaf) {
            if (nd->span == len && (len == 0 || memcmp(nd->u.tail, seq, len) == 0)) {
                nd->hits += mult; /* :235-239 */
                return FQO_OK;
            }
            uint32_t old_len = nd->span, old_hits = nd->hits;
            uint8_t *old = malloc(old_len ? old_len : 1);
            if (!old)
                return FQO_E_NOMEM;
            memcpy(old, nd->u.tail, old_len);
            nd->leaf = 0;
            nd->span = 0;
            nd->hits = 0;
            int rc = node_insert(slot, old, old_len, old_hits, ab);
            free(old);
            if (rc != FQO_OK)
                return rc;
            nd = *slot;
        }
        if (len == 0) { /* :261-264 key ends on an inner node */
            nd->hits += mult;
            return FQO_OK;
        }
        uint8_t s = alphabet_slot_or_add(ab, seq[0]);
        if (s >= nd->span) {
            if ((uint32_t)s + 1 > AB_MAX)
                return FQO_E_RUNTIME;
            tnode *wider = inner_widen(nd, (uint32_t)s + 1);
            if (!wider)
                return FQO_E_NOMEM;
            *slot = nd = wider;
        }
        slot = &nd->u.kid[s];
        seq++, len--;
    }
}

/* ------------------------------------------------------------------------ */
/* Trie object                                                              */
/* ------------------------------------------------------------------------ */

struct fqo_trie {
    alphabet_t ab;
    int64_t n_sequences;     /* total multiplicity (_triemodule.c:599,699,837,883) */
    uint32_t max_len;        /* never shrinks (:700-702) */
    tnode *root;
    uint8_t *scratch;        /* found-key buffer, >= max_len bytes (:799-810) */
    size_t scratch_cap;
    tnode ***path;           /* slot stack for deletion */
    size_t path_cap;
    /* last popped cluster */
    uint8_t *cl_bytes;
    size_t cl_bytes_cap;
    uint64_t *cl_off;
    uint32_t *cl_cnt;
    size_t cl_cap, cl_n;
};

fqo_trie *fqo_trie_new(const uint8_t *alphabet, size_t alphabet_len, int *err, uint8_t *repeated_char)
{ /* :613-642 */
    fqo_trie *t = calloc(1, sizeof *t);
    if (!t) {
        if (err)
            *err = FQO_E_NOMEM;
        return NULL;
    }
    int rc = alphabet_init(&t->ab, alphabet, alphabet_len, repeated_char);
    if (rc != FQO_OK) {
        free(t);
        if (err)
            *err = rc;
        return NULL;
    }
    if (err)
        *err = FQO_OK;
    return t;
}

void fqo_trie_free(fqo_trie *t)
{ /* :606-611 */
    if (!t)
        return;
    node_destroy(t->root);
    free(t->scratch);
    free(t->path);
    free(t->cl_bytes);
    free(t->cl_off);
    free(t->cl_cnt);
    free(t);
}

int fqo_trie_add(fqo_trie *t, const uint8_t *seq, uint32_t len, uint32_t count)
{ /* :677-706 (ASCII/type checks live in the Python wrapper) */
    int rc = node_insert(&t->root, seq, len, count, &t->ab);
    if (rc != FQO_OK)
        return rc;
    t->n_sequences += count;
    if (len > t->max_len)
        t->max_len = len;
    return FQO_OK;
}

int64_t fqo_trie_number_of_sequences(const fqo_trie *t) { return t->n_sequences; }
uint32_t fqo_trie_max_sequence_size(const fqo_trie *t) { return t->max_len; }

size_t fqo_trie_alphabet(const fqo_trie *t, uint8_t *out)
{ /* :644-648 */
    memcpy(out, t->ab.symbol_of, t->ab.n);
    return t->ab.n;
}

/* :301-363 TrieNode_DeleteSequence: remove the key, return its multiplicity
 * (0 = not stored); afterwards drop every ancestor left without children,
 * turning one that still counts a key of its own into an empty-tail leaf. */
static uint32_t trie_remove(fqo_trie *t, const uint8_t *seq, uint32_t len)
{
    if (!t->root)
        return 0;
    if ((size_t)len + 2 > t->path_cap) {
        size_t cap = (size_t)len + 2;
        tnode ***p = realloc(t->path, cap * sizeof *p);
        if (!p)
            return 0;
        t->path = p;
        t->path_cap = cap;
    }
    size_t depth = 0;
    tnode **slot = &t->root;
    uint32_t removed = 0;
    for (;;) {
        tnode *nd = *slot;
        if (nd->leaf) {
            if (nd->span != len || (len && memcmp(nd->u.tail, seq, len) != 0))
                return 0;
            removed = nd->hits;
            free(nd);
            *slot = NULL;
            break;
        }
        if (len == 0) { /* :324-328 key ends on an inner node: no pruning needed */
            removed = nd->hits;
            nd->hits = 0;
            return removed;
        }
        uint8_t s = t->ab.slot_of[seq[0]];
        if (s == AB_UNKNOWN || !inner_child(nd, s))
            return 0;
        t->path[depth++] = slot;
        slot = &nd->u.kid[s];
        seq++, len--;
    }
    while (depth) { /* :344-361 */
        tnode **up = t->path[--depth];
        tnode *nd = *up;
        for (uint32_t i = 0; i < nd->span; i++)
            if (nd->u.kid[i])
                return removed;
        if (nd->hits) {
            *up = leaf_new(NULL, 0, nd->hits);
            free(nd);
            return removed; /* slot stays occupied: ancestors keep a child */
        }
        *up = NULL;
        free(nd);
    }
    return removed;
}

/* :380-495 TrieNode_FindNearest. Returns the length of SOME stored key within
 * the budget (its bytes in out[0..len) when out != NULL) or -1. Search order:
 * exact-match child first (free), then with one unit spent: (edit) skip one
 * query byte at the same node; every other child in slot order, each followed
 * (edit) by "take the child's byte without consuming the query". */
static int64_t node_nearest(const tnode *nd, const uint8_t *q, uint32_t qlen, int budget,
                            const alphabet_t *ab, uint8_t *out, int edit)
{
    if (nd->leaf) { /* :390-409 */
        int ok = edit ? fqo_within_edit(q, qlen, nd->u.tail, nd->span, budget)
                      : fqo_within_hamming(q, qlen, nd->u.tail, nd->span, budget);
        if (!ok)
            return -1;
        if (out && nd->span)
            memcpy(out, nd->u.tail, nd->span);
        return nd->span;
    }
    if (edit && (int64_t)qlen <= (int64_t)budget && nd->hits)
        return 0; /* :410-413 */
    uint8_t *deeper = out ? out + 1 : NULL;
    const tnode *straight = NULL;
    uint32_t straight_slot = AB_UNKNOWN;
    if (qlen == 0) { /* :423-434 */
        if (nd->hits)
            return 0;
        if (!edit)
            return -1;
    } else {
        straight_slot = ab->slot_of[q[0]];
        straight = inner_child(nd, straight_slot);
    }
    if (straight) { /* :439-450 */
        if (out)
            out[0] = q[0];
        int64_t r = node_nearest(straight, q + 1, qlen - 1, budget, ab, deeper, edit);
        if (r >= 0)
            return r + 1;
    }
    if (--budget < 0) /* :452-455 */
        return -1;
    if (edit && qlen > 0) { /* :456-464 (with qlen==0 the reference underflows and always fails) */
        int64_t r = node_nearest(nd, q + 1, qlen - 1, budget, ab, out, edit);
        if (r >= 0)
            return r;
    }
    for (uint32_t i = 0; i < nd->span; i++) { /* :466-492 */
        const tnode *kid = nd->u.kid[i];
        if (i == straight_slot || !kid)
            continue;
        if (out)
            out[0] = ab->symbol_of[i];
        if (qlen > 0) {
            int64_t r = node_nearest(kid, q + 1, qlen - 1, budget, ab, deeper, edit);
            if (r >= 0)
                return r + 1;
        }
        if (edit) {
            int64_t r = node_nearest(kid, q, qlen, budget, ab, deeper, edit);
            if (r >= 0)
                return r + 1;
        }
    }
    return -1;
}

int fqo_trie_contains(fqo_trie *t, const uint8_t *seq, uint32_t len, int max_distance, int use_edit)
{ /* :730-758 */
    if (!t->root)
        return 0;
    return node_nearest(t->root, seq, len, max_distance, &t->ab, NULL, use_edit) >= 0;
}

/* :510-551 TrieNode_GetSequence: leftmost key in slot order; a node's
 * children are tried before its own count, so a longer key precedes its
 * prefix. Iterative. Returns length or -1. */
static int64_t trie_leftmost(const fqo_trie *t, uint8_t *out, size_t cap)
{
    const tnode *nd = t->root;
    size_t depth = 0;
    for (;;) {
        if (nd->leaf) {
            if (depth + nd->span > cap)
                return -1;
            if (nd->span)
                memcpy(out + depth, nd->u.tail, nd->span);
            return (int64_t)(depth + nd->span);
        }
        const tnode *next = NULL;
        uint32_t i;
        for (i = 0; i < nd->span; i++)
            if ((next = nd->u.kid[i]) != NULL)
                break;
        if (!next)
            return nd->hits ? (int64_t)depth : -1;
        if (depth >= cap)
            return -1;
        out[depth++] = t->ab.symbol_of[i];
        nd = next;
    }
}

/* :553-570 */
static size_t node_bytes(const tnode *nd)
{
    if (!nd)
        return 0;
    if (nd->leaf)
        return NODE_HEADER + nd->span;
    size_t total = NODE_HEADER + sizeof(tnode *) * (size_t)nd->span;
    for (uint32_t i = 0; i < nd->span; i++)
        total += node_bytes(nd->u.kid[i]);
    return total;
}

size_t fqo_trie_memory_size(const fqo_trie *t) { return node_bytes(t->root); } /* :909-913 */

/* :572-594 */
static void node_census(const tnode *nd, size_t layer, size_t layers, size_t cols, size_t *out)
{
    if (!nd || layer >= layers)
        return;
    size_t *row = out + layer * cols;
    if (nd->leaf) {
        row[0]++;
        return;
    }
    if (nd->span < cols)
        row[nd->span]++;
    for (uint32_t i = 0; i < nd->span; i++)
        node_census(nd->u.kid[i], layer + 1, layers, cols, out);
}

int fqo_trie_raw_stats(const fqo_trie *t, size_t *out)
{ /* :929-964 */
    size_t cols = (size_t)t->ab.n + 1, layers = (size_t)t->max_len + 1;
    memset(out, 0, cols * layers * sizeof *out);
    node_census(t->root, 0, layers, cols, out);
    return FQO_OK;
}

static int cluster_push(fqo_trie *t, const uint8_t *key, uint64_t len, uint32_t count)
{
    if (t->cl_n + 2 > t->cl_cap) {
        size_t cap = t->cl_cap ? t->cl_cap * 2 : 16;
        uint64_t *o = realloc(t->cl_off, (cap + 1) * sizeof *o);
        if (!o)
            return FQO_E_NOMEM;
        t->cl_off = o;
        uint32_t *c = realloc(t->cl_cnt, cap * sizeof *c);
        if (!c)
            return FQO_E_NOMEM;
        t->cl_cnt = c;
        t->cl_cap = cap;
    }
    uint64_t at = t->cl_n ? t->cl_off[t->cl_n] : 0;
    if (at + len + 1 > t->cl_bytes_cap) {
        size_t cap = t->cl_bytes_cap ? t->cl_bytes_cap * 2 : 256;
        while (cap < at + len + 1)
            cap *= 2;
        uint8_t *b = realloc(t->cl_bytes, cap);
        if (!b)
            return FQO_E_NOMEM;
        t->cl_bytes = b;
        t->cl_bytes_cap = cap;
    }
    if (len)
        memcpy(t->cl_bytes + at, key, len);
    t->cl_off[t->cl_n] = at;
    t->cl_off[t->cl_n + 1] = at + len;
    t->cl_cnt[t->cl_n] = count;
    t->cl_n++;
    return FQO_OK;
}

/* :778-897 Trie.pop_cluster: seed = leftmost key; remove it; then, member by
 * member, keep pulling (find + remove) neighbours of the current template
 * until it has none left, and move on to the next member. */
int64_t fqo_trie_pop_cluster(fqo_trie *t, int max_distance, int use_edit)
{
    if (max_distance < 0)
        return FQO_E_VALUE; /* :789-793 */
    if (!t->root)
        return FQO_E_LOOKUP; /* :794-797 */
    if (t->scratch_cap < (size_t)t->max_len + 1) {
        uint8_t *s = realloc(t->scratch, (size_t)t->max_len + 1);
        if (!s)
            return FQO_E_NOMEM;
        t->scratch = s;
        t->scratch_cap = (size_t)t->max_len + 1;
    }
    t->cl_n = 0;
    int64_t len = trie_leftmost(t, t->scratch, t->max_len);
    if (len < 0)
        return FQO_E_RUNTIME; /* :815-817 */
    uint32_t hits = trie_remove(t, t->scratch, (uint32_t)len);
    if (!hits)
        return FQO_E_RUNTIME; /* :832-836 */
    t->n_sequences -= hits;
    int rc = cluster_push(t, t->scratch, (uint64_t)len, hits);
    if (rc != FQO_OK)
        return rc;
    if (max_distance == 0)
        return 1; /* :844-846 */
    size_t cursor = 0;
    while (cursor != t->cl_n && t->root) { /* :865-895 */
        const uint8_t *tpl = t->cl_bytes + t->cl_off[cursor];
        uint32_t tpl_len = (uint32_t)(t->cl_off[cursor + 1] - t->cl_off[cursor]);
        int64_t got = node_nearest(t->root, tpl, tpl_len, max_distance, &t->ab, t->scratch, use_edit);
        if (got < 0) {
            cursor++;
            continue;
        }
        hits = trie_remove(t, t->scratch, (uint32_t)got);
        if (!hits)
            return FQO_E_RUNTIME; /* :877-882 */
        t->n_sequences -= hits;
        rc = cluster_push(t, t->scratch, (uint64_t)got, hits);
        if (rc != FQO_OK)
            return rc;
    }
    return (int64_t)t->cl_n;
}

const uint8_t *fqo_cluster_bytes(const fqo_trie *t) { return t->cl_bytes; }
const uint64_t *fqo_cluster_offsets(const fqo_trie *t) { return t->cl_off; }
const uint32_t *fqo_cluster_counts(const fqo_trie *t) { return t->cl_cnt; }

/* ------------------------------------------------------------------------ */
/* Cluster dissection (__init__.py:60-130)                                  */
/* ------------------------------------------------------------------------ */

typedef struct {
    uint32_t count;
    const uint8_t *s;
    uint64_t len;
    uint64_t idx;
} member_t;

/* Python tuple order of (count:int, key:str): count, then bytes, then
 * "a proper prefix is smaller". */
static int member_cmp(const void *pa, const void *pb)
{
    const member_t *a = pa, *b = pb;
    if (a->count != b->count)
        return a->count < b->count ? -1 : 1;
    uint64_t m = a->len < b->len ? a->len : b->len;
    int c = m ? memcmp(a->s, b->s, m) : 0;
    if (c)
        return c;
    if (a->len != b->len)
        return a->len < b->len ? -1 : 1;
    return a->idx < b->idx ? -1 : (a->idx > b->idx);
}

static int member_near(const member_t *a, const member_t *b, int d, int edit)
{ /* _distancemodule.c:46-93 */
    return edit ? fqo_within_edit(a->s, a->len, b->s, b->len, d)
                : fqo_within_hamming(a->s, a->len, b->s, b->len, d);
}

int64_t fqo_dissect(int method, const uint32_t *counts, const uint8_t *bytes,
                    const uint64_t *offsets, uint64_t n, int max_distance, int use_edit,
                    uint64_t *kept_idx_out)
{
    if (n == 0)
        return 0;
    member_t *pool = malloc(n * sizeof *pool);
    member_t *chain = malloc(n * sizeof *chain);
    if (!pool || !chain) {
        free(pool);
        free(chain);
        return FQO_E_NOMEM;
    }
    for (uint64_t i = 0; i < n; i++) {
        pool[i].count = counts[i];
        pool[i].s = bytes + offsets[i];
        pool[i].len = offsets[i + 1] - offsets[i];
        pool[i].idx = i;
    }
    qsort(pool, n, sizeof *pool, member_cmp); /* ascending */
    int64_t kept = 0;
    uint64_t left = n;

    if (method == FQO_METHOD_HIGHEST_COUNT) { /* :94-102 */
        kept_idx_out[kept++] = pool[n - 1].idx;
    } else if (method == FQO_METHOD_ADJACENCY) { /* :105-122 */
        /* the reference sorts descending and takes element 0; here the pool is
         * ascending and the root is the last element. Survivors keep order. */
        while (left) {
            member_t root = pool[left - 1];
            uint64_t w = 0;
            for (uint64_t i = 0; i + 1 < left; i++)
                if (!member_near(&root, &pool[i], max_distance, use_edit))
                    pool[w++] = pool[i];
            kept_idx_out[kept++] = root.idx;
            left = w;
        }
    } else if (method == FQO_METHOD_DIRECTIONAL) { /* :60-91 */
        while (left) {
            member_t root = pool[--left];
            uint64_t chain_n = 0;
            chain[chain_n++] = root;
            for (uint64_t c = 0; c < chain_n && left; c++) {
                member_t tpl = chain[c];
                uint64_t w = 0;
                for (uint64_t i = 0; i < left; i++) {
                    /* :84 (2*n-1) <= template count */
                    if (2 * (int64_t)pool[i].count - 1 <= (int64_t)tpl.count &&
                        member_near(&tpl, &pool[i], max_distance, use_edit))
                        chain[chain_n++] = pool[i];
                    else
                        pool[w++] = pool[i];
                }
                left = w;
            }
            kept_idx_out[kept++] = root.idx;
        }
    } else {
        kept = FQO_E_VALUE;
    }
    free(pool);
    free(chain);
    return kept;
}

/* ------------------------------------------------------------------------ */
/* Whole path (__init__.py:240-276 + pass-2 rule :201-206)                  */
/* ------------------------------------------------------------------------ */

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static uint64_t bytes_hash(const uint8_t *s, uint64_t len)
{
    uint64_t h = 0xcbf29ce484222325ull ^ len;
    for (uint64_t i = 0; i < len; i++) {
        h ^= s[i];
        h *= 0x100000001b3ull;
    }
    h ^= h >> 29;
    h *= 0xbf58476d1ce4e5b9ull;
    h ^= h >> 32;
    return h;
}

static int u64_cmp(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y);
}

int fqo_dedup(const uint8_t *bytes, const uint64_t *offsets, uint64_t n,
              const uint32_t *weights, int max_distance, int use_edit, int method,
              uint64_t *kept_first_ids, uint64_t *n_kept, uint64_t *n_clusters,
              uint64_t *n_unique, double *stage_seconds)
{
    int err = FQO_OK;
    double t_insert = 0, t_pop = 0, t_dissect = 0;
    *n_kept = *n_clusters = *n_unique = 0;
    if (max_distance < 0)
        return FQO_E_VALUE;

    /* first holder of every distinct key over ALL inputs, counted or not */
    uint64_t cap = 16;
    while (cap < 2 * n + 2)
        cap <<= 1;
    uint64_t *first = calloc(cap, sizeof *first); /* index+1, 0 = empty */
    if (!first)
        return FQO_E_NOMEM;
    for (uint64_t i = 0; i < n; i++) {
        const uint8_t *s = bytes + offsets[i];
        uint64_t len = offsets[i + 1] - offsets[i];
        uint64_t h = bytes_hash(s, len) & (cap - 1);
        for (;;) {
            uint64_t e = first[h];
            if (!e) {
                first[h] = i + 1;
                break;
            }
            uint64_t j = e - 1, jl = offsets[j + 1] - offsets[j];
            if (jl == len && (len == 0 || memcmp(bytes + offsets[j], s, len) == 0))
                break;
            h = (h + 1) & (cap - 1);
        }
    }

    fqo_trie *t = fqo_trie_new((const uint8_t *)"ACGTN", 5, &err, NULL); /* :240 */
    if (!t) {
        free(first);
        return err;
    }
    double t0 = now_s();
    for (uint64_t i = 0; i < n; i++) { /* :242-252 */
        uint32_t w = weights ? weights[i] : 1;
        if (!w)
            continue;
        uint64_t len = offsets[i + 1] - offsets[i];
        if (len > UINT32_MAX) {
            err = FQO_E_VALUE;
            goto done;
        }
        err = fqo_trie_add(t, bytes + offsets[i], (uint32_t)len, w);
        if (err != FQO_OK)
            goto done;
    }
    t_insert = now_s() - t0;

    uint64_t kept = 0, clusters = 0, unique = 0;
    uint64_t *scratch_idx = NULL;
    size_t scratch_cap = 0;
    while (fqo_trie_number_of_sequences(t) > 0) { /* :272 */
        t0 = now_s();
        int64_t m = fqo_trie_pop_cluster(t, max_distance, use_edit);
        t_pop += now_s() - t0;
        if (m < 0) {
            err = (int)m;
            free(scratch_idx);
            goto done;
        }
        clusters++;
        unique += (uint64_t)m;
        if ((size_t)m > scratch_cap) {
            uint64_t *p = realloc(scratch_idx, (size_t)m * sizeof *p);
            if (!p) {
                err = FQO_E_NOMEM;
                free(scratch_idx);
                goto done;
            }
            scratch_idx = p;
            scratch_cap = (size_t)m;
        }
        t0 = now_s();
        int64_t k = fqo_dissect(method, t->cl_cnt, t->cl_bytes, t->cl_off, (uint64_t)m,
                                max_distance, use_edit, scratch_idx); /* :275 */
        t_dissect += now_s() - t0;
        if (k < 0) {
            err = (int)k;
            free(scratch_idx);
            goto done;
        }
        for (int64_t q = 0; q < k; q++) { /* :276 + :201-206 */
            uint64_t ci = scratch_idx[q];
            const uint8_t *s = t->cl_bytes + t->cl_off[ci];
            uint64_t len = t->cl_off[ci + 1] - t->cl_off[ci];
            uint64_t h = bytes_hash(s, len) & (cap - 1);
            for (;;) {
                uint64_t e = first[h];
                if (!e) {
                    err = FQO_E_RUNTIME;
                    free(scratch_idx);
                    goto done;
                }
                uint64_t j = e - 1, jl = offsets[j + 1] - offsets[j];
                if (jl == len && (len == 0 || memcmp(bytes + offsets[j], s, len) == 0)) {
                    kept_first_ids[kept++] = j;
                    break;
                }
                h = (h + 1) & (cap - 1);
            }
        }
    }
    free(scratch_idx);
    qsort(kept_first_ids, kept, sizeof *kept_first_ids, u64_cmp);
    *n_kept = kept;
    *n_clusters = clusters;
    *n_unique = unique;
done:
    if (stage_seconds) {
        stage_seconds[0] = t_insert;
        stage_seconds[1] = t_pop;
        stage_seconds[2] = t_dissect;
    }
    fqo_trie_free(t);
    free(first);
    return err;
}

/* ------------------------------------------------------------------------ */
/* Quality gate (_fastqmodule.c:38-76, score_to_error_rate.py)              */
/* ------------------------------------------------------------------------ */

int fqo_average_error_rate(const uint8_t *phred, size_t len, uint8_t phred_offset, double *out,
                           uint8_t *bad_char)
{
    static double table[128];
    static int ready = 0;
    if (!ready) { /* score_to_error_rate.py: 10 ** -(i / 10) */
        for (int i = 0; i < 128; i++)
            table[i] = pow(10.0, -((double)i / 10.0));
        ready = 1;
    }
    const uint8_t max_score = (uint8_t)(126 - phred_offset); /* :20, :65 */
    double total = 0.0;
    for (size_t i = 0; i < len; i++) {
        const uint8_t score = (uint8_t)(phred[i] - phred_offset);
        if (score > max_score) { /* :68-74 */
            if (bad_char)
                *bad_char = phred[i];
            return FQO_E_VALUE;
        }
        total += table[score];
    }
    *out = total / (double)len; /* :76 */
    return FQO_OK;
}

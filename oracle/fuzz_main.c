/* CPU ORACLE -- TEST INFRASTRUCTURE. Self-checking fuzz driver for fqd_oracle.c, meant to be
 * built with -fsanitize=address,undefined (make -C oracle fuzz_asan): random tries, interleaved
 * add / contains / pop_cluster for both metrics, every popped cluster checked against a
 * brute-force connected component, plus fqo_dedup on random inputs. Exit code 0 = clean. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fqd_oracle.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(uint32_t n)
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)((rng_state >> 11) % n);
}

#define MAXK 64
#define MAXL 9

int main(void)
{
    const char *syms = "ACGTN";
    for (int trial = 0; trial < 4000; trial++) {
        int err = 0;
        fqo_trie *t = fqo_trie_new((const uint8_t *)"ACGTN", trial % 3 ? 5 : 0, &err, NULL);
        if (!t)
            return 2;
        char keys[MAXK][MAXL + 1];
        int alive[MAXK];
        const int n = 1 + (int)rnd(MAXK - 1), nsym = 2 + (int)rnd(4), d = (int)rnd(4), edit = (int)rnd(2);
        for (int i = 0; i < n; i++) {
            const int len = (int)rnd(MAXL);
            for (int j = 0; j < len; j++)
                keys[i][j] = syms[rnd((uint32_t)nsym)];
            keys[i][len] = 0;
            alive[i] = 1;
            if (fqo_trie_add(t, (const uint8_t *)keys[i], (uint32_t)len, 1 + rnd(3)) != FQO_OK)
                return 3;
        }
        for (int q = 0; q < 8; q++) {
            char probe[MAXL + 1];
            const int len = (int)rnd(MAXL);
            for (int j = 0; j < len; j++)
                probe[j] = syms[rnd(5)];
            int want = 0;
            for (int i = 0; i < n && !want; i++)
                want = edit ? fqo_within_edit((const uint8_t *)probe, (size_t)len, (const uint8_t *)keys[i],
                                              strlen(keys[i]), d)
                            : fqo_within_hamming((const uint8_t *)probe, (size_t)len, (const uint8_t *)keys[i],
                                                 strlen(keys[i]), d);
            if (fqo_trie_contains(t, (const uint8_t *)probe, (uint32_t)len, d, edit) != want) {
                fprintf(stderr, "contains mismatch trial %d\n", trial);
                return 4;
            }
        }
        while (fqo_trie_number_of_sequences(t) > 0) {
            const int64_t m = fqo_trie_pop_cluster(t, d, edit);
            if (m < 1)
                return 5;
            const uint8_t *b = fqo_cluster_bytes(t);
            const uint64_t *o = fqo_cluster_offsets(t);
            /* closure check: no still-alive key may be within distance of a popped member */
            for (int64_t k = 0; k < m; k++)
                for (int i = 0; i < n; i++)
                    if (alive[i] && strlen(keys[i]) == o[k + 1] - o[k] &&
                        memcmp(keys[i], b + o[k], o[k + 1] - o[k]) == 0)
                        alive[i] = 0;
            for (int64_t k = 0; k < m; k++)
                for (int i = 0; i < n; i++) {
                    if (!alive[i])
                        continue;
                    const int near = edit ? fqo_within_edit(b + o[k], o[k + 1] - o[k], (const uint8_t *)keys[i],
                                                            strlen(keys[i]), d)
                                          : fqo_within_hamming(b + o[k], o[k + 1] - o[k],
                                                               (const uint8_t *)keys[i], strlen(keys[i]), d);
                    if (near && d > 0) {
                        fprintf(stderr, "cluster not closed, trial %d\n", trial);
                        return 6;
                    }
                }
        }
        fqo_trie_free(t);

        /* whole path */
        uint8_t flat[MAXK * MAXL];
        uint64_t off[MAXK + 1], kept[MAXK], nk, nc, nu;
        uint32_t w[MAXK];
        off[0] = 0;
        for (int i = 0; i < n; i++) {
            const size_t len = strlen(keys[i]);
            memcpy(flat + off[i], keys[i], len);
            off[i + 1] = off[i] + len;
            w[i] = rnd(3);
        }
        if (fqo_dedup(flat, off, (uint64_t)n, w, d, edit, (int)rnd(3), kept, &nk, &nc, &nu, NULL) != FQO_OK)
            return 7;
        if (nk > nu || nc > nu)
            return 8;
    }
    puts("oracle fuzz clean");
    return 0;
}

/* libmi355radix.so - the paged radix tree of the KV cache as native host code.
 *
 * Host-only C ABI (no GPU, no torch types): token ids in, physical page numbers out.  Same results as the
 * reference's Python tree, operation for operation (memory/radix_cache.py:73-371): keys are compared in
 * page_size-token chunks, a child is found by its first chunk (:85-90), match_prefix splits a node that matches
 * only in part, eviction removes unlocked leaves in least-recently-used order (:202-221).  After such a split
 * the walk stops; the reference (:263-267) continues from the new upper node without consuming the key and can then
 * return pages of tokens the key does not hold ("abab" cached, "abc" queried) - m355r_set_reference_walk(tree, 1)
 * restores that walk for parity tests against the reference's own traces.  The LRU stamp is a strictly
 * increasing counter instead of time.time().  One tree per handle; calls on one handle must not overlap. */
#ifndef MI355RADIX_H
#define MI355RADIX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int m355r_version(void);

/* PagedRadixCache.__init__ / reset (radix_cache.py:73-99): a root that is never evicted (lock_ref 1). */
void *m355r_create(int32_t page_size);
void m355r_destroy(void *tree);
void m355r_reset(void *tree);
void m355r_set_reference_walk(void *tree, int32_t on);   /* default 0: stop at a split (see above) */

/* match_prefix (radix_cache.py:101-125).  `key` is cut to a multiple of page_size.  Writes the matched pages to
 * out_pages (up to cap) and the last node reached to *node (an opaque handle for the lock calls; the root for an
 * empty key).  Returns the number of matched pages - which, with the reference walk switched on, can be LARGER than
 * the key's page count; when it exceeds cap, fetch the full list with m355r_last_match (the tree is not walked
 * again) - or -1 on a bad argument. */
int64_t m355r_match_prefix(void *tree, const int64_t *key, int64_t n_tokens, int32_t *out_pages, int64_t cap,
                           int64_t *node);
/* The pages of the latest m355r_match_prefix on this tree (no counterpart in the reference: a buffer hand-over). */
int64_t m355r_last_match(void *tree, int32_t *out_pages, int64_t cap);

/* insert (radix_cache.py:127-143): key cut to full pages, pages[i] = physical page of chunk i.  Returns the
 * number of TOKENS of the key that were already in the tree, or -1 on a bad argument. */
int64_t m355r_insert(void *tree, const int64_t *key, int64_t n_tokens, const int32_t *pages, int64_t n_pages);

/* inc_lock_ref / dec_lock_ref (radix_cache.py:223-252) from `node` up to the root; return the change of the
 * evictable size in tokens (<= 0 for inc, >= 0 for dec).  A handle names its node until that node is evicted or the
 * tree is reset; a handle whose node is gone is ignored (returns 0) - lock a matched node before anything that can
 * evict, as the reference's callers do (engine/batch.py:128-150). */
int64_t m355r_inc_lock_ref(void *tree, int64_t node);
int64_t m355r_dec_lock_ref(void *tree, int64_t node);

/* evict (radix_cache.py:202-221): frees unlocked leaves, least recently used first, until num_pages pages are
 * freed or none is left.  The freed pages are written to out_pages in the order the reference hands them to
 * KVCache._free_pages (leaf by leaf).  Returns their number (may exceed num_pages by the last leaf's rest), or
 * -1 when cap is too small (nothing is evicted then: cap >= total_size / page_size is always enough). */
int64_t m355r_evict(void *tree, int64_t num_pages, int32_t *out_pages, int64_t cap);

int64_t m355r_evictable_size(void *tree);   /* tokens (radix_cache.py:194-196) */
int64_t m355r_protected_size(void *tree);   /* tokens (:198-200) */
int64_t m355r_total_size(void *tree);       /* tokens in all nodes (:323-334) */
int64_t m355r_node_count(void *tree);       /* nodes including the root (diagnostics) */

#ifdef __cplusplus
}
#endif
#endif

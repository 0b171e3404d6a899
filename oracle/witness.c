/* ORACLE (test infrastructure, not product code).
 *
 * Deterministic witness synthesis: the counterpart of TransactionMetadata::build_random
 * (src/lib.rs:235-465) and schnorr::sign (src/schnorr/mod.rs:197-217), seeded with SplitMix64
 * instead of OsRng.  Structure followed: random sender accounts -> random distinct receivers ->
 * per transaction: record root, sender path, apply transfer, record receiver path
 * (src/lib.rs:341-422) -> sign (src/lib.rs:435-447).
 *
 * Two documented departures (neither touches the AIR):
 *  - The account tree is our own Rescue `merge` tree (the fork's MerkleTree::build_empty /
 *    update_leaf are not in the tree); empty leaves are the all-zero digest.
 *  - The scalar field order of the curve is not available in the reference tree, so signatures
 *    are made with integer arithmetic only: secret keys are small integers
 *    sk in [1,8], the nonce r is a ~258-bit integer, R = r*G, h = hash(R.x, msg) as a 255-bit
 *    integer and s = r - sk*h is accepted when 0 <= s < 2^255.  Then s*G + h*(sk*G) = R holds in
 *    the group whatever its order, which is what the in-circuit verification checks.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "gadgets.h"

static uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* k * base (affine) -> affine, MSB-first double-and-add with the complete formulas of ecc.rs */
void cso_ecc_scalar_mul_affine(const uint64_t *k, unsigned n_limbs, const uint64_t *base12, uint64_t *out12) {
    fp acc[PROJ + 1];
    memset(acc, 0, sizeof acc);
    acc[COORD] = FP_ONE; /* identity (0 : 1 : 0) */
    for (int i = (int)n_limbs * 64 - 1; i >= 0; i--) {
        ecc_double(acc);
        if ((k[i / 64] >> (i % 64)) & 1) ecc_add_mixed(acc, base12);
    }
    fp6 zi = fp6_inv(fp6_load(acc + 12));
    fp6_store(out12, fp6_mul(fp6_load(acc), zi));
    fp6_store(out12 + 6, fp6_mul(fp6_load(acc + 6), zi));
}

/* y^2 == x^3 + x + B3/3 */
int cso_ecc_on_curve_affine(const uint64_t *q) {
    fp6 x = fp6_load(q), y = fp6_load(q + 6), b3 = fp6_b3();
    fp inv3 = fp_inv(fp_from_u64(3));
    fp6 b;
    for (int i = 0; i < 6; i++) b.c[i] = fp_mul(b3.c[i], inv3);
    fp6 rhs = fp6_add(fp6_add(fp6_mul(fp6_sqr(x), x), x), b);
    fp6 lhs = fp6_sqr(y);
    return memcmp(lhs.c, rhs.c, sizeof lhs.c) == 0;
}

/* ---- Rescue account tree, sparse: node ids in heap order (1 = root, 2^depth + i = leaf i) ------ */
/* Only touched nodes are stored; an absent node of level l has the digest of an empty subtree (empty leaves are the all-zero
 * digest).  Depth 31 -- the largest the 512-row Merkle cycle admits (src/merkle/constants.rs:27-29: 8 * 31 + 7 = 255 rows)
 * and the nearest legal value to BASELINE.json's "depth 32" -- therefore costs memory for the touched paths only. */
typedef struct { uint64_t *keys; size_t *slot; size_t cap, used; fp *vals; size_t stride, nvals, vcap; } u64map_t;
static void map_init(u64map_t *m, size_t stride) {
    m->cap = 1024; m->used = 0; m->stride = stride; m->nvals = 0; m->vcap = 256;
    m->keys = calloc(m->cap, sizeof(uint64_t)); m->slot = calloc(m->cap, sizeof(size_t));
    m->vals = calloc(m->vcap * stride, sizeof(fp));
}
static void map_free(u64map_t *m) { free(m->keys); free(m->slot); free(m->vals); }
static size_t map_probe(const u64map_t *m, uint64_t key) { /* keys are nonzero (node ids >= 1, account ids stored + 1) */
    size_t i = (size_t)((key * 0x9E3779B97F4A7C15ULL) >> 20) & (m->cap - 1);
    while (m->keys[i] && m->keys[i] != key) i = (i + 1) & (m->cap - 1);
    return i;
}
static fp *map_get(const u64map_t *m, uint64_t key) {
    size_t i = map_probe(m, key);
    return m->keys[i] ? m->vals + m->slot[i] * m->stride : NULL;
}
static fp *map_insert(u64map_t *m, uint64_t key) { /* existing entry or a new zeroed one */
    if (2 * (m->used + 1) > m->cap) {
        uint64_t *ok = m->keys; size_t *os = m->slot; size_t oc = m->cap;
        m->cap *= 2; m->keys = calloc(m->cap, sizeof(uint64_t)); m->slot = calloc(m->cap, sizeof(size_t));
        for (size_t j = 0; j < oc; j++) if (ok[j]) { size_t i = map_probe(m, ok[j]); m->keys[i] = ok[j]; m->slot[i] = os[j]; }
        free(ok); free(os);
    }
    size_t i = map_probe(m, key);
    if (!m->keys[i]) {
        if (m->nvals == m->vcap) { m->vcap *= 2; m->vals = realloc(m->vals, m->vcap * m->stride * sizeof(fp)); }
        memset(m->vals + m->nvals * m->stride, 0, m->stride * sizeof(fp));
        m->keys[i] = key; m->slot[i] = m->nvals++; m->used++;
    }
    return m->vals + m->slot[i] * m->stride;
}

typedef struct { unsigned depth; uint64_t size; u64map_t nodes; fp empty[33][7]; } tree_t; /* empty[l]: digest of an empty subtree rooted at level l */

static void tree_init_empty(tree_t *t, unsigned depth) {
    t->depth = depth;
    t->size = (uint64_t)1 << depth;
    map_init(&t->nodes, 7);
    memset(t->empty, 0, sizeof t->empty);
    for (int lvl = (int)depth - 1; lvl >= 0; lvl--) rescue_merge(t->empty[lvl + 1], t->empty[lvl + 1], t->empty[lvl]);
}
static const fp *tree_node(const tree_t *t, uint64_t id, unsigned lvl) {
    const fp *v = map_get(&t->nodes, id);
    return v ? v : t->empty[lvl];
}
static void tree_update_leaf(tree_t *t, uint64_t index, const fp *leaf) {
    uint64_t i = t->size + index;
    unsigned lvl = t->depth;
    memcpy(map_insert(&t->nodes, i), leaf, 7 * sizeof(fp));
    for (i >>= 1; i >= 1; i >>= 1) {
        fp h[7];
        rescue_merge(tree_node(t, 2 * i, lvl), tree_node(t, 2 * i + 1, lvl), h);
        lvl--;
        memcpy(map_insert(&t->nodes, i), h, sizeof h);
    }
}
/* [leaf, sibling_0 .. sibling_{d-1}] as MerkleTree::prove returns it (src/merkle/update/trace.rs:113 uses [k+1]) */
static void tree_prove(const tree_t *t, uint64_t index, fp *path) {
    uint64_t i = t->size + index;
    memcpy(path, tree_node(t, i, t->depth), 7 * sizeof(fp));
    for (unsigned k = 0; k < t->depth; k++, i >>= 1) memcpy(path + 7 * (k + 1), tree_node(t, i ^ 1, t->depth - k), 7 * sizeof(fp));
}
static const fp *tree_root(const tree_t *t) { return tree_node(t, 1, 0); }
static void leaf_hash(const fp *val, fp *out) { rescue_merge(val, val + 7, out); } /* src/lib.rs:287-290 */

static void make_account(uint64_t *rng, fp *val, uint64_t *sk_out) {
    uint64_t sk = 1 + splitmix64(rng) % 8;
    cso_ecc_scalar_mul_affine(&sk, 1, CS_GENERATOR_MONT, val);
    val[12] = fp_from_u64(splitmix64(rng)); /* balance: BaseElement::from(next_u64) */
    val[13] = fp_from_u64(splitmix64(rng)); /* nonce */
    *sk_out = sk;
}

/* 320-bit little-endian integers for the nonce arithmetic */
typedef struct { uint64_t w[5]; } u320;
static int u320_sub(u320 *r, const u320 *a, const u320 *b) { /* returns borrow */
    unsigned __int128 br = 0;
    for (int i = 0; i < 5; i++) {
        unsigned __int128 d = (unsigned __int128)a->w[i] - b->w[i] - br;
        r->w[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
    return (int)br;
}
static void u320_mul_small(u320 *r, const uint64_t h[4], uint64_t k) {
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; i++) { c += (unsigned __int128)h[i] * k; r->w[i] = (uint64_t)c; c >>= 64; }
    r->w[4] = (uint64_t)c;
}

static void sign_tx(uint64_t *rng, const fp *msg, uint64_t sk, const fp *pkey, fp *rx_out, uint8_t *s_out) {
    (void)pkey;
    for (;;) {
        u320 r;
        for (int i = 0; i < 4; i++) r.w[i] = splitmix64(rng);
        /* r uniform in [0, (sk+2) * 2^254), which contains sk*h + [0, 2^255) for h up to 2^254 */
        uint64_t hi = splitmix64(rng) % (sk + 2); /* multiple of 2^254 */
        r.w[3] &= 0x3FFFFFFFFFFFFFFFULL;          /* low 254 bits random */
        unsigned __int128 acc = (unsigned __int128)r.w[3] + ((unsigned __int128)hi << 62);
        r.w[3] = (uint64_t)acc;
        r.w[4] = (uint64_t)(acc >> 64);
        fp R[12];
        cso_ecc_scalar_mul_affine(r.w, 5, CS_GENERATOR_MONT, R);
        fp h[7];
        cso_schnorr_hash_message(R, msg, h);
        uint64_t hl[4];
        for (int i = 0; i < 4; i++) hl[i] = fp_to_u64(h[i]);
        u320 skh, s;
        u320_mul_small(&skh, hl, sk);
        if (u320_sub(&s, &r, &skh)) continue;                 /* s < 0 */
        if (s.w[4] != 0 || (s.w[3] >> 63) != 0) continue;     /* s >= 2^255 */
        memcpy(rx_out, R, 6 * sizeof(fp));
        for (int i = 0; i < 4; i++) for (int b = 0; b < 8; b++) s_out[8 * i + b] = (uint8_t)(s.w[i] >> (8 * b));
        return;
    }
}

void cso_sign_message(uint64_t *rng_state, const uint64_t *msg28, uint64_t sk, uint64_t *rx6, uint8_t *s32) { sign_tx(rng_state, msg28, sk, msg28, rx6, s32); }

int cso_tx_witness_generate(cstark_tx_witness *w, uint64_t seed) {
    const uint32_t n = w->n_tx, depth = w->merkle_depth;
    if (n == 0 || depth == 0 || depth > 31) return -1;
    uint64_t rng = seed;
    tree_t tree;
    tree_init_empty(&tree, depth);
    const uint64_t tree_size = tree.size;
    u64map_t accounts; /* leaf index + 1 -> 14 leaf values | secret key (0 = no account yet) */
    map_init(&accounts, 15);
    uint64_t *s_idx = (uint64_t *)w->s_indices, *r_idx = (uint64_t *)w->r_indices;
    fp leaf[7];

    for (uint32_t t = 0; t < n; t++) { /* senders, src/lib.rs:273-296 */
        uint64_t i = splitmix64(&rng) % tree_size;
        s_idx[t] = i;
        fp *acc = map_insert(&accounts, i + 1);
        make_account(&rng, acc, &acc[14]);
        leaf_hash(acc, leaf);
        tree_update_leaf(&tree, i, leaf);
    }
    for (uint32_t t = 0; t < n; t++) { /* receivers, src/lib.rs:305-333 */
        uint64_t i = splitmix64(&rng) % tree_size;
        while (i == s_idx[t]) i = splitmix64(&rng) % tree_size;
        r_idx[t] = i;
        fp *acc = map_insert(&accounts, i + 1);
        if (acc[14] == 0) {
            make_account(&rng, acc, &acc[14]);
            leaf_hash(acc, leaf);
            tree_update_leaf(&tree, i, leaf);
        }
    }
    uint64_t *tx_sk = calloc(n, sizeof(uint64_t));
    for (uint32_t t = 0; t < n; t++) { /* transfers, src/lib.rs:347-422 */
        uint64_t si = s_idx[t], ri = r_idx[t];
        fp *sv = map_get(&accounts, si + 1), *rv = map_get(&accounts, ri + 1);
        uint64_t sb = fp_to_u64(sv[12]), rb = fp_to_u64(rv[12]);
        uint64_t bound = sb < UINT64_MAX - rb ? sb : UINT64_MAX - rb;
        uint64_t dv = bound ? splitmix64(&rng) % bound : 0;
        fp delta = fp_from_u64(dv);
        memcpy((fp *)w->initial_roots + 7 * t, tree_root(&tree), 7 * sizeof(fp));
        tx_sk[t] = sv[14];
        memcpy((fp *)w->s_old_values + 14 * t, sv, 14 * sizeof(fp));
        memcpy((fp *)w->r_old_values + 14 * t, rv, 14 * sizeof(fp));
        ((fp *)w->deltas)[t] = delta;
        tree_prove(&tree, si, (fp *)w->s_paths + 7 * (size_t)(depth + 1) * t);
        sv[12] = fp_sub(sv[12], delta);
        sv[13] = fp_add(sv[13], FP_ONE);
        rv[12] = fp_add(rv[12], delta);
        leaf_hash(sv, leaf);
        tree_update_leaf(&tree, si, leaf);
        leaf_hash(rv, leaf);
        tree_update_leaf(&tree, ri, leaf);
        tree_prove(&tree, ri, (fp *)w->r_paths + 7 * (size_t)(depth + 1) * t);
    }
    memcpy((fp *)w->final_root, tree_root(&tree), 7 * sizeof(fp));

    /* signatures, src/lib.rs:435-447; independent per transaction -> per-tx RNG streams */
    uint64_t sig_seed = splitmix64(&rng);
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t t = 0; t < n; t++) {
        uint64_t r2 = sig_seed ^ (0xD1B54A32D192ED03ULL * (t + 1));
        fp msg[28];
        memset(msg, 0, sizeof msg);
        memcpy(msg, w->s_old_values + 14 * t, 12 * sizeof(fp));
        memcpy(msg + 12, w->r_old_values + 14 * t, 12 * sizeof(fp));
        msg[24] = w->deltas[t];
        msg[25] = w->s_old_values[14 * t + 13];
        sign_tx(&r2, msg, tx_sk[t], msg, (fp *)w->sig_rx + 6 * t, (uint8_t *)w->sig_s + 32 * t);
    }
    free(tx_sk); map_free(&accounts); map_free(&tree.nodes);
    return 0;
}

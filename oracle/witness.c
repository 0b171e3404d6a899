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

/* ---- Rescue account tree: nodes[1] root, nodes[size + i] leaf i ------------------------------- */
typedef struct { unsigned depth; size_t size; fp *nodes; } tree_t;

static void tree_init_empty(tree_t *t, unsigned depth) {
    t->depth = depth;
    t->size = (size_t)1 << depth;
    t->nodes = calloc(2 * t->size * 7, sizeof(fp));
    fp h[7] = {0};
    for (int lvl = (int)depth - 1; lvl >= 0; lvl--) {
        fp nh[7];
        rescue_merge(h, h, nh);
        memcpy(h, nh, sizeof h);
        for (size_t i = (size_t)1 << lvl; i < ((size_t)2 << lvl); i++) memcpy(t->nodes + 7 * i, h, sizeof h);
    }
}
static void tree_update_leaf(tree_t *t, size_t index, const fp *leaf) {
    size_t i = t->size + index;
    memcpy(t->nodes + 7 * i, leaf, 7 * sizeof(fp));
    for (i >>= 1; i >= 1; i >>= 1) rescue_merge(t->nodes + 7 * (2 * i), t->nodes + 7 * (2 * i + 1), t->nodes + 7 * i);
}
/* [leaf, sibling_0 .. sibling_{d-1}] as MerkleTree::prove returns it (src/merkle/update/trace.rs:113 uses [k+1]) */
static void tree_prove(const tree_t *t, size_t index, fp *path) {
    size_t i = t->size + index;
    memcpy(path, t->nodes + 7 * i, 7 * sizeof(fp));
    for (unsigned k = 0; k < t->depth; k++, i >>= 1) memcpy(path + 7 * (k + 1), t->nodes + 7 * (i ^ 1), 7 * sizeof(fp));
}
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
    if (n == 0 || depth == 0 || depth > 24) return -1;
    uint64_t rng = seed;
    tree_t tree;
    tree_init_empty(&tree, depth);
    const size_t tree_size = tree.size;
    fp *values = calloc(tree_size * 14, sizeof(fp));
    uint64_t *sks = calloc(tree_size, sizeof(uint64_t)); /* 0 = no account yet */
    uint64_t *s_idx = (uint64_t *)w->s_indices, *r_idx = (uint64_t *)w->r_indices;
    fp leaf[7];

    for (uint32_t t = 0; t < n; t++) { /* senders, src/lib.rs:273-296 */
        size_t i = splitmix64(&rng) % tree_size;
        s_idx[t] = i;
        make_account(&rng, values + 14 * i, &sks[i]);
        leaf_hash(values + 14 * i, leaf);
        tree_update_leaf(&tree, i, leaf);
    }
    for (uint32_t t = 0; t < n; t++) { /* receivers, src/lib.rs:305-333 */
        size_t i = splitmix64(&rng) % tree_size;
        while (i == s_idx[t]) i = splitmix64(&rng) % tree_size;
        r_idx[t] = i;
        if (sks[i] == 0) {
            make_account(&rng, values + 14 * i, &sks[i]);
            leaf_hash(values + 14 * i, leaf);
            tree_update_leaf(&tree, i, leaf);
        }
    }
    uint64_t *tx_sk = calloc(n, sizeof(uint64_t));
    for (uint32_t t = 0; t < n; t++) { /* transfers, src/lib.rs:347-422 */
        size_t si = s_idx[t], ri = r_idx[t];
        fp *sv = values + 14 * si, *rv = values + 14 * ri;
        uint64_t sb = fp_to_u64(sv[12]), rb = fp_to_u64(rv[12]);
        uint64_t bound = sb < UINT64_MAX - rb ? sb : UINT64_MAX - rb;
        uint64_t dv = bound ? splitmix64(&rng) % bound : 0;
        fp delta = fp_from_u64(dv);
        memcpy((fp *)w->initial_roots + 7 * t, tree.nodes + 7, 7 * sizeof(fp));
        tx_sk[t] = sks[si];
        memcpy((fp *)w->s_old_values + 14 * t, sv, 14 * sizeof(fp));
        memcpy((fp *)w->r_old_values + 14 * t, rv, 14 * sizeof(fp));
        ((fp *)w->deltas)[t] = delta;
        tree_prove(&tree, si, (fp *)w->s_paths + 7 * (depth + 1) * t);
        sv[12] = fp_sub(sv[12], delta);
        sv[13] = fp_add(sv[13], FP_ONE);
        rv[12] = fp_add(rv[12], delta);
        leaf_hash(sv, leaf);
        tree_update_leaf(&tree, si, leaf);
        leaf_hash(rv, leaf);
        tree_update_leaf(&tree, ri, leaf);
        tree_prove(&tree, ri, (fp *)w->r_paths + 7 * (depth + 1) * t);
    }
    memcpy((fp *)w->final_root, tree.nodes + 7, 7 * sizeof(fp));

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
    free(tx_sk); free(values); free(sks); free(tree.nodes);
    return 0;
}

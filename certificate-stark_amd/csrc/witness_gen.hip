// Deterministic witness synthesis on the host: the counterpart of TransactionMetadata::build_random
// (/root/reference/src/lib.rs:235-465) and of SchnorrExample::new (src/schnorr/mod.rs:86-141) with schnorr::sign (:197-217),
// seeded with SplitMix64 instead of OsRng.  Like the reference this is CPU work outside the timed region
// (benches/state_transition.rs:21-24 builds the example before b.iter).  Structure followed: random sender accounts ->
// random distinct receivers -> per transaction: record the root and the sender's path, apply the transfer, record the
// receiver's path (src/lib.rs:341-422) -> sign (src/lib.rs:435-447).
//
// Two documented departures, neither visible to the AIR:
//  - the account tree is a plain Rescue `merge` tree whose empty leaves are the all-zero digest (the fork's
//    MerkleTree::build_empty / update_leaf are not in the reference tree);
//  - the order of the curve's scalar field is not in the reference tree either, so signatures use integer arithmetic only:
//    secret keys are small integers sk in [1,8], the nonce r is a ~258-bit integer, R = r*G, h = hash(R.x, msg) read as a
//    255-bit integer and s = r - sk*h is accepted when 0 <= s < 2^255; then s*G + h*(sk*G) = R holds in the group whatever
//    its order, which is what the in-circuit verification checks.
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <unordered_map>
#include <vector>
#include "../../include/cstark.h"
#include "ctx.h"
#include "hostgadgets.h"

namespace {
using namespace cs::hostg;

uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// Sparse account tree: node ids in heap order (1 = root, 2^depth + i = leaf i); only touched nodes are stored, an absent node of
// level l has the digest of an empty subtree (empty leaves are the all-zero digest).  Depth 31 -- the largest the 512-row Merkle
// cycle admits (src/merkle/constants.rs:27-29) and the nearest legal value to BASELINE.json's "depth 32" -- costs memory for
// the touched paths only.
struct Digest { fp v[7]; };
struct Tree {
    unsigned depth;
    uint64_t size;
    std::unordered_map<uint64_t, Digest> nodes;
    Digest empty[33]; // empty[l]: digest of an empty subtree rooted at level l
    explicit Tree(unsigned d) : depth(d), size((uint64_t)1 << d) {
        memset(empty, 0, sizeof empty);
        for (int lvl = (int)d - 1; lvl >= 0; lvl--) merge(empty[lvl + 1].v, empty[lvl + 1].v, empty[lvl].v);
    }
    const fp *node(uint64_t id, unsigned lvl) const {
        const auto it = nodes.find(id);
        return it == nodes.end() ? empty[lvl].v : it->second.v;
    }
    const fp *root() const { return node(1, 0); }
    void update_leaf(uint64_t index, const fp *leaf) {
        uint64_t i = size + index;
        unsigned lvl = depth;
        memcpy(nodes[i].v, leaf, 56);
        for (i >>= 1; i >= 1; i >>= 1) {
            Digest h;
            merge(node(2 * i, lvl), node(2 * i + 1, lvl), h.v);
            lvl--;
            nodes[i] = h;
        }
    }
    void prove(uint64_t index, fp *path) const { // [leaf, sibling_0 .. sibling_{d-1}] (src/merkle/update/trace.rs:113 reads [k+1])
        uint64_t i = size + index;
        memcpy(path, node(i, depth), 56);
        for (unsigned k = 0; k < depth; k++, i >>= 1) memcpy(path + 7 * (k + 1), node(i ^ 1, depth - k), 56);
    }
};
struct Account { fp val[14]; uint64_t sk; }; // sk = 0: no account yet
void leaf_hash(const fp *val, fp *out) { merge(val, val + 7, out); } // src/lib.rs:287-290

void make_account(uint64_t *rng, fp *val, uint64_t *sk_out) {
    const uint64_t sk = 1 + splitmix64(rng) % 8;
    scalar_mul_affine(&sk, 1, CS_GENERATOR_MONT, val);
    val[12] = from_u64(splitmix64(rng)); // balance: BaseElement::from(next_u64)
    val[13] = from_u64(splitmix64(rng)); // nonce
    *sk_out = sk;
}

void sign(uint64_t *rng, const fp *msg28, uint64_t sk, fp *rx_out, uint8_t *s_out) {
    typedef unsigned __int128 u128;
    for (;;) {
        uint64_t r[5];
        for (int i = 0; i < 4; i++) r[i] = splitmix64(rng);
        const uint64_t hi = splitmix64(rng) % (sk + 2); // r uniform in [0, (sk+2) * 2^254): contains sk*h + [0, 2^255) for h < 2^254
        r[3] &= 0x3FFFFFFFFFFFFFFFULL;
        const u128 top = (u128)r[3] + ((u128)hi << 62);
        r[3] = (uint64_t)top;
        r[4] = (uint64_t)(top >> 64);
        fp R[12], h[7];
        scalar_mul_affine(r, 5, CS_GENERATOR_MONT, R);
        hash_message(R, msg28, h);
        uint64_t skh[5], s[5];
        u128 c = 0;
        for (int i = 0; i < 4; i++) { c += (u128)to_u64(h[i]) * sk; skh[i] = (uint64_t)c; c >>= 64; }
        skh[4] = (uint64_t)c;
        u128 br = 0;
        for (int i = 0; i < 5; i++) { const u128 d = (u128)r[i] - skh[i] - br; s[i] = (uint64_t)d; br = (d >> 64) & 1; }
        if (br || s[4] != 0 || (s[3] >> 63) != 0) continue; // s < 0 or s >= 2^255
        memcpy(rx_out, R, 48);
        for (int i = 0; i < 4; i++) for (int b = 0; b < 8; b++) s_out[8 * i + b] = (uint8_t)(s[i] >> (8 * b));
        return;
    }
}

template <class F>
void parallel_for(uint32_t n, F f) {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > n) nt = n;
    std::vector<std::thread> th;
    for (unsigned w = 0; w < nt; w++) th.emplace_back([=] { for (uint32_t t = w; t < n; t += nt) f(t); });
    for (auto &x : th) x.join();
}

} // namespace

extern "C" {

int cstark_tx_witness_generate(cstark_tx_witness *w, uint64_t seed) {
    if (!w) return cs::fail(CSTARK_ERR_INVALID_ARG, "cstark_tx_witness_generate: null argument");
    const uint32_t n = w->n_tx, depth = w->merkle_depth;
    if (n == 0 || depth == 0 || depth > 31) return cs::fail(CSTARK_ERR_INVALID_ARG, "bad transaction count / tree depth");
    if (!w->initial_roots || !w->final_root || !w->s_old_values || !w->r_old_values || !w->s_indices || !w->r_indices || !w->s_paths || !w->r_paths ||
        !w->deltas || !w->sig_rx || !w->sig_s)
        return cs::fail(CSTARK_ERR_INVALID_ARG, "witness array pointer is null (the caller allocates every array)");
    uint64_t rng = seed;
    Tree tree(depth);
    const uint64_t tree_size = tree.size;
    std::unordered_map<uint64_t, Account> accounts; // by leaf index
    std::vector<uint64_t> tx_sk(n, 0);
    uint64_t *s_idx = (uint64_t *)w->s_indices, *r_idx = (uint64_t *)w->r_indices;
    fp leaf[7];
    for (uint32_t t = 0; t < n; t++) { // senders, src/lib.rs:273-296
        const uint64_t i = splitmix64(&rng) % tree_size;
        s_idx[t] = i;
        Account &a = accounts[i];
        make_account(&rng, a.val, &a.sk);
        leaf_hash(a.val, leaf);
        tree.update_leaf(i, leaf);
    }
    for (uint32_t t = 0; t < n; t++) { // receivers, src/lib.rs:305-333
        uint64_t i = splitmix64(&rng) % tree_size;
        while (i == s_idx[t]) i = splitmix64(&rng) % tree_size;
        r_idx[t] = i;
        Account &a = accounts[i]; // value-initialised (sk = 0) when new
        if (a.sk == 0) {
            make_account(&rng, a.val, &a.sk);
            leaf_hash(a.val, leaf);
            tree.update_leaf(i, leaf);
        }
    }
    for (uint32_t t = 0; t < n; t++) { // transfers, src/lib.rs:347-422
        const uint64_t si = s_idx[t], ri = r_idx[t];
        Account &sa = accounts.at(si), &ra = accounts.at(ri);
        fp *sv = sa.val, *rv = ra.val;
        const uint64_t sb = to_u64(sv[12]), rb = to_u64(rv[12]);
        const uint64_t bound = sb < UINT64_MAX - rb ? sb : UINT64_MAX - rb;
        const fp delta = from_u64(bound ? splitmix64(&rng) % bound : 0);
        memcpy((fp *)w->initial_roots + 7 * t, tree.root(), 56);
        tx_sk[t] = sa.sk;
        memcpy((fp *)w->s_old_values + 14 * t, sv, 112);
        memcpy((fp *)w->r_old_values + 14 * t, rv, 112);
        ((fp *)w->deltas)[t] = delta;
        tree.prove(si, (fp *)w->s_paths + 7 * (depth + 1) * (size_t)t);
        sv[12] = sub(sv[12], delta);
        sv[13] = add(sv[13], ONE);
        rv[12] = add(rv[12], delta);
        leaf_hash(sv, leaf);
        tree.update_leaf(si, leaf);
        leaf_hash(rv, leaf);
        tree.update_leaf(ri, leaf);
        tree.prove(ri, (fp *)w->r_paths + 7 * (depth + 1) * (size_t)t);
    }
    memcpy((fp *)w->final_root, tree.root(), 56);
    // signatures, src/lib.rs:435-447: independent per transaction -> per-transaction random streams, all host threads
    const uint64_t sig_seed = splitmix64(&rng);
    parallel_for(n, [&](uint32_t t) {
        uint64_t r2 = sig_seed ^ (0xD1B54A32D192ED03ULL * (t + 1));
        fp msg[28] = {0}; // build_tx_message, src/lib.rs:467-481
        memcpy(msg, w->s_old_values + 14 * (size_t)t, 96);
        memcpy(msg + 12, w->r_old_values + 14 * (size_t)t, 96);
        msg[24] = w->deltas[t];
        msg[25] = w->s_old_values[14 * (size_t)t + 13];
        sign(&r2, msg, tx_sk[t], (fp *)w->sig_rx + 6 * (size_t)t, (uint8_t *)w->sig_s + 32 * (size_t)t);
    });
    return CSTARK_OK;
}

int cstark_schnorr_witness_generate(uint32_t n_sig, uint64_t seed, uint64_t *messages, uint64_t *sig_rx, uint8_t *sig_s) {
    if (!messages || !sig_rx || !sig_s || n_sig == 0) return cs::fail(CSTARK_ERR_INVALID_ARG, "cstark_schnorr_witness_generate: bad argument");
    uint64_t rng = seed;
    for (uint32_t t = 0; t < n_sig; t++) { // message = public key || 16 random elements, src/schnorr/mod.rs:94-99
        fp *msg = messages + 28 * (size_t)t;
        const uint64_t sk = 1 + splitmix64(&rng) % 8;
        scalar_mul_affine(&sk, 1, CS_GENERATOR_MONT, msg);
        for (int i = 12; i < 28; i++) msg[i] = from_u64(splitmix64(&rng));
        uint64_t r2 = rng ^ 0xA5A5A5A5DEADBEEFULL;
        sign(&r2, msg, sk, sig_rx + 6 * (size_t)t, sig_s + 32 * (size_t)t);
    }
    return CSTARK_OK;
}

} // extern "C"

// Whole-proof orchestrator: cstark_tx_prove = TransactionExample::prove (/root/reference/src/lib.rs:116-141), i.e.
// TransactionProver::build_trace followed by the engine's Prover::prove, as one C call.  Every stage runs on the GPU through
// the stage entry points of capi.hip; the host side here is only the Fiat-Shamir channel (a few hundred BLAKE3 calls), the
// FRI layer loop, and the serialisation of the openings.
//
// Protocol [UPSTREAM-RECALL winterfell v0.3, parity unpinned -- the engine is absent from the reference tree]:
//   coin      seed = H(context || public inputs); reseed(d) = H(seed || d); reseed_int(v) = H(seed || v_le64);
//             draw: counter += 1, H(seed || counter_le64), first 8 bytes LE as integer, rejected unless < p
//   order     trace root -> 115+4 coefficient pairs -> constraint root -> z -> H(T(z) || T(z w)), H(H_i(z^b)) ->
//             DEEP coefficients (alpha, beta, gamma per register; one per composition column; two for the degree
//             adjustment) -> per FRI layer: root, alpha -> H(remainder) -> proof-of-work nonce -> query positions
//   FRI       folding factor f = 4, 8 or 16, layers while the domain exceeds fri_max_remainder; layer rows are the f evaluations
//             { e[i + t N/f] } that fold into position i
//   domains   blowup factor b = 2, 4, 8 or 16, at least the AIR's constraint-evaluation blowup ce (8 / 4 / 8 / 2).  b > ce: the trace
//             table is kept in BLOCK ORDER (blake3.h, lde_coset_slot) -- b / ce blocks of ce cosets, block 0 = the constraint-
//             evaluation domain -- so the evaluators always read a plain [ce][width][n] table; commitment leaves, query positions and
//             the proof bytes are in natural order (leaf b j + k)
// The byte layout of the proof is this library's own (documented in include/cstark.h); the tests check it with a restated verifier.
#include <hip/hip_runtime.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <new>
#include <chrono>
#include <thread>
#include <vector>
#include "../../include/cstark.h"
#include "ctx.h"
#include "blake3.h"
#include "channel.h"
#include "deep.h"
#include "ext.h"
#include "range_batch.h"
#include "hostblake3.h"
#include "keccak.cuh"
#include "air_tx_host.h"
#include "hostfield.h"

namespace cs {

enum { PROVE_EVENTS = CSTARK_PROVE_NUM_STAGES + 1 };

struct ProveArena {
    int air = -1;
    unsigned log_n = 0, log_b = 0, log_f = 0;
    uint64_t *trace = nullptr, *coeffs = nullptr, *lde = nullptr, *combined = nullptr, *ccoef = nullptr, *clde = nullptr, *deep = nullptr;
    std::vector<void *> extra; // per-AIR buffers (materialised transition evaluations, SchnorrAir's public columns)
    std::vector<size_t> extra_bytes; // allocated size of every slot of `extra`
    uint8_t *tnodes = nullptr, *cnodes = nullptr;
    std::vector<uint64_t *> layer;   // FRI layer evaluations (layer[0] = DEEP composition in natural order), last = remainder
    std::vector<uint8_t *> lnodes;   // FRI layer trees
    uint32_t *d_pos = nullptr;
    uint64_t *d_pub = nullptr;    // [16] the public inputs read from the trace (first / last row of seven registers)
    uint64_t *d_shifts = nullptr; // [b] g w_(b n)^k: the coset offsets of the LDE domain (k_deep)
    uint8_t *d_open = nullptr;
    uint64_t *h_pub = nullptr; // pinned
    uint8_t *h_open = nullptr; // pinned: the openings land here (a pageable destination is staged by the runtime: ~60 us for 0.5 MB)
    size_t h_open_bytes = 0;
    size_t open_bytes = 0;
    hipEvent_t ev[PROVE_EVENTS] = {};
    bool timed = false;
    std::vector<void *> owned;
    struct ProofRun *run = nullptr; // a proof in progress between the phases of the sharded entry points (cstark_tx_shard_*)
};
void proof_run_free(struct ProofRun *r);

void prove_arena_free(ProveArena *a) {
    if (!a) return;
    proof_run_free(a->run);
    for (void *p : a->owned) (void)hipFree(p);
    if (a->h_pub) (void)hipHostFree(a->h_pub);
    if (a->h_open) (void)hipHostFree(a->h_open);
    for (hipEvent_t e : a->ev) if (e) (void)hipEventDestroy(e);
    delete a;
}

namespace {

// the proof's hash function on the host (channel, small commitments): 0 = Blake3_256, 1 = Sha3_256
void digest(uint32_t hash_fn, const uint8_t *p, size_t n, uint8_t out[32]) {
    if (hash_fn == 1) keccak::sha3_256(p, n, out);
    else hostb3::hash(p, n, out);
}

struct Coin {
    uint8_t seed[32];
    uint64_t counter = CSTARK_CONV_COIN_FIRST_COUNTER - 1; // pre-incremented by every draw
    uint32_t hash_fn = 0;
    void init(const uint8_t *p, size_t n) { digest(hash_fn, p, n, seed); counter = CSTARK_CONV_COIN_FIRST_COUNTER - 1; }
    void reseed(const uint8_t d[32]) {
        uint8_t buf[64];
        memcpy(buf, seed, 32); memcpy(buf + 32, d, 32);
        digest(hash_fn, buf, 64, seed);
        counter = CSTARK_CONV_COIN_FIRST_COUNTER - 1;
    }
    void with_int(const uint8_t s[32], uint64_t v, uint8_t out[32]) const {
        uint8_t buf[40];
        memcpy(buf, s, 32);
        for (int i = 0; i < 8; i++) buf[32 + i] = (uint8_t)(v >> (8 * i));
        digest(hash_fn, buf, 40, out);
    }
    void reseed_int(uint64_t v) { with_int(seed, v, seed); counter = CSTARK_CONV_COIN_FIRST_COUNTER - 1; }
    uint64_t next_u64() {
        uint8_t out[32];
        with_int(seed, ++counter, out);
        uint64_t v = 0;
        for (int i = 0; i < 8; i++) v |= (uint64_t)out[i] << (8 * i);
        return v;
    }
    uint64_t draw() { // a field element, memory form
        for (;;) {
            const uint64_t v = next_u64();
            if (!CSTARK_CONV_COIN_REJECT_ABOVE_P || v < host::P) return host::from_u64(v); // from_u64 reduces
        }
    }
    // the next `count` draws, in order -- the same values and the same final counter as `count` calls of draw().  Blake3 coin: the
    // candidates of eight consecutive counters per pass of the vectorised compression (hostblake3.h); candidates computed beyond the
    // last accepted one are simply not consumed.
    void draw_many(size_t count, uint64_t *out) {
        static const bool scalar = [] { const char *e = getenv("CSTARK_COIN_SCALAR"); return e && atoi(e) != 0; }(); // tuning / debugging
        if (hash_fn != 0 || scalar) { for (size_t i = 0; i < count; i++) out[i] = draw(); return; }
        size_t got = 0;
        while (got < count) {
            uint64_t cand[8];
            hostb3::coin_candidates_x8(seed, counter + 1, cand);
            for (int l = 0; l < 8 && got < count; l++) {
                counter++;
                if (!CSTARK_CONV_COIN_REJECT_ABOVE_P || cand[l] < host::P) out[got++] = host::from_u64(cand[l]);
            }
        }
    }
    void draw_integers(size_t count, uint64_t domain, std::vector<uint32_t> &out) {
        out.clear();
        while (out.size() < count) {
            const uint32_t v = (uint32_t)(next_u64() & (domain - 1));
            if (!CSTARK_CONV_QUERY_DEDUP || std::find(out.begin(), out.end(), v) == out.end()) out.push_back(v);
        }
    }
};

// digest of field elements: their little-endian bytes in memory form, or canonical (CSTARK_CONV_HASHED_ELEMENT_BYTES_MONTGOMERY)
void hash_elements(uint32_t hash_fn, const uint64_t *e, size_t n, uint8_t out[32]) {
#if CSTARK_CONV_HASHED_ELEMENT_BYTES_MONTGOMERY
    digest(hash_fn, (const uint8_t *)e, 8 * n, out); // little-endian host
#else
    std::vector<uint64_t> can(n);
    for (size_t i = 0; i < n; i++) can[i] = host::to_u64(e[i]);
    digest(hash_fn, (const uint8_t *)can.data(), 8 * n, out);
#endif
}

struct Writer {
    std::vector<uint8_t> b;
    void raw(const void *p, size_t n) { const uint8_t *q = (const uint8_t *)p; b.insert(b.end(), q, q + n); }
    void u32(uint32_t v) { raw(&v, 4); }
    void u64(uint64_t v) { raw(&v, 8); }
};
// The proof bytes go straight into the caller's buffer: a first pass only counts (dst = null), so that a buffer that is too small is left
// untouched and the required size is known; a 0.6 MB temporary per proof would be fresh pages from the allocator every time.
struct ProofWriter {
    uint8_t *dst = nullptr;
    size_t len = 0;
    void raw(const void *p, size_t n) { if (dst) memcpy(dst + len, p, n); len += n; }
    void u32(uint32_t v) { raw(&v, 4); }
    void u64(uint64_t v) { raw(&v, 8); }
};

// positions folded into the next layer's row indices, first occurrence order
std::vector<uint32_t> fold_positions(const std::vector<uint32_t> &pos, uint32_t rows) {
    std::vector<uint32_t> out;
    for (uint32_t p : pos) {
        const uint32_t r = p & (rows - 1);
        if (std::find(out.begin(), out.end(), r) == out.end()) out.push_back(r);
    }
    return out;
}

__global__ void k_gather_rows(const uint64_t *__restrict__ lde, uint32_t width, uint32_t log_n, uint32_t log_b, const uint32_t *__restrict__ pos,
                              uint64_t *__restrict__ out) {
    const uint32_t q = blockIdx.x, i = pos[q], k = i & ((1u << log_b) - 1), j = i >> log_b;
    for (uint32_t c = threadIdx.x; c < width; c += blockDim.x) out[(size_t)q * width + c] = lde[(((size_t)k * width + c) << log_n) + j];
}
// the same for a table that holds cosets [k0, k0 + nk) only, each row followed by the bottom log2(nk) siblings of its authentication path
// from the rank's own subtree heap `sub` (leaf nk j + (k - k0) at nk n + ...; phase_commit): [nq][width + 4 log_nk] words.  Rows of
// other cosets are written as zeros (the owners' rows are summed in).
__global__ void k_gather_rows_window(const uint64_t *__restrict__ lde, uint32_t width, uint32_t log_n, uint32_t log_b, uint32_t k0, uint32_t nk,
                                     const uint32_t *__restrict__ pos, uint64_t *__restrict__ out, const uint64_t *__restrict__ sub, uint32_t log_nk) {
    const uint32_t q = blockIdx.x, i = pos[q], k = i & ((1u << log_b) - 1), j = i >> log_b, stride = width + 4 * log_nk;
    const bool mine = k >= k0 && k < k0 + nk;
    for (uint32_t c = threadIdx.x; c < width; c += blockDim.x)
        out[(size_t)q * stride + c] = mine ? lde[(((size_t)(k - k0) * width + c) << log_n) + j] : 0;
    const size_t leaf = ((size_t)nk << log_n) + ((size_t)j << log_nk) + (k - k0);
    for (uint32_t t = threadIdx.x; t < 4 * log_nk; t += blockDim.x)
        out[(size_t)q * stride + width + t] = mine ? sub[4 * ((leaf >> (t >> 2)) ^ 1) + (t & 3)] : 0;
}
// the summed rows of the ranks [nq][width + 4 log_nk] -> the opened rows [nq][width] and levels [0, log_nk) of the authentication paths
// [nq][log_leaves][32 bytes] (the levels above come from the upper tree: GatherJob::lvl0)
__global__ void k_split_shard_rows(const uint64_t *__restrict__ rows, uint32_t width, uint32_t log_nk, uint32_t log_leaves, uint64_t *__restrict__ out_rows,
                                   uint64_t *__restrict__ out_paths) {
    const uint32_t q = blockIdx.x, stride = width + 4 * log_nk;
    for (uint32_t c = threadIdx.x; c < width; c += blockDim.x) out_rows[(size_t)q * width + c] = rows[(size_t)q * stride + c];
    for (uint32_t t = threadIdx.x; t < 4 * log_nk; t += blockDim.x) out_paths[(size_t)q * log_leaves * 4 + t] = rows[(size_t)q * stride + width + t];
}
// leaf digests [b][n] (coset-major, as the ranks' all-gather delivers them) -> natural order: leaf b*j + k = digest (k, j)
__global__ void k_interleave_leaves(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n, uint32_t log_b) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; // one 16-byte half of a digest
    if (t >= (n << (log_b + 1))) return;
    const size_t i = t >> 1, half = t & 1, k = i & ((1u << log_b) - 1), j = i >> log_b;
    out[t] = in[2 * (k * n + j) + half];
}
// authentication path of leaf pos[q]: siblings from the leaf level upwards
__global__ void k_gather_paths(const uint4 *__restrict__ nodes, uint32_t log_leaves, const uint32_t *__restrict__ pos, uint4 *__restrict__ out) {
    const uint32_t q = blockIdx.x;
    for (uint32_t t = threadIdx.x; t < 2 * log_leaves; t += blockDim.x) {
        const uint32_t lvl = t >> 1, half = t & 1;
        const size_t node = ((((size_t)1 << log_leaves) + pos[q]) >> lvl) ^ 1;
        out[((size_t)q * log_leaves + lvl) * 2 + half] = nodes[2 * node + half];
    }
}
// All opening gathers of a proof in ONE launch (they were twenty launches of a few microseconds each, 0.14 ms of launch latency):
// job = blockIdx.y, query = blockIdx.x.  kind 0: row pos[q] of a coset-major table (as k_gather_rows); kind 1: authentication path
// of leaf pos[q] (as k_gather_paths; a = log2 of the leaf count).
struct GatherJob { const void *src; void *out; const uint32_t *pos; uint32_t kind, a, log_n, log_b, count, log_s; const uint32_t *dcount; uint32_t lvl0; };
// log_s: block order of the table's cosets (blake3.h); dcount != null: the number of positions is read from the device (at most `count`:
// the device-side channel folds the query positions itself, channel.hip); lvl0 (paths): the levels below it are left alone (sharded
// proofs: they come from the rank that owns the leaf)
constexpr int MAX_GATHER_JOBS = 32;
struct GatherBatch { GatherJob job[MAX_GATHER_JOBS]; };
__global__ void k_gather_batch(GatherBatch b) {
    const GatherJob g = b.job[blockIdx.y];
    const uint32_t q = blockIdx.x;
    if (q >= (g.dcount ? *g.dcount : g.count)) return;
    if (g.kind == 0) {
        const uint64_t *lde = (const uint64_t *)g.src;
        uint64_t *out = (uint64_t *)g.out;
        const uint32_t width = g.a, i = g.pos[q], k = lde_coset_slot(i & ((1u << g.log_b) - 1), g.log_b, g.log_s), j = i >> g.log_b;
        for (uint32_t c = threadIdx.x; c < width; c += blockDim.x) out[(size_t)q * width + c] = lde[(((size_t)k * width + c) << g.log_n) + j];
    } else {
        const uint4 *nodes = (const uint4 *)g.src;
        uint4 *out = (uint4 *)g.out;
        const uint32_t log_leaves = g.a;
        for (uint32_t t = threadIdx.x + 2 * g.lvl0; t < 2 * log_leaves; t += blockDim.x) {
            const uint32_t lvl = t >> 1, half = t & 1;
            const size_t node = ((((size_t)1 << log_leaves) + g.pos[q]) >> lvl) ^ 1;
            out[((size_t)q * log_leaves + lvl) * 2 + half] = nodes[2 * node + half];
        }
    }
}
struct GatherList {
    GatherBatch b{};
    int n = 0;
    uint32_t max_count = 0;
    void rows(const uint64_t *lde, uint32_t width, uint32_t log_n, uint32_t log_b, const uint32_t *pos, void *out, uint32_t count, uint32_t log_s = 0,
              const uint32_t *dcount = nullptr) {
        b.job[n++] = GatherJob{lde, out, pos, 0u, width, log_n, log_b, count, log_s, dcount, 0u};
        if (count > max_count) max_count = count;
    }
    void paths(const uint8_t *nodes, uint32_t log_leaves, const uint32_t *pos, void *out, uint32_t count, const uint32_t *dcount = nullptr, uint32_t lvl0 = 0) {
        b.job[n++] = GatherJob{nodes, out, pos, 1u, log_leaves, 0u, 0u, count, 0u, dcount, lvl0};
        if (count > max_count) max_count = count;
    }
    hipError_t launch(hipStream_t st) {
        if (n == 0 || max_count == 0) return hipSuccess;
        k_gather_batch<<<dim3(max_count, (unsigned)n), 128, 0, st>>>(b);
        return hipGetLastError();
    }
};
// first / last row of seven registers from reg0: 58 = PREV_TREE_ROOT_POS (src/prover.rs:106-129), 0 = the hash chain (benches/rescue.rs:331-354)
__global__ void k_gather_pub(const uint64_t *__restrict__ trace, size_t n, uint64_t *__restrict__ out, uint32_t reg0) {
    const uint32_t t = threadIdx.x;
    if (t < 14) out[t] = trace[(size_t)(reg0 + (t % 7)) * n + (t < 7 ? 0 : n - 1)];
}

template <class T>
int dev_alloc(ProveArena *a, T **p, size_t bytes) {
    HIP_TRY(hipMalloc((void **)p, bytes));
    a->owned.push_back(*p);
    return CSTARK_OK;
}

// Slot `slot` of the arena's per-AIR buffers with at least `bytes` bytes.  A slot outlives the proof that created it (the arena is
// reused across proofs with other query counts, field extensions or AIR options), so its size is recorded and a larger request
// replaces the allocation once the stream has drained.
template <class T>
int arena_extra(cstark_ctx *c, ProveArena *a, size_t slot, T **p, size_t bytes) {
    if (a->extra.size() <= slot) { a->extra.resize(slot + 1, nullptr); a->extra_bytes.resize(slot + 1, 0); }
    if (a->extra[slot] && a->extra_bytes[slot] < bytes) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        void *old = a->extra[slot];
        a->extra[slot] = nullptr; a->extra_bytes[slot] = 0;
        a->owned.erase(std::find(a->owned.begin(), a->owned.end(), old));
        HIP_TRY(hipFree(old));
    }
    if (!a->extra[slot]) {
        void *q;
        HIP_TRY(hipMalloc(&q, bytes));
        a->owned.push_back(q); a->extra[slot] = q; a->extra_bytes[slot] = bytes;
    }
    *p = (T *)a->extra[slot];
    return CSTARK_OK;
}

unsigned ceil_log2(uint64_t x) { unsigned l = 0; while ((1ull << l) < x) l++; return l; }
unsigned num_fri_layers(unsigned log_domain, unsigned log_max_remainder, unsigned log_f) {
    unsigned l = 0;
    while (log_domain > log_max_remainder) { log_domain -= log_f; l++; }
    return l;
}

// What differs between the AIRs: how the trace is built, what goes into the channel seed, and how the combined constraint
// evaluations are produced from the extended trace and the drawn coefficients.
struct AirJob {
    int air = 0;
    uint32_t width = 0, log_n = 0, log_ce = 0, n_constraints = 0, n_assertions = 0, item = 0; // item: Merkle depth / signature count / 0
    std::vector<uint64_t> pub;      // public-input elements (memory form), appended to the seed in canonical form
    std::vector<uint8_t> pub_bytes; // further public material appended verbatim (Schnorr: the s halves of the signatures)
    int (*build)(cstark_ctx *, ProveArena *, AirJob &) = nullptr;
    // merged constraint evaluations [b][n] for ONE set of (base-field) coefficients -> out
    int (*combine)(cstark_ctx *, ProveArena *, AirJob &, const uint64_t *ta, const uint64_t *tb, const uint64_t *ba, const uint64_t *bb, uint64_t *out) = nullptr;
    // optional: the merged evaluations of m coefficient sets in one pass over the frame (extension proofs); falls back to m calls
    int (*combine_sets)(cstark_ctx *, ProveArena *, AirJob &, unsigned m, const uint64_t *const *ta, const uint64_t *const *tb, const uint64_t *const *ba,
                        const uint64_t *const *bb, uint64_t *const *outs) = nullptr;
    // Optional: the order in which the trace columns become complete when build() returns with parts of the trace still being
    // written on internal streams (TransactionAir).  The prover interpolates and extends batch after batch, waiting for a batch's
    // events first; empty = all columns at once.
    struct ColumnBatch { uint32_t col0, ncols; hipEvent_t wait[2]; };
    std::vector<ColumnBatch> batches;
    const uint64_t *pub_staging = nullptr; // pinned host copy of the public inputs, valid once every batch has been waited for
    bool public_ready = false;      // SchnorrAir: the extended public-input columns of this proof are in the arena
    bool evals_ready = false;       // sub-AIRs: the materialised transition evaluations of this proof are already in the arena
    uint32_t k0 = 0, nk = 8;        // sharded proofs (blowup 8): the LDE cosets this GPU owns
    bool sharded = false;
    bool dev_channel = false;       // the Fiat-Shamir channel runs on the device (prove_core_dev): the host never reads the public inputs
    const uint64_t *d_coefs = nullptr;   // ... and the coefficients are drawn there: alpha[115] | beta[115] | b_alpha[na] | b_beta[na] (device)
    const uint64_t *d_avalues = nullptr; // the assertion values on the device (null: the AIR's built-in constants)
    uint32_t log_b = 3;             // log2 of the blowup factor; the trace table holds its cosets in block order when log_b > log_ce
    uint64_t seed[7] = {};          // RescueAir
    uint64_t number = 0;            // RangeProofAir
    const uint64_t *bits = nullptr; // RangeProofAir, long form: the n/64 words of the value (host)
};

// A sharded proof in progress (cstark_tx_shard_*) lives in the arena's buffers between its phases.  Anything else that takes the arena --
// a whole proof on the same context, a new sharded proof, a replacement of the arena -- ends it: the later phases then fail with "phase
// called out of order" instead of building a proof from foreign buffers.
void drop_run(ProveArena *a) {
    if (a && a->run) { proof_run_free(a->run); a->run = nullptr; }
}

int get_arena(cstark_ctx *c, const AirJob &job, unsigned log_b, unsigned log_f, unsigned n_layers, size_t nq, ProveArena **out) {
    drop_run(c->arena);
    if (c->arena && c->arena->air == job.air && c->arena->log_n == job.log_n && c->arena->log_b == log_b && c->arena->log_f == log_f &&
        c->arena->layer.size() == n_layers + 1 && c->arena->open_bytes >= nq) { *out = c->arena; return CSTARK_OK; }
    if (c->arena) { HIP_TRY(hipStreamSynchronize(c->stream)); prove_arena_free(c->arena); c->arena = nullptr; }
    ProveArena *a = new (std::nothrow) ProveArena();
    if (!a) return fail(CSTARK_ERR_OOM, "host allocation failed");
    c->arena = a; // owned by the context from here on (freed with it, also after a partial failure)
    a->air = job.air; a->log_n = job.log_n; a->log_b = log_b; a->log_f = log_f;
    const size_t n = (size_t)1 << job.log_n, b = (size_t)1 << log_b, N = n * b, W = job.width, ce = (size_t)1 << job.log_ce, f = (size_t)1 << log_f;
    RC_TRY(dev_alloc(a, &a->trace, W * n * 8));
    RC_TRY(dev_alloc(a, &a->coeffs, W * n * 8));
    RC_TRY(dev_alloc(a, &a->lde, W * N * 8));
    RC_TRY(dev_alloc(a, &a->tnodes, 2 * N * 32));
    RC_TRY(dev_alloc(a, &a->combined, N * 8)); // merged evaluations on the constraint-evaluation domain, [ce][n]
    RC_TRY(dev_alloc(a, &a->ccoef, ce * n * 8));
    RC_TRY(dev_alloc(a, &a->clde, b * ce * n * 8)); // [b cosets][ce columns][n]
    RC_TRY(dev_alloc(a, &a->cnodes, 2 * N * 32));
    RC_TRY(dev_alloc(a, &a->deep, N * 8));
    size_t sz = N;
    for (unsigned l = 0; l <= n_layers; l++) {
        uint64_t *e; uint8_t *t = nullptr;
        RC_TRY(dev_alloc(a, &e, sz * 8));
        if (l < n_layers) RC_TRY(dev_alloc(a, &t, 2 * (sz / f) * 32));
        a->layer.push_back(e); a->lnodes.push_back(t);
        sz /= f;
    }
    RC_TRY(dev_alloc(a, &a->d_pos, 4 * 256 * (n_layers + 2)));
    RC_TRY(dev_alloc(a, &a->d_pub, 16 * 8));
    RC_TRY(dev_alloc(a, &a->d_shifts, b * 8));
    {
        std::vector<uint64_t> sh(b);
        const uint64_t wbn = host::root_of_unity(job.log_n + log_b);
        uint64_t v = host::lde_offset();
        for (size_t k = 0; k < b; k++) { sh[k] = v; v = host::mul(v, wbn); }
        HIP_TRY(hipMemcpy(a->d_shifts, sh.data(), b * 8, hipMemcpyHostToDevice)); // once per arena
    }
    // openings: per query a trace row + path, a composition row + path, per layer a row of f + path
    a->open_bytes = nq;
    const size_t log_N = job.log_n + log_b;
    const size_t per_q = W * 8 + ce * 8 + 2 * log_N * 32 + (size_t)n_layers * (f * 8 + log_N * 32);
    RC_TRY(dev_alloc(a, &a->d_open, per_q * nq + 256));
    for (hipEvent_t &e : a->ev) HIP_TRY(hipEventCreate(&e));
    *out = a;
    return CSTARK_OK;
}

// The option values the reference passes (src/lib.rs:78-86; blowup 4 in src/merkle/update/tests.rs:41-52, src/range/tests.rs:87-98 and
// benches/rescue.rs:370-378; -b / -f on the command line, examples/state-transition.rs:33-34, :46-47).  log_ce: the AIR's
// constraint-evaluation blowup -- a blowup factor below it cannot hold the composition polynomial (the engine refuses it too).
int check_options(const cstark_options *opt, unsigned log_ce, unsigned *log_rem_out, unsigned *log_b_out, unsigned *log_f_out) {
    const uint32_t b = opt->blowup_factor, f = opt->fri_folding_factor;
    if (b != 2 && b != 4 && b != 8 && b != 16) return fail(CSTARK_ERR_UNSUPPORTED, "blowup_factor must be 2, 4, 8 or 16");
    if (b < (1u << log_ce)) return fail(CSTARK_ERR_INVALID_ARG, "blowup_factor below the AIR's constraint-evaluation blowup (TransactionAir / SchnorrAir 8, MerkleAir 4, RangeProofAir 2)");
    if (opt->hash_fn > 1) return fail(CSTARK_ERR_UNSUPPORTED, "hash_fn must be Blake3_256 (0) or Sha3_256 (1)");
    if (opt->field_extension > 2) return fail(CSTARK_ERR_INVALID_ARG, "field_extension must be None (0), Quadratic (1) or Cubic (2)");
    if (f != 4 && f != 8 && f != 16) return fail(CSTARK_ERR_UNSUPPORTED, "fri_folding_factor must be 4, 8 or 16");
    *log_b_out = b == 2 ? 1 : b == 4 ? 2 : b == 8 ? 3 : 4;
    *log_f_out = f == 4 ? 2 : f == 8 ? 3 : 4;
    if (opt->num_queries == 0 || opt->num_queries > 128) return fail(CSTARK_ERR_INVALID_ARG, "num_queries must be 1..128");
    if (opt->grinding_factor > 32) return fail(CSTARK_ERR_INVALID_ARG, "grinding_factor must be at most 32");
    unsigned log_rem = 0;
    while ((1u << log_rem) < opt->fri_max_remainder) log_rem++;
    if ((1u << log_rem) != opt->fri_max_remainder || log_rem < 7 || log_rem > 10) return fail(CSTARK_ERR_INVALID_ARG, "fri_max_remainder must be a power of two in 128..1024");
    *log_rem_out = log_rem;
    return CSTARK_OK;
}

// Interpolation and extension of the trace columns (cosets [job.k0, job.k0 + job.nk)); records the two stage events (after the
// interpolation, after the extension).  With column batches (AirJob::batches) the complete columns go first -- interpolated AND
// extended while the internal streams still write the later ones -- so the "interpolate" stage time then also holds the extension
// of the earlier batches.
// Extension of columns [col0, col0 + ncols) of the trace: the cosets [k0, k0 + nk) of a sharded proof, or all of them -- in block
// order when the blowup factor exceeds the AIR's constraint-evaluation blowup: block r = the blowup-ce extension with offset g w_(b n)^r
int lde_trace(cstark_ctx *c, ProveArena *a, const AirJob &job, uint32_t col0, uint32_t ncols) {
    const uint32_t W = job.width, log_n = job.log_n, log_b = job.log_b;
    if (job.sharded || log_b <= job.log_ce) return lde_column_range(c, a->coeffs, a->lde, W, col0, ncols, log_n, log_b, host::lde_offset(), job.k0, job.nk);
    const size_t n = (size_t)1 << log_n, ce = (size_t)1 << job.log_ce;
    const uint64_t wbn = host::root_of_unity(log_n + log_b);
    uint64_t offset = host::lde_offset();
    for (uint32_t r = 0; r < (1u << (log_b - job.log_ce)); r++) {
        RC_TRY(lde_column_range(c, a->coeffs, a->lde + (size_t)r * ce * W * n, W, col0, ncols, log_n, job.log_ce, offset, 0, (uint32_t)ce));
        offset = host::mul(offset, wbn);
    }
    return CSTARK_OK;
}
int commit_columns(cstark_ctx *c, ProveArena *a, AirJob &job, hipStream_t st, int &evi) {
    const uint32_t W = job.width, log_n = job.log_n;
    const size_t n = (size_t)1 << log_n;
    if (job.batches.empty()) {
        RC_TRY(cstark_interpolate_columns(c, a->trace, a->coeffs, W, log_n));
        HIP_TRY(hipEventRecord(a->ev[evi++], st));
        RC_TRY(lde_trace(c, a, job, 0, W));
        HIP_TRY(hipEventRecord(a->ev[evi++], st));
        return CSTARK_OK;
    }
    for (size_t i = 0; i < job.batches.size(); i++) {
        const AirJob::ColumnBatch &cb = job.batches[i];
        for (hipEvent_t e : cb.wait)
            if (e) HIP_TRY(hipStreamWaitEvent(st, e, 0));
        RC_TRY(cstark_interpolate_columns(c, a->trace + (size_t)cb.col0 * n, a->coeffs + (size_t)cb.col0 * n, cb.ncols, log_n));
        if (i + 1 == job.batches.size()) HIP_TRY(hipEventRecord(a->ev[evi++], st));
        RC_TRY(lde_trace(c, a, job, cb.col0, cb.ncols));
    }
    HIP_TRY(hipEventRecord(a->ev[evi++], st));
    return CSTARK_OK;
}

} // namespace

// Proof of work: the smallest nonce >= 1 whose digest with the seed has `bits` low zero bits (0 bits: nonce 1).  A search of 2^bits
// hashes in sequence on the host costs 10 ms at 16 bits (43 ms with the Sha3 coin); from 12 bits on the GPU searches 2^22 nonces per launch --
// chunks in increasing order and an atomic minimum inside a chunk, so the nonce is the one the sequential search finds.
// CSTARK_GRIND_DEVICE=0: always on the host.
int grind_nonce(cstark_ctx *c, ProveArena *a, const Coin &coin, unsigned bits, uint64_t *nonce_out) {
    static const bool dev_env = [] { const char *e = getenv("CSTARK_GRIND_DEVICE"); return !e || atoi(e) != 0; }();
    if (bits == 0 || !dev_env || bits < 12) {
        uint64_t nonce = 1;
        for (;; nonce++) {
            uint8_t out[32];
            coin.with_int(coin.seed, nonce, out);
            uint64_t v = 0;
            for (int i = 0; i < 8; i++) v |= (uint64_t)out[i] << (8 * i);
            if (bits == 0 || (v & ((1ull << bits) - 1)) == 0) break;
        }
        *nonce_out = nonce;
        return CSTARK_OK;
    }
    unsigned long long *d_found; // [found | seed (Sha3 coin: read from device memory)]
    RC_TRY(arena_extra(c, a, 42, &d_found, 64));
    constexpr uint64_t CHUNK = (uint64_t)1 << 22;
    if (coin.hash_fn == 1) {
        HIP_TRY(hipMemcpyAsync(d_found + 1, coin.seed, 32, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemsetAsync(d_found, 0xFF, 8, c->stream));
    }
    for (uint64_t base = 1;; base += CHUNK) {
        if (coin.hash_fn == 1) HIP_TRY(cs::grind_batch_chunk_sha3((const uint64_t *)(d_found + 1), 1, base, CHUNK, bits, d_found, c->stream));
        else HIP_TRY(cs::grind_chunk(coin.seed, base, CHUNK, bits, d_found, c->stream));
        unsigned long long found = 0;
        HIP_TRY(hipMemcpyAsync(&found, d_found, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(cs::stream_wait(c->stream));
        if (found != ~0ull) { *nonce_out = found; return CSTARK_OK; }
        if (base > ((uint64_t)1 << 44)) return fail(CSTARK_ERR_HIP, "proof of work: no nonce found"); // 2^-(2^12) to get here at 32 bits
    }
}

// Prover::prove for any of the AIRs over the base field, as a sequence of phases.  On one GPU (prove_core) they run back to back;
// the sharded entry points (cstark_tx_shard_*: one proof across several GPUs by LDE coset) run them with the ranks' all-gathers in
// between -- leaf digests after `commit`, merged evaluations after `evaluate`, the opened trace rows after `compose`.
struct ProofRun {
    cstark_options opt{};
    AirJob job;
    unsigned log_rem = 0, n_layers = 0, log_b = 3, log_f = 2;
    Coin coin;
    uint8_t trace_root[32] = {}, cons_root[32] = {}, rem_commit[32] = {};
    std::vector<uint64_t> ta, tb, ba, bb, ood_trace, ood_comp, remainder;
    std::vector<uint8_t> layer_roots;
    uint64_t nonce = 0;
    std::vector<uint32_t> positions;
    std::vector<std::vector<uint32_t>> lpos;
    int evi = 0, phase = 0; // phase: 1 commit, 2 evaluate, 3 compose done
    bool sharded() const { return job.sharded; }
    unsigned log_s() const { return job.sharded ? 0 : log_b - job.log_ce; } // block order of the trace table's cosets
};
void proof_run_free(ProofRun *r) { delete r; }

namespace {

#define STAGE() HIP_TRY(hipEventRecord(a->ev[R.evi++], st))

// ---- phase 1: trace, its interpolation and extension, row hashes of the owned cosets ---------------------------------------------
// d_leaves_local (sharded only): [n][32], the roots of the rank's subtrees -- the nk leaves 8 j + k0 .. 8 j + k0 + nk - 1 of row j are a
// complete subtree of the trace tree, so the rank hashes the bottom log2(nk) levels itself (its own heap `sub`, kept for the openings)
// and the all-gather moves 32 n bytes per rank at every world size.  Otherwise the leaves go straight into the tree.
int phase_commit(cstark_ctx *c, ProveArena *a, ProofRun &R, uint8_t *d_leaves_local) {
    AirJob &job = R.job;
    const unsigned log_n = job.log_n, log_b = R.log_b;
    const size_t n = (size_t)1 << log_n, N = n << log_b, W = job.width;
    hipStream_t st = c->stream;
    a->timed = false;
    R.evi = 0;
    STAGE();
    RC_TRY(job.build(c, a, job));
    STAGE();
    RC_TRY(commit_columns(c, a, job, st, R.evi));
    const uint32_t hf = R.opt.hash_fn;
    if (!R.sharded()) {
        RC_TRY(hash_rows_slots(c, hf, a->lde, a->tnodes + 32 * N, (uint32_t)W, log_n, log_b, R.log_s()));
    } else {
        const unsigned log_nk = ceil_log2(job.nk);
        if (log_nk == 0) { // one coset: its row digests are the subtree roots
            RC_TRY(cstark_hash_rows_fn(c, hf, a->lde, d_leaves_local, (uint32_t)W, log_n, 0, 0, 1));
        } else {           // the rank's cosets as a tree of their own: leaf nk j + (k - k0); the level with n nodes = the subtree roots
            uint8_t *sub;
            RC_TRY(arena_extra(c, a, 45, &sub, 2 * ((size_t)job.nk << log_n) * 32));
            RC_TRY(cstark_hash_rows_fn(c, hf, a->lde, sub + 32 * ((size_t)job.nk << log_n), (uint32_t)W, log_n, log_nk, 0, job.nk));
            RC_TRY(cstark_merkle_build_fn(c, hf, sub, log_n + log_nk));
            HIP_TRY(hipMemcpyAsync(d_leaves_local, sub + 32 * n, n * 32, hipMemcpyDeviceToDevice, st));
        }
    }
    R.phase = 1;
    return CSTARK_OK;
}

// ---- phase 2: trace tree, channel, coefficients, merged constraint evaluations of the owned cosets -----------------------------------
// d_leaves_all (sharded only): the all-gathered subtree roots [W][n][32], rank-major (W = 8 / nk).  d_out: [nk][n].
int phase_evaluate(cstark_ctx *c, ProveArena *a, ProofRun &R, const uint8_t *d_leaves_all, uint64_t *d_out) {
    AirJob &job = R.job;
    const unsigned log_n = job.log_n, log_b = R.log_b, log_N = log_n + log_b;
    const size_t n = (size_t)1 << log_n, N = n << log_b, W = job.width;
    hipStream_t st = c->stream;
    const uint32_t hf = R.opt.hash_fn;
    unsigned log_top = log_N; // leaves of the tree that is built here
    if (R.sharded()) { // node W j + r of the level with W n nodes = rank r's subtree root of row j; the levels from there up
        const unsigned log_w = log_b - ceil_log2(job.nk);
        log_top = log_n + log_w;
        k_interleave_leaves<<<(unsigned)(((2 * n << log_w) + 255) / 256), 256, 0, st>>>((const uint4 *)d_leaves_all, (uint4 *)(a->tnodes + 32 * (n << log_w)), n, log_w);
        HIP_TRY(hipGetLastError());
    }
    RC_TRY(cstark_merkle_build_fn(c, hf, a->tnodes, log_top));
    HIP_TRY(hipMemcpyAsync(R.trace_root, a->tnodes + 32, 32, hipMemcpyDeviceToHost, st));
    STAGE();
    HIP_TRY(cs::stream_wait(st)); // also completes the public-input copy of job.build
    static const bool hostprof = getenv("CSTARK_HOSTPROF") != nullptr; // debugging: host time between the root and the evaluation launches
    const auto hp0 = std::chrono::steady_clock::now();
    if (job.pub_staging) job.pub.assign(job.pub_staging, job.pub_staging + 14);

    Coin &coin = R.coin;
    coin.hash_fn = hf;
    {
        Writer s;
        const uint8_t ctxb[2] = {(uint8_t)W, (uint8_t)log_n};
        s.raw(ctxb, 2);
        s.u64(host::P);
        const uint8_t ob[7] = {(uint8_t)R.opt.num_queries, (uint8_t)log_b, (uint8_t)R.opt.grinding_factor, (uint8_t)R.opt.hash_fn,
                               (uint8_t)R.opt.field_extension, (uint8_t)R.opt.fri_folding_factor, (uint8_t)R.log_rem};
        s.raw(ob, 7);
        for (uint64_t v : job.pub) s.u64(host::to_u64(v)); // PublicInputs::write_into (src/air.rs:57-62 and the sub-AIRs' equivalents)
        s.raw(job.pub_bytes.data(), job.pub_bytes.size());
        coin.init(s.b.data(), s.b.size());
    }
    coin.reseed(R.trace_root);
    const size_t nc = job.n_constraints, na = job.n_assertions;
    R.ta.resize(nc); R.tb.resize(nc); R.ba.resize(na); R.bb.resize(na);
    {   // (alpha, beta) per transition constraint, then per assertion: 2 (nc + na) draws in the coin's order
        std::vector<uint64_t> dr(2 * (nc + na));
        coin.draw_many(dr.size(), dr.data());
        for (size_t i = 0; i < nc; i++) { R.ta[i] = dr[2 * i]; R.tb[i] = dr[2 * i + 1]; }
        for (size_t i = 0; i < na; i++) { R.ba[i] = dr[2 * nc + 2 * i]; R.bb[i] = dr[2 * nc + 2 * i + 1]; }
    }
    const auto hp1 = std::chrono::steady_clock::now();
    RC_TRY(job.combine(c, a, job, R.ta.data(), R.tb.data(), R.ba.data(), R.bb.data(), d_out));
    if (hostprof) {
        const auto hp2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[cstark hostprof] coefficients drawn in %.1f us, evaluation enqueued in %.1f us\n",
                std::chrono::duration<double, std::micro>(hp1 - hp0).count(), std::chrono::duration<double, std::micro>(hp2 - hp1).count());
    }
    STAGE();
    R.phase = 2;
    return CSTARK_OK;
}

// ---- phase 3: composition polynomial and its commitment, out-of-domain frame, DEEP composition, FRI, query positions -----------------
// Needs the merged evaluations on the constraint-evaluation domain, [ce][n], in a->combined and coset 0 of the extended trace at a->lde
// (the owner of coset 0).
int phase_compose(cstark_ctx *c, ProveArena *a, ProofRun &R) {
    AirJob &job = R.job;
    const cstark_options *opt = &R.opt;
    const unsigned log_n = job.log_n, log_b = R.log_b, log_f = R.log_f, log_N = log_n + log_b, log_ce = job.log_ce, n_layers = R.n_layers;
    const size_t n = (size_t)1 << log_n, b = (size_t)1 << log_b, N = n * b, W = job.width, ce = (size_t)1 << log_ce, nq = opt->num_queries;
    const uint32_t fold = 1u << log_f;
    hipStream_t st = c->stream;
    const uint32_t hf = opt->hash_fn;
    Coin &coin = R.coin;
    if (job.k0 != 0) return fail(CSTARK_ERR_INVALID_ARG, "the composition phase runs on the rank that owns coset 0");
    RC_TRY(cstark_composition_columns(c, a->combined, a->ccoef, log_n, log_ce));
    RC_TRY(cstark_lde_columns(c, a->ccoef, a->clde, (uint32_t)ce, log_n, log_b, host::lde_offset(), 0, (uint32_t)b));
    RC_TRY(cstark_hash_rows_fn(c, hf, a->clde, a->cnodes + 32 * N, (uint32_t)ce, log_n, log_b, 0, (uint32_t)b));
    RC_TRY(cstark_merkle_build_fn(c, hf, a->cnodes, log_N));
    HIP_TRY(hipMemcpyAsync(R.cons_root, a->cnodes + 32, 32, hipMemcpyDeviceToHost, st));
    STAGE();
    HIP_TRY(cs::stream_wait(st));
    coin.reseed(R.cons_root);

    // ---- out-of-domain frame ----------------------------------------------------------------------------------------------
    const uint64_t z = coin.draw();
    const uint64_t zpts[2] = {z, host::mul(z, host::root_of_unity(log_n))};
    const uint64_t zb = host::pow(z, ce);
    std::vector<uint64_t> &ood_trace = R.ood_trace, &ood_comp = R.ood_comp;
    ood_trace.assign(2 * W, 0); ood_comp.assign(ce, 0);
    RC_TRY(evaluate_ood_frames(c, a->coeffs, (uint32_t)W, a->ccoef, (uint32_t)ce, log_n, zpts, zb, ood_trace.data(), ood_comp.data()));
    uint8_t dg[32];
    hash_elements(hf, ood_trace.data(), 2 * W, dg); coin.reseed(dg);
    hash_elements(hf, ood_comp.data(), ce, dg); coin.reseed(dg);
    STAGE();

    // ---- DEEP composition -------------------------------------------------------------------------------------------------
    std::vector<uint64_t> d_alpha(W), d_beta(W), d_delta(ce);
    uint64_t deg_a, deg_b;
    {   // per register: alpha (point z), beta (point z w), then the draws only extension fields use (conjugate term); one per composition
        // column; two for the degree adjustment -- drawn in this order
        constexpr size_t PER = CSTARK_CONV_DEEP_DRAWS_PER_REGISTER;
        std::vector<uint64_t> dr(PER * W + ce + 2);
        coin.draw_many(dr.size(), dr.data());
        for (size_t i = 0; i < W; i++) { d_alpha[i] = dr[PER * i]; d_beta[i] = dr[PER * i + 1]; }
        for (size_t i = 0; i < ce; i++) d_delta[i] = dr[PER * W + i];
        deg_a = dr[PER * W + ce]; deg_b = dr[PER * W + ce + 1];
    }
    // The DEEP composition polynomial has degree < n (quotients of degree n - 2 times the linear degree adjustment): its values on
    // ONE coset determine it.  Evaluate the quotient sums on coset 0 only (1/8 of the extended trace read), interpolate there (the
    // coefficients of P(g y)) and extend to all cosets with offset 1 -- the same values as evaluating the sums at every point.
    {
        uint64_t *dcoef;
        RC_TRY(arena_extra(c, a, 40, &dcoef, n * 8));
        RC_TRY(cstark_deep_composition(c, a->lde, a->clde, (uint32_t)W, (uint32_t)ce, z, ood_trace.data(), ood_comp.data(), d_alpha.data(),
                                       d_beta.data(), d_delta.data(), deg_a, deg_b, a->deep, log_n, log_b, 0, 1));
        RC_TRY(cstark_interpolate_columns(c, a->deep, dcoef, 1, log_n));
        RC_TRY(cstark_lde_columns(c, dcoef, a->deep, 1, log_n, log_b, host::from_u64(1), 0, (uint32_t)b));
    }
    RC_TRY(cstark_interleave_cosets(c, a->deep, a->layer[0], log_n, log_b));
    STAGE();

    // ---- FRI commit phase -----------------------------------------------------------------------------------------------------
    R.layer_roots.assign(32 * (size_t)n_layers, 0);
    uint64_t offset = host::lde_offset();
    unsigned lg = log_N;
    // Blake3 coin: the layers' coin lives on the device (k_fri_coin: reseed with the layer's root, draw the folding point), so the host
    // enqueues all layers at once and collects the roots with the remainder -- one wait instead of one per layer (8 x ~20 us of GPU
    // idle time at 2^20 steps); it then replays the reseeds on its own coin.  CSTARK_FRI_DEVICE_COIN=0, and the Sha3 coin: the host
    // draws every folding point between two launches.
    static const bool dev_coin_env = [] { const char *e = getenv("CSTARK_FRI_DEVICE_COIN"); return !e || atoi(e) != 0; }();
    const bool dev_coin = dev_coin_env && hf == 0 && n_layers > 0;
    uint32_t *d_fri = nullptr; // [seed 8 words][alpha: 2 words per layer][roots: 8 words per layer]
    if (dev_coin) {
        RC_TRY(arena_extra(c, a, 41, &d_fri, (size_t)(8 + 10 * 32) * 4));
        HIP_TRY(hipMemcpyAsync(d_fri, coin.seed, 32, hipMemcpyHostToDevice, st));
    }
    for (unsigned l = 0; l < n_layers; l++) {
        const size_t rows = (size_t)1 << (lg - log_f);
        RC_TRY(cstark_hash_rows_fn(c, hf, a->layer[l], a->lnodes[l] + 32 * rows, fold, lg - log_f, 0, 0, 1));
        RC_TRY(cstark_merkle_build_fn(c, hf, a->lnodes[l], lg - log_f));
        if (dev_coin) {
            RC_TRY(fri_coin_fold_dev(c, d_fri, a->lnodes[l] + 32, (uint64_t *)(d_fri + 8) + l, d_fri + 8 + 2 * 32 + 8 * l, a->layer[l], a->layer[l + 1], lg, log_f, offset));
        } else {
            HIP_TRY(hipMemcpyAsync(&R.layer_roots[32 * l], a->lnodes[l] + 32, 32, hipMemcpyDeviceToHost, st));
            HIP_TRY(cs::stream_wait(st));
            coin.reseed(&R.layer_roots[32 * l]);
            const uint64_t alpha = coin.draw();
            RC_TRY(cstark_fri_fold(c, a->layer[l], a->layer[l + 1], lg, fold, offset, alpha));
        }
        offset = host::pow(offset, fold);
        lg -= log_f;
    }
    R.remainder.assign((size_t)1 << lg, 0);
    if (dev_coin) HIP_TRY(hipMemcpyAsync(R.layer_roots.data(), d_fri + 8 + 2 * 32, 32 * (size_t)n_layers, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(R.remainder.data(), a->layer[n_layers], R.remainder.size() * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(cs::stream_wait(st));
    if (dev_coin)
        for (unsigned l = 0; l < n_layers; l++) coin.reseed(&R.layer_roots[32 * l]); // the draws in between left no trace: reseed resets the counter
    hash_elements(hf, R.remainder.data(), R.remainder.size(), R.rem_commit);
    coin.reseed(R.rem_commit);
    STAGE();

    // ---- proof of work, query positions -------------------------------------------------------------------------------------
    uint64_t nonce = 1;
    RC_TRY(grind_nonce(c, a, coin, opt->grinding_factor, &nonce));
    R.nonce = nonce;
    coin.reseed_int(nonce);
    coin.draw_integers(nq, N, R.positions);
    R.lpos.assign(n_layers, {});
    {
        std::vector<uint32_t> cur = R.positions;
        unsigned g2 = log_N;
        for (unsigned l = 0; l < n_layers; l++) { R.lpos[l] = fold_positions(cur, 1u << (g2 - log_f)); cur = R.lpos[l]; g2 -= log_f; }
    }
    R.phase = 3;
    return CSTARK_OK;
}

// ---- phase 4: openings (gathered on the device, one copy back) and the proof bytes ---------------------------------------------------
// d_trace_rows (sharded only): the opened rows of the extended trace [nq][W], complete (summed over the ranks).
int phase_open(cstark_ctx *c, ProveArena *a, ProofRun &R, const uint64_t *d_trace_rows, uint8_t *proof, size_t capacity, size_t *proof_len) {
    AirJob &job = R.job;
    const cstark_options *opt = &R.opt;
    const unsigned log_n = job.log_n, log_b = R.log_b, log_f = R.log_f, log_N = log_n + log_b, log_ce = job.log_ce, n_layers = R.n_layers;
    const size_t W = job.width, ce = (size_t)1 << log_ce, nq = opt->num_queries, fold = (size_t)1 << log_f;
    hipStream_t st = c->stream;
    const std::vector<uint32_t> &positions = R.positions;
    const std::vector<std::vector<uint32_t>> &lpos = R.lpos;
    std::vector<uint32_t> hpos(256 * (n_layers + 1), 0);
    memcpy(hpos.data(), positions.data(), nq * 4);
    for (unsigned l = 0; l < n_layers; l++) memcpy(hpos.data() + 256 * (l + 1), lpos[l].data(), lpos[l].size() * 4);
    HIP_TRY(hipMemcpyAsync(a->d_pos, hpos.data(), hpos.size() * 4, hipMemcpyHostToDevice, st));
    uint8_t *o = a->d_open;
    size_t off = 0;
    const size_t o_trows = off; off += nq * W * 8;
    const size_t o_tpath = off; off += nq * log_N * 32;
    const size_t o_crows = off; off += nq * ce * 8;
    const size_t o_cpath = off; off += nq * log_N * 32;
    if (2 * (size_t)n_layers + 4 > MAX_GATHER_JOBS) return fail(CSTARK_ERR_UNSUPPORTED, "too many FRI layers for one opening launch");
    GatherList gl;
    uint32_t trace_lvl0 = 0;
    if (d_trace_rows) { // sharded: rows and the bottom levels of the paths come from the owning ranks
        trace_lvl0 = ceil_log2(job.nk);
        k_split_shard_rows<<<(unsigned)nq, 128, 0, st>>>(d_trace_rows, (uint32_t)W, trace_lvl0, log_N, (uint64_t *)(o + o_trows), (uint64_t *)(o + o_tpath));
        HIP_TRY(hipGetLastError());
    } else gl.rows(a->lde, (uint32_t)W, log_n, log_b, a->d_pos, o + o_trows, (uint32_t)nq, R.log_s());
    gl.paths(a->tnodes, log_N, a->d_pos, o + o_tpath, (uint32_t)nq, nullptr, trace_lvl0);
    gl.rows(a->clde, (uint32_t)ce, log_n, log_b, a->d_pos, o + o_crows, (uint32_t)nq);
    gl.paths(a->cnodes, log_N, a->d_pos, o + o_cpath, (uint32_t)nq);
    std::vector<size_t> o_lrows(n_layers), o_lpath(n_layers);
    {
        unsigned g2 = log_N;
        for (unsigned l = 0; l < n_layers; l++) {
            const unsigned np = (unsigned)lpos[l].size(), lr = g2 - log_f;
            o_lrows[l] = off; off += (size_t)np * fold * 8;
            o_lpath[l] = off; off += (size_t)np * lr * 32;
            gl.rows(a->layer[l], (uint32_t)fold, lr, 0, a->d_pos + 256 * (l + 1), o + o_lrows[l], np);
            gl.paths(a->lnodes[l], lr, a->d_pos + 256 * (l + 1), o + o_lpath[l], np);
            g2 -= log_f;
        }
    }
    HIP_TRY(gl.launch(st));
    if (a->h_open_bytes < off) {
        if (a->h_open) { HIP_TRY(hipHostFree(a->h_open)); a->h_open = nullptr; a->h_open_bytes = 0; }
        HIP_TRY(hipHostMalloc((void **)&a->h_open, off, hipHostMallocDefault));
        a->h_open_bytes = off;
    }
    const struct { const uint8_t *p; const uint8_t *data() const { return p; } } open{a->h_open};
    HIP_TRY(hipMemcpyAsync(a->h_open, o, off, hipMemcpyDeviceToHost, st));
    STAGE();
    HIP_TRY(cs::stream_wait(st));
    a->timed = true;

    // ---- serialise ----------------------------------------------------------------------------------------------------------------
    auto emit = [&](ProofWriter &wr) {
    wr.raw("CSTK", 4); wr.u32(CSTARK_PROOF_VERSION);
    wr.u32((uint32_t)job.air); wr.u32((uint32_t)W); wr.u32(log_n); wr.u32(job.item);
    wr.u32(opt->num_queries); wr.u32(opt->blowup_factor); wr.u32(opt->grinding_factor); wr.u32(opt->hash_fn); wr.u32(opt->field_extension);
    wr.u32(opt->fri_folding_factor); wr.u32(opt->fri_max_remainder);
    wr.raw(R.trace_root, 32); wr.raw(R.cons_root, 32);
    wr.u32(n_layers); wr.raw(R.layer_roots.data(), R.layer_roots.size()); wr.raw(R.rem_commit, 32);
    wr.raw(R.ood_trace.data(), R.ood_trace.size() * 8); wr.raw(R.ood_comp.data(), R.ood_comp.size() * 8);
    wr.u64(R.nonce);
    wr.raw(open.data() + o_trows, nq * W * 8); wr.raw(open.data() + o_tpath, nq * log_N * 32);
    wr.raw(open.data() + o_crows, nq * ce * 8); wr.raw(open.data() + o_cpath, nq * log_N * 32);
    {
        unsigned g2 = log_N;
        for (unsigned l = 0; l < n_layers; l++) {
            const size_t np = lpos[l].size();
            wr.u32((uint32_t)np);
            wr.raw(open.data() + o_lrows[l], np * fold * 8);
            wr.raw(open.data() + o_lpath[l], np * (g2 - log_f) * 32);
            g2 -= log_f;
        }
    }
    wr.u32((uint32_t)R.remainder.size()); wr.raw(R.remainder.data(), R.remainder.size() * 8);
    };
    ProofWriter count, out;
    emit(count);
    *proof_len = count.len;
    if (!proof || capacity < count.len) return fail(CSTARK_ERR_INVALID_ARG, "proof buffer too small (required size returned in *proof_len)");
    out.dst = proof;
    emit(out);
    return CSTARK_OK;
}
#undef STAGE

// options / sizes of a run, its arena
int run_setup(cstark_ctx *c, const cstark_options *opt, const AirJob &job, ProofRun &R, ProveArena **a) {
    RC_TRY(check_options(opt, job.log_ce, &R.log_rem, &R.log_b, &R.log_f));
    const unsigned log_N = job.log_n + R.log_b;
    if (log_N > 24) return fail(CSTARK_ERR_UNSUPPORTED, "the LDE domain holds at most 2^24 points (2^21 trace rows at blowup 8)");
    if (job.sharded && R.log_b != 3) return fail(CSTARK_ERR_UNSUPPORTED, "sharded proofs use blowup factor 8 (one to four of its eight cosets per rank)");
    R.opt = *opt; R.job = job;
    R.job.log_b = R.log_b;
    if (!job.sharded) { R.job.k0 = 0; R.job.nk = 1u << R.log_b; }
    R.n_layers = num_fri_layers(log_N, R.log_rem, R.log_f);
    if (opt->num_queries > ((size_t)1 << log_N) / 2) return fail(CSTARK_ERR_INVALID_ARG, "more queries than the domain supports"); // (distinct positions are drawn)
    HIP_TRY(hipSetDevice(c->device));
    return get_arena(c, R.job, R.log_b, R.log_f, R.n_layers, opt->num_queries, a);
}

int prove_core(cstark_ctx *c, const cstark_options *opt, AirJob &job, uint8_t *proof, size_t capacity, size_t *proof_len) {
    ProofRun R;
    ProveArena *a;
    RC_TRY(run_setup(c, opt, job, R, &a));
    RC_TRY(phase_commit(c, a, R, nullptr));
    RC_TRY(phase_evaluate(c, a, R, nullptr, a->combined));
    RC_TRY(phase_compose(c, a, R));
    return phase_open(c, a, R, nullptr, proof, capacity, proof_len);
}

// ---- the same proof with the Fiat-Shamir channel on the device (channel.hip) ---------------------------------------------------------
// Any of the AIRs, base field, Blake3 coin, no proof of work: every channel step -- seed, reseeds, the 238 coefficient draws, the
// out-of-domain point, the DEEP coefficients, the FRI layers' folding points (k_fri_coin), the remainder commitment, the query positions
// and their folded forms -- is a launch on the context's stream, the kernels read what was drawn from device memory, and the host waits
// ONCE, for the block that holds everything the proof bytes are written from.  Same bytes as prove_core (the tests compare both with the
// CPU prover); CSTARK_HOST_CHANNEL=1 keeps the host channel for the A/B.  One-at-a-time proving on the host channel leaves the GPU idle
// for 0.6 ms of a 28.4 ms proof, 0.39 ms of it in the five waits (profiles/r04_timeline_host_channel.txt).
int prove_core_dev(cstark_ctx *c, const cstark_options *opt, AirJob &job0, uint8_t *proof, size_t capacity, size_t *proof_len) {
    const auto hp0 = std::chrono::steady_clock::now();
    ProofRun R;
    ProveArena *a;
    job0.dev_channel = true;
    RC_TRY(run_setup(c, opt, job0, R, &a));
    AirJob &job = R.job;
    const unsigned log_n = job.log_n, log_b = R.log_b, log_f = R.log_f, log_N = log_n + log_b, log_ce = job.log_ce, n_layers = R.n_layers;
    const size_t n = (size_t)1 << log_n, b = (size_t)1 << log_b, N = n * b, W = job.width, ce = (size_t)1 << log_ce, nq = opt->num_queries;
    const uint32_t fold = 1u << log_f, hf = 0;
    const size_t rem_len = (size_t)1 << (log_N - n_layers * log_f), n_ood = 2 * W + ce;
    hipStream_t st = c->stream;
    // ---- the result block: everything the proof bytes are written from, one copy to the host at the end ------------------------------------
    size_t off = 0;
    auto take = [&off](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_troot = take(32), o_croot = take(32), o_rem = take(32), o_cnt = take(4 * 64), o_ood = take(n_ood * 8), o_lroots = take(32 * (size_t)n_layers),
                 o_remainder = take(rem_len * 8);
    const size_t o_trows = take(nq * W * 8), o_tpath = take(nq * log_N * 32), o_crows = take(nq * ce * 8), o_cpath = take(nq * log_N * 32);
    std::vector<size_t> o_lrows(n_layers), o_lpath(n_layers);
    {
        unsigned g2 = log_N;
        for (unsigned l = 0; l < n_layers; l++) { o_lrows[l] = take(nq * fold * 8); o_lpath[l] = take(nq * (g2 - log_f) * 32); g2 -= log_f; }
    }
    uint8_t *d_res;
    uint32_t *d_fri;   // [seed 8 words][alpha: 2 words per layer x 32]: the coin and the layers' folding points
    uint64_t *d_chan;  // [pts 4][scal 8][deep coefficients 2 W + ce]
    uint64_t *dcoef;
    RC_TRY(arena_extra(c, a, 44, &d_res, off));
    RC_TRY(arena_extra(c, a, 41, &d_fri, (size_t)(8 + 10 * 32) * 4));
    RC_TRY(arena_extra(c, a, 43, &d_chan, (12 + n_ood) * 8));
    RC_TRY(arena_extra(c, a, 40, &dcoef, n * 8));
    uint64_t *d_pts = d_chan, *d_scal = d_chan + 4, *d_deepc = d_chan + 12, *d_ood = (uint64_t *)(d_res + o_ood);
    uint32_t *d_cnt = (uint32_t *)(d_res + o_cnt);
    if (a->h_open_bytes < off) {
        if (a->h_open) { HIP_TRY(hipHostFree(a->h_open)); a->h_open = nullptr; a->h_open_bytes = 0; }
        HIP_TRY(hipHostMalloc((void **)&a->h_open, off, hipHostMallocDefault));
        a->h_open_bytes = off;
    }
    // the coefficient block the channel draws into: TransactionAir's evaluator reads the context's cstark_tx_coeffs block; the sub-AIRs'
    // merge takes alpha[115] | beta[115] | b_alpha[na] | b_beta[na]
    uint64_t *d_coef_block;
    const bool is_tx = job.air == CSTARK_AIR_STATE_TRANSITION;
    if (is_tx) RC_TRY(tx_coef_device_block(c, &d_coef_block));
    else RC_TRY(arena_extra(c, a, 46, &d_coef_block, (230 + 2 * (size_t)job.n_assertions) * 8));
    const uint64_t *pw, *pwinv;
    RC_TRY(plan_tables(c, log_n, &pw, &pwinv));

#define STAGE() HIP_TRY(hipEventRecord(a->ev[R.evi++], st))
    a->timed = false;
    R.evi = 0;
    STAGE();
    RC_TRY(job.build(c, a, job));
    STAGE();
    RC_TRY(commit_columns(c, a, job, st, R.evi));
    RC_TRY(hash_rows_slots(c, hf, a->lde, a->tnodes + 32 * N, (uint32_t)W, log_n, log_b, R.log_s()));
    RC_TRY(cstark_merkle_build_fn(c, hf, a->tnodes, log_N));
    STAGE();
    {   // the coin: context || public inputs (read from the trace, on the device), the trace root, the coefficient pairs
        ChanStep s{};
        s.seed = d_fri;
        if (job.air == CSTARK_AIR_SCHNORR) {
            // SchnorrAir's public inputs -- every message, R.x and s half: 304 bytes per signature -- are host data: the seed is hashed
            // here (once per uploaded witness and option set: c->schnorr_seed) and uploaded
            uint8_t key[12] = {(uint8_t)W, (uint8_t)log_n, (uint8_t)opt->num_queries, (uint8_t)log_b, (uint8_t)opt->grinding_factor, (uint8_t)opt->hash_fn,
                               (uint8_t)opt->field_extension, (uint8_t)opt->fri_folding_factor, (uint8_t)R.log_rem, 0, 0, 1};
            if (memcmp(key, c->schnorr_seed_key, sizeof key) != 0) {
                Writer sd;
                sd.raw(key, 2); sd.u64(host::P); sd.raw(key + 2, 7);
                for (uint64_t v : job.pub) sd.u64(host::to_u64(v));
                sd.raw(job.pub_bytes.data(), job.pub_bytes.size());
                hostb3::hash(sd.b.data(), sd.b.size(), c->schnorr_seed);
                memcpy(c->schnorr_seed_key, key, sizeof key);
            }
            HIP_TRY(hipMemcpyAsync(d_fri, c->schnorr_seed, 32, hipMemcpyHostToDevice, st));
        } else {
            s.init = 1; s.pub = a->d_pub; s.npub = job.air == CSTARK_AIR_RANGE ? 1 : 14;
        }
        const uint8_t hdr[17] = {(uint8_t)W, (uint8_t)log_n, (uint8_t)host::P, (uint8_t)(host::P >> 8), (uint8_t)(host::P >> 16), (uint8_t)(host::P >> 24),
                                 (uint8_t)(host::P >> 32), (uint8_t)(host::P >> 40), (uint8_t)(host::P >> 48), (uint8_t)(host::P >> 56),
                                 (uint8_t)opt->num_queries, (uint8_t)log_b, (uint8_t)opt->grinding_factor, (uint8_t)opt->hash_fn, (uint8_t)opt->field_extension,
                                 (uint8_t)opt->fri_folding_factor, (uint8_t)R.log_rem};
        memcpy(s.prefix, hdr, sizeof hdr); s.prefix_len = sizeof hdr;
        s.absorb[0].kind = CHAN_DIGEST; s.absorb[0].ptr = a->tnodes + 32; s.absorb[0].copy_out = d_res + o_troot;
        s.draw = CHAN_DRAW_COEFFS; s.a = job.n_constraints; s.b = job.n_assertions; s.stride = CSTARK_TX_NUM_CONSTRAINTS;
        s.count = 2 * (job.n_constraints + job.n_assertions); s.out = d_coef_block;
        HIP_TRY(channel_step(s, st));
    }
    if (is_tx) {
        uint64_t *outs[1] = {a->combined};
        RC_TRY(tx_evaluate_constraints_sets(c, a->lde, nullptr, 1, nullptr, outs, job.item, log_n, 3, 0, 8, true, a->d_pub));
    } else { // the sub-AIRs: the assertion values are the public inputs (MerkleAir, RescueAir), (0, number) (RangeProofAir) or built in (SchnorrAir)
        job.d_coefs = d_coef_block;
        job.d_avalues = job.air == CSTARK_AIR_RANGE ? a->d_pub + 1 : job.air == CSTARK_AIR_SCHNORR ? nullptr : a->d_pub;
        RC_TRY(job.combine(c, a, job, nullptr, nullptr, nullptr, nullptr, a->combined));
    }
    STAGE();
    RC_TRY(cstark_composition_columns(c, a->combined, a->ccoef, log_n, log_ce));
    RC_TRY(cstark_lde_columns(c, a->ccoef, a->clde, (uint32_t)ce, log_n, log_b, host::lde_offset(), 0, (uint32_t)b));
    RC_TRY(cstark_hash_rows_fn(c, hf, a->clde, a->cnodes + 32 * N, (uint32_t)ce, log_n, log_b, 0, (uint32_t)b));
    RC_TRY(cstark_merkle_build_fn(c, hf, a->cnodes, log_N));
    STAGE();
    {   // the constraint root -> the out-of-domain point z; z w and z^ce beside it
        ChanStep s{};
        s.seed = d_fri;
        s.absorb[0].kind = CHAN_DIGEST; s.absorb[0].ptr = a->cnodes + 32; s.absorb[0].copy_out = d_res + o_croot;
        s.draw = CHAN_DRAW_POINT; s.count = 1; s.b = (uint32_t)ce; s.w = host::root_of_unity(log_n); s.out = d_pts; s.out2 = d_scal;
        HIP_TRY(channel_step(s, st));
    }
    RC_TRY(ood_frames_dev(c, a->coeffs, (uint32_t)W, a->ccoef, (uint32_t)ce, log_n, d_pts, d_ood));
    STAGE();
    {   // the two halves of the frame -> the DEEP coefficients
        ChanStep s{};
        s.seed = d_fri;
        s.absorb[0].kind = CHAN_ELEMS; s.absorb[0].ptr = d_ood; s.absorb[0].count = (uint32_t)(2 * W);
        s.absorb[1].kind = CHAN_ELEMS; s.absorb[1].ptr = d_ood + 2 * W; s.absorb[1].count = (uint32_t)ce;
        s.draw = CHAN_DRAW_DEEP; s.a = (uint32_t)W; s.b = (uint32_t)ce; s.per = CSTARK_CONV_DEEP_DRAWS_PER_REGISTER;
        s.count = (uint32_t)(s.per * W + ce + 2); s.out = d_deepc; s.out2 = d_scal;
        HIP_TRY(channel_step(s, st));
    }
    {   // degree < n: the quotient sums on coset 0, interpolation, extension (prove_core)
        cs::DeepParams p{};
        p.trace_lde = a->lde; p.comp_lde = a->clde; p.w = pw; p.coef = d_deepc; p.ood = d_ood; p.shifts = a->d_shifts; p.out = a->deep;
        p.width = (uint32_t)W; p.nb = (uint32_t)ce; p.log_n = log_n; p.k0 = 0; p.scal = d_scal;
        HIP_TRY(cs::deep_composition(p, 1, st));
        RC_TRY(cstark_interpolate_columns(c, a->deep, dcoef, 1, log_n));
        RC_TRY(cstark_lde_columns(c, dcoef, a->deep, 1, log_n, log_b, host::from_u64(1), 0, (uint32_t)b));
    }
    RC_TRY(cstark_interleave_cosets(c, a->deep, a->layer[0], log_n, log_b));
    STAGE();
    {
        uint64_t offset = host::lde_offset();
        unsigned lg = log_N;
        for (unsigned l = 0; l < n_layers; l++) {
            const size_t rows = (size_t)1 << (lg - log_f);
            RC_TRY(cstark_hash_rows_fn(c, hf, a->layer[l], a->lnodes[l] + 32 * rows, fold, lg - log_f, 0, 0, 1));
            RC_TRY(cstark_merkle_build_fn(c, hf, a->lnodes[l], lg - log_f));
            uint64_t *next = l + 1 == n_layers ? (uint64_t *)(d_res + o_remainder) : a->layer[l + 1]; // the remainder lands in the result block
            RC_TRY(fri_coin_fold_dev(c, d_fri, a->lnodes[l] + 32, (uint64_t *)(d_fri + 8) + l, (uint32_t *)(d_res + o_lroots + 32 * (size_t)l), a->layer[l], next, lg,
                                     log_f, offset));
            offset = host::pow(offset, fold);
            lg -= log_f;
        }
    }
    STAGE();
    {   // remainder commitment, proof of work (none: nonce 1), query positions and their folded forms
        ChanStep s{};
        s.seed = d_fri;
        s.absorb[0].kind = CHAN_ELEMS; s.absorb[0].ptr = d_res + o_remainder; s.absorb[0].count = (uint32_t)rem_len; s.absorb[0].copy_out = d_res + o_rem;
        s.absorb[1].kind = CHAN_INT; s.absorb[1].value = 1;
        s.draw = CHAN_DRAW_QUERIES; s.count = (uint32_t)nq; s.log_domain = log_N; s.log_f = log_f; s.n_layers = n_layers; s.slot = 256;
        s.pos = a->d_pos; s.cnt = d_cnt;
        HIP_TRY(channel_step(s, st));
    }
    if (2 * (size_t)n_layers + 4 > MAX_GATHER_JOBS) return fail(CSTARK_ERR_UNSUPPORTED, "too many FRI layers for one opening launch");
    {
        GatherList gl;
        gl.rows(a->lde, (uint32_t)W, log_n, log_b, a->d_pos, d_res + o_trows, (uint32_t)nq, R.log_s());
        gl.paths(a->tnodes, log_N, a->d_pos, d_res + o_tpath, (uint32_t)nq);
        gl.rows(a->clde, (uint32_t)ce, log_n, log_b, a->d_pos, d_res + o_crows, (uint32_t)nq);
        gl.paths(a->cnodes, log_N, a->d_pos, d_res + o_cpath, (uint32_t)nq);
        unsigned g2 = log_N;
        for (unsigned l = 0; l < n_layers; l++) { // at most nq rows per layer; how many: cnt[l + 1], on the device
            const unsigned lr = g2 - log_f;
            gl.rows(a->layer[l], fold, lr, 0, a->d_pos + 256 * (l + 1), d_res + o_lrows[l], (uint32_t)nq, 0, d_cnt + l + 1);
            gl.paths(a->lnodes[l], lr, a->d_pos + 256 * (l + 1), d_res + o_lpath[l], (uint32_t)nq, d_cnt + l + 1);
            g2 -= log_f;
        }
        HIP_TRY(gl.launch(st));
    }
    HIP_TRY(hipMemcpyAsync(a->h_open, d_res, off, hipMemcpyDeviceToHost, st));
    STAGE();
    static const bool hostprof = getenv("CSTARK_HOSTPROF") != nullptr; // debugging: where the host's time goes
    const auto hp1 = std::chrono::steady_clock::now();
    HIP_TRY(cs::stream_wait_tail(st, a->ev[8])); // the only wait of the proof: sleep until the DEEP stage is done (event 8), poll through the FRI tail
    const auto hp2 = std::chrono::steady_clock::now();
    a->timed = true;
#undef STAGE
    const uint8_t *h = a->h_open;
    const uint32_t *cnt = (const uint32_t *)(h + o_cnt);
    if (cnt[0] != nq) return fail(CSTARK_ERR_HIP, "device channel: the query positions could not be drawn");
    for (unsigned l = 0; l < n_layers; l++)
        if (cnt[l + 1] == 0 || cnt[l + 1] > nq) return fail(CSTARK_ERR_HIP, "device channel: bad folded position count");
    auto emit = [&](ProofWriter &wr) {
        wr.raw("CSTK", 4); wr.u32(CSTARK_PROOF_VERSION);
        wr.u32((uint32_t)job.air); wr.u32((uint32_t)W); wr.u32(log_n); wr.u32(job.item);
        wr.u32(opt->num_queries); wr.u32(opt->blowup_factor); wr.u32(opt->grinding_factor); wr.u32(opt->hash_fn); wr.u32(opt->field_extension);
        wr.u32(opt->fri_folding_factor); wr.u32(opt->fri_max_remainder);
        wr.raw(h + o_troot, 32); wr.raw(h + o_croot, 32);
        wr.u32(n_layers); wr.raw(h + o_lroots, 32 * (size_t)n_layers); wr.raw(h + o_rem, 32);
        wr.raw(h + o_ood, n_ood * 8);
        wr.u64(1); // pow nonce: grinding_factor 0
        wr.raw(h + o_trows, nq * W * 8); wr.raw(h + o_tpath, nq * log_N * 32);
        wr.raw(h + o_crows, nq * ce * 8); wr.raw(h + o_cpath, nq * log_N * 32);
        unsigned g2 = log_N;
        for (unsigned l = 0; l < n_layers; l++) {
            const size_t np = cnt[l + 1];
            wr.u32((uint32_t)np);
            wr.raw(h + o_lrows[l], np * fold * 8);
            wr.raw(h + o_lpath[l], np * (g2 - log_f) * 32);
            g2 -= log_f;
        }
        wr.u32((uint32_t)rem_len); wr.raw(h + o_remainder, rem_len * 8);
    };
    ProofWriter count, out;
    emit(count);
    *proof_len = count.len;
    if (!proof || capacity < count.len) return fail(CSTARK_ERR_INVALID_ARG, "proof buffer too small (required size returned in *proof_len)");
    out.dst = proof;
    emit(out);
    if (hostprof) {
        const auto hp3 = std::chrono::steady_clock::now();
        auto us = [](auto d) { return std::chrono::duration<double, std::micro>(d).count(); };
        fprintf(stderr, "[cstark hostprof] device channel: enqueued in %.0f us, waited %.0f us, proof written in %.0f us\n", us(hp1 - hp0), us(hp2 - hp1), us(hp3 - hp2));
    }
    return CSTARK_OK;
}
// which channel: the device's for what prove_core_dev covers, unless CSTARK_HOST_CHANNEL=1
bool use_dev_channel(const cstark_options *opt, const AirJob &job) {
    static const bool host_env = [] { const char *e = getenv("CSTARK_HOST_CHANNEL"); return e && atoi(e) != 0; }();
    if (host_env || job.sharded || job.n_constraints > CSTARK_TX_NUM_CONSTRAINTS) return false;
    if (opt->hash_fn != 0 || opt->field_extension != 0 || opt->grinding_factor != 0) return false;
    uint32_t lb = 0, lf = 0, lr = 0;
    while ((1u << lb) < opt->blowup_factor && lb < 8) lb++;
    while ((1u << lf) < opt->fri_folding_factor && lf < 8) lf++;
    while ((1u << lr) < opt->fri_max_remainder && lr < 12) lr++;
    return lf >= 2 && job.log_n + lb > lr; // at least one FRI layer (always, but for a 64-row range proof with a large remainder)
}

// Sharded proofs: a rank that holds 2 or 4 cosets evaluates its share of the degree-split form (CSTARK_SHARD_SPLIT=0, tuning /
// debugging: every point of its cosets directly, as a rank with a single coset always does).  shard_rows = rows of n merged
// evaluations the rank hands to the all-gather: its nk cosets, or its nk / 2 even cosets + its share of the four odd ones.
bool shard_split(uint32_t nk) {
    static const bool on = [] { const char *e = getenv("CSTARK_SHARD_SPLIT"); return !e || atoi(e) != 0; }();
    return on && (nk == 2 || nk == 4);
}
uint32_t shard_rows(uint32_t nk) { return shard_split(nk) ? nk / 2 + 4 : nk; }

// first / last row of registers 58..64 -> job.pub (TransactionProver::get_pub_inputs src/prover.rs:106-129; MerkleProver alike)
int gather_roots(cstark_ctx *c, ProveArena *a, AirJob &job, uint32_t reg0 = 58) {
    const size_t n = (size_t)1 << job.log_n;
    k_gather_pub<<<1, 64, 0, c->stream>>>(a->trace, n, a->d_pub, reg0);
    HIP_TRY(hipGetLastError());
    if (job.dev_channel) return CSTARK_OK; // the device-side channel reads them where they are
    job.pub.assign(14, 0);
    HIP_TRY(hipMemcpyAsync(job.pub.data(), a->d_pub, 14 * 8, hipMemcpyDeviceToHost, c->stream)); // complete at the commitment sync
    return CSTARK_OK;
}


// ---- TransactionAir ---------------------------------------------------------------------------------------------------------------
int tx_build(cstark_ctx *c, ProveArena *a, AirJob &job) {
    // the curve ladders are latency-bound (two waves per SIMD for ~2.8 ms): they run beside the interpolation and extension of
    // the 57 registers that do not depend on them
    static const int mode = [] { const char *e = getenv("CSTARK_TRACE_OVERLAP"); return e ? atoi(e) : 1; }(); // tuning / debugging: 0 = no overlap, 2 = split launch joined at once
    if (mode == 0) {
        RC_TRY(cstark_tx_build_trace(c, a->trace));
        return gather_roots(c, a, job);
    }
    RC_TRY(tx_build_trace_split(c, a->trace));
    if (mode == 2) {
        for (hipEvent_t e : {c->ev_join, c->ev_mid, c->ev_join2}) HIP_TRY(hipStreamWaitEvent(c->stream, e, 0));
        return gather_roots(c, a, job);
    }
    // public inputs (tree roots in registers 58..64, written by the Merkle recurrence): gathered on its stream, before the event the
    // second batch waits for, into pinned memory; collected after the commitment sync
    const size_t n = (size_t)1 << job.log_n;
    if (!a->h_pub) HIP_TRY(hipHostMalloc((void **)&a->h_pub, 14 * 8, hipHostMallocDefault));
    k_gather_pub<<<1, 64, 0, c->side>>>(a->trace, n, a->d_pub, 58u);
    HIP_TRY(hipGetLastError());
    if (!job.dev_channel) HIP_TRY(hipMemcpyAsync(a->h_pub, a->d_pub, 14 * 8, hipMemcpyDeviceToHost, c->side));
    HIP_TRY(hipEventRecord(c->ev_join, c->side));
    job.pub_staging = a->h_pub;
    if (mode == 3) // two batches: everything but the curve registers once the Merkle recurrence and the message hash are done
        job.batches = {{TX_LATE_COLS, job.width - TX_LATE_COLS, {c->ev_join, c->ev_mid}}, {0, TX_LATE_COLS, {c->ev_join2, nullptr}}};
    else
        job.batches = {{TX_COPY_COLS, job.width - TX_COPY_COLS, {nullptr, nullptr}},
                       {TX_LATE_COLS, TX_COPY_COLS - TX_LATE_COLS, {c->ev_join, c->ev_mid}},
                       {0, TX_LATE_COLS, {c->ev_join2, nullptr}}};
    return CSTARK_OK;
}
int tx_combine(cstark_ctx *c, ProveArena *a, AirJob &job, const uint64_t *ta, const uint64_t *tb, const uint64_t *ba, const uint64_t *bb, uint64_t *out) {
    cstark_tx_coeffs cf;
    memcpy(cf.t_alpha, ta, sizeof cf.t_alpha); memcpy(cf.t_beta, tb, sizeof cf.t_beta);
    memcpy(cf.b_alpha, ba, sizeof cf.b_alpha); memcpy(cf.b_beta, bb, sizeof cf.b_beta);
    const uint64_t pub4[4] = {job.pub[0], job.pub[1], job.pub[7], job.pub[8]}; // get_assertions, src/air.rs:175-184
    uint64_t *outs[1] = {out};
    // all cosets on this GPU: the degree-split evaluation (the table is this prover's own extension); a window of 2 or 4 cosets of a
    // sharded proof: the rank's share of the split evaluation (rows: shard_rows); a single coset (8 ranks): every point directly
    // (blowup 16: block 0 of the trace table is the eight cosets of the constraint-evaluation domain)
    if (!job.sharded) return tx_evaluate_constraints_sets(c, a->lde, &cf, 1, pub4, outs, job.item, job.log_n, 3, 0, 8, true);
    if (shard_split(job.nk)) return tx_evaluate_constraints_shard(c, a->lde, a->coeffs, &cf, pub4, out, job.item, job.log_n, job.k0, job.nk);
    return tx_evaluate_constraints_sets(c, a->lde, &cf, 1, pub4, outs, job.item, job.log_n, 3, job.k0, job.nk, false);
}
int tx_combine_sets(cstark_ctx *c, ProveArena *a, AirJob &job, unsigned m, const uint64_t *const *ta, const uint64_t *const *tb, const uint64_t *const *ba,
                    const uint64_t *const *bb, uint64_t *const *outs) {
    cstark_tx_coeffs cf[3];
    for (unsigned q = 0; q < m; q++) {
        memcpy(cf[q].t_alpha, ta[q], sizeof cf[q].t_alpha); memcpy(cf[q].t_beta, tb[q], sizeof cf[q].t_beta);
        memcpy(cf[q].b_alpha, ba[q], sizeof cf[q].b_alpha); memcpy(cf[q].b_beta, bb[q], sizeof cf[q].b_beta);
    }
    const uint64_t pub4[4] = {job.pub[0], job.pub[1], job.pub[7], job.pub[8]};
    return tx_evaluate_constraints_sets(c, a->lde, cf, m, pub4, outs, job.item, job.log_n, 3, 0, 8, true);
}
// The sub-AIRs are evaluated on their constraint-evaluation domain -- blowup 2^log_ce: MerkleAir 4, RangeProofAir 2 -- which is block 0 of
// the trace table whatever the blowup factor of the proof: a plain [ce][width][n] extension with the domain offset.
// ---- MerkleAir (src/merkle/update) ---------------------------------------------------------------------------------------------
int merkle_build(cstark_ctx *c, ProveArena *a, AirJob &job) {
    RC_TRY(cstark_merkle_build_trace(c, a->trace));
    return gather_roots(c, a, job);
}
int merkle_combine(cstark_ctx *c, ProveArena *a, AirJob &job, const uint64_t *ta, const uint64_t *tb, const uint64_t *ba, const uint64_t *bb, uint64_t *out) {
    const size_t n = (size_t)1 << job.log_n;
    const uint32_t log_ce = job.log_ce, ce = 1u << log_ce;
    // CSTARK_MERKLE_FUSED=0 (tuning / debugging): materialise the 106 transition values and merge them generically
    static const bool fused = [] { const char *e = getenv("CSTARK_MERKLE_FUSED"); return !e || atoi(e) != 0; }();
    if (fused && job.d_coefs) return air_combine_dev(c, CSTARK_AIR_MERKLE_UPDATE, 0, 1, job.item, a->lde, nullptr, nullptr, job.d_coefs, job.d_avalues, nullptr, 0, out, job.log_n, log_ce, ce);
    if (fused) return cstark_merkle_evaluate_constraints(c, job.item, a->lde, ta, tb, ba, bb, job.pub.data(), out, job.log_n, log_ce, 0, ce);
    uint64_t *evals;
    RC_TRY(arena_extra(c, a, 0, &evals, (size_t)ce * job.n_constraints * n * 8));
    if (!job.evals_ready) RC_TRY(cstark_air_evaluate_transitions(c, CSTARK_AIR_MERKLE_UPDATE, a->lde, evals, job.item, job.log_n, log_ce, 0, ce));
    job.evals_ready = true;
    if (job.d_coefs) return air_combine_dev(c, CSTARK_AIR_MERKLE_UPDATE, 0, 0, 0, a->lde, evals, nullptr, job.d_coefs, job.d_avalues, nullptr, 0, out, job.log_n, log_ce, ce);
    return cstark_air_combine(c, CSTARK_AIR_MERKLE_UPDATE, 0, a->lde, evals, ta, tb, ba, bb, job.pub.data(), nullptr, 0, out, job.log_n, log_ce, 0, ce);
}
// ---- RangeProofAir (src/range) -------------------------------------------------------------------------------------------------------
int range_build(cstark_ctx *c, ProveArena *a, AirJob &job) {
    if (job.dev_channel) { // the public input (the number) and the two assertion values (0, the number: src/range/air.rs:79-86) for the device-side channel
        const uint64_t v[3] = {job.number, 0, job.number};
        HIP_TRY(hipMemcpyAsync(a->d_pub, v, sizeof v, hipMemcpyHostToDevice, c->stream)); // (a small pageable source is staged by the runtime before the call returns)
    }
    if (job.bits) return cstark_range_build_trace_bits(c, job.bits, job.log_n, a->trace, nullptr);
    return cstark_range_build_trace(c, job.number, a->trace);
}
int range_combine(cstark_ctx *c, ProveArena *a, AirJob &job, const uint64_t *ta, const uint64_t *tb, const uint64_t *ba, const uint64_t *bb, uint64_t *out) {
    const size_t n = (size_t)1 << job.log_n;
    const uint32_t log_ce = job.log_ce, ce = 1u << log_ce;
    uint64_t *evals;
    RC_TRY(arena_extra(c, a, 0, &evals, (size_t)ce * job.n_constraints * n * 8));
    if (!job.evals_ready) RC_TRY(cstark_air_evaluate_transitions(c, CSTARK_AIR_RANGE, a->lde, evals, job.item, job.log_n, log_ce, 0, ce));
    job.evals_ready = true;
    if (job.d_coefs) return air_combine_dev(c, CSTARK_AIR_RANGE, 0, 0, 0, a->lde, evals, nullptr, job.d_coefs, job.d_avalues, nullptr, 0, out, job.log_n, log_ce, ce);
    const uint64_t vals[2] = {0, job.number}; // get_assertions, src/range/air.rs:79-86
    return cstark_air_combine(c, CSTARK_AIR_RANGE, 0, a->lde, evals, ta, tb, ba, bb, vals, nullptr, 0, out, job.log_n, log_ce, 0, ce);
}
// ---- RescueAir (benches/rescue.rs:145-356) ---------------------------------------------------------------------------------------------
int rescue_build(cstark_ctx *c, ProveArena *a, AirJob &job) {
    RC_TRY(cstark_rescue_chain_build_trace(c, job.seed, job.item, a->trace));
    return gather_roots(c, a, job, 0); // get_pub_inputs :331-354: seed and result are the first / last row of registers 0..6
}
int rescue_combine(cstark_ctx *c, ProveArena *a, AirJob &job, const uint64_t *ta, const uint64_t *tb, const uint64_t *ba, const uint64_t *bb, uint64_t *out) {
    const size_t n = (size_t)1 << job.log_n;
    const uint32_t log_ce = job.log_ce, ce = 1u << log_ce;
    uint64_t *evals;
    RC_TRY(arena_extra(c, a, 0, &evals, (size_t)ce * job.n_constraints * n * 8));
    if (!job.evals_ready) RC_TRY(cstark_air_evaluate_transitions(c, CSTARK_AIR_RESCUE_CHAIN, a->lde, evals, 0, job.log_n, log_ce, 0, ce));
    job.evals_ready = true;
    if (job.d_coefs) return air_combine_dev(c, CSTARK_AIR_RESCUE_CHAIN, 0, 0, 0, a->lde, evals, nullptr, job.d_coefs, job.d_avalues, nullptr, 0, out, job.log_n, log_ce, ce);
    return cstark_air_combine(c, CSTARK_AIR_RESCUE_CHAIN, 0, a->lde, evals, ta, tb, ba, bb, job.pub.data(), nullptr, 0, out, job.log_n, log_ce, 0, ce);
}
// ---- SchnorrAir (src/schnorr) ---------------------------------------------------------------------------------------------------------
// the public-input columns (src/schnorr/air.rs:228-290; not committed: both sides derive them from the messages) and the sequence
// polynomials of the assertions, extended over the constraint-evaluation domain (blowup 8, whatever the proof's blowup factor): they
// depend on the public inputs only
int schnorr_public_columns(cstark_ctx *c, ProveArena *a, AirJob &job, uint64_t **aux_lde_out, uint64_t **av_lde_out, bool compute) {
    const size_t n = (size_t)1 << job.log_n;
    uint64_t *aux, *aux_co, *aux_lde, *av_co, *av_lde;
    RC_TRY(arena_extra(c, a, 1, &aux, 19 * n * 8));
    RC_TRY(arena_extra(c, a, 2, &aux_co, 19 * n * 8));
    RC_TRY(arena_extra(c, a, 3, &aux_lde, 8 * 19 * n * 8));
    RC_TRY(arena_extra(c, a, 4, &av_co, 12 * n * 8));
    RC_TRY(arena_extra(c, a, 5, &av_lde, 8 * 12 * n * 8));
    if (compute) {
        RC_TRY(cstark_schnorr_aux_columns(c, aux));
        RC_TRY(cstark_interpolate_columns(c, aux, aux_co, 19, job.log_n));
        RC_TRY(cstark_lde_columns(c, aux_co, aux_lde, 19, job.log_n, 3, host::lde_offset(), 0, 8));
        RC_TRY(cstark_schnorr_assertion_polys(c, av_co, job.log_n));
        RC_TRY(cstark_lde_columns(c, av_co, av_lde, 12, job.log_n, 3, host::lde_offset(), 0, 8));
    }
    *aux_lde_out = aux_lde; *av_lde_out = av_lde;
    return CSTARK_OK;
}
// The ladders are latency-bound (two waves per signature, one per SIMD at 512 signatures): they run on an internal stream beside the
// work that does not need them -- the public-input columns above, then the interpolation and extension of registers 37..55 (message
// hash, bit registers, limb accumulators).  CSTARK_SCHNORR_OVERLAP=0 (tuning / debugging): one stream, one thing after the other.
int schnorr_build(cstark_ctx *c, ProveArena *a, AirJob &job) {
    static const bool overlap = [] { const char *e = getenv("CSTARK_SCHNORR_OVERLAP"); return !e || atoi(e) != 0; }();
    if (!overlap) return cstark_schnorr_build_trace(c, a->trace);
    if (!c->wit_buf || c->wit.n_tx == 0 || !c->wit.msg_tail) return fail(CSTARK_ERR_INVALID_ARG, "no Schnorr witness uploaded");
    uint64_t *aux_lde, *av_lde;
    RC_TRY(schnorr_public_columns(c, a, job, &aux_lde, &av_lde, false)); // allocations (they may synchronise) before the fork
    HIP_TRY(cs::launch_schnorr_trace_split(c->wit, a->trace, c->stream, c->side, c->ev_fork, c->ev_join, c->ev_join2));
    RC_TRY(schnorr_public_columns(c, a, job, &aux_lde, &av_lde, true));
    job.public_ready = true;
    job.batches = {{37, job.width - 37, {c->ev_join, nullptr}}, {0, 37, {c->ev_join2, nullptr}}};
    return CSTARK_OK;
}
int schnorr_combine(cstark_ctx *c, ProveArena *a, AirJob &job, const uint64_t *ta, const uint64_t *tb, const uint64_t *ba, const uint64_t *bb, uint64_t *out) {
    const size_t n = (size_t)1 << job.log_n;
    // CSTARK_SCHNORR_FUSED=0 (tuning / debugging): materialise the 56 transition values and merge them generically
    static const bool fused = [] { const char *e = getenv("CSTARK_SCHNORR_FUSED"); return !e || atoi(e) != 0; }();
    uint64_t *evals = nullptr, *aux_lde, *av_lde;
    if (!fused) RC_TRY(arena_extra(c, a, 0, &evals, 8 * (size_t)job.n_constraints * n * 8));
    RC_TRY(schnorr_public_columns(c, a, job, &aux_lde, &av_lde, !job.public_ready)); // once per proof, in schnorr_build when it overlaps
    job.public_ready = true;
    if (!job.evals_ready) { // once per proof (extension proofs merge with m coefficient sets)
        if (!fused) RC_TRY(cstark_schnorr_evaluate_transitions(c, a->lde, aux_lde, evals, job.log_n, 3, 0, 8));
        job.evals_ready = true;
    }
    if (fused && job.d_coefs) return air_combine_dev(c, CSTARK_AIR_SCHNORR, job.item, 2, 0, a->lde, nullptr, aux_lde, job.d_coefs, nullptr, av_lde, 12, out, job.log_n, 3, 8);
    if (fused) return cstark_schnorr_evaluate_constraints_lde(c, job.item, a->lde, aux_lde, ta, tb, ba, bb, av_lde, 12, out, job.log_n); // own extensions: split form
    if (job.d_coefs) return air_combine_dev(c, CSTARK_AIR_SCHNORR, job.item, 0, 0, a->lde, evals, nullptr, job.d_coefs, nullptr, av_lde, 12, out, job.log_n, 3, 8);
    return cstark_air_combine(c, CSTARK_AIR_SCHNORR, job.item, a->lde, evals, ta, tb, ba, bb, nullptr, av_lde, 12, out, job.log_n, 3, 0, 8);
}


// ---- any AIR with FieldExtension::Quadratic / Cubic ------------------------------------------------------------------------------
// Base-field trace; everything the coin draws lives in the degree-m extension (ext.hip).  Coefficients multiply base-field
// constraint values, so the merged evaluations are the AIR's evaluator applied with m coefficient sets (one per component):
// TransactionAir merges all sets in one pass over the frame, the sub-AIRs merge their materialised evaluations m times.  Layout
// differences of the proof: out-of-domain values are m-tuples, composition rows hold 8 m-tuples, FRI rows and the remainder are
// component-major.
int prove_ext(cstark_ctx *c, const cstark_options *opt, AirJob &job, uint8_t *proof, size_t capacity, size_t *proof_len) {
    using namespace host;
    const unsigned m = opt->field_extension + 1;
    unsigned log_rem = 0, log_b = 3, log_f = 2;
    RC_TRY(check_options(opt, job.log_ce, &log_rem, &log_b, &log_f));
    job.log_b = log_b; job.k0 = 0; job.nk = 1u << log_b; job.sharded = false;
    const unsigned log_n = job.log_n, log_N = log_n + log_b, log_ce = job.log_ce, log_s = log_b - log_ce;
    if (log_N > 24) return fail(CSTARK_ERR_UNSUPPORTED, "the LDE domain holds at most 2^24 points (2^21 trace rows at blowup 8)");
    const size_t n = (size_t)1 << log_n, b = (size_t)1 << log_b, N = n * b, W = job.width, ce = (size_t)1 << log_ce, CW = m * ce; // CW: base columns of the composition table
    const size_t CN = ce * n; // points of the constraint-evaluation domain
    const unsigned n_layers = num_fri_layers(log_N, log_rem, log_f);
    const uint32_t fold = 1u << log_f;
    const size_t nq = opt->num_queries;
    if (nq > N / 2) return fail(CSTARK_ERR_INVALID_ARG, "more queries than the domain supports");
    HIP_TRY(hipSetDevice(c->device));
    ProveArena *a;
    RC_TRY(get_arena(c, job, log_b, log_f, n_layers, nq, &a));
    uint64_t *combined_x, *ccoef_x, *ccoefs, *cldes, *deepx;
    uint8_t *d_open;
    RC_TRY(arena_extra(c, a, 16, &combined_x, 2 * CN * 8));  // components 1, 2 of the merged evaluations
    RC_TRY(arena_extra(c, a, 17, &ccoef_x, 2 * CN * 8));     // their column coefficients
    RC_TRY(arena_extra(c, a, 18, &ccoefs, 3 * CN * 8));      // interleaved: column m i + k
    RC_TRY(arena_extra(c, a, 19, &cldes, 3 * b * CN * 8));
    RC_TRY(arena_extra(c, a, 20, &deepx, 3 * N * 8));
    RC_TRY(arena_extra(c, a, 21, &d_open, nq * (W * 8 + 192 + 2 * log_N * 32 + (size_t)n_layers * (fold * 24 + log_N * 32)) + 256));
    std::vector<uint64_t *> layer(n_layers + 1);
    {
        size_t sz = N;
        for (unsigned l = 0; l <= n_layers; l++) { RC_TRY(arena_extra(c, a, 22 + l, &layer[l], 3 * sz * 8)); sz >>= log_f; }
    }
    hipStream_t st = c->stream;
    int evi = 0;
#define STAGE() HIP_TRY(hipEventRecord(a->ev[evi++], st))
    a->timed = false;
    const uint32_t hf = opt->hash_fn;

    STAGE();
    RC_TRY(job.build(c, a, job));
    STAGE();
    RC_TRY(commit_columns(c, a, job, st, evi));
    RC_TRY(hash_rows_slots(c, hf, a->lde, a->tnodes + 32 * N, (uint32_t)W, log_n, log_b, log_s));
    RC_TRY(cstark_merkle_build_fn(c, hf, a->tnodes, log_N));
    uint8_t trace_root[32], cons_root[32];
    HIP_TRY(hipMemcpyAsync(trace_root, a->tnodes + 32, 32, hipMemcpyDeviceToHost, st));
    STAGE();
    HIP_TRY(cs::stream_wait(st));
    if (job.pub_staging) job.pub.assign(job.pub_staging, job.pub_staging + 14);

    Coin coin;
    coin.hash_fn = hf;
    {
        Writer s;
        const uint8_t ctxb[2] = {(uint8_t)W, (uint8_t)log_n};
        s.raw(ctxb, 2);
        s.u64(P);
        const uint8_t ob[7] = {(uint8_t)opt->num_queries, (uint8_t)log_b, (uint8_t)opt->grinding_factor, (uint8_t)opt->hash_fn,
                               (uint8_t)opt->field_extension, (uint8_t)opt->fri_folding_factor, (uint8_t)log_rem};
        s.raw(ob, 7);
        for (uint64_t v : job.pub) s.u64(to_u64(v));
        s.raw(job.pub_bytes.data(), job.pub_bytes.size());
        coin.init(s.b.data(), s.b.size());
    }
    coin.reseed(trace_root);
    auto draw_e = [&coin, m]() { EX x = ex_zero(); for (unsigned q = 0; q < m; q++) x.c[q] = coin.draw(); return x; };
    const size_t nc = job.n_constraints, na = job.n_assertions;
    std::vector<uint64_t> ta[3], tb[3], ba[3], bb[3];
    for (unsigned q = 0; q < m; q++) { ta[q].resize(nc); tb[q].resize(nc); ba[q].resize(na); bb[q].resize(na); }
    for (size_t i = 0; i < nc; i++) {
        const EX al = draw_e(), be = draw_e();
        for (unsigned q = 0; q < m; q++) { ta[q][i] = al.c[q]; tb[q][i] = be.c[q]; }
    }
    for (size_t i = 0; i < na; i++) {
        const EX al = draw_e(), be = draw_e();
        for (unsigned q = 0; q < m; q++) { ba[q][i] = al.c[q]; bb[q][i] = be.c[q]; }
    }
    uint64_t *comb[3] = {a->combined, combined_x, combined_x + CN}, *cco[3] = {a->ccoef, ccoef_x, ccoef_x + CN};
    if (job.combine_sets) {
        const uint64_t *pa[3], *pb[3], *qa[3], *qb[3];
        for (unsigned q = 0; q < m; q++) { pa[q] = ta[q].data(); pb[q] = tb[q].data(); qa[q] = ba[q].data(); qb[q] = bb[q].data(); }
        RC_TRY(job.combine_sets(c, a, job, m, pa, pb, qa, qb, comb));
    } else {
        for (unsigned q = 0; q < m; q++) RC_TRY(job.combine(c, a, job, ta[q].data(), tb[q].data(), ba[q].data(), bb[q].data(), comb[q]));
    }
    STAGE();
    for (unsigned q = 0; q < m; q++) RC_TRY(cstark_composition_columns(c, comb[q], cco[q], log_n, log_ce)); // [ce][n] each
    HIP_TRY(cs::interleave_set_columns(ccoefs, cco, m, (unsigned)ce, n, st)); // column m i + q = component q of composition column i
    RC_TRY(cstark_lde_columns(c, ccoefs, cldes, (uint32_t)CW, log_n, log_b, lde_offset(), 0, (uint32_t)b));
    RC_TRY(cstark_hash_rows_fn(c, hf, cldes, a->cnodes + 32 * N, (uint32_t)CW, log_n, log_b, 0, (uint32_t)b));
    RC_TRY(cstark_merkle_build_fn(c, hf, a->cnodes, log_N));
    HIP_TRY(hipMemcpyAsync(cons_root, a->cnodes + 32, 32, hipMemcpyDeviceToHost, st));
    STAGE();
    HIP_TRY(cs::stream_wait(st));
    coin.reseed(cons_root);

    const EX z = draw_e(), zw = ex_scale(z, root_of_unity(log_n)), zb = ex_pow(z, ce, m);
    std::vector<uint64_t> ood_trace(2 * m * W), raw(m * CW), ood_comp(m * ce);
    RC_TRY(cstark_evaluate_polys_at_ext(c, a->coeffs, (uint32_t)W, log_n, m, z.c, ood_trace.data()));
    RC_TRY(cstark_evaluate_polys_at_ext(c, a->coeffs, (uint32_t)W, log_n, m, zw.c, ood_trace.data() + m * W));
    RC_TRY(cstark_evaluate_polys_at_ext(c, ccoefs, (uint32_t)CW, log_n, m, zb.c, raw.data()));
    {
        EX gen = ex_zero();
        gen.c[1] = ONE; // the adjoined root
        for (size_t i = 0; i < ce; i++) { // H_i = sum_q root^q H_i,q, each component polynomial evaluated at z^ce
            EX h = ex_zero(), gq = ex_one();
            for (unsigned q = 0; q < m; q++) {
                h = ex_add(h, ex_mul(gq, ex_load(raw.data() + m * (m * i + q), m), m));
                gq = ex_mul(gq, gen, m);
            }
            for (unsigned q = 0; q < m; q++) ood_comp[m * i + q] = h.c[q];
        }
    }
    uint8_t dg[32];
    hash_elements(hf, ood_trace.data(), ood_trace.size(), dg); coin.reseed(dg);
    hash_elements(hf, ood_comp.data(), ood_comp.size(), dg); coin.reseed(dg);
    STAGE();

    std::vector<uint64_t> d_alpha(m * W), d_beta(m * W), d_delta(m * ce);
    for (size_t i = 0; i < W; i++) {
        const EX al = draw_e(), be = draw_e();
        for (int k = 2; k < CSTARK_CONV_DEEP_DRAWS_PER_REGISTER; k++) (void)draw_e(); // conjugate-term coefficient of the engine, unused here
        for (unsigned q = 0; q < m; q++) { d_alpha[m * i + q] = al.c[q]; d_beta[m * i + q] = be.c[q]; }
    }
    for (size_t i = 0; i < ce; i++) { const EX dl = draw_e(); for (unsigned q = 0; q < m; q++) d_delta[m * i + q] = dl.c[q]; }
    const EX dga = draw_e(), dgb = draw_e();
    {   // degree < n in every component: coset 0 only, then interpolation and extension per component (see prove_core)
        uint64_t *dev0, *dcoef;
        RC_TRY(arena_extra(c, a, 40, &dev0, 3 * n * 8));
        RC_TRY(arena_extra(c, a, 41, &dcoef, 3 * n * 8));
        RC_TRY(deep_composition_ext_cosets(c, a->lde, cldes, (uint32_t)W, (uint32_t)ce, m, z.c, ood_trace.data(), ood_comp.data(), d_alpha.data(),
                                           d_beta.data(), d_delta.data(), dga.c, dgb.c, dev0, log_n, log_b, 1));
        RC_TRY(cstark_interpolate_columns(c, dev0, dcoef, m, log_n));
        for (unsigned q = 0; q < m; q++) {
            RC_TRY(cstark_lde_columns(c, dcoef + q * n, deepx + q * N, 1, log_n, log_b, from_u64(1), 0, (uint32_t)b));
            RC_TRY(cstark_interleave_cosets(c, deepx + q * N, layer[0] + q * N, log_n, log_b));
        }
    }
    STAGE();

    std::vector<uint8_t> layer_roots(32 * (size_t)n_layers);
    uint64_t offset = lde_offset();
    unsigned lg = log_N;
    for (unsigned l = 0; l < n_layers; l++) {
        const size_t rows = (size_t)1 << (lg - log_f);
        RC_TRY(cstark_hash_rows_fn(c, hf, layer[l], a->lnodes[l] + 32 * rows, fold * m, lg - log_f, 0, 0, 1)); // [m][f][rows]: component-major rows
        RC_TRY(cstark_merkle_build_fn(c, hf, a->lnodes[l], lg - log_f));
        HIP_TRY(hipMemcpyAsync(&layer_roots[32 * l], a->lnodes[l] + 32, 32, hipMemcpyDeviceToHost, st));
        HIP_TRY(cs::stream_wait(st));
        coin.reseed(&layer_roots[32 * l]);
        const EX alpha = draw_e();
        RC_TRY(cstark_fri_fold_ext(c, layer[l], layer[l + 1], lg, fold, offset, m, alpha.c));
        offset = pow(offset, fold);
        lg -= log_f;
    }
    const size_t R = (size_t)1 << lg;
    std::vector<uint64_t> remainder(m * R);
    HIP_TRY(hipMemcpyAsync(remainder.data(), layer[n_layers], remainder.size() * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(cs::stream_wait(st));
    uint8_t rem_commit[32];
    hash_elements(hf, remainder.data(), remainder.size(), rem_commit);
    coin.reseed(rem_commit);
    STAGE();

    uint64_t nonce = 1;
    RC_TRY(grind_nonce(c, a, coin, opt->grinding_factor, &nonce));
    coin.reseed_int(nonce);
    std::vector<uint32_t> positions;
    coin.draw_integers(nq, N, positions);
    std::vector<std::vector<uint32_t>> lpos(n_layers);
    {
        std::vector<uint32_t> cur = positions;
        unsigned g2 = log_N;
        for (unsigned l = 0; l < n_layers; l++) { lpos[l] = fold_positions(cur, 1u << (g2 - log_f)); cur = lpos[l]; g2 -= log_f; }
    }
    std::vector<uint32_t> hpos(256 * (n_layers + 1), 0);
    memcpy(hpos.data(), positions.data(), nq * 4);
    for (unsigned l = 0; l < n_layers; l++) memcpy(hpos.data() + 256 * (l + 1), lpos[l].data(), lpos[l].size() * 4);
    HIP_TRY(hipMemcpyAsync(a->d_pos, hpos.data(), hpos.size() * 4, hipMemcpyHostToDevice, st));
    uint8_t *o = d_open;
    size_t off = 0;
    const size_t o_trows = off; off += nq * W * 8;
    const size_t o_tpath = off; off += nq * log_N * 32;
    const size_t o_crows = off; off += nq * CW * 8;
    const size_t o_cpath = off; off += nq * log_N * 32;
    if (2 * (size_t)n_layers + 4 > MAX_GATHER_JOBS) return fail(CSTARK_ERR_UNSUPPORTED, "too many FRI layers for one opening launch");
    GatherList gl;
    gl.rows(a->lde, (uint32_t)W, log_n, log_b, a->d_pos, o + o_trows, (uint32_t)nq, log_s);
    gl.paths(a->tnodes, log_N, a->d_pos, o + o_tpath, (uint32_t)nq);
    gl.rows(cldes, (uint32_t)CW, log_n, log_b, a->d_pos, o + o_crows, (uint32_t)nq);
    gl.paths(a->cnodes, log_N, a->d_pos, o + o_cpath, (uint32_t)nq);
    std::vector<size_t> o_lrows(n_layers), o_lpath(n_layers);
    {
        unsigned g2 = log_N;
        for (unsigned l = 0; l < n_layers; l++) {
            const unsigned np = (unsigned)lpos[l].size(), lr = g2 - log_f;
            o_lrows[l] = off; off += (size_t)np * fold * 8 * m;
            o_lpath[l] = off; off += (size_t)np * lr * 32;
            gl.rows(layer[l], fold * m, lr, 0, a->d_pos + 256 * (l + 1), o + o_lrows[l], np);
            gl.paths(a->lnodes[l], lr, a->d_pos + 256 * (l + 1), o + o_lpath[l], np);
            g2 -= log_f;
        }
    }
    HIP_TRY(gl.launch(st));
    if (a->h_open_bytes < off) {
        if (a->h_open) { HIP_TRY(hipHostFree(a->h_open)); a->h_open = nullptr; a->h_open_bytes = 0; }
        HIP_TRY(hipHostMalloc((void **)&a->h_open, off, hipHostMallocDefault));
        a->h_open_bytes = off;
    }
    const struct { const uint8_t *p; const uint8_t *data() const { return p; } } open{a->h_open};
    HIP_TRY(hipMemcpyAsync(a->h_open, o, off, hipMemcpyDeviceToHost, st));
    STAGE();
    HIP_TRY(cs::stream_wait(st));
    a->timed = true;
#undef STAGE

    Writer wr;
    wr.b.reserve(off + ((size_t)64 << 10) + 8 * ((size_t)opt->fri_max_remainder * (opt->field_extension + 1)));
    wr.raw("CSTK", 4); wr.u32(CSTARK_PROOF_VERSION);
    wr.u32((uint32_t)job.air); wr.u32((uint32_t)W); wr.u32(log_n); wr.u32(job.item);
    wr.u32(opt->num_queries); wr.u32(opt->blowup_factor); wr.u32(opt->grinding_factor); wr.u32(opt->hash_fn); wr.u32(opt->field_extension);
    wr.u32(opt->fri_folding_factor); wr.u32(opt->fri_max_remainder);
    wr.raw(trace_root, 32); wr.raw(cons_root, 32);
    wr.u32(n_layers); wr.raw(layer_roots.data(), layer_roots.size()); wr.raw(rem_commit, 32);
    wr.raw(ood_trace.data(), ood_trace.size() * 8); wr.raw(ood_comp.data(), ood_comp.size() * 8);
    wr.u64(nonce);
    wr.raw(open.data() + o_trows, nq * W * 8); wr.raw(open.data() + o_tpath, nq * log_N * 32);
    wr.raw(open.data() + o_crows, nq * CW * 8); wr.raw(open.data() + o_cpath, nq * log_N * 32);
    {
        unsigned g2 = log_N;
        for (unsigned l = 0; l < n_layers; l++) {
            const size_t np = lpos[l].size();
            wr.u32((uint32_t)np);
            wr.raw(open.data() + o_lrows[l], np * fold * 8 * m);
            wr.raw(open.data() + o_lpath[l], np * (g2 - log_f) * 32);
            g2 -= log_f;
        }
    }
    wr.u32((uint32_t)R); wr.raw(remainder.data(), remainder.size() * 8);
    *proof_len = wr.b.size();
    if (!proof || capacity < wr.b.size()) return fail(CSTARK_ERR_INVALID_ARG, "proof buffer too small (required size returned in *proof_len)");
    memcpy(proof, wr.b.data(), wr.b.size());
    return CSTARK_OK;
}

} // namespace
} // namespace cs

namespace {
// f(i) for every i < count on up to eight host threads.  Nothing escapes: an exception inside a worker (the channel code allocates) or a
// thread that cannot be created ends in CSTARK_ERR_OOM -- these lambdas run inside extern "C" functions, where an escaping exception
// would be std::terminate.
template <class F>
int parallel_for(size_t count, F f) {
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 1 : nt > 8 ? 8 : nt;
    std::atomic<bool> failed{false};
    auto guarded = [&](size_t first, size_t step) {
        try { for (size_t i = first; i < count && !failed.load(std::memory_order_relaxed); i += step) f(i); }
        catch (...) { failed.store(true); }
    };
    if (count < 64 || nt == 1) {
        guarded(0, 1);
    } else {
        std::vector<std::thread> th;
        try {
            th.reserve(nt);
            for (unsigned w = 0; w < nt; w++) th.emplace_back(guarded, (size_t)w, (size_t)nt);
        } catch (...) { failed.store(true); }
        for (std::thread &t : th) t.join();
    }
    return failed.load() ? cs::fail(CSTARK_ERR_OOM, "host allocation or thread creation failed inside a batched channel step") : CSTARK_OK;
}
struct Carver { // consecutive 256-byte aligned pieces of one block
    uint8_t *base; size_t off = 0;
    template <class T> T *take(size_t bytes) { T *q = (T *)(base + off); off += (bytes + 255) & ~(size_t)255; return q; }
};
} // namespace

using namespace cs;

extern "C" {

int cstark_tx_prove(cstark_ctx *c, const cstark_options *opt, uint8_t *proof, size_t capacity, size_t *proof_len) {
    if (!c || !opt || !proof_len) return fail(CSTARK_ERR_INVALID_ARG, "cstark_tx_prove: null argument");
    if (!c->wit_buf || c->wit.n_tx == 0 || c->wit.msg_tail) return fail(CSTARK_ERR_INVALID_ARG, "no transaction witness uploaded");
    if (c->wit.n_tx & (c->wit.n_tx - 1)) return fail(CSTARK_ERR_INVALID_ARG, "the number of transactions must be a power of two");
    AirJob job;
    job.air = CSTARK_AIR_STATE_TRANSITION; job.width = CSTARK_TX_TRACE_WIDTH; job.log_n = 10 + ceil_log2(c->wit.n_tx); job.log_ce = 3;
    job.n_constraints = CSTARK_TX_NUM_CONSTRAINTS; job.n_assertions = 4; job.item = c->wit.depth;
    job.build = tx_build; job.combine = tx_combine; job.combine_sets = tx_combine_sets;
    if (opt->field_extension == 1 || opt->field_extension == 2) return prove_ext(c, opt, job, proof, capacity, proof_len);
    if (use_dev_channel(opt, job)) return prove_core_dev(c, opt, job, proof, capacity, proof_len);
    return prove_core(c, opt, job, proof, capacity, proof_len);
}

int cstark_air_prove(cstark_ctx *c, int air, const cstark_options *opt, uint64_t number, uint8_t *proof, size_t capacity, size_t *proof_len) {
    if (!c || !opt || !proof_len) return fail(CSTARK_ERR_INVALID_ARG, "cstark_air_prove: null argument");
    if (air == CSTARK_AIR_STATE_TRANSITION) return cstark_tx_prove(c, opt, proof, capacity, proof_len);
    AirJob job;
    job.air = air;
    host::AirShape s;
    if (air == CSTARK_AIR_MERKLE_UPDATE) {
        if (!c->wit_buf || c->wit.n_tx == 0 || c->wit.msg_tail) return fail(CSTARK_ERR_INVALID_ARG, "no transaction witness uploaded");
        if (c->wit.n_tx & (c->wit.n_tx - 1)) return fail(CSTARK_ERR_INVALID_ARG, "the number of transactions must be a power of two");
        host::air_shape(air, s, 0);
        job.log_n = 9 + ceil_log2(c->wit.n_tx); job.item = c->wit.depth;
        job.build = merkle_build; job.combine = merkle_combine;
    } else if (air == CSTARK_AIR_RANGE) {
        if (number >= host::P || (host::to_u64(number) >> 63)) return fail(CSTARK_ERR_INVALID_ARG, "range proofs cover 63-bit field elements (src/range/tests.rs:54-62)");
        // One 64-row proof is host-API bound either way.  The batch prover with a batch of one makes the same bytes from ~25 launches
        // instead of ~150 (CSTARK_RANGE_VIA_BATCH=1): 0.30 against 0.36 ms in a process that does nothing else, but 0.48 against 0.38 ms
        // inside bench.py's process (other contexts alive) -- so the generic path stays the default (profiles/r03_range_single.txt).
        static const bool via_batch = [] { const char *e = getenv("CSTARK_RANGE_VIA_BATCH"); return e && atoi(e) != 0; }();
        if (via_batch && opt->field_extension == 0 && opt->blowup_factor == 8 && opt->fri_folding_factor == 4 && proof && capacity >= cstark_tx_proof_size_bound(1, opt))
            return cstark_range_prove_batch(c, opt, &number, 1, proof, capacity, proof_len);
        host::air_shape(air, s, 0);
        job.log_n = 6; job.item = 0; job.number = number; // RANGE_LOG = 64 rows, src/range/mod.rs:34
        job.pub = {number};
        job.build = range_build; job.combine = range_combine;
    } else if (air == CSTARK_AIR_SCHNORR) {
        if (!c->wit_buf || c->wit.n_tx == 0 || !c->wit.msg_tail) return fail(CSTARK_ERR_INVALID_ARG, "no Schnorr witness uploaded");
        const uint32_t ns = c->wit.n_tx;
        if (ns & (ns - 1)) return fail(CSTARK_ERR_INVALID_ARG, "the number of signatures must be a power of two");
        host::air_shape(air, s, ns);
        job.log_n = 9 + ceil_log2(ns); job.item = ns;
        job.pub = c->schnorr_pub; // messages [ns][28] then R.x [ns][6] (src/schnorr/air.rs:29-38)
        job.pub_bytes = c->schnorr_s;
        job.build = schnorr_build; job.combine = schnorr_combine;
    } else {
        return fail(CSTARK_ERR_UNSUPPORTED, "no prover for this AIR");
    }
    job.width = s.width; job.n_constraints = s.n_constraints; job.n_assertions = (uint32_t)s.a_reg.size(); job.log_ce = s.log_ce_blowup();
    if (opt->field_extension == 1 || opt->field_extension == 2) return prove_ext(c, opt, job, proof, capacity, proof_len);
    if (use_dev_channel(opt, job)) return prove_core_dev(c, opt, job, proof, capacity, proof_len);
    return prove_core(c, opt, job, proof, capacity, proof_len);
}

// ---- one proof across several GPUs by LDE coset -------------------------------------------------------------------------
// Every rank holds the witness and calls the phases in the same order; the caller moves the three exchanged buffers between the
// ranks (RCCL all-gather / all-reduce through torch.distributed in sharding.py).  Trace generation and interpolation are replicated
// (every rank needs all coefficient columns for its cosets), extension / row hashing / constraint evaluation run on the rank's cosets,
// everything after the merged evaluations on the rank that owns coset 0.
static int shard_run(cstark_ctx *c, int min_phase, ProofRun **out) {
    if (!c || !c->arena || !c->arena->run || c->arena->run->phase < min_phase) return fail(CSTARK_ERR_INVALID_ARG, "sharded proof: phase called out of order");
    *out = c->arena->run;
    return CSTARK_OK;
}
int cstark_tx_shard_commit(cstark_ctx *c, const cstark_options *opt, uint32_t k0, uint32_t nk, uint8_t *d_leaves_local) {
    if (!c || !opt || !d_leaves_local) return fail(CSTARK_ERR_INVALID_ARG, "cstark_tx_shard_commit: null argument");
    if (!c->wit_buf || c->wit.n_tx == 0 || c->wit.msg_tail) return fail(CSTARK_ERR_INVALID_ARG, "no transaction witness uploaded");
    if (c->wit.n_tx & (c->wit.n_tx - 1)) return fail(CSTARK_ERR_INVALID_ARG, "the number of transactions must be a power of two");
    if (nk == 0 || nk >= 8 || (nk & (nk - 1)) || k0 % nk || k0 + nk > 8) return fail(CSTARK_ERR_INVALID_ARG, "a rank owns 1, 2 or 4 consecutive cosets of the 8 (world size 8, 4 or 2)");
    if (opt->field_extension != 0) return fail(CSTARK_ERR_UNSUPPORTED, "sharded proofs use FieldExtension::None");
    AirJob job;
    job.air = CSTARK_AIR_STATE_TRANSITION; job.width = CSTARK_TX_TRACE_WIDTH; job.log_n = 10 + ceil_log2(c->wit.n_tx); job.log_ce = 3;
    job.n_constraints = CSTARK_TX_NUM_CONSTRAINTS; job.n_assertions = 4; job.item = c->wit.depth;
    job.build = tx_build; job.combine = tx_combine;
    job.k0 = k0; job.nk = nk; job.sharded = true;
    ProofRun *R = new (std::nothrow) ProofRun();
    if (!R) return fail(CSTARK_ERR_OOM, "host allocation failed");
    ProveArena *a = nullptr;
    int rc = run_setup(c, opt, job, *R, &a); // ends any earlier run on this context (get_arena)
    if (rc) { delete R; return rc; }
    a->run = R;
    rc = phase_commit(c, a, *R, d_leaves_local);
    if (rc) { proof_run_free(a->run); a->run = nullptr; }
    return rc;
}
uint32_t cstark_tx_shard_rows(uint32_t nk) { return (nk == 1 || nk == 2 || nk == 4) ? shard_rows(nk) : 0; }
uint32_t cstark_tx_shard_open_words(uint32_t nk) { return (nk == 1 || nk == 2 || nk == 4) ? CSTARK_TX_TRACE_WIDTH + 4 * ceil_log2(nk) : 0; }
int cstark_tx_shard_evaluate(cstark_ctx *c, const uint8_t *d_leaves_all, uint64_t *d_combined_local, uint32_t rows) {
    ProofRun *R;
    RC_TRY(shard_run(c, 1, &R));
    if (!d_leaves_all || !d_combined_local) return fail(CSTARK_ERR_INVALID_ARG, "cstark_tx_shard_evaluate: null argument");
    // the caller sized d_combined_local [rows][n]: the count depends on nk AND on the evaluation mode of this process (shard_rows)
    if (rows != shard_rows(R->job.nk)) return fail(CSTARK_ERR_INVALID_ARG, "cstark_tx_shard_evaluate: rows differs from cstark_tx_shard_rows(nk)");
    return phase_evaluate(c, c->arena, *R, d_leaves_all, d_combined_local);
}
int cstark_tx_shard_compose(cstark_ctx *c, const uint64_t *d_combined_all, uint32_t total_rows, uint32_t *positions /* host [num_queries] */) {
    ProofRun *R;
    RC_TRY(shard_run(c, 2, &R));
    if (!d_combined_all || !positions) return fail(CSTARK_ERR_INVALID_ARG, "cstark_tx_shard_compose: null argument");
    // every rank must have handed over the same number of rows (ranks whose CSTARK_SHARD_SPLIT differs would not)
    if (total_rows != (8 / R->job.nk) * shard_rows(R->job.nk)) return fail(CSTARK_ERR_INVALID_ARG, "cstark_tx_shard_compose: total_rows differs from W * cstark_tx_shard_rows(nk)");
    ProveArena *a = c->arena;
    const size_t N = (size_t)8 << R->job.log_n;
    if (shard_split(R->job.nk)) RC_TRY(tx_shard_combine(c, d_combined_all, a->combined, R->job.log_n, R->job.nk)); // the ranks' shares -> [8][n]
    else if (d_combined_all != a->combined) HIP_TRY(hipMemcpyAsync(a->combined, d_combined_all, N * 8, hipMemcpyDeviceToDevice, c->stream));
    RC_TRY(phase_compose(c, a, *R));
    memcpy(positions, R->positions.data(), R->positions.size() * 4);
    return CSTARK_OK;
}
int cstark_tx_shard_open_rows(cstark_ctx *c, const uint32_t *positions, uint32_t nq, uint64_t *d_rows) {
    ProofRun *R;
    RC_TRY(shard_run(c, 2, &R));
    if (!positions || !d_rows) return fail(CSTARK_ERR_INVALID_ARG, "cstark_tx_shard_open_rows: null argument");
    // d_rows is sized by the caller's nq and read back as num_queries rows by cstark_tx_shard_finish: they must agree
    if (nq != R->opt.num_queries) return fail(CSTARK_ERR_INVALID_ARG, "cstark_tx_shard_open_rows: nq differs from the proof's num_queries");
    const uint32_t N = 8u << R->job.log_n;
    for (uint32_t q = 0; q < nq; q++)
        if (positions[q] >= N) return fail(CSTARK_ERR_INVALID_ARG, "cstark_tx_shard_open_rows: position outside the LDE domain");
    ProveArena *a = c->arena;
    HIP_TRY(hipMemcpyAsync(a->d_pos, positions, nq * 4, hipMemcpyHostToDevice, c->stream));
    const uint32_t log_nk = ceil_log2(R->job.nk);
    const uint64_t *sub = log_nk ? (const uint64_t *)a->extra[45] : nullptr; // the rank's subtree heap (phase_commit)
    k_gather_rows_window<<<nq, 128, 0, c->stream>>>(a->lde, R->job.width, R->job.log_n, 3, R->job.k0, R->job.nk, a->d_pos, d_rows, sub, log_nk);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream)); // the caller's positions may be transient
    return CSTARK_OK;
}
int cstark_tx_shard_finish(cstark_ctx *c, const uint64_t *d_rows, uint8_t *proof, size_t capacity, size_t *proof_len) {
    ProofRun *R;
    RC_TRY(shard_run(c, 3, &R));
    if (!d_rows || !proof_len) return fail(CSTARK_ERR_INVALID_ARG, "cstark_tx_shard_finish: null argument");
    if (R->job.k0 != 0) return fail(CSTARK_ERR_INVALID_ARG, "cstark_tx_shard_finish runs on the rank that owns coset 0");
    const int rc = phase_open(c, c->arena, *R, d_rows, proof, capacity, proof_len);
    if (rc == CSTARK_OK) { proof_run_free(c->arena->run); c->arena->run = nullptr; }
    return rc;
}

// RescueExample::prove (benches/rescue.rs:66-86): a chain of `chain_length` Rescue hashes from `seed` (7 elements, memory form)
int cstark_rescue_prove(cstark_ctx *c, const cstark_options *opt, const uint64_t seed[7], uint32_t chain_length, uint8_t *proof, size_t capacity,
                        size_t *proof_len) {
    if (!c || !opt || !seed || !proof_len) return fail(CSTARK_ERR_INVALID_ARG, "cstark_rescue_prove: null argument");
    if (chain_length < 8 || (chain_length & (chain_length - 1)) || chain_length > (1u << 21))
        return fail(CSTARK_ERR_INVALID_ARG, "chain length must be a power of two, 8 .. 2^21 (benches/rescue.rs:34-37)");
    for (int i = 0; i < 7; i++) if (seed[i] >= host::P) return fail(CSTARK_ERR_INVALID_ARG, "seed is not a field element");
    AirJob job;
    job.air = CSTARK_AIR_RESCUE_CHAIN;
    host::AirShape s;
    host::air_shape(CSTARK_AIR_RESCUE_CHAIN, s, 0);
    job.log_n = 3 + ceil_log2(chain_length); job.item = chain_length;
    memcpy(job.seed, seed, sizeof job.seed);
    job.build = rescue_build; job.combine = rescue_combine;
    job.width = s.width; job.n_constraints = s.n_constraints; job.n_assertions = (uint32_t)s.a_reg.size(); job.log_ce = s.log_ce_blowup();
    if (opt->field_extension == 1 || opt->field_extension == 2) return prove_ext(c, opt, job, proof, capacity, proof_len);
    if (use_dev_channel(opt, job)) return prove_core_dev(c, opt, job, proof, capacity, proof_len);
    return prove_core(c, opt, job, proof, capacity, proof_len);
}

// RangeProofAir over 2^log_n rows (synthetic long form; log_n = 6 with a one-word value is cstark_air_prove(CSTARK_AIR_RANGE))
int cstark_range_prove_bits(cstark_ctx *c, const cstark_options *opt, const uint64_t *words, uint32_t log_n, uint8_t *proof, size_t capacity,
                            size_t *proof_len) {
    if (!c || !opt || !words || !proof_len) return fail(CSTARK_ERR_INVALID_ARG, "cstark_range_prove_bits: null argument");
    if (log_n < 6 || log_n > 21) return fail(CSTARK_ERR_INVALID_ARG, "trace length must be 2^6 .. 2^21");
    const size_t nw = (size_t)1 << (log_n - 6);
    if (words[nw - 1] >> 63) return fail(CSTARK_ERR_INVALID_ARG, "the value must have at most n - 1 bits");
    uint64_t number = 0; // V mod p, memory form: Horner in base 2^64
    for (size_t i = nw; i-- > 0;) number = host::add(host::mul(number, host::R2), host::from_u64(words[i] % host::P));
    AirJob job;
    job.air = CSTARK_AIR_RANGE;
    host::AirShape s;
    host::air_shape(CSTARK_AIR_RANGE, s, 0);
    job.log_n = log_n; job.item = 0; job.number = number; job.bits = words;
    job.pub = {number};
    job.build = range_build; job.combine = range_combine;
    job.width = s.width; job.n_constraints = s.n_constraints; job.n_assertions = (uint32_t)s.a_reg.size(); job.log_ce = s.log_ce_blowup();
    if (opt->field_extension == 1 || opt->field_extension == 2) return prove_ext(c, opt, job, proof, capacity, proof_len);
    if (use_dev_channel(opt, job)) return prove_core_dev(c, opt, job, proof, capacity, proof_len);
    return prove_core(c, opt, job, proof, capacity, proof_len);
}

// ---- B reference-shaped range proofs in one call (RangeProofExample::prove, src/range/mod.rs:75-100, benches/range.rs:15-37) ---------
// Every stage is one launch over the batch (range_batch.hip; interpolation and extension of the 2 B columns through the generic
// transform kernels), the host walks the B Fiat-Shamir channels between the stages on a few threads.  Same protocol, same bytes as
// prove_core for CSTARK_AIR_RANGE: the channel order, the proof layout and every formula are the ones documented there.
int cstark_range_prove_batch(cstark_ctx *c, const cstark_options *opt, const uint64_t *numbers, uint32_t count, uint8_t *proofs, size_t stride, size_t *lens) {
    using namespace cs::host;
    if (!c || !opt || !numbers || !proofs || !lens || count == 0) return fail(CSTARK_ERR_INVALID_ARG, "cstark_range_prove_batch: null argument");
    // the transforms of the 2 * count columns are one launch with the column index in grid.y (at most 65535): 32768 proofs per call
    if (count > 32768) return fail(CSTARK_ERR_INVALID_ARG, "cstark_range_prove_batch: at most 32768 proofs per call");
    if (opt->field_extension != 0) return fail(CSTARK_ERR_UNSUPPORTED, "cstark_range_prove_batch: FieldExtension::None only (use cstark_air_prove)");
    unsigned log_rem = 0, log_b_opt = 3, log_f_opt = 2;
    RC_TRY(check_options(opt, 1, &log_rem, &log_b_opt, &log_f_opt));
    if (log_b_opt != 3 || log_f_opt != 2) {
        // the one-launch-per-stage kernels (range_batch.hip) are laid out for the reference's get_example options (blowup 8, folding 4,
        // src/range/mod.rs:44-50); other option values go through the generic prover one proof at a time -- same bytes
        for (uint32_t t = 0; t < count; t++)
            RC_TRY(cstark_air_prove(c, CSTARK_AIR_RANGE, opt, numbers[t], proofs + stride * t, stride, &lens[t]));
        return CSTARK_OK;
    }
    const unsigned log_n = RB_LOG_N, log_N = log_n + 3, n_layers = num_fri_layers(log_N, log_rem, 2);
    const size_t B = count, n = RB_N, N = RB_LDE, nq = opt->num_queries, W = 2, ce = RB_CE;
    if (nq > N / 4) return fail(CSTARK_ERR_INVALID_ARG, "more queries than the domain supports");
    std::vector<uint64_t> canon(B);
    for (size_t t = 0; t < B; t++) {
        if (numbers[t] >= P) return fail(CSTARK_ERR_INVALID_ARG, "number is not a field element");
        canon[t] = to_u64(numbers[t]);
        if (canon[t] >> 63) return fail(CSTARK_ERR_INVALID_ARG, "range proofs cover 63-bit field elements (src/range/tests.rs:54-62)");
    }
    const size_t rem_len = n_layers ? N / 4 : N, slot = nq * 864;
    const size_t per_proof = 4096 + 32 * 3 + 8 * (2 * W + ce) + nq * (W * 8 + ce * 8 + 2 * log_N * 32) + n_layers * (4 + nq * (32 + 7 * 32)) + 4 + 8 * rem_len;
    if (stride < per_proof) return fail(CSTARK_ERR_INVALID_ARG, "cstark_range_prove_batch: stride too small (cstark_tx_proof_size_bound(1, opt) is sufficient)");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const uint32_t hf = opt->hash_fn;
    if (c->arena) c->arena->timed = false; // cstark_prove_stage_ms describes the generic prover's last proof: none after a batch

    // ---- buffers: one device block, one pinned host block ----------------------------------------------------------------------------
    const size_t dev_need = B * (16 + 1024 * 2 + 8192 + 32768 + 64 + 1024 + 1024 + 8192 + 32768 + 8 + 48 + 64 + 4096 + 8192 + 8 + 8 * rem_len + 8 * nq + 4 + slot + 40) + 66 * 256;
    const size_t host_need = B * (32 * 3 + 64 + 48 + 64 + 8 + 8 + 8 * rem_len + 8 * nq + 4 + slot + 40) + 34 * 256;
    if (c->rb_dev_bytes < dev_need) {
        HIP_TRY(hipStreamSynchronize(st));
        if (c->rb_dev) { HIP_TRY(hipFree(c->rb_dev)); c->rb_dev = nullptr; c->rb_dev_bytes = 0; }
        HIP_TRY(hipMalloc(&c->rb_dev, dev_need));
        c->rb_dev_bytes = dev_need;
    }
    if (c->rb_host_bytes < host_need) {
        HIP_TRY(hipStreamSynchronize(st));
        if (c->rb_host) { HIP_TRY(hipHostFree(c->rb_host)); c->rb_host = nullptr; c->rb_host_bytes = 0; }
        HIP_TRY(hipHostMalloc(&c->rb_host, host_need, hipHostMallocDefault));
        c->rb_host_bytes = host_need;
    }
    Carver D{(uint8_t *)c->rb_dev}, H{(uint8_t *)c->rb_host};
    uint64_t *d_canon = D.take<uint64_t>(B * 8), *d_num = D.take<uint64_t>(B * 8), *d_trace = D.take<uint64_t>(B * 1024), *d_coeffs = D.take<uint64_t>(B * 1024);
    uint64_t *d_lde = D.take<uint64_t>(B * 8192), *d_coefs = D.take<uint64_t>(B * 64), *d_comb = D.take<uint64_t>(B * 1024), *d_ccoef = D.take<uint64_t>(B * 1024);
    uint64_t *d_clde = D.take<uint64_t>(B * 8192), *d_z = D.take<uint64_t>(B * 8), *d_ood = D.take<uint64_t>(B * 48), *d_dcoef = D.take<uint64_t>(B * 64);
    uint64_t *d_layer = D.take<uint64_t>(B * 4096), *d_alpha = D.take<uint64_t>(B * 8), *d_rem = D.take<uint64_t>(B * 8 * rem_len);
    uint8_t *d_tnodes = D.take<uint8_t>(B * 32768), *d_cnodes = D.take<uint8_t>(B * 32768), *d_lnodes = D.take<uint8_t>(B * 8192);
    uint32_t *d_pos = D.take<uint32_t>(B * nq * 4), *d_lpos = D.take<uint32_t>(B * nq * 4), *d_lcount = D.take<uint32_t>(B * 4);
    uint8_t *d_open = D.take<uint8_t>(B * slot);
    uint32_t *d_gseed = D.take<uint32_t>(B * 32);
    unsigned long long *d_gfound = D.take<unsigned long long>(B * 8);
    uint8_t *h_troot = H.take<uint8_t>(B * 32), *h_croot = H.take<uint8_t>(B * 32), *h_lroot = H.take<uint8_t>(B * 32);
    uint64_t *h_coefs = H.take<uint64_t>(B * 64), *h_ood = H.take<uint64_t>(B * 48), *h_dcoef = H.take<uint64_t>(B * 64), *h_z = H.take<uint64_t>(B * 8);
    uint64_t *h_alpha = H.take<uint64_t>(B * 8), *h_rem = H.take<uint64_t>(B * 8 * rem_len);
    uint32_t *h_pos = H.take<uint32_t>(B * nq * 4), *h_lpos = H.take<uint32_t>(B * nq * 4), *h_lcount = H.take<uint32_t>(B * 4);
    uint8_t *h_open = H.take<uint8_t>(B * slot);
    uint32_t *h_gseed = H.take<uint32_t>(B * 32);
    unsigned long long *h_gfound = H.take<unsigned long long>(B * 8);

    RangeBatchConsts K{};
    {
        const uint64_t wN = root_of_unity(log_N), g = lde_offset();
        uint64_t sh = g;
        for (int k = 0; k < 8; k++) { K.shift[k] = sh; K.zinv[k] = inv(sub(pow(sh, n), ONE)); sh = mul(sh, wN); }
        K.w_last = inv(root_of_unity(log_n));
        const uint64_t cen = n * ce;
        K.adj[0] = CSTARK_CONV_TRANSITION_ADJUSTMENT(cen, n, 2 * (n - 1)); // degrees (2), (1): src/range/air.rs:100-105
        K.adj[1] = CSTARK_CONV_TRANSITION_ADJUSTMENT(cen, n, 1 * (n - 1));
        K.badj = CSTARK_CONV_BOUNDARY_ADJUSTMENT(cen, n, 1);
        K.inv128 = inv(from_u64(cen)); K.ginv = inv(g); K.offset_inv = inv(g); K.inv4 = inv(from_u64(4));
        const uint64_t *unused;
        RC_TRY(plan_tables(c, log_n, &K.w64, &unused));
        RC_TRY(plan_tables(c, log_n + 1, &unused, &K.winv128));
        RC_TRY(plan_tables(c, log_N, &unused, &K.winv512));
    }
    auto hash_rows_b = [&](const uint64_t *tab, uint8_t *leaves, unsigned gw, unsigned ln, unsigned lb, size_t leaf_stride) {
        return hf == 1 ? hash_rows_batch_sha3(tab, leaves, gw, (unsigned)(gw * B), ln, lb, (unsigned)B, leaf_stride, st)
                       : hash_rows_batch(tab, leaves, gw, (unsigned)(gw * B), ln, lb, (unsigned)B, leaf_stride, st);
    };
    auto merkle_b = [&](uint8_t *nodes, unsigned log_leaves, size_t node_stride) {
        return hf == 1 ? merkle_build_batch_sha3(nodes, log_leaves, (unsigned)B, node_stride, st) : merkle_build_batch(nodes, log_leaves, (unsigned)B, node_stride, st);
    };

    static const bool rb_prof = getenv("CSTARK_RB_PROF") != nullptr; // debugging: host wall-clock of the phases on stderr
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!rb_prof) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[cstark range batch] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(now - t_prev).count());
        t_prev = now;
    };
    mark("setup");
    // ---- trace, extension, commitment ----------------------------------------------------------------------------------------------------
    HIP_TRY(hipMemcpyAsync(d_canon, canon.data(), B * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_num, numbers, B * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(rb_trace(d_canon, d_trace, (unsigned)B, st));
    RC_TRY(cstark_interpolate_columns(c, d_trace, d_coeffs, (uint32_t)(2 * B), log_n));
    RC_TRY(cstark_lde_columns(c, d_coeffs, d_lde, (uint32_t)(2 * B), log_n, 3, lde_offset(), 0, 8));
    HIP_TRY(hash_rows_b(d_lde, d_tnodes + 32 * N, 2, log_n, 3, 2 * N * 32));
    HIP_TRY(merkle_b(d_tnodes, log_N, 2 * N * 32));
    HIP_TRY(hipMemcpy2DAsync(h_troot, 32, d_tnodes + 32, 2 * N * 32, 32, B, hipMemcpyDeviceToHost, st));
    HIP_TRY(cs::stream_wait(st)); // (also: `canon` and the caller's numbers have been read)

    mark("trace..trace roots (gpu)");
    std::vector<Coin> coins(B);
    RC_TRY(parallel_for(B, [&](size_t t) {
        Coin &coin = coins[t];
        coin.hash_fn = hf;
        Writer sd;
        const uint8_t ctxb[2] = {(uint8_t)W, (uint8_t)log_n};
        sd.raw(ctxb, 2);
        sd.u64(P);
        const uint8_t ob[7] = {(uint8_t)opt->num_queries, 3, (uint8_t)opt->grinding_factor, (uint8_t)opt->hash_fn, (uint8_t)opt->field_extension,
                               (uint8_t)opt->fri_folding_factor, (uint8_t)log_rem};
        sd.raw(ob, 7);
        sd.u64(canon[t]); // PublicInputs: the number (src/range/air.rs:26-36)
        coin.init(sd.b.data(), sd.b.size());
        coin.reseed(h_troot + 32 * t);
        uint64_t *cf = h_coefs + 8 * t; // t_alpha[2] t_beta[2] b_alpha[2] b_beta[2]
        for (int i = 0; i < 2; i++) { cf[i] = coin.draw(); cf[2 + i] = coin.draw(); }
        for (int i = 0; i < 2; i++) { cf[4 + i] = coin.draw(); cf[6 + i] = coin.draw(); }
    }));
    mark("coefficients (host)");
    HIP_TRY(hipMemcpyAsync(d_coefs, h_coefs, B * 64, hipMemcpyHostToDevice, st));

    // ---- constraint evaluation, composition polynomial and its commitment -------------------------------------------------------------
    HIP_TRY(rb_combine(K, d_lde, d_coefs, d_num, d_comb, (unsigned)B, st));
    HIP_TRY(rb_composition(K, d_comb, d_ccoef, (unsigned)B, st));
    RC_TRY(cstark_lde_columns(c, d_ccoef, d_clde, (uint32_t)(2 * B), log_n, 3, lde_offset(), 0, 8));
    HIP_TRY(hash_rows_b(d_clde, d_cnodes + 32 * N, 2, log_n, 3, 2 * N * 32));
    HIP_TRY(merkle_b(d_cnodes, log_N, 2 * N * 32));
    HIP_TRY(hipMemcpy2DAsync(h_croot, 32, d_cnodes + 32, 2 * N * 32, 32, B, hipMemcpyDeviceToHost, st));
    HIP_TRY(cs::stream_wait(st));
    mark("constraints..comp roots (gpu)");
    RC_TRY(parallel_for(B, [&](size_t t) { coins[t].reseed(h_croot + 32 * t); h_z[t] = coins[t].draw(); }));
    HIP_TRY(hipMemcpyAsync(d_z, h_z, B * 8, hipMemcpyHostToDevice, st));

    // ---- out-of-domain frame, DEEP composition -----------------------------------------------------------------------------------------
    HIP_TRY(rb_ood(K, d_coeffs, d_ccoef, d_z, d_ood, (unsigned)B, st));
    HIP_TRY(hipMemcpyAsync(h_ood, d_ood, B * 48, hipMemcpyDeviceToHost, st));
    HIP_TRY(cs::stream_wait(st));
    RC_TRY(parallel_for(B, [&](size_t t) {
        Coin &coin = coins[t];
        uint8_t dg[32];
        hash_elements(hf, h_ood + 6 * t, 4, dg); coin.reseed(dg);
        hash_elements(hf, h_ood + 6 * t + 4, 2, dg); coin.reseed(dg);
        uint64_t *cf = h_dcoef + 8 * t; // alpha[2] beta[2] delta[2] deg_a deg_b
        for (int i = 0; i < 2; i++) {
            cf[i] = coin.draw(); cf[2 + i] = coin.draw();
            for (int k = 2; k < CSTARK_CONV_DEEP_DRAWS_PER_REGISTER; k++) (void)coin.draw();
        }
        for (int i = 0; i < 2; i++) cf[4 + i] = coin.draw();
        cf[6] = coin.draw(); cf[7] = coin.draw();
    }));
    mark("ood + deep coefficients");
    HIP_TRY(hipMemcpyAsync(d_dcoef, h_dcoef, B * 64, hipMemcpyHostToDevice, st));
    HIP_TRY(rb_deep(K, d_lde, d_clde, d_z, d_ood, d_dcoef, d_layer, (unsigned)B, st));

    // ---- FRI: at most one layer for a 512-point domain (remainder 128 .. 1024) ----------------------------------------------------------
    if (n_layers) {
        HIP_TRY(hash_rows_b(d_layer, d_lnodes + 32 * 128, 4, 7, 0, 256 * 32)); // rows { e[i + t 128] }: table t = [4][128] inside its 512 words
        HIP_TRY(merkle_b(d_lnodes, 7, 256 * 32));
        HIP_TRY(hipMemcpy2DAsync(h_lroot, 32, d_lnodes + 32, 256 * 32, 32, B, hipMemcpyDeviceToHost, st));
        HIP_TRY(cs::stream_wait(st));
        RC_TRY(parallel_for(B, [&](size_t t) { coins[t].reseed(h_lroot + 32 * t); h_alpha[t] = coins[t].draw(); }));
        HIP_TRY(hipMemcpyAsync(d_alpha, h_alpha, B * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(rb_fold(K, d_layer, d_alpha, d_rem, (unsigned)B, st));
        HIP_TRY(hipMemcpyAsync(h_rem, d_rem, B * 8 * rem_len, hipMemcpyDeviceToHost, st));
    } else {
        HIP_TRY(hipMemcpyAsync(h_rem, d_layer, B * 8 * rem_len, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(cs::stream_wait(st));
    mark("deep + fri (gpu + host)");
    std::vector<uint64_t> nonces(B);
    std::vector<uint8_t> rem_commit(32 * B);
    // proof of work: from 12 bits on (Blake3 coin) all B searches run on the device, chunk after chunk in increasing order, until every
    // proof has its smallest nonce (grind_nonce above: the single-proof form)
    static const bool grind_dev_env = [] { const char *e = getenv("CSTARK_GRIND_DEVICE"); return !e || atoi(e) != 0; }();
    const bool grind_dev = grind_dev_env && opt->grinding_factor >= 12;
    if (grind_dev) {
        RC_TRY(parallel_for(B, [&](size_t t) {
            hash_elements(hf, h_rem + rem_len * t, rem_len, &rem_commit[32 * t]);
            coins[t].reseed(&rem_commit[32 * t]);
            memcpy(h_gseed + 8 * t, coins[t].seed, 32);
            h_gfound[t] = ~0ull;
        }));
        HIP_TRY(hipMemcpyAsync(d_gseed, h_gseed, B * 32, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_gfound, h_gfound, B * 8, hipMemcpyHostToDevice, st));
        uint64_t chunk = (uint64_t)4 << opt->grinding_factor; // four expected hits per proof and chunk
        while (chunk * B > ((uint64_t)1 << 28)) chunk >>= 1;   // at most 2^28 nonces per launch
        if (chunk < 256) chunk = 256;
        for (uint64_t base = 1;; base += chunk) {
            if (hf == 1) HIP_TRY(cs::grind_batch_chunk_sha3((const uint64_t *)d_gseed, (unsigned)B, base, chunk, opt->grinding_factor, d_gfound, st));
            else HIP_TRY(cs::grind_batch_chunk(d_gseed, (unsigned)B, base, chunk, opt->grinding_factor, d_gfound, st));
            HIP_TRY(hipMemcpyAsync(h_gfound, d_gfound, B * 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(cs::stream_wait(st));
            bool all = true;
            for (size_t t = 0; t < B; t++) all = all && h_gfound[t] != ~0ull;
            if (all) break;
            if (base > ((uint64_t)1 << 44)) return fail(CSTARK_ERR_HIP, "proof of work: no nonce found");
        }
    }
    RC_TRY(parallel_for(B, [&](size_t t) {
        Coin &coin = coins[t];
        uint64_t nonce = 1;
        if (grind_dev) {
            nonce = h_gfound[t];
        } else {
            hash_elements(hf, h_rem + rem_len * t, rem_len, &rem_commit[32 * t]);
            coin.reseed(&rem_commit[32 * t]);
            for (;; nonce++) {
                uint8_t out[32];
                coin.with_int(coin.seed, nonce, out);
                uint64_t v = 0;
                for (int i = 0; i < 8; i++) v |= (uint64_t)out[i] << (8 * i);
                if (opt->grinding_factor == 0 || (v & ((1ull << opt->grinding_factor) - 1)) == 0) break;
            }
        }
        nonces[t] = nonce;
        coin.reseed_int(nonce);
        std::vector<uint32_t> pos;
        coin.draw_integers(nq, N, pos);
        memcpy(h_pos + nq * t, pos.data(), nq * 4);
        if (n_layers) {
            const std::vector<uint32_t> lp = fold_positions(pos, 128);
            h_lcount[t] = (uint32_t)lp.size();
            memcpy(h_lpos + nq * t, lp.data(), lp.size() * 4);
            for (size_t q = lp.size(); q < nq; q++) h_lpos[nq * t + q] = 0;
        } else h_lcount[t] = 0;
    }));

    mark("remainder, positions (host)");
    // ---- openings, proof bytes ----------------------------------------------------------------------------------------------------------------
    HIP_TRY(hipMemcpyAsync(d_pos, h_pos, B * nq * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_lpos, h_lpos, B * nq * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_lcount, h_lcount, B * 4, hipMemcpyHostToDevice, st));
    RangeBatchOpen o{d_lde, d_clde, d_layer, d_tnodes, d_cnodes, d_lnodes, d_pos, d_lpos, d_lcount, d_open, (uint32_t)nq, (uint32_t)B, n_layers, slot};
    HIP_TRY(rb_open(o, st));
    HIP_TRY(hipMemcpyAsync(h_open, d_open, B * slot, hipMemcpyDeviceToHost, st));
    HIP_TRY(cs::stream_wait(st));
    RC_TRY(parallel_for(B, [&](size_t t) {
        const uint8_t *op = h_open + slot * t;
        const size_t o_trows = 0, o_tpath = o_trows + nq * 16, o_crows = o_tpath + nq * 288, o_cpath = o_crows + nq * 16, o_lrows = o_cpath + nq * 288,
                     o_lpath = o_lrows + nq * 32;
        Writer wr;
        wr.b.reserve(per_proof);
        wr.raw("CSTK", 4); wr.u32(CSTARK_PROOF_VERSION);
        wr.u32((uint32_t)CSTARK_AIR_RANGE); wr.u32((uint32_t)W); wr.u32(log_n); wr.u32(0);
        wr.u32(opt->num_queries); wr.u32(opt->blowup_factor); wr.u32(opt->grinding_factor); wr.u32(opt->hash_fn); wr.u32(opt->field_extension);
        wr.u32(opt->fri_folding_factor); wr.u32(opt->fri_max_remainder);
        wr.raw(h_troot + 32 * t, 32); wr.raw(h_croot + 32 * t, 32);
        wr.u32(n_layers);
        if (n_layers) wr.raw(h_lroot + 32 * t, 32);
        wr.raw(&rem_commit[32 * t], 32);
        wr.raw(h_ood + 6 * t, 6 * 8);
        wr.u64(nonces[t]);
        wr.raw(op + o_trows, nq * 16); wr.raw(op + o_tpath, nq * log_N * 32);
        wr.raw(op + o_crows, nq * 16); wr.raw(op + o_cpath, nq * log_N * 32);
        if (n_layers) {
            const size_t np = h_lcount[t];
            wr.u32((uint32_t)np);
            wr.raw(op + o_lrows, np * 32);
            wr.raw(op + o_lpath, np * 7 * 32);
        }
        wr.u32((uint32_t)rem_len); wr.raw(h_rem + rem_len * t, rem_len * 8);
        lens[t] = wr.b.size();
        if (wr.b.size() <= stride) memcpy(proofs + stride * t, wr.b.data(), wr.b.size());
    }));
    mark("openings + serialise");
    for (size_t t = 0; t < B; t++)
        if (lens[t] > stride) return fail(CSTARK_ERR_INVALID_ARG, "cstark_range_prove_batch: stride too small");
    return CSTARK_OK;
}

size_t cstark_tx_proof_size_bound(uint32_t n_tx, const cstark_options *opt) {
    if (!opt || n_tx == 0) return 0;
    unsigned log_b = 0, log_f = 2;
    while ((1u << log_b) < opt->blowup_factor && log_b < 6) log_b++;
    while ((1u << log_f) < opt->fri_folding_factor && log_f < 4) log_f++;
    unsigned log_N = 10 + log_b;
    while ((1u << (log_N - 10 - log_b)) < n_tx) log_N++;
    const size_t nq = opt->num_queries, layers = log_N / log_f + 1, em = opt->field_extension + 1; // em: words per drawn-field element
    return 4096 + 32 * layers + (2 * 94 + 8) * 8 * em + nq * (94 * 8 + 8 * 8 * em + 2 * log_N * 32) +
           layers * (4 + nq * (((size_t)8 << log_f) * em + log_N * 32)) + 8 * em * (size_t)opt->fri_max_remainder;
}

int cstark_prove_stage_ms(cstark_ctx *c, float *ms /* [CSTARK_PROVE_NUM_STAGES] */) {
    if (!c || !ms) return fail(CSTARK_ERR_INVALID_ARG, "null argument");
    if (!c->arena || !c->arena->timed) return fail(CSTARK_ERR_INVALID_ARG, "no proof has been generated on this context");
    for (int i = 0; i < CSTARK_PROVE_NUM_STAGES; i++) HIP_TRY(hipEventElapsedTime(&ms[i], c->arena->ev[i], c->arena->ev[i + 1]));
    return CSTARK_OK;
}

} // extern "C"

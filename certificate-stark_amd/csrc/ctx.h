// Shared between capi.hip (stage entry points) and prove.hip (the whole-proof orchestrator): the context object behind the
// opaque cstark_ctx handle, its cached tables, and the error helpers.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <deque>
#include <vector>
#include "../../include/cstark.h"
#include "constraints.h"
#include "trace_gen.h"

// The prover waits for a root, a frame or an upload a dozen times per proof with the GPU idle until the host answers: poll the stream
// instead of blocking in hipStreamSynchronize (whose wake-up costs tens of microseconds; FRI stage 0.69 -> 0.58 ms).  The poll is
// BOUNDED: the short waits (FRI layers, frames: tens of microseconds) end inside it; a long one (the 10 ms of the trace commitment, a
// wedged stream) falls back to the blocking call after CSTARK_SPIN_US microseconds (default 200), so a context costs a host core only
// in the latency-critical tail of a proof and never spins forever.  CSTARK_SYNC_BLOCK=1: always block.
#include <chrono>
namespace cs {
inline hipError_t stream_wait(hipStream_t st) {
    static const bool block = [] { const char *e = getenv("CSTARK_SYNC_BLOCK"); return e && atoi(e) != 0; }();
    static const long spin_us = [] { const char *e = getenv("CSTARK_SPIN_US"); return e ? atol(e) : 200L; }();
    if (block || spin_us <= 0) return hipStreamSynchronize(st);
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e;
    for (unsigned it = 0; (e = hipStreamQuery(st)) == hipErrorNotReady; it++) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if ((it & 15) == 15 && std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us)
            return hipStreamSynchronize(st);
    }
    return e;
}
// The ONE wait of a proof whose channel runs on the device (prove.hip, prove_core_dev): the host has enqueued the whole proof and would
// poll for 25 ms; `near_end` is an event recorded before the last ~0.5 ms of work.  Sleep on that event, then poll the stream: the
// blocking wait's wake-up latency (50 - 150 us) is paid while the GPU is still busy, not after the proof.
inline hipError_t stream_wait_tail(hipStream_t st, hipEvent_t near_end) {
    static const bool block = [] { const char *e = getenv("CSTARK_SYNC_BLOCK"); return e && atoi(e) != 0; }();
    if (block) return hipStreamSynchronize(st);
    hipError_t e = hipEventSynchronize(near_end);
    if (e != hipSuccess) return e;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned it = 0; (e = hipStreamQuery(st)) == hipErrorNotReady; it++) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if ((it & 15) == 15 && std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > 5000)
            return hipStreamSynchronize(st);
    }
    return e;
}
} // namespace cs

namespace cs {

extern thread_local char g_err[512];

inline int fail(int code, const char *fmt, const char *detail = "") {
    snprintf(g_err, sizeof g_err, fmt, detail);
    return code;
}
#define HIP_TRY(expr)                                                                                                              \
    do {                                                                                                                           \
        hipError_t e_ = (expr);                                                                                                    \
        if (e_ != hipSuccess) return cs::fail(e_ == hipErrorOutOfMemory ? CSTARK_ERR_OOM : CSTARK_ERR_HIP, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)
#define RC_TRY(expr)            \
    do {                        \
        int rc_ = (expr);       \
        if (rc_) return rc_;    \
    } while (0)

struct NttPlan {
    unsigned log_n;
    uint64_t *w, *winv; // [n] each: powers of w_n and of its inverse
    uint64_t n_inv;
    uint64_t *aux_w = nullptr, *aux_winv = nullptr; // compact tables of the three-step kernels (ntt.h), null for other sizes
};
struct CosetTable {
    unsigned log_n, log_b;
    uint64_t offset;
    uint64_t *s; // [b][n]: (offset * w_{bn}^k)^m
    uint64_t *aux = nullptr; // [b][aux_words]: compact prescale / output-factor tables per coset (ntt.h), null for other sizes
    size_t aux_words = 0;
};
struct PeriodicTable {
    unsigned depth, log_n, log_b;
    uint64_t *tab;   // [b][48][1024]
    uint64_t *coset; // [b][CE_COSET_CONSTS]
    uint64_t *binv;  // [b][2][n] inverses of the boundary divisors
    int air = 0;     // cstark_air_id the table belongs to
};
// 1 / (x^m - zc) over the LDE domain for one assertion divisor: [b cosets][n / m] (x^m has period n / m in the row index of a coset)
struct AssertInverseTable {
    unsigned log_n, log_b;
    uint64_t m, zc;
    uint64_t *tab;
};
// what cstark_air_combine needs of an AIR that does not change from proof to proof (capi.hip, air_combine_impl): keyed by
// (AIR, items, trace length, blowup)
struct AirCombineStatic {
    int air;
    uint32_t n_items, log_n, log_b;
    AirCombineParams p;          // groups, coset powers, device pointers of the static tables; the per-call fields are patched in
    std::vector<uint32_t> t_grp; // degree group of every transition constraint (host copy)
    uint64_t *d_static;          // device: shifts[b] | a_reg | a_seq | a_grp | t_grp
};
struct ProveArena; // prove.hip
void prove_arena_free(ProveArena *a);

} // namespace cs

struct cstark_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false; // cstark_ctx_create_own_stream: destroyed with the context
    hipStream_t side = nullptr, side2 = nullptr; // internal streams (forked from / joined into `stream`)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr, ev_mid = nullptr;
    // uploaded witness
    void *wit_buf = nullptr;
    size_t wit_bytes = 0;
    cs::TxWitnessDev wit{};
    // cached tables (deque: references stay valid as entries are added) and workspace
    std::deque<cs::NttPlan> plans;
    std::deque<cs::CosetTable> cosets;
    std::deque<cs::PeriodicTable> periodic;
    uint64_t *coef_buf = nullptr; // device copy of the composition coefficients
    void *coef_stage = nullptr;   // pinned host staging of the same block: the upload is asynchronous, no wait for the caller's struct
    hipEvent_t coef_ev = nullptr; // recorded behind the upload; waited on before the staging block is rewritten
    std::deque<cs::AssertInverseTable> assert_inv;  // k_air_combine's divisor inverses, keyed by (m, zc, log_n, log_b)
    std::deque<cs::AirCombineStatic> air_static;    // cstark_air_combine's per-AIR tables
    uint64_t *air_coef_buf = nullptr, *air_coef_stage = nullptr; // its per-proof coefficient block (device) and the pinned staging of the upload
    size_t air_coef_words = 0;
    hipEvent_t air_coef_ev = nullptr;
    uint64_t *deep_buf = nullptr, *deep_stage = nullptr; // cstark_deep_composition's coefficient / frame block, likewise
    size_t deep_words = 0;
    hipEvent_t deep_ev = nullptr;
    std::deque<cs::PeriodicTable> small_periodic; // standalone sub-AIRs: keyed by (air, depth, log_n, log_b); coset/binv unused
    void *desc_buf = nullptr;     // device copy of a generic AIR description (cstark_air_combine)
    hipEvent_t part_ev[cs::CE_NUM_PARTS + 1] = {}; // optional per-launch timing of the fused constraint evaluation
    bool part_timing = false, part_valid = false;
    // with part_timing: one event pair around every low-degree extension (cstark_lde_columns and the prover's own calls) since the
    // last cstark_lde_timing_ms; lde_units = column x coset transforms inside the pairs
    std::vector<hipEvent_t> lde_ev;
    size_t lde_ev_used = 0;
    uint64_t lde_units = 0;
    uint64_t *tail_buf = nullptr; // standalone SchnorrAir: message[26..28] per signature
    std::vector<uint64_t> schnorr_rx; // host copy of the signatures' R.x ([n][6]) for the sequence assertions
    std::vector<uint64_t> schnorr_av_stage; // [12][n / 512]: the non-zero coefficients of their value polynomials, staged for the upload
    std::vector<uint64_t> schnorr_pub; // messages [n][28] then R.x [n][6]: SchnorrAir's public inputs, for the channel seed
    std::vector<uint8_t> schnorr_s;    // the s halves [n][32]
    uint8_t schnorr_seed[32] = {}, schnorr_seed_key[12] = {}; // the channel seed of the uploaded Schnorr witness under one option set (key[11] = 1: valid)
    size_t tail_bytes = 0;
    size_t desc_bytes = 0;
    cs::ProveArena *arena = nullptr; // device buffers of cstark_tx_prove (prove.hip)
    void *ws = nullptr;
    size_t ws_bytes = 0;
    uint64_t *shard_bit37 = nullptr; // sharded split evaluation: register 37 on all eight cosets
    size_t shard_bit37_words = 0;
    void *rb_dev = nullptr, *rb_host = nullptr; // cstark_range_prove_batch: one device block and one pinned host block, carved per call
    size_t rb_dev_bytes = 0, rb_host_bytes = 0;
};
// internal (capi.hip): the coin of one FRI layer on the device followed by the fold with the drawn point (prove.hip)
int fri_coin_fold_dev(cstark_ctx *c, uint32_t *d_seed, const uint8_t *d_root, uint64_t *d_alpha, uint32_t *d_root_out, const uint64_t *d_evals,
                      uint64_t *d_out, uint32_t log_n, uint32_t log_f, uint64_t domain_offset);
// internal (capi.hip): row hashes of a whole table whose cosets are in block order (blake3.h): leaf b j + k = row j of LDE coset k
int hash_rows_slots(cstark_ctx *c, uint32_t hash_fn, const uint64_t *d_lde, uint8_t *d_leaves, uint32_t width, uint32_t log_n, uint32_t log_blowup, uint32_t log_s);
// internal (capi.hip): the cached twiddle tables of a 2^log_n-point domain: powers of w and of its inverse (device, n entries each)
int plan_tables(cstark_ctx *c, unsigned log_n, const uint64_t **w, const uint64_t **winv);

// internal (capi.hip): merged TransactionAir constraint evaluations for m coefficient sets in one pass over the frame
struct cstark_tx_coeffs;
// input_is_lde: the table is the low-degree extension of columns of degree < n (true for the prover's own table): allows the split
// evaluation of the Rescue windows on half of the cosets
int tx_evaluate_constraints_sets(cstark_ctx *c, const uint64_t *d_lde, const cstark_tx_coeffs *coeffs, uint32_t m, const uint64_t pub_inputs[4],
                                 uint64_t *const *d_outs, uint32_t merkle_depth, uint32_t log_n, uint32_t log_blowup, uint32_t k0, uint32_t nk,
                                 bool input_is_lde, const uint64_t *d_pub = nullptr);
int tx_coef_device_block(cstark_ctx *c, uint64_t **d_coef);
// internal (capi.hip): cstark_air_combine / the fused sub-AIR evaluators with coefficients (and assertion values) on the device
int air_combine_dev(cstark_ctx *c, int air, uint32_t n_items, int mode, uint32_t merkle_depth, const uint64_t *d_lde, const uint64_t *d_evals,
                    const uint64_t *d_aux_lde, const uint64_t *d_coefs, const uint64_t *d_avalues, const uint64_t *d_avals_lde, uint32_t n_avals, uint64_t *d_out,
                    uint32_t log_n, uint32_t log_blowup, uint32_t nk);
// internal (capi.hip): one rank's share of the degree-split evaluation of a proof sharded by LDE coset.  d_lde: the rank's cosets
// [k0, k0 + nk) (k0 even, nk = 2 or 4) of its own extension; d_coeffs: the 94 coefficient columns (register 37 is extended to all
// cosets here: the recombination reads it on cosets the rank does not hold).  d_out [nk / 2 + 4][n]: merged evaluations of the rank's
// even cosets, then its share of the four odd cosets (summed over the ranks by tx_shard_combine: d_parts [8 / nk][nk / 2 + 4][n] -> [8][n]).
int tx_evaluate_constraints_shard(cstark_ctx *c, const uint64_t *d_lde, const uint64_t *d_coeffs, const cstark_tx_coeffs *coeffs, const uint64_t pub_inputs[4],
                                  uint64_t *d_out, uint32_t merkle_depth, uint32_t log_n, uint32_t k0, uint32_t nk);
int tx_shard_combine(cstark_ctx *c, const uint64_t *d_parts, uint64_t *d_out, uint32_t log_n, uint32_t nk);
// internal (capi.hip): TransactionAir trace spread over the context's streams, nothing joined.  In stream order `stream` holds
// registers >= TX_COPY_COLS (closed forms); c->side: the Merkle recurrence, c->ev_join recorded behind it; c->side2: message hash
// (c->ev_mid behind it), then the curve ladders (c->ev_join2 behind them).  Registers [TX_LATE_COLS, TX_COPY_COLS) are complete
// after ev_join and ev_mid, registers [0, TX_LATE_COLS) after ev_join2 as well.
constexpr uint32_t TX_LATE_COLS = 37; // registers 0..36: the two curve points and the s-bit register between them
constexpr uint32_t TX_COPY_COLS = 65; // registers 65..93: key / amount copies and the sigma range accumulator
int tx_build_trace_split(cstark_ctx *c, uint64_t *d_trace);
// LDE of columns [col0, col0 + ncols) of a table of `width` columns (same layout and arguments as cstark_lde_columns)
int lde_column_range(cstark_ctx *c, const uint64_t *d_coeffs, uint64_t *d_lde, uint32_t width, uint32_t col0, uint32_t ncols, uint32_t log_n,
                     uint32_t log_blowup, uint64_t domain_offset, uint32_t k0, uint32_t nk);
// internal (capi.hip): cstark_deep_composition_ext restricted to the first nk cosets, d_out = [m][nk][n]
int evaluate_ood_frames(cstark_ctx *c, const uint64_t *d_coeffs, uint32_t width, const uint64_t *d_ccoef, uint32_t n_comp, uint32_t log_n,
                        const uint64_t zpts[2], uint64_t zb, uint64_t *out_trace, uint64_t *out_comp);
int ood_frames_dev(cstark_ctx *c, const uint64_t *d_coeffs, uint32_t width, const uint64_t *d_ccoef, uint32_t n_comp, uint32_t log_n, const uint64_t *d_pts,
                   uint64_t *d_out);
int deep_composition_ext_cosets(cstark_ctx *c, const uint64_t *d_trace_lde, const uint64_t *d_comp_lde, uint32_t width, uint32_t n_comp, uint32_t m,
                                const uint64_t *z, const uint64_t *ood_trace, const uint64_t *ood_comp, const uint64_t *alpha, const uint64_t *beta,
                                const uint64_t *delta, const uint64_t *deg_a, const uint64_t *deg_b, uint64_t *d_out, uint32_t log_n, uint32_t log_blowup,
                                uint32_t nk);

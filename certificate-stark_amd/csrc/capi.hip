// C ABI of the backend (include/cstark.h): context management, cached tables, witness upload and the stage
// entry points.  No torch types, no CPU fallback: without a HIP device every compute entry point returns
// CSTARK_ERR_NO_DEVICE (context creation fails).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <deque>
#include <new>
#include <vector>
#include "../../include/cstark.h"
#include "air_tx_host.h"
#include "blake3.h"
#include "ctx.h"
#include "constraints.h"
#include "deep.h"
#include "ext.h"
#include "hostfield.h"
#include "ntt.h"
#include "trace_gen.h"

namespace cs { thread_local char g_err[512] = ""; }
using cs::fail;
using cs::NttPlan;
using cs::CosetTable;
using cs::PeriodicTable;
using cs::g_err;

namespace {

// frees the device allocations registered with it unless release() is reached: error paths of the table builders leak nothing
struct DevGuard {
    std::vector<void *> ptrs;
    bool armed = true;
    template <class T> hipError_t alloc(T **p, size_t bytes) {
        const hipError_t e = hipMalloc((void **)p, bytes);
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
    void release() { armed = false; }
    ~DevGuard() { if (armed) for (void *q : ptrs) (void)hipFree(q); }
};

int get_plan(cstark_ctx *c, unsigned log_n, const NttPlan **out) {
    for (const NttPlan &p : c->plans)
        if (p.log_n == log_n) { *out = &p; return CSTARK_OK; }
    const size_t n = (size_t)1 << log_n;
    NttPlan p{log_n, nullptr, nullptr, 0};
    DevGuard g;
    HIP_TRY(g.alloc(&p.w, n * 8));
    HIP_TRY(g.alloc(&p.winv, n * 8));
    const uint64_t w = cs::host::root_of_unity(log_n);
    HIP_TRY(cs::ntt_power_table(p.w, n, w, c->stream));
    HIP_TRY(cs::ntt_power_table(p.winv, n, cs::host::inv(w), c->stream));
    p.n_inv = cs::host::inv(cs::host::from_u64(n));
    cs::NttV4Shape shape;
    if (cs::ntt_v4_shape(log_n, &shape)) {
        const size_t words = cs::ntt_aux_plan_words(shape);
        HIP_TRY(g.alloc(&p.aux_w, words * 8));
        HIP_TRY(g.alloc(&p.aux_winv, words * 8));
        HIP_TRY(cs::ntt_build_aux_plan(p.aux_w, p.w, shape, c->stream));
        HIP_TRY(cs::ntt_build_aux_plan(p.aux_winv, p.winv, shape, c->stream));
    }
    c->plans.push_back(p);
    g.release();
    *out = &c->plans.back();
    return CSTARK_OK;
}

int get_coset_table(cstark_ctx *c, unsigned log_n, unsigned log_b, uint64_t offset, const CosetTable **out) {
    for (const CosetTable &t : c->cosets)
        if (t.log_n == log_n && t.log_b == log_b && t.offset == offset) { *out = &t; return CSTARK_OK; }
    const size_t n = (size_t)1 << log_n, b = (size_t)1 << log_b;
    CosetTable t{log_n, log_b, offset, nullptr};
    DevGuard g;
    HIP_TRY(g.alloc(&t.s, b * n * 8));
    const uint64_t wbn = cs::host::root_of_unity(log_n + log_b);
    uint64_t shift = offset;
    for (size_t k = 0; k < b; k++) {
        HIP_TRY(cs::ntt_power_table(t.s + k * n, n, shift, c->stream));
        shift = cs::host::mul(shift, wbn);
    }
    cs::NttV4Shape shape;
    if (cs::ntt_v4_shape(log_n, &shape)) {
        const NttPlan *p;
        RC_TRY(get_plan(c, log_n, &p));
        t.aux_words = cs::ntt_aux_coset_words(shape);
        HIP_TRY(g.alloc(&t.aux, b * t.aux_words * 8));
        for (size_t k = 0; k < b; k++) HIP_TRY(cs::ntt_build_aux_coset(t.aux + k * t.aux_words, p->w, t.s + k * n, shape, c->stream));
    }
    c->cosets.push_back(t);
    g.release();
    *out = &c->cosets.back();
    return CSTARK_OK;
}

int ensure_ws(cstark_ctx *c, size_t bytes) {
    if (bytes <= c->ws_bytes) return CSTARK_OK;
    if (c->ws) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->ws)); c->ws = nullptr; c->ws_bytes = 0; }
    HIP_TRY(hipMalloc(&c->ws, bytes));
    c->ws_bytes = bytes;
    return CSTARK_OK;
}

} // namespace
int plan_tables(cstark_ctx *c, unsigned log_n, const uint64_t **w, const uint64_t **winv) {
    const NttPlan *p;
    RC_TRY(get_plan(c, log_n, &p));
    *w = p->w; *winv = p->winv;
    return CSTARK_OK;
}
namespace {

int interpolate_impl(cstark_ctx *c, uint64_t *d_evals, uint64_t *d_coeffs, uint32_t width, uint32_t log_n) {
    if (!c || !d_evals || !d_coeffs || width == 0) return fail(CSTARK_ERR_INVALID_ARG, "cstark_interpolate_columns: bad argument");
    if (d_evals == d_coeffs) return fail(CSTARK_ERR_INVALID_ARG, "cstark_interpolate_columns: output must not alias input");
    if (log_n < cs::NTT_MIN_LOG_N || log_n > cs::NTT_MAX_LOG_N) return fail(CSTARK_ERR_UNSUPPORTED, "domain size must be 2^6 .. 2^24");
    HIP_TRY(hipSetDevice(c->device));
    const NttPlan *p;
    RC_TRY(get_plan(c, log_n, &p));
    cs::NttArgs a{};
    a.in = d_evals; a.scratch = d_evals; a.out = d_coeffs;
    a.width = width; a.batch = 1; a.log_n = log_n;
    a.w = p->winv; a.post_scale = p->n_inv; a.do_scale = true; a.inverse = true; a.aux = p->aux_winv;
    HIP_TRY(cs::ntt_columns(a, c->stream));
    return CSTARK_OK;
}

// part timing: an event on the stream, from a pool that grows on demand up to LDE_EVENT_CAP (reset by cstark_lde_timing_ms; a caller
// that never collects stops being timed instead of growing the pool without bound)
constexpr size_t LDE_EVENT_CAP = 4096;
int lde_mark(cstark_ctx *c) {
    if (c->lde_ev_used == c->lde_ev.size()) {
        if (c->lde_ev.size() >= LDE_EVENT_CAP) return CSTARK_ERR_UNSUPPORTED; // pool full: this extension is not timed

        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        c->lde_ev.push_back(e);
    }
    HIP_TRY(hipEventRecord(c->lde_ev[c->lde_ev_used++], c->stream));
    return CSTARK_OK;
}

// All cosets of an extension go into ONE launch pair (the grid's batch dimension) when their intermediate fits this budget: the eight
// cosets of a coefficient tile then run together on one XCD and the coefficients are read from HBM once (ntt.hip, k_ntt_cols_v5).  7 GiB
// covers the 94 x 2^20 x 8 trace (6.3 GB; HBM holds 288 GB); CSTARK_LDE_BATCH_MB overrides (tuning; 0 = coset by coset as in round 2).
// (clamped to 0 .. 64 GiB: a negative or absurd value must not become a huge size_t).  Footprint: the workspace of a context grows to the
// largest batch it has extended -- 6.3 GB for the 94 x 2^20 x 8 trace -- and stays: every context (ProverPool worker, shard rank) holds
// its own.  If that allocation fails the extension falls back to one coset at a time (0.8 GB).
const size_t LDE_BATCH_WS_BYTES = [] {
    const char *e = getenv("CSTARK_LDE_BATCH_MB");
    if (!e) return (size_t)7 << 30;
    const long long mb = atoll(e);
    return mb <= 0 ? (size_t)0 : mb > (64ll << 10) ? (size_t)64 << 30 : (size_t)mb << 20;
}();
constexpr uint32_t LDE_COLUMN_GROUP = 1u << 30; // columns per group of the LDE (see lde_impl); 2^30 = all columns at once
int lde_impl_inner(cstark_ctx *c, const uint64_t *d_coeffs, uint64_t *d_lde, uint32_t width, uint32_t col0, uint32_t ncols, uint32_t log_n,
                   uint32_t log_blowup, uint64_t domain_offset, uint32_t k0, uint32_t nk);
int lde_impl(cstark_ctx *c, const uint64_t *d_coeffs, uint64_t *d_lde, uint32_t width, uint32_t col0, uint32_t ncols, uint32_t log_n, uint32_t log_blowup,
             uint64_t domain_offset, uint32_t k0, uint32_t nk) {
    // events come in pairs: without room for both, or if the first cannot be recorded, the extension simply is not timed
    bool timed = c && c->part_timing && c->lde_ev_used + 2 <= LDE_EVENT_CAP;
    const size_t mark = timed ? c->lde_ev_used : 0;
    if (timed && lde_mark(c) != CSTARK_OK) { c->lde_ev_used = mark; timed = false; }
    const int rc = lde_impl_inner(c, d_coeffs, d_lde, width, col0, ncols, log_n, log_blowup, domain_offset, k0, nk);
    if (timed) {
        if (rc == CSTARK_OK && lde_mark(c) == CSTARK_OK) c->lde_units += (uint64_t)ncols * nk << log_n;
        else c->lde_ev_used = mark; // never leave an unpaired event behind
    }
    return rc;
}
int lde_impl_inner(cstark_ctx *c, const uint64_t *d_coeffs, uint64_t *d_lde, uint32_t width, uint32_t col0, uint32_t ncols, uint32_t log_n,
                   uint32_t log_blowup, uint64_t domain_offset, uint32_t k0, uint32_t nk) {
    if (!c || !d_coeffs || !d_lde || width == 0 || ncols == 0 || (uint64_t)col0 + ncols > width) return fail(CSTARK_ERR_INVALID_ARG, "cstark_lde_columns: bad argument");
    if (log_n < cs::NTT_MIN_LOG_N || log_n > cs::NTT_MAX_LOG_N || log_blowup > 6) return fail(CSTARK_ERR_UNSUPPORTED, "unsupported domain size");
    if (domain_offset == 0 || domain_offset >= cs::host::P) return fail(CSTARK_ERR_INVALID_ARG, "domain offset must be a nonzero field element");
    if ((uint64_t)k0 + nk > (1ull << log_blowup)) return fail(CSTARK_ERR_INVALID_ARG, "coset range exceeds the blowup factor");
    HIP_TRY(hipSetDevice(c->device));
    const NttPlan *p;
    const CosetTable *t;
    RC_TRY(get_plan(c, log_n, &p));
    RC_TRY(get_coset_table(c, log_n, log_blowup, domain_offset, &t));
    const size_t n = (size_t)1 << log_n;
    RC_TRY(ensure_ws(c, (size_t)width * n * 8));
    // Optional column groups with the cosets inside (CSTARK_NTT_GROUP, tuning): the group's coefficients and the intermediate of
    // the two-pass transform would then stay within the 256 MB Infinity Cache.  Measured on MI355X at 2^20 x 94 x 8: all columns
    // at once 10.33 ms, groups of 32 / 16 / 8 / 4 columns 10.49 / 10.80 / 11.13 / 12.17 ms -- the transform is bound by its field
    // multiplications (11 per element for 20 butterfly levels), not by HBM, so the default stays one launch pair per coset.
    static const uint32_t group_env = [] { const char *e = getenv("CSTARK_NTT_GROUP"); return e ? (uint32_t)atoi(e) : 0u; }();
    const uint32_t group = group_env ? group_env : LDE_COLUMN_GROUP;
    // narrow tables (composition columns, periodic columns): all cosets in one launch pair -- the grid's batch dimension -- when the
    // intermediate of every coset fits the workspace budget; 2 launches of nk x the workgroups instead of 2 nk small ones
    if (nk > 1 && (size_t)nk * (group < ncols ? group : ncols) * n * 8 <= LDE_BATCH_WS_BYTES) {
        // (with CSTARK_NTT_GROUP: column groups, all cosets of a group in one launch pair -- the group's intermediate, nk x group x n
        // words, is then written and read back within a short window)
        bool batched_ok = true;
        for (uint32_t g0 = col0; g0 < col0 + ncols; g0 += group) {
            const uint32_t gw = col0 + ncols - g0 < group ? col0 + ncols - g0 : group;
            if (ensure_ws(c, (size_t)nk * gw * n * 8) != CSTARK_OK) { // no room for the batched intermediate: coset by coset below
                (void)hipGetLastError();
                if (g0 != col0) return fail(CSTARK_ERR_OOM, "workspace allocation failed in the middle of an extension");
                batched_ok = false;
                break;
            }
            cs::NttArgs a{};
            a.in = d_coeffs + (size_t)g0 * n; a.scratch = (uint64_t *)c->ws; a.out = d_lde + (size_t)g0 * n;
            a.width = gw; a.batch = nk; a.log_n = log_n;
            a.w = p->w; a.prescale = t->s + (size_t)k0 * n; a.prescale_batch_stride = n; a.do_scale = false;
            a.aux = p->aux_w; a.aux_ps = t->aux ? t->aux + (size_t)k0 * t->aux_words : nullptr; a.aux_ps_batch_stride = t->aux_words;
            a.in_batch_stride = 0; a.scratch_batch_stride = (size_t)gw * n; a.out_batch_stride = (size_t)width * n;
            HIP_TRY(cs::ntt_columns(a, c->stream));
        }
        if (batched_ok) return CSTARK_OK;
        RC_TRY(ensure_ws(c, (size_t)width * n * 8));
    }
    for (uint32_t g0 = col0; g0 < col0 + ncols; g0 += group) {
        const uint32_t gw = col0 + ncols - g0 < group ? col0 + ncols - g0 : group;
        for (uint32_t k = k0; k < k0 + nk; k++) {
            cs::NttArgs a{};
            a.in = d_coeffs + (size_t)g0 * n; a.scratch = (uint64_t *)c->ws; a.out = d_lde + ((size_t)(k - k0) * width + g0) * n;
            a.width = gw; a.batch = 1; a.log_n = log_n;
            a.w = p->w; a.prescale = t->s + (size_t)k * n; a.do_scale = false;
            a.aux = p->aux_w; a.aux_ps = t->aux ? t->aux + (size_t)k * t->aux_words : nullptr;
            HIP_TRY(cs::ntt_columns(a, c->stream));
        }
    }
    return CSTARK_OK;
}

// K7: periodic values of the 48 mask / round-constant columns over the constraint-evaluation domain, plus the
// per-coset scalars of the driver.  Built once per (depth, trace length, blowup) and cached.
int get_periodic(cstark_ctx *c, unsigned depth, unsigned log_n, unsigned log_b, const PeriodicTable **out) {
    for (const PeriodicTable &t : c->periodic)
        if (t.depth == depth && t.log_n == log_n && t.log_b == log_b) { *out = &t; return CSTARK_OK; }
    if (log_n < 10) return fail(CSTARK_ERR_INVALID_ARG, "the trace must hold at least one 1024-row transaction");
    std::vector<uint64_t> cols;
    if (!cs::host::tx_periodic_columns(depth, cols)) return fail(CSTARK_ERR_INVALID_ARG, "unsupported Merkle depth");
    const size_t n = (size_t)1 << log_n, b = (size_t)1 << log_b, C = cs::host::TX_CYCLE, NP = cs::host::TX_NUM_PERIODIC;
    PeriodicTable t{depth, log_n, log_b, nullptr, nullptr, nullptr};
    uint64_t *d_cols = nullptr, *d_poly = nullptr;
    DevGuard keep, tmp; // g: the table (kept on success); tmp: scratch (always freed)
    HIP_TRY(tmp.alloc(&d_cols, NP * C * 8));
    HIP_TRY(tmp.alloc(&d_poly, NP * C * 8));
    HIP_TRY(keep.alloc(&t.tab, b * NP * C * 8));
    HIP_TRY(keep.alloc(&t.coset, b * cs::CE_COSET_CONSTS * 8));
    HIP_TRY(hipMemcpyAsync(d_cols, cols.data(), NP * C * 8, hipMemcpyHostToDevice, c->stream));
    RC_TRY(interpolate_impl(c, d_cols, d_poly, (uint32_t)NP, 10));
    // a column of period 1024 is a polynomial in x^(n/1024): evaluate it over offset' * <w_{b*1024}>, offset' = g^(n/1024)
    const uint64_t g = cs::host::lde_offset();
    RC_TRY(lde_impl(c, d_poly, t.tab, (uint32_t)NP, 0, (uint32_t)NP, 10, log_b, cs::host::pow(g, n / C), 0, (uint32_t)b));
    // per-coset scalars: shift_k = g w_{bn}^k, 1/(shift^n - 1), shift^adj_g, shift^badj
    std::vector<uint64_t> cc(b * cs::CE_COSET_CONSTS);
    const uint64_t wbn = cs::host::root_of_unity(log_n + log_b);
    uint64_t shift = g;
    for (size_t k = 0; k < b; k++) {
        uint64_t *o = cc.data() + k * cs::CE_COSET_CONSTS;
        o[0] = shift;
        o[1] = cs::host::inv(cs::host::sub(cs::host::pow(shift, n), cs::host::ONE));
        for (int gi = 0; gi < 5; gi++) o[2 + gi] = cs::host::pow(shift, cs::host::tx_group_adjustment(gi, n, n * b));
        o[7] = cs::host::pow(shift, cs::host::tx_boundary_adjustment(n, n * b));
        o[8] = cs::host::pow(shift, n - 1); // lift between neighbouring degree groups (merged split polynomials)
        o[9] = 0;
        shift = cs::host::mul(shift, wbn);
    }
    HIP_TRY(hipMemcpyAsync(t.coset, cc.data(), cc.size() * 8, hipMemcpyHostToDevice, c->stream));
    const NttPlan *plan;
    RC_TRY(get_plan(c, log_n, &plan));
    HIP_TRY(keep.alloc(&t.binv, b * 2 * n * 8));
    HIP_TRY(cs::build_boundary_inverses(t.binv, plan->w, t.coset, cs::host::inv(cs::host::root_of_unity(log_n)), log_n, log_b, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream)); // cc / cols go out of scope; tmp frees the scratch
    keep.release();
    c->periodic.push_back(t);
    *out = &c->periodic.back();
    return CSTARK_OK;
}

int ce_params(cstark_ctx *c, const uint64_t *d_lde, uint64_t *d_out, uint32_t merkle_depth, uint32_t log_n, uint32_t log_blowup, uint32_t k0,
              uint32_t nk, cs::CeParams *p) {
    if (!c || !d_lde || !d_out) return fail(CSTARK_ERR_INVALID_ARG, "constraint evaluation: null argument");
    if (((uintptr_t)d_lde & 15) != 0) return fail(CSTARK_ERR_INVALID_ARG, "constraint evaluation: d_lde must be 16-byte aligned (16-byte LDS-DMA pieces)");
    if (log_n < 10 || log_n > cs::NTT_MAX_LOG_N) return fail(CSTARK_ERR_INVALID_ARG, "trace length must be 2^10 .. 2^24");
    if (log_blowup != 3) return fail(CSTARK_ERR_UNSUPPORTED, "TransactionAir needs a constraint-evaluation blowup of 8 (max degree 7, src/air.rs:76-108)");
    if ((uint64_t)k0 + nk > (1ull << log_blowup) || nk == 0) return fail(CSTARK_ERR_INVALID_ARG, "coset range exceeds the blowup factor");
    HIP_TRY(hipSetDevice(c->device));
    const PeriodicTable *pt;
    const NttPlan *plan;
    RC_TRY(get_periodic(c, merkle_depth, log_n, log_blowup, &pt));
    RC_TRY(get_plan(c, log_n, &plan));
    const uint64_t n = 1ull << log_n, ce = n << log_blowup;
    *p = cs::CeParams{};
    p->lde = d_lde; p->ptab = pt->tab; p->w = plan->w; p->coset = pt->coset; p->binv = pt->binv; p->out = d_out;
    p->w_last = cs::host::inv(cs::host::root_of_unity(log_n));
    for (int g = 0; g < 5; g++) p->adj_mod_n[g] = (uint32_t)(cs::host::tx_group_adjustment(g, n, ce) & (n - 1));
    p->badj_mod_n = (uint32_t)(cs::host::tx_boundary_adjustment(n, ce) & (n - 1));
    p->log_n = log_n; p->log_b = log_blowup; p->k0 = k0;
    return CSTARK_OK;
}

} // namespace

extern "C" {

const char *cstark_last_error(void) { return g_err; }
const char *cstark_version(void) { return "certificate-stark_amd 0.2 (gfx950)"; }

int cstark_ctx_create(int device, void *stream, cstark_ctx **out) {
    if (!out) return fail(CSTARK_ERR_INVALID_ARG, "cstark_ctx_create: out is null");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail(CSTARK_ERR_NO_DEVICE, "no HIP device visible (this backend has no CPU fallback)");
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= count) return fail(CSTARK_ERR_INVALID_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    cstark_ctx *c = new (std::nothrow) cstark_ctx();
    if (!c) return fail(CSTARK_ERR_OOM, "host allocation failed");
    c->device = device;
    c->stream = (hipStream_t)stream; // NULL is HIP's default stream
    // the internal streams carry latency-bound recurrences (one wave per transaction and chain) that run beside chip-filling
    // transforms on the caller's stream: highest dispatch priority, so that their workgroups are placed as soon as they are ready
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipStreamCreateWithPriority(&c->side2, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join2, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_mid, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return fail(CSTARK_ERR_HIP, "could not create the internal stream / events");
    }
    *out = c;
    return CSTARK_OK;
}
int cstark_ctx_create_own_stream(int device, cstark_ctx **out) {
    if (!out) return fail(CSTARK_ERR_INVALID_ARG, "cstark_ctx_create_own_stream: out is null");
    RC_TRY(cstark_ctx_create(device, nullptr, out));
    cstark_ctx *c = *out;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        c->stream = nullptr;
        cstark_ctx_destroy(c);
        *out = nullptr;
        return fail(CSTARK_ERR_HIP, "could not create the context's stream");
    }
    c->owns_stream = true;
    return CSTARK_OK;
}

void cstark_ctx_destroy(cstark_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
    if (c->side2) { (void)hipStreamSynchronize(c->side2); (void)hipStreamDestroy(c->side2); }
    if (c->ev_join2) (void)hipEventDestroy(c->ev_join2);
    if (c->ev_mid) (void)hipEventDestroy(c->ev_mid);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->wit_buf) (void)hipFree(c->wit_buf);
    if (c->ws) (void)hipFree(c->ws);
    if (c->coef_buf) (void)hipFree(c->coef_buf);
    if (c->coef_stage) (void)hipHostFree(c->coef_stage);
    if (c->coef_ev) (void)hipEventDestroy(c->coef_ev);
    for (NttPlan &p : c->plans) { (void)hipFree(p.w); (void)hipFree(p.winv); (void)hipFree(p.aux_w); (void)hipFree(p.aux_winv); }
    for (CosetTable &t : c->cosets) { (void)hipFree(t.s); (void)hipFree(t.aux); }
    for (PeriodicTable &t : c->periodic) { (void)hipFree(t.tab); (void)hipFree(t.coset); (void)hipFree(t.binv); }
    for (PeriodicTable &t : c->small_periodic) (void)hipFree(t.tab);
    for (cs::AssertInverseTable &t : c->assert_inv) (void)hipFree(t.tab);
    for (cs::AirCombineStatic &t : c->air_static) (void)hipFree(t.d_static);
    if (c->air_coef_buf) (void)hipFree(c->air_coef_buf);
    if (c->air_coef_stage) (void)hipHostFree(c->air_coef_stage);
    if (c->air_coef_ev) (void)hipEventDestroy(c->air_coef_ev);
    if (c->deep_buf) (void)hipFree(c->deep_buf);
    if (c->deep_stage) (void)hipHostFree(c->deep_stage);
    if (c->deep_ev) (void)hipEventDestroy(c->deep_ev);
    if (c->desc_buf) (void)hipFree(c->desc_buf);
    if (c->rb_dev) (void)hipFree(c->rb_dev);
    if (c->shard_bit37) (void)hipFree(c->shard_bit37);
    if (c->rb_host) (void)hipHostFree(c->rb_host);
    if (c->tail_buf) (void)hipFree(c->tail_buf);
    if (c->arena) cs::prove_arena_free(c->arena);
    for (hipEvent_t e : c->part_ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->lde_ev) (void)hipEventDestroy(e);
    if (c->owns_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int cstark_ctx_synchronize(cstark_ctx *c) {
    if (!c) return fail(CSTARK_ERR_INVALID_ARG, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CSTARK_OK;
}

int cstark_malloc(cstark_ctx *c, size_t bytes, void **d_ptr) {
    if (!c || !d_ptr) return fail(CSTARK_ERR_INVALID_ARG, "cstark_malloc: null argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(d_ptr, bytes));
    return CSTARK_OK;
}
int cstark_free(cstark_ctx *c, void *d_ptr) {
    if (!c) return fail(CSTARK_ERR_INVALID_ARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(d_ptr));
    return CSTARK_OK;
}
int cstark_memcpy_h2d(cstark_ctx *c, void *d_dst, const void *src, size_t bytes) {
    if (!c || (!d_dst && bytes) || (!src && bytes)) return fail(CSTARK_ERR_INVALID_ARG, "cstark_memcpy_h2d: null argument");
    HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CSTARK_OK;
}
int cstark_memcpy_d2h(cstark_ctx *c, void *dst, const void *d_src, size_t bytes) {
    if (!c || (!dst && bytes) || (!d_src && bytes)) return fail(CSTARK_ERR_INVALID_ARG, "cstark_memcpy_d2h: null argument");
    HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CSTARK_OK;
}

// ---- K1 ------------------------------------------------------------------------------------------
int cstark_tx_witness_upload(cstark_ctx *c, const cstark_tx_witness *w) {
    if (!c || !w) return fail(CSTARK_ERR_INVALID_ARG, "cstark_tx_witness_upload: null argument");
    const size_t n = w->n_tx, d = w->merkle_depth;
    if (n == 0) return fail(CSTARK_ERR_INVALID_ARG, "n_tx must be positive");
    // (depth+1) must be a power of two (src/lib.rs:102-105) and 8*depth+7 <= 511 rows (src/merkle/constants.rs:27-29)
    if (d == 0 || ((d + 1) & d) != 0 || 8 * d + 7 > 511) return fail(CSTARK_ERR_INVALID_ARG, "tree depth must be one less than a power of 2 and at most 31");
    if (!w->initial_roots || !w->s_old_values || !w->r_old_values || !w->s_indices || !w->r_indices || !w->s_paths || !w->r_paths ||
        !w->deltas || !w->sig_rx || !w->sig_s)
        return fail(CSTARK_ERR_INVALID_ARG, "witness array pointer is null");
    HIP_TRY(hipSetDevice(c->device));
    // one allocation, 8-byte fields first
    const size_t sz[] = {n * 7 * 8, n * 14 * 8, n * 14 * 8, n * 8, n * 8, n * (d + 1) * 7 * 8, n * (d + 1) * 7 * 8, n * 8, n * 6 * 8, n * 4 * 8, n * 32};
    const void *src[] = {w->initial_roots, w->s_old_values, w->r_old_values, w->s_indices, w->r_indices, w->s_paths, w->r_paths, w->deltas, w->sig_rx, nullptr, w->sig_s};
    size_t off[12] = {0};
    for (int i = 0; i < 11; i++) off[i + 1] = off[i] + ((sz[i] + 255) & ~(size_t)255);
    if (off[11] > c->wit_bytes) {
        if (c->wit_buf) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->wit_buf)); c->wit_buf = nullptr; c->wit_bytes = 0; }
        HIP_TRY(hipMalloc(&c->wit_buf, off[11]));
        c->wit_bytes = off[11];
    }
    char *base = (char *)c->wit_buf;
    for (int i = 0; i < 11; i++)
        if (src[i]) HIP_TRY(hipMemcpyAsync(base + off[i], src[i], sz[i], hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream)); // the caller may free its host arrays on return
    cs::TxWitnessDev &dv = c->wit;
    dv.n_tx = (uint32_t)n;
    dv.depth = (uint32_t)d;
    dv.initial_roots = (const uint64_t *)(base + off[0]);
    dv.s_old = (const uint64_t *)(base + off[1]);
    dv.r_old = (const uint64_t *)(base + off[2]);
    dv.s_idx = (const uint64_t *)(base + off[3]);
    dv.r_idx = (const uint64_t *)(base + off[4]);
    dv.s_paths = (const uint64_t *)(base + off[5]);
    dv.r_paths = (const uint64_t *)(base + off[6]);
    dv.deltas = (const uint64_t *)(base + off[7]);
    dv.sig_rx = (const uint64_t *)(base + off[8]);
    dv.h_limbs = (uint64_t *)(base + off[9]);
    dv.sig_s = (const uint8_t *)(base + off[10]);
    dv.msg_tail = nullptr;
    return CSTARK_OK;
}

int cstark_tx_build_trace(cstark_ctx *c, uint64_t *d_trace) {
    if (!c || !d_trace) return fail(CSTARK_ERR_INVALID_ARG, "cstark_tx_build_trace: null argument");
    if (!c->wit_buf || c->wit.n_tx == 0) return fail(CSTARK_ERR_INVALID_ARG, "no witness uploaded");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(cs::launch_trace_gen(c->wit, d_trace, c->stream, c->side, c->ev_fork, c->ev_join));
    return CSTARK_OK;
}

// ---- K2 / K3 ---------------------------------------------------------------------------------------
uint64_t cstark_field_generator(void) { return cs::host::generator(); }
uint64_t cstark_field_lde_offset(void) { return cs::host::lde_offset(); }
uint64_t cstark_field_root_of_unity(uint32_t log_n) { return log_n <= 55 ? cs::host::root_of_unity(log_n) : 0; }

int cstark_interpolate_columns(cstark_ctx *c, uint64_t *d_evals, uint64_t *d_coeffs, uint32_t width, uint32_t log_n) {
    return interpolate_impl(c, d_evals, d_coeffs, width, log_n);
}
int cstark_lde_columns(cstark_ctx *c, const uint64_t *d_coeffs, uint64_t *d_lde, uint32_t width, uint32_t log_n, uint32_t log_blowup,
                       uint64_t domain_offset, uint32_t k0, uint32_t nk) {
    return lde_impl(c, d_coeffs, d_lde, width, 0, width, log_n, log_blowup, domain_offset, k0, nk);
}

// ---- composition polynomial (first "next" row: engine into_poly + column split) -----------------------------------
int cstark_composition_columns(cstark_ctx *c, const uint64_t *d_combined, uint64_t *d_cols, uint32_t log_n, uint32_t log_blowup) {
    if (!c || !d_combined || !d_cols) return fail(CSTARK_ERR_INVALID_ARG, "cstark_composition_columns: null argument");
    if (log_n < cs::NTT_MIN_LOG_N || log_n + log_blowup > cs::NTT_MAX_LOG_N || log_blowup == 0 || log_blowup > 6)
        return fail(CSTARK_ERR_UNSUPPORTED, "composition domain must be 2^7 .. 2^24 points");
    HIP_TRY(hipSetDevice(c->device));
    const size_t N = (size_t)1 << (log_n + log_blowup);
    RC_TRY(ensure_ws(c, 2 * N * 8));
    uint64_t *nat = (uint64_t *)c->ws, *h = nat + N;
    if (log_blowup <= 3) {
        // The evaluation domain is the union of b cosets of the n-point domain and the table is coset-major: b interpolations of
        // size n (the register-tiled kernels, cosets as columns), then per row a twist and a b-point inverse DFT across the cosets
        // -- instead of a transposition and one transform of size b n.  Same coefficients (exact arithmetic).
        const NttPlan *pn, *pN;
        RC_TRY(get_plan(c, log_n, &pn));
        RC_TRY(get_plan(c, log_n + log_blowup, &pN));
        cs::NttArgs a{};
        a.in = d_combined; a.scratch = h; a.out = nat; a.width = 1u << log_blowup; a.batch = 1; a.log_n = log_n;
        a.w = pn->winv; a.post_scale = pn->n_inv; a.do_scale = true; a.inverse = true; a.aux = pn->aux_winv;
        HIP_TRY(cs::ntt_columns(a, c->stream));
        const uint64_t b_inv = cs::host::inv(cs::host::from_u64(1ull << log_blowup));
        HIP_TRY(cs::coset_combine(nat, h, log_n, log_blowup, pN->winv, b_inv, c->stream));
        HIP_TRY(cs::split_columns(h, d_cols, log_n, log_blowup, cs::host::inv(cs::host::lde_offset()), c->stream));
        return CSTARK_OK;
    }
    HIP_TRY(cs::interleave_cosets(d_combined, nat, log_n, log_blowup, c->stream));
    // interpolate over the whole evaluation domain (offset handled by the g^-m scaling of the split)
    const NttPlan *p;
    RC_TRY(get_plan(c, log_n + log_blowup, &p));
    cs::NttArgs a{};
    a.in = nat; a.scratch = nat; a.out = h; a.width = 1; a.batch = 1; a.log_n = log_n + log_blowup;
    a.w = p->winv; a.post_scale = p->n_inv; a.do_scale = true; a.inverse = true; a.aux = p->aux_winv;
    HIP_TRY(cs::ntt_columns(a, c->stream));
    HIP_TRY(cs::split_columns(h, d_cols, log_n, log_blowup, cs::host::inv(cs::host::lde_offset()), c->stream));
    return CSTARK_OK;
}

// ---- out-of-domain frame and DEEP composition ("next" rows) -----------------------------------------------------------
int cstark_evaluate_polys_at(cstark_ctx *c, const uint64_t *d_coeffs, uint32_t width, uint32_t log_n, const uint64_t *points, uint32_t npts,
                             uint64_t *out /* host [npts][width] */) {
    if (!c || !d_coeffs || !points || !out || width == 0 || npts == 0 || npts > 16) return fail(CSTARK_ERR_INVALID_ARG, "cstark_evaluate_polys_at: bad argument");
    if (log_n > 30) return fail(CSTARK_ERR_INVALID_ARG, "bad polynomial size");
    HIP_TRY(hipSetDevice(c->device));
    const size_t need = ((size_t)npts + (size_t)npts * width + cs::poly_eval_scratch_words(width, log_n, npts)) * 8;
    if (need > c->desc_bytes) {
        if (c->desc_buf) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->desc_buf)); c->desc_buf = nullptr; c->desc_bytes = 0; }
        HIP_TRY(hipMalloc(&c->desc_buf, need));
        c->desc_bytes = need;
    }
    uint64_t *d_pts = (uint64_t *)c->desc_buf, *d_out = d_pts + npts, *d_scr = d_out + (size_t)npts * width;
    HIP_TRY(hipMemcpyAsync(d_pts, points, (size_t)npts * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(cs::poly_eval(d_coeffs, width, log_n, d_pts, npts, d_out, d_scr, c->stream));
    HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)npts * width * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(cs::stream_wait(c->stream));
    return CSTARK_OK;
}

int cstark_deep_composition(cstark_ctx *c, const uint64_t *d_trace_lde, const uint64_t *d_comp_lde, uint32_t width, uint32_t n_comp,
                            uint64_t z, const uint64_t *ood_trace, const uint64_t *ood_comp, const uint64_t *alpha, const uint64_t *beta,
                            const uint64_t *delta, uint64_t deg_a, uint64_t deg_b, uint64_t *d_out, uint32_t log_n, uint32_t log_blowup,
                            uint32_t k0, uint32_t nk) {
    if (!c || !d_trace_lde || !d_comp_lde || !ood_trace || !ood_comp || !alpha || !beta || !delta || !d_out || nk == 0 || width == 0 || n_comp == 0)
        return fail(CSTARK_ERR_INVALID_ARG, "cstark_deep_composition: bad argument");
    if (log_n < cs::NTT_MIN_LOG_N || log_n > cs::NTT_MAX_LOG_N || log_blowup > 6 || (uint64_t)k0 + nk > (1ull << log_blowup)) return fail(CSTARK_ERR_INVALID_ARG, "bad domain parameters");
    HIP_TRY(hipSetDevice(c->device));
    const NttPlan *plan;
    RC_TRY(get_plan(c, log_n, &plan));
    const size_t b = (size_t)1 << log_blowup, nco = 2 * (size_t)width + n_comp, words = 2 * nco + b;
    // coefficients | frame | coset offsets through a pinned staging block into a device block of the context: no wait between the upload
    // and the launch (round 3 uploaded from a transient vector and waited)
    if (c->deep_words < words) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->deep_buf) { HIP_TRY(hipFree(c->deep_buf)); c->deep_buf = nullptr; }
        if (c->deep_stage) { HIP_TRY(hipHostFree(c->deep_stage)); c->deep_stage = nullptr; }
        c->deep_words = 0;
        HIP_TRY(hipMalloc((void **)&c->deep_buf, words * 8));
        HIP_TRY(hipHostMalloc((void **)&c->deep_stage, words * 8, hipHostMallocDefault));
        if (!c->deep_ev) HIP_TRY(hipEventCreateWithFlags(&c->deep_ev, hipEventDisableTiming));
        c->deep_words = words;
    } else {
        HIP_TRY(hipEventSynchronize(c->deep_ev));
    }
    uint64_t *blk = c->deep_stage;
    memcpy(blk, alpha, width * 8); memcpy(blk + width, beta, width * 8); memcpy(blk + 2 * width, delta, n_comp * 8);
    memcpy(blk + nco, ood_trace, 2 * (size_t)width * 8); memcpy(blk + nco + 2 * width, ood_comp, n_comp * 8);
    const uint64_t wbn = cs::host::root_of_unity(log_n + log_blowup);
    uint64_t shift = cs::host::lde_offset();
    for (size_t k = 0; k < b; k++) { blk[2 * nco + k] = shift; shift = cs::host::mul(shift, wbn); }
    HIP_TRY(hipMemcpyAsync(c->deep_buf, blk, words * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipEventRecord(c->deep_ev, c->stream));
    const uint64_t *d = c->deep_buf;
    cs::DeepParams p{};
    p.trace_lde = d_trace_lde; p.comp_lde = d_comp_lde; p.w = plan->w; p.coef = d; p.ood = d + nco; p.shifts = d + 2 * nco; p.out = d_out;
    p.z = z; p.zw = cs::host::mul(z, cs::host::root_of_unity(log_n)); p.zb = cs::host::pow(z, n_comp);
    p.deg_a = deg_a; p.deg_b = deg_b; p.width = width; p.nb = n_comp; p.log_n = log_n; p.k0 = k0;
    HIP_TRY(cs::deep_composition(p, nk, c->stream));
    return CSTARK_OK;
}

// ---- FRI: natural-order view of coset-major evaluations and one folding step ("next" rows) ---------------------------
int cstark_interleave_cosets(cstark_ctx *c, const uint64_t *d_coset_major, uint64_t *d_natural, uint32_t log_n, uint32_t log_blowup) {
    if (!c || !d_coset_major || !d_natural || d_coset_major == d_natural) return fail(CSTARK_ERR_INVALID_ARG, "cstark_interleave_cosets: bad argument");
    if (log_n + log_blowup > 30 || log_blowup > 6) return fail(CSTARK_ERR_INVALID_ARG, "bad domain parameters");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(cs::interleave_cosets(d_coset_major, d_natural, log_n, log_blowup, c->stream));
    return CSTARK_OK;
}
int cstark_fri_fold4(cstark_ctx *c, const uint64_t *d_evals, uint64_t *d_out, uint32_t log_n, uint64_t domain_offset, uint64_t alpha) {
    if (!c || !d_evals || !d_out || d_evals == d_out) return fail(CSTARK_ERR_INVALID_ARG, "cstark_fri_fold4: bad argument");
    if (log_n < cs::NTT_MIN_LOG_N || log_n > cs::NTT_MAX_LOG_N) return fail(CSTARK_ERR_UNSUPPORTED, "layer size must be 2^6 .. 2^24");
    if (domain_offset == 0 || domain_offset >= cs::host::P || alpha >= cs::host::P) return fail(CSTARK_ERR_INVALID_ARG, "offset / alpha must be field elements");
    HIP_TRY(hipSetDevice(c->device));
    const NttPlan *p;
    RC_TRY(get_plan(c, log_n, &p));
    HIP_TRY(cs::fri_fold4(d_evals, d_out, log_n, p->winv, cs::host::inv(domain_offset), alpha, cs::host::inv(cs::host::from_u64(4)), c->stream));
    return CSTARK_OK;
}
// folding_factor = 4, 8 or 16 (FriOptions, examples/state-transition.rs:46-47)
int cstark_fri_fold(cstark_ctx *c, const uint64_t *d_evals, uint64_t *d_out, uint32_t log_n, uint32_t folding_factor, uint64_t domain_offset, uint64_t alpha) {
    if (!c || !d_evals || !d_out || d_evals == d_out) return fail(CSTARK_ERR_INVALID_ARG, "cstark_fri_fold: bad argument");
    if (folding_factor != 4 && folding_factor != 8 && folding_factor != 16) return fail(CSTARK_ERR_UNSUPPORTED, "FRI folding factor must be 4, 8 or 16");
    if (log_n < cs::NTT_MIN_LOG_N || log_n > cs::NTT_MAX_LOG_N) return fail(CSTARK_ERR_UNSUPPORTED, "layer size must be 2^6 .. 2^24");
    if (domain_offset == 0 || domain_offset >= cs::host::P || alpha >= cs::host::P) return fail(CSTARK_ERR_INVALID_ARG, "offset / alpha must be field elements");
    HIP_TRY(hipSetDevice(c->device));
    const NttPlan *p;
    RC_TRY(get_plan(c, log_n, &p));
    const unsigned log_f = folding_factor == 4 ? 2 : folding_factor == 8 ? 3 : 4;
    HIP_TRY(cs::fri_fold(d_evals, d_out, log_n, log_f, p->winv, cs::host::inv(domain_offset), alpha, cs::host::inv(cs::host::from_u64(folding_factor)), c->stream));
    return CSTARK_OK;
}

// ---- FieldExtension::Quadratic / Cubic: the same three stages over the degree-m extension (ext.hip) ---------------------------
int cstark_evaluate_polys_at_ext(cstark_ctx *c, const uint64_t *d_coeffs, uint32_t width, uint32_t log_n, uint32_t m, const uint64_t *z, uint64_t *out) {
    if (!c || !d_coeffs || !z || !out || width == 0 || (m != 2 && m != 3)) return fail(CSTARK_ERR_INVALID_ARG, "cstark_evaluate_polys_at_ext: bad argument");
    if (log_n > 30) return fail(CSTARK_ERR_INVALID_ARG, "bad polynomial size");
    for (uint32_t i = 0; i < m; i++) if (z[i] >= cs::host::P) return fail(CSTARK_ERR_INVALID_ARG, "the point is not an extension element");
    HIP_TRY(hipSetDevice(c->device));
    const size_t need = ((size_t)m * width + cs::poly_eval_ext_scratch_words(width, log_n, m)) * 8;
    if (need > c->desc_bytes) {
        if (c->desc_buf) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->desc_buf)); c->desc_buf = nullptr; c->desc_bytes = 0; }
        HIP_TRY(hipMalloc(&c->desc_buf, need));
        c->desc_bytes = need;
    }
    uint64_t *d_out = (uint64_t *)c->desc_buf, *d_scr = d_out + (size_t)m * width;
    HIP_TRY(cs::poly_eval_ext(d_coeffs, width, log_n, z, m, d_out, d_scr, c->stream));
    HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)m * width * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(cs::stream_wait(c->stream));
    return CSTARK_OK;
}
} // extern "C"

// internal (ctx.h): one FRI layer's coin on the device (Blake3 coin: reseed with the layer root at d_root, draw the folding point) and the
// fold with that point; d_state = [seed: 8 words][alpha: one element per layer][roots: 8 words per layer]
int fri_coin_fold_dev(cstark_ctx *c, uint32_t *d_seed, const uint8_t *d_root, uint64_t *d_alpha, uint32_t *d_root_out, const uint64_t *d_evals,
                      uint64_t *d_out, uint32_t log_n, uint32_t log_f, uint64_t domain_offset) {
    if (log_n < cs::NTT_MIN_LOG_N || log_n > cs::NTT_MAX_LOG_N) return fail(CSTARK_ERR_UNSUPPORTED, "layer size must be 2^6 .. 2^24");
    const NttPlan *p;
    RC_TRY(get_plan(c, log_n, &p));
    HIP_TRY(cs::fri_coin(d_seed, d_root, d_alpha, d_root_out, c->stream));
    HIP_TRY(cs::fri_fold(d_evals, d_out, log_n, log_f, p->winv, cs::host::inv(domain_offset), 0, cs::host::inv(cs::host::from_u64(1ull << log_f)), c->stream, d_alpha));
    return CSTARK_OK;
}
// internal (ctx.h): row hashes of a whole table whose cosets are in block order (blake3.h, lde_coset_slot): leaf b j + k = row j of LDE coset k
int hash_rows_slots(cstark_ctx *c, uint32_t hash_fn, const uint64_t *d_lde, uint8_t *d_leaves, uint32_t width, uint32_t log_n, uint32_t log_blowup, uint32_t log_s) {
    if (!c || !d_lde || !d_leaves || width == 0 || width > 128 || hash_fn > 1 || log_blowup > 6 || log_s > log_blowup || ((uintptr_t)d_leaves & 15) != 0)
        return fail(CSTARK_ERR_INVALID_ARG, "hash_rows_slots: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    if (hash_fn == 1) HIP_TRY(cs::hash_rows_sha3(d_lde, d_leaves, width, log_n, log_blowup, 0, 1u << log_blowup, c->stream, log_s));
    else HIP_TRY(cs::hash_rows(d_lde, d_leaves, width, log_n, log_blowup, 0, 1u << log_blowup, c->stream, log_s));
    return CSTARK_OK;
}
// internal (ctx.h): both halves of the out-of-domain frame with ONE upload, one readback and one wait -- the trace polynomials at
// (z, z w) and the composition columns at z^b; out_trace [2][width], out_comp [n_comp] (host)
int evaluate_ood_frames(cstark_ctx *c, const uint64_t *d_coeffs, uint32_t width, const uint64_t *d_ccoef, uint32_t n_comp, uint32_t log_n,
                        const uint64_t zpts[2], uint64_t zb, uint64_t *out_trace, uint64_t *out_comp) {
    if (!c || !d_coeffs || !d_ccoef || !out_trace || !out_comp || width == 0 || n_comp == 0) return fail(CSTARK_ERR_INVALID_ARG, "evaluate_ood_frames: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t n_out = 2 * (size_t)width + n_comp;
    const size_t scr_t = cs::poly_eval_scratch_words(width, log_n, 2), scr_c = cs::poly_eval_scratch_words(n_comp, log_n, 1);
    const size_t need = (4 + n_out + scr_t + scr_c) * 8;
    if (need > c->desc_bytes) {
        if (c->desc_buf) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->desc_buf)); c->desc_buf = nullptr; c->desc_bytes = 0; }
        HIP_TRY(hipMalloc(&c->desc_buf, need));
        c->desc_bytes = need;
    }
    uint64_t *d_pts = (uint64_t *)c->desc_buf, *d_out = d_pts + 4, *d_scr = d_out + n_out;
    const uint64_t pts[3] = {zpts[0], zpts[1], zb};
    HIP_TRY(hipMemcpyAsync(d_pts, pts, sizeof pts, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(cs::poly_eval(d_coeffs, width, log_n, d_pts, 2, d_out, d_scr, c->stream));
    HIP_TRY(cs::poly_eval(d_ccoef, n_comp, log_n, d_pts + 2, 1, d_out + 2 * (size_t)width, d_scr + scr_t, c->stream));
    std::vector<uint64_t> host(n_out); // (pageable on purpose: a pinned landing area measured 0.03 ms SLOWER per proof over these small copies)
    HIP_TRY(hipMemcpyAsync(host.data(), d_out, n_out * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(cs::stream_wait(c->stream));
    memcpy(out_trace, host.data(), 2 * (size_t)width * 8);
    memcpy(out_comp, host.data() + 2 * (size_t)width, (size_t)n_comp * 8);
    return CSTARK_OK;
}

// internal (ctx.h): the same frame with the points and the values staying on the device (the device-side channel): d_pts = z, z w, z^b;
// d_out = T(z)[width] | T(z w)[width] | H_i(z^b)[n_comp]
int ood_frames_dev(cstark_ctx *c, const uint64_t *d_coeffs, uint32_t width, const uint64_t *d_ccoef, uint32_t n_comp, uint32_t log_n, const uint64_t *d_pts,
                   uint64_t *d_out) {
    if (!c || !d_coeffs || !d_ccoef || !d_pts || !d_out || width == 0 || n_comp == 0) return fail(CSTARK_ERR_INVALID_ARG, "ood_frames_dev: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t scr_t = cs::poly_eval_scratch_words(width, log_n, 2), scr_c = cs::poly_eval_scratch_words(n_comp, log_n, 1);
    const size_t need = (scr_t + scr_c) * 8;
    if (need > c->desc_bytes) {
        if (c->desc_buf) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->desc_buf)); c->desc_buf = nullptr; c->desc_bytes = 0; }
        HIP_TRY(hipMalloc(&c->desc_buf, need));
        c->desc_bytes = need;
    }
    uint64_t *d_scr = (uint64_t *)c->desc_buf;
    HIP_TRY(cs::poly_eval(d_coeffs, width, log_n, d_pts, 2, d_out, d_scr, c->stream));
    HIP_TRY(cs::poly_eval(d_ccoef, n_comp, log_n, d_pts + 2, 1, d_out + 2 * (size_t)width, d_scr + scr_t, c->stream));
    return CSTARK_OK;
}
// internal (ctx.h): the first nk cosets only, d_out = [m][nk][n]
int deep_composition_ext_cosets(cstark_ctx *c, const uint64_t *d_trace_lde, const uint64_t *d_comp_lde, uint32_t width, uint32_t n_comp, uint32_t m,
                                const uint64_t *z, const uint64_t *ood_trace, const uint64_t *ood_comp, const uint64_t *alpha, const uint64_t *beta,
                                const uint64_t *delta, const uint64_t *deg_a, const uint64_t *deg_b, uint64_t *d_out, uint32_t log_n, uint32_t log_blowup,
                                uint32_t nk) {
    if (!c || !d_trace_lde || !d_comp_lde || !z || !ood_trace || !ood_comp || !alpha || !beta || !delta || !deg_a || !deg_b || !d_out || width == 0 || n_comp == 0 ||
        (m != 2 && m != 3))
        return fail(CSTARK_ERR_INVALID_ARG, "cstark_deep_composition_ext: bad argument");
    if (log_n < cs::NTT_MIN_LOG_N || log_n > cs::NTT_MAX_LOG_N || log_blowup > 6) return fail(CSTARK_ERR_INVALID_ARG, "bad domain parameters");
    using namespace cs::host;
    HIP_TRY(hipSetDevice(c->device));
    const NttPlan *plan;
    RC_TRY(get_plan(c, log_n, &plan));
    const size_t b = (size_t)1 << log_blowup, nco = (size_t)m * (2 * (size_t)width + n_comp);
    std::vector<uint64_t> blk(nco + b);
    memcpy(blk.data(), alpha, (size_t)m * width * 8);
    memcpy(blk.data() + (size_t)m * width, beta, (size_t)m * width * 8);
    memcpy(blk.data() + (size_t)2 * m * width, delta, (size_t)m * n_comp * 8);
    const uint64_t wbn = root_of_unity(log_n + log_blowup);
    uint64_t shift = lde_offset();
    for (size_t k = 0; k < b; k++) { blk[nco + k] = shift; shift = mul(shift, wbn); }
    const size_t bytes = blk.size() * 8;
    if (bytes > c->desc_bytes) {
        if (c->desc_buf) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->desc_buf)); c->desc_buf = nullptr; c->desc_bytes = 0; }
        HIP_TRY(hipMalloc(&c->desc_buf, bytes));
        c->desc_bytes = bytes;
    }
    HIP_TRY(hipMemcpyAsync(c->desc_buf, blk.data(), bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(cs::stream_wait(c->stream));
    cs::DeepExtParams p{};
    p.trace_lde = d_trace_lde; p.comp_lde = d_comp_lde; p.w = plan->w; p.coef = (const uint64_t *)c->desc_buf; p.shifts = p.coef + nco; p.out = d_out;
    const EX zz = ex_load(z, m), zw = ex_scale(zz, root_of_unity(log_n)), zb = ex_pow(zz, n_comp, m), da = ex_load(deg_a, m), db = ex_load(deg_b, m);
    EX k1 = ex_zero(), k2 = ex_zero(), k3 = ex_zero();
    for (uint32_t i = 0; i < width; i++) {
        k1 = ex_add(k1, ex_mul(ex_load(alpha + (size_t)m * i, m), ex_load(ood_trace + (size_t)m * i, m), m));
        k2 = ex_add(k2, ex_mul(ex_load(beta + (size_t)m * i, m), ex_load(ood_trace + (size_t)m * (width + i), m), m));
    }
    for (uint32_t i = 0; i < n_comp; i++) k3 = ex_add(k3, ex_mul(ex_load(delta + (size_t)m * i, m), ex_load(ood_comp + (size_t)m * i, m), m));
    for (int q = 0; q < 3; q++) {
        p.z[q] = zz.c[q]; p.zw[q] = zw.c[q]; p.zb[q] = zb.c[q]; p.deg_a[q] = da.c[q]; p.deg_b[q] = db.c[q];
        p.k1[q] = k1.c[q]; p.k2[q] = k2.c[q]; p.k3[q] = k3.c[q];
    }
    p.width = width; p.nb = n_comp; p.log_n = log_n; p.log_b = log_blowup; p.m = m;
    if (nk > b) return fail(CSTARK_ERR_INVALID_ARG, "coset range exceeds the blowup factor");
    p.nk = nk == b ? 0 : nk;
    HIP_TRY(cs::deep_composition_ext(p, c->stream));
    return CSTARK_OK;
}
extern "C" {
int cstark_deep_composition_ext(cstark_ctx *c, const uint64_t *d_trace_lde, const uint64_t *d_comp_lde, uint32_t width, uint32_t n_comp, uint32_t m,
                                const uint64_t *z, const uint64_t *ood_trace, const uint64_t *ood_comp, const uint64_t *alpha, const uint64_t *beta,
                                const uint64_t *delta, const uint64_t *deg_a, const uint64_t *deg_b, uint64_t *d_out, uint32_t log_n, uint32_t log_blowup) {
    if (log_blowup > 6) return fail(CSTARK_ERR_INVALID_ARG, "bad domain parameters");
    return deep_composition_ext_cosets(c, d_trace_lde, d_comp_lde, width, n_comp, m, z, ood_trace, ood_comp, alpha, beta, delta, deg_a, deg_b, d_out, log_n,
                                       log_blowup, 1u << log_blowup);
}
int cstark_fri_fold4_ext(cstark_ctx *c, const uint64_t *d_evals, uint64_t *d_out, uint32_t log_n, uint64_t domain_offset, uint32_t m, const uint64_t *alpha) {
    if (!c || !d_evals || !d_out || !alpha || d_evals == d_out || (m != 2 && m != 3)) return fail(CSTARK_ERR_INVALID_ARG, "cstark_fri_fold4_ext: bad argument");
    if (log_n < cs::NTT_MIN_LOG_N || log_n > cs::NTT_MAX_LOG_N) return fail(CSTARK_ERR_UNSUPPORTED, "layer size must be 2^6 .. 2^24");
    if (domain_offset == 0 || domain_offset >= cs::host::P) return fail(CSTARK_ERR_INVALID_ARG, "offset must be a nonzero field element");
    for (uint32_t i = 0; i < m; i++) if (alpha[i] >= cs::host::P) return fail(CSTARK_ERR_INVALID_ARG, "alpha is not an extension element");
    HIP_TRY(hipSetDevice(c->device));
    const NttPlan *p;
    RC_TRY(get_plan(c, log_n, &p));
    HIP_TRY(cs::fri_fold4_ext(d_evals, d_out, log_n, p->winv, cs::host::inv(domain_offset), alpha, m, cs::host::inv(cs::host::from_u64(4)), c->stream));
    return CSTARK_OK;
}
int cstark_fri_fold_ext(cstark_ctx *c, const uint64_t *d_evals, uint64_t *d_out, uint32_t log_n, uint32_t folding_factor, uint64_t domain_offset, uint32_t m,
                        const uint64_t *alpha) {
    if (!c || !d_evals || !d_out || !alpha || d_evals == d_out || (m != 2 && m != 3)) return fail(CSTARK_ERR_INVALID_ARG, "cstark_fri_fold_ext: bad argument");
    if (folding_factor != 4 && folding_factor != 8 && folding_factor != 16) return fail(CSTARK_ERR_UNSUPPORTED, "FRI folding factor must be 4, 8 or 16");
    if (log_n < cs::NTT_MIN_LOG_N || log_n > cs::NTT_MAX_LOG_N) return fail(CSTARK_ERR_UNSUPPORTED, "layer size must be 2^6 .. 2^24");
    if (domain_offset == 0 || domain_offset >= cs::host::P) return fail(CSTARK_ERR_INVALID_ARG, "offset must be a nonzero field element");
    for (uint32_t i = 0; i < m; i++) if (alpha[i] >= cs::host::P) return fail(CSTARK_ERR_INVALID_ARG, "alpha is not an extension element");
    HIP_TRY(hipSetDevice(c->device));
    const NttPlan *p;
    RC_TRY(get_plan(c, log_n, &p));
    const unsigned log_f = folding_factor == 4 ? 2 : folding_factor == 8 ? 3 : 4;
    HIP_TRY(cs::fri_fold_ext(d_evals, d_out, log_n, log_f, p->winv, cs::host::inv(domain_offset), alpha, m, cs::host::inv(cs::host::from_u64(folding_factor)), c->stream));
    return CSTARK_OK;
}

// ---- K4 / K5 ---------------------------------------------------------------------------------------
int cstark_hash_rows(cstark_ctx *c, const uint64_t *d_lde, uint8_t *d_leaves, uint32_t width, uint32_t log_n, uint32_t log_blowup,
                     uint32_t k0, uint32_t nk) {
    if (!c || !d_lde || !d_leaves) return fail(CSTARK_ERR_INVALID_ARG, "cstark_hash_rows: null argument");
    if (width == 0 || width > 128) return fail(CSTARK_ERR_UNSUPPORTED, "row width must be 1..128 elements (single Blake3 chunk)");
    if (log_n > 30 || log_blowup > 6 || (uint64_t)k0 + nk > (1ull << log_blowup)) return fail(CSTARK_ERR_INVALID_ARG, "bad domain parameters");
    if (((uintptr_t)d_leaves & 15) != 0) return fail(CSTARK_ERR_INVALID_ARG, "d_leaves must be 16-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(cs::hash_rows(d_lde, d_leaves, width, log_n, log_blowup, k0, nk, c->stream));
    return CSTARK_OK;
}

int cstark_hash_rows_fn(cstark_ctx *c, uint32_t hash_fn, const uint64_t *d_lde, uint8_t *d_leaves, uint32_t width, uint32_t log_n, uint32_t log_blowup,
                        uint32_t k0, uint32_t nk) {
    if (hash_fn == 0) return cstark_hash_rows(c, d_lde, d_leaves, width, log_n, log_blowup, k0, nk);
    if (hash_fn != 1) return fail(CSTARK_ERR_UNSUPPORTED, "hash_fn must be 0 (Blake3_256) or 1 (Sha3_256)");
    if (!c || !d_lde || !d_leaves) return fail(CSTARK_ERR_INVALID_ARG, "cstark_hash_rows_fn: null argument");
    if (width == 0 || width > 128) return fail(CSTARK_ERR_UNSUPPORTED, "row width must be 1..128 elements");
    if (log_n > 30 || log_blowup > 6 || (uint64_t)k0 + nk > (1ull << log_blowup)) return fail(CSTARK_ERR_INVALID_ARG, "bad domain parameters");
    if (((uintptr_t)d_leaves & 15) != 0) return fail(CSTARK_ERR_INVALID_ARG, "d_leaves must be 16-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(cs::hash_rows_sha3(d_lde, d_leaves, width, log_n, log_blowup, k0, nk, c->stream));
    return CSTARK_OK;
}
int cstark_merkle_build_fn(cstark_ctx *c, uint32_t hash_fn, uint8_t *d_nodes, uint32_t log_leaves) {
    if (hash_fn == 0) return cstark_merkle_build(c, d_nodes, log_leaves);
    if (hash_fn != 1) return fail(CSTARK_ERR_UNSUPPORTED, "hash_fn must be 0 (Blake3_256) or 1 (Sha3_256)");
    if (!c || !d_nodes) return fail(CSTARK_ERR_INVALID_ARG, "cstark_merkle_build_fn: null argument");
    if (log_leaves == 0 || log_leaves > 30) return fail(CSTARK_ERR_INVALID_ARG, "tree must have 2 .. 2^30 leaves");
    if (((uintptr_t)d_nodes & 15) != 0) return fail(CSTARK_ERR_INVALID_ARG, "d_nodes must be 16-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(cs::merkle_build_sha3(d_nodes, log_leaves, c->stream));
    return CSTARK_OK;
}

int cstark_merkle_build(cstark_ctx *c, uint8_t *d_nodes, uint32_t log_leaves) {
    if (!c || !d_nodes) return fail(CSTARK_ERR_INVALID_ARG, "cstark_merkle_build: null argument");
    if (log_leaves == 0 || log_leaves > 30) return fail(CSTARK_ERR_INVALID_ARG, "tree must have 2 .. 2^30 leaves");
    if (((uintptr_t)d_nodes & 15) != 0) return fail(CSTARK_ERR_INVALID_ARG, "d_nodes must be 16-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(cs::merkle_build(d_nodes, log_leaves, c->stream));
    return CSTARK_OK;
}

// ---- K6 / K7 ---------------------------------------------------------------------------------------
int cstark_tx_evaluate_transitions(cstark_ctx *c, const uint64_t *d_lde, uint64_t *d_out, uint32_t merkle_depth, uint32_t log_n,
                                   uint32_t log_blowup, uint32_t k0, uint32_t nk) {
    cs::CeParams p;
    RC_TRY(ce_params(c, d_lde, d_out, merkle_depth, log_n, log_blowup, k0, nk, &p));
    HIP_TRY(cs::launch_eval_transitions(p, nk, c->stream));
    return CSTARK_OK;
}

} // extern "C"
int lde_column_range(cstark_ctx *c, const uint64_t *d_coeffs, uint64_t *d_lde, uint32_t width, uint32_t col0, uint32_t ncols, uint32_t log_n,
                     uint32_t log_blowup, uint64_t domain_offset, uint32_t k0, uint32_t nk) {
    return lde_impl(c, d_coeffs, d_lde, width, col0, ncols, log_n, log_blowup, domain_offset, k0, nk);
}
int tx_build_trace_split(cstark_ctx *c, uint64_t *d_trace) {
    if (!c || !d_trace) return fail(CSTARK_ERR_INVALID_ARG, "tx_build_trace_split: null argument");
    if (!c->wit_buf || c->wit.n_tx == 0) return fail(CSTARK_ERR_INVALID_ARG, "no witness uploaded");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(cs::launch_trace_gen_split(c->wit, d_trace, c->stream, c->side, c->side2, c->ev_fork, c->ev_join, c->ev_mid, c->ev_join2));
    return CSTARK_OK;
}
// m coefficient sets over the same frame (the components of an extension-field proof): the constraint values are computed once.
// The m coefficient sets of a proof into the context's device block (coefficients, then the per-proof tables of the Rescue windows):
// the caller's struct may be transient, so it is copied into a pinned staging block and uploaded from there without waiting.
static constexpr size_t TX_COEF_WORDS = (size_t)cs::CE_MAX_SETS * cs::CE_COEF_WORDS;
static int ensure_coef_buf(cstark_ctx *c) {
    static_assert(sizeof(cstark_tx_coeffs) == cs::CE_COEF_WORDS * 8, "coefficient block layout");
    if (!c->coef_buf) HIP_TRY(hipMalloc((void **)&c->coef_buf, (TX_COEF_WORDS + (size_t)cs::CE_MAX_SETS * cs::CE_RTAB_WORDS) * 8));
    return CSTARK_OK;
}
// internal (ctx.h): the context's device block of composition coefficients (cstark_tx_coeffs layout, one set): the device-side channel
// draws them there, tx_evaluate_constraints_sets(coeffs = null) reads them
int tx_coef_device_block(cstark_ctx *c, uint64_t **d_coef) {
    HIP_TRY(hipSetDevice(c->device));
    RC_TRY(ensure_coef_buf(c));
    *d_coef = c->coef_buf;
    return CSTARK_OK;
}
static int upload_coeffs(cstark_ctx *c, const cstark_tx_coeffs *coeffs, uint32_t m, cs::CeParams &p) {
    constexpr size_t COEF_WORDS = TX_COEF_WORDS;
    RC_TRY(ensure_coef_buf(c));
    if (!c->coef_stage) { // the event first: the staging block is only published once both exist
        if (!c->coef_ev) HIP_TRY(hipEventCreateWithFlags(&c->coef_ev, hipEventDisableTiming));
        HIP_TRY(hipHostMalloc(&c->coef_stage, (size_t)cs::CE_MAX_SETS * sizeof(cstark_tx_coeffs), hipHostMallocDefault));
    } else {
        HIP_TRY(hipEventSynchronize(c->coef_ev)); // the previous upload has left the staging block (long ago, normally)
    }
    memcpy(c->coef_stage, coeffs, (size_t)m * sizeof(cstark_tx_coeffs));
    HIP_TRY(hipMemcpyAsync(c->coef_buf, c->coef_stage, (size_t)m * sizeof(cstark_tx_coeffs), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipEventRecord(c->coef_ev, c->stream));
    p.coef = c->coef_buf;
    p.rtab = c->coef_buf + COEF_WORDS;
    p.m = m;
    return CSTARK_OK;
}

// (internal: declared in ctx.h for the prover)
int tx_evaluate_constraints_sets(cstark_ctx *c, const uint64_t *d_lde, const cstark_tx_coeffs *coeffs, uint32_t m, const uint64_t pub_inputs[4],
                                 uint64_t *const *d_outs, uint32_t merkle_depth, uint32_t log_n, uint32_t log_blowup, uint32_t k0, uint32_t nk,
                                 bool input_is_lde, const uint64_t *d_pub) {
    // coeffs == null: one coefficient set already in the context's device block (tx_coef_device_block); d_pub != null: the 14 public
    // inputs on the device instead of pub_inputs (both: the device-side channel of prove.hip)
    if ((!coeffs && m != 1) || (!pub_inputs && !d_pub) || !d_outs) return fail(CSTARK_ERR_INVALID_ARG, "cstark_tx_evaluate_constraints: null argument");
    if (m < 1 || m > (uint32_t)cs::CE_MAX_SETS) return fail(CSTARK_ERR_INVALID_ARG, "cstark_tx_evaluate_constraints: 1..3 coefficient sets");
    for (uint32_t q = 1; q < m; q++)
        if (!d_outs[q]) return fail(CSTARK_ERR_INVALID_ARG, "cstark_tx_evaluate_constraints: null output");
    cs::CeParams p;
    RC_TRY(ce_params(c, d_lde, d_outs[0], merkle_depth, log_n, log_blowup, k0, nk, &p));
    if (coeffs) RC_TRY(upload_coeffs(c, coeffs, m, p));
    else { RC_TRY(ensure_coef_buf(c)); p.coef = c->coef_buf; p.rtab = c->coef_buf + TX_COEF_WORDS; p.m = 1; }
    for (uint32_t q = 1; q < m; q++) p.out_ext[q - 1] = d_outs[q];
    for (int i = 0; i < 4; i++) p.pub[i] = pub_inputs ? pub_inputs[i] : 0;
    p.pubd = d_pub;
    static const bool split_env = [] { const char *e = getenv("CSTARK_ROUNDS_SPLIT"); return !e || atoi(e) != 0; }(); // tuning / debugging
    hipEvent_t *pev = c->part_timing ? c->part_ev : nullptr;
    // input_is_lde: d_lde is the extension of columns of degree < n (the prover's own table), which the split evaluation relies on;
    // the public stage entry points evaluate every point directly and are exact for ANY table
    static const bool split_ext_env = [] { const char *e = getenv("CSTARK_ROUNDS_SPLIT_EXT"); return !e || atoi(e) != 0; }();
    if (split_env && input_is_lde && (m == 1 || split_ext_env) && k0 == 0 && nk == 8 && log_n + 3 <= cs::NTT_MAX_LOG_N) { // (twiddle tables of size 8n)
        // Parts whose merged polynomials have degree < 4n (constraints.hip: the Rescue windows with their flags; the doublings and the
        // addition of the generator with the flag factored out) run on the even cosets only; their thirteen polynomials are
        // interpolated over that 4n-point sub-domain, extended to the odd cosets by transforms of thirteen columns and recombined
        // at every point.  (Part timing: the extension and the recombination are counted with the last of these parts.)
        const size_t n = (size_t)1 << log_n;
        const unsigned T = cs::CE_SPLIT_TABLES * m; // every coefficient set has its own block of polynomials
        const NttPlan *pn, *p4, *p8;
        const CosetTable *t1;
        RC_TRY(get_plan(c, log_n, &pn));
        RC_TRY(get_plan(c, log_n + 2, &p4));
        RC_TRY(get_plan(c, log_n + 3, &p8));
        RC_TRY(get_coset_table(c, log_n, 3, cs::host::from_u64(1), &t1));
        const size_t region = (size_t)T * 4 * n; // words per array of T x 4 columns
        const size_t hcol = (size_t)2 * m * n;   // the final addition's two sums per set: one n-point table each
        RC_TRY(ensure_ws(c, (5 * region + 9 * hcol) * 8));
        uint64_t *even = (uint64_t *)c->ws, *sa = even + region, *sb = sa + region, *sc = sb + region, *odd = sc + region;
        uint64_t *fin_direct = odd + region, *fin_hi = fin_direct + hcol /* [4 odd cosets][2 m][n] */, *fin_co = fin_hi + 4 * hcol, *fin_scr = fin_co + hcol /* [3] */;
        if (pev) HIP_TRY(hipEventRecord(pev[0], c->stream));
        HIP_TRY(cs::launch_rounds_setup(p, c->stream));
        HIP_TRY(cs::launch_rounds_split(p, even, c->stream));
        if (pev) HIP_TRY(hipEventRecord(pev[1], c->stream));
        uint64_t *fam_dbl = even + (size_t)cs::CE_SPLIT_FAM0 * 4 * n, *fam_add = fam_dbl + 12 * n, *fam_addbit = fam_add + 8 * n; // 3 | 2 | 2 tables
        HIP_TRY(cs::launch_ec_split(p, 1, fam_dbl, nullptr, c->stream));
        if (pev) HIP_TRY(hipEventRecord(pev[2], c->stream));
        HIP_TRY(cs::launch_ec_split(p, 2, fam_add, nullptr, c->stream));
        if (pev) HIP_TRY(hipEventRecord(pev[3], c->stream));
        HIP_TRY(cs::launch_ec_split(p, 3, fam_dbl, nullptr, c->stream));
        if (pev) HIP_TRY(hipEventRecord(pev[4], c->stream));
        HIP_TRY(cs::launch_ec_split(p, 4, fam_addbit, fam_add, c->stream));
        // the final addition reaches degree 5 (n - 1): its two sums on the four even cosets join the tables, ONE odd coset pins the
        // n coefficients above 4n (below, after the extension); constraints.hip, k_final_split
        if (pev) HIP_TRY(hipEventRecord(pev[5], c->stream));
        uint64_t *fam_final = fam_addbit + 8 * n;
        HIP_TRY(cs::launch_final_split(p, -1, fam_final, c->stream));
        // the three linear groups: one pass over the frame (k_lin_all); CSTARK_LIN_MERGED=0 (tuning / debugging): the three launches
        // of round 2.  Part timing: the merged pass is reported as lin_a, lin_b = 0, lin_c = the extension and recombination below.
        static const bool lin_merged = [] { const char *e = getenv("CSTARK_LIN_MERGED"); return !e || atoi(e) != 0; }();
        if (lin_merged) {
            if (pev) HIP_TRY(hipEventRecord(pev[6], c->stream));
            HIP_TRY(cs::launch_lin_all(p, even, c->stream));
            if (pev) { HIP_TRY(hipEventRecord(pev[7], c->stream)); HIP_TRY(hipEventRecord(pev[8], c->stream)); }
        } else {
            for (int part = 6; part <= 8; part++) {
                if (pev) HIP_TRY(hipEventRecord(pev[part], c->stream));
                HIP_TRY(cs::launch_lin_split(p, part, even, c->stream));
            }
        }
        cs::NttArgs a{};
        a.in = even; a.scratch = sa; a.out = sb; a.width = 4 * T; a.batch = 1; a.log_n = log_n; // every polynomial on every even coset
        a.w = pn->winv; a.post_scale = pn->n_inv; a.do_scale = true; a.inverse = true; a.aux = pn->aux_winv;
        HIP_TRY(cs::ntt_columns(a, c->stream));
        // interpolants of the even cosets -> inputs of the odd cosets' transforms (the 4n coefficients are never written)
        HIP_TRY(cs::coset_even_to_odd(sb, sa, log_n, T, p4->winv, p8->w, cs::host::inv(cs::host::from_u64(4)), c->stream)); // sa = [4 odd cosets][T][n]
        cs::NttArgs f{};
        f.in = sa; f.scratch = sc; f.out = odd; f.width = T; f.batch = 4; f.log_n = log_n;
        f.w = pn->w; f.prescale = t1->s + n; f.prescale_batch_stride = 2 * n; f.do_scale = false; f.inverse = false;
        f.aux = pn->aux_w; f.aux_ps = t1->aux ? t1->aux + t1->aux_words : nullptr; f.aux_ps_batch_stride = 2 * t1->aux_words;
        f.in_batch_stride = (size_t)T * n; f.scratch_batch_stride = (size_t)T * n; f.out_batch_stride = (size_t)T * n;
        HIP_TRY(cs::ntt_columns(f, c->stream));
        {   // high parts of the final addition's sums: H = (T - Q) / 2 on LDE coset 1, interpolated there and extended to cosets 3, 5, 7
            HIP_TRY(cs::launch_final_split(p, 1, fin_direct, c->stream));
            HIP_TRY(cs::launch_final_hi(p, odd, fin_direct, fin_hi, cs::host::inv(cs::host::from_u64(2)), c->stream));
            cs::NttArgs hi_inv{};
            hi_inv.in = fin_hi; hi_inv.scratch = fin_scr; hi_inv.out = fin_co; hi_inv.width = 2 * m; hi_inv.batch = 1; hi_inv.log_n = log_n;
            hi_inv.w = pn->winv; hi_inv.post_scale = pn->n_inv; hi_inv.do_scale = true; hi_inv.inverse = true; hi_inv.aux = pn->aux_winv;
            HIP_TRY(cs::ntt_columns(hi_inv, c->stream));
            // coefficients of H(w_8n z) in z -> values on coset k: prescale by (w_8n^(k-1))^s, k - 1 = 2, 4, 6: rows 2, 4, 6 of the offset-1 table
            cs::NttArgs hi_fwd{};
            hi_fwd.in = fin_co; hi_fwd.scratch = fin_scr; hi_fwd.out = fin_hi + hcol; hi_fwd.width = 2 * m; hi_fwd.batch = 3; hi_fwd.log_n = log_n;
            hi_fwd.w = pn->w; hi_fwd.prescale = t1->s + 2 * n; hi_fwd.prescale_batch_stride = 2 * n; hi_fwd.do_scale = false; hi_fwd.inverse = false;
            hi_fwd.aux = pn->aux_w; hi_fwd.aux_ps = t1->aux ? t1->aux + 2 * t1->aux_words : nullptr; hi_fwd.aux_ps_batch_stride = 2 * t1->aux_words;
            hi_fwd.in_batch_stride = 0; hi_fwd.scratch_batch_stride = hcol; hi_fwd.out_batch_stride = hcol;
            HIP_TRY(cs::ntt_columns(hi_fwd, c->stream));
        }
        HIP_TRY(cs::launch_split_finish(p, even, odd, fin_hi, c->stream));
        if (pev) HIP_TRY(hipEventRecord(pev[cs::CE_NUM_PARTS], c->stream));
    } else {
        HIP_TRY(cs::launch_eval_constraints(p, nk, c->stream, pev));
    }
    c->part_valid = c->part_timing;
    return CSTARK_OK;
}
// One rank of a proof sharded by LDE coset: the split evaluation on the rank's even cosets, the extension of its (partial) tables to the
// odd cosets and the recombination of its share (constraints.hip: k_split_finish_shard).  Everything between the even-coset values
// and the merged evaluations of an odd coset is linear in those values, so the ranks' shares add up to what the single-GPU path
// writes; exact field arithmetic, hence bit-identical proofs.
int tx_evaluate_constraints_shard(cstark_ctx *c, const uint64_t *d_lde, const uint64_t *d_coeffs, const cstark_tx_coeffs *coeffs, const uint64_t pub_inputs[4],
                                  uint64_t *d_out, uint32_t merkle_depth, uint32_t log_n, uint32_t k0, uint32_t nk) {
    if (!coeffs || !pub_inputs || !d_out || !d_coeffs) return fail(CSTARK_ERR_INVALID_ARG, "sharded constraint evaluation: null argument");
    if ((nk != 2 && nk != 4) || (k0 % nk) || k0 + nk > 8) return fail(CSTARK_ERR_INVALID_ARG, "sharded split evaluation: a rank holds 2 or 4 consecutive cosets");
    if (log_n + 3 > cs::NTT_MAX_LOG_N) return fail(CSTARK_ERR_UNSUPPORTED, "trace too long for the split evaluation");
    cs::CeParams p;
    RC_TRY(ce_params(c, d_lde, d_out, merkle_depth, log_n, 3, k0, nk, &p));
    RC_TRY(upload_coeffs(c, coeffs, 1, p));
    for (int i = 0; i < 4; i++) p.pub[i] = pub_inputs[i];
    p.nkc = nk / 2;
    const unsigned kc0 = k0 / 2, nkc = nk / 2;
    const size_t n = (size_t)1 << log_n;
    const unsigned T = cs::CE_SPLIT_TABLES;
    const NttPlan *pn, *p4, *p8;
    const CosetTable *t1;
    RC_TRY(get_plan(c, log_n, &pn));
    RC_TRY(get_plan(c, log_n + 2, &p4));
    RC_TRY(get_plan(c, log_n + 3, &p8));
    RC_TRY(get_coset_table(c, log_n, 3, cs::host::from_u64(1), &t1));
    const size_t region = (size_t)T * 4 * n, hcol = (size_t)2 * n;
    // register 37 on all eight cosets first: the extension uses the workspace itself
    uint64_t *bit37 = nullptr;
    {
        if (!c->shard_bit37 || c->shard_bit37_words < 8 * n) {
            if (c->shard_bit37) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->shard_bit37)); c->shard_bit37 = nullptr; }
            HIP_TRY(hipMalloc((void **)&c->shard_bit37, 8 * n * 8));
            c->shard_bit37_words = 8 * n;
        }
        bit37 = c->shard_bit37;
        RC_TRY(lde_impl(c, d_coeffs + (size_t)37 * n, bit37, 1, 0, 1, log_n, 3, cs::host::lde_offset(), 0, 8));
    }
    RC_TRY(ensure_ws(c, (5 * region + 9 * hcol) * 8));
    uint64_t *even = (uint64_t *)c->ws, *sa = even + region, *sb = sa + region, *sc = sb + region, *odd = sc + region;
    uint64_t *fin_direct = odd + region, *fin_hi = fin_direct + hcol, *fin_co = fin_hi + 4 * hcol, *fin_scr = fin_co + hcol;
    HIP_TRY(cs::launch_rounds_setup(p, c->stream));
    HIP_TRY(cs::launch_rounds_split(p, even, c->stream));
    uint64_t *fam_dbl = even + (size_t)cs::CE_SPLIT_FAM0 * 4 * n, *fam_add = fam_dbl + 12 * n, *fam_addbit = fam_add + 8 * n, *fam_final = fam_addbit + 8 * n;
    HIP_TRY(cs::launch_ec_split(p, 1, fam_dbl, nullptr, c->stream));
    HIP_TRY(cs::launch_ec_split(p, 2, fam_add, nullptr, c->stream));
    HIP_TRY(cs::launch_ec_split(p, 3, fam_dbl, nullptr, c->stream));
    HIP_TRY(cs::launch_ec_split(p, 4, fam_addbit, fam_add, c->stream));
    HIP_TRY(cs::launch_final_split(p, -1, fam_final, c->stream));
    HIP_TRY(cs::launch_lin_all(p, even, c->stream));
    // interpolation of every table on the rank's even cosets: columns [kc0, kc0 + nkc) of each table's four (batch = table)
    cs::NttArgs a{};
    a.in = even + (size_t)kc0 * n; a.scratch = sa + (size_t)kc0 * n; a.out = sb + (size_t)kc0 * n; a.width = nkc; a.batch = T; a.log_n = log_n;
    a.in_batch_stride = 4 * n; a.scratch_batch_stride = 4 * n; a.out_batch_stride = 4 * n;
    a.w = pn->winv; a.post_scale = pn->n_inv; a.do_scale = true; a.inverse = true; a.aux = pn->aux_winv;
    HIP_TRY(cs::ntt_columns(a, c->stream));
    HIP_TRY(cs::coset_even_to_odd(sb, sa, log_n, T, p4->winv, p8->w, cs::host::inv(cs::host::from_u64(4)), c->stream, kc0, nkc));
    cs::NttArgs f{};
    f.in = sa; f.scratch = sc; f.out = odd; f.width = T; f.batch = 4; f.log_n = log_n;
    f.w = pn->w; f.prescale = t1->s + n; f.prescale_batch_stride = 2 * n; f.do_scale = false; f.inverse = false;
    f.aux = pn->aux_w; f.aux_ps = t1->aux ? t1->aux + t1->aux_words : nullptr; f.aux_ps_batch_stride = 2 * t1->aux_words;
    f.in_batch_stride = (size_t)T * n; f.scratch_batch_stride = (size_t)T * n; f.out_batch_stride = (size_t)T * n;
    HIP_TRY(cs::ntt_columns(f, c->stream));
    {   // high parts of the final addition: H = (T - Q) / 2 on LDE coset 1 -- Q (the direct evaluation there) only on the rank that holds coset 1
        if (k0 == 0) HIP_TRY(cs::launch_final_split(p, 1, fin_direct, c->stream));
        else HIP_TRY(hipMemsetAsync(fin_direct, 0, hcol * 8, c->stream));
        HIP_TRY(cs::launch_final_hi(p, odd, fin_direct, fin_hi, cs::host::inv(cs::host::from_u64(2)), c->stream));
        cs::NttArgs hi_inv{};
        hi_inv.in = fin_hi; hi_inv.scratch = fin_scr; hi_inv.out = fin_co; hi_inv.width = 2; hi_inv.batch = 1; hi_inv.log_n = log_n;
        hi_inv.w = pn->winv; hi_inv.post_scale = pn->n_inv; hi_inv.do_scale = true; hi_inv.inverse = true; hi_inv.aux = pn->aux_winv;
        HIP_TRY(cs::ntt_columns(hi_inv, c->stream));
        cs::NttArgs hi_fwd{};
        hi_fwd.in = fin_co; hi_fwd.scratch = fin_scr; hi_fwd.out = fin_hi + hcol; hi_fwd.width = 2; hi_fwd.batch = 3; hi_fwd.log_n = log_n;
        hi_fwd.w = pn->w; hi_fwd.prescale = t1->s + 2 * n; hi_fwd.prescale_batch_stride = 2 * n; hi_fwd.do_scale = false; hi_fwd.inverse = false;
        hi_fwd.aux = pn->aux_w; hi_fwd.aux_ps = t1->aux ? t1->aux + 2 * t1->aux_words : nullptr; hi_fwd.aux_ps_batch_stride = 2 * t1->aux_words;
        hi_fwd.in_batch_stride = 0; hi_fwd.scratch_batch_stride = hcol; hi_fwd.out_batch_stride = hcol;
        HIP_TRY(cs::ntt_columns(hi_fwd, c->stream));
    }
    HIP_TRY(cs::launch_split_finish_shard(p, even, odd, fin_hi, bit37, d_out, c->stream));
    return CSTARK_OK;
}
int tx_shard_combine(cstark_ctx *c, const uint64_t *d_parts, uint64_t *d_out, uint32_t log_n, uint32_t nk) {
    if (!c || !d_parts || !d_out || (nk != 2 && nk != 4)) return fail(CSTARK_ERR_INVALID_ARG, "tx_shard_combine: bad argument");
    HIP_TRY(cs::launch_shard_combine(d_parts, d_out, log_n, nk / 2, c->stream));
    return CSTARK_OK;
}

extern "C" {

int cstark_tx_evaluate_constraints(cstark_ctx *c, const uint64_t *d_lde, const cstark_tx_coeffs *coeffs, const uint64_t pub_inputs[4],
                                   uint64_t *d_out, uint32_t merkle_depth, uint32_t log_n, uint32_t log_blowup, uint32_t k0, uint32_t nk) {
    uint64_t *outs[1] = {d_out};
    return tx_evaluate_constraints_sets(c, d_lde, coeffs, 1, pub_inputs, outs, merkle_depth, log_n, log_blowup, k0, nk, false);
}

int cstark_tx_evaluate_constraints_lde(cstark_ctx *c, const uint64_t *d_lde, const cstark_tx_coeffs *coeffs, const uint64_t pub_inputs[4],
                                       uint64_t *d_out, uint32_t merkle_depth, uint32_t log_n) {
    uint64_t *outs[1] = {d_out};
    return tx_evaluate_constraints_sets(c, d_lde, coeffs, 1, pub_inputs, outs, merkle_depth, log_n, 3, 0, 8, true);
}

int cstark_tx_evaluate_constraints_ext(cstark_ctx *c, const uint64_t *d_lde, const cstark_tx_coeffs *coeffs, uint32_t m, const uint64_t pub_inputs[4],
                                       uint64_t *d_out, uint32_t merkle_depth, uint32_t log_n, uint32_t log_blowup, uint32_t k0, uint32_t nk) {
    if (!d_out || m < 1 || m > 3) return fail(CSTARK_ERR_INVALID_ARG, "cstark_tx_evaluate_constraints_ext: bad argument");
    const size_t comp = (size_t)nk << log_n;
    uint64_t *outs[3] = {d_out, d_out + comp, d_out + 2 * comp};
    return tx_evaluate_constraints_sets(c, d_lde, coeffs, m, pub_inputs, outs, merkle_depth, log_n, log_blowup, k0, nk, false);
}

// ---- standalone sub-AIRs -------------------------------------------------------------
int cstark_merkle_build_trace(cstark_ctx *c, uint64_t *d_trace) {
    if (!c || !d_trace) return fail(CSTARK_ERR_INVALID_ARG, "cstark_merkle_build_trace: null argument");
    if (!c->wit_buf || c->wit.n_tx == 0) return fail(CSTARK_ERR_INVALID_ARG, "no witness uploaded");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(cs::launch_merkle_trace(c->wit, d_trace, c->stream));
    return CSTARK_OK;
}
int cstark_range_build_trace(cstark_ctx *c, uint64_t number, uint64_t *d_trace) {
    if (!c || !d_trace) return fail(CSTARK_ERR_INVALID_ARG, "cstark_range_build_trace: null argument");
    if (number >= cs::host::P) return fail(CSTARK_ERR_INVALID_ARG, "number is not a field element");
    const uint64_t canonical = cs::host::to_u64(number);
    if (canonical >> 63) return fail(CSTARK_ERR_INVALID_ARG, "range proofs cover 63-bit values (src/range/tests.rs:54-62 panics above)");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(cs::launch_range_trace(canonical, d_trace, c->stream));
    return CSTARK_OK;
}
// Synthetic long form of the range accumulator (BASELINE.json "range-proof AIR, 2^16 steps"; the reference's trace is fixed at 64
// rows): words = the n/64 little-endian words of an (n-1)-bit integer V, host memory.  *number_out (optional) = V mod p, memory form.
int cstark_range_build_trace_bits(cstark_ctx *c, const uint64_t *words, uint32_t log_n, uint64_t *d_trace, uint64_t *number_out) {
    if (!c || !words || !d_trace) return fail(CSTARK_ERR_INVALID_ARG, "cstark_range_build_trace_bits: null argument");
    if (log_n < 6 || log_n > 21) return fail(CSTARK_ERR_INVALID_ARG, "trace length must be 2^6 .. 2^21");
    const size_t nw = (size_t)1 << (log_n - 6);
    if (words[nw - 1] >> 63) return fail(CSTARK_ERR_INVALID_ARG, "the value must have at most n - 1 bits (top bit of the last word clear)");
    HIP_TRY(hipSetDevice(c->device));
    RC_TRY(ensure_ws(c, 2 * nw * 8));
    uint64_t *d_words = (uint64_t *)c->ws, *d_prefix = d_words + nw;
    HIP_TRY(hipMemcpyAsync(d_words, words, nw * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(cs::launch_range_trace_bits(d_words, d_prefix, d_trace, log_n, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream)); // the caller's words may be transient
    if (number_out) { // Horner in base 2^64, most significant word first
        uint64_t v = 0;
        const uint64_t B = cs::host::R2; // 2^64 in memory form
        for (size_t i = nw; i-- > 0;) v = cs::host::add(cs::host::mul(v, B), cs::host::from_u64(words[i] % cs::host::P));
        *number_out = v;
    }
    return CSTARK_OK;
}

// SchnorrAir (src/schnorr): messages [n][28], signatures (R.x [n][6], s bytes [n][32]); host arrays are copied
int cstark_schnorr_witness_upload(cstark_ctx *c, uint32_t n_sig, const uint64_t *messages, const uint64_t *sig_rx, const uint8_t *sig_s) {
    if (!c || !messages || !sig_rx || !sig_s || n_sig == 0) return fail(CSTARK_ERR_INVALID_ARG, "cstark_schnorr_witness_upload: bad argument");
    // reuse the transaction-witness device view: message[0..12] -> s_old[0..12], [12..24] -> r_old[0..12], [24] -> deltas,
    // [25] -> s_old[13]; the two trailing elements go to msg_tail
    std::vector<uint64_t> s_old((size_t)n_sig * 14, 0), r_old((size_t)n_sig * 14, 0), deltas(n_sig), zeros((size_t)n_sig * 28, 0), tail((size_t)n_sig * 2);
    for (uint32_t t = 0; t < n_sig; t++) {
        const uint64_t *m = messages + 28 * (size_t)t;
        memcpy(&s_old[14 * (size_t)t], m, 12 * 8);
        s_old[14 * (size_t)t + 13] = m[25];
        memcpy(&r_old[14 * (size_t)t], m + 12, 12 * 8);
        deltas[t] = m[24];
        tail[2 * (size_t)t] = m[26];
        tail[2 * (size_t)t + 1] = m[27];
    }
    cstark_tx_witness w{};
    w.n_tx = n_sig; w.merkle_depth = 3;
    w.initial_roots = zeros.data(); w.final_root = zeros.data(); w.s_old_values = s_old.data(); w.r_old_values = r_old.data();
    w.s_indices = zeros.data(); w.r_indices = zeros.data(); w.s_paths = zeros.data(); w.r_paths = zeros.data();
    w.deltas = deltas.data(); w.sig_rx = sig_rx; w.sig_s = sig_s;
    RC_TRY(cstark_tx_witness_upload(c, &w));
    if (c->tail_bytes < tail.size() * 8) {
        if (c->tail_buf) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->tail_buf)); c->tail_buf = nullptr; c->tail_bytes = 0; }
        HIP_TRY(hipMalloc((void **)&c->tail_buf, tail.size() * 8));
        c->tail_bytes = tail.size() * 8;
    }
    HIP_TRY(hipMemcpyAsync(c->tail_buf, tail.data(), tail.size() * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->wit.msg_tail = c->tail_buf;
    c->schnorr_rx.assign(sig_rx, sig_rx + (size_t)n_sig * 6);
    c->schnorr_pub.assign(messages, messages + (size_t)n_sig * 28);
    c->schnorr_pub.insert(c->schnorr_pub.end(), sig_rx, sig_rx + (size_t)n_sig * 6);
    c->schnorr_s.assign(sig_s, sig_s + (size_t)n_sig * 32);
    memset(c->schnorr_seed_key, 0, sizeof c->schnorr_seed_key); // another witness: the cached channel seed is stale
    return CSTARK_OK;
}
int cstark_schnorr_build_trace(cstark_ctx *c, uint64_t *d_trace) {
    if (!c || !d_trace) return fail(CSTARK_ERR_INVALID_ARG, "cstark_schnorr_build_trace: null argument");
    if (!c->wit_buf || c->wit.n_tx == 0 || !c->wit.msg_tail) return fail(CSTARK_ERR_INVALID_ARG, "no Schnorr witness uploaded");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(cs::launch_schnorr_trace(c->wit, d_trace, c->stream));
    return CSTARK_OK;
}
int cstark_schnorr_aux_columns(cstark_ctx *c, uint64_t *d_out) {
    if (!c || !d_out) return fail(CSTARK_ERR_INVALID_ARG, "cstark_schnorr_aux_columns: null argument");
    if (!c->wit_buf || c->wit.n_tx == 0 || !c->wit.msg_tail) return fail(CSTARK_ERR_INVALID_ARG, "no Schnorr witness uploaded");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(cs::launch_schnorr_aux_columns(c->wit, d_out, c->stream));
    return CSTARK_OK;
}
int cstark_schnorr_mask_columns(uint64_t *out /* [36][512] host */) {
    if (!out) return fail(CSTARK_ERR_INVALID_ARG, "null argument");
    std::vector<uint64_t> cols;
    cs::host::schnorr_mask_columns(cols);
    memcpy(out, cols.data(), cols.size() * 8);
    return CSTARK_OK;
}
// SchnorrAir's 36 periodic columns over the LDE domain, [b][36][512]; built once per (trace length, blowup)
static int schnorr_periodic(cstark_ctx *c, uint32_t log_n, uint32_t log_blowup, const PeriodicTable **out) {
    const PeriodicTable *pt = nullptr;
    for (const PeriodicTable &t : c->small_periodic)
        if (t.air == CSTARK_AIR_SCHNORR && t.log_n == log_n && t.log_b == log_blowup) pt = &t;
    if (!pt) {
        std::vector<uint64_t> cols;
        cs::host::schnorr_mask_columns(cols);
        const size_t n = (size_t)1 << log_n, b = (size_t)1 << log_blowup;
        PeriodicTable t{0, log_n, log_blowup, nullptr, nullptr, nullptr, CSTARK_AIR_SCHNORR};
        uint64_t *d_cols = nullptr, *d_poly = nullptr;
        HIP_TRY(hipMalloc((void **)&d_cols, cols.size() * 8));
        HIP_TRY(hipMalloc((void **)&d_poly, cols.size() * 8));
        HIP_TRY(hipMalloc((void **)&t.tab, b * cols.size() * 8));
        HIP_TRY(hipMemcpyAsync(d_cols, cols.data(), cols.size() * 8, hipMemcpyHostToDevice, c->stream));
        RC_TRY(interpolate_impl(c, d_cols, d_poly, 36, 9));
        RC_TRY(lde_impl(c, d_poly, t.tab, 36, 0, 36, 9, log_blowup, cs::host::pow(cs::host::lde_offset(), n / 512), 0, (uint32_t)b));
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(d_cols));
        HIP_TRY(hipFree(d_poly));
        c->small_periodic.push_back(t);
        pt = &c->small_periodic.back();
    }
    *out = pt;
    return CSTARK_OK;
}
int cstark_schnorr_evaluate_transitions(cstark_ctx *c, const uint64_t *d_lde, const uint64_t *d_aux_lde, uint64_t *d_out, uint32_t log_n,
                                        uint32_t log_blowup, uint32_t k0, uint32_t nk) {
    if (!c || !d_lde || !d_aux_lde || !d_out || nk == 0) return fail(CSTARK_ERR_INVALID_ARG, "cstark_schnorr_evaluate_transitions: bad argument");
    if (log_n < 9 || log_n > cs::NTT_MAX_LOG_N || log_blowup > 6 || (uint64_t)k0 + nk > (1ull << log_blowup)) return fail(CSTARK_ERR_INVALID_ARG, "bad domain parameters");
    HIP_TRY(hipSetDevice(c->device));
    const PeriodicTable *pt;
    RC_TRY(schnorr_periodic(c, log_n, log_blowup, &pt));
    HIP_TRY(cs::launch_eval_transitions_schnorr(d_lde, d_aux_lde, pt->tab, d_out, log_n, k0, nk, c->stream));
    return CSTARK_OK;
}

int cstark_air_shape(int air, uint32_t n_items, uint32_t *width, uint32_t *n_constraints, uint32_t *n_assertions, uint32_t *log_ce_blowup) {
    cs::host::AirShape s;
    if (!cs::host::air_shape(air, s, n_items) || !width || !n_constraints || !n_assertions || !log_ce_blowup) return fail(CSTARK_ERR_UNSUPPORTED, "AIR not available through the generic entry points");
    *width = s.width; *n_constraints = s.n_constraints; *n_assertions = (uint32_t)s.a_reg.size(); *log_ce_blowup = s.log_ce_blowup();
    return CSTARK_OK;
}
int cstark_air_constraint_degree(int air, uint32_t n_items, uint32_t i, uint32_t *base, uint32_t *cycles) {
    cs::host::AirShape s;
    if (!cs::host::air_shape(air, s, n_items) || i >= s.n_constraints || !base || !cycles) return fail(CSTARK_ERR_INVALID_ARG, "bad AIR / constraint index");
    *base = s.base[i]; *cycles = s.cycles[i];
    return CSTARK_OK;
}
int cstark_merkle_periodic_columns(uint32_t merkle_depth, uint64_t *out /* [33][512] host */) {
    std::vector<uint64_t> cols;
    if (!out || !cs::host::merkle_periodic_columns(merkle_depth, cols)) return fail(CSTARK_ERR_INVALID_ARG, "unsupported Merkle depth");
    memcpy(out, cols.data(), cols.size() * 8);
    return CSTARK_OK;
}

static int merkle_periodic(cstark_ctx *c, uint32_t merkle_depth, uint32_t log_n, uint32_t log_blowup, const PeriodicTable **out);
// RescueAir's 29 periodic columns (cycle 8) over the LDE domain, [b][29][8]: a column of period 8 is a polynomial of degree < 8 in
// x^(n/8); 8 x 8 x 29 x b values, computed on the host (the transform kernels start at 64 points)
static int rescue_periodic(cstark_ctx *c, uint32_t log_n, uint32_t log_blowup, const PeriodicTable **out) {
    const int air = CSTARK_AIR_RESCUE_CHAIN;
    for (const PeriodicTable &t : c->small_periodic)
        if (t.air == air && t.log_n == log_n && t.log_b == log_blowup) { *out = &t; return CSTARK_OK; }
    using namespace cs::host;
    std::vector<uint64_t> cols;
    rescue_chain_periodic_columns(cols);
    const size_t n = (size_t)1 << log_n, b = (size_t)1 << log_blowup;
    for (int cidx = 0; cidx < 29; cidx++) intt_small(cols.data() + (size_t)cidx * 8, 3);
    std::vector<uint64_t> tab(b * 29 * 8);
    const uint64_t wbn = root_of_unity(log_n + log_blowup), w8 = root_of_unity(3);
    uint64_t shift = lde_offset();
    for (size_t k = 0; k < b; k++) {
        uint64_t x = pow(shift, n / 8); // point m of coset k in the variable x^(n/8): shift^(n/8) w_8^m
        for (int m = 0; m < 8; m++) {
            for (int cidx = 0; cidx < 29; cidx++) {
                const uint64_t *co = cols.data() + (size_t)cidx * 8;
                uint64_t v = 0;
                for (int d = 7; d >= 0; d--) v = add(mul(v, x), co[d]);
                tab[(k * 29 + cidx) * 8 + m] = v;
            }
            x = mul(x, w8);
        }
        shift = mul(shift, wbn);
    }
    PeriodicTable t{0, log_n, log_blowup, nullptr, nullptr, nullptr, air};
    HIP_TRY(hipMalloc((void **)&t.tab, tab.size() * 8));
    HIP_TRY(hipMemcpyAsync(t.tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->small_periodic.push_back(t);
    *out = &c->small_periodic.back();
    return CSTARK_OK;
}
int cstark_rescue_chain_periodic_columns(uint64_t *out /* [29][8] host */) {
    if (!out) return fail(CSTARK_ERR_INVALID_ARG, "null argument");
    std::vector<uint64_t> cols;
    cs::host::rescue_chain_periodic_columns(cols);
    memcpy(out, cols.data(), cols.size() * 8);
    return CSTARK_OK;
}
// RescueProver::build_trace (benches/rescue.rs:277-322): 14 x 8 * chain_length
int cstark_rescue_chain_build_trace(cstark_ctx *c, const uint64_t seed[7], uint32_t chain_length, uint64_t *d_trace) {
    if (!c || !seed || !d_trace) return fail(CSTARK_ERR_INVALID_ARG, "cstark_rescue_chain_build_trace: null argument");
    if (chain_length < 8 || (chain_length & (chain_length - 1)) || chain_length > (1u << 21)) // benches/rescue.rs:34-37: a power of two
        return fail(CSTARK_ERR_INVALID_ARG, "chain length must be a power of two, 8 .. 2^21 (64 .. 2^24 trace rows)");
    for (int i = 0; i < 7; i++) if (seed[i] >= cs::host::P) return fail(CSTARK_ERR_INVALID_ARG, "seed is not a field element");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(cs::launch_rescue_chain_trace(seed, chain_length, d_trace, c->stream));
    return CSTARK_OK;
}
int cstark_air_evaluate_transitions(cstark_ctx *c, int air, const uint64_t *d_lde, uint64_t *d_out, uint32_t merkle_depth, uint32_t log_n,
                                    uint32_t log_blowup, uint32_t k0, uint32_t nk) {
    if (!c || !d_lde || !d_out || nk == 0) return fail(CSTARK_ERR_INVALID_ARG, "cstark_air_evaluate_transitions: bad argument");
    if (log_n < cs::NTT_MIN_LOG_N || log_n > cs::NTT_MAX_LOG_N || log_blowup > 6 || (uint64_t)k0 + nk > (1ull << log_blowup))
        return fail(CSTARK_ERR_INVALID_ARG, "bad domain parameters");
    HIP_TRY(hipSetDevice(c->device));
    if (air == CSTARK_AIR_RANGE) {
        HIP_TRY(cs::launch_eval_transitions_range(d_lde, d_out, log_n, nk, c->stream));
        return CSTARK_OK;
    }
    if (air == CSTARK_AIR_RESCUE_CHAIN) {
        const PeriodicTable *pt;
        RC_TRY(rescue_periodic(c, log_n, log_blowup, &pt));
        HIP_TRY(cs::launch_eval_transitions_rescue(d_lde, pt->tab, d_out, log_n, k0, nk, c->stream));
        return CSTARK_OK;
    }
    if (air != CSTARK_AIR_MERKLE_UPDATE) return fail(CSTARK_ERR_UNSUPPORTED, "AIR not available through the generic entry points");
    if (log_n < 9) return fail(CSTARK_ERR_INVALID_ARG, "the trace must hold at least one 512-row transaction");
    const PeriodicTable *pt;
    RC_TRY(merkle_periodic(c, merkle_depth, log_n, log_blowup, &pt));
    HIP_TRY(cs::launch_eval_transitions_merkle(d_lde, pt->tab, d_out, log_n, k0, nk, c->stream));
    return CSTARK_OK;
}
// MerkleAir's 33 periodic columns over the LDE domain, [b][33][512]; built once per (depth, trace length, blowup)
static int merkle_periodic(cstark_ctx *c, uint32_t merkle_depth, uint32_t log_n, uint32_t log_blowup, const PeriodicTable **out) {
    const int air = CSTARK_AIR_MERKLE_UPDATE;
    const PeriodicTable *pt = nullptr;
    for (const PeriodicTable &t : c->small_periodic)
        if (t.air == air && t.depth == merkle_depth && t.log_n == log_n && t.log_b == log_blowup) pt = &t;
    if (!pt) {
        std::vector<uint64_t> cols;
        if (!cs::host::merkle_periodic_columns(merkle_depth, cols)) return fail(CSTARK_ERR_INVALID_ARG, "unsupported Merkle depth");
        const size_t n = (size_t)1 << log_n, b = (size_t)1 << log_blowup;
        PeriodicTable t{merkle_depth, log_n, log_blowup, nullptr, nullptr, nullptr, air};
        uint64_t *d_cols = nullptr, *d_poly = nullptr;
        HIP_TRY(hipMalloc((void **)&d_cols, cols.size() * 8));
        HIP_TRY(hipMalloc((void **)&d_poly, cols.size() * 8));
        HIP_TRY(hipMalloc((void **)&t.tab, b * cols.size() * 8));
        HIP_TRY(hipMemcpyAsync(d_cols, cols.data(), cols.size() * 8, hipMemcpyHostToDevice, c->stream));
        RC_TRY(interpolate_impl(c, d_cols, d_poly, 33, 9));
        RC_TRY(lde_impl(c, d_poly, t.tab, 33, 0, 33, 9, log_blowup, cs::host::pow(cs::host::lde_offset(), n / 512), 0, (uint32_t)b));
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(d_cols));
        HIP_TRY(hipFree(d_poly));
        c->small_periodic.push_back(t);
        pt = &c->small_periodic.back();
    }
    *out = pt;
    return CSTARK_OK;
}

// coefficient columns [12][n] of the value polynomials of SchnorrAir's sequence assertions (R.x at step 0 and at step 511 of
// every 512-row block, src/schnorr/air.rs:172-224); the caller extends them with cstark_lde_columns
int cstark_schnorr_assertion_polys(cstark_ctx *c, uint64_t *d_out, uint32_t log_n) {
    if (!c || !d_out) return fail(CSTARK_ERR_INVALID_ARG, "cstark_schnorr_assertion_polys: null argument");
    const size_t n = (size_t)1 << log_n, m = c->schnorr_rx.size() / 6;
    if (m == 0 || m * 512 != n) return fail(CSTARK_ERR_INVALID_ARG, "no Schnorr witness uploaded for this trace length");
    unsigned log_m = 0;
    while (((size_t)1 << log_m) < m) log_m++;
    // Only the first m = n / 512 coefficients of a column are non-zero (the polynomials have degree < m): they are staged in a
    // buffer of the context and copied as 12 rows of m words into the zero-filled columns -- 48 KB at 2^18 rows.  (Round 3 built
    // the 12 n words on the host and uploaded all of them, 25 MB of zeros per proof through the runtime's staging chunks: 0.6 ms of
    // copies on the stream that the interpolation of registers 37..55 waits on, and a host synchronisation.)
    std::vector<uint64_t> &cols = c->schnorr_av_stage;
    cols.assign(12 * m, 0);
    const uint64_t winv = cs::host::inv(cs::host::root_of_unity(log_n));
    for (int half = 0; half < 2; half++) {
        const uint64_t off = cs::host::pow(winv, half ? 511 : 0); // c(x) = P(x * w^-first_step)
        for (int k = 0; k < 6; k++) {
            uint64_t *o = cols.data() + (size_t)(6 * half + k) * m;
            for (size_t t = 0; t < m; t++) o[t] = c->schnorr_rx[6 * t + k];
            if (m > 1) cs::host::intt_small(o, log_m);
            uint64_t sc = cs::host::ONE;
            for (size_t t = 0; t < m; t++) { o[t] = cs::host::mul(o[t], sc); sc = cs::host::mul(sc, off); }
        }
    }
    HIP_TRY(hipMemsetAsync(d_out, 0, 12 * n * 8, c->stream));
    HIP_TRY(hipMemcpy2DAsync(d_out, n * 8, cols.data(), m * 8, m * 8, 12, hipMemcpyHostToDevice, c->stream));
    return CSTARK_OK;
}

// d_schnorr_aux_lde != null (SchnorrAir only): the transition sum comes from the fused evaluator instead of d_evals
static int air_combine_impl(cstark_ctx *c, int air, uint32_t n_items, const uint64_t *d_lde, const uint64_t *d_evals, const uint64_t *t_alpha,
                            const uint64_t *t_beta, const uint64_t *b_alpha, const uint64_t *b_beta, const uint64_t *assertion_values,
                            const uint64_t *d_avals_lde, uint32_t n_avals, uint64_t *d_out, uint32_t log_n, uint32_t log_blowup, uint32_t k0, uint32_t nk,
                            const uint64_t *d_schnorr_aux_lde, int merkle_depth_fused = -1, bool schnorr_input_is_lde = false,
                            const uint64_t *d_coefs = nullptr, const uint64_t *d_avalues = nullptr) {
    // d_coefs != null (the device-side channel of prove.hip): the coefficients are already on the device, in the cstark_tx_coeffs-like
    // block alpha[115] | beta[115] | b_alpha[na] | b_beta[na] (what channel.hip draws with stride 115); d_avalues: the assertion values on
    // the device (null: the AIR's built-in constants)
    const bool fused_merkle = merkle_depth_fused >= 0;
    if (!c || !d_lde || (!d_evals && !d_schnorr_aux_lde && !fused_merkle) || (!d_coefs && (!t_alpha || !t_beta || !b_alpha || !b_beta)) || !d_out || nk == 0)
        return fail(CSTARK_ERR_INVALID_ARG, "cstark_air_combine: null argument");
    if (d_schnorr_aux_lde && (air != CSTARK_AIR_SCHNORR || log_n < 9)) return fail(CSTARK_ERR_INVALID_ARG, "the fused evaluator is SchnorrAir's");
    if (fused_merkle && (air != CSTARK_AIR_MERKLE_UPDATE || log_n < 9)) return fail(CSTARK_ERR_INVALID_ARG, "the fused evaluator is MerkleAir's");
    cs::host::AirShape s;
    if (!cs::host::air_shape(air, s, n_items)) return fail(CSTARK_ERR_UNSUPPORTED, "AIR not available through the generic entry points");
    if (s.a_const.empty() && !assertion_values && !d_avalues) return fail(CSTARK_ERR_INVALID_ARG, "cstark_air_combine: assertion values required");
    if (d_coefs && s.n_constraints > 115) return fail(CSTARK_ERR_UNSUPPORTED, "cstark_air_combine: device coefficient block holds at most 115 constraints");
    const uint32_t log_ce = s.log_ce_blowup();
    if (log_blowup < log_ce || log_blowup > 6 || (uint64_t)k0 + nk > (1ull << log_blowup)) return fail(CSTARK_ERR_INVALID_ARG, "blowup factor below the constraint degree");
    if (log_n < cs::NTT_MIN_LOG_N || log_n > cs::NTT_MAX_LOG_N) return fail(CSTARK_ERR_INVALID_ARG, "bad trace length");
    const uint64_t n = 1ull << log_n, ce = n << log_ce, b = 1ull << log_blowup;
    const size_t nc = s.n_constraints, na = s.a_reg.size();
    bool needs_avals = false;
    for (int32_t q : s.a_seq) needs_avals |= q >= 0;
    if (needs_avals && (!d_avals_lde || n_avals == 0)) return fail(CSTARK_ERR_INVALID_ARG, "this AIR has sequence assertions: pass the extended value polynomials");
    if (b > 8) return fail(CSTARK_ERR_UNSUPPORTED, "cstark_air_combine: blowup factor at most 8");
    HIP_TRY(hipSetDevice(c->device));
    const NttPlan *plan;
    RC_TRY(get_plan(c, log_n, &plan));
    // ---- everything that depends on (AIR, trace length, blowup) only: degree / divisor groups, the powers of the coset offsets, the
    // assertion tables on the device, the cached divisor inverses -- built once per context and key (round 3 rebuilt it per call: ~150
    // modular exponentiations and a blocking upload between the trace root and the evaluation launches of EVERY sub-AIR proof)
    cs::AirCombineStatic *st = nullptr;
    for (cs::AirCombineStatic &q : c->air_static)
        if (q.air == air && q.n_items == n_items && q.log_n == log_n && q.log_b == log_blowup) st = &q;
    if (!st) {
        cs::AirCombineStatic q{};
        q.air = air; q.n_items = n_items; q.log_n = log_n; q.log_b = log_blowup;
        cs::AirCombineParams &p = q.p;
        const uint64_t wn = cs::host::root_of_unity(log_n);
        q.t_grp.resize(nc);
        std::vector<uint32_t> a_grp(na);
        for (size_t i = 0; i < nc; i++) { // distinct degree adjustments
            const uint64_t adj = CSTARK_CONV_TRANSITION_ADJUSTMENT(ce, n, s.eval_degree(i, n));
            uint32_t g = 0;
            while (g < p.n_tgrp && p.tgrp_adj[g] != adj) g++;
            if (g == p.n_tgrp) {
                if (g == cs::AIR_MAX_GROUPS) return fail(CSTARK_ERR_UNSUPPORTED, "too many distinct constraint degrees");
                p.tgrp_adj[p.n_tgrp++] = adj;
            }
            q.t_grp[i] = g;
        }
        for (size_t a = 0; a < na; a++) { // distinct assertion divisors x^m - w^(first m)
            const uint64_t first = s.a_stride.empty() ? (s.a_last[a] ? n - 1 : 0) : s.a_first[a];
            const uint64_t m = (!s.a_stride.empty() && s.a_stride[a]) ? n / s.a_stride[a] : 1;
            const uint64_t zc = cs::host::pow(wn, (first * m) % n);
            uint32_t g = 0;
            while (g < p.n_agrp && !(p.agrp_m[g] == m && p.agrp_zc[g] == zc)) g++;
            if (g == p.n_agrp) {
                if (g == cs::AIR_MAX_GROUPS) return fail(CSTARK_ERR_UNSUPPORTED, "too many distinct assertion divisors");
                p.agrp_m[g] = m; p.agrp_zc[g] = zc; p.agrp_badj[g] = CSTARK_CONV_BOUNDARY_ADJUSTMENT(ce, n, m);
                p.n_agrp++;
            }
            a_grp[a] = g;
        }
        // device block of the static part: shifts[b] | built-in assertion constants[na] (u64), then u32: a_reg | a_seq | a_grp | t_grp
        std::vector<uint64_t> blk(b + na + (3 * na + nc + 1) / 2 + 1);
        const uint64_t wbn = cs::host::root_of_unity(log_n + log_blowup);
        {
            uint64_t shift = cs::host::lde_offset();
            for (uint64_t k = 0; k < b; k++) { blk[k] = shift; shift = cs::host::mul(shift, wbn); }
        }
        for (size_t a = 0; a < na; a++) blk[b + a] = s.a_const.empty() ? 0 : s.a_const[a];
        uint32_t *q32 = (uint32_t *)(blk.data() + b + na);
        for (size_t a = 0; a < na; a++) q32[a] = s.a_reg[a];
        for (size_t a = 0; a < na; a++) ((int32_t *)q32)[na + a] = s.a_seq.empty() ? -1 : s.a_seq[a];
        for (size_t a = 0; a < na; a++) q32[2 * na + a] = a_grp[a];
        for (size_t i = 0; i < nc; i++) q32[3 * na + i] = q.t_grp[i];
        HIP_TRY(hipMalloc((void **)&q.d_static, blk.size() * 8));
        HIP_TRY(hipMemcpy(q.d_static, blk.data(), blk.size() * 8, hipMemcpyHostToDevice));
        p.shifts = q.d_static;
        p.a_reg = (const uint32_t *)(q.d_static + b + na); p.a_seq = (const int32_t *)(p.a_reg + na);
        p.a_grp = p.a_reg + 2 * na; p.t_grp = p.a_reg + 3 * na;
        p.w = plan->w;
        p.w_last = cs::host::inv(wn);
        p.width = s.width; p.n_constraints = (uint32_t)nc; p.n_assertions = (uint32_t)na;
        p.stride = 1u << (log_blowup - log_ce); p.log_n = log_n;
        {   // per-coset powers of the coset shift (the kernel completes them with a twiddle-table product per point)
            uint64_t sh = cs::host::lde_offset();
            for (uint64_t k = 0; k < b; k++) {
                for (uint32_t g = 0; g < p.n_tgrp; g++) p.tgrp_shift[k][g] = cs::host::pow(sh, p.tgrp_adj[g]);
                for (uint32_t g = 0; g < p.n_agrp; g++) {
                    p.agrp_bshift[k][g] = cs::host::pow(sh, p.agrp_badj[g]);
                    p.agrp_mshift[k][g] = cs::host::pow(sh, p.agrp_m[g]);
                }
                p.zinv_coset[k] = cs::host::inv(cs::host::sub(cs::host::pow(sh, n), cs::host::ONE));
                sh = cs::host::mul(sh, wbn);
            }
        }
        {   // 1 / (x^m - zc) of every assertion divisor over the domain: cached per (m, zc, trace length, blowup).
            // CSTARK_AIR_INV_TABLES=0 (tuning / debugging): one inversion per point inside k_air_combine
            static const bool inv_tables = [] { const char *e = getenv("CSTARK_AIR_INV_TABLES"); return !e || atoi(e) != 0; }();
            for (uint32_t g = 0; inv_tables && g < p.n_agrp; g++) {
                const cs::AssertInverseTable *t = nullptr;
                for (const cs::AssertInverseTable &qq : c->assert_inv)
                    if (qq.log_n == log_n && qq.log_b == log_blowup && qq.m == p.agrp_m[g] && qq.zc == p.agrp_zc[g]) t = &qq;
                if (!t) {
                    cs::AssertInverseTable qq{log_n, log_blowup, p.agrp_m[g], p.agrp_zc[g], nullptr};
                    HIP_TRY(hipMalloc((void **)&qq.tab, (size_t)b * (n / qq.m) * 8));
                    uint64_t sm[8] = {};
                    for (uint64_t k = 0; k < b; k++) sm[k] = p.agrp_mshift[k][g];
                    const hipError_t e = cs::launch_assert_inverses(qq.tab, plan->w, sm, (unsigned)b, qq.m, qq.zc, log_n, c->stream);
                    if (e != hipSuccess) { (void)hipFree(qq.tab); HIP_TRY(e); }
                    c->assert_inv.push_back(qq);
                    t = &c->assert_inv.back();
                }
                p.agrp_inv[g] = t->tab;
            }
        }
        c->air_static.push_back(std::move(q));
        st = &c->air_static.back();
    }
    const std::vector<uint32_t> &t_grp = st->t_grp;
    // ---- per proof: coefficients and assertion values through a pinned staging block into the context's device block, no wait:
    // t_alpha | t_beta | b_alpha | b_beta | a_value | the transition coefficients once more in the cstark_tx_coeffs layout (SchnorrAir's
    // split evaluation reads alpha[i] at word i, beta[i] at word 115 + i) | device scratch of k_merkle_rounds
    constexpr size_t TXL = 230;
    const size_t words = 2 * nc + 3 * na + TXL, total = words + cs::MERKLE_RTAB_WORDS;
    const size_t txl_off = 2 * nc + 3 * na, mrt_off = words;
    if (c->air_coef_words < total) {
        if (d_coefs && c->air_coef_words) HIP_TRY(hipEventSynchronize(c->air_coef_ev));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->air_coef_buf) { HIP_TRY(hipFree(c->air_coef_buf)); c->air_coef_buf = nullptr; }
        if (c->air_coef_stage) { HIP_TRY(hipHostFree(c->air_coef_stage)); c->air_coef_stage = nullptr; }
        c->air_coef_words = 0;
        HIP_TRY(hipMalloc((void **)&c->air_coef_buf, total * 8));
        HIP_TRY(hipHostMalloc((void **)&c->air_coef_stage, total * 8, hipHostMallocDefault));
        if (!c->air_coef_ev) HIP_TRY(hipEventCreateWithFlags(&c->air_coef_ev, hipEventDisableTiming));
        c->air_coef_words = total;
    } else if (!d_coefs) {
        HIP_TRY(hipEventSynchronize(c->air_coef_ev)); // the previous upload has left the staging block (long ago, normally)
    }
    if (!d_coefs) {
        uint64_t *q = c->air_coef_stage;
        memcpy(q, t_alpha, nc * 8); q += nc;
        memcpy(q, t_beta, nc * 8); q += nc;
        memcpy(q, b_alpha, na * 8); q += na;
        memcpy(q, b_beta, na * 8); q += na;
        for (size_t a = 0; a < na; a++) *q++ = s.a_const.empty() ? assertion_values[a] : s.a_const[a];
        memset(q, 0, TXL * 8);
        if (nc <= 115) { memcpy(q, t_alpha, nc * 8); memcpy(q + 115, t_beta, nc * 8); }
        HIP_TRY(hipMemcpyAsync(c->air_coef_buf, c->air_coef_stage, words * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipEventRecord(c->air_coef_ev, c->stream));
    }
    const uint64_t *d = c->air_coef_buf;
    cs::AirCombineParams p = st->p;
    p.lde = d_lde; p.evals = d_evals; p.out = d_out; p.avals = d_avals_lde; p.n_avals = n_avals; p.k0 = k0;
    if (d_coefs) {
        p.t_alpha = d_coefs; p.t_beta = d_coefs + 115; p.b_alpha = d_coefs + 230; p.b_beta = d_coefs + 230 + na;
        p.a_value = d_avalues ? d_avalues : st->d_static + b;
    } else {
        p.t_alpha = d; p.t_beta = d + nc;
        p.b_alpha = d + 2 * nc; p.b_beta = d + 2 * nc + na; p.a_value = d + 2 * nc + 2 * na;
    }
    const uint64_t *d_txl = d_coefs ? d_coefs : d + txl_off; // the transition coefficients in the cstark_tx_coeffs layout
    if (d_schnorr_aux_lde) {
        const PeriodicTable *pt;
        RC_TRY(schnorr_periodic(c, log_n, log_blowup, &pt));
        // The doubling / addition gadgets in the degree-split form of the TransactionAir evaluator when the whole extension is at hand
        // (every coset, blowup 8, at least eight signatures): their sums have degree < 4n without the periodic flags, so they are
        // evaluated on the four even cosets, interpolated, extended to the odd cosets and recombined (constraints.hip,
        // k_schnorr_ec_split).  The table must be a genuine extension -- it is the prover's own; the stage entry point passes any table,
        // so it takes this path only under CSTARK_SCHNORR_SPLIT_STAGE=1 (tests).  CSTARK_SCHNORR_SPLIT=0: every point directly.
        static const bool split_env = [] { const char *e = getenv("CSTARK_SCHNORR_SPLIT"); return !e || atoi(e) != 0; }();
        if (split_env && schnorr_input_is_lde && k0 == 0 && nk == 8 && log_blowup == 3 && log_n >= 12 && log_n + 3 <= cs::NTT_MAX_LOG_N && n_items > 1) {
            // the final addition (degree 5 (n - 1) without its flag) on five cosets: three more tables on the even cosets, LDE coset 1
            // directly (CSTARK_SCHNORR_FINAL5=0: on all eight cosets through the frame evaluator)
            static const bool final5 = [] { const char *e = getenv("CSTARK_SCHNORR_FINAL5"); return !e || atoi(e) != 0; }();
            const unsigned T = final5 ? cs::SCHNORR_SPLIT_TABLES : cs::SCHNORR_SPLIT_EC_TABLES;
            const NttPlan *pn, *p4, *p8;
            const CosetTable *t1;
            RC_TRY(get_plan(c, log_n, &pn));
            RC_TRY(get_plan(c, log_n + 2, &p4));
            RC_TRY(get_plan(c, log_n + 3, &p8));
            RC_TRY(get_coset_table(c, log_n, 3, cs::host::from_u64(1), &t1));
            const size_t region = (size_t)T * 4 * n, hcol = (size_t)3 * n; // hcol: the final addition's three sums, one n-point table each
            RC_TRY(ensure_ws(c, (5 * region + 9 * hcol) * 8));
            uint64_t *even = (uint64_t *)c->ws, *sa = even + region, *sb = sa + region, *sc = sb + region, *odd = sc + region;
            uint64_t *fin_direct = odd + region, *fin_hi = fin_direct + hcol /* [4 odd cosets][3][n] */, *fin_co = fin_hi + 4 * hcol, *fin_scr = fin_co + hcol /* [3] */;
            HIP_TRY(cs::launch_schnorr_ec_split(p, d_schnorr_aux_lde, d_txl, even, c->stream));
            if (final5) HIP_TRY(cs::launch_schnorr_final_split(p, d_txl, even + (size_t)cs::SCHNORR_SPLIT_EC_TABLES * 4 * n, -1, c->stream));
            cs::NttArgs a{};
            a.in = even; a.scratch = sa; a.out = sb; a.width = 4 * T; a.batch = 1; a.log_n = log_n;
            a.w = pn->winv; a.post_scale = pn->n_inv; a.do_scale = true; a.inverse = true; a.aux = pn->aux_winv;
            HIP_TRY(cs::ntt_columns(a, c->stream));
            HIP_TRY(cs::coset_even_to_odd(sb, sa, log_n, T, p4->winv, p8->w, cs::host::inv(cs::host::from_u64(4)), c->stream));
            cs::NttArgs f{};
            f.in = sa; f.scratch = sc; f.out = odd; f.width = T; f.batch = 4; f.log_n = log_n;
            f.w = pn->w; f.prescale = t1->s + n; f.prescale_batch_stride = 2 * n; f.do_scale = false; f.inverse = false;
            f.aux = pn->aux_w; f.aux_ps = t1->aux ? t1->aux + t1->aux_words : nullptr; f.aux_ps_batch_stride = 2 * t1->aux_words;
            f.in_batch_stride = (size_t)T * n; f.scratch_batch_stride = (size_t)T * n; f.out_batch_stride = (size_t)T * n;
            HIP_TRY(cs::ntt_columns(f, c->stream));
            if (final5) { // H = (T - F) / 2 on LDE coset 1, interpolated there and extended to cosets 3, 5, 7 (as for TransactionAir)
                HIP_TRY(cs::launch_schnorr_final_split(p, d_txl, fin_direct, 1, c->stream));
                HIP_TRY(cs::launch_schnorr_final_hi(p, odd, fin_direct, fin_hi, cs::host::inv(cs::host::from_u64(2)), c->stream));
                cs::NttArgs hi_inv{};
                hi_inv.in = fin_hi; hi_inv.scratch = fin_scr; hi_inv.out = fin_co; hi_inv.width = 3; hi_inv.batch = 1; hi_inv.log_n = log_n;
                hi_inv.w = pn->winv; hi_inv.post_scale = pn->n_inv; hi_inv.do_scale = true; hi_inv.inverse = true; hi_inv.aux = pn->aux_winv;
                HIP_TRY(cs::ntt_columns(hi_inv, c->stream));
                cs::NttArgs hi_fwd{}; // coefficients of H(w_8n z) in z -> coset k: prescale by (w_8n^(k-1))^s, k - 1 = 2, 4, 6: rows 2, 4, 6 of the offset-1 table
                hi_fwd.in = fin_co; hi_fwd.scratch = fin_scr; hi_fwd.out = fin_hi + hcol; hi_fwd.width = 3; hi_fwd.batch = 3; hi_fwd.log_n = log_n;
                hi_fwd.w = pn->w; hi_fwd.prescale = t1->s + 2 * n; hi_fwd.prescale_batch_stride = 2 * n; hi_fwd.do_scale = false; hi_fwd.inverse = false;
                hi_fwd.aux = pn->aux_w; hi_fwd.aux_ps = t1->aux ? t1->aux + 2 * t1->aux_words : nullptr; hi_fwd.aux_ps_batch_stride = 2 * t1->aux_words;
                hi_fwd.in_batch_stride = 0; hi_fwd.scratch_batch_stride = hcol; hi_fwd.out_batch_stride = hcol;
                HIP_TRY(cs::ntt_columns(hi_fwd, c->stream));
            }
            // the round gadget of the message hash in the folded form (CSTARK_SCHNORR_ROUNDS=0: inside the frame evaluator)
            static const bool rounds_env = [] { const char *e = getenv("CSTARK_SCHNORR_ROUNDS"); return !e || atoi(e) != 0; }();
            HIP_TRY(cs::launch_schnorr_split_finish(p, d_schnorr_aux_lde, pt->tab, even, odd, t_grp[0], t_grp[6], c->stream,
                                                    rounds_env ? c->air_coef_buf + mrt_off : nullptr, t_grp[42], final5 ? fin_hi : nullptr));
        } else {
            HIP_TRY(cs::launch_schnorr_fused(p, d_schnorr_aux_lde, pt->tab, nk, c->stream));
        }
        p.tsum = d_out;
    }
    if (fused_merkle) {
        const PeriodicTable *pt;
        RC_TRY(merkle_periodic(c, (uint32_t)merkle_depth_fused, log_n, log_blowup, &pt));
        // the round gadgets in the folded form of the TransactionAir evaluator (CSTARK_MERKLE_ROUNDS=0: the generic frame evaluator)
        static const bool rounds_env = [] { const char *e = getenv("CSTARK_MERKLE_ROUNDS"); return !e || atoi(e) != 0; }();
        const bool folded = rounds_env && b <= 8 && log_n >= 9;
        HIP_TRY(cs::launch_merkle_fused(p, pt->tab, nk, c->stream, folded ? c->air_coef_buf + mrt_off : nullptr, t_grp[0]));
        p.tsum = d_out;
    }
    HIP_TRY(cs::launch_air_combine(p, nk, c->stream));
    return CSTARK_OK;
}
int cstark_air_combine(cstark_ctx *c, int air, uint32_t n_items, const uint64_t *d_lde, const uint64_t *d_evals, const uint64_t *t_alpha,
                       const uint64_t *t_beta, const uint64_t *b_alpha, const uint64_t *b_beta, const uint64_t *assertion_values,
                       const uint64_t *d_avals_lde, uint32_t n_avals, uint64_t *d_out, uint32_t log_n, uint32_t log_blowup, uint32_t k0, uint32_t nk) {
    if (!d_evals) return fail(CSTARK_ERR_INVALID_ARG, "cstark_air_combine: null argument");
    return air_combine_impl(c, air, n_items, d_lde, d_evals, t_alpha, t_beta, b_alpha, b_beta, assertion_values, d_avals_lde, n_avals, d_out, log_n,
                            log_blowup, k0, nk, nullptr);
}
int cstark_merkle_evaluate_constraints(cstark_ctx *c, uint32_t merkle_depth, const uint64_t *d_lde, const uint64_t *t_alpha, const uint64_t *t_beta,
                                       const uint64_t *b_alpha, const uint64_t *b_beta, const uint64_t *assertion_values, uint64_t *d_out, uint32_t log_n,
                                       uint32_t log_blowup, uint32_t k0, uint32_t nk) {
    if (merkle_depth > 31) return fail(CSTARK_ERR_INVALID_ARG, "cstark_merkle_evaluate_constraints: unsupported Merkle depth");
    return air_combine_impl(c, CSTARK_AIR_MERKLE_UPDATE, 0, d_lde, nullptr, t_alpha, t_beta, b_alpha, b_beta, assertion_values, nullptr, 0, d_out, log_n,
                            log_blowup, k0, nk, nullptr, (int)merkle_depth);
}
int cstark_schnorr_evaluate_constraints(cstark_ctx *c, uint32_t n_sig, const uint64_t *d_lde, const uint64_t *d_aux_lde, const uint64_t *t_alpha,
                                        const uint64_t *t_beta, const uint64_t *b_alpha, const uint64_t *b_beta, const uint64_t *d_avals_lde,
                                        uint32_t n_avals, uint64_t *d_out, uint32_t log_n, uint32_t log_blowup, uint32_t k0, uint32_t nk) {
    if (!d_aux_lde) return fail(CSTARK_ERR_INVALID_ARG, "cstark_schnorr_evaluate_constraints: null argument");
    return air_combine_impl(c, CSTARK_AIR_SCHNORR, n_sig, d_lde, nullptr, t_alpha, t_beta, b_alpha, b_beta, nullptr, d_avals_lde, n_avals, d_out, log_n,
                            log_blowup, k0, nk, d_aux_lde);
}

int cstark_schnorr_evaluate_constraints_lde(cstark_ctx *c, uint32_t n_sig, const uint64_t *d_lde, const uint64_t *d_aux_lde, const uint64_t *t_alpha,
                                            const uint64_t *t_beta, const uint64_t *b_alpha, const uint64_t *b_beta, const uint64_t *d_avals_lde,
                                            uint32_t n_avals, uint64_t *d_out, uint32_t log_n) {
    if (!d_aux_lde) return fail(CSTARK_ERR_INVALID_ARG, "cstark_schnorr_evaluate_constraints_lde: null argument");
    return air_combine_impl(c, CSTARK_AIR_SCHNORR, n_sig, d_lde, nullptr, t_alpha, t_beta, b_alpha, b_beta, nullptr, d_avals_lde, n_avals, d_out, log_n, 3, 0,
                            8, d_aux_lde, -1, true);
}

} // extern "C"
// internal (ctx.h): the merged evaluations of a sub-AIR with the coefficients already on the device (drawn there by the device-side
// channel): d_coefs = alpha[115] | beta[115] | b_alpha[na] | b_beta[na]; d_avalues = the assertion values (null: built-in constants).
// mode 0: from materialised transition values d_evals; 1: MerkleAir fused (depth); 2: SchnorrAir fused on the prover's own extensions
int air_combine_dev(cstark_ctx *c, int air, uint32_t n_items, int mode, uint32_t merkle_depth, const uint64_t *d_lde, const uint64_t *d_evals,
                    const uint64_t *d_aux_lde, const uint64_t *d_coefs, const uint64_t *d_avalues, const uint64_t *d_avals_lde, uint32_t n_avals, uint64_t *d_out,
                    uint32_t log_n, uint32_t log_blowup, uint32_t nk) {
    if (!d_coefs) return fail(CSTARK_ERR_INVALID_ARG, "air_combine_dev: null argument");
    if (mode == 1) return air_combine_impl(c, air, n_items, d_lde, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, d_out, log_n, log_blowup, 0, nk,
                                           nullptr, (int)merkle_depth, false, d_coefs, d_avalues);
    if (mode == 2) return air_combine_impl(c, air, n_items, d_lde, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d_avals_lde, n_avals, d_out, log_n, log_blowup, 0,
                                           nk, d_aux_lde, -1, true, d_coefs, d_avalues);
    return air_combine_impl(c, air, n_items, d_lde, d_evals, nullptr, nullptr, nullptr, nullptr, nullptr, d_avals_lde, n_avals, d_out, log_n, log_blowup, 0, nk, nullptr,
                            -1, false, d_coefs, d_avalues);
}
extern "C" {

// per-launch timing of the fused constraint evaluation (HIP events on the context's stream)
int cstark_ctx_set_part_timing(cstark_ctx *c, int enable) {
    if (!c) return fail(CSTARK_ERR_INVALID_ARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    if (enable && !c->part_ev[0])
        for (hipEvent_t &e : c->part_ev) HIP_TRY(hipEventCreate(&e));
    c->part_timing = enable != 0;
    c->part_valid = false;
    return CSTARK_OK;
}
int cstark_lde_timing_ms(cstark_ctx *c, float *total_ms, uint64_t *elements) {
    if (!c || !total_ms || !elements) return fail(CSTARK_ERR_INVALID_ARG, "null argument");
    float sum = 0;
    if (c->lde_ev_used) HIP_TRY(hipEventSynchronize(c->lde_ev[c->lde_ev_used - 1]));
    for (size_t i = 0; i + 1 < c->lde_ev_used; i += 2) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->lde_ev[i], c->lde_ev[i + 1]));
        sum += ms;
    }
    *total_ms = sum;
    *elements = c->lde_units;
    c->lde_ev_used = 0;
    c->lde_units = 0;
    return CSTARK_OK;
}
int cstark_tx_constraint_part_ms(cstark_ctx *c, float *ms /* [9] */) {
    if (!c || !ms) return fail(CSTARK_ERR_INVALID_ARG, "null argument");
    if (!c->part_valid) return fail(CSTARK_ERR_INVALID_ARG, "no timed constraint evaluation has run (cstark_ctx_set_part_timing)");
    HIP_TRY(hipEventSynchronize(c->part_ev[cs::CE_NUM_PARTS]));
    for (int i = 0; i < cs::CE_NUM_PARTS; i++) HIP_TRY(hipEventElapsedTime(&ms[i], c->part_ev[i], c->part_ev[i + 1]));
    return CSTARK_OK;
}

// host-side view of the AIR description, for the CPU tests
int cstark_tx_constraint_degree(uint32_t i, uint32_t *base, uint32_t *cycles) {
    if (i >= CSTARK_TX_NUM_CONSTRAINTS || !base || !cycles) return fail(CSTARK_ERR_INVALID_ARG, "constraint index out of range");
    const int g = cs::tx_degree_group((int)i);
    *base = cs::TX_GROUP_BASE[g];
    *cycles = cs::TX_GROUP_CYCLES[g];
    return CSTARK_OK;
}
int cstark_tx_periodic_columns(uint32_t merkle_depth, uint64_t *out /* [48][1024] host */) {
    std::vector<uint64_t> cols;
    if (!out || !cs::host::tx_periodic_columns(merkle_depth, cols)) return fail(CSTARK_ERR_INVALID_ARG, "unsupported Merkle depth");
    memcpy(out, cols.data(), cols.size() * 8);
    return CSTARK_OK;
}

} // extern "C"

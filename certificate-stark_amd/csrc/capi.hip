// C ABI of the backend (include/cstark.h): context management, witness upload and the stage entry
// points.  No torch types, no CPU fallback: without a HIP device every compute entry point returns
// CSTARK_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include "../../include/cstark.h"
#include <deque>
#include "trace_gen.h"
#include "ntt.h"
#include "blake3.h"
#include "hostfield.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *detail = "") {
    snprintf(g_err, sizeof g_err, fmt, detail);
    return code;
}
#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? CSTARK_ERR_OOM : CSTARK_ERR_HIP, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)

} // namespace

struct NttPlan {
    unsigned log_n;
    uint64_t *w, *winv; // [n] each: powers of w_n and of its inverse
    uint64_t n_inv;
};
struct CosetTable {
    unsigned log_n, log_b;
    uint64_t offset;
    uint64_t *s; // [b][n]: (offset * w_{bn}^k)^m
};

struct cstark_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // uploaded witness
    void *wit_buf = nullptr;
    size_t wit_bytes = 0;
    cs::TxWitnessDev wit{};
    // cached tables and workspace
    std::deque<NttPlan> plans;   // deque: references stay valid as entries are added
    std::deque<CosetTable> cosets;
    void *ws = nullptr;
    size_t ws_bytes = 0;
};

namespace {

int get_plan(cstark_ctx *c, unsigned log_n, const NttPlan **out) {
    for (const NttPlan &p : c->plans)
        if (p.log_n == log_n) { *out = &p; return CSTARK_OK; }
    const size_t n = (size_t)1 << log_n;
    NttPlan p{log_n, nullptr, nullptr, 0};
    HIP_TRY(hipMalloc((void **)&p.w, n * 8));
    HIP_TRY(hipMalloc((void **)&p.winv, n * 8));
    const uint64_t w = cs::host::root_of_unity(log_n);
    HIP_TRY(cs::ntt_power_table(p.w, n, w, c->stream));
    HIP_TRY(cs::ntt_power_table(p.winv, n, cs::host::inv(w), c->stream));
    p.n_inv = cs::host::inv(cs::host::from_u64(n));
    c->plans.push_back(p);
    *out = &c->plans.back();
    return CSTARK_OK;
}

int get_coset_table(cstark_ctx *c, unsigned log_n, unsigned log_b, uint64_t offset, const CosetTable **out) {
    for (const CosetTable &t : c->cosets)
        if (t.log_n == log_n && t.log_b == log_b && t.offset == offset) { *out = &t; return CSTARK_OK; }
    const size_t n = (size_t)1 << log_n, b = (size_t)1 << log_b;
    CosetTable t{log_n, log_b, offset, nullptr};
    HIP_TRY(hipMalloc((void **)&t.s, b * n * 8));
    const uint64_t wbn = cs::host::root_of_unity(log_n + log_b);
    uint64_t shift = offset;
    for (size_t k = 0; k < b; k++) {
        HIP_TRY(cs::ntt_power_table(t.s + k * n, n, shift, c->stream));
        shift = cs::host::mul(shift, wbn);
    }
    c->cosets.push_back(t);
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

extern "C" {

const char *cstark_last_error(void) { return g_err; }
const char *cstark_version(void) { return "certificate-stark_amd 0.1 (gfx950)"; }

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
    *out = c;
    return CSTARK_OK;
}

void cstark_ctx_destroy(cstark_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->wit_buf) (void)hipFree(c->wit_buf);
    if (c->ws) (void)hipFree(c->ws);
    for (NttPlan &p : c->plans) { (void)hipFree(p.w); (void)hipFree(p.winv); }
    for (CosetTable &t : c->cosets) (void)hipFree(t.s);
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
    return CSTARK_OK;
}

int cstark_tx_build_trace(cstark_ctx *c, uint64_t *d_trace) {
    if (!c || !d_trace) return fail(CSTARK_ERR_INVALID_ARG, "cstark_tx_build_trace: null argument");
    if (!c->wit_buf || c->wit.n_tx == 0) return fail(CSTARK_ERR_INVALID_ARG, "no witness uploaded");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(cs::launch_trace_gen(c->wit, d_trace, c->stream));
    return CSTARK_OK;
}

// ---- K2 / K3 ---------------------------------------------------------------------------------------
uint64_t cstark_field_generator(void) { return cs::host::generator(); }
uint64_t cstark_field_root_of_unity(uint32_t log_n) { return log_n <= 55 ? cs::host::root_of_unity(log_n) : 0; }

int cstark_interpolate_columns(cstark_ctx *c, uint64_t *d_evals, uint64_t *d_coeffs, uint32_t width, uint32_t log_n) {
    if (!c || !d_evals || !d_coeffs || width == 0) return fail(CSTARK_ERR_INVALID_ARG, "cstark_interpolate_columns: bad argument");
    if (d_evals == d_coeffs) return fail(CSTARK_ERR_INVALID_ARG, "cstark_interpolate_columns: output must not alias input");
    if (log_n < cs::NTT_MIN_LOG_N || log_n > cs::NTT_MAX_LOG_N) return fail(CSTARK_ERR_UNSUPPORTED, "domain size must be 2^6 .. 2^24");
    HIP_TRY(hipSetDevice(c->device));
    const NttPlan *p;
    int rc = get_plan(c, log_n, &p);
    if (rc) return rc;
    cs::NttArgs a{};
    a.in = d_evals; a.scratch = d_evals; a.out = d_coeffs;
    a.width = width; a.batch = 1; a.log_n = log_n;
    a.w = p->winv; a.post_scale = p->n_inv; a.do_scale = true;
    HIP_TRY(cs::ntt_columns(a, c->stream));
    return CSTARK_OK;
}

int cstark_lde_columns(cstark_ctx *c, const uint64_t *d_coeffs, uint64_t *d_lde, uint32_t width, uint32_t log_n, uint32_t log_blowup,
                       uint64_t domain_offset, uint32_t k0, uint32_t nk) {
    if (!c || !d_coeffs || !d_lde || width == 0) return fail(CSTARK_ERR_INVALID_ARG, "cstark_lde_columns: bad argument");
    if (log_n < cs::NTT_MIN_LOG_N || log_n > cs::NTT_MAX_LOG_N || log_blowup > 6) return fail(CSTARK_ERR_UNSUPPORTED, "unsupported domain size");
    if (domain_offset == 0 || domain_offset >= cs::host::P) return fail(CSTARK_ERR_INVALID_ARG, "domain offset must be a nonzero field element");
    if ((uint64_t)k0 + nk > (1ull << log_blowup)) return fail(CSTARK_ERR_INVALID_ARG, "coset range exceeds the blowup factor");
    HIP_TRY(hipSetDevice(c->device));
    const NttPlan *p;
    const CosetTable *t;
    int rc = get_plan(c, log_n, &p);
    if (rc) return rc;
    if ((rc = get_coset_table(c, log_n, log_blowup, domain_offset, &t))) return rc;
    const size_t n = (size_t)1 << log_n;
    if ((rc = ensure_ws(c, (size_t)width * n * 8))) return rc;
    for (uint32_t k = k0; k < k0 + nk; k++) {
        cs::NttArgs a{};
        a.in = d_coeffs; a.scratch = (uint64_t *)c->ws; a.out = d_lde + (size_t)(k - k0) * width * n;
        a.width = width; a.batch = 1; a.log_n = log_n;
        a.w = p->w; a.prescale = t->s + (size_t)k * n; a.do_scale = false;
        HIP_TRY(cs::ntt_columns(a, c->stream));
    }
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

int cstark_merkle_build(cstark_ctx *c, uint8_t *d_nodes, uint32_t log_leaves) {
    if (!c || !d_nodes) return fail(CSTARK_ERR_INVALID_ARG, "cstark_merkle_build: null argument");
    if (log_leaves == 0 || log_leaves > 30) return fail(CSTARK_ERR_INVALID_ARG, "tree must have 2 .. 2^30 leaves");
    if (((uintptr_t)d_nodes & 15) != 0) return fail(CSTARK_ERR_INVALID_ARG, "d_nodes must be 16-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(cs::merkle_build(d_nodes, log_leaves, c->stream));
    return CSTARK_OK;
}

} // extern "C"

// C ABI of the backend (include/cstark.h): context management, witness upload and the stage entry
// points.  No torch types, no CPU fallback: without a HIP device every compute entry point returns
// CSTARK_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include "../../include/cstark.h"
#include "trace_gen.h"

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

struct cstark_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // uploaded witness
    void *wit_buf = nullptr;
    size_t wit_bytes = 0;
    cs::TxWitnessDev wit{};
};

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

} // extern "C"

/* ORACLE (test infrastructure, not product code).  Thin exported wrappers around the static-inline
 * field and gadget restatements (fp.h, gadgets.h) so the Python tests can call them. */
#include "oracle.h"
#include "gadgets.h"
#include <omp.h>

void cso_fp_from_u64(const uint64_t *in, uint64_t *out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = fp_from_u64(in[i]); }
void cso_fp_to_u64(const uint64_t *in, uint64_t *out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = fp_to_u64(in[i]); }
void cso_fp_mul(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = fp_mul(a[i], b[i]); }
void cso_fp_add(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = fp_add(a[i], b[i]); }
void cso_fp_sub(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = fp_sub(a[i], b[i]); }
void cso_fp_inv(const uint64_t *a, uint64_t *out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = fp_inv(a[i]); }
void cso_fp_pow(const uint64_t *a, uint64_t e, uint64_t *out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = fp_pow(a[i], e); }
uint64_t cso_fp_root_of_unity(unsigned log_n) { return fp_root_of_unity(log_n); }

void cso_rescue_permutation(uint64_t *s) { rescue_apply_permutation(s); }
void cso_rescue_round(uint64_t *s, uint32_t step) { rescue_apply_round(s, step); }
void cso_rescue_enforce_round(uint64_t *result, const uint64_t *cur, const uint64_t *next, const uint64_t *ark, uint64_t flag) {
    rescue_enforce_round(result, cur, next, ark, flag);
}
void cso_rescue_merge(const uint64_t *a, const uint64_t *b, uint64_t *out) { rescue_merge(a, b, out); }
void cso_rescue_digest(const uint64_t *data, size_t n, uint64_t *out) { rescue_digest(data, n, out); }
void cso_fp6_mul(const uint64_t *a, const uint64_t *b, uint64_t *out) { fp6_store(out, fp6_mul(fp6_load(a), fp6_load(b))); }
void cso_fp6_sqr(const uint64_t *a, uint64_t *out) { fp6_store(out, fp6_sqr(fp6_load(a))); }
void cso_fp6_inv(const uint64_t *a, uint64_t *out) { fp6_store(out, fp6_inv(fp6_load(a))); }
void cso_ecc_double(uint64_t *p) { ecc_double(p); }
void cso_ecc_add(uint64_t *p, const uint64_t *q) { ecc_add(p, q); }
void cso_ecc_add_mixed(uint64_t *p, const uint64_t *q) { ecc_add_mixed(p, q); }

int cso_num_threads(void) { return omp_get_max_threads(); }
void cso_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }

// Host-visible declarations for the FieldExtension::Quadratic kernels (ext.hip).  E = F_p[u]/(u^2 - 2u - 2), elements as pairs.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace cs {

size_t poly_eval_ext_scratch_words(unsigned width, unsigned log_n);
// d_out[c][2] = column c (base coefficients) at the point z of E
hipError_t poly_eval_ext(const uint64_t *d_coeffs, unsigned width, unsigned log_n, uint64_t za, uint64_t zb, uint64_t *d_out, uint64_t *d_scratch,
                         hipStream_t stream);

struct DeepExtParams {
    const uint64_t *trace_lde, *comp_lde; // [b][width][n] base; [b][2 nb][n] (column 2i + k = component k of composition column i)
    const uint64_t *w, *shifts;           // powers of w_n; g * w_{bn}^k per coset
    const uint64_t *coef;                 // device: alpha[width][2] | beta[width][2] | delta[nb][2]
    uint64_t *out;                        // [2][b][n] component-major
    uint64_t z[2], zw[2], zb[2], deg_a[2], deg_b[2];
    uint64_t k1[2], k2[2], k3[2];         // sum alpha_c T_c(z), sum beta_c T_c(z w), sum delta_i H_i(z^nb)
    uint32_t width, nb, log_n, log_b;
};
hipError_t deep_composition_ext(const DeepExtParams &p, hipStream_t stream);
// evals [2][N] component-major over offset * <w_N> -> [2][N/4]
hipError_t fri_fold4_ext(const uint64_t *d_evals, uint64_t *d_out, unsigned log_n, const uint64_t *d_winv, uint64_t offset_inv, uint64_t alpha_a,
                         uint64_t alpha_b, uint64_t inv4, hipStream_t stream);

} // namespace cs

// Host-visible declarations for the FieldExtension::Quadratic / Cubic kernels (ext.hip).  An element of the degree-m extension
// (m = 2: F_p[u]/(u^2 - 2u - 2); m = 3: F_p[v]/(v^3 + v + 1)) is m consecutive base elements.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace cs {

size_t poly_eval_ext_scratch_words(unsigned width, unsigned log_n, unsigned m);
// d_out[c][m] = column c (base coefficients) at the point z of the extension
hipError_t poly_eval_ext(const uint64_t *d_coeffs, unsigned width, unsigned log_n, const uint64_t *z, unsigned m, uint64_t *d_out, uint64_t *d_scratch,
                         hipStream_t stream);

struct DeepExtParams {
    const uint64_t *trace_lde, *comp_lde; // [b][width][n] base; [b][m nb][n] (column m i + k = component k of composition column i)
    const uint64_t *w, *shifts;           // powers of w_n; g * w_{bn}^k per coset
    const uint64_t *coef;                 // device: alpha[width][m] | beta[width][m] | delta[nb][m]
    uint64_t *out;                        // [m][b][n] component-major
    uint64_t z[3], zw[3], zb[3], deg_a[3], deg_b[3];
    uint64_t k1[3], k2[3], k3[3];         // sum alpha_c T_c(z), sum beta_c T_c(z w), sum delta_i H_i(z^nb)
    uint32_t width, nb, log_n, log_b, m;
    uint32_t nk;                          // cosets to evaluate (the first nk; 0 = all b): out is then [m][nk][n]
};
hipError_t deep_composition_ext(const DeepExtParams &p, hipStream_t stream);
// evals [m][N] component-major over offset * <w_N> -> [m][N/4]
hipError_t fri_fold4_ext(const uint64_t *d_evals, uint64_t *d_out, unsigned log_n, const uint64_t *d_winv, uint64_t offset_inv, const uint64_t *alpha,
                         unsigned m, uint64_t inv4, hipStream_t stream);
hipError_t fri_fold_ext(const uint64_t *d_evals, uint64_t *d_out, unsigned log_n, unsigned log_f, const uint64_t *d_winv, uint64_t offset_inv,
                        const uint64_t *alpha, unsigned m, uint64_t inv_f, hipStream_t stream);

// dst column m i + q <- column i of src[q] (i < cols, q < m, columns of n words): the components of the composition columns of an
// extension-field proof side by side, one launch (24 device-to-device copies before)
hipError_t interleave_set_columns(uint64_t *d_dst, const uint64_t *const src[3], unsigned m, unsigned cols, size_t n, hipStream_t stream);
} // namespace cs

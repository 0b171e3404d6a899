// Host-visible declarations for the out-of-domain / DEEP kernels (deep.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace cs {

struct DeepParams {
    const uint64_t *trace_lde, *comp_lde; // [nk][width][n], [nk][nb][n]
    const uint64_t *w, *shifts;           // powers of w_n; g * w_{bn}^k per coset
    const uint64_t *coef;                 // alpha[width] | beta[width] | delta[nb]        (device)
    const uint64_t *ood;                  // T(z)[width] | T(z w)[width] | H(z^nb)[nb]      (device)
    uint64_t *out;                        // [nk][n]
    uint64_t z, zw, zb, deg_a, deg_b;
    uint32_t width, nb, log_n, k0;
    const uint64_t *scal; // non-null: z, z w, z^nb, deg_a, deg_b are read from scal[0..5) (drawn on the device: channel.hip) instead of the fields above
};

// out[p * width + c] = value of coefficient column c at points[p]  (all device memory)
// d_scratch: poly_eval_scratch_words(width, log_n, npts) words
size_t poly_eval_scratch_words(unsigned width, unsigned log_n, unsigned npts);
hipError_t poly_eval(const uint64_t *d_coeffs, unsigned width, unsigned log_n, const uint64_t *d_points, unsigned npts, uint64_t *d_out,
                     uint64_t *d_scratch, hipStream_t stream);
hipError_t deep_composition(const DeepParams &p, unsigned nk, hipStream_t stream);
// N = 2^log_n evaluations over offset*<w_N> (natural order) -> N/4 evaluations of the alpha-folding over offset^4*<w_{N/4}>
// d_alpha != null: the folding point is read from device memory instead of `alpha`
hipError_t fri_fold4(const uint64_t *d_evals, uint64_t *d_out, unsigned log_n, const uint64_t *d_winv, uint64_t offset_inv, uint64_t alpha,
                     uint64_t inv4, hipStream_t stream, const uint64_t *d_alpha = nullptr);
// folding factor 2^log_f = 4, 8, 16; inv_f = 1 / 2^log_f (memory form)
hipError_t fri_fold(const uint64_t *d_evals, uint64_t *d_out, unsigned log_n, unsigned log_f, const uint64_t *d_winv, uint64_t offset_inv, uint64_t alpha,
                    uint64_t inv_f, hipStream_t stream, const uint64_t *d_alpha = nullptr);

} // namespace cs

// Host-visible declarations for the NTT kernels (ntt.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace cs {

constexpr unsigned NTT_MIN_LOG_N = 6, NTT_MAX_LOG_N = 24;

// One batched transform: `batch` x `width` independent length-2^log_n sequences, column stride n.
struct NttArgs {
    const uint64_t *in;    // natural order
    uint64_t *scratch;     // same shape as one batch of `in` times batch; may alias `in` (destroys it)
    uint64_t *out;         // natural order; must not alias scratch
    unsigned width, batch, log_n;
    const uint64_t *w;         // [n] powers of the root of unity to use (forward or inverse table)
    const uint64_t *prescale;  // optional [n] per batch: input element m is multiplied by prescale[m] (coset shift^m)
    size_t prescale_batch_stride;
    uint64_t post_scale;       // applied to every output when do_scale (n^-1 for the inverse transform)
    bool do_scale;
    bool inverse;              // w is the inverse table (selects the compile-time twiddles of the register kernels)
    size_t in_batch_stride, scratch_batch_stride, out_batch_stride; // in elements
    // optional compact tables of the three-step kernels (ntt_build_aux_*): of `w`, and per batch of `prescale`
    const uint64_t *aux = nullptr, *aux_ps = nullptr;
    size_t aux_ps_batch_stride = 0;
};

// Shape of the three-step kernels for a size (false: the size runs on other kernels and takes no compact tables):
// column pass over 2^log_r points, row pass over 2^log_c, output factors tabulated for 2^log_kb row classes.
struct NttV4Shape { unsigned log_r, log_c, log_kb; };
bool ntt_v4_shape(unsigned log_n, NttV4Shape *s);
size_t ntt_aux_plan_words(const NttV4Shape &s);
size_t ntt_aux_coset_words(const NttV4Shape &s);
// d_w: the n-entry power table of the plan (forward or inverse); d_s: one coset's n-entry prescale table (powers of its shift)
hipError_t ntt_build_aux_plan(uint64_t *d_aux, const uint64_t *d_w, const NttV4Shape &s, hipStream_t stream);
hipError_t ntt_build_aux_coset(uint64_t *d_aux, const uint64_t *d_w, const uint64_t *d_s, const NttV4Shape &s, hipStream_t stream);

hipError_t ntt_columns(const NttArgs &a, hipStream_t stream);
// table[e] = base^e, e < n
hipError_t ntt_power_table(uint64_t *d_table, size_t n, uint64_t base, hipStream_t stream);

// composition-polynomial helpers: coset-major -> natural order; split of H's coefficients into b columns (with the g^-m scaling)
hipError_t interleave_cosets(const uint64_t *d_in, uint64_t *d_out, unsigned log_n, unsigned log_b, hipStream_t stream);
// Second half of the interpolation over the whole b n-point domain from per-coset interpolants (B cosets = 2^log_b <= 8):
// d_b [B][n] = iNTT_n of every coset of a coset-major table; d_h [B n] = coefficients a_t in natural order.  With t = q + n i:
//   a_{q + n i} = (1 / B) sum_k w_B^(-k i) (w_N^(-k q) B_k[q]),   N = B n;  winv_N = powers of w_N^-1.
hipError_t coset_combine(const uint64_t *d_b, uint64_t *d_h, unsigned log_n, unsigned log_b, const uint64_t *d_winv_N, uint64_t b_inv,
                         hipStream_t stream, unsigned tables = 1);
// The way to the odd cosets for polynomials of degree < 4n given by the interpolants of the four even cosets (d_b [tables][4][n]):
// coset_combine with log_b = 2 (coefficients a_t of P(g y), t = q + n i) followed by b_k'[q] = sum_{i<4} w_8^((2k'+1) i) a_{q + n i}, the
// inputs of the n-point transforms over the odd cosets (whose prescale table supplies the twist w_8n^((2k'+1) q)), in one pass: the
// 4n coefficients are never written.  d_out [4 odd cosets][tables][n]; d_winv_4n = powers of w_4n^-1, d_w_8n = powers of w_8n, quarter = 1/4.
hipError_t coset_even_to_odd(const uint64_t *d_b, uint64_t *d_out, unsigned log_n, unsigned tables, const uint64_t *d_winv_4n, const uint64_t *d_w_8n,
                             uint64_t quarter, hipStream_t stream, unsigned kc0 = 0, unsigned nkc = 4); // (only even cosets [kc0, kc0 + nkc) present)
hipError_t split_columns(const uint64_t *d_h, uint64_t *d_out, unsigned log_n, unsigned log_b, uint64_t ginv, hipStream_t stream);

} // namespace cs

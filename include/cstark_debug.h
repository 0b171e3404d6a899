/* cstark_debug.h -- TEST-ONLY companion of cstark.h (libcstark_debug.so, built from certificate-stark_amd/csrc/debug/).
 *
 * Element-wise entry points for the device field arithmetic (csrc/fp.cuh, csrc/tower.cuh) so that the GPU parity tests can pin
 * every primitive against the oracle, and two micro-benchmarks.  None of this is in the product library libcstark_hip.so and
 * nothing in the product calls it.  Field elements: uint64_t in BaseElement memory form (see cstark.h); F_p6 elements are six
 * consecutive base elements; `stream` is a hipStream_t; all pointers are device memory unless noted.  Return: cstark_status.
 */
#ifndef CSTARK_DEBUG_H
#define CSTARK_DEBUG_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* out[i] = op(a[i], b[i]); op: 0 mul, 1 add, 2 sub, 3 inverse, 4 x^(1/alpha) (Rescue inverse S-box, src/utils/rescue.rs:337-341),
 * 5 from canonical, 6 to canonical, 7 negate, 8 double, 9 wave_next (lane l gets a of lane l + 1, lane 63 its own b),
 * 10 (a - b) * b with the unreduced difference, 11 / 12 Montgomery reduction of the 128-bit accumulator b 2^64 + a
 * (acc_reduce_below_p: b <= p - 2^32; acc_reduce: b < 2p). */
int cstark_debug_fp_op(void *stream, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t n, int op);
/* n independent F_p6 operations (src/utils/ecc.rs:506-591); op: 0 mul, 1 square, 2 inverse. */
int cstark_debug_fp6_op(void *stream, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t n, int op);
/* blocks x 256 lanes x ilp independent chains x iters Montgomery products, no memory traffic; *ms (host) = elapsed milliseconds. */
int cstark_debug_modmul_bench(void *stream, uint64_t *d_out, int blocks, int iters, int ilp, float *ms);
/* INV_MDS * v for npts 14-element vectors ([14][npts] column-major): limb dot products (use_mfma = 0) or the int8 matrix-core
 * variant of csrc/mds_mfma.cuh (1); same bits. */
int cstark_debug_mds(void *stream, const uint64_t *d_in, uint64_t *d_out, size_t npts, int use_mfma, int iters, float *ms);
#ifdef __cplusplus
}
#endif
#endif

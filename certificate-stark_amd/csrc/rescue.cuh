// Rescue-Prime over f63 (state 14, rate 7, 7 rounds; /root/reference/src/utils/rescue.rs:25-37) --
// device-side pieces shared by the trace-generation and constraint-evaluation kernels.
#pragma once
#include "fp.cuh"
#include "constants_gen.h"

namespace cs {

// Parameter tables (Montgomery form) in constant memory; one copy per translation unit.
__constant__ fp c_mds[196] = CS_MDS_MONT_INIT;
__constant__ fp c_inv_mds[196] = CS_INV_MDS_MONT_INIT;
__constant__ fp c_ark[8 * 28] = CS_ARK_MONT_INIT;
__constant__ fp c_b3[6] = CS_B3_MONT_INIT;
__constant__ fp c_generator[12] = CS_GENERATOR_MONT_INIT;

// x^3 for reduced x: the square stays unreduced in (0, 2p) -- it is the FIRST factor of the second product, which takes such values
// (fp.cuh) -- so a cube costs two products and ONE conditional subtraction
__device__ __forceinline__ fp fp_cube(fp x) { return fp_reduce_once(fp_mul_lazy(fp_mul_lazy(x, x), x)); }

} // namespace cs

// The five Rescue windows of TransactionAir's round gadgets and the layout of CeParams::rtab, shared by the vector-ALU kernels
// (constraints.hip: k_rounds_setup, k_rounds_split, k_eval_fused) and the matrix-core kernel (rounds_mfma.hip).
// Reference: src/air.rs:383-430 (the gadget calls), src/utils/rescue.rs:269-300, 345-375 (the round itself).
#pragma once
#include "constraints.h"
#include "mds_mfma.cuh"

namespace cs {

// {first register, result base A, flag A, result base B, flag B (-1: none)}; flag 3 = setup + hash (window 0 writes the same result
// slots under both flags, so one sum serves both).  Registers: S_INIT = 0, S_UPD = 15, R_INIT = 29, R_UPD = 44; 42 = the message hash.
struct RoundWindow { int reg, res_a, flag_a, res_b, flag_b; }; // dwords: sub-dword constants cannot be scalar loads
#define CS_ROUND_WINDOWS_INIT {{0, 0, 3, 0, -1}, {15, 14, 0, 15, 1}, {29, 28, 0, 29, 1}, {44, 42, 0, 44, 1}, {42, 42, 2, 0, -1}}
// degree groups present among the 14 result slots of each (window, flag set); -1 = unused
#define CS_WINDOW_GROUPS_INIT {{{0, 1, -1}, {-1, -1, -1}}, {{1, 2, 0}, {1, 2, 0}}, {{0, 2, -1}, {0, 2, -1}}, {{2, -1, -1}, {2, -1, -1}}, {{2, -1, -1}, {-1, -1, -1}}}

constexpr int RT_SECTIONS = 5 * 2 * 4; // (window, flag set, {alpha, beta of up to 3 groups})
// CeParams::rtab (u64 words): A[sections][64] | limbs of U[sections][14] (4 dwords each) | limbs of INV_MDS | byte-Toeplitz table of
// INV_MDS (the opt-in -DCS_ROUNDS_MFMA variant of the vector-ALU kernels) | the sections' coefficient vectors | tables of k_rounds_mfma
constexpr int RT_A = 0, RT_UL = RT_SECTIONS * 64, RT_ML = RT_UL + RT_SECTIONS * 14 * 2, RT_MT = RT_ML + 14 * 14 * 2;
constexpr size_t MT_BYTES = (mdsmfma::table_bytes(14) + 15) & ~(size_t)15;
constexpr int RT_G = RT_MT + (int)(MT_BYTES / 8); // the sections' coefficient vectors themselves (split evaluation)
constexpr int RT_MF = (RT_G + RT_SECTIONS * 14 + 1) & ~1; // 16-byte aligned

// ---- tables of k_rounds_mfma (rounds_mfma.hip), COMPACT: one 64-bit word of eight signed base-256 digits per matrix entry; the
// lanes expand a word into their bytes of the Toeplitz fragment with two v_perm_b32 (selectors fixed per lane).
// A 32-row tile holds two matrix rows ("outputs" g = 0, 1: the lane half that receives all 15 byte diagonals of the output) and is
// multiplied with the points' vectors in k-steps of 32 bytes = two values from each lane half h.
//   inverse matrix: tile T = outputs i = 2T + g; half h, index m <-> column j = 4 (m >> 1) + 2 h + (m & 1)   (j >= 14: zero)
//   sections:       tile t = (window, flag set, pair u), output g = slot 2u + g; half h, index m < 7: the section's coefficient of
//                   cube(INV_MDS (next - ark2))_i, i = 2 m + h; 7 <= m < 14: minus its MDS-folded coefficient of cube(cur_j), j = 2 (m - 7) + h
constexpr int MF_TILES_INV = 7, MF_TILES_SEC = 13, MF_KS_INV = 4, MF_KS_SEC = 7;
constexpr int MF_INV_D = RT_MF;                                  // [7][2 g][2 h][8 m]
constexpr int MF_SEC_D = MF_INV_D + MF_TILES_INV * 4 * 8;        // [13][2 g][2 h][16 m]
constexpr int MF_K = MF_SEC_D + MF_TILES_SEC * 4 * 16;           // [14 + 26][4]: the 32-bit words of the 128-bit row constants, one per u64
constexpr int RT_SIZE = MF_K + (14 + 2 * MF_TILES_SEC) * 4;
static_assert(RT_SIZE <= CE_RTAB_WORDS, "rtab size");
// section tile t -> (window, flag set, pair); window w owns the tiles [mf_tile_base(w), mf_tile_base(w + 1))
__host__ __device__ constexpr int mf_tile_base(int w) { return w == 0 ? 0 : w == 1 ? 2 : w == 2 ? 6 : w == 3 ? 10 : w == 4 ? 12 : 13; }
__host__ __device__ constexpr int mf_tile_window(int t) { return t < 2 ? 0 : t < 6 ? 1 : t < 10 ? 2 : t < 12 ? 3 : 4; }
__host__ __device__ constexpr int mf_tile_fs(int w, int local) { return (w == 1 || w == 2) ? local >> 1 : w == 3 ? local : 0; }
__host__ __device__ constexpr int mf_tile_pair(int w, int local) { return (w == 1 || w == 2) ? local & 1 : w == 0 ? local : 0; }

// MerkleAir's / SchnorrAir's folded round gadgets (k_merkle_rounds_setup, constraints.hip), rtab in u64 words; sections = (window,
// {alpha, beta}): A[8][8 cosets][8] | limbs of U[8][14] | limbs of INV_MDS[196] | G[8][14] | compact tables of k_merkle_rounds_mfma
constexpr int MR_SECTIONS = 8, MR_A = 0, MR_UL = MR_SECTIONS * 64, MR_ML = MR_UL + MR_SECTIONS * 14 * 2, MR_G = MR_ML + 196 * 2;
constexpr int MRF_INV_D = MR_G + MR_SECTIONS * 14;          // [7][2][2][8]
constexpr int MRF_SEC_D = MRF_INV_D + MF_TILES_INV * 32;   // [4 windows][2][2][16]
constexpr int MRF_K = MRF_SEC_D + 4 * 64;                  // [14 + 8][4]
constexpr int MR_SIZE = MRF_K + (14 + 8) * 4;
static_assert(MR_SIZE <= MERKLE_RTAB_WORDS, "MerkleAir rounds table");

hipError_t launch_merkle_rounds_mfma(const AirCombineParams &p, const uint64_t *ptab, uint64_t *d_rtab, unsigned nk, unsigned round_group, int which,
                                     hipStream_t stream); // rounds_mfma.hip; after k_merkle_rounds_setup
hipError_t launch_rounds_mfma(const CeParams &p, uint64_t *d_even, hipStream_t stream); // rounds_mfma.hip; after launch_rounds_setup

} // namespace cs

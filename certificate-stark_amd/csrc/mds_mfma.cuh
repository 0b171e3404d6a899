// 14x14 constant-matrix times per-point vector products over f63 on the matrix cores (v_mfma_i32_32x32x32_i8).
//
// y_i = REDC( sum_j M_ij x_j ),  M uniform (Rescue MDS / INV_MDS / per-proof folded rows), x one vector of 14 field elements
// per evaluation point.  The 64-bit integer products are decomposed into bytes:
//     sum_j M_ij x_j = sum_d 2^(8d) D_(i,d),   D_(i,d) = sum_j sum_(a+b=d) M_ij^(a) x_j^(b)
// and the diagonal sums D are ONE int8 GEMM: rows (i, d), contraction index k = (j, b) (112 of 128), columns = points, with the
// Toeplitz matrix A[(i,d)][(j,b)] = M_ij^(d-b).  The B operand column of a point is simply the 112 bytes of its vector.
// The matrix cores multiply SIGNED bytes, so M is stored in signed base-256 digits (exact, precomputed) and the vector bytes
// 0..6 are offset by 128 (x = s + X0, X0 = 0x0080808080808080; byte 7 of a field element is < 0x42 and stays as it is):
//     sum_j M_ij x_j = sum_j M_ij s_j + X0 sum_j M_ij,
// the last term being a constant of row i.  Accumulators start at 2^23 so every diagonal leaves the MFMA non-negative and
// below 2^24; both constants are folded into one 128-bit constant per row.  All integer arithmetic is exact: the value fed to
// the Montgomery reduction equals the carry-propagated 128-bit sum of the scalar code, bit for bit.
//
// Lane layout (32x32x32): A fragment of lane (r = l & 31, h = l >> 5): 16 bytes k = 32 s + 16 h .. +15 of tile row r;
// B fragment: the same k range of column n = l & 31; C: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).
// Tile row r of tile T is (i, d) = (2T + ((r >> 2) & 1), (r & 3) + 4 (r >> 3)), so that register v of lane half g holds
// diagonal d = v of output i = 2T + g: all 15 diagonals of one output meet in one lane.
#pragma once
#include "fp.cuh"

namespace cs {
namespace mdsmfma {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned __int128 u128;

constexpr int ROW_BYTES = 144;                 // 128 data bytes + 16 pad: conflict-free ds_read_b128 across a lane group
constexpr uint64_t X0 = 0x0080808080808080ULL; // byte offset of the vector operand
constexpr int ACC0 = 1 << 23;                  // accumulator start: |diagonal| <= 14 * 8 * 128 * 128 < 2^21

// table for `nrows` matrix rows (padded to an even count): A tiles [nrows / 2][32][ROW_BYTES] then nrows 128-bit constants.
__host__ __device__ constexpr size_t table_bytes(int nrows) { return (size_t)((nrows + 1) / 2) * 32 * ROW_BYTES + (size_t)((nrows + 1) / 2) * 2 * 16; }

// signed base-256 digits of v < 2^63: v = sum_a dig[a] 2^(8a), dig in [-128, 127]
__device__ __forceinline__ void signed_digits(uint64_t v, int8_t dig[8]) {
    int carry = 0;
    for (int a = 0; a < 8; a++) {
        int b = (int)((v >> (8 * a)) & 0xff) + carry;
        carry = b >= 128 ? 1 : 0;
        dig[a] = (int8_t)(b - 256 * carry);
    }
}
// One thread builds one (row i, column j) entry set; call with i < nrows_padded, j < 16.  m = row-major [nrows][14] matrix
// (rows >= nrows and columns >= 14 are zero).  Also writes the row constants (thread j == 0).
__device__ inline void build_table_entry(uint8_t *tab, const fp *m, int nrows, int i, int j) {
    const int T = i >> 1, gsel = i & 1, ntiles = (nrows + 1) / 2;
    int8_t dig[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (i < nrows && j < 14) signed_digits(m[i * 14 + j], dig);
    for (int d = 0; d < 16; d++) {
        const int r = (d & 3) + 8 * (d >> 2) + 4 * gsel; // tile row holding (i, d)
        uint8_t *row = tab + ((size_t)T * 32 + r) * ROW_BYTES + 8 * j;
        for (int b = 0; b < 8; b++) row[b] = (d < 15 && d - b >= 0 && d - b <= 7) ? (uint8_t)dig[d - b] : 0;
    }
    if (j == 0) {
        u128 k = 0;
        if (i < nrows)
            for (int jj = 0; jj < 14; jj++) k += (u128)m[i * 14 + jj] * X0;
        u128 off = 0;
        for (int d = 0; d < 15; d++) off += (u128)ACC0 << (8 * d);
        k -= off; // mod 2^128
        uint64_t *kc = (uint64_t *)(tab + (size_t)ntiles * 32 * ROW_BYTES) + 2 * i;
        kc[0] = (uint64_t)k; kc[1] = (uint64_t)(k >> 64);
    }
}

// the 15 non-negative 24-bit diagonals of one output -> Montgomery-reduced field element.
// Diagonals d, d+3, d+6, ... are disjoint 24-bit fields, so the three residue classes are packed without carries -- directly at
// their bit offsets 0 / 8 / 16 -- and the 128-bit value is three word-wise additions (plus the row constant).  32-bit
// operations throughout: v_lshl_or_b32 / v_lshrrev_b32 for the packing, v_add_co / v_addc chains for the sums.
__device__ __forceinline__ void add128(uint32_t (&a)[4], const uint32_t (&b)[4]) {
    unsigned c;
    a[0] = __builtin_addc(a[0], b[0], 0u, &c);
    a[1] = __builtin_addc(a[1], b[1], c, &c);
    a[2] = __builtin_addc(a[2], b[2], c, &c);
    a[3] = a[3] + b[3] + c;
}
__device__ __forceinline__ fp recombine(const v16i &acc, uint64_t klo, uint64_t khi) {
    const uint32_t a0 = acc[0], a3 = acc[3], a6 = acc[6], a9 = acc[9], a12 = acc[12];   // offset 0:  bits 0, 24, 48, 72, 96
    const uint32_t b1 = acc[1], b4 = acc[4], b7 = acc[7], b10 = acc[10], b13 = acc[13]; // offset 8:  bits 8, 32, 56, 80, 104
    const uint32_t c2 = acc[2], c5 = acc[5], c8 = acc[8], c11 = acc[11], c14 = acc[14]; // offset 16: bits 16, 40, 64, 88, 112
    uint32_t v[4] = {a0 | (a3 << 24), (a3 >> 8) | (a6 << 16), (a6 >> 16) | (a9 << 8), a12};
    const uint32_t g1[4] = {b1 << 8, b4 | (b7 << 24), (b7 >> 8) | (b10 << 16), (b10 >> 16) | (b13 << 8)};
    const uint32_t g2[4] = {c2 << 16, (c2 >> 16) | (c5 << 8), c8 | (c11 << 24), (c11 >> 8) | (c14 << 16)};
    const uint32_t kk[4] = {(uint32_t)klo, (uint32_t)(klo >> 32), (uint32_t)khi, (uint32_t)(khi >> 32)};
    add128(v, g1);
    add128(v, g2);
    add128(v, kk);
    Acc128 a{((uint64_t)v[1] << 32) | v[0], ((uint64_t)v[3] << 32) | v[2]};
    acc_fold(a);
    return acc_reduce(a);
}

// stage one vector (14 field elements of this lane's point) into the wave's B image: stage + pt * ROW_BYTES
__device__ __forceinline__ void stage_vector(uint8_t *stage, int pt, const fp (&x)[14]) {
    uint64_t *row = (uint64_t *)(stage + (size_t)pt * ROW_BYTES);
#pragma unroll
    for (int j = 0; j < 14; j++) row[j] = x[j] ^ X0;
    row[14] = 0; row[15] = 0; // k >= 112: the matching A bytes are zero as well
}
struct BFrags { v4i f[4][2]; };
__device__ __forceinline__ void load_bfrags(BFrags &b, const uint8_t *stage, int lane) {
    const int n = lane & 31, h = lane >> 5;
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) b.f[s][nt] = *(const v4i *)(stage + (size_t)(32 * nt + n) * ROW_BYTES + 32 * s + 16 * h);
}
// tile T of the table against the staged vectors: y[nt] = output i = 2T + (lane >> 5) at point 32 nt + (lane & 31)
__device__ __forceinline__ void tile_product(const uint8_t *tab, int ntiles, int T, const BFrags &b, int lane, fp (&y)[2]) {
    const int r = lane & 31, h = lane >> 5;
    v16i acc0, acc1;
#pragma unroll
    for (int v = 0; v < 16; v++) { acc0[v] = ACC0; acc1[v] = ACC0; }
    const uint8_t *arow = tab + ((size_t)T * 32 + r) * ROW_BYTES + 16 * h;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const v4i a = *(const v4i *)(arow + 32 * s);
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b.f[s][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b.f[s][1], acc1, 0, 0, 0);
    }
    const uint64_t *kc = (const uint64_t *)(tab + (size_t)ntiles * 32 * ROW_BYTES) + 2 * (2 * T + h);
    y[0] = recombine(acc0, kc[0], kc[1]);
    y[1] = recombine(acc1, kc[0], kc[1]);
}

} // namespace mdsmfma
} // namespace cs

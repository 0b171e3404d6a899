// Host-side description of the composite TransactionAir that the engine would obtain through the Air trait:
// periodic (mask) columns, constraint degrees and degree adjustments.  Restates
//   periodic_columns()          /root/reference/src/air.rs:194-380
//     merkle masks              src/merkle/update/air.rs:182-212
//     schnorr masks             src/schnorr/air.rs:335-391
//     round constants           src/utils/rescue.rs:306-320
//   TransactionAir::new degrees src/air.rs:76-108
// It runs once per (depth, trace length); the resulting table is kernel input (K7).
#pragma once
#include "../../include/cstark_conventions.h"
#include <stdint.h>
#include <vector>
#include "constants_gen.h"
#include "constraints.h"
#include "hostfield.h"

namespace cs {
namespace host {

constexpr int TX_NUM_PERIODIC = 48, TX_CYCLE = 1024;

// 48 columns x 1024 rows, column-major.  The length-8 columns of the reference (HASH_INPUT mask and the 28
// round-constant columns) are written out over the full 1024-row cycle: same periodic polynomial.
inline bool tx_periodic_columns(unsigned depth, std::vector<uint64_t> &out) {
    const unsigned hash_len = 8 * depth + 7; // TRANSACTION_HASH_LENGTH, src/merkle/constants.rs:27
    if (depth == 0 || hash_len > 511) return false;
    out.assign((size_t)TX_NUM_PERIODIC * TX_CYCLE, 0);
    auto col = [&](int c) { return out.data() + (size_t)c * TX_CYCLE; };
    const unsigned S = 512; // Schnorr half starts here (MERKLE_UPDATE_LENGTH)
    col(0)[0] = ONE;                                                     // SETUP
    for (unsigned i = 0; i < hash_len; i++) {
        col(1)[i] = ONE;                                                 // MERKLE (hashing in progress)
        col(4)[i] = (i % 8) != 7 ? ONE : 0;                              // HASH (round steps)
    }
    col(3)[hash_len - 1] = ONE;                                          // FINISH
    for (unsigned i = 0; i < TX_CYCLE; i++) col(2)[i] = (i % 8) == 7 ? ONE : 0; // HASH_INPUT (period 8)
    for (unsigned i = 0; i < 511; i++) col(5)[S + i] = ONE;              // SCHNORR global mask (SCALAR_MUL_LENGTH + 1)
    for (unsigned i = 0; i < 510; i++) {
        col(6)[S + i] = ONE;                                             // SCALAR_MULT
        col(7)[S + i] = (i % 2 == 0) ? ONE : 0;                          // DOUBLING
    }
    const unsigned lo[4] = {0, 126, 254, 382}, hi[4] = {126, 254, 382, 510};
    for (int k = 0; k < 4; k++)
        for (unsigned i = lo[k]; i < hi[k]; i++) col(8 + k)[S + i] = ONE; // h-limb selectors
    for (unsigned i = 0; i < 40; i++) col(12)[S + i] = (i % 8) != 7 ? ONE : 0; // SCHNORR_HASH
    for (int k = 0; k < 4; k++) col(13 + k)[S + 8 * (k + 1) - 1] = ONE;  // message-chunk insertion steps
    for (unsigned i = 0; i < 64; i++) col(17)[S + i] = ONE;              // RANGE_STEP
    col(18)[S + 63] = ONE;                                               // RANGE_FINISH
    for (unsigned i = 1; i < S + 64; i++) col(19)[i] = ONE;              // VALUE_COPY
    for (int j = 0; j < 28; j++)
        for (unsigned i = 0; i < TX_CYCLE; i++) col(20 + j)[i] = CS_ARK_MONT[(i % 8) * 28 + j];
    return true;
}

// evaluation degree of degree group g for trace length n (TransitionConstraintDegree [UPSTREAM-RECALL])
inline uint64_t tx_group_eval_degree(int g, uint64_t n) { return TX_GROUP_BASE[g] * (n - 1) + TX_GROUP_CYCLES[g] * (n / TX_CYCLE) * (TX_CYCLE - 1); }
// adjustment so that every merged constraint reaches degree (ce_size - 1) + (n - 1) before division
inline uint64_t tx_group_adjustment(int g, uint64_t n, uint64_t ce_size) { return CSTARK_CONV_TRANSITION_ADJUSTMENT(ce_size, n, tx_group_eval_degree(g, n)); }
inline uint64_t tx_boundary_adjustment(uint64_t n, uint64_t ce_size) { return CSTARK_CONV_BOUNDARY_ADJUSTMENT(ce_size, n, 1); }
static_assert(CSTARK_CONV_TRANSITION_EXEMPTIONS == 1, "the evaluators divide by (x^n - 1) / (x - w^(n-1)): one exempted step");

// ---- standalone sub-AIRs ------------------------------------------------------------
// MerkleAir periodic columns (src/merkle/update/air.rs:182-212): setup, hash, hash_input (period 8), finish,
// hash_mask, 28 round constants; 33 columns x 512 rows
inline bool merkle_periodic_columns(unsigned depth, std::vector<uint64_t> &out) {
    const unsigned hash_len = 8 * depth + 7;
    if (depth == 0 || hash_len > 511) return false;
    out.assign((size_t)33 * 512, 0);
    auto col = [&](int c) { return out.data() + (size_t)c * 512; };
    col(0)[0] = ONE;
    for (unsigned i = 0; i < hash_len; i++) { col(1)[i] = ONE; col(4)[i] = (i % 8) != 7 ? ONE : 0; }
    for (unsigned i = 0; i < 512; i++) col(2)[i] = (i % 8) == 7 ? ONE : 0;
    col(3)[hash_len - 1] = ONE;
    for (int j = 0; j < 28; j++)
        for (unsigned i = 0; i < 512; i++) col(5 + j)[i] = CS_ARK_MONT[(i % 8) * 28 + j];
    return true;
}

// SchnorrAir input-independent periodic columns (src/schnorr/air.rs:335-391): global mask, scalar-mult, doubling, four
// h-limb selectors, hash flag, then the 28 round constants; 36 columns x 512 rows
inline void schnorr_mask_columns(std::vector<uint64_t> &out) {
    out.assign((size_t)36 * 512, 0);
    auto col = [&](int c) { return out.data() + (size_t)c * 512; };
    for (unsigned i = 0; i < 511; i++) col(0)[i] = ONE;
    for (unsigned i = 0; i < 510; i++) { col(1)[i] = ONE; col(2)[i] = (i % 2 == 0) ? ONE : 0; }
    const unsigned lo[4] = {0, 126, 254, 382}, hi[4] = {126, 254, 382, 510};
    for (int k = 0; k < 4; k++)
        for (unsigned i = lo[k]; i < hi[k]; i++) col(3 + k)[i] = ONE;
    for (unsigned i = 0; i < 40; i++) col(7)[i] = (i % 8) != 7 ? ONE : 0;
    for (int j = 0; j < 28; j++)
        for (unsigned i = 0; i < 512; i++) col(8 + j)[i] = CS_ARK_MONT[(i % 8) * 28 + j];
}

// RescueAir periodic columns (benches/rescue.rs:133-142, :245-249): the cycle mask (seven ones, one zero), then the 28 round constants;
// 29 columns x 8 rows
inline void rescue_chain_periodic_columns(std::vector<uint64_t> &out) {
    out.assign((size_t)29 * 8, 0);
    for (unsigned i = 0; i < 7; i++) out[i] = ONE;
    for (int j = 0; j < 28; j++)
        for (unsigned i = 0; i < 8; i++) out[(size_t)(1 + j) * 8 + i] = CS_ARK_MONT[i * 28 + j];
}

// Static description of an AIR as the engine sees it: width, constraint degrees (base; cycles of length cycle_len),
// single-step assertions.  air ids as in cstark_air_id.
struct AirShape {
    uint32_t width = 0, n_constraints = 0, cycle_len = 0, n_periodic = 0;
    std::vector<uint32_t> base, cycles;
    std::vector<uint32_t> a_reg, a_last; // assertion register, 0 = first step / 1 = last step (single assertions)
    // generalisation (Assertion::periodic / ::sequence): when a_stride is non-empty assertion a holds at steps
    // a_first[a] + k * a_stride[a]; its value is the caller's assertion_values[a] or, if a_seq[a] >= 0, column a_seq[a] of
    // the extended sequence-value polynomials
    std::vector<uint32_t> a_first, a_stride;
    std::vector<int32_t> a_seq;
    std::vector<uint64_t> a_const; // built-in constant values (SchnorrAir), empty when the caller supplies them
    uint32_t log_ce_blowup() const {     // next power of two >= max(base + cycles), at least 2 [UPSTREAM-RECALL]
        uint32_t m = 2;
        for (size_t i = 0; i < base.size(); i++) m = base[i] + (cycle_len ? cycles[i] : 0) > m ? base[i] + (cycle_len ? cycles[i] : 0) : m;
        uint32_t l = 0;
        while ((1u << l) < m) l++;
        return l;
    }
    uint64_t eval_degree(size_t i, uint64_t n) const { return base[i] * (n - 1) + (cycle_len ? cycles[i] * (n / cycle_len) * (cycle_len - 1) : 0); }
};
inline bool air_shape(int air, AirShape &s, uint32_t n_items = 2) {
    s = AirShape{};
    if (air == 1) { // MerkleAir: transition_constraint_degrees(512), src/merkle/update/air.rs:371-401; 14 root assertions :142-170
        s.width = 65; s.n_constraints = 106; s.cycle_len = 512; s.n_periodic = 33;
        s.base.assign(106, 1); s.cycles.assign(106, 1);
        for (int b = 0; b < 58; b += 29) { for (int i = 0; i < 29; i++) s.base[b + i] = 3; s.base[b + 14] = 2; }
        for (int a = 0; a < 14; a++) { s.a_reg.push_back(58 + a % 7); s.a_last.push_back(a / 7); }
        return true;
    }
    if (air == 2) { // SchnorrAir: degrees src/schnorr/air.rs:533-585 (bit degree depends on the number of signatures),
                    // the 61 periodic / sequence assertions of get_assertions (:111-226) in order
        s.width = 56; s.n_constraints = 56; s.cycle_len = 512; s.n_periodic = 36;
        s.base.assign(56, 0); s.cycles.assign(56, 0);
        const uint32_t bit_degree = n_items == 1 ? 3 : 5;
        for (int i = 0; i < 6; i++) { s.base[i] = 5; s.cycles[i] = 2; }
        for (int i = 6; i < 18; i++) { s.base[i] = 4; s.cycles[i] = 2; }
        s.base[18] = 2; s.cycles[18] = 1;
        for (int i = 19; i < 37; i++) { s.base[i] = bit_degree; s.cycles[i] = 2; }
        s.base[37] = 2; s.cycles[37] = 1;
        for (int i = 38; i < 42; i++) { s.base[i] = 1; s.cycles[i] = 2; }
        for (int i = 42; i < 56; i++) { s.base[i] = 3; s.cycles[i] = 1; }
        auto add = [&](uint32_t r, uint32_t first, uint64_t v, int32_t q) {
            s.a_reg.push_back(r); s.a_last.push_back(0); s.a_first.push_back(first); s.a_stride.push_back(512); s.a_const.push_back(v); s.a_seq.push_back(q);
        };
        for (int i = 0; i < 18; i++) add(i, 0, i == 6 ? ONE : 0, -1);
        add(18, 0, 0, -1);
        for (int i = 0; i < 18; i++) add(19 + i, 0, i == 6 ? ONE : 0, -1);
        for (int i = 0; i < 5; i++) add(37 + i, 0, 0, -1);
        for (int k = 0; k < 6; k++) add(42 + k, 0, 0, k);
        for (int i = 0; i < 7; i++) add(48 + i, 0, 0, -1);
        for (int k = 0; k < 6; k++) add(k, 511, 0, 6 + k);
        return true;
    }
    if (air == 4) { // RescueAir of benches/rescue.rs: 14 x (3; one cycle of 8) :169-191, seed / result assertions :224-243
        s.width = 14; s.n_constraints = 14; s.cycle_len = 8; s.n_periodic = 29;
        s.base.assign(14, 3); s.cycles.assign(14, 1);
        for (int a = 0; a < 14; a++) { s.a_reg.push_back(a % 7); s.a_last.push_back(a / 7); }
        return true;
    }
    if (air == 3) { // RangeProofAir: degrees (2), (1), src/range/air.rs:100-105; assertions :79-86
        s.width = 2; s.n_constraints = 2; s.cycle_len = 0; s.n_periodic = 0;
        s.base = {2, 1}; s.cycles = {0, 0};
        s.a_reg = {1, 1}; s.a_last = {0, 1};
        return true;
    }
    return false;
}

} // namespace host
} // namespace cs

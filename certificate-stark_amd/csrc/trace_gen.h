// Host-visible declarations for the trace-generation kernels (trace_gen.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cs {

// Device-resident copy of a cstark_tx_witness (include/cstark.h); all pointers are HBM addresses.
struct TxWitnessDev {
    uint32_t n_tx, depth;
    const uint64_t *initial_roots, *s_old, *r_old, *s_idx, *r_idx, *s_paths, *r_paths, *deltas, *sig_rx;
    const uint8_t *sig_s;
    uint64_t *h_limbs; // [n_tx][4] canonical limbs of hash_message(), produced by k_trace_schnorr_hash
    const uint64_t *msg_tail; // standalone SchnorrAir only: message[26], message[27] per signature (zero in the composite AIR)
};

// `side` is a second stream used for the part of the trace that is independent of the rest; `fork`/`join` are events
// owned by the caller (fork: stream -> side, join: side -> stream)
hipError_t launch_trace_gen(const TxWitnessDev &w, uint64_t *d_trace, hipStream_t stream, hipStream_t side, hipEvent_t fork, hipEvent_t join);
// The same trace spread over three streams, nothing joined: `stream` gets the closed-form registers (k_trace_aux: 65..93 and the
// bit / accumulator registers of the Schnorr rows); side_a the Merkle recurrence (join_a recorded behind it); side_b the message
// hash (mid_b behind it: registers 42..55 and the scalar h) and then the curve ladders (join_b behind them).
hipError_t launch_trace_gen_split(const TxWitnessDev &w, uint64_t *d_trace, hipStream_t stream, hipStream_t side_a, hipStream_t side_b,
                                  hipEvent_t fork, hipEvent_t join_a, hipEvent_t mid_b, hipEvent_t join_b);

// standalone sub-AIR traces
hipError_t launch_merkle_trace(const TxWitnessDev &w, uint64_t *d_trace, hipStream_t stream); // 65 x 512*n_tx
hipError_t launch_range_trace(uint64_t number_canonical, uint64_t *d_trace, hipStream_t stream); // 2 x 64
// RescueProver::build_trace (benches/rescue.rs:277-322): 14 x 8*iterations, seed = 7 elements in memory form; iterations a multiple of 8
hipError_t launch_rescue_chain_trace(const uint64_t seed[7], unsigned iterations, uint64_t *d_trace, hipStream_t stream);
// the same accumulator over 2^log_n rows of an (n-1)-bit integer (d_words: n/64 little-endian words; d_prefix: n/64 words scratch)
hipError_t launch_range_trace_bits(const uint64_t *d_words, uint64_t *d_prefix, uint64_t *d_trace, unsigned log_n, hipStream_t stream);
// SchnorrProver::build_trace (src/schnorr/prover.rs:40-67): 56 x 512*n; the witness view holds message[0..12] in s_old,
// [12..24] in r_old, [24] in deltas, [25] in s_old[13], [26..28] in msg_tail
hipError_t launch_schnorr_trace(const TxWitnessDev &w, uint64_t *d_trace, hipStream_t stream);
// the same on the internal stream `side` (forked from `stream`): join_a = registers 37..55 complete, join_b = all registers complete
hipError_t launch_schnorr_trace_split(const TxWitnessDev &w, uint64_t *d_trace, hipStream_t stream, hipStream_t side, hipEvent_t fork, hipEvent_t join_a,
                                      hipEvent_t join_b);
// the 19 public-input columns of SchnorrAir (pkey x12, message chunks x7; src/schnorr/air.rs:228-290): 19 x 512*n
hipError_t launch_schnorr_aux_columns(const TxWitnessDev &w, uint64_t *d_out, hipStream_t stream);

} // namespace cs

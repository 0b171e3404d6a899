/* cstark.h -- C ABI of the MI355X-native prover backend for the Topos state-transition AIR.
 *
 * This is the drop-in boundary for the reference's hot path.  The reference (a Rust crate) has no
 * FFI of its own; these entry points are what a Rust shim bound at
 *     impl Prover for TransactionProver            /root/reference/src/prover.rs:101-134
 *     TransactionProver::build_trace               /root/reference/src/prover.rs:37-98
 *     TransactionExample::prove                    /root/reference/src/lib.rs:116-141
 * would call (the binding a maintainer adds is shown in INTEGRATION.md).
 *
 * NOT winterfell-compatible: the reference's engine (winterfell fork, Cargo.toml:20) is absent from its tree.  Field conventions, the
 * Fiat-Shamir coin and the extension polynomials are recalled / assumed (every one a macro in cstark_conventions.h), the channel
 * order and the proof byte layout are this library's own (documented at cstark_tx_prove).  "Bit-exact" in this header always means:
 * against the CPU restatement under oracle/, which is pinned by algebraic known answers and hash test vectors, not by a Rust run.
 * The stage entry points (K1..K6) hand an engine its own objects (LDE values, digests, merged evaluations) and depend on two
 * conventions only (domain offset / root of unity, byte form of hashed elements).
 *
 * Conventions
 *  - Field elements cross the ABI as uint64_t holding the in-memory representation of the
 *    reference's `f63::BaseElement`: Montgomery form, R = 2^64, p = 2^62+2^56+2^55+1, reduced to
 *    [0,p).  A Rust `&[BaseElement]` can therefore be passed without conversion.
 *  - Matrices of field elements are column-major ("one Vec per register", as TraceTable stores
 *    them): element (column c, row r) of a W x N table lives at  base[c * N + r].
 *  - `d_*` pointers are device (HBM) addresses, everything else is host memory.
 *  - All functions return CSTARK_OK (0) or a negative cstark_status; they never abort.  The
 *    reference reports failure through Result / panics (src/lib.rs:140, :212-218).
 *  - The library is re-entrant per context: one cstark_ctx owns one HIP stream and its workspace.
 */
#ifndef CSTARK_H
#define CSTARK_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cstark_status {
    CSTARK_OK = 0,
    CSTARK_ERR_INVALID_ARG = -1,   /* malformed sizes / null pointers (reference: assert! panics) */
    CSTARK_ERR_NO_DEVICE = -2,     /* no HIP device / kernel image missing: fail loudly, no CPU fallback */
    CSTARK_ERR_HIP = -3,           /* a HIP runtime call failed; see cstark_last_error() */
    CSTARK_ERR_OOM = -4,
    CSTARK_ERR_UNSUPPORTED = -5
} cstark_status;

/* which AIR program (BASELINE.json configs; reference benches/{rescue,range,merkle,schnorr,state_transition}.rs) */
typedef enum cstark_air_id {
    CSTARK_AIR_STATE_TRANSITION = 0, /* TransactionAir, src/air.rs:64-189; 94 registers, 1024 rows / tx */
    CSTARK_AIR_MERKLE_UPDATE = 1,    /* MerkleAir, src/merkle/update/air.rs:36-177; 65 registers, 512 rows / tx */
    CSTARK_AIR_SCHNORR = 2,          /* SchnorrAir, src/schnorr/air.rs:41-300; 56 registers, 512 rows / sig */
    CSTARK_AIR_RANGE = 3,            /* RangeProofAir, src/range/air.rs:23-105; 2 registers, 64 rows */
    CSTARK_AIR_RESCUE_CHAIN = 4      /* RescueAir, benches/rescue.rs:128-360; 14 registers, 8 rows / link (cstark_rescue_prove) */
} cstark_air_id;

#define CSTARK_TX_TRACE_WIDTH 94      /* src/constants.rs:35 */
#define CSTARK_TX_CYCLE_LENGTH 1024   /* src/constants.rs:83 */
#define CSTARK_TX_NUM_CONSTRAINTS 115 /* src/air.rs:76-100 */
#define CSTARK_TX_NUM_PERIODIC 48     /* src/air.rs:195, src/constants.rs:116 */
#define CSTARK_DIGEST_BYTES 32        /* Blake3_256, src/lib.rs:82 */

/* Witness for a batch of transfers: field-for-field the reference's TransactionMetadata
 * (src/lib.rs:183-194).  Field elements in BaseElement memory form (see Conventions). */
typedef struct cstark_tx_witness {
    uint32_t n_tx;                /* number of transactions; a power of two for proving */
    uint32_t merkle_depth;        /* MERKLE_TREE_DEPTH: 15 (src/merkle/constants.rs:25), 3 under cfg(test) (:22) */
    const uint64_t *initial_roots; /* [n_tx][7]   root before each transaction */
    const uint64_t *final_root;    /* [7] */
    const uint64_t *s_old_values;  /* [n_tx][14]  sender leaf: pk.x(6) pk.y(6) balance nonce */
    const uint64_t *r_old_values;  /* [n_tx][14]  receiver leaf */
    const uint64_t *s_indices;     /* [n_tx] */
    const uint64_t *r_indices;     /* [n_tx] */
    const uint64_t *s_paths;       /* [n_tx][merkle_depth+1][7]  [leaf, sibling_0 .. sibling_{d-1}] */
    const uint64_t *r_paths;       /* [n_tx][merkle_depth+1][7]  (after the sender update) */
    const uint64_t *deltas;        /* [n_tx] */
    const uint64_t *sig_rx;        /* [n_tx][6]   signature.0 = R.x */
    const uint8_t *sig_s;          /* [n_tx][32]  signature.1.to_bytes(), little-endian (src/schnorr/trace.rs:133) */
} cstark_tx_witness;

/* Mirror of the 7 ProofOptions fields (src/lib.rs:78-86). */
typedef struct cstark_options {
    uint32_t num_queries;      /* 42; 1..128 */
    uint32_t blowup_factor;    /* 8; 2, 4, 8 or 16 and at least the AIR's constraint-evaluation blowup (see cstark_tx_prove) */
    uint32_t grinding_factor;  /* 0; up to 32 bits of proof of work (Blake3 coin, 12 bits and more: searched on the device) */
    uint32_t hash_fn;          /* 0 = Blake3_256, 1 = Sha3_256 */
    uint32_t field_extension;  /* 0 = None, 1 = Quadratic, 2 = Cubic */
    uint32_t fri_folding_factor; /* 4; 4, 8 or 16 */
    uint32_t fri_max_remainder;  /* 256; 128, 256, 512 or 1024 */
} cstark_options;

/* Composition coefficients for the constraint-evaluation driver (what the engine draws from the
 * public coin after the trace commitment; inputs here so the stage can be tested in isolation). */
typedef struct cstark_tx_coeffs {
    uint64_t t_alpha[CSTARK_TX_NUM_CONSTRAINTS]; /* per transition constraint: (alpha, beta) */
    uint64_t t_beta[CSTARK_TX_NUM_CONSTRAINTS];
    uint64_t b_alpha[4];                         /* boundary assertions in get_assertions order, src/air.rs:175-184 */
    uint64_t b_beta[4];
} cstark_tx_coeffs;

typedef struct cstark_ctx cstark_ctx;

/* ---- context ------------------------------------------------------------------------------- */
/* device < 0: current HIP device.  stream: the hipStream_t every launch of this context goes to
 * (NULL = HIP's default stream).  Work is asynchronous; cstark_ctx_synchronize() waits for it. */
int cstark_ctx_create(int device, void *stream, cstark_ctx **out);
/* The same with a stream of the context's own (non-blocking, destroyed with the context): for callers that do not link HIP, and
 * for several contexts working side by side -- contexts that share a stream (e.g. the default one) serialise their work on it. */
int cstark_ctx_create_own_stream(int device, cstark_ctx **out);
void cstark_ctx_destroy(cstark_ctx *ctx);
const char *cstark_last_error(void);
const char *cstark_version(void);
int cstark_ctx_synchronize(cstark_ctx *ctx);

/* ---- K1: execution trace (replaces TransactionProver::build_trace, src/prover.rs:37-98) ---- */
/* Copies the witness to device memory owned by the context (host -> HBM, ~2.3 KB / tx). */
int cstark_tx_witness_upload(cstark_ctx *ctx, const cstark_tx_witness *w);
/* Fills the 94 x (1024*n_tx) column-major trace at d_trace from the uploaded witness. */
int cstark_tx_build_trace(cstark_ctx *ctx, uint64_t *d_trace);

/* ---- K2/K3: low-degree extension (engine: trace.extend) ------------------------------------- */
/* Field conventions of the engine [UPSTREAM-RECALL]: multiplicative generator (domain offset) and
 * the primitive 2^log_n-th root of unity, in memory form. */
uint64_t cstark_field_generator(void);   /* multiplicative generator of the field (CSTARK_CONV_GENERATOR) */
uint64_t cstark_field_lde_offset(void);  /* domain offset of the STARK's LDE domain (CSTARK_CONV_LDE_OFFSET; the generator unless flipped) */
uint64_t cstark_field_root_of_unity(uint32_t log_n);
/* d_evals: width x n evaluations over the trace domain <w_n> (natural order); it is used as scratch
 * and destroyed.  d_coeffs (distinct buffer): width x n polynomial coefficients, natural order. */
int cstark_interpolate_columns(cstark_ctx *ctx, uint64_t *d_evals, uint64_t *d_coeffs, uint32_t width, uint32_t log_n);
/* d_coeffs: width x n coefficients.  d_lde: cosets [k0, k0+nk) of the blowup-times larger domain,
 * coset-major:  d_lde[((k - k0) * width + c) * n + j] = f_c(offset * w_{bn}^k * w_n^j),
 * i.e. natural LDE-domain index i = b*j + k.  offset = cstark_field_lde_offset() for the STARK domain.
 * Sharding by coset is what distributes one proof over several GPUs. */
int cstark_lde_columns(cstark_ctx *ctx, const uint64_t *d_coeffs, uint64_t *d_lde, uint32_t width, uint32_t log_n,
                       uint32_t log_blowup, uint64_t domain_offset, uint32_t k0, uint32_t nk);

/* FieldExtension::Quadratic / Cubic: the same three stages over the degree-m extension, m = 2: F_p[u]/(u^2 - 2u - 2), m = 3:
 * F_p[v]/(v^3 + v + 1) [assumption: the two polynomials of the reference's own curve tower, src/utils/ecc.rs:407-648; the fork's
 * choices for f63 are not in the tree].  An element is m consecutive base elements (coefficients of 1, x, x^2).  The trace stays
 * in the base field.
 *   cstark_evaluate_polys_at_ext: base-coefficient columns at one point; out[c][m] on the host.
 *   cstark_deep_composition_ext: all cosets; d_comp_lde holds m n_comp base columns per coset (column m i + k = component k of
 *     composition column i); coefficient / OOD arrays are host arrays of m-tuples; d_out = [m][b][n], component-major.
 *   cstark_fri_fold4_ext / cstark_fri_fold_ext: d_evals = [m][N] component-major -> d_out = [m][N/4] / [m][N/folding_factor]. */
int cstark_evaluate_polys_at_ext(cstark_ctx *ctx, const uint64_t *d_coeffs, uint32_t width, uint32_t log_n, uint32_t m, const uint64_t *z, uint64_t *out);
int cstark_deep_composition_ext(cstark_ctx *ctx, const uint64_t *d_trace_lde, const uint64_t *d_comp_lde, uint32_t width, uint32_t n_comp, uint32_t m,
                                const uint64_t *z, const uint64_t *ood_trace, const uint64_t *ood_comp, const uint64_t *alpha,
                                const uint64_t *beta, const uint64_t *delta, const uint64_t *deg_a, const uint64_t *deg_b, uint64_t *d_out,
                                uint32_t log_n, uint32_t log_blowup);
int cstark_fri_fold4_ext(cstark_ctx *ctx, const uint64_t *d_evals, uint64_t *d_out, uint32_t log_n, uint64_t domain_offset, uint32_t m,
                         const uint64_t *alpha);
int cstark_fri_fold_ext(cstark_ctx *ctx, const uint64_t *d_evals, uint64_t *d_out, uint32_t log_n, uint32_t folding_factor, uint64_t domain_offset,
                        uint32_t m, const uint64_t *alpha);

/* ---- K4/K5: Blake3 row hashing + Merkle tree (engine: build_commitment) ---------------------- */
/* Hash row j of coset k (width elements, 8 bytes LE each, memory form) into leaf i = b*j + k:
 *   d_leaves[32 * i .. 32 * i + 32]. */
int cstark_hash_rows(cstark_ctx *ctx, const uint64_t *d_lde, uint8_t *d_leaves, uint32_t width,
                     uint32_t log_n, uint32_t log_blowup, uint32_t k0, uint32_t nk);
/* d_nodes: 2 * n_leaves digests; nodes[n_leaves + i] = leaf i on input; on return nodes[1] is the
 * root and nodes[i] = Blake3(nodes[2i] || nodes[2i+1]). nodes[0] is zero. */
int cstark_merkle_build(cstark_ctx *ctx, uint8_t *d_nodes, uint32_t log_leaves);
/* The same two stages with the hash function chosen as in cstark_options::hash_fn: 0 = Blake3_256 (identical to the calls above),
 * 1 = Sha3_256 (FIPS 202; a row is absorbed as its 8-byte little-endian words, a parent is SHA3-256 of its two 32-byte children). */
int cstark_hash_rows_fn(cstark_ctx *ctx, uint32_t hash_fn, const uint64_t *d_lde, uint8_t *d_leaves, uint32_t width, uint32_t log_n,
                        uint32_t log_blowup, uint32_t k0, uint32_t nk);
int cstark_merkle_build_fn(cstark_ctx *ctx, uint32_t hash_fn, uint8_t *d_nodes, uint32_t log_leaves);

/* ---- K6/K7: constraint evaluation (Air::evaluate_transition, src/air.rs:114-173, + driver) --- */
/* d_lde: cosets [k0,k0+nk) of the extended 94-column trace, coset-major as cstark_lde_columns writes them; 16-byte aligned.
 * merkle_depth selects the mask columns (src/merkle/constants.rs:21-27).  log_blowup must be 3.
 *
 * All 115 transition-constraint values at every point:  d_out[((k - k0) * 115 + i) * n + j].
 * Parity / debugging entry point (the production path never materialises them). */
int cstark_tx_evaluate_transitions(cstark_ctx *ctx, const uint64_t *d_lde, uint64_t *d_out, uint32_t merkle_depth,
                                   uint32_t log_n, uint32_t log_blowup, uint32_t k0, uint32_t nk);
/* Fused production path: combined constraint evaluations
 *   d_out[(k - k0) * n + j] = sum_i (alpha_i + beta_i x^adj_i) C_i(x) / Z(x) + boundary terms,
 * x = g * w_{bn}^k * w_n^j.  pub_inputs = initial_root[0..2], final_root[0..2] (src/air.rs:175-184). */
int cstark_tx_evaluate_constraints(cstark_ctx *ctx, const uint64_t *d_lde, const cstark_tx_coeffs *coeffs,
                                   const uint64_t pub_inputs[4], uint64_t *d_out, uint32_t merkle_depth,
                                   uint32_t log_n, uint32_t log_blowup, uint32_t k0, uint32_t nk);
/* The same result for a table that IS the low-degree extension of 94 columns of degree < n over all 8 cosets (k0 = 0, nk = 8), e.g.
 * the output of cstark_lde_columns: allows the degree-split evaluation the prover uses (every part except the final
 * addition on the even cosets only, their merged polynomials extended to the odd cosets).  On such a table the output equals
 * cstark_tx_evaluate_constraints bit for bit; on any other table it is undefined (use cstark_tx_evaluate_constraints). */
int cstark_tx_evaluate_constraints_lde(cstark_ctx *ctx, const uint64_t *d_lde, const cstark_tx_coeffs *coeffs,
                                       const uint64_t pub_inputs[4], uint64_t *d_out, uint32_t merkle_depth, uint32_t log_n);
/* The same for m = 1..3 coefficient sets in ONE pass over the frame (the components of a FieldExtension::Quadratic / Cubic
 * proof: extension coefficients multiply base-field constraint values, so the values are computed once and merged m times).
 * coeffs: m consecutive blocks; d_out[(q * nk + (k - k0)) * n + j] = the merged evaluations for block q. */
int cstark_tx_evaluate_constraints_ext(cstark_ctx *ctx, const uint64_t *d_lde, const cstark_tx_coeffs *coeffs, uint32_t m,
                                       const uint64_t pub_inputs[4], uint64_t *d_out, uint32_t merkle_depth,
                                       uint32_t log_n, uint32_t log_blowup, uint32_t k0, uint32_t nk);
/* Measurement aid: when enabled, cstark_tx_evaluate_constraints records HIP events around each of its 9 launches
 * (Rescue windows; doubling / mixed addition of s*G; of h*P; final addition; three linear groups) on the context's
 * stream; cstark_tx_constraint_part_ms waits for the last one and returns the 9 durations in milliseconds. */
/* Inside cstark_tx_prove (base field) the parts run as the degree-split evaluation: every part except the final
 * addition on the even cosets only; the last figure (third linear group) then also holds the extension of the 13 merged
 * polynomials (11 + the final addition's 2) to the odd cosets and the recombination over all cosets. */
int cstark_ctx_set_part_timing(cstark_ctx *ctx, int enable);
int cstark_tx_constraint_part_ms(cstark_ctx *ctx, float *ms /* [9] */);
/* With part timing enabled every low-degree extension on this context (cstark_lde_columns and the prover's own extensions: trace,
 * composition columns, split polynomials, DEEP) is bracketed by a HIP event pair on the context's stream.  Returns the summed
 * duration and the number of evaluations written (columns x cosets x n) since the previous call, and resets both. */
int cstark_lde_timing_ms(cstark_ctx *ctx, float *total_ms, uint64_t *elements);
/* Host-side AIR description (no GPU needed): degree (base; number of 1024-row cycles) of transition constraint i
 * (TransactionAir::new, src/air.rs:76-108) and the 48 periodic columns (src/air.rs:194-380), [48][1024]. */
int cstark_tx_constraint_degree(uint32_t i, uint32_t *base, uint32_t *cycles);  /* CSTARK_AIR_STATE_TRANSITION */
int cstark_tx_periodic_columns(uint32_t merkle_depth, uint64_t *out);

/* ---- composition polynomial (engine: evaluations -> H(x) -> column split; first of the "next" rows) ----------
 * d_combined: [b][n] combined constraint evaluations, coset-major (output of cstark_tx_evaluate_constraints with all b
 * cosets).  d_cols: [b][n] coefficients of the column polynomials H_i, H(x) = sum_i x^i H_i(x^b); extend and commit
 * them with cstark_lde_columns / cstark_hash_rows (width b) / cstark_merkle_build. */
int cstark_composition_columns(cstark_ctx *ctx, const uint64_t *d_combined, uint64_t *d_cols, uint32_t log_n, uint32_t log_blowup);

/* Out-of-domain frame: values of `width` coefficient columns (device) at up to 16 points (host); out[p][c] on the host. */
int cstark_evaluate_polys_at(cstark_ctx *ctx, const uint64_t *d_coeffs, uint32_t width, uint32_t log_n, const uint64_t *points,
                             uint32_t npts, uint64_t *out);
/* DEEP composition over LDE cosets [k0,k0+nk):
 *   d_out[(k-k0)*n + j] = [ sum_c alpha_c (T_c(x)-T_c(z))/(x-z) + beta_c (T_c(x)-T_c(z w))/(x-z w)
 *                         + sum_i delta_i (H_i(x)-H_i(z^n_comp))/(x-z^n_comp) ] * (deg_a + deg_b x).
 * ood_trace = T(z)[width] | T(z w)[width], ood_comp = H_i(z^n_comp); all coefficient / OOD arrays are host memory. */
int cstark_deep_composition(cstark_ctx *ctx, const uint64_t *d_trace_lde, const uint64_t *d_comp_lde, uint32_t width, uint32_t n_comp,
                            uint64_t z, const uint64_t *ood_trace, const uint64_t *ood_comp, const uint64_t *alpha,
                            const uint64_t *beta, const uint64_t *delta, uint64_t deg_a, uint64_t deg_b, uint64_t *d_out,
                            uint32_t log_n, uint32_t log_blowup, uint32_t k0, uint32_t nk);

/* FRI (FriOptions::folding_factor f = 4, 8 or 16; the reference's examples pass -f, examples/state-transition.rs:46-47).
 * cstark_interleave_cosets: [b][n] coset-major -> natural LDE order (i = b*j + k).
 * cstark_fri_fold: N = 2^log_n evaluations over domain_offset * <w_N> (natural order) -> N/f evaluations of the
 * alpha-folded polynomial over domain_offset^f * <w_{N/f}>.  A layer is committed by viewing its N evaluations as the
 * f x (N/f) column-major table of rows { e[i + t N/f] } and calling cstark_hash_rows (width f, log_blowup 0) and
 * cstark_merkle_build.  cstark_fri_fold4 = cstark_fri_fold with f = 4. */
int cstark_interleave_cosets(cstark_ctx *ctx, const uint64_t *d_coset_major, uint64_t *d_natural, uint32_t log_n, uint32_t log_blowup);
int cstark_fri_fold4(cstark_ctx *ctx, const uint64_t *d_evals, uint64_t *d_out, uint32_t log_n, uint64_t domain_offset, uint64_t alpha);
int cstark_fri_fold(cstark_ctx *ctx, const uint64_t *d_evals, uint64_t *d_out, uint32_t log_n, uint32_t folding_factor, uint64_t domain_offset,
                    uint64_t alpha);

/* ---- whole proof (replaces TransactionExample::prove, src/lib.rs:116-141 = build_trace + Prover::prove) ------------
 * Proves the uploaded witness (cstark_tx_witness_upload) under `opt` and writes the serialised proof to `proof`
 * (host, `capacity` bytes; cstark_tx_proof_size_bound gives a sufficient capacity).  *proof_len receives the length; if the
 * buffer is too small the call fails with CSTARK_ERR_INVALID_ARG and *proof_len still holds the required size.
 * The public inputs are read from the trace as TransactionProver::get_pub_inputs does (src/prover.rs:106-129).
 * Supported options (the values the reference's own tests, benches and command line pass: src/lib.rs:78-86, src/merkle/update/tests.rs:41-52,
 * src/range/tests.rs:87-98, benches/rescue.rs:370-378, examples/state-transition.rs:33-34, :46-47): blowup_factor 2, 4, 8 or 16 and at least
 * the AIR's constraint-evaluation blowup (TransactionAir / SchnorrAir 8, MerkleAir 4, RangeProofAir 2 -- below it the engine refuses the
 * options too); Blake3_256 or Sha3_256; FieldExtension::None / Quadratic / Cubic; fri_folding_factor 4, 8 or 16; fri_max_remainder 128..1024;
 * num_queries 1..128; grinding_factor 0..32.  The sharded entry points and cstark_range_prove_batch's one-launch-per-stage path take
 * blowup 8 / folding 4 (the batch call falls back to one proof at a time for other values).
 *
 * Proof layout (little-endian; field elements as 8-byte memory form; this library's own format, the engine's
 * StarkProof::to_bytes layout is not available in the reference tree):
 *   "CSTK" u32 version | u32 air, trace_width, log2(trace_length), merkle_depth | u32 x 7 options
 *   trace_root[32] constraint_root[32] | u32 n_layers, layer_root[n_layers][32], remainder_commitment[32]
 *   T(z)[94] T(z w)[94] H_i(z^8)[8] | u64 pow_nonce                                 (8 = the AIR's composition columns: 8 / 4 / 8 / 2)
 *   trace rows [q][94], paths [q][log N][32] | composition rows [q][8], paths [q][log N][32]      (q = num_queries, N = blowup * n)
 *   per layer: u32 n_positions, rows [n_positions][f], paths [n_positions][log rows][32]          (f = fri_folding_factor, rows = N_layer / f)
 *   u32 remainder_len, remainder[remainder_len]
 * Query positions are not stored: the verifier re-derives them from the channel.  Paths list siblings leaf-upwards. */
#define CSTARK_PROOF_VERSION 1
#define CSTARK_PROVE_NUM_STAGES 10
int cstark_tx_prove(cstark_ctx *ctx, const cstark_options *opt, uint8_t *proof, size_t capacity, size_t *proof_len);
/* The same for the standalone AIRs (MerkleExample / SchnorrExample / RangeProofExample ::prove, src/merkle/update/mod.rs:84-107,
 * src/schnorr/mod.rs:150-173, src/range/mod.rs:78-101): CSTARK_AIR_MERKLE_UPDATE proves the uploaded transaction witness with the
 * 65-register MerkleAir, CSTARK_AIR_SCHNORR the uploaded Schnorr witness, CSTARK_AIR_RANGE the field element `number` (ignored
 * otherwise).  Same proof layout with the AIR's width and its number of composition columns (4 / 8 / 2); the header's 4th word
 * holds the Merkle depth / the number of signatures / 0.  Use cstark_tx_proof_size_bound(rows / 1024 rounded up, opt) * 2 as capacity. */
int cstark_air_prove(cstark_ctx *ctx, int air, const cstark_options *opt, uint64_t number, uint8_t *proof, size_t capacity, size_t *proof_len);
size_t cstark_tx_proof_size_bound(uint32_t n_tx, const cstark_options *opt);
/* RescueExample::prove (benches/rescue.rs:66-86, options :370-378: blowup 4): a chain of chain_length Rescue hashes from `seed` (7
 * elements, memory form; the bench uses 42..48); chain_length a power of two, 8 .. 2^21; the trace is 14 x 8 chain_length.  Same proof
 * layout (4 composition columns; the header's 4th word holds chain_length); public inputs = seed and result, read from the trace as
 * RescueProver::get_pub_inputs does (:331-354).  cstark_tx_proof_size_bound(chain_length / 128 rounded up, opt) is a sufficient capacity. */
int cstark_rescue_prove(cstark_ctx *ctx, const cstark_options *opt, const uint64_t seed[7], uint32_t chain_length, uint8_t *proof, size_t capacity,
                        size_t *proof_len);
/* RescueProver::build_trace (benches/rescue.rs:277-322) and the AIR's 29 periodic columns [29][8] (host; :245-249) */
int cstark_rescue_chain_build_trace(cstark_ctx *ctx, const uint64_t seed[7], uint32_t chain_length, uint64_t *d_trace);
int cstark_rescue_chain_periodic_columns(uint64_t *out);

/* ---- one proof across several GPUs, sharded by LDE coset (the reference's only parallel axis is the rayon loop
 * over trace fragments, src/prover.rs:50-52) -------------------------------------------------------------------------------------
 * Every rank (one process per GPU, one cstark_ctx each) uploads the same witness and calls the phases in this order; the caller
 * moves three device buffers between the ranks -- RCCL collectives, see certificate-stark_amd/sharding.py.  Rank r of W in {2, 4, 8}
 * owns the nk = 8 / W cosets [k0, k0 + nk), k0 = r nk.  The proof bytes equal cstark_tx_prove's bit for bit.
 *   1 cstark_tx_shard_commit     trace + interpolation (replicated), extension and row hashes of the rank's cosets.  The rank's nk leaves
 *                                of a row are a complete subtree of the trace tree: it hashes the bottom log2(nk) levels itself and hands
 *                                over the subtree roots, d_leaves_local [n][32] -> all-gather -> d_leaves_all [W][n][32] (rank-major):
 *                                32 n bytes per rank at every world size (34 MB at 2^20 steps)
 *   2 cstark_tx_shard_evaluate   trace tree + root (every rank: the channel is replayed everywhere), coefficients, the rank's share of the
 *                                merged constraint evaluations: d_combined_local [R][n], R = cstark_tx_shard_rows(nk)
 *                                -> all-gather -> d_combined_all [W][R][n].  W = 8: R = 1, the rank's coset evaluated point by point.
 *                                W = 2, 4: the degree-split evaluation, sharded -- R = nk / 2 + 4: the rank's nk / 2 even cosets
 *                                (complete), then ITS SHARE of each of the four odd cosets (the split polynomials are evaluated on the
 *                                rank's even cosets only; their extension to the odd cosets is linear, so the ranks' shares add up)
 *   3 cstark_tx_shard_compose    rank 0 only (it owns coset 0, which the DEEP composition reads): sums the shares into the merged
 *                                evaluations of all cosets, composition polynomial and its commitment, out-of-domain frame, DEEP, FRI;
 *                                positions[num_queries] (host) -> broadcast
 *   4 cstark_tx_shard_open_rows  every rank: the opened rows of the extended trace that lie in its cosets, each followed by the bottom
 *                                log2(nk) siblings of its authentication path (the levels only the owner holds), zeros elsewhere;
 *                                d_rows [nq][cstark_tx_shard_open_words(nk)] -> all-reduce (sum) -> complete rows and path bottoms
 *   5 cstark_tx_shard_finish     rank 0: paths, remaining openings, proof bytes. */
uint32_t cstark_tx_shard_rows(uint32_t nk); /* rows of n evaluations a rank with nk cosets contributes in phase 2 (0: invalid nk) */
uint32_t cstark_tx_shard_open_words(uint32_t nk); /* 64-bit words per query in phase 4: 94 + 4 log2(nk) (0: invalid nk) */
int cstark_tx_shard_commit(cstark_ctx *ctx, const cstark_options *opt, uint32_t k0, uint32_t nk, uint8_t *d_leaves_local);
/* rows / total_rows: the row counts the caller sized d_combined_local ([rows][n]) and d_combined_all ([total_rows][n]) by; a value other
 * than cstark_tx_shard_rows(nk) / W * cstark_tx_shard_rows(nk) is refused before anything is written (version 0.2 of this library). */
int cstark_tx_shard_evaluate(cstark_ctx *ctx, const uint8_t *d_leaves_all, uint64_t *d_combined_local, uint32_t rows);
int cstark_tx_shard_compose(cstark_ctx *ctx, const uint64_t *d_combined_all, uint32_t total_rows, uint32_t *positions);
int cstark_tx_shard_open_rows(cstark_ctx *ctx, const uint32_t *positions, uint32_t nq, uint64_t *d_rows);
int cstark_tx_shard_finish(cstark_ctx *ctx, const uint64_t *d_rows, uint8_t *proof, size_t capacity, size_t *proof_len);
/* Wall-clock of the stages of the last cstark_tx_prove on this context (HIP events on its stream), milliseconds:
 * trace, interpolate, LDE, row hashes + tree, constraint evaluation, composition polynomial + commitment,
 * out-of-domain frame, DEEP composition, FRI layers, query openings.
 * cstark_tx_prove runs the curve ladders of the trace (registers 0..36) on an internal stream beside the interpolation AND
 * extension of the other 57 registers; its first three figures are therefore: trace = the rest of the trace on the context's
 * stream; interpolate = registers 37..93 interpolated and extended, the wait for the ladders, registers 0..36 interpolated;
 * LDE = extension of registers 0..36.  Their sum is the time from the start of the proof to the complete extended trace. */
int cstark_prove_stage_ms(cstark_ctx *ctx, float *ms /* [CSTARK_PROVE_NUM_STAGES] */);

/* ---- standalone sub-AIRs (reference src/merkle/update, src/range; BASELINE configs 1-2) ---------- */
/* MerkleProver::build_trace (src/merkle/update/prover.rs:28-80): 65 x (512*n_tx) from the uploaded witness. */
int cstark_merkle_build_trace(cstark_ctx *ctx, uint64_t *d_trace);
/* RangeProver::build_trace (src/range/prover.rs:24-43): 2 x 64; `number` is a field element in memory form
 * whose canonical value must be below 2^63. */
int cstark_range_build_trace(cstark_ctx *ctx, uint64_t number, uint64_t *d_trace);
/* SYNTHETIC long form of the same accumulator for BASELINE.json's config "range-proof AIR, 2^16 steps" (the reference's range trace
 * is fixed at 64 rows, src/range/mod.rs:34, so this size has no reference counterpart): 2 x 2^log_n rows of the (n-1)-bit integer V
 * whose n/64 little-endian words are `words` (host; top bit of the last word clear).  Row q >= 1 holds bit (n-1-q) of V and
 * acc_q = 2 acc_(q-1) + bit (src/range/prover.rs:74-84), the AIR is RangeProofAir unchanged (src/range/air.rs:60-105) with the
 * assertion acc[n-1] = V mod p.  log_n = 6 with words[0] = number reproduces cstark_range_build_trace / cstark_air_prove exactly.
 * *number_out (optional): V mod p in memory form. */
int cstark_range_build_trace_bits(cstark_ctx *ctx, const uint64_t *words, uint32_t log_n, uint64_t *d_trace, uint64_t *number_out);
int cstark_range_prove_bits(cstark_ctx *ctx, const cstark_options *opt, const uint64_t *words, uint32_t log_n, uint8_t *proof, size_t capacity,
                            size_t *proof_len);
/* `count` reference-shaped range proofs (RangeProofExample::prove, src/range/mod.rs:75-100; the loop of benches/range.rs:15-37) in ONE
 * call: every stage is one launch over the batch, the host walks the `count` Fiat-Shamir channels between the stages.  numbers[count]:
 * field elements in memory form, canonical value below 2^63.  Proof t is written to proofs + t * stride (host memory;
 * cstark_tx_proof_size_bound(1, opt) is a sufficient stride), its length to lens[t]; it equals
 * cstark_air_prove(ctx, CSTARK_AIR_RANGE, opt, numbers[t], ...) byte for byte.  FieldExtension::None; both hash functions. */
int cstark_range_prove_batch(cstark_ctx *ctx, const cstark_options *opt, const uint64_t *numbers, uint32_t count, uint8_t *proofs, size_t stride,
                             size_t *lens);
/* SchnorrAir (src/schnorr/air.rs:41-300, src/schnorr/prover.rs:21-67): n signatures over 28-element messages
 * (message[0..12] = public key).  Trace 56 x (512*n); the 19 public-input columns (pkey x12, message chunks x7;
 * src/schnorr/air.rs:228-290) as a 19 x (512*n) table that the caller extends like trace columns; the 8 mask + 28
 * round-constant periodic columns [36][512] (host).  The 61 periodic / sequence assertions (:111-226) are merged by
 * cstark_air_combine. */
int cstark_schnorr_witness_upload(cstark_ctx *ctx, uint32_t n_sig, const uint64_t *messages, const uint64_t *sig_rx, const uint8_t *sig_s);
int cstark_schnorr_build_trace(cstark_ctx *ctx, uint64_t *d_trace);
int cstark_schnorr_aux_columns(cstark_ctx *ctx, uint64_t *d_out);
int cstark_schnorr_mask_columns(uint64_t *out /* [36][512] host */);
int cstark_schnorr_evaluate_transitions(cstark_ctx *ctx, const uint64_t *d_lde, const uint64_t *d_aux_lde, uint64_t *d_out,
                                        uint32_t log_n, uint32_t log_blowup, uint32_t k0, uint32_t nk);
/* Coefficient columns [12][n] of the value polynomials of SchnorrAir's sequence assertions (R.x asserted at step 0 and at
 * step 511 of every block, src/schnorr/air.rs:172-224); extend them with cstark_lde_columns and pass them to
 * cstark_air_combine. */
int cstark_schnorr_assertion_polys(cstark_ctx *ctx, uint64_t *d_out, uint32_t log_n);
/* Shape of an AIR as the engine sees it (host side): trace width, number of transition constraints and of
 * assertions, log2 of the constraint-evaluation blowup; degree (base; cycles) of constraint i.  n_items = number of
 * signatures for CSTARK_AIR_SCHNORR (its degrees depend on it, src/schnorr/air.rs:537), ignored otherwise. */
int cstark_air_shape(int air, uint32_t n_items, uint32_t *width, uint32_t *n_constraints, uint32_t *n_assertions, uint32_t *log_ce_blowup);
int cstark_air_constraint_degree(int air, uint32_t n_items, uint32_t i, uint32_t *base, uint32_t *cycles);
int cstark_merkle_periodic_columns(uint32_t merkle_depth, uint64_t *out /* [33][512] host */);
/* All transition constraints of AIR `air` (CSTARK_AIR_MERKLE_UPDATE, CSTARK_AIR_RANGE) on LDE cosets [k0,k0+nk):
 *   d_out[((k - k0) * n_constraints + i) * n + j].  (SchnorrAir: cstark_schnorr_evaluate_transitions.) */
int cstark_air_evaluate_transitions(cstark_ctx *ctx, int air, const uint64_t *d_lde, uint64_t *d_out, uint32_t merkle_depth,
                                    uint32_t log_n, uint32_t log_blowup, uint32_t k0, uint32_t nk);
/* Generic merge of materialised transition evaluations with the AIR's assertions (single, periodic and sequence) into
 * the combined constraint evaluations d_out[(k - k0) * n + j].  Coefficient arrays (per transition constraint / per
 * assertion in get_assertions order) are host memory.  assertion_values: host, one per assertion, for Merkle (7+7 roots)
 * and Range (0, number); NULL for Schnorr whose constants are built in and whose sequence values arrive as
 * d_avals_lde = [nk][n_avals][n] (extension of cstark_schnorr_assertion_polys).  The constraint-evaluation domain may be
 * smaller than the LDE domain (MerkleAir: blowup 4); cosets outside it are written as 0. */
int cstark_air_combine(cstark_ctx *ctx, int air, uint32_t n_items, const uint64_t *d_lde, const uint64_t *d_evals,
                       const uint64_t *t_alpha, const uint64_t *t_beta, const uint64_t *b_alpha, const uint64_t *b_beta,
                       const uint64_t *assertion_values, const uint64_t *d_avals_lde, uint32_t n_avals, uint64_t *d_out,
                       uint32_t log_n, uint32_t log_blowup, uint32_t k0, uint32_t nk);

/* SchnorrAir's combined constraint evaluations WITHOUT the 56 materialised transition values: the transition sum
 * sum_i (alpha_i + beta_i x^adj_i) C_i(x) is accumulated per point by the fused gadget evaluators (the ones behind
 * cstark_tx_evaluate_constraints) and merged with the 61 assertions like cstark_air_combine does.  Same values as
 * cstark_schnorr_evaluate_transitions followed by cstark_air_combine (exact arithmetic); what cstark_air_prove uses for
 * CSTARK_AIR_SCHNORR.  Replaces the evaluation loop of winterfell's ConstraintEvaluator for SchnorrAir
 * (/root/reference/src/schnorr/air.rs:72-109, :394-531) [UPSTREAM-RECALL for the driver]. */
/* MerkleAir likewise (one launch over the cosets of its constraint-evaluation domain, blowup 4; src/merkle/update/air.rs:64-141,
 * :215-369): same values as cstark_air_evaluate_transitions + cstark_air_combine; what cstark_air_prove uses for
 * CSTARK_AIR_MERKLE_UPDATE.  assertion_values: the 7 + 7 root elements (host). */
int cstark_merkle_evaluate_constraints(cstark_ctx *ctx, uint32_t merkle_depth, const uint64_t *d_lde, const uint64_t *t_alpha,
                                       const uint64_t *t_beta, const uint64_t *b_alpha, const uint64_t *b_beta,
                                       const uint64_t *assertion_values, uint64_t *d_out, uint32_t log_n, uint32_t log_blowup,
                                       uint32_t k0, uint32_t nk);
int cstark_schnorr_evaluate_constraints(cstark_ctx *ctx, uint32_t n_sig, const uint64_t *d_lde, const uint64_t *d_aux_lde,
                                        const uint64_t *t_alpha, const uint64_t *t_beta, const uint64_t *b_alpha, const uint64_t *b_beta,
                                        const uint64_t *d_avals_lde, uint32_t n_avals, uint64_t *d_out, uint32_t log_n,
                                        uint32_t log_blowup, uint32_t k0, uint32_t nk);
/* The same result for tables that ARE low-degree extensions over all 8 cosets (d_lde: 56 columns of degree < n, d_aux_lde: the 19
 * public-input columns; e.g. outputs of cstark_lde_columns): allows the degree-split evaluation of the doubling / addition gadgets that
 * cstark_air_prove uses (on the even cosets only, their eight merged polynomials extended to the odd cosets).  On such tables
 * the output equals cstark_schnorr_evaluate_constraints bit for bit; on any other table it is undefined. */
int cstark_schnorr_evaluate_constraints_lde(cstark_ctx *ctx, uint32_t n_sig, const uint64_t *d_lde, const uint64_t *d_aux_lde,
                                            const uint64_t *t_alpha, const uint64_t *t_beta, const uint64_t *b_alpha, const uint64_t *b_beta,
                                            const uint64_t *d_avals_lde, uint32_t n_avals, uint64_t *d_out, uint32_t log_n);

/* ---- witness synthesis (host; counterpart of TransactionMetadata::build_random, src/lib.rs:235-465, and of
 * SchnorrExample::new + schnorr::sign, src/schnorr/mod.rs:86-141, :197-217; seeded and deterministic) -----------------
 * cstark_tx_witness_generate fills the arrays of `w` (every pointer caller-allocated with the sizes documented on
 * cstark_tx_witness; n_tx and merkle_depth set by the caller).  Like the reference this is CPU work outside the timed
 * region.  Signatures are produced with integer arithmetic and small secret keys (see csrc/witness_gen.hip). */
int cstark_tx_witness_generate(cstark_tx_witness *w, uint64_t seed);
/* n_sig messages [n][28] (public key || 16 elements) with their signatures (R.x [n][6], s [n][32]). */
int cstark_schnorr_witness_generate(uint32_t n_sig, uint64_t seed, uint64_t *messages, uint64_t *sig_rx, uint8_t *sig_s);

/* ---- device memory helpers for callers without a HIP runtime of their own (the Rust shim) ---- */
int cstark_malloc(cstark_ctx *ctx, size_t bytes, void **d_ptr);
int cstark_free(cstark_ctx *ctx, void *d_ptr);
int cstark_memcpy_h2d(cstark_ctx *ctx, void *d_dst, const void *src, size_t bytes);
int cstark_memcpy_d2h(cstark_ctx *ctx, void *dst, const void *d_src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* CSTARK_H */

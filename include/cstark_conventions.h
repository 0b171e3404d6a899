/* cstark_conventions.h -- every choice this library makes ON BEHALF OF THE ABSENT ENGINE, in one place.
 *
 * The reference crate delegates field, transforms, hashing of rows, constraint-evaluation driver, DEEP / FRI, the Fiat-Shamir
 * channel and the proof format to a git dependency that is not in its tree (winterfell fork, Cargo.toml:20, rev 8e37310).  What the
 * reference itself fixes (AIR, trace, masks, Rescue, curve) is restated from its sources and is NOT configurable here.  What only the
 * engine fixes is recalled from upstream winterfell v0.3 [UPSTREAM-RECALL] or, for the extension fields, assumed -- parity with the
 * real engine is UNPINNED until someone diffs against a Rust run (INTEGRATION.md section 5 says which diff flips which line).
 *
 * This header is plain C preprocessor: the product (certificate-stark_amd/csrc, C++/HIP) and the CPU oracle (oracle/, C; its Python
 * half reads the values back through cso_conventions()) include the SAME file, so flipping a line and rebuilding moves both sides
 * together and the parity suite stays meaningful.  Every value can also be overridden with -D for both builds at once:
 * tools/flip_conventions.py does that and re-runs the GPU-against-oracle parity tests under the flipped conventions.
 */
#ifndef CSTARK_CONVENTIONS_H
#define CSTARK_CONVENTIONS_H

/* ---- field f63: p = 2^62 + 2^56 + 2^55 + 1 (fixed by the reference: src/range/tests.rs:59) ---------------------------------------- */
#ifndef CSTARK_CONV_FIELD_GENERATOR
#define CSTARK_CONV_FIELD_GENERATOR 3        /* multiplicative generator of F_p* (the smallest primitive root) */
#endif
#ifndef CSTARK_CONV_TWO_ADIC_ROOT_EXP
#define CSTARK_CONV_TWO_ADIC_ROOT_EXP 131    /* the 2^55-th root of unity is GENERATOR^131, 131 = (p - 1) / 2^55; get_root_of_unity(k) =
                                                that root squared 55 - k times */
#endif
/* ---- domains ------------------------------------------------------------------------------------------------------------------- */
#ifndef CSTARK_CONV_LDE_OFFSET
#define CSTARK_CONV_LDE_OFFSET CSTARK_CONV_FIELD_GENERATOR /* the LDE / constraint-evaluation domain is OFFSET * <w_(blowup n)> */
#endif
/* ---- bytes of a field element wherever elements are HASHED (rows of every committed table: trace, composition, FRI layers; the
 * out-of-domain frames; the FRI remainder): 1 = little-endian bytes of the in-memory form (Montgomery, R = 2^64: what as_bytes() of
 * a slice of BaseElement yields), 0 = little-endian bytes of the canonical value.  Public inputs are always canonical (src/air.rs:57-62
 * writes them through Serializable); the proof body stores memory form either way (this library's own format, cstark.h). */
#ifndef CSTARK_CONV_HASHED_ELEMENT_BYTES_MONTGOMERY
#define CSTARK_CONV_HASHED_ELEMENT_BYTES_MONTGOMERY 1
#endif
/* ---- constraint composition ------------------------------------------------------------------------------------------------------ */
/* composition polynomial degree for a constraint-evaluation domain of ce_size points */
#define CSTARK_CONV_COMPOSITION_DEGREE(ce_size) ((ce_size) - 1)
/* transition constraints vanish on all steps but the last CSTARK_CONV_TRANSITION_EXEMPTIONS: divisor (x^n - 1) / (x - w^(n-1)) */
#ifndef CSTARK_CONV_TRANSITION_EXEMPTIONS
#define CSTARK_CONV_TRANSITION_EXEMPTIONS 1
#endif
/* x^adjustment lifts a constraint of evaluation degree d to the composition degree before the division by the divisor (degree
 * n - EXEMPTIONS); each constraint contributes (alpha + beta x^adjustment) * value */
#define CSTARK_CONV_TRANSITION_ADJUSTMENT(ce_size, n, d) \
    (CSTARK_CONV_COMPOSITION_DEGREE(ce_size) + ((n) - CSTARK_CONV_TRANSITION_EXEMPTIONS) - (d))
/* boundary constraints on m steps (divisor of degree m) of a trace polynomial of degree n - 1 */
#define CSTARK_CONV_BOUNDARY_ADJUSTMENT(ce_size, n, m) (CSTARK_CONV_COMPOSITION_DEGREE(ce_size) + (m) - ((n) - 1))
/* ---- public coin ----------------------------------------------------------------------------------------------------------------- */
/* seed = H(context || public inputs); reseed(d) = H(seed || d); reseed_int(v) = H(seed || v_le64);
 * draw: counter += 1 starting at FIRST_COUNTER, H(seed || counter_le64), first 8 bytes little-endian as an integer */
#ifndef CSTARK_CONV_COIN_FIRST_COUNTER
#define CSTARK_CONV_COIN_FIRST_COUNTER 1
#endif
#ifndef CSTARK_CONV_COIN_REJECT_ABOVE_P
#define CSTARK_CONV_COIN_REJECT_ABOVE_P 1    /* 1: values >= p are skipped and the next counter is tried; 0: reduced mod p */
#endif
#ifndef CSTARK_CONV_QUERY_DEDUP
#define CSTARK_CONV_QUERY_DEDUP 1            /* 1: query positions are drawn until num_queries DISTINCT ones exist; 0: duplicates kept */
#endif
#ifndef CSTARK_CONV_DEEP_DRAWS_PER_REGISTER
#define CSTARK_CONV_DEEP_DRAWS_PER_REGISTER 3 /* DEEP coefficients drawn per trace register (alpha for z, beta for z w, one for the
                                                conjugate point that only extension fields use: drawn and discarded) */
#endif
/* ---- extension fields of f63 (FieldExtension::Quadratic / Cubic).  ASSUMED: the polynomials of the reference's own curve tower
 * (src/utils/ecc.rs:424-439, :506-548); the fork's choice for f63 is not in the tree.
 *   E2 = F_p[u] / (u^2 - E2_C1 u - E2_C0)          E3 = F_p[v] / (v^3 - E3_C2 v^2 - E3_C1 v - E3_C0)
 * small signed integers; the arithmetic of product, oracle and verifier is generic in them */
#ifndef CSTARK_CONV_E2_C0
#define CSTARK_CONV_E2_C0 2
#endif
#ifndef CSTARK_CONV_E2_C1
#define CSTARK_CONV_E2_C1 2
#endif
#ifndef CSTARK_CONV_E3_C0
#define CSTARK_CONV_E3_C0 (-1)
#endif
#ifndef CSTARK_CONV_E3_C1
#define CSTARK_CONV_E3_C1 (-1)
#endif
#ifndef CSTARK_CONV_E3_C2
#define CSTARK_CONV_E3_C2 0
#endif
/* ---- not configurable by a macro, listed for completeness (each is one function): order of the channel (prove.hip, header
 * comment), the proof byte layout (cstark.h, cstark_tx_prove), leaf = H(row bytes), node = H(left || right), nodes[1] = root, FRI
 * layer rows = the folding-factor evaluations that fold into one position, remainder committed as H(elements). */
#endif /* CSTARK_CONVENTIONS_H */

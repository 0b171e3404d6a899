// The Fiat-Shamir channel on the device (channel.hip): one launch of one workgroup per channel step -- absorb (reseed), draw, and the
// small derived values the next kernels need -- so that a whole proof is enqueued without the host reading anything back in between.
// Same bytes as the host coin of prove.hip (Coin): seed = Blake3(context || public inputs); reseed(d) = Blake3(seed || d);
// reseed_int(v) = Blake3(seed || le64(v)); draw: Blake3(seed || le64(counter))[0..8), counter from CSTARK_CONV_COIN_FIRST_COUNTER,
// rejected unless below p (CSTARK_CONV_COIN_REJECT_ABOVE_P).  Blake3 coin only.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace cs {

enum ChanAbsorb : uint32_t {
    CHAN_NONE = 0,
    CHAN_DIGEST = 1, // 32 bytes at ptr
    CHAN_ELEMS = 2,  // digest of `count` field elements at ptr (their hashed byte form: CSTARK_CONV_HASHED_ELEMENT_BYTES_MONTGOMERY)
    CHAN_INT = 3     // the 64-bit integer `value`
};
enum ChanDraw : uint32_t {
    CHAN_DRAW_NONE = 0,
    // field elements in memory form.  Where draw i goes:
    CHAN_DRAW_LINEAR = 1,  // out[i]
    CHAN_DRAW_COEFFS = 2,  // (alpha, beta) pairs of a transition constraints then b assertions -> alpha[i] at out[i], beta[i] at
                           // out[stride + i], assertion alphas at out[2 stride + i], betas at out[2 stride + b + i]  (cstark_tx_coeffs)
    CHAN_DRAW_DEEP = 3,    // per register `per` draws (alpha, beta, unused...), then b composition columns, then two: alpha[a] | beta[a] |
                           // delta[b] at out, the two degree-adjustment coefficients at out2[3], out2[4]
    CHAN_DRAW_POINT = 4,   // one draw z -> out[0] = z, out[1] = z w, out[2] = z^e (w, e below); also out2[0..3) = the same (DEEP scalars)
    CHAN_DRAW_QUERIES = 5  // `count` distinct integers below 2^log_domain -> positions; then their folded positions layer by layer
};

struct ChanStep {
    uint32_t *seed;            // [8] the coin (device); read unless `init`, always written
    // init: seed = Blake3(prefix || canonical little-endian words of pub[0..npub))      (one chunk: prefix_len + 8 npub <= 1024)
    uint32_t init, prefix_len, npub;
    uint8_t prefix[32];
    const uint64_t *pub;
    struct { uint32_t kind, count; const void *ptr; uint64_t value; uint8_t *copy_out; } absorb[3]; // in order; copy_out: the 32-byte digest absorbed (or null)
    uint32_t draw, count, a, b, stride, per;
    uint64_t *out, *out2;
    uint64_t w;                // CHAN_DRAW_POINT: w_n (memory form); e = b
    // CHAN_DRAW_QUERIES: positions -> pos[0..count); layer l < n_layers: the distinct values of (previous list mod 2^(log_domain - (l + 1) log_f)), first
    // occurrences in order -> pos[slot * (l + 1) ..), their number -> cnt[l + 1]; cnt[0] = count
    uint32_t log_domain, log_f, n_layers, slot;
    uint32_t *pos, *cnt;
};
// one workgroup of 1024 threads
hipError_t channel_step(const ChanStep &s, hipStream_t stream);

} // namespace cs

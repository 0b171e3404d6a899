// cstark.hpp -- C++ host-side mirror of the reference's prover interface over the C ABI of cstark.h.
//
// The reference is a Rust crate; no Rust toolchain exists in the build environment, so the host layer a Rust maintainer would
// write is provided in C++ with the reference's names, argument meaning and error behaviour:
//   ProofOptions            winterfell::ProofOptions::new(...) as used at /root/reference/src/lib.rs:78-86
//   TransactionMetadata     src/lib.rs:183-232 (field for field), ::build_random src/lib.rs:235-465 (seeded)
//   TransactionProver       src/prover.rs:20-134: new(options), build_trace, get_pub_inputs, prove
//   TransactionExample      src/lib.rs:92-150: new(options, num_transactions), prove()
//   get_example             src/lib.rs:75-89
//   MerkleExample / SchnorrExample / RangeProofExample    the sub-AIR examples (src/merkle/update/mod.rs, src/schnorr/mod.rs,
//                                                         src/range/mod.rs)
// Failures are exceptions (cstark::Error) where the reference returns Err / panics.  Header-only; link libcstark_hip.so.
#ifndef CSTARK_HPP
#define CSTARK_HPP
#include <array>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <future>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#include "cstark.h"

namespace cstark {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error("cstark error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int rc) {
    if (rc != CSTARK_OK) throw Error(rc, cstark_last_error());
}

using BaseElement = uint64_t; // memory form of f63::BaseElement (Montgomery, reduced)
using Hash = std::array<BaseElement, 7>;

enum class HashFunction : uint32_t { Blake3_256 = 0, Sha3_256 = 1 };
enum class FieldExtension : uint32_t { None = 0, Quadratic = 1, Cubic = 2 };

struct ProofOptions {
    uint32_t num_queries = 42, blowup_factor = 8, grinding_factor = 0;
    HashFunction hash_fn = HashFunction::Blake3_256;
    FieldExtension field_extension = FieldExtension::None;
    uint32_t fri_folding_factor = 4, fri_max_remainder = 256;
    ProofOptions() = default;
    ProofOptions(uint32_t q, uint32_t b, uint32_t g, HashFunction h, FieldExtension e, uint32_t f, uint32_t r)
        : num_queries(q), blowup_factor(b), grinding_factor(g), hash_fn(h), field_extension(e), fri_folding_factor(f), fri_max_remainder(r) {}
    cstark_options raw() const {
        return {num_queries, blowup_factor, grinding_factor, (uint32_t)hash_fn, (uint32_t)field_extension, fri_folding_factor, fri_max_remainder};
    }
};

struct PublicInputs { // src/air.rs:52-62
    Hash initial_root{}, final_root{};
};

// One GPU context (device, stream).  Not copyable; one per host thread.
class Context {
  public:
    explicit Context(int device = -1, void *stream = nullptr) { check(cstark_ctx_create(device, stream, &ctx_)); }
    struct OwnStream {}; // Context(OwnStream{}, device): a stream of the context's own (several contexts side by side: ProverPool)
    explicit Context(OwnStream, int device = -1) { check(cstark_ctx_create_own_stream(device, &ctx_)); }
    ~Context() { cstark_ctx_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    cstark_ctx *raw() const { return ctx_; }
    void synchronize() const { check(cstark_ctx_synchronize(ctx_)); }
    std::array<float, CSTARK_PROVE_NUM_STAGES> prove_stage_ms() const {
        std::array<float, CSTARK_PROVE_NUM_STAGES> ms{};
        check(cstark_prove_stage_ms(ctx_, ms.data()));
        return ms;
    }

  private:
    cstark_ctx *ctx_ = nullptr;
};

// Series of transfers in the account tree (src/lib.rs:183-194).  Paths are [leaf, sibling_0 .. sibling_{depth-1}].
struct TransactionMetadata {
    unsigned merkle_depth = 15;
    std::vector<Hash> initial_roots;
    Hash final_root{};
    std::vector<std::array<BaseElement, 14>> s_old_values, r_old_values;
    std::vector<uint64_t> s_indices, r_indices;
    std::vector<BaseElement> s_paths, r_paths; // [n][depth + 1][7]
    std::vector<BaseElement> deltas;
    std::vector<std::array<BaseElement, 6>> sig_rx;
    std::vector<std::array<uint8_t, 32>> sig_s;

    size_t len() const { return initial_roots.size(); }
    // "Enforce that all vectors are of equal length" (src/lib.rs:211-218)
    void validate() const {
        const size_t n = len();
        if (n == 0 || s_old_values.size() != n || r_old_values.size() != n || s_indices.size() != n || r_indices.size() != n || deltas.size() != n ||
            sig_rx.size() != n || sig_s.size() != n || s_paths.size() != n * (merkle_depth + 1) * 7 || r_paths.size() != s_paths.size())
            throw Error(CSTARK_ERR_INVALID_ARG, "transaction metadata vectors are not of equal length");
        if ((merkle_depth + 1) & merkle_depth) throw Error(CSTARK_ERR_INVALID_ARG, "tree depth must be one less than a power of 2"); // src/lib.rs:102-105
    }
    cstark_tx_witness view() const {
        cstark_tx_witness w{};
        w.n_tx = (uint32_t)len(); w.merkle_depth = merkle_depth;
        w.initial_roots = initial_roots[0].data(); w.final_root = final_root.data();
        w.s_old_values = s_old_values[0].data(); w.r_old_values = r_old_values[0].data();
        w.s_indices = s_indices.data(); w.r_indices = r_indices.data();
        w.s_paths = s_paths.data(); w.r_paths = r_paths.data();
        w.deltas = deltas.data(); w.sig_rx = sig_rx[0].data(); w.sig_s = sig_s[0].data();
        return w;
    }
    // TransactionMetadata::build_random (src/lib.rs:235-465), deterministic in `seed`
    static TransactionMetadata build_random(size_t num_transactions, unsigned depth = 15, uint64_t seed = 0x5EED) {
        TransactionMetadata m;
        const size_t n = num_transactions;
        m.merkle_depth = depth;
        m.initial_roots.resize(n); m.s_old_values.resize(n); m.r_old_values.resize(n); m.s_indices.resize(n); m.r_indices.resize(n);
        m.s_paths.resize(n * (depth + 1) * 7); m.r_paths.resize(n * (depth + 1) * 7); m.deltas.resize(n); m.sig_rx.resize(n); m.sig_s.resize(n);
        if (n == 0) throw Error(CSTARK_ERR_INVALID_ARG, "no transactions");
        cstark_tx_witness w = m.view();
        check(cstark_tx_witness_generate(&w, seed));
        return m;
    }
};

// src/prover.rs:20-134.  The trace lives in device memory (the reference's TraceTable is a host object); prove() is the
// whole of Prover::prove including trace generation.
class TransactionProver {
  public:
    TransactionProver(const ProofOptions &options, Context &ctx) : options_(options), ctx_(ctx) {}
    const ProofOptions &options() const { return options_; }

    // uploads the witness; the trace itself is built inside prove() / build_trace()
    void load(const TransactionMetadata &m) {
        m.validate();
        if (m.len() & (m.len() - 1)) throw Error(CSTARK_ERR_INVALID_ARG, "the number of transactions must be a power of two");
        const cstark_tx_witness w = m.view();
        check(cstark_tx_witness_upload(ctx_.raw(), &w));
        n_tx_ = m.len();
    }
    // 94 x (1024 n) column-major trace into caller-provided device memory (cstark_malloc)
    void build_trace(const TransactionMetadata &m, uint64_t *d_trace) {
        load(m);
        check(cstark_tx_build_trace(ctx_.raw(), d_trace));
    }
    std::vector<uint8_t> prove(const TransactionMetadata &m) {
        load(m);
        const cstark_options o = options_.raw();
        std::vector<uint8_t> proof(cstark_tx_proof_size_bound((uint32_t)n_tx_, &o));
        size_t len = 0;
        check(cstark_tx_prove(ctx_.raw(), &o, proof.data(), proof.size(), &len));
        proof.resize(len);
        return proof;
    }
    // src/prover.rs:106-129 (from the metadata: the first initial root and the final root)
    static PublicInputs get_pub_inputs(const TransactionMetadata &m) { return {m.initial_roots.at(0), m.final_root}; }

  private:
    ProofOptions options_;
    Context &ctx_;
    size_t n_tx_ = 0;
};

// Several proofs in flight on ONE GPU.  A proof is a chain of launches with a dozen host round trips (the Fiat-Shamir channel) during
// which the GPU idles (~0.6 ms of a 28 ms proof); a second proof on a context and host thread of its own fills those gaps: 36.7
// instead of 35.2 proofs/s at 2^20 steps (bench.py --inflight 2).  Every worker owns a Context with its own stream; submit() hands a
// witness to the next free worker and returns a future of the proof bytes (exceptions travel through the future).
class ProverPool {
  public:
    explicit ProverPool(const ProofOptions &options, unsigned workers = 2, int device = -1) : options_(options) {
        if (workers == 0) throw Error(CSTARK_ERR_INVALID_ARG, "ProverPool: at least one worker");
        for (unsigned w = 0; w < workers; w++) ctx_.emplace_back(new Context(Context::OwnStream{}, device));
        try {
            for (unsigned w = 0; w < workers; w++) threads_.emplace_back([this, w] { run(w); });
        } catch (...) { // a thread could not be created: stop and join the ones already running (a joinable std::thread must not be destroyed)
            { std::lock_guard<std::mutex> g(m_); stop_ = true; }
            cv_.notify_all();
            for (std::thread &t : threads_) t.join();
            throw;
        }
    }
    ~ProverPool() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        for (std::thread &t : threads_) t.join();
    }
    ProverPool(const ProverPool &) = delete;
    ProverPool &operator=(const ProverPool &) = delete;
    std::future<std::vector<uint8_t>> submit(TransactionMetadata m) {
        std::packaged_task<std::vector<uint8_t>(Context &)> task(
            [opt = options_, m = std::move(m)](Context &c) { return TransactionProver(opt, c).prove(m); });
        std::future<std::vector<uint8_t>> f = task.get_future();
        { std::lock_guard<std::mutex> g(m_); q_.push_back(std::move(task)); }
        cv_.notify_one();
        return f;
    }
    unsigned workers() const { return (unsigned)threads_.size(); }

  private:
    void run(unsigned w) {
        for (;;) {
            std::packaged_task<std::vector<uint8_t>(Context &)> task;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return; // stop_ and nothing left
                task = std::move(q_.front());
                q_.pop_front();
            }
            task(*ctx_[w]);
        }
    }
    ProofOptions options_;
    std::vector<std::unique_ptr<Context>> ctx_;
    std::vector<std::thread> threads_;
    std::deque<std::packaged_task<std::vector<uint8_t>(Context &)>> q_;
    std::mutex m_;
    std::condition_variable cv_;
    bool stop_ = false;
};

// src/lib.rs:92-150
class TransactionExample {
  public:
    TransactionExample(const ProofOptions &options, size_t num_transactions, Context &ctx, unsigned depth = 15, uint64_t seed = 0x5EED)
        : options_(options), tx_metadata_(TransactionMetadata::build_random(num_transactions, depth, seed)), ctx_(ctx) {}
    std::vector<uint8_t> prove() { return TransactionProver(options_, ctx_).prove(tx_metadata_); }
    PublicInputs pub_inputs() const { return TransactionProver::get_pub_inputs(tx_metadata_); }
    const TransactionMetadata &metadata() const { return tx_metadata_; }

  private:
    ProofOptions options_;
    TransactionMetadata tx_metadata_;
    Context &ctx_;
};
inline TransactionExample get_example(size_t num_transactions, Context &ctx) { // src/lib.rs:75-89
    return TransactionExample(ProofOptions(42, 8, 0, HashFunction::Blake3_256, FieldExtension::None, 4, 256), num_transactions, ctx);
}

namespace detail {
inline std::vector<uint8_t> air_prove(Context &ctx, int air, const ProofOptions &options, uint64_t number, size_t rows) {
    const cstark_options o = options.raw();
    std::vector<uint8_t> proof(2 * cstark_tx_proof_size_bound((uint32_t)((rows + 1023) / 1024), &o));
    size_t len = 0;
    check(cstark_air_prove(ctx.raw(), air, &o, number, proof.data(), proof.size(), &len));
    proof.resize(len);
    return proof;
}
} // namespace detail

// merkle::update::MerkleExample (src/merkle/update/mod.rs:36-127)
class MerkleExample {
  public:
    MerkleExample(const ProofOptions &options, TransactionMetadata m, Context &ctx) : options_(options), m_(std::move(m)), ctx_(ctx) { m_.validate(); }
    std::vector<uint8_t> prove() {
        const cstark_tx_witness w = m_.view();
        check(cstark_tx_witness_upload(ctx_.raw(), &w));
        return detail::air_prove(ctx_, CSTARK_AIR_MERKLE_UPDATE, options_, 0, m_.len() * 512);
    }
    PublicInputs pub_inputs() const { return {m_.initial_roots.at(0), m_.final_root}; }

  private:
    ProofOptions options_;
    TransactionMetadata m_;
    Context &ctx_;
};
// range::RangeProofExample (src/range/mod.rs:28-110)
class RangeProofExample {
  public:
    RangeProofExample(const ProofOptions &options, BaseElement number, Context &ctx) : options_(options), number_(number), ctx_(ctx) {}
    std::vector<uint8_t> prove() { return detail::air_prove(ctx_, CSTARK_AIR_RANGE, options_, number_, 64); }

  private:
    ProofOptions options_;
    BaseElement number_;
    Context &ctx_;
};
// The loop of benches/range.rs:15-37 in one call: one 64-row range proof per number (cstark_range_prove_batch), each byte-identical to
// RangeProofExample(options, number, ctx).prove()
inline std::vector<std::vector<uint8_t>> prove_range_batch(const ProofOptions &options, const std::vector<BaseElement> &numbers, Context &ctx) {
    const cstark_options o = options.raw();
    const size_t stride = cstark_tx_proof_size_bound(1, &o);
    std::vector<uint8_t> buf(stride * numbers.size());
    std::vector<size_t> lens(numbers.size());
    check(cstark_range_prove_batch(ctx.raw(), &o, numbers.data(), (uint32_t)numbers.size(), buf.data(), stride, lens.data()));
    std::vector<std::vector<uint8_t>> out(numbers.size());
    for (size_t t = 0; t < numbers.size(); t++) out[t].assign(buf.begin() + t * stride, buf.begin() + t * stride + lens[t]);
    return out;
}
// RescueExample of benches/rescue.rs:25-102: a chain of chain_length Rescue hashes from the bench's seed 42..48; the public inputs (seed
// and result) are read from the trace inside the prover, as RescueProver::get_pub_inputs does (:331-354)
class RescueExample {
  public:
    RescueExample(size_t chain_length, const ProofOptions &options, Context &ctx) : options_(options), chain_length_(chain_length), ctx_(ctx) {
        if (chain_length < 8 || (chain_length & (chain_length - 1))) throw Error(CSTARK_ERR_INVALID_ARG, "chain length must a power of 2"); // :34-37
        const unsigned __int128 p = ((unsigned __int128)1 << 62) + ((unsigned __int128)1 << 56) + ((unsigned __int128)1 << 55) + 1;
        for (int i = 0; i < 7; i++) seed[i] = (BaseElement)((((unsigned __int128)(42 + i)) << 64) % p); // BaseElement::from(42u8 + i): memory form, R = 2^64
    }
    std::vector<uint8_t> prove() {
        const cstark_options o = options_.raw();
        std::vector<uint8_t> proof(2 * cstark_tx_proof_size_bound((uint32_t)((chain_length_ + 127) / 128), &o));
        size_t len = 0;
        check(cstark_rescue_prove(ctx_.raw(), &o, seed, (uint32_t)chain_length_, proof.data(), proof.size(), &len));
        proof.resize(len);
        return proof;
    }
    BaseElement seed[7];

  private:
    ProofOptions options_;
    size_t chain_length_;
    Context &ctx_;
};
// schnorr::SchnorrExample (src/schnorr/mod.rs:52-186)
class SchnorrExample {
  public:
    SchnorrExample(const ProofOptions &options, size_t num_signatures, Context &ctx, uint64_t seed = 0x5EED)
        : options_(options), messages(num_signatures * 28), sig_rx(num_signatures * 6), sig_s(num_signatures * 32), ctx_(ctx) {
        check(cstark_schnorr_witness_generate((uint32_t)num_signatures, seed, messages.data(), sig_rx.data(), sig_s.data()));
    }
    std::vector<uint8_t> prove() {
        const uint32_t n = (uint32_t)(messages.size() / 28);
        check(cstark_schnorr_witness_upload(ctx_.raw(), n, messages.data(), sig_rx.data(), sig_s.data()));
        return detail::air_prove(ctx_, CSTARK_AIR_SCHNORR, options_, 0, (size_t)n * 512);
    }
    ProofOptions options_;
    std::vector<BaseElement> messages, sig_rx;
    std::vector<uint8_t> sig_s;

  private:
    Context &ctx_;
};

} // namespace cstark
#endif // CSTARK_HPP

// Exercises include/cstark.hpp the way the reference's tests exercise the crate (src/tests.rs:11-38): build an example of 2
// transfers, prove it, hand proof and public inputs to the caller (the Python test verifies them with the restated verifier).
// usage: host_mirror <out_prefix>      writes <prefix>.proof, <prefix>.pub (14 u64), <prefix>.range (proof of 17), prints a summary
#include <cstdio>
#include <fstream>
#include "cstark.hpp"

static void dump(const std::string &path, const void *p, size_t n) {
    std::ofstream f(path, std::ios::binary);
    f.write((const char *)p, (std::streamsize)n);
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::string prefix = argv[1];
    try {
        cstark::Context ctx;
        cstark::ProofOptions options(42, 8, 0, cstark::HashFunction::Blake3_256, cstark::FieldExtension::None, 4, 256); // build_options(1)
        cstark::TransactionExample transaction(options, 2, ctx, /*depth=*/3, /*seed=*/0x5EED);
        const std::vector<uint8_t> proof = transaction.prove();
        const cstark::PublicInputs pub = transaction.pub_inputs();
        dump(prefix + ".proof", proof.data(), proof.size());
        uint64_t pi[14];
        for (int i = 0; i < 7; i++) { pi[i] = pub.initial_root[i]; pi[7 + i] = pub.final_root[i]; }
        dump(prefix + ".pub", pi, sizeof pi);
        // error behaviour: a blowup factor below the AIR's constraint-evaluation blowup is refused, the metadata length check fires (src/lib.rs:211-218)
        bool refused = false, length_checked = false;
        try {
            cstark::TransactionExample bad(cstark::ProofOptions(42, 4, 0, cstark::HashFunction::Blake3_256, cstark::FieldExtension::None, 4, 256), 2, ctx, 3);
            bad.prove();
        } catch (const cstark::Error &) { refused = true; }
        try {
            cstark::TransactionMetadata m = transaction.metadata();
            m.deltas.pop_back();
            cstark::TransactionProver(options, ctx).prove(m);
        } catch (const cstark::Error &) { length_checked = true; }
        const auto ms = ctx.prove_stage_ms(); // of the last successful proof on this context
        // the range-proof loop of benches/range.rs in one call: every proof equals the single-proof path's bytes
        bool batch_ok = true;
        {
            cstark::Context rctx;
            const std::vector<cstark::BaseElement> numbers = {cstark_field_root_of_unity(1), cstark_field_generator(), cstark_field_root_of_unity(5)};
            const auto proofs = cstark::prove_range_batch(options, numbers, rctx);
            for (size_t t = 0; t < numbers.size(); t++) batch_ok = batch_ok && proofs[t] == cstark::RangeProofExample(options, numbers[t], rctx).prove();
            dump(prefix + ".range", proofs[1].data(), proofs[1].size());
        }
        if (!batch_ok) { std::fprintf(stderr, "batched range proofs differ from the single proofs\n"); return 4; }
        {   // the rescue bench's example at its own options (benches/rescue.rs:370-378: blowup 4), chain length 16
            cstark::RescueExample rescue(16, cstark::ProofOptions(42, 4, 0, cstark::HashFunction::Blake3_256, cstark::FieldExtension::None, 4, 256), ctx);
            const std::vector<uint8_t> rp = rescue.prove();
            dump(prefix + ".rescue", rp.data(), rp.size());
        }
        // several proofs in flight on one GPU: every proof equals the one proved alone; a bad witness fails its own future only
        bool pool_ok = true;
        {
            cstark::ProverPool pool(options, 2);
            std::vector<cstark::TransactionMetadata> metas;
            std::vector<std::future<std::vector<uint8_t>>> futures;
            for (int t = 0; t < 5; t++) metas.push_back(cstark::TransactionMetadata::build_random(4, 3, 100 + t));
            for (const auto &m : metas) futures.push_back(pool.submit(m));
            cstark::TransactionMetadata broken = metas[0];
            broken.deltas.pop_back();
            std::future<std::vector<uint8_t>> bad = pool.submit(broken);
            for (size_t t = 0; t < metas.size(); t++) pool_ok = pool_ok && futures[t].get() == cstark::TransactionProver(options, ctx).prove(metas[t]);
            bool threw = false;
            try { bad.get(); } catch (const cstark::Error &) { threw = true; }
            pool_ok = pool_ok && threw;
        }
        if (!pool_ok) { std::fprintf(stderr, "proofs of the pool differ from proofs proved alone\n"); return 5; }
        std::printf("proof_bytes=%zu refused=%d length_checked=%d trace_ms=%.3f\n", proof.size(), (int)refused, (int)length_checked, ms[0]);
        return refused && length_checked ? 0 : 1;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 3;
    }
}

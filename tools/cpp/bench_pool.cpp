// Throughput of cstark::ProverPool: `proofs` proofs of one 1024-transfer witness (2^20 steps, 96 queries) through 1, 2 and 3 workers on
// one GPU.  Build and run: tools/gpu_jobs/bench_pool.sh
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "cstark.hpp"

int main(int argc, char **argv) {
    const int proofs = argc > 1 ? atoi(argv[1]) : 24;
    try {
        cstark::ProofOptions options(96, 8, 0, cstark::HashFunction::Blake3_256, cstark::FieldExtension::None, 4, 256);
        const cstark::TransactionMetadata meta = cstark::TransactionMetadata::build_random(1024, 15, 7);
        for (unsigned workers = 1; workers <= 3; workers++) {
            cstark::ProverPool pool(options, workers);
            for (unsigned w = 0; w < 2 * workers; w++) pool.submit(meta).get(); // warm every context (first proof allocates)
            std::vector<std::future<std::vector<uint8_t>>> f;
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < proofs; i++) f.push_back(pool.submit(meta));
            size_t bytes = 0;
            for (auto &x : f) bytes = x.get().size();
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("ProverPool(%u): %d proofs of %zu bytes in %.3f s = %.2f proofs/s (%.2f ms per proof)\n", workers, proofs, bytes, s, proofs / s,
                        1e3 * s / proofs);
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}

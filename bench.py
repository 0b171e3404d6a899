#!/usr/bin/env python3
"""Benchmark of the MI355X hot path for the Topos state-transition AIR (BASELINE.json north_star).

One "step" = one complete proof of one synthetic batch: the witness of 1024 transfers (2^20 trace rows, 94 registers,
Merkle depth 15) is already resident in HBM; the timed region is one cstark_tx_prove call =
    K1 trace generation -> K2 interpolation -> K3 coset LDE (blowup 8) -> K4 Blake3 row hashing -> K5 Merkle tree
    -> K6 constraint evaluation (115 constraints, fused random linear combination + boundary terms)
    -> composition polynomial + commitment -> out-of-domain frame -> DEEP composition -> FRI layers -> 96 query openings
    -> proof bytes on the host,
i.e. exactly the region the reference times (TransactionExample::prove = build_trace + Prover::prove, src/lib.rs:116-141,
benches/state_transition.rs:21-24).  --mode hotpath times only K1..K6 (the SURVEY 8(a) rows) stage by stage.

Launch:  python bench.py --gpus N --steps K --warmup W
  N > 1 without a torch.distributed environment (RANK / WORLD_SIZE unset): this process starts the N ranks itself as a child
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>` BEFORE it
  imports torch or touches the GPU, relays rank 0's JSON line and exits with the child's code.  Launched by torch.distributed.run
  (the driver's form) it is one rank; --gpus must then equal WORLD_SIZE.
Multi-GPU, default (--mode prove): independent proofs per GPU (replicas, no data-path collective; the only RCCL traffic is the
  barrier / max-over-ranks timing): weak scaling.  --mode shard: ONE proof across the GPUs by LDE coset with RCCL all-gathers of
  leaf digests, merged evaluations and query rows: strong scaling.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

P = 2**62 + 2**56 + 2**55 + 1
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a streaming copy reaches
LOG_N, WIDTH, LOG_B, DEPTH = 20, 94, 3, 15
FIXTURE = os.path.join(ROOT, "tests", "golden", "witness_1024_d15.npz")


def algorithmic_bytes(n, w, b):
    """every stage reads its input once and writes its output once."""
    return {
        "trace_gen": n * w * 8,
        "interpolate": 2 * n * w * 8,
        "lde": n * w * 8 + b * n * w * 8,
        "hash_rows": b * n * w * 8 + b * n * 32,
        "merkle": 2 * b * n * 32,
        "constraints": b * n * w * 8 + b * n * 8,
    }


def cpu_baseline(meta, sample_tx, full_options=None):
    """The oracle (CPU port, OpenMP) on a bounded sample of the same workload, all host cores."""
    from oracle import oracle as O
    if "OMP_NUM_THREADS" not in os.environ:  # the one big proof: many cores, but not every hardware thread of a shared 256-thread host
        O.set_num_threads(min(64, len(os.sched_getaffinity(0))))  # (measured: 128 threads 22 s, 256 threads 49 s)
    w = O.TxWitness(sample_tx, meta.depth)
    for f in w.FIELDS:
        src = getattr(meta, f)
        getattr(w, f)[...] = src if f == "final_root" else src[:sample_tx]
    w.final_root[...] = meta.initial_roots[sample_tx] if sample_tx < meta.n_tx else meta.final_root
    if full_options is not None:
        from oracle import prover as OP
        t0 = time.perf_counter()
        proof = (OP.prove_ext if full_options[4] else OP.prove)(w, full_options)
        total = time.perf_counter() - t0
        return {
            "value": round(sample_tx / meta.n_tx / total, 5), "unit": "proofs/s", "cores": O.num_threads(), "kind": "port",
            "sample": "complete proof of %d of %d transactions (2^%d of 2^20 rows, %d proof bytes) by the OpenMP oracle prover in %.2f s%s" % (
                sample_tx, meta.n_tx, (sample_tx * 1024).bit_length() - 1, len(proof), total,
                "" if sample_tx == meta.n_tx else ", linearly extrapolated"),
        }
    cf = O.make_coeffs(17)
    pub = np.concatenate([w.initial_roots[0][:2], w.final_root[:2]])
    t0 = time.perf_counter()
    trace = O.tx_build_trace(w)
    t1 = time.perf_counter()
    co = O.interpolate_columns(trace)
    lde = O.lde_columns(co, LOG_B)
    t2 = time.perf_counter()
    leaves = O.hash_rows(lde, LOG_B)
    O.merkle_build(leaves)
    t3 = time.perf_counter()
    O.tx_evaluate_constraints(lde, cf, pub, w.depth, LOG_B)
    t4 = time.perf_counter()
    total = t4 - t0
    frac = sample_tx / meta.n_tx
    return {
        "value": round(frac / total, 5), "unit": "proofs/s", "cores": O.num_threads(), "kind": "port",
        "sample": "%d of %d transactions (2^%d of 2^20 rows) through the same stages, linearly extrapolated; "
                  "OpenMP oracle: trace %.2fs, lde %.2fs, commit %.2fs, constraints %.2fs" % (
                      sample_tx, meta.n_tx, (sample_tx * 1024).bit_length() - 1, t1 - t0, t2 - t1, t3 - t2, t4 - t3),
    }


def b_core(n, w, b=8):
    """algorithmic bytes of the hot path (trace .. constraint evaluation) for an n x w trace at blowup b."""
    return 8 * n * w * (4 + 3 * b) + 104 * b * n


def _config_traffic(name):
    """HBM bytes per proof of a sub-AIR configuration: every kernel of the newest profiles/r*_<name>_hbm_traffic_pmc.csv (tools/profile.sh
    over tools/bench_air_one.py; 2 x FETCH_SIZE + WRITE_SIZE per MI355X_MICROARCH.md), or None"""
    import csv
    import glob
    import re
    key = lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))]
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_hbm_traffic_pmc.csv" % name)), key=key, reverse=True)
    if not files:
        return None, None
    rows = list(csv.DictReader(open(files[0])))
    return int(sum(float(r["fetch_x2_plus_write_GB_per_proof"]) for r in rows) * 1e9), "profiles/" + os.path.basename(files[0])


def other_configs(backend, queries=42):
    """BASELINE.json configs 1-3 (range / merkle / schnorr) through the product path, after the timed headline region: complete proofs
    (cstark_air_prove / cstark_range_prove_bits), milliseconds per proof from the host clock around `reps` back-to-back proofs, proof
    bytes, the stage split of the last proof (HIP events inside the library), and a roofline entry per configuration: B_core of
    SURVEY.md 8(d) over the hot-path stages (trace .. constraints).  Witnesses come from the product's own seeded generators with the
    seeds of tools/proof_configs.py, so every proof timed here is one the GPU suite pins to the CPU prover byte for byte
    (tests/test_gpu_pinned_proofs.py; range 1024 x 64: tests/test_gpu_baseline_configs.py)."""
    from certificate_stark_amd.prover import MerkleExample, ProofOptions, RangeProofExample, SchnorrExample, TransactionMetadata
    opt = ProofOptions(queries, 8, 0, 0, 0, 4, 256)
    out = []

    def timed(name, workload, prove, rows, width, reps, note=None, stages=True, blowup=8):
        proof = prove()
        prove()
        t0 = time.perf_counter()
        for _ in range(reps):
            proof = prove()
        ms = (time.perf_counter() - t0) / reps * 1e3
        if not stages:      # the batch prover (a 64-row proof is a batch of one) has no per-stage events
            out.append({"config": name, "workload": workload, "ms_per_proof": round(ms, 3), "proofs_per_s": round(1e3 / ms, 2), "proof_bytes": len(proof),
                        **({"note": note} if note else {})})
            return
        st = backend.prove_stage_ms()
        hot = sum(st[k] for k in ("trace", "interpolate", "lde", "commit", "constraints"))
        alg = b_core(rows, width, blowup)
        traffic, tsrc = _config_traffic({"merkle_2_18_d15": "merkle_2_18", "schnorr_2_18": "schnorr_2_18", "range_2_16": "range_2_16"}.get(name, name))
        e = {"config": name, "workload": workload, "ms_per_proof": round(ms, 3), "proofs_per_s": round(1e3 / ms, 2), "proof_bytes": len(proof),
             "stage_ms": {k: round(v, 3) for k, v in st.items()},
             "roofline": {"bound": "hbm", "achieved": round(alg / (hot * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(alg / (hot * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": alg, "hot_path_ms": round(hot, 3), "traffic": traffic,
                          "traffic_source": tsrc, "traffic_scope": None if traffic is None else "every kernel of a whole proof (hot path and the stages after it)"}}
        if note:
            e["note"] = note
        out.append(e)

    rng_words = np.random.default_rng(16).integers(0, 2**64, size=(1 << 16) // 64, dtype=np.uint64)
    rng_words[-1] &= np.uint64(2**63 - 1)
    timed("range_2_16", "benches/range.rs range-proof AIR, 2^16 steps, blowup 8 (SYNTHETIC long accumulator: the reference's range trace is fixed at 64 rows)",
          lambda: backend.range_prove_bits(opt, rng_words, 16), 1 << 16, 2, 10)
    one = RangeProofExample(opt, 12345 << 3, backend)
    timed("range_64", "benches/range.rs range-proof AIR, the reference's own shape: 64 rows x 2 registers", one.prove, 64, 2, 20,
          note="one call per proof: host-API bound (about 150 launches and round trips); the batched call below is the throughput path")
    numbers = [(12345 + i) << 3 for i in range(1024)]
    if hasattr(backend, "range_prove_batch"):
        backend.range_prove_batch(opt, numbers[:64])
        t0 = time.perf_counter()
        proofs = backend.range_prove_batch(opt, numbers)
        ms = (time.perf_counter() - t0) * 1e3
        out.append({"config": "range_1024x64", "workload": "1024 independent 64-row range proofs (2^16 rows in total) in ONE batched call (cstark_range_prove_batch)",
                    "ms_per_proof": round(ms / 1024, 4), "ms_per_batch": round(ms, 3), "proofs_per_s": round(1024e3 / ms, 1),
                    "proof_bytes": len(proofs[0])})
    else:
        t0 = time.perf_counter()
        for v in numbers:
            proof = RangeProofExample(opt, v, backend).prove()
        ms = (time.perf_counter() - t0) * 1e3
        out.append({"config": "range_1024x64", "workload": "1024 independent 64-row range proofs (2^16 rows in total), one cstark_air_prove call each",
                    "ms_per_proof": round(ms / 1024, 4), "ms_per_batch": round(ms, 3), "proofs_per_s": round(1024e3 / ms, 1), "proof_bytes": len(proof),
                    "note": "host-API bound: about 150 launches and round trips per 64-row proof"})
    full = TransactionMetadata.load(FIXTURE)
    m512 = TransactionMetadata(*[getattr(full, f) if f == "final_root" else getattr(full, f)[:512] for f in TransactionMetadata.FIELDS])
    m512.final_root = full.initial_roots[512].copy()
    # (MerkleExample / SchnorrExample keep their witness in device memory after the first call: the timed proofs exclude the upload, as
    # the headline does -- "witness resident" in the workload strings)
    timed("merkle_2_18_d15", "benches/merkle.rs Merkle AIR, 512 transfers = 2^18 steps, depth 15 (the reference's constant); witness resident, upload excluded",
          MerkleExample(opt, m512, backend).prove, 1 << 18, 65, 5)
    m31 = TransactionMetadata.build_random(512, 31, seed=31)
    timed("merkle_2_18_d31", "benches/merkle.rs Merkle AIR, 512 transfers = 2^18 steps, depth 31 (nearest legal depth to BASELINE's 32: depth + 1 must be a power of two); witness resident, upload excluded",
          MerkleExample(opt, m31, backend).prove, 1 << 18, 65, 5)
    sch = SchnorrExample.build_random(opt, 512, seed=1, backend=backend)
    timed("schnorr_2_18", "benches/schnorr.rs Schnorr AIR, 512 signatures = 2^18 steps; witness resident, upload excluded", sch.prove, 1 << 18, 56, 5)
    if hasattr(backend, "rescue_prove"):  # BASELINE config 0: the reference times it on the CPU; here the same AIR through cstark_rescue_prove
        from certificate_stark_amd.prover import RescueExample
        rex = RescueExample(512, ProofOptions(queries, 4, 0, 0, 0, 4, 256), backend)
        timed("rescue_2_12", "benches/rescue.rs Rescue-Prime hash chain, 512 links = 2^12 trace steps, blowup 4 (the bench's options, :370-378); one "
              "sequential chain: the trace is a single wave's recurrence", rex.prove, 1 << 12, 14, 10, blowup=4)
    return out


def two_in_flight(meta, options, device, proofs_each=4):
    """Throughput with TWO independent proofs in flight on the GPU (two contexts, streams and host threads; witnesses resident): the host
    round trips of one proof's Fiat-Shamir channel (about 0.9 ms of GPU idle time per proof) are filled by the other.  Reported beside
    the headline, which stays one proof at a time -- the shape of the reference's benchmark."""
    import threading
    import torch
    from certificate_stark_amd.backend import Backend
    from certificate_stark_amd.prover import TransactionProver
    provers = []
    for _ in range(2):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            pr = TransactionProver(options, Backend(device))
            pr.load_witness(meta)
            pr.prove()
        provers.append((st, pr))
    torch.cuda.synchronize()

    def run(st, pr):
        with torch.cuda.stream(st):
            for _ in range(proofs_each):
                pr.prove()

    t0 = time.perf_counter()
    threads = [threading.Thread(target=run, args=o) for o in provers]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for _, pr in provers:
        pr.backend.close()
    return {"value": round(2 * proofs_each / dt, 3), "unit": "proofs/s", "ms_per_proof": round(dt / (2 * proofs_each) * 1e3, 3), "proofs": 2 * proofs_each,
            "note": "two independent proofs in flight on one GPU (two contexts / streams / host threads)"}


class PmcTraffic:
    """HBM bytes per proof and kernel from the newest PMC summary under profiles/ (tools/profile.sh + tools/pmc_summary.py run on the
    GPU box for this build; separate FETCH_SIZE / WRITE_SIZE passes, 2 x FETCH + WRITE as MI355X_MICROARCH.md prescribes for gfx950).
    No file, or a kernel that is not in it: None -- never a literal."""

    def __init__(self):
        import csv
        import glob
        import re
        self.gb, self.source = {}, None
        files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic_pmc.csv"))
                 if re.match(r"r\d+_[a-z0-9]+_hbm_traffic_pmc.csv", os.path.basename(f))]  # the headline's set (<round>_<tag>_...), not a sub-AIR configuration's
        key = lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))]
        for f in sorted(files, key=key, reverse=True):
            rows = list(csv.DictReader(open(f)))
            if rows and "fetch_x2_plus_write_GB_per_proof" in rows[0]:
                self.gb = {r["kernel"]: float(r["fetch_x2_plus_write_GB_per_proof"]) for r in rows}
                self.source = "profiles/" + os.path.basename(f)
                break

    def per_proof(self, names):
        """names: exact kernel names; a name ending in '<' matches every instantiation of that template"""
        if not names or not self.gb:
            return None
        hit = [v for k, v in self.gb.items() if any(k == nm or (nm.endswith("<") and k.startswith(nm)) for nm in names)]
        return sum(hit) if hit else None


class PmcValu:
    """The vector-instruction counters of the newest profile set under profiles/ (tools/profile.sh: <tag>_valu_pmc.csv, one row per
    kernel: SQ_INSTS_VALU, SQ_BUSY_CYCLES, SQ_WAVES, GRBM_GUI_ACTIVE per dispatch) together with the static issue cost per instruction
    of the same kernels (tools/isa_mix.py --csv -> <tag>_isa_mix.csv: 4 SIMD cycles for v_mad_u64_u32 / 64-bit adds / carries / VOP3,
    2 for the other 32-bit vector instructions, profiles/r02_valu_issue_bench*.txt).  Nothing here is a literal: no file, no entry."""
    N_SIMD, N_XCD = 1024, 8  # 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs

    def __init__(self, suffix="_valu_pmc.csv"):
        import csv
        import glob
        import re
        self.rows, self.cpi, self.source, self.cpi_source = {}, {}, None, None
        key = lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))]
        files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*" + suffix)) if (suffix != "_valu_pmc.csv" or re.match(r"r\d+_[a-z0-9]+_valu_pmc.csv", os.path.basename(f)))]
        for f in sorted(files, key=key, reverse=True)[:1]:
            self.rows = {r["kernel"]: r for r in csv.DictReader(open(f))}
            self.source = "profiles/" + os.path.basename(f)
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_isa_mix.csv")), key=key, reverse=True)[:1]:
            self.cpi = {r["kernel"]: float(r["cycles_per_valu_instruction"]) for r in csv.DictReader(open(f))}
            self.cpi_source = "profiles/" + os.path.basename(f)

    def entry(self, kernel, points):
        """kernel: exact name in the PMC summary; points: evaluation points (or elements) one dispatch processes"""
        r = self.rows.get(kernel)
        if r is None:
            return None
        insts = float(r["SQ_INSTS_VALU_per_dispatch"])
        gui = float(r["GRBM_GUI_ACTIVE_per_dispatch"]) / self.N_XCD
        ns = float(r["avg_ns_under_pmc"])
        e = {"kernel": kernel, "wave_instructions": int(insts), "insts_per_point": round(insts * 64 / points, 1) if points else None,
             "gui_cycles": int(gui), "clock_ghz": round(gui / ns, 3) if ns else None, "kernel_us_under_pmc": round(ns / 1e3, 1), "source": self.source,
             # bracket: every vector instruction at 4 SIMD cycles (v_mad_u64_u32, 64-bit adds, carries, VOP3) -- the ceiling of the estimate below
             "frac_of_issue_at_4_cycles": round(insts * 4 / self.N_SIMD / gui, 3) if gui else None}
        cpi = self.cpi.get(kernel)
        if cpi is not None:
            issue = insts * cpi / self.N_SIMD
            e.update({"cycles_per_instruction_static": cpi, "issue_cycles_est": int(issue), "frac_of_issue": round(issue / gui, 3) if gui else None,
                      "cpi_source": self.cpi_source})
        return e


def spawn_ranks(n):
    """--gpus N without a torch.distributed environment: start the N ranks as a fresh child process tree (this process has not
    imported torch and never touches the GPU), pass the child's output through, and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:  # rank 0 prints the one JSON line; anything else the ranks write to stdout is passed through to stderr
        if out.startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks exited without a result line\n")
        rc = 1
    if line is not None and rc == 0:
        print(line, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-tx", type=int, default=1024, help="transactions per proof (1024 = BASELINE's 2^20 rows)")
    ap.add_argument("--cpu-sample-tx", type=int, default=1024,
                    help="transactions the CPU baseline proves (default: the whole 1024-transaction witness, no extrapolation; about 25 s on the GPU "
                         "box's host cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip BASELINE.json's configs 1-3 (range / merkle / schnorr proofs, about 2 s after the timed region; N = 1, --mode prove only)")
    ap.add_argument("--with-composition", action="store_true",
                    help="also time the next stage of prove(): composition polynomial columns + their LDE + Blake3 commitment "
                         "(reported as extra_stage_ms, not part of the hot-path metric)")
    ap.add_argument("--mode", choices=["prove", "hotpath", "shard"], default="prove",
                    help="prove: complete proofs, independent per GPU (weak scaling, no collective); hotpath: K1..K6 only, "
                         "independent per GPU; shard: ONE complete proof sharded by LDE coset over the GPUs with RCCL all-gathers of "
                         "digests and merged evaluations (strong scaling; needs --gpus 2, 4 or 8)")
    ap.add_argument("--hash-fn", choices=["blake3", "sha3"], default="blake3", help="ProofOptions hash (the headline metric uses Blake3_256, src/lib.rs:82)")
    ap.add_argument("--field-extension", choices=["none", "quadratic", "cubic"], default="none",
                    help="ProofOptions field extension (the headline metric uses None; the reference's CLI defaults to cubic)")
    ap.add_argument("--inflight", type=int, default=1,
                    help="prove mode: independent proofs in flight per GPU, one context + stream + host thread each (default 1: one proof "
                         "at a time, the reference benchmark's shape; 2 was measured +7%% proofs/s, see DESIGN.md). --steps must be a multiple")
    ap.add_argument("--queries", type=int, default=96, help="FRI queries (BASELINE.json: 96; the reference's get_example: 42)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d; pass --gpus equal to the number of ranks torch.distributed.run started "
                         "(or run `python bench.py --gpus N` alone: it starts the ranks itself)" % (args.gpus, world))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    # CSTARK_BENCH_REHEARSE=1 (never the driver's form): every rank on GPU 0 and the control collectives over gloo, to exercise the spawner,
    # the rendezvous, the barriers and the max-over-ranks reduction on a one-GPU box; the line then carries "rehearsal" and no rccl_ranks
    rehearse = world > 1 and os.environ.get("CSTARK_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    rccl_ranks = None
    if rehearse:
        dist.init_process_group("gloo")
    elif world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        ones = torch.ones(1, device="cuda", dtype=torch.int64)
        dist.all_reduce(ones)  # one real RCCL collective over every rank's device buffer: the sum is the number of ranks that took part
        rccl_ranks = int(ones.item())
        if rccl_ranks != world or dist.get_world_size() != world:
            raise SystemExit("bench.py: RCCL all-reduce saw %d ranks, expected %d" % (rccl_ranks, world))

    from certificate_stark_amd import _lib
    from certificate_stark_amd.backend import Backend
    from certificate_stark_amd.prover import ProofOptions, TransactionMetadata, TransactionProver

    meta_full = TransactionMetadata.load(FIXTURE)
    n_tx = args.n_tx
    meta = meta_full if n_tx == meta_full.n_tx else TransactionMetadata(
        *[getattr(meta_full, f) if f == "final_root" else getattr(meta_full, f)[:n_tx] for f in TransactionMetadata.FIELDS])
    if n_tx != meta_full.n_tx:
        meta.final_root = meta_full.initial_roots[n_tx].copy()
    n = n_tx * 1024
    log_n = n.bit_length() - 1

    hash_fn = 1 if args.hash_fn == "sha3" else 0
    field_ext = {"none": 0, "quadratic": 1, "cubic": 2}[args.field_extension]
    prover = TransactionProver(ProofOptions(num_queries=args.queries, hash_fn=hash_fn, field_extension=field_ext), Backend(local))
    prover.load_witness(meta)  # witness resident in HBM before the timed region
    rng = np.random.default_rng(1234 + rank)  # one proof = one set of coefficients
    cf = _lib.TxCoeffsStruct()
    for name, k in (("t_alpha", 115), ("t_beta", 115), ("b_alpha", 4), ("b_beta", 4)):
        v = rng.integers(1, P, size=k, dtype=np.uint64)
        for i in range(k):
            getattr(cf, name)[i] = int(v[i])
    pub = [int(meta.initial_roots[0][0]), int(meta.initial_roots[0][1]), int(meta.final_root[0]), int(meta.final_root[1])]

    stages = ["trace_gen", "interpolate", "lde", "hash_rows", "merkle", "constraints"]
    acc_ms = {s: 0.0 for s in stages}
    coset_mode = args.mode == "shard"
    if coset_mode and world not in (2, 4, 8):
        raise SystemExit("bench.py --mode shard splits one proof over 2, 4 or 8 GPUs (--gpus)")
    if coset_mode:
        from certificate_stark_amd import sharding

    def step_coset(timed):
        """one complete proof over all ranks (cstark_tx_shard_* phases, RCCL all-gathers in between); bytes on rank 0"""
        proof = sharding.prove_sharded(prover.backend, prover.options)
        if proof is not None:
            proof_len[0] = len(proof)
            if timed:
                for k, v in prover.backend.prove_stage_ms().items():
                    pacc[k] += v
        return None

    prove_mode = args.mode in ("prove", "shard")
    pstages = list(Backend.PROVE_STAGES)
    pacc = {k: 0.0 for k in pstages}
    proof_len = [0]

    def step_prove(timed):
        proof = prover.prove()
        proof_len[0] = len(proof)
        if timed:
            for k, v in prover.backend.prove_stage_ms().items():
                pacc[k] += v
        return None

    def step(timed):
        if coset_mode:
            return step_coset(timed)
        if prove_mode:
            return step_prove(timed)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(stages) + 1)] if timed else None
        b = prover.backend
        if timed: ev[0].record()
        trace = prover.build_trace()
        if timed: ev[1].record()
        coeffs = b.interpolate_columns(trace, out=prover._buf("coeffs", (WIDTH, n)))
        if timed: ev[2].record()
        lde = b.lde_columns(coeffs, LOG_B, out=prover._buf("lde", (1 << LOG_B, WIDTH, n)))
        if timed: ev[3].record()
        L = n << LOG_B
        nodes = prover._buf("nodes", (2 * L, 32), torch.uint8)
        b.hash_rows(lde, LOG_B, leaves=nodes[L:])
        if timed: ev[4].record()
        b.merkle_build(nodes)
        if timed: ev[5].record()
        prover.evaluate_constraints(lde, cf, pub, input_is_lde=True)  # the table is this run's own extension: degree-split evaluation
        if timed: ev[6].record()
        return ev

    def collect_parts():
        if coset_mode:  # the ranks' shares of the split evaluation are not timed part by part
            prover.backend.synchronize()
            return
        for k, v in prover.backend.constraint_part_ms().items():
            part_ms[k] += v

    prover.backend.set_part_timing(True)
    part_ms = {k: 0.0 for k in prover.backend.CE_PARTS}
    for _ in range(args.warmup):
        step(False)
    prover.backend.lde_timing_ms()  # reset: only the timed steps count
    inflight = args.inflight if prove_mode else 1
    if inflight > 1 and args.steps % inflight:
        raise SystemExit("--steps must be a multiple of --inflight")
    others = []
    for _ in range(inflight - 1):  # further provers: own context, own stream, witness resident, warmed up
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            pr = TransactionProver(ProofOptions(num_queries=args.queries, hash_fn=hash_fn, field_extension=field_ext), Backend(local))
            pr.load_witness(meta)
            for _ in range(max(args.warmup, 1)):
                pr.prove()
        others.append((st, pr))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    events = []

    def run_other(st, pr):
        with torch.cuda.stream(st):
            for _ in range(args.steps // inflight):
                pr.prove()

    import threading
    threads = [threading.Thread(target=run_other, args=o) for o in others]
    for t in threads:
        t.start()
    for _ in range(args.steps // inflight):
        events.append(step(True))
        collect_parts()  # waits for this step's last constraint launch: the steps are serialised on one stream anyway
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if rehearse else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if prove_mode:
        stage_ms = {k: pacc[k] / (args.steps // inflight) for k in pstages}
    else:
        for ev in events:
            for i, s in enumerate(stages):
                acc_ms[s] += ev[i].elapsed_time(ev[i + 1])
        stage_ms = {s: acc_ms[s] / args.steps for s in stages}

    extra = None
    if args.with_composition and args.mode == "hotpath":
        b = prover.backend
        comb = prover._bufs["combined"]
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for rep in range(2):
            e[0].record()
            cols = b.composition_columns(comb, out=prover._buf("comp_cols", (8, n)))
            e[1].record()
            clde = b.lde_columns(cols, LOG_B, out=prover._buf("comp_lde", (8, 8, n)))
            e[2].record()
            L = n << LOG_B
            cnodes = prover._buf("comp_nodes", (2 * L, 32), torch.uint8)
            b.hash_rows(clde, LOG_B, leaves=cnodes[L:])
            b.merkle_build(cnodes)
            e[3].record()
            torch.cuda.synchronize()
        extra = {"composition_columns": round(e[0].elapsed_time(e[1]), 3), "composition_lde": round(e[1].elapsed_time(e[2]), 3),
                 "composition_commit": round(e[2].elapsed_time(e[3]), 3)}

    lde_ms, lde_elements = prover.backend.lde_timing_ms()  # every low-degree extension of the timed steps (HIP events on the context's stream)
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        per = args.steps // inflight
        ab = algorithmic_bytes(n, WIDTH, 1 << LOG_B)
        part_avg = {k: v / per for k, v in part_ms.items()}
        split = (args.mode == "prove" and field_ext == 0) or args.mode == "hotpath"
        traffic = PmcTraffic()
        # ---- roofline entries: ALGORITHMIC bytes (input read once, output written once) / HIP-event duration measured
        # in the timed region above; `traffic` = HBM bytes per launch set from the PMC passes of tools/profile.sh on this build
        # (profiles/<tag>_hbm_traffic_pmc.csv: 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md) ----
        def entry(kernel, alg_bytes, ms, pmc_names, note=None, launches=1):
            ach = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else None
            t = traffic.per_proof(pmc_names) if n_tx == 1024 and args.mode == "prove" and field_ext == 0 and hash_fn == 0 else None
            e = {"bound": "hbm", "kernel": kernel, "achieved": None if ach is None else round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": None if ach is None else round(ach / HBM_PEAK_GBS, 4), "traffic": None if t is None else round(t * 1e9 / launches),
                 "algorithmic_bytes": int(alg_bytes / launches), "kernel_ms": round(ms / launches, 4), "launches_per_proof": launches,
                 "traffic_source": traffic.source if t is not None else None}
            if note:
                e["note"] = note
            return e
        # the transform pair of the low-degree extensions: per evaluation written 8 bytes out + 1/8 coefficient read (blowup 8)
        lde_alg = lde_elements / per * 8 * (1 + 1.0 / (1 << LOG_B))
        lde_pairs = max(1, round(lde_elements / per / (WIDTH * n)))  # in units of "94 columns of one coset"
        ntt_names = ["k_ntt_cols_v5<4, 3, 3, false>", "k_ntt_rows_v5<4, 3, 3, false>", "k_ntt_cols_v4<4, 3, 3, false>", "k_ntt_rows_v4<4, 3, 3, false>",
                     "k_ntt_cols_v2<5, 5, false>", "k_ntt_rows_v2<5, 5, false>"]
        roofline_lde = entry("forward transform pair of the low-degree extension (k_ntt_cols_v5 + k_ntt_rows_v5; all LDE calls of a proof: trace 94 "
                             "columns x 8 cosets, composition 8 x 8, DEEP 1 x 8; traffic also covers the 58 forward transforms of the split polynomials, same kernels)",
                             lde_alg, lde_ms / per, ntt_names,
                             "the dominant kernel family by GPU time: about 300 vector instructions per element and coset at 2-4 issue cycles each "
                             "(profiles/*_valu_issue_bench*.txt); bound by vector-instruction issue, not by HBM: the two kernels spend 87 / 96 % of their "
                             "cycles issuing at the clock they reach beside their HBM traffic (1.8-1.9 GHz; 2.1-2.2 GHz with the traffic compiled out, "
                             "2.4 GHz for pure arithmetic: profiles/r03_ntt_clock_and_phase_skip.txt, DESIGN.md 5); priced against the HBM roofline as "
                             "BASELINE.json asks; kernel_ms = per 94 columns x 1 coset; inside cstark_tx_prove the first two column batches are extended "
                             "beside the trace recurrences",
                             launches=lde_pairs)
        nb = (8 // world) if coset_mode else 8
        rounds_bytes = (nb // 2) * n * (58 * 8 + 6 * 8) if split else nb * n * (58 * 8 + 8)
        rounds_mfma = split and os.environ.get("CSTARK_ROUNDS_MFMA", "1") != "0"  # csrc/rounds_mfma.hip
        roofline_rounds = entry(("k_rounds_mfma (Rescue windows of the constraint evaluation, even cosets; INV_MDS and the sections' sums as int8 GEMMs "
                                 "of byte diagonals on the matrix cores)" if rounds_mfma else "k_rounds_split (Rescue windows of the constraint evaluation, even cosets)")
                                if split else "k_eval_fused<0> (Rescue windows of the constraint evaluation)", rounds_bytes, part_avg["rounds"],
                                (["k_rounds_mfma<", "k_rounds_mfma_tables"] if rounds_mfma else ["k_rounds_split<1>"]) if split else ["k_eval_fused<0, 1>"],
                                "round 4: 3.08 ms on the vector ALU alone (22.9 k instructions per point) -> the constant-matrix products on the matrix "
                                "cores, cubes / recombination / reductions left on the vector ALU; still bound by vector-instruction issue and by what two "
                                "waves per SIMD can overlap (profiles/*_valu_pmc.csv; DESIGN.md 5.7)")
        if rounds_mfma and part_avg["rounds"] > 0:
            # matrix-core side of the same launch: 462 v_mfma_i32_32x32x32_i8 per wave of 64 points (5 windows x 7 tiles x 4 k-steps x 2
            # point halves + 13 section tiles x 7 x 2), 32 * 32 * 32 * 2 integer operations each; peak = dense int8 (2 x the bf16 rate,
            # MI355X_MICROARCH.md).  kernel time = the 'rounds' part minus nothing: the two setup launches are 20 microseconds
            n_mfma = 5 * 56 + (field_ext + 1) * 182  # per wave of 64 points: the inverse matrix once, the section tiles per coefficient set
            ops = ((nb // 2) * n / 64) * n_mfma * 65536
            tops = ops / (part_avg["rounds"] * 1e-3) / 1e12
            roofline_rounds["mfma"] = {"bound": "mfma", "achieved": round(tops, 1), "peak": 5000.0, "unit": "TOP/s (int8, dense)", "frac": round(tops / 5000.0, 4),
                                       "ops_per_launch": int(ops), "mfma_per_wave": n_mfma,
                                       "note": "the matrix pipe is busy for about a quarter of the kernel; what bounds the launch is the vector work per output "
                                               "(one recombination and Montgomery reduction per byte-diagonal set, the cubes) at two waves per SIMD"}
        valu = PmcValu()
        pts_headline = n_tx == 1024 and args.mode == "prove" and field_ext == 0 and hash_fn == 0  # the workload the profile set was taken on
        if pts_headline:
            # the transform pair: one dispatch = all 8 cosets of a table; the trace table (94 columns) dominates the per-dispatch average
            # of the counter file, whose rows average over EVERY dispatch of a kernel name in a proof: points = elements per average dispatch
            # are not recoverable from it, so the per-element figure is taken from the instruction and element totals of a proof
            cols, rows_ = valu.entry("k_ntt_cols_v5<4, 3, 3, false>", 0), valu.entry("k_ntt_rows_v5<4, 3, 3, false>", 0)
            if cols and rows_:
                disp = {k: float(valu.rows[k]["dispatches"]) / 4 for k in ("k_ntt_cols_v5<4, 3, 3, false>", "k_ntt_rows_v5<4, 3, 3, false>")}  # per proof (4 proofs per profiled run)
                elems = lde_elements / per + 58 * n  # forward transforms of a proof: the LDE calls + the 58 split-polynomial columns
                tot = cols["wave_instructions"] * disp["k_ntt_cols_v5<4, 3, 3, false>"] + rows_["wave_instructions"] * disp["k_ntt_rows_v5<4, 3, 3, false>"]
                roofline_lde["valu"] = {"column_pass": cols, "row_pass": rows_, "insts_per_element": round(tot * 64 / elems, 1),
                                        "note": "lane-instructions per output element of the forward transform pair, all forward transforms of a proof"}
            r = valu.entry("k_rounds_mfma<2, 1, 256>", 4 * n) if rounds_mfma else valu.entry("k_rounds_split<1>", 4 * n)
            if r:
                roofline_rounds["valu"] = r
        roofline_stages = [roofline_lde] + ([] if coset_mode else [roofline_rounds])
        if args.mode == "prove":
            roofline_stages.append(entry("row hashes + Merkle tree of the trace commitment (k_hash_rows + k_merkle_level2 / k_merkle_quad)", ab["hash_rows"] + ab["merkle"],
                                         stage_ms["commit"], ["k_hash_rows", "k_merkle_level2", "k_merkle_level", "k_merkle_top", "k_merkle_quad"],
                                         "the one HBM-shaped stage; k_hash_rows moves exactly its algorithmic bytes; `traffic` is per kernel NAME and so also "
                                         "holds the small FRI-layer hashes and the composition / layer trees of a proof (about +0.5 GB)"))
            roofline_stages.append(entry("constraint evaluation stage (all launches)", ab["constraints"], stage_ms["constraints"],
                                         ["k_rounds_split<", "k_rounds_mfma<", "k_rounds_mfma_tables", "k_rounds_setup", "k_ec_split<", "k_final_split<", "k_final_hi<", "k_lin_split<", "k_lin_all",
                                          "k_coset_even_to_odd", "k_split_finish<"] if split else ["k_eval_fused<", "k_rounds_setup"],
                                         "`traffic` = the stage's own kernels; the interpolation and extension of its split polynomials run through the "
                                         "transform kernels and are counted in the first entry (about 4 GB)" if split else None))
        # ---- the bound that explains the times: vector-instruction issue.  Per kernel: wave-instructions and active cycles from the counter
        # pass, the static issue cost per instruction, and their ratio = the fraction of the kernel's cycles spent issuing VALU work ----
        issue_roofline = None
        if pts_headline and args.mode == "prove":
            pts = {"k_ntt_cols_v5<4, 3, 3, false>": 0, "k_ntt_rows_v5<4, 3, 3, false>": 0, "k_rounds_mfma<2, 1, 256>": 4 * n, "k_rounds_split<1>": 4 * n, "k_ec_split<1, false, 1>": 4 * n,
                   "k_ec_split<2, false, 1>": 4 * n, "k_ec_split<3, true, 1>": 4 * n, "k_ec_split<4, false, 1>": 4 * n, "k_final_split<1>": 4 * n,
                   "k_lin_all<true>": 4 * n, "k_split_finish<1>": 8 * n, "k_hash_rows": 0, "k_trace_schnorr_ec<false, 16>": 2 * n_tx * 511, "k_deep": n}
            issue_roofline = [e for e in (valu.entry(k, p) for k, p in pts.items()) if e is not None] or None
        out = {
            "metric": ("proofs/sec, state_transition AIR @ 2^%d steps (complete prove(): trace gen, LDE, Blake3 commitments, constraint "
                       "evaluation, composition, DEEP, FRI, %d queries)" % (log_n, args.queries)) if prove_mode else
                      ("proofs/sec, state_transition AIR @ 2^%d steps (hot path only: trace gen + LDE + Blake3 commitment + constraint evaluation)" % log_n),
            "value": round((1 if coset_mode else world) / (ms_per_step * 1e-3), 4),
            "unit": "proofs/s",
            "n_gpus": world, "rccl_ranks": rccl_ranks, **({"rehearsal": "%d ranks on ONE GPU, control collectives over gloo" % world} if rehearse else {}),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong" if coset_mode else "weak", "vs_baseline": None,
            "dtype": "u64 (63-bit prime field, Montgomery) + u32 (Blake3)", "data": "synthetic",
            "config": {"workload": "benches/state_transition.rs full TransactionAir, %d transactions = 2^%d steps, blowup 8, "
                                   "Merkle depth %d, %s, %s" % (n_tx, log_n, meta.depth, "Sha3_256" if hash_fn else "Blake3_256",
                                                                      ["no field extension", "quadratic extension", "cubic extension"][field_ext]),
                       "queries": args.queries, "proof_bytes": proof_len[0] or None,
                       "parallelism": ("one proof sharded by LDE coset over %d GPUs: RCCL all-gather of leaf digests and merged evaluations, "
                                       "broadcast of query positions, reduction of opened rows" % world)
                       if coset_mode else ("replica x%d (independent proofs per GPU, no data-path collective)%s" % (
                           world, ", %d proofs in flight per GPU" % inflight if inflight > 1 else "")),
                       "trace": "%d x 2^%d" % (WIDTH, log_n)},
            "stage_ms": {s: round(v, 3) for s, v in stage_ms.items()},
            "stage_gbs": ({"trace+interpolate+lde": round((ab["trace_gen"] + ab["interpolate"] + ab["lde"]) / ((stage_ms["trace"] + stage_ms["interpolate"] + stage_ms["lde"]) * 1e-3) / 1e9, 1),
                           "constraints": round(ab["constraints"] / (stage_ms["constraints"] * 1e-3) / 1e9, 1)} if prove_mode else
                          {s: round(ab[s] / (stage_ms[s] * 1e-3) / 1e9, 1) for s in stages}),
            "stage_note": ("cstark_tx_prove overlaps trace generation with the transforms: 'trace' = launch of the closed-form registers, "
                           "'interpolate' = registers 65..93 and 37..64 interpolated AND extended while the recurrences / curve ladders still run, "
                           "then registers 0..36 interpolated, 'lde' = their extension; the three add up to the time to the complete extended trace "
                           "(include/cstark.h, cstark_prove_stage_ms)") if prove_mode else None,
            "constraint_part_ms": None if coset_mode else {k: round(v, 3) for k, v in part_avg.items()},
            "constraint_part_note": ("all parts run on the even cosets only (split evaluation; final_add also on LDE coset 1); lin_c includes the "
                                     "extension of the 11 + 2 split polynomials to the odd cosets and the recombination over all cosets") if split else None,
            "roofline": roofline_lde,
            "roofline_stages": roofline_stages,
            "issue_roofline": issue_roofline,  # per kernel: the vector-instruction issue fraction (PmcValu), the bound DESIGN.md 5.2 names
        }
        if extra:
            out["extra_stage_ms"] = extra
        if world == 1 and args.mode == "prove" and inflight == 1 and not args.no_other_configs:
            try:
                out["two_proofs_in_flight"] = two_in_flight(meta, prover.options, local)
            except Exception as e:
                out["two_proofs_in_flight"] = {"failed": repr(e)}
        if world == 1 and args.mode == "prove" and not args.no_other_configs:
            try:  # after the timed region and on its own context; a report beside the headline, never a reason to lose it
                with torch.cuda.stream(torch.cuda.Stream()):  # a stream of its own, as a caller with several contexts would give it
                    ob = Backend(local)
                    out["other_configs"] = other_configs(ob)
                    ob.close()
            except Exception as e:
                out["other_configs"] = {"failed": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(meta_full, min(args.cpu_sample_tx, n_tx),
                                                   (args.queries, 8, 0, hash_fn, field_ext, 4, 256) if prove_mode else None)
            except Exception as e:  # the baseline is a report, never a reason to lose the measurement
                out["cpu_baseline"] = {"value": None, "unit": "proofs/s", "cores": len(os.sched_getaffinity(0)), "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Host-side mirror of the reference's prover interface for the state-transition AIR.

Names and argument meaning follow the reference so tests read like its own:
  ProofOptions(num_queries, blowup_factor, grinding_factor, hash_fn, field_extension, fri_folding_factor,
               fri_max_remainder)                          -- winterfell::ProofOptions as used at src/lib.rs:78-86
  TransactionMetadata                                      -- src/lib.rs:183-232 (field for field)
  TransactionProver(options).build_trace(tx_metadata)     -- src/prover.rs:20-98
  TransactionProver.get_pub_inputs(trace)                  -- src/prover.rs:106-129
  TransactionProver.commit_and_evaluate(...)               -- the hot half of Prover::prove (src/lib.rs:140):
        interpolate -> LDE -> Blake3 row hashes -> Merkle tree -> constraint evaluations
All device work goes through the C ABI (include/cstark.h); there is no CPU path here.
"""
import numpy as np

_P = 2**62 + 2**56 + 2**55 + 1  # the prime of f63::BaseElement
import torch

from . import _lib
from .backend import Backend, to_numpy_u64

TRACE_WIDTH = _lib.TX_TRACE_WIDTH
TRANSACTION_CYCLE_LENGTH = _lib.TX_CYCLE_LENGTH
PREV_TREE_ROOT_POS = 58  # src/merkle/constants.rs:45


class ProofOptions:
    """The 7 fields of winterfell::ProofOptions; defaults are the reference's (src/lib.rs:78-86)."""
    BLAKE3_256, SHA3_256 = 0, 1
    EXT_NONE, EXT_QUADRATIC, EXT_CUBIC = 0, 1, 2

    def __init__(self, num_queries=42, blowup_factor=8, grinding_factor=0, hash_fn=0, field_extension=0,
                 fri_folding_factor=4, fri_max_remainder=256):
        if blowup_factor & (blowup_factor - 1) or blowup_factor < 2:
            raise ValueError("blowup factor must be a power of two >= 2")
        self.num_queries, self.blowup_factor, self.grinding_factor = num_queries, blowup_factor, grinding_factor
        self.hash_fn, self.field_extension = hash_fn, field_extension
        self.fri_folding_factor, self.fri_max_remainder = fri_folding_factor, fri_max_remainder

    @property
    def log_blowup(self):
        return self.blowup_factor.bit_length() - 1


class TransactionMetadata:
    """Series of transfers in the account tree (src/lib.rs:183-194); arrays are uint64 in BaseElement memory form."""
    FIELDS = ("initial_roots", "final_root", "s_old_values", "r_old_values", "s_indices", "r_indices",
              "s_paths", "r_paths", "deltas", "sig_rx", "sig_s")

    def __init__(self, initial_roots, final_root, s_old_values, r_old_values, s_indices, r_indices, s_paths, r_paths,
                 deltas, sig_rx, sig_s):
        n = len(initial_roots)
        # "Enforce that all vectors are of equal length" (src/lib.rs:211-218)
        for name, a in (("s_old_values", s_old_values), ("r_old_values", r_old_values), ("s_indices", s_indices),
                        ("r_indices", r_indices), ("s_paths", s_paths), ("r_paths", r_paths), ("deltas", deltas),
                        ("sig_rx", sig_rx), ("sig_s", sig_s)):
            if len(a) != n:
                raise ValueError("%s has %d entries, expected %d" % (name, len(a), n))
        self.initial_roots = np.ascontiguousarray(initial_roots, np.uint64).reshape(n, 7)
        self.final_root = np.ascontiguousarray(final_root, np.uint64).reshape(7)
        self.s_old_values = np.ascontiguousarray(s_old_values, np.uint64).reshape(n, 14)
        self.r_old_values = np.ascontiguousarray(r_old_values, np.uint64).reshape(n, 14)
        self.s_indices = np.ascontiguousarray(s_indices, np.uint64).reshape(n)
        self.r_indices = np.ascontiguousarray(r_indices, np.uint64).reshape(n)
        self.s_paths = np.ascontiguousarray(s_paths, np.uint64)
        self.r_paths = np.ascontiguousarray(r_paths, np.uint64)
        self.deltas = np.ascontiguousarray(deltas, np.uint64).reshape(n)
        self.sig_rx = np.ascontiguousarray(sig_rx, np.uint64).reshape(n, 6)
        self.sig_s = np.ascontiguousarray(sig_s, np.uint8).reshape(n, 32)
        self.n_tx = n
        self.depth = self.s_paths.shape[1] - 1
        if self.s_paths.shape != (n, self.depth + 1, 7) or self.r_paths.shape != self.s_paths.shape:
            raise ValueError("authentication paths must be [n][depth+1][7]")
        if (self.depth + 1) & self.depth:
            raise ValueError("tree depth must be one less than a power of 2")  # src/lib.rs:102-105

    @classmethod
    def load(cls, path):
        z = np.load(path)
        return cls(*[z[f] for f in cls.FIELDS])

    @classmethod
    def build_random(cls, num_transactions, depth=15, seed=0x5EED):
        """Counterpart of TransactionMetadata::build_random (src/lib.rs:235-465), seeded: cstark_tx_witness_generate
        (host code of the library; no GPU involved)."""
        import ctypes as C
        n, d = int(num_transactions), int(depth)
        arrays = dict(initial_roots=np.zeros((n, 7), np.uint64), final_root=np.zeros(7, np.uint64),
                      s_old_values=np.zeros((n, 14), np.uint64), r_old_values=np.zeros((n, 14), np.uint64),
                      s_indices=np.zeros(n, np.uint64), r_indices=np.zeros(n, np.uint64),
                      s_paths=np.zeros((n, d + 1, 7), np.uint64), r_paths=np.zeros((n, d + 1, 7), np.uint64),
                      deltas=np.zeros(n, np.uint64), sig_rx=np.zeros((n, 6), np.uint64), sig_s=np.zeros((n, 32), np.uint8))
        s = _lib.TxWitnessStruct()
        s.n_tx, s.merkle_depth = n, d
        for f, a in arrays.items():
            setattr(s, f, a.ctypes.data_as(_lib.u8p if f == "sig_s" else _lib.u64p))
        _lib.check(_lib.load().cstark_tx_witness_generate(C.byref(s), C.c_uint64(seed)))
        return cls(*[arrays[f] for f in cls.FIELDS])

    def save(self, path):
        np.savez_compressed(path, **{f: getattr(self, f) for f in self.FIELDS})


class TransactionProver:
    """MI355X counterpart of src/prover.rs::TransactionProver plus the hot half of Prover::prove."""

    def __init__(self, options=None, backend=None):
        self.options = options or ProofOptions()
        self.backend = backend or Backend()
        self._bufs = {}

    def _buf(self, name, shape, dtype=torch.int64):
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.backend.device)
            self._bufs[name] = t
        return t

    # -- src/prover.rs:37-98
    def load_witness(self, tx_metadata):
        self.backend.upload_witness(tx_metadata)
        self.n_tx, self.depth = tx_metadata.n_tx, tx_metadata.depth
        self.trace_length = self.n_tx * TRANSACTION_CYCLE_LENGTH
        if self.trace_length & (self.trace_length - 1):
            raise ValueError("the number of transactions must be a power of two")

    def build_trace(self, tx_metadata=None):
        if tx_metadata is not None:
            self.load_witness(tx_metadata)
        return self.backend.build_trace(self._buf("trace", (TRACE_WIDTH, self.trace_length)))

    # -- src/prover.rs:106-129
    @staticmethod
    def get_pub_inputs(trace):
        first = to_numpy_u64(trace[PREV_TREE_ROOT_POS:PREV_TREE_ROOT_POS + 7, 0])
        last = to_numpy_u64(trace[PREV_TREE_ROOT_POS:PREV_TREE_ROOT_POS + 7, -1])
        return first, last  # initial_root, final_root

    # -- hot half of Prover::prove
    def extend_and_commit(self, trace, k0=0, nk=None):
        """interpolate -> LDE (cosets [k0,k0+nk)) -> leaf hashes; returns (coeffs, lde, nodes)."""
        b, lb = self.backend, self.options.log_blowup
        n = trace.shape[1]
        nk = (1 << lb) - k0 if nk is None else nk
        coeffs = b.interpolate_columns(trace, out=self._buf("coeffs", (TRACE_WIDTH, n)))
        lde = b.lde_columns(coeffs, lb, k0=k0, nk=nk, out=self._buf("lde", (nk, TRACE_WIDTH, n)))
        L = n << lb
        nodes = self._buf("nodes", (2 * L, 32), torch.uint8)
        b.hash_rows(lde, lb, k0=k0, leaves=nodes[L:])
        return coeffs, lde, nodes

    def build_tree(self, nodes):
        return self.backend.merkle_build(nodes)

    def evaluate_constraints(self, lde, coeffs_struct, pub_inputs4, k0=0, input_is_lde=False):
        """input_is_lde: `lde` is this prover's own extension of the whole trace (all cosets): degree-split evaluation, same values."""
        n = lde.shape[2]
        return self.backend.evaluate_constraints(lde, coeffs_struct, pub_inputs4, self.depth, self.options.log_blowup, k0=k0,
                                                 out=self._buf("combined", (lde.shape[0], n)), input_is_lde=input_is_lde)

    # -- Prover::prove as called at src/lib.rs:140 (trace generation included: the trace never leaves HBM)
    def prove(self, tx_metadata=None):
        """Full proof of the loaded transactions -> serialised proof bytes (layout: include/cstark.h)."""
        if tx_metadata is not None:
            self.load_witness(tx_metadata)
        return self.backend.prove(self.options)


def get_example_options():
    """The options of get_example (src/lib.rs:75-89)."""
    return ProofOptions(42, 8, 0, ProofOptions.BLAKE3_256, ProofOptions.EXT_NONE, 4, 256)


class TransactionExample:
    """src/lib.rs:92-150 without the random witness synthesis: holds options + metadata, proves on the GPU.
    Verification is CPU work in the reference (winterfell::verify) and outside this backend; the tests use the
    restated CPU verifier of the test suite."""

    def __init__(self, options, tx_metadata, backend=None):
        self.options, self.tx_metadata = options, tx_metadata
        self.prover = TransactionProver(options, backend)

    def prove(self):
        return self.prover.prove(self.tx_metadata)

    def pub_inputs(self):
        return self.tx_metadata.initial_roots[0], self.tx_metadata.final_root


# ---- the standalone examples of the reference: same prove() surface over cstark_air_prove ---------------------------------------
class _ResidentWitness:
    """An example's witness is uploaded by its first prove() and stays in device memory (any other upload on the same backend replaces
    it).  While it is resident the host arrays it was uploaded from are READ-ONLY (numpy refuses writes), so a changed witness can never
    be proved from the stale device copy: call invalidate() before editing them (or assign new arrays: that is detected)."""

    def _witness_arrays(self):
        raise NotImplementedError

    def _ensure_resident(self, upload):
        arrays = self._witness_arrays()
        key = tuple(id(a) for a in arrays)
        if getattr(self.backend, "resident", None) is not self or getattr(self, "_resident_key", None) != key:
            upload()
            self.backend.resident, self._resident_key = self, key
            self._was_writeable = [bool(a.flags.writeable) for a in arrays]
            for a in arrays:
                a.flags.writeable = False

    def invalidate(self):
        """the host witness is about to change: make it writable again and upload it anew at the next prove()"""
        if getattr(self, "_resident_key", None) is not None:
            for a, w in zip(self._witness_arrays(), self._was_writeable):
                if w:
                    a.flags.writeable = True
        self._resident_key = None
        if getattr(self.backend, "resident", None) is self:
            self.backend.resident = None


class MerkleExample(_ResidentWitness):
    """merkle::update::MerkleExample (src/merkle/update/mod.rs:36-127): proves the Merkle-update half of the transfers with the
    65-register MerkleAir."""

    def __init__(self, options, tx_metadata, backend=None):
        self.options, self.tx_metadata = options, tx_metadata
        self.backend = backend or Backend()

    def _witness_arrays(self):
        return [getattr(self.tx_metadata, f) for f in TransactionMetadata.FIELDS]

    def prove(self):
        """The witness is uploaded by the first call and stays resident (_ResidentWitness: read-only on the host until invalidate())."""
        self._ensure_resident(lambda: self.backend.upload_witness(self.tx_metadata))
        return self.backend.air_prove(Backend.AIR_MERKLE, self.options)

    def pub_inputs(self):
        return self.tx_metadata.initial_roots[0], self.tx_metadata.final_root


class SchnorrExample(_ResidentWitness):
    """schnorr::SchnorrExample (src/schnorr/mod.rs:52-186): messages [n][28] (public key || 16 elements) with signatures."""

    def __init__(self, options, messages, sig_rx, sig_s, backend=None):
        self.options, self.messages, self.sig_rx, self.sig_s = options, messages, sig_rx, sig_s
        self.backend = backend or Backend()

    @classmethod
    def build_random(cls, options, num_signatures, seed=0x5EED, backend=None):
        import ctypes as C
        n = int(num_signatures)
        msg, rx, s = np.zeros((n, 28), np.uint64), np.zeros((n, 6), np.uint64), np.zeros((n, 32), np.uint8)
        _lib.check(_lib.load().cstark_schnorr_witness_generate(C.c_uint32(n), C.c_uint64(seed), msg.ctypes.data_as(_lib.u64p),
                                                               rx.ctypes.data_as(_lib.u64p), s.ctypes.data_as(_lib.u8p)))
        return cls(options, msg, rx, s, backend)

    def _witness_arrays(self):
        return [self.messages, self.sig_rx, self.sig_s]

    def prove(self):
        """The witness is uploaded by the first call and stays resident (_ResidentWitness: read-only on the host until invalidate())."""
        self._ensure_resident(lambda: self.backend.upload_schnorr_witness(self.messages, self.sig_rx, self.sig_s))
        return self.backend.air_prove(Backend.AIR_SCHNORR, self.options)


class RescueExample:
    """RescueExample of benches/rescue.rs:25-102: a chain of `chain_length` Rescue hashes from the bench's seed 42..48 (held in memory
    form, R = 2^64).  The public inputs -- seed and result -- are read from the trace inside the prover, as get_pub_inputs does."""

    def __init__(self, chain_length, options, backend=None, seed=None):
        if chain_length & (chain_length - 1) or chain_length < 8:
            raise ValueError("chain length must a power of 2")   # benches/rescue.rs:34-37 (at least 8 links: 64 trace rows)
        self.options, self.chain_length = options, int(chain_length)
        self.backend = backend or Backend()
        r2 = pow(2, 64, _P)
        self.seed = np.array([(v * r2) % _P for v in range(42, 49)], np.uint64) if seed is None else np.ascontiguousarray(seed, np.uint64)

    def prove(self):
        return self.backend.rescue_prove(self.options, self.seed, self.chain_length)


class RangeProofExample:
    """range::RangeProofExample (src/range/mod.rs:28-110): `number` is a field element in memory form whose canonical value is
    below 2^63 (larger inputs are refused, src/range/tests.rs:54-62)."""

    def __init__(self, options, number, backend=None):
        self.options, self.number = options, int(number)
        self.backend = backend or Backend()

    def prove(self):
        return self.backend.air_prove(Backend.AIR_RANGE, self.options, self.number)

"""Device plumbing for the C ABI: torch owns HBM allocations and the HIP stream, the library does the work.

Field-element buffers are torch.int64 tensors holding the uint64 bit patterns (BaseElement memory form).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, u64p, u8p


def _np_u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def to_numpy_u64(t):
    return t.detach().cpu().contiguous().numpy().view(np.uint64)


class ProofBatch:
    """The proofs of one batched call: a sequence of bytes objects, materialised on access."""

    def __init__(self, buf, stride, lens):
        self._buf, self._stride, self._lens = buf, stride, lens

    def __len__(self):
        return len(self._lens)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self._buf[i * self._stride:i * self._stride + int(self._lens[i])].tobytes()

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __eq__(self, other):
        return list(self) == list(other)


class Backend:
    """One cstark_ctx bound to a torch device and the current torch stream."""

    def __init__(self, device=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.CstarkError(-2, "no HIP device visible (torch.cuda.is_available() is False); no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.current_stream(self.device)
        ctx = C.c_void_p()
        check(self.lib.cstark_ctx_create(C.c_int(self.device.index), C.c_void_p(self.stream.cuda_stream), C.byref(ctx)))
        self.ctx = ctx
        self.n_tx = 0
        self.depth = 0
        self.resident = None   # the example object whose witness is in device memory (prover.MerkleExample / SchnorrExample), if any

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.cstark_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(self.lib.cstark_ctx_synchronize(self.ctx))

    def empty_u64(self, *shape):
        return torch.empty(shape, dtype=torch.int64, device=self.device)

    def from_numpy_u64(self, a):
        return torch.from_numpy(_np_u64(a).view(np.int64)).to(self.device)

    @staticmethod
    def _ptr(t, typ=u64p):
        assert t.is_contiguous()
        return C.cast(C.c_void_p(t.data_ptr()), typ)

    # ---- K1 ----
    def upload_witness(self, w):
        """w: any object with the TransactionMetadata arrays as numpy attributes (see prover.TransactionMetadata)."""
        self.resident = None
        s = _lib.TxWitnessStruct()
        s.n_tx, s.merkle_depth = int(w.n_tx), int(w.depth)
        keep = []
        for f, typ in (("initial_roots", u64p), ("final_root", u64p), ("s_old_values", u64p), ("r_old_values", u64p),
                       ("s_indices", u64p), ("r_indices", u64p), ("s_paths", u64p), ("r_paths", u64p),
                       ("deltas", u64p), ("sig_rx", u64p), ("sig_s", u8p)):
            a = np.ascontiguousarray(getattr(w, f), dtype=np.uint8 if typ is u8p else np.uint64)
            keep.append(a)
            setattr(s, f, a.ctypes.data_as(typ))
        check(self.lib.cstark_tx_witness_upload(self.ctx, C.byref(s)))
        self.n_tx, self.depth = int(w.n_tx), int(w.depth)

    def build_trace(self, out=None):
        n = self.n_tx * _lib.TX_CYCLE_LENGTH
        if out is None:
            out = self.empty_u64(_lib.TX_TRACE_WIDTH, n)
        assert out.shape == (_lib.TX_TRACE_WIDTH, n) and out.dtype == torch.int64
        check(self.lib.cstark_tx_build_trace(self.ctx, self._ptr(out)))
        return out

    # ---- K2 / K3 ----
    def field_generator(self):
        self.lib.cstark_field_generator.restype = C.c_uint64
        return int(self.lib.cstark_field_generator())

    def field_lde_offset(self):
        self.lib.cstark_field_lde_offset.restype = C.c_uint64
        return int(self.lib.cstark_field_lde_offset())

    def interpolate_columns(self, evals, out=None):
        """evals: int64 [width, n] tensor (destroyed). Returns the coefficient tensor."""
        width, n = evals.shape
        if out is None:
            out = self.empty_u64(width, n)
        check(self.lib.cstark_interpolate_columns(self.ctx, self._ptr(evals), self._ptr(out), C.c_uint32(width),
                                                  C.c_uint32(n.bit_length() - 1)))
        return out

    def lde_columns(self, coeffs, log_blowup, offset=None, k0=0, nk=None, out=None):
        width, n = coeffs.shape
        nk = (1 << log_blowup) - k0 if nk is None else nk
        if out is None:
            out = self.empty_u64(nk, width, n)
        off = self.field_lde_offset() if offset is None else int(offset)
        check(self.lib.cstark_lde_columns(self.ctx, self._ptr(coeffs), self._ptr(out), C.c_uint32(width),
                                          C.c_uint32(n.bit_length() - 1), C.c_uint32(log_blowup), C.c_uint64(off),
                                          C.c_uint32(k0), C.c_uint32(nk)))
        return out

    # ---- K4 / K5 ----
    def empty_u8(self, *shape):
        return torch.empty(shape, dtype=torch.uint8, device=self.device)

    def hash_rows_fn(self, hash_fn, lde, log_blowup, k0=0):
        """cstark_hash_rows_fn: row hashes with Blake3_256 (0) or Sha3_256 (1)"""
        nk, width, n = lde.shape
        leaves = self.empty_u8(n << log_blowup, 32)
        check(self.lib.cstark_hash_rows_fn(self.ctx, C.c_uint32(hash_fn), self._ptr(lde), self._ptr(leaves, u8p), C.c_uint32(width),
                                           C.c_uint32(n.bit_length() - 1), C.c_uint32(log_blowup), C.c_uint32(k0), C.c_uint32(nk)))
        return leaves

    def merkle_build_fn(self, hash_fn, nodes):
        check(self.lib.cstark_merkle_build_fn(self.ctx, C.c_uint32(hash_fn), self._ptr(nodes, u8p), C.c_uint32((nodes.shape[0] // 2).bit_length() - 1)))
        return nodes

    def hash_rows(self, lde, log_blowup, k0=0, leaves=None):
        """lde: [nk, width, n]. leaves: uint8 [(n << log_blowup), 32]; rows of cosets [k0, k0+nk) are written."""
        nk, width, n = lde.shape
        if leaves is None:
            leaves = torch.zeros((n << log_blowup, 32), dtype=torch.uint8, device=self.device)
        check(self.lib.cstark_hash_rows(self.ctx, self._ptr(lde), self._ptr(leaves, u8p), C.c_uint32(width),
                                        C.c_uint32(n.bit_length() - 1), C.c_uint32(log_blowup), C.c_uint32(k0), C.c_uint32(nk)))
        return leaves

    def merkle_build(self, nodes):
        """nodes: uint8 [2 L, 32] with the leaves in the upper half; filled in place, nodes[1] is the root."""
        L2 = nodes.shape[0]
        check(self.lib.cstark_merkle_build(self.ctx, self._ptr(nodes, u8p), C.c_uint32((L2 // 2).bit_length() - 1)))
        return nodes

    # ---- K6 / K7 ----
    def evaluate_transitions(self, lde, depth, log_blowup=3, k0=0):
        nk, width, n = lde.shape
        out = self.empty_u64(nk, _lib.TX_NUM_CONSTRAINTS, n)
        check(self.lib.cstark_tx_evaluate_transitions(self.ctx, self._ptr(lde), self._ptr(out), C.c_uint32(depth),
                                                      C.c_uint32(n.bit_length() - 1), C.c_uint32(log_blowup), C.c_uint32(k0), C.c_uint32(nk)))
        return out

    def evaluate_constraints(self, lde, coeffs, pub_inputs, depth, log_blowup=3, k0=0, out=None, input_is_lde=False):
        """coeffs: _lib.TxCoeffsStruct (or any ctypes struct of the same layout); pub_inputs: 4 uint64.
        input_is_lde: the table is the extension of columns of degree < n over all 8 cosets (cstark_tx_evaluate_constraints_lde:
        same output, degree-split evaluation)."""
        nk, width, n = lde.shape
        if out is None:
            out = self.empty_u64(nk, n)
        pub = (C.c_uint64 * 4)(*[int(v) for v in pub_inputs])
        if input_is_lde:
            assert k0 == 0 and nk == 8 and log_blowup == 3
            check(self.lib.cstark_tx_evaluate_constraints_lde(self.ctx, self._ptr(lde), C.byref(coeffs), pub, self._ptr(out), C.c_uint32(depth),
                                                              C.c_uint32(n.bit_length() - 1)))
            return out
        check(self.lib.cstark_tx_evaluate_constraints(self.ctx, self._ptr(lde), C.byref(coeffs), pub, self._ptr(out), C.c_uint32(depth),
                                                      C.c_uint32(n.bit_length() - 1), C.c_uint32(log_blowup), C.c_uint32(k0), C.c_uint32(nk)))
        return out

    def evaluate_constraints_ext(self, lde, coeff_sets, pub_inputs, depth, log_blowup=3, k0=0):
        """The merged evaluations for 1..3 coefficient sets in one pass over the frame (cstark_tx_evaluate_constraints_ext):
        returns [m][nk][n]."""
        nk, width, n = lde.shape
        m = len(coeff_sets)
        out = self.empty_u64(m, nk, n)
        arr = (type(coeff_sets[0]) * m)(*coeff_sets)
        pub = (C.c_uint64 * 4)(*[int(v) for v in pub_inputs])
        check(self.lib.cstark_tx_evaluate_constraints_ext(self.ctx, self._ptr(lde), arr, C.c_uint32(m), pub, self._ptr(out), C.c_uint32(depth),
                                                          C.c_uint32(n.bit_length() - 1), C.c_uint32(log_blowup), C.c_uint32(k0), C.c_uint32(nk)))
        return out

    # ---- standalone sub-AIRs (MerkleAir, RangeProofAir) ----
    AIR_MERKLE, AIR_SCHNORR, AIR_RANGE, AIR_RESCUE_CHAIN = 1, 2, 3, 4

    def rescue_chain_build_trace(self, seed, chain_length):
        """cstark_rescue_chain_build_trace: 14 x 8 * chain_length (benches/rescue.rs:277-322); seed: 7 elements, memory form"""
        s = _np_u64(seed)
        out = self.empty_u64(14, 8 * chain_length)
        check(self.lib.cstark_rescue_chain_build_trace(self.ctx, self._hptr(s), C.c_uint32(chain_length), self._ptr(out)))
        return out

    def rescue_prove(self, options, seed, chain_length):
        """cstark_rescue_prove: complete RescueAir proof of a chain of `chain_length` hashes from `seed` (benches/rescue.rs:66-86)"""
        s = _np_u64(seed)
        o = self._options_struct(options)
        self.lib.cstark_tx_proof_size_bound.restype = C.c_size_t
        cap = 2 * self.lib.cstark_tx_proof_size_bound(C.c_uint32(max(1, chain_length // 128)), C.byref(o))
        buf = (C.c_uint8 * cap)()
        n = C.c_size_t(0)
        check(self.lib.cstark_rescue_prove(self.ctx, C.byref(o), self._hptr(s), C.c_uint32(chain_length), buf, C.c_size_t(cap), C.byref(n)))
        return bytes(memoryview(buf)[:n.value])

    def merkle_build_trace(self):
        out = self.empty_u64(65, self.n_tx * 512)
        check(self.lib.cstark_merkle_build_trace(self.ctx, self._ptr(out)))
        return out

    def range_build_trace(self, number_mont):
        out = self.empty_u64(2, 64)
        check(self.lib.cstark_range_build_trace(self.ctx, C.c_uint64(int(number_mont)), self._ptr(out)))
        return out

    def range_build_trace_bits(self, words, log_n):
        """cstark_range_build_trace_bits: the synthetic long accumulator (2 x 2^log_n); returns (trace, V mod p in memory form)."""
        w = _np_u64(words)
        out = self.empty_u64(2, 1 << log_n)
        num = C.c_uint64()
        check(self.lib.cstark_range_build_trace_bits(self.ctx, w.ctypes.data_as(u64p), C.c_uint32(log_n), self._ptr(out), C.byref(num)))
        return out, num.value

    def range_prove_bits(self, options, words, log_n):
        """cstark_range_prove_bits: complete RangeProofAir proof over 2^log_n rows."""
        w = _np_u64(words)
        o = _lib.OptionsStruct(options.num_queries, options.blowup_factor, options.grinding_factor, options.hash_fn,
                               options.field_extension, options.fri_folding_factor, options.fri_max_remainder)
        self.lib.cstark_tx_proof_size_bound.restype = C.c_size_t
        cap = 2 * self.lib.cstark_tx_proof_size_bound(C.c_uint32(max(1, (1 << log_n) // 1024)), C.byref(o))
        buf = (C.c_uint8 * cap)()
        n = C.c_size_t(0)
        check(self.lib.cstark_range_prove_bits(self.ctx, C.byref(o), w.ctypes.data_as(u64p), C.c_uint32(log_n), buf, C.c_size_t(cap), C.byref(n)))
        return bytes(memoryview(buf)[:n.value])

    def range_prove_batch(self, options, numbers):
        """cstark_range_prove_batch: one reference-shaped (64-row) range proof per element of `numbers` (memory form).  Returns a
        sequence of proofs (bytes on access: the call itself leaves them in one host buffer)."""
        nums = _np_u64(numbers)
        o = self._options_struct(options)
        self.lib.cstark_tx_proof_size_bound.restype = C.c_size_t
        stride = int(self.lib.cstark_tx_proof_size_bound(C.c_uint32(1), C.byref(o)))
        buf = np.empty(nums.size * stride, np.uint8)
        lens = np.zeros(nums.size, np.uint64)
        check(self.lib.cstark_range_prove_batch(self.ctx, C.byref(o), self._hptr(nums), C.c_uint32(nums.size), buf.ctypes.data_as(u8p), C.c_size_t(stride),
                                                lens.ctypes.data_as(C.POINTER(C.c_size_t))))
        return ProofBatch(buf, stride, lens)

    @staticmethod
    def _hptr(a, typ=u64p):
        return a.ctypes.data_as(typ)

    def air_shape(self, air, n_items=2):
        w, nc, na, lce = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(self.lib.cstark_air_shape(C.c_int(air), C.c_uint32(n_items), C.byref(w), C.byref(nc), C.byref(na), C.byref(lce)))
        return w.value, nc.value, na.value, lce.value

    def air_evaluate_transitions(self, air, lde, depth, log_blowup, k0=0):
        nk, width, n = lde.shape
        nc = self.air_shape(air)[1]
        out = self.empty_u64(nk, nc, n)
        check(self.lib.cstark_air_evaluate_transitions(self.ctx, C.c_int(air), self._ptr(lde), self._ptr(out), C.c_uint32(depth),
                                                       C.c_uint32(n.bit_length() - 1), C.c_uint32(log_blowup), C.c_uint32(k0), C.c_uint32(nk)))
        return out

    def air_combine(self, air, lde, evals, t_alpha, t_beta, b_alpha, b_beta, assertion_values, log_blowup, k0=0, n_items=2, avals_lde=None):
        nk, width, n = lde.shape
        out = self.empty_u64(nk, n)
        arrs = [_np_u64(a) for a in (t_alpha, t_beta, b_alpha, b_beta)]
        av = None if assertion_values is None else _np_u64(assertion_values)
        check(self.lib.cstark_air_combine(self.ctx, C.c_int(air), C.c_uint32(n_items), self._ptr(lde), self._ptr(evals),
                                          *[a.ctypes.data_as(u64p) for a in arrs], None if av is None else av.ctypes.data_as(u64p),
                                          None if avals_lde is None else self._ptr(avals_lde), C.c_uint32(0 if avals_lde is None else avals_lde.shape[1]),
                                          self._ptr(out), C.c_uint32(n.bit_length() - 1), C.c_uint32(log_blowup), C.c_uint32(k0), C.c_uint32(nk)))
        return out

    def merkle_evaluate_constraints(self, lde, depth, t_alpha, t_beta, b_alpha, b_beta, assertion_values, log_blowup, k0=0):
        """MerkleAir's combined constraint evaluations through the fused evaluator (cstark_merkle_evaluate_constraints)."""
        nk, width, n = lde.shape
        out = self.empty_u64(nk, n)
        arrs = [_np_u64(a) for a in (t_alpha, t_beta, b_alpha, b_beta, assertion_values)]
        check(self.lib.cstark_merkle_evaluate_constraints(self.ctx, C.c_uint32(depth), self._ptr(lde), *[a.ctypes.data_as(u64p) for a in arrs],
                                                          self._ptr(out), C.c_uint32(n.bit_length() - 1), C.c_uint32(log_blowup), C.c_uint32(k0),
                                                          C.c_uint32(nk)))
        return out

    def schnorr_evaluate_constraints(self, lde, aux_lde, t_alpha, t_beta, b_alpha, b_beta, avals_lde, log_blowup, k0=0, n_sig=2):
        """SchnorrAir's combined constraint evaluations through the fused evaluator (cstark_schnorr_evaluate_constraints)."""
        nk, width, n = lde.shape
        out = self.empty_u64(nk, n)
        arrs = [_np_u64(a) for a in (t_alpha, t_beta, b_alpha, b_beta)]
        check(self.lib.cstark_schnorr_evaluate_constraints(self.ctx, C.c_uint32(n_sig), self._ptr(lde), self._ptr(aux_lde),
                                                           *[a.ctypes.data_as(u64p) for a in arrs], self._ptr(avals_lde),
                                                           C.c_uint32(avals_lde.shape[1]), self._ptr(out), C.c_uint32(n.bit_length() - 1),
                                                           C.c_uint32(log_blowup), C.c_uint32(k0), C.c_uint32(nk)))
        return out

    # ---- standalone SchnorrAir ----
    def schnorr_evaluate_constraints_lde(self, lde, aux_lde, t_alpha, t_beta, b_alpha, b_beta, avals_lde, n_sig=2):
        """cstark_schnorr_evaluate_constraints_lde: all 8 cosets of genuine extensions -> the degree-split evaluation of the curve gadgets"""
        nk, width, n = lde.shape
        assert nk == 8 and width == 56
        out = self.empty_u64(8, n)
        arrs = [_np_u64(a) for a in (t_alpha, t_beta, b_alpha, b_beta)]
        check(self.lib.cstark_schnorr_evaluate_constraints_lde(self.ctx, C.c_uint32(n_sig), self._ptr(lde), self._ptr(aux_lde), *[a.ctypes.data_as(u64p) for a in arrs],
                                                               self._ptr(avals_lde), C.c_uint32(avals_lde.shape[1]), self._ptr(out), C.c_uint32(n.bit_length() - 1)))
        return out

    def upload_schnorr_witness(self, messages, sig_rx, sig_s):
        self.resident = None
        m, rx = _np_u64(messages), _np_u64(sig_rx)
        s = np.ascontiguousarray(sig_s, np.uint8)
        n_sig = m.shape[0]
        check(self.lib.cstark_schnorr_witness_upload(self.ctx, C.c_uint32(n_sig), m.ctypes.data_as(u64p), rx.ctypes.data_as(u64p),
                                                     s.ctypes.data_as(u8p)))
        self.n_tx = n_sig

    def schnorr_build_trace(self):
        out = self.empty_u64(56, self.n_tx * 512)
        check(self.lib.cstark_schnorr_build_trace(self.ctx, self._ptr(out)))
        return out

    def schnorr_aux_columns(self):
        out = self.empty_u64(19, self.n_tx * 512)
        check(self.lib.cstark_schnorr_aux_columns(self.ctx, self._ptr(out)))
        return out

    def schnorr_evaluate_transitions(self, lde, aux_lde, log_blowup, k0=0):
        nk, width, n = lde.shape
        out = self.empty_u64(nk, 56, n)
        check(self.lib.cstark_schnorr_evaluate_transitions(self.ctx, self._ptr(lde), self._ptr(aux_lde), self._ptr(out),
                                                           C.c_uint32(n.bit_length() - 1), C.c_uint32(log_blowup), C.c_uint32(k0), C.c_uint32(nk)))
        return out

    # ---- measurement ----
    CE_PARTS = ("rounds", "dbl_sG", "add_sG", "dbl_hP", "add_hP", "final_add", "lin_a", "lin_b", "lin_c")

    def set_part_timing(self, enable=True):
        check(self.lib.cstark_ctx_set_part_timing(self.ctx, C.c_int(1 if enable else 0)))

    def constraint_part_ms(self):
        ms = (C.c_float * 9)()
        check(self.lib.cstark_tx_constraint_part_ms(self.ctx, ms))
        return dict(zip(self.CE_PARTS, [float(v) for v in ms]))

    def lde_timing_ms(self):
        """(milliseconds, evaluations written) of all low-degree extensions since the last call (part timing on)"""
        ms, el = C.c_float(), C.c_uint64()
        check(self.lib.cstark_lde_timing_ms(self.ctx, C.byref(ms), C.byref(el)))
        return float(ms.value), int(el.value)

    def schnorr_assertion_polys(self, log_n):
        out = self.empty_u64(12, 1 << log_n)
        check(self.lib.cstark_schnorr_assertion_polys(self.ctx, self._ptr(out), C.c_uint32(log_n)))
        return out

    # ---- composition polynomial ----
    def composition_columns(self, combined, out=None):
        b, n = combined.shape
        if out is None:
            out = self.empty_u64(b, n)
        check(self.lib.cstark_composition_columns(self.ctx, self._ptr(combined), self._ptr(out), C.c_uint32(n.bit_length() - 1),
                                                  C.c_uint32(b.bit_length() - 1)))
        return out

    # ---- out-of-domain frame + DEEP composition ----
    def evaluate_polys_at(self, coeffs, points):
        width, n = coeffs.shape
        pts = _np_u64(points)
        out = np.zeros((pts.size, width), np.uint64)
        check(self.lib.cstark_evaluate_polys_at(self.ctx, self._ptr(coeffs), C.c_uint32(width), C.c_uint32(n.bit_length() - 1),
                                                pts.ctypes.data_as(u64p), C.c_uint32(pts.size), out.ctypes.data_as(u64p)))
        return out

    def deep_composition(self, trace_lde, comp_lde, z, ood_trace, ood_comp, alpha, beta, delta, deg_a, deg_b, log_blowup, k0=0, out=None):
        nk, width, n = trace_lde.shape
        nb = comp_lde.shape[1]
        if out is None:
            out = self.empty_u64(nk, n)
        arrs = [_np_u64(a) for a in (ood_trace, ood_comp, alpha, beta, delta)]
        check(self.lib.cstark_deep_composition(self.ctx, self._ptr(trace_lde), self._ptr(comp_lde), C.c_uint32(width), C.c_uint32(nb),
                                               C.c_uint64(int(z)), *[a.ctypes.data_as(u64p) for a in arrs], C.c_uint64(int(deg_a)),
                                               C.c_uint64(int(deg_b)), self._ptr(out), C.c_uint32(n.bit_length() - 1), C.c_uint32(log_blowup),
                                               C.c_uint32(k0), C.c_uint32(nk)))
        return out

    # ---- FRI ----
    def interleave_cosets(self, coset_major):
        b, n = coset_major.shape
        out = self.empty_u64(b * n)
        check(self.lib.cstark_interleave_cosets(self.ctx, self._ptr(coset_major), self._ptr(out), C.c_uint32(n.bit_length() - 1),
                                                C.c_uint32(b.bit_length() - 1)))
        return out

    def fri_fold4(self, evals, offset, alpha):
        N = evals.numel()
        out = self.empty_u64(N // 4)
        check(self.lib.cstark_fri_fold4(self.ctx, self._ptr(evals), self._ptr(out), C.c_uint32(N.bit_length() - 1), C.c_uint64(int(offset)),
                                        C.c_uint64(int(alpha))))
        return out

    def fri_fold(self, evals, offset, alpha, folding=4):
        """cstark_fri_fold: one FRI layer with folding factor 4, 8 or 16"""
        N = evals.numel()
        out = self.empty_u64(N // folding)
        check(self.lib.cstark_fri_fold(self.ctx, self._ptr(evals), self._ptr(out), C.c_uint32(N.bit_length() - 1), C.c_uint32(folding),
                                       C.c_uint64(int(offset)), C.c_uint64(int(alpha))))
        return out

    def fri_fold_ext(self, evals, offset, alpha, folding=4):
        """cstark_fri_fold_ext: evals [m][N] component-major, alpha: m memory-form words"""
        m, N = evals.shape
        out = self.empty_u64(m, N // folding)
        al = (C.c_uint64 * m)(*[int(v) for v in alpha])
        check(self.lib.cstark_fri_fold_ext(self.ctx, self._ptr(evals), self._ptr(out), C.c_uint32(N.bit_length() - 1), C.c_uint32(folding),
                                           C.c_uint64(int(offset)), C.c_uint32(m), al))
        return out

    def fri_commit_layer(self, evals, folding=4):
        """Merkle tree over the rows { e[i + t N/f] } of a layer; returns the node array (nodes[1] = root)."""
        N = evals.numel()
        q = N // folding
        nodes = torch.zeros((2 * q, 32), dtype=torch.uint8, device=self.device)
        self.hash_rows(evals.view(1, folding, q), 0, leaves=nodes[q:])
        self.merkle_build(nodes)
        return nodes

    # ---- whole proof -------------------------------------------------------------------------------------------
    def prove(self, options):
        """cstark_tx_prove on the uploaded witness; `options` has the 7 ProofOptions attributes.  Returns proof bytes."""
        o = _lib.OptionsStruct(options.num_queries, options.blowup_factor, options.grinding_factor, options.hash_fn,
                               options.field_extension, options.fri_folding_factor, options.fri_max_remainder)
        self.lib.cstark_tx_proof_size_bound.restype = C.c_size_t
        cap = self.lib.cstark_tx_proof_size_bound(C.c_uint32(self.n_tx), C.byref(o))
        buf = self._proof_buffer(cap)
        n = C.c_size_t(0)
        check(self.lib.cstark_tx_prove(self.ctx, C.byref(o), buf, C.c_size_t(cap), C.byref(n)))
        return bytes(memoryview(buf)[:n.value])

    def _proof_buffer(self, cap):
        """host buffer the library writes a proof into, kept across calls (a fresh ctypes array is zero-filled: 1 MB per proof)"""
        if getattr(self, "_pbuf_cap", 0) < cap:
            self._pbuf, self._pbuf_cap = (C.c_uint8 * cap)(), cap
        return self._pbuf

    # ---- one proof across several GPUs by LDE coset: the phases of cstark_tx_shard_* (driver: sharding.prove_sharded) -------------
    def _options_struct(self, options):
        return _lib.OptionsStruct(options.num_queries, options.blowup_factor, options.grinding_factor, options.hash_fn,
                                  options.field_extension, options.fri_folding_factor, options.fri_max_remainder)

    def shard_commit(self, options, k0, nk):
        """phase 1 -> the roots of this rank's subtrees (its nk leaves of every row), uint8 [1][n][32]"""
        self._shard_options = options
        n = self.n_tx * _lib.TX_CYCLE_LENGTH
        leaves = self.empty_u8(1, n, 32)
        o = self._options_struct(options)
        check(self.lib.cstark_tx_shard_commit(self.ctx, C.byref(o), C.c_uint32(k0), C.c_uint32(nk), self._ptr(leaves, u8p)))
        self._shard_nk = nk
        return leaves

    def shard_evaluate(self, leaves_all):
        """phase 2: all-gathered subtree roots [W][n][32] -> this rank's share of the merged constraint evaluations, int64 [R][n]
        (R = cstark_tx_shard_rows(nk): its cosets, or its even cosets + its share of the four odd ones; include/cstark.h)"""
        n = self.n_tx * _lib.TX_CYCLE_LENGTH
        assert tuple(leaves_all.shape) == (8 // self._shard_nk, n, 32) and leaves_all.dtype == torch.uint8
        self.lib.cstark_tx_shard_rows.restype = C.c_uint32
        out = self.empty_u64(int(self.lib.cstark_tx_shard_rows(C.c_uint32(self._shard_nk))), n)
        check(self.lib.cstark_tx_shard_evaluate(self.ctx, self._ptr(leaves_all.contiguous(), u8p), self._ptr(out), C.c_uint32(out.shape[0])))
        return out

    def shard_compose(self, combined_all):
        """phase 3 (the rank that owns coset 0): the ranks' shares [W * R][n] -> query positions, int32 [num_queries]"""
        nq = self._shard_options.num_queries
        pos = np.zeros(nq, np.uint32)
        n = self.n_tx * _lib.TX_CYCLE_LENGTH
        assert combined_all.dim() == 2 and combined_all.shape[1] == n
        check(self.lib.cstark_tx_shard_compose(self.ctx, self._ptr(combined_all.contiguous()), C.c_uint32(combined_all.shape[0]),
                                               pos.ctypes.data_as(C.POINTER(C.c_uint32))))
        return torch.from_numpy(pos.view(np.int32)).to(self.device)

    def shard_open_rows(self, positions):
        """phase 4: rows of the extended trace at `positions` that lie in this rank's cosets, each with the bottom log2(nk) siblings of
        its authentication path (zeros elsewhere), int64 [nq][94 + 4 log2(nk)]"""
        pos = np.ascontiguousarray(positions.detach().cpu().numpy().view(np.uint32))
        self.lib.cstark_tx_shard_open_words.restype = C.c_uint32
        rows = self.empty_u64(pos.size, int(self.lib.cstark_tx_shard_open_words(C.c_uint32(self._shard_nk))))
        check(self.lib.cstark_tx_shard_open_rows(self.ctx, pos.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(pos.size), self._ptr(rows)))
        return rows

    def shard_finish(self, rows):
        """phase 5 (the rank that ran phase 3): complete opened rows -> proof bytes"""
        o = self._options_struct(self._shard_options)
        self.lib.cstark_tx_proof_size_bound.restype = C.c_size_t
        cap = self.lib.cstark_tx_proof_size_bound(C.c_uint32(self.n_tx), C.byref(o))
        buf = (C.c_uint8 * cap)()
        n = C.c_size_t(0)
        check(self.lib.cstark_tx_shard_finish(self.ctx, self._ptr(rows.contiguous()), buf, C.c_size_t(cap), C.byref(n)))
        return bytes(memoryview(buf)[:n.value])

    PROVE_STAGES = ("trace", "interpolate", "lde", "commit", "constraints", "composition", "ood", "deep", "fri", "queries")

    def prove_stage_ms(self):
        ms = (C.c_float * len(self.PROVE_STAGES))()
        check(self.lib.cstark_prove_stage_ms(self.ctx, ms))
        return dict(zip(self.PROVE_STAGES, [float(v) for v in ms]))

    def air_prove(self, air, options, number=0):
        """cstark_air_prove: complete proof of the uploaded witness under MerkleAir / SchnorrAir, or of `number` (memory form)
        under RangeProofAir."""
        o = _lib.OptionsStruct(options.num_queries, options.blowup_factor, options.grinding_factor, options.hash_fn,
                               options.field_extension, options.fri_folding_factor, options.fri_max_remainder)
        self.lib.cstark_tx_proof_size_bound.restype = C.c_size_t
        cap = 2 * self.lib.cstark_tx_proof_size_bound(C.c_uint32(max(1, self.n_tx)), C.byref(o))
        buf = (C.c_uint8 * cap)()
        n = C.c_size_t(0)
        check(self.lib.cstark_air_prove(self.ctx, C.c_int(air), C.byref(o), C.c_uint64(number), buf, C.c_size_t(cap), C.byref(n)))
        return bytes(memoryview(buf)[:n.value])

"""One proof across several GPUs: sharding of the hot path by LDE coset.

The b-times larger LDE domain is the union of b cosets of the trace domain; LDE index i = b*j + k belongs to coset k
and the constraint frame pairs row i with row i + b (same coset), so extension, row hashing and constraint evaluation
need no halo.  Rank r owns cosets [k0, k0 + nk).  Two real exchange steps remain, both all-gathers:
  * subtree roots of the trace tree (n * 32 bytes per rank: a rank's nk leaves of a row are a complete subtree, whose bottom
    log2(nk) levels it hashes itself) -> every rank builds the upper levels of the Merkle tree
  * combined constraint evaluations (n * nk * 8 bytes per rank) -> composition polynomial
The functions work on any torch.distributed backend/device (RCCL on GPUs; gloo on CPU tensors in the tests).

prove_sharded() is the whole proof: the phases of cstark_tx_shard_* (include/cstark.h) with the collectives between them.  Besides the
two all-gathers, the query positions are broadcast from the rank that owns coset 0 (it alone runs composition / DEEP / FRI), and the
opened trace rows -- each lies in exactly one rank's cosets -- are summed onto that rank.
"""
import torch
import torch.distributed as dist


def coset_range(rank, world, blowup):
    """Cosets owned by `rank`: contiguous, balanced; world must divide blowup (or be 1)."""
    if blowup % world != 0:
        raise ValueError("world size %d must divide the blowup factor %d" % (world, blowup))
    nk = blowup // world
    return rank * nk, nk


def _via_host(t, group):
    """A gloo group moving device tensors (the one-GPU rehearsal: RCCL refuses two ranks on one device): stage through the host.
    RCCL groups and CPU tensors go straight through."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_gather_cosets(local, group=None):
    """local: [nk, ...] tensor of this rank's cosets (same shape on every rank) -> [world * nk, ...] in coset order."""
    world = dist.get_world_size(group)
    if _via_host(local, group):
        host = local.contiguous().cpu()
        out = torch.empty((world * host.shape[0],) + tuple(host.shape[1:]), dtype=host.dtype)
        dist.all_gather_into_tensor(out, host, group=group)
        return out.to(local.device)
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


def broadcast_(t, src, group=None):
    if _via_host(t, group):
        host = t.cpu()
        dist.broadcast(host, src=src, group=group)
        t.copy_(host)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


def reduce_sum_(t, dst, group=None):
    if _via_host(t, group):
        host = t.cpu()
        dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
    else:
        dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return t


def leaves_to_natural_order(coset_major):
    """[b, n, 32] digests per (coset k, row j) -> [n * b, 32] in LDE order i = b*j + k."""
    b, n = coset_major.shape[0], coset_major.shape[1]
    return coset_major.permute(1, 0, 2).reshape(n * b, 32).contiguous()


def prove_sharded(backend, options, group=None):
    """One complete proof of the witness uploaded on every rank's `backend`, sharded by LDE coset over the ranks of `group`.
    Returns the proof bytes on rank 0 (bit-identical to the single-GPU proof) and None elsewhere.  `backend` provides the phases
    shard_commit / shard_evaluate / shard_compose / shard_open_rows / shard_finish on its own device's tensors
    (certificate_stark_amd.backend.Backend over the C ABI; the CPU tests pass a stand-in built on the oracle)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    k0, nk = coset_range(rank, world, 8)
    leaves = all_gather_cosets(backend.shard_commit(options, k0, nk), group)       # C2: [world][n][32] subtree roots, rank-major
    combined = all_gather_cosets(backend.shard_evaluate(leaves), group)            # C3: [8][n] merged evaluations
    if rank == 0:
        positions = backend.shard_compose(combined)
    else:
        positions = torch.zeros(options.num_queries, dtype=torch.int32, device=combined.device)
    src = dist.get_global_rank(group, 0) if group is not None else 0
    broadcast_(positions, src, group)
    rows = backend.shard_open_rows(positions)
    reduce_sum_(rows, src, group)                                                   # every row (+ path bottom) is nonzero on exactly one rank
    return backend.shard_finish(rows) if rank == 0 else None

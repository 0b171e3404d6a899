"""One proof across several GPUs: sharding of the hot path by LDE coset.

The b-times larger LDE domain is the union of b cosets of the trace domain; LDE index i = b*j + k belongs to coset k
and the constraint frame pairs row i with row i + b (same coset), so extension, row hashing and constraint evaluation
need no halo.  Rank r owns cosets [k0, k0 + nk).  Two real exchange steps remain, both all-gathers:
  * leaf digests  (n * nk * 32 bytes per rank)  -> every rank (or rank 0) builds the Merkle tree
  * combined constraint evaluations (n * nk * 8 bytes per rank) -> composition polynomial
The functions work on any torch.distributed backend/device (RCCL on GPUs; gloo on CPU tensors in the tests).
"""
import torch
import torch.distributed as dist


def coset_range(rank, world, blowup):
    """Cosets owned by `rank`: contiguous, balanced; world must divide blowup (or be 1)."""
    if blowup % world != 0:
        raise ValueError("world size %d must divide the blowup factor %d" % (world, blowup))
    nk = blowup // world
    return rank * nk, nk


def all_gather_cosets(local, group=None):
    """local: [nk, ...] tensor of this rank's cosets (same shape on every rank) -> [world * nk, ...] in coset order."""
    world = dist.get_world_size(group)
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


def leaves_to_natural_order(coset_major):
    """[b, n, 32] digests per (coset k, row j) -> [n * b, 32] in LDE order i = b*j + k."""
    b, n = coset_major.shape[0], coset_major.shape[1]
    return coset_major.permute(1, 0, 2).reshape(n * b, 32).contiguous()

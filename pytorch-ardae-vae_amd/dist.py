"""Data-parallel host logic: shard the IMAGE batch, replicate parameters, average the two flat gradient
buffers with one all-reduce each (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).

Every quantity of the step is per image or per Monte-Carlo row, and both losses are plain means, so with equal shards
    grad(global batch) = mean over ranks of grad(local shard)
provided the entropy-gradient seed of ivae_ardae.py:834, `beta*g/(B*nz)`, is formed with the LOCAL B on each rank
(mean over ranks of 1/B_local == 1/B_global times the sum).  All nz rows of an image stay on its rank.

The exchange itself (round 4): `DpComm` - an RCCL communicator owned by the C ABI (`ardae_dp_comm_create`, include/ardae_hip.h), whose
in-place mean all-reduce is issued on the launch stream like a kernel and is CAPTURED into the step's HIP graphs.  `torch.distributed`
only ships the 128-byte unique id to the ranks; its own collectives (`allreduce_mean_`) remain the gloo rehearsal path (CPU tests, two
ranks sharing one GPU).
"""
import ctypes

import torch


def world_size(group=None):
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_world_size(group)
    return 1


def rank(group=None):
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_rank(group)
    return 0


def shard_rows(global_batch, group=None):
    """[begin, end) image indices of this rank's shard; the global batch must divide evenly (mean-of-means exactness)."""
    w, r = world_size(group), rank(group)
    if global_batch % w != 0:
        raise ValueError(f"global batch {global_batch} does not divide over {w} ranks")
    per = global_batch // w
    return r * per, (r + 1) * per


def allreduce_mean_(flat, group=None, force=False):
    """In-place average of a flat gradient buffer over the ranks (no-op for a single process, unless `force`: a one-rank group
    then still issues the collective - how the RCCL path is exercised on a single GPU)."""
    w = world_size(group)
    if w > 1 or (force and torch.distributed.is_available() and torch.distributed.is_initialized()):
        torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM, group=group)
        if w > 1:
            flat.mul_(1.0 / w)
    return flat


def entropy_seed_scale(std_scale, beta, local_batch, nz_model):
    """Factor applied to the score g to form d/dz of (s (z - z0)).backward(beta g / (B nz)) on one rank."""
    return std_scale * beta / float(local_batch * nz_model)


class DpComm:
    """One RCCL communicator per rank behind the C ABI (SURVEY 8(b) `dp_allreduce_flat`).  Construction is collective: every rank of
    `group` (default: the world) must call it, each with ITS device current (`torch.cuda.set_device`); rank 0's unique id travels through
    `torch.distributed.broadcast` (on the host for gloo, through a device tensor when the group only moves device memory).  With no process
    group: a one-rank communicator (what a single GPU can exercise)."""

    def __init__(self, group=None):
        from . import _lib as L
        self._L, self._h = L, ctypes.c_void_p()
        lib = L.lib()
        self.world, self.rank = world_size(group), rank(group)
        uid = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            L.check(lib.ardae_dp_unique_id(ctypes.c_void_p(uid.data_ptr())), "ardae_dp_unique_id")
        if self.world > 1:
            backend = str(torch.distributed.get_backend(group))
            if "gloo" in backend:
                torch.distributed.broadcast(uid, src=torch.distributed.get_global_rank(group, 0) if group is not None else 0, group=group)
            else:
                d = uid.cuda()
                torch.distributed.broadcast(d, src=torch.distributed.get_global_rank(group, 0) if group is not None else 0, group=group)
                uid = d.cpu()
        L.check(lib.ardae_dp_comm_create(ctypes.c_void_p(uid.data_ptr()), self.world, self.rank, ctypes.byref(self._h)), "ardae_dp_comm_create")
        self.backend = lib.ardae_dp_backend().decode()

    def query(self):
        """(ranks, rank, device) as RCCL reports them."""
        n, r, d = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self._L.check(self._L.lib().ardae_dp_comm_query(self._h, ctypes.byref(n), ctypes.byref(r), ctypes.byref(d)), "ardae_dp_comm_query")
        return n.value, r.value, d.value

    def allreduce_mean_(self, flat):
        """In-place mean over the ranks on the current stream (capturable)."""
        if not flat.is_contiguous():
            raise ValueError("the gradient buffer must be contiguous")
        self._L.check(self._L.lib().ardae_dp_allreduce_mean(self._h, self._L.ptr(flat), flat.numel(), self._L.stream_ptr()), "ardae_dp_allreduce_mean")
        return flat

    def close(self):
        if self._h:
            self._L.check(self._L.lib().ardae_dp_comm_destroy(self._h), "ardae_dp_comm_destroy")
            self._h = ctypes.c_void_p()

"""Data-parallel host logic: shard the IMAGE batch, replicate parameters, average the two flat gradient
buffers with one all-reduce each (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).

Every quantity of the step is per image or per Monte-Carlo row, and both losses are plain means, so with equal shards
    grad(global batch) = mean over ranks of grad(local shard)
provided the entropy-gradient seed of ivae_ardae.py:834, `beta*g/(B*nz)`, is formed with the LOCAL B on each rank
(mean over ranks of 1/B_local == 1/B_global times the sum).  All nz rows of an image stay on its rank.
"""
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

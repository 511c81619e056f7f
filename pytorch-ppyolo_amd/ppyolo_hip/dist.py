"""Multi-GPU inference: one process per GPU, batch sharded by rank, ONE all-gather of
fixed-size detection records at the end.

The reference has no multi-GPU path at all (its README lists it as not implemented), so this
is a new capability.  Images are independent in eval mode (BN uses running statistics, NMS
is per image), hence the forward needs no collective; the only exchange is the result:
per image `keep_top_k` rows of 6 floats + a count = 2404 B, 19.2 KB per GPU at 8 images --
latency-bound, so a single `all_gather_into_tensor` (RCCL over xGMI with backend "nccl",
gloo on CPU for tests) of one packed buffer is used, not a ring of small messages.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  No-op for a single process."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)            # one process per GPU
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(n_items, rank, world):
    """Contiguous, balanced split of a global batch: rank r owns [lo, hi)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class DetectionGatherer(object):
    """Packs (dets [n,keep,6], count [n]) into one [n, keep+1, 6] record block and all-gathers
    it.  Row `keep` of every image carries the count in column 0."""

    def __init__(self, n_local, keep_top_k, device, world=None):
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.n_local, self.keep = n_local, keep_top_k
        self.local = torch.zeros((n_local, keep_top_k + 1, 6), dtype=torch.float32, device=device)
        self.all = torch.zeros((self.world * n_local, keep_top_k + 1, 6), dtype=torch.float32, device=device)

    def gather(self, dets, count):
        self.local[:, :self.keep].copy_(dets)
        self.local[:, self.keep, 0].copy_(count)
        if self.world > 1:
            dist.all_gather_into_tensor(self.all, self.local)
        else:
            self.all.copy_(self.local)
        return self.all

    def unpack(self, packed=None):
        """-> list (global image order: rank-major) of [K,6] tensors / [[-1]*6] sentinel."""
        packed = self.all if packed is None else packed
        counts = packed[:, self.keep, 0].round().to(torch.int64).cpu().tolist()
        return [packed[i, :max(k, 1)].clone() for i, k in enumerate(counts)]

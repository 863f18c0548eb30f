"""Slice sharding of the per-slice encoder across the GPUs of one node.

The reference has no explicit collective (multi-GPU only through Lightning's implicit DDP); this
is new design for MI355X.  Slices are independent through the whole ViT and only the Slice
Transformer mixes them, so rank r of G encodes slices [r*ceil(D/G), (r+1)*ceil(D/G)) of every
volume and ONE all-gather of the slice embeddings ([B, D/G, E] fp32: 98 KB per volume at D=64)
precedes the replicated fusion stage.  The message is latency-bound, so it is issued as a single
``all_gather_into_tensor`` (RCCL over xGMI with backend "nccl"; gloo on CPU tensors in the tests).
Pure tensor plumbing: no arithmetic lives here.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


class SliceSharding:
    def __init__(self, group=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("slice sharding needs an initialised torch.distributed process group")
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def shard_range(self, D: int) -> Tuple[int, int, int]:
        """(first slice, one-past-last slice, padded shard length) of this rank."""
        return shard_range(D, self.world_size, self.rank)

    def _all_gather(self, out: torch.Tensor, local: torch.Tensor):
        """One collective on tensors of the group's backend (device tensors over RCCL)."""
        dist.all_gather_into_tensor(out, local, group=self.group)    # rank-major concatenation on dim 0

    def all_reduce_max(self, t: torch.Tensor) -> torch.Tensor:
        """Element-wise maximum over the ranks, in place (the fp8 scale table: every rank must quantise alike).  Built on the one
        collective `_all_gather`, so a transport that overrides it (tools/rehearsal.py) needs nothing else."""
        flat = t.contiguous().view(1, -1)
        out = torch.empty((self.world_size, flat.shape[1]), dtype=t.dtype, device=t.device)
        self._all_gather(out, flat)
        t.copy_(out.max(dim=0).values.view(t.shape))
        return t

    def all_gather_slices(self, local: torch.Tensor, D: int) -> torch.Tensor:
        """local [B, dpad, X] (this rank's shard, zero-padded to dpad) -> [B, D, X] on every rank."""
        B, dpad, X = local.shape
        local = local.contiguous()
        out = torch.empty((self.world_size * B, dpad, X), dtype=local.dtype, device=local.device)
        self._all_gather(out, local)
        out = out.view(self.world_size, B, dpad, X)
        return out.permute(1, 0, 2, 3).reshape(B, self.world_size * dpad, X)[:, :D].contiguous()


def shard_range(D: int, world_size: int, rank: int) -> Tuple[int, int, int]:
    dpad = (D + world_size - 1) // world_size
    d0 = min(rank * dpad, D)
    d1 = min(d0 + dpad, D)
    return d0, d1, dpad

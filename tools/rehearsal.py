"""Rehearsal transport for the slice-sharded path on a ONE-GPU box (tests and `MST_BENCH_SINGLE_DEVICE=1` only).

RCCL refuses two ranks on the same device, so a rehearsal runs several ranks on cuda:0 over gloo; gloo has no device
all_gather, so the (small) messages take the host path here.  The product transport is mst.parallel.SliceSharding on
backend "nccl" (= RCCL over xGMI): nothing in new-vit_amd/ imports this file."""
import torch
import torch.distributed as dist

from mst.parallel import SliceSharding


class HostStagedSharding(SliceSharding):
    def _all_gather(self, out: torch.Tensor, local: torch.Tensor):
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local.cpu(), group=self.group)
        out.copy_(host)

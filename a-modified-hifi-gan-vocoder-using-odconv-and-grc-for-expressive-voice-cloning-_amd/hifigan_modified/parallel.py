"""Data parallelism: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).

Samples are independent in every op of the path (GroupNorm and the ODConv attention are per-sample; there is no
BatchNorm), so the only exchange is one all-reduce(mean) of gradients per optimizer step.  `FlatAdamW` already holds the
gradients of a parameter group in one flat fp32 buffer; `GradSynchronizer` cuts it into buckets (default 32 MiB: the
33.6 MB `upsample_layers.0.0.kernels` is its own bucket) and issues asynchronous all-reduces on them.  xGMI is
point-to-point (7 links per GPU), so a few large buckets are preferred over many small ones; RCCL picks ring/direct.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, local_rank, world)."""
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def broadcast_parameters(module, src=0):
    """Identical initial weights on every rank."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)
        from . import ops
        ops.bump_param_epoch()          # .data writes do not bump _version: drop every cached cast / packed weight


def bucket_ranges(numel, bucket_elems):
    out, o = [], 0
    while o < numel:
        n = min(bucket_elems, numel - o)
        out.append((o, n))
        o += n
    return out


class GradSynchronizer:
    """Bucketed asynchronous all-reduce of a flat gradient buffer; `finish()` returns the 1/world scale that the
    optimizer folds into its update (no separate division pass)."""

    def __init__(self, bucket_mib=32):
        self.bucket_elems = bucket_mib * (1 << 20) // 4
        self.pending = []

    @property
    def world(self):
        return dist.get_world_size() if dist.is_initialized() else 1

    def start(self, flat_grad):
        if self.world == 1:
            return
        for o, n in bucket_ranges(flat_grad.numel(), self.bucket_elems):
            self.pending.append(dist.all_reduce(flat_grad[o:o + n], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        return 1.0 / self.world

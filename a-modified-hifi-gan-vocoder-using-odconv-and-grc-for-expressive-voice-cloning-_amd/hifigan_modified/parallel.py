"""Data parallelism: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).

Samples are independent in every op of the path (GroupNorm and the ODConv attention are per-sample; there is no
BatchNorm), so the only exchange is one all-reduce(mean) of gradients per optimizer step.  `FlatAdamW` already holds the
gradients of a parameter group in one flat fp32 buffer.  Two ways to reduce it:

* `GradSynchronizer` - after the backward: the whole buffer, cut into fixed-size buckets, asynchronous all-reduces.
* `OverlappedGradSync` - UNDER the backward: the arena is cut into buckets of whole parameters; a post-accumulate-grad hook per
  parameter counts arrivals, and the moment a bucket's last gradient exists that bucket is gathered into the flat buffer and its
  all-reduce is enqueued (RCCL runs it on its own stream, ordered after the gather by an event), while autograd keeps producing
  the remaining gradients on the compute stream.  The discriminator buckets thus reduce under the rest of the discriminator
  backward, the generator buckets under the rest of the generator backward.  xGMI is point-to-point
  (7 links per GPU), so buckets stay large (default 8 MiB; `upsample_layers.0.0.kernels`, 33.6 MB, is a bucket of its own).
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


class OverlappedGradSync:
    """Gradient all-reduce overlapped with the backward of ONE FlatAdamW parameter group (see the module docstring).

    usage per step:   sync.begin(); loss.backward(); scale = sync.finish(); opt.step(grad_scale=scale, gathered=True)
    `finish()` gathers and reduces whatever did not complete during the backward (parameters without a gradient count as
    zeros, exactly like FlatAdamW.gather_grads), waits for every bucket and returns 1/world.  With world size 1 it degrades to
    one gather.  `exposed_ms` accumulates the time the compute stream spent waiting in finish() (what the overlap did not hide)."""

    def __init__(self, opt, bucket_mib=8, timing=False):
        self.opt = opt
        be = max(1, bucket_mib * (1 << 20) // 4)
        # buckets = runs of whole parameters in arena order, closed once they reach the bucket size
        self.buckets, i0, acc = [], 0, 0
        for i, p in enumerate(opt.params):
            acc += p.numel()
            if acc >= be or i == len(opt.params) - 1:
                o0 = opt.offsets[i0]
                o1 = opt.numel if i == len(opt.params) - 1 else opt.offsets[i + 1]
                self.buckets.append((i0, i + 1, o0, o1 - o0))
                i0, acc = i + 1, 0
        self._bucket_of = {}
        for b, (a, e, _, _) in enumerate(self.buckets):
            for i in range(a, e):
                self._bucket_of[i] = b
        self._need = [e - a for a, e, _, _ in self.buckets]
        self._have = [0] * len(self.buckets)
        self._launched = [True] * len(self.buckets)     # nothing armed until begin()
        self._next = -1
        self._lazy = set()
        self._works = []
        self._armed = False
        self.timing = timing
        self.exposed_ms = 0.0
        self.reduced_bytes = 0
        self._ev = []
        for i, p in enumerate(opt.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))

    @property
    def world(self):
        return dist.get_world_size() if dist.is_initialized() else 1

    def _make_hook(self, i):
        def hook(_p):
            if not self._armed:
                return
            b = self._bucket_of[i]
            self._have[b] += 1
            # Collectives must be issued in the SAME order on every rank.  Gradients arrive roughly from the last parameter to the
            # first, so the order is fixed as descending bucket index: a complete bucket is launched only once every higher-index
            # bucket has been (whatever order the hooks fire in on this rank, the sequence of all-reduces is the same everywhere).
            # (Buckets that were still incomplete at finish() of the previous step - parameters without a gradient - are not waited
            #  for: they go last, in finish(); every rank runs the same model, so every rank skips the same ones.)
            while self._next >= 0 and (self._next in self._lazy or self._have[self._next] == self._need[self._next]):
                if self._next not in self._lazy:
                    self._launch(self._next)
                self._next -= 1
        return hook

    def begin(self):
        """Arm the hooks for the next backward of this parameter group."""
        self._have = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._next = len(self.buckets) - 1
        self._works = []
        self._armed = True

    def _launch(self, b):
        a, e, off, n = self.buckets[b]
        self.opt.gather_range(a, e)
        self._launched[b] = True
        if self.world > 1:
            self._works.append(dist.all_reduce(self.opt.flat_g[off:off + n], op=dist.ReduceOp.SUM, async_op=True))
            self.reduced_bytes += 4 * n

    def finish(self):
        self._armed = False
        self._lazy = {b for b in range(len(self.buckets)) if self._have[b] < self._need[b]}
        for b in range(len(self.buckets) - 1, -1, -1):          # the rest, in the same descending order
            if not self._launched[b]:
                self._launch(b)
        self._next = -1
        if self._works:
            if self.timing and torch.cuda.is_available():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for w in self._works:
                    w.wait()
                e1.record()
                self._ev.append((e0, e1))
            else:
                for w in self._works:
                    w.wait()
        self._works = []
        return 1.0 / self.world

    def collect_timing(self):
        """Sum the recorded waits (call after a device synchronize)."""
        for e0, e1 in self._ev:
            self.exposed_ms += e0.elapsed_time(e1)
        self._ev = []
        return self.exposed_ms

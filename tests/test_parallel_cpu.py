"""N>1 path on CPU (gloo, world_size 2): the bucketed gradient all-reduce used by the trainers
(hifigan_modified/parallel.py) averages a flat gradient buffer exactly and cuts buckets at the documented sizes."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, numel, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    from conftest import PKG, ROOT  # noqa: F401  (puts the package on sys.path)
    from hifigan_modified.parallel import GradSynchronizer, broadcast_parameters, init_distributed
    r, lr, w = init_distributed("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(rank)
    lin = torch.nn.Linear(8, 4)
    broadcast_parameters(lin)                    # identical weights afterwards
    wsum = lin.weight.detach().sum().item()
    g = torch.full((numel,), float(rank + 1))
    g[::7] = rank * 10.0
    sync = GradSynchronizer(bucket_mib=1)        # 262144 elements per bucket -> several buckets
    sync.start(g)
    n_buckets = len(sync.pending)
    scale = sync.finish()
    q.put((rank, wsum, (g * scale).tolist()[:16], float((g * scale).sum()), n_buckets, scale))
    dist.destroy_process_group()


def test_bucket_ranges():
    from hifigan_modified.parallel import bucket_ranges
    r = bucket_ranges(10, 4)
    assert r == [(0, 4), (4, 4), (8, 2)]
    assert bucket_ranges(4, 4) == [(0, 4)]
    # default 32 MiB buckets: the 8.39 M-element upsample_layers.0.0.kernels tensor (33.6 MB) spans two buckets at most
    assert len(bucket_ranges(8_388_608, 32 * (1 << 20) // 4)) == 1


def test_grad_allreduce_gloo_world2():
    world, numel = 2, 600_000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, numel, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, w0, head0, sum0, nb0, sc0), (r1, w1, head1, sum1, nb1, sc1) = res
    assert abs(w0 - w1) < 1e-9                    # broadcast made the weights identical
    assert head0 == head1 and abs(sum0 - sum1) < 1e-3
    assert nb0 == nb1 == 3 and sc0 == sc1 == 0.5
    # mean of rank values: (1+2)/2 = 1.5 everywhere except every 7th element: (0+10)/2 = 5
    assert head0[1] == 1.5 and head0[0] == 5.0 and head0[7] == 5.0


class _CpuArena:
    """The slice of FlatAdamW that OverlappedGradSync uses (params, offsets, numel, flat_g, gather_range), in plain torch on the
    CPU: lets the bucket / hook / collective logic run under gloo without a GPU."""

    def __init__(self, params):
        self.params = list(params)
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 7) // 8 * 8
        self.numel = off
        self.flat_g = torch.zeros(off)
        self.gathered = []

    def gather_range(self, i0, i1):
        self.gathered.append((i0, i1))
        for p, o in zip(self.params[i0:i1], self.offsets[i0:i1]):
            self.flat_g[o:o + p.numel()] = 0 if p.grad is None else p.grad.reshape(-1)


def _overlap_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from conftest import PKG, ROOT  # noqa: F401
    from hifigan_modified.parallel import OverlappedGradSync, init_distributed
    init_distributed("gloo")
    torch.manual_seed(0)                                   # same weights on both ranks
    net = torch.nn.Sequential(torch.nn.Linear(64, 300), torch.nn.Tanh(), torch.nn.Linear(300, 300), torch.nn.Tanh(),
                              torch.nn.Linear(300, 8))
    frozen = torch.nn.Parameter(torch.randn(5))            # a parameter that never gets a gradient (zeros in the flat buffer)
    arena = _CpuArena(list(net.parameters()) + [frozen])
    sync = OverlappedGradSync(arena, bucket_mib=0)         # bucket size 1 element -> every parameter is its own bucket
    sync2 = OverlappedGradSync(_CpuArena(list(net.parameters())), bucket_mib=1)   # 262144 elements: one bucket for this net
    torch.manual_seed(10 + rank)                           # different data per rank
    x, y = torch.randn(16, 64), torch.randn(16, 8)
    order = []
    orig = sync._launch
    sync._launch = lambda b: (order.append(b), orig(b))[1]
    # step 1: launches go in one fixed (descending) order, so the gradient-less last bucket holds everything back until finish(),
    # which learns that this bucket never completes; from step 2 on it is skipped during the backward and goes last
    sync.begin()
    ((net(x) - y) ** 2).mean().backward()
    assert order == [], order
    sync.finish()
    assert order == [6, 5, 4, 3, 2, 1, 0], order
    for p in net.parameters():
        p.grad = None
    order.clear()
    sync.begin()
    ((net(x) - y) ** 2).mean().backward()
    launched_in_backward = list(order)
    scale = sync.finish()
    local = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    flat = torch.cat([arena.flat_g[o:o + p.numel()] for p, o in zip(arena.params, arena.offsets)]) * scale
    q.put((rank, local.tolist(), flat.tolist(), launched_in_backward, order, len(sync.buckets), len(sync2.buckets), scale))
    dist.destroy_process_group()


def test_overlapped_bucket_allreduce_gloo_world2():
    """OverlappedGradSync: buckets are launched from grad-ready hooks DURING the backward, always in descending bucket order
    (the same on every rank, whatever order the hooks fire in), the rest in finish(); the reduced buffer is the mean of the two ranks' gradients; a gradient-less parameter contributes zeros."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, l0, f0, inb0, ord0, nb, nb2, sc), (_, l1, f1, inb1, ord1, _, _, _) = res
    assert sc == 0.5 and nb == 7 and nb2 == 1
    assert f0 == f1                                              # identical on both ranks
    mean = [(a + b) / 2 for a, b in zip(l0, l1)]
    n = len(mean)
    assert max(abs(a - b) for a, b in zip(f0[:n], mean)) < 1e-6   # = mean of the local gradients
    assert all(v == 0.0 for v in f0[n:])                          # the frozen parameter
    # the six trainable parameters were reduced from hooks, output layer first; only the frozen one waited for finish()
    assert inb0 == inb1 == [5, 4, 3, 2, 1, 0] and ord0[-1] == 6


def _order_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from conftest import PKG, ROOT  # noqa: F401
    from hifigan_modified.parallel import OverlappedGradSync, init_distributed
    init_distributed("gloo")
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(n)) for n in (40, 24, 8, 56, 16, 32)]
    arena = _CpuArena(params)
    sync = OverlappedGradSync(arena, bucket_mib=0)          # one bucket per parameter
    order = []
    orig = sync._launch
    sync._launch = lambda b: (order.append(b), orig(b))[1]
    torch.manual_seed(20 + rank)
    grads = [torch.randn_like(p) for p in params]
    # the gradients become ready in a DIFFERENT order on each rank (rank 0: first parameter first; rank 1: a shuffle)
    ready = list(range(len(params))) if rank == 0 else [3, 5, 0, 4, 1, 2]
    sync.begin()
    for i in ready:
        params[i].grad = grads[i]
        sync._make_hook(i)(params[i])
    scale = sync.finish()
    flat = torch.cat([arena.flat_g[o:o + p.numel()] for p, o in zip(arena.params, arena.offsets)]) * scale
    q.put((rank, order, flat.tolist(), torch.cat(grads).tolist()))
    dist.destroy_process_group()


def test_bucket_collectives_are_issued_in_one_order_on_every_rank():
    """Gradients that become ready in different orders on the two ranks still produce the same sequence of bucket all-reduces
    (descending bucket index - the order RCCL needs on one communicator), and the reduced buffer is the mean of the ranks' gradients."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_order_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, o0, f0, g0), (_, o1, f1, g1) = res
    assert o0 == o1 == [5, 4, 3, 2, 1, 0]
    assert f0 == f1
    assert max(abs(a - (b + c) / 2) for a, b, c in zip(f0, g0, g1)) < 1e-6

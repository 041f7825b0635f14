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

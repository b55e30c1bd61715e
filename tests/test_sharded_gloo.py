"""World-size-2 gloo test of the sharded-MSM combine path (no GPU): each rank's partial is
produced by the oracle here (standing in for bpgpu_msm_device), the exchange and the sum are
the product code in bulletproofs_amd/sharded.py."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from bulletproofs_amd import sharded
    from oracle import c_oracle as co
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pts, sc = co.msm_bench_inputs(0, n)
    lo, hi = sharded.shard_range(n, rank, world)
    norm = co.msm_pippenger(0, pts[lo:hi], sc[lo:hi])
    part = np.zeros(9, np.uint64)
    if norm[8:].any():
        part[:8] = norm[:8]
    else:
        part[8] = 1
    total = sharded.combine_partials(0, part)
    want = co.msm_pippenger(0, pts, sc)
    ok = bool((total[:8] == want[:8]).all()) and int(total[8]) == 0
    q.put((rank, ok, (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_combine_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, n = 2, 101
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res)
    ranges = sorted(r for _, _, r in res)
    assert ranges == [(0, 50), (50, 101)]

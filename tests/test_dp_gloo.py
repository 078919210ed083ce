"""world_size-2 gloo tests (CPU) of the data-parallel plumbing around the WC path:
the flat gradient bucket's all-reduce(mean), the start-up broadcast, and the sync-WC moment exchange."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import wc_oracle as o
        from wc_gan_amd.functional import _allreduce_
        from wc_gan_amd.train import FlatGradBucket, broadcast_state
        torch.manual_seed(rank)
        net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 2))
        net.register_buffer("moving_cov", torch.full((3, 3), float(rank + 1)))
        broadcast_state(net)
        same = [p.detach().clone() for p in net.parameters()] + [net.moving_cov.clone()]
        bucket = FlatGradBucket(net.parameters())
        x = torch.full((4, 6), float(rank + 1))
        net(x).sum().backward()
        local = bucket.flat.clone()
        bucket.allreduce_mean()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        ok_mean = torch.allclose(bucket.flat, sum(gathered) / world)
        # sync-WC: per-rank raw moments add up to the global-batch statistics
        rng = np.random.default_rng(7)
        X = o.synth_activation(rng, (64, 8))
        shard = X[rank::world]
        s, xtx, M = o.batch_moments(shard)
        ts, txtx = torch.tensor(s), torch.tensor(xtx)
        _allreduce_([ts, txtx], None)
        mu, sig = o.moments_to_stats(ts.numpy(), txtx.numpy(), M * world)
        mu_ref, sig_ref = o.moments_to_stats(*o.batch_moments(X))
        ok_sync = np.allclose(mu, mu_ref) and np.allclose(sig, sig_ref)
        q.put((rank, ok_mean, ok_sync, [t.sum().item() for t in same]))
    finally:
        dist.destroy_process_group()


def test_dp_bucket_broadcast_and_sync_moments():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "flat bucket all-reduce(mean) wrong"
    assert all(r[2] for r in res), "sync-WC moment exchange wrong"
    assert res[0][3] == res[1][3], "broadcast_state did not make the replicas identical"

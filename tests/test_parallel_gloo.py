"""World-size-2 gloo test of the multi-GPU host logic (CPU): batch sharding with no data-path
collective, output gather, flat-bucket gradient all-reduce, one-time parameter broadcast."""
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


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from snvc_amd import parallel as P
        assert P.world() == (rank, world)
        # 5 "crops" over 2 ranks: rank 0 gets 3, rank 1 gets 2 (uneven shard)
        crops = torch.arange(5 * 3, dtype=torch.float32).view(5, 3)
        extra = torch.arange(5)
        mine, mine_extra = P.shard([crops, extra])
        lo, hi = P.shard_range(5, rank, world)
        assert (lo, hi) == ((0, 3) if rank == 0 else (3, 5))
        assert torch.equal(mine, crops[lo:hi]) and torch.equal(mine_extra, extra[lo:hi])
        # per-rank "inference": no collective; then gather the small outputs
        out_local = mine * 2 + rank * 0            # independent units
        full = P.gather_outputs(out_local, 5)
        assert torch.equal(full, crops * 2)
        # gradient all-reduce: one flat bucket
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 2))
        if rank == 1:
            for p in net.parameters():
                p.data.add_(1.0)
        P.broadcast_parameters(net, src=0)
        ref = [p.detach().clone() for p in net.parameters()]
        loss = net(mine).pow(2).sum()
        loss.backward()
        local = [p.grad.clone() for p in net.parameters()]
        moved = P.all_reduce_gradients(net.parameters(), average=False)
        assert moved == sum(p.numel() for p in net.parameters()) * 4
        # reference: gradient of the full-batch loss on one process
        net2 = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 2))
        for p, r in zip(net2.parameters(), ref):
            p.data.copy_(r)
        net2(crops).pow(2).sum().backward()
        for p, p2, l in zip(net.parameters(), net2.parameters(), local):
            assert torch.allclose(p.grad, p2.grad, rtol=1e-5, atol=1e-5)
            assert not torch.allclose(l, p2.grad)          # the local gradient alone is different
        # a parameter without a gradient on ONE rank only (an unused head): the bucket keeps the same layout
        # on every rank, the missing gradient counts as zeros and comes back as the reduced value
        head = torch.nn.Linear(2, 2)
        P.broadcast_parameters(head, src=0)
        for p in list(net.parameters()) + list(head.parameters()):
            p.grad = None
        y = net(mine)
        (y.pow(2).sum() + (head(y).sum() if rank == 0 else 0.0)).backward()
        assert (head.weight.grad is None) == (rank == 1)
        params = list(net.parameters()) + list(head.parameters())
        moved = P.all_reduce_gradients(params, average=False)
        assert moved == sum(p.numel() for p in params) * 4
        g = [torch.zeros_like(head.weight.grad) for _ in range(world)]
        dist.all_gather(g, head.weight.grad)
        assert torch.equal(g[0], g[1]) and g[0].abs().sum() > 0
        # a parameter without a gradient on ANY rank stays without one (as DDP / a single process leave it)
        unused = torch.nn.Linear(2, 2)
        for p in params:
            p.grad = None
        net(mine).pow(2).sum().backward()
        P.all_reduce_gradients(params + list(unused.parameters()), average=True)
        assert unused.weight.grad is None and unused.bias.grad is None and head.weight.grad is None
        assert all(p.grad is not None for p in net.parameters())
        # the synthetic-bucket timer runs on any backend (gloo has no reduce-scatter: "auto" stays on all_reduce)
        us, used = P.all_reduce_bucket(1 << 16, "cpu", reps=2)
        assert used == "all_reduce" and us > 0
        q.put((rank, "ok"))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_grad_allreduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(0, "ok"), (1, "ok")]


def test_shard_range_partitions_exactly():
    from snvc_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


# ------------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[2] as bench.py shards it (run_cfg3 -> cfg3_shard_inputs / cfg3_shard_step), rehearsed at WORLD SIZE 8 on the CPU
# with a stub in place of the HIP model: 64 crops (8 per rank) and an uneven 65 (rank 0 takes 9).  What is checked is the host logic
# an 8-GPU run depends on and no 1-GPU box can show: every crop is evaluated exactly once, by the rank that owns it, the gathered
# result is in dim-0 order and identical on all ranks, and a rank with per_call not dividing its shard still covers it.
# ------------------------------------------------------------------------------------------------------------------------------
class _StubModel:
    """construct_voxel / trunk_3d with the model's shapes, computed on the CPU: occupancy[i] is a function of crop i's inputs alone."""

    def __init__(self, grid, rank):
        self.grid, self.rank, self.calls = grid, rank, []

    def construct_voxel_x3(self, *a):
        return None                                 # "split mode does not qualify": the fp32 gather is taken

    def construct_voxel(self, lf, rf, gl, gr):
        self.calls.append(lf.shape[0])
        n = lf.shape[0]
        v = gl.shape[2]
        base = lf.mean(dim=(1, 2, 3)) + 2.0 * rf.mean(dim=(1, 2, 3))                     # [n]
        return (base.view(n, 1) + 1e-3 * (gl[:, 0] - gr[:, 1])).view((n, 1) + self.grid) + 0.0 * v

    def trunk_3d(self, vox):
        return vox.mean(dim=(2, 3, 4)), torch.tanh(vox), None


def _cfg3_worker(rank, world, port, total, per_call, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from snvc_amd import parallel as P
        grid, F = (2, 3, 4), 2
        lo, hi = P.shard_range(total, rank, world)
        lf, rf, gl, gr = bench.cfg3_shard_inputs(lo, hi, grid, F, "cpu", fh=4, fw=4)
        assert lf.shape[0] == hi - lo
        m = _StubModel(grid, rank)
        out = bench.cfg3_shard_step(m, lf, rf, gl, gr, per_call, total, grid, gather=True)
        assert sum(m.calls) == hi - lo and max(m.calls, default=0) <= per_call        # this rank's crops, once, per_call at a time
        # the single-process evaluation of ALL crops, in order
        alf, arf, agl, agr = bench.cfg3_shard_inputs(0, total, grid, F, "cpu", fh=4, fw=4)
        ref = bench.cfg3_shard_step(_StubModel(grid, -1), alf, arf, agl, agr, total, total, grid, gather=False)
        assert out.shape == (total, 1) + grid and torch.equal(out, ref)
        local = bench.cfg3_shard_step(_StubModel(grid, rank), lf, rf, gl, gr, per_call, total, grid, gather=False)
        assert torch.equal(local, ref[lo:hi])                                           # no data-path collective was needed for it
        q.put((rank, hi - lo))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total,per_call", [(64, 8), (65, 8), (65, 4), (5, 8)])
def test_cfg3_sharding_at_world_size_8(total, per_call):
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cfg3_worker, args=(r, world, port, total, per_call, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    got = dict(q.get(timeout=5) for _ in range(world))
    assert sum(got.values()) == total and max(got.values()) - min(got.values()) <= 1
    assert got[0] == -(-total // world)                                                 # the first total % world ranks take one more

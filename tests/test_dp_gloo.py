"""CPU, world_size 2, gloo: the N>1 path of the data-parallel wrapper (sharding by image, flat-arena
gradient all-reduce == mean over ranks, coalesced scalar reduction, ragged all-gather)."""
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
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from semi_detr_amd import dp
    r, lr, w = dp.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    try:
        # --- image sharding: disjoint, covering, contiguous
        shard = list(dp.shard_images(10, rank, world))
        gathered = [None] * world
        dist.all_gather_object(gathered, shard)
        assert sorted(sum(gathered, [])) == list(range(10))
        # --- flat gradient arena + bucketed async all-reduce == mean of the per-rank gradients
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(64, 96), torch.nn.ReLU(), torch.nn.Linear(96, 8))
        net[0].bias.requires_grad_(False)           # frozen parameters are not part of the arena
        arena = dp.FlatGradArena(net.parameters(), bucket_bytes=4096)
        assert len(arena.buckets) > 2 and arena.numel == sum(p.numel() for p in net.parameters() if p.requires_grad)
        red = dp.GradAllReducer(arena.flat, arena.buckets)
        x = torch.randn(5, 64, generator=torch.Generator().manual_seed(100 + rank))
        arena.zero_()
        red.start()
        net(x).square().sum().backward()
        assert net[0].weight.grad.data_ptr() == arena.flat.data_ptr()          # grads ARE the arena
        local = arena.flat.clone()
        red.launch_ready(0.5)
        red.finish()
        both = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        assert torch.allclose(arena.flat, sum(both) / world, rtol=1e-6, atol=1e-7)
        # --- coalesced scalar reduce_mean
        sr = dp.ScalarReducer(torch.device("cpu"))
        sr.add(float(rank + 1))
        sr.add(torch.tensor(10.0 * (rank + 1)))
        a, b = sr.reduce_mean()
        assert abs(a.item() - 1.5) < 1e-6 and abs(b.item() - 15.0) < 1e-5
        # --- ragged all_gather (GMM costs)
        t = torch.arange(3 + 2 * rank, dtype=torch.float32) + 100 * rank
        g = dp.concat_all_gather_ragged(t)
        want = torch.cat([torch.arange(3 + 2 * k, dtype=torch.float32) + 100 * k for k in range(world)])
        assert torch.equal(g, want)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_dp_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def test_single_process_paths_are_noops():
    from semi_detr_amd import dp
    t = torch.arange(4.0)
    assert dp.concat_all_gather_ragged(t) is t
    red = dp.GradAllReducer(torch.ones(10), bucket_bytes=16)
    red.start()
    red.launch_ready(1.0)
    red.finish()
    assert torch.equal(red.flat, torch.ones(10))
    assert list(dp.shard_images(5, 0, 1)) == [0, 1, 2, 3, 4]

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


# ---------------------------------------------------------------------------------------------------
# FlatDDP: the drop-in for MMDistributedDataParallel(model, broadcast_buffers=False, find_unused_parameters=...)
# (detr_ssod/apis/train.py:88-93).  World 2, gloo, a small real model, real autograd driving the hooks:
#   * one optimizer step == torch's own DistributedDataParallel (what MMDistributedDataParallel is) BIT FOR BIT,
#     and == the single-process run on the concatenated batch (fp32 rounding);
#   * parameters/buffers are broadcast from rank 0 at construction; buffers are NOT synchronised afterwards;
#   * optimizer.zero_grad(set_to_none=True) between steps cannot detach gradients from the reduction;
#   * find_unused_parameters False raises, True reduces zeros; no_sync() accumulates locally.
# ---------------------------------------------------------------------------------------------------
class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(3, 8, 3, padding=1)
        self.bn = torch.nn.BatchNorm2d(8)
        self.fc1 = torch.nn.Linear(8 * 4 * 4, 64)
        self.fc2 = torch.nn.Linear(64, 10)
        self.extra = torch.nn.Linear(64, 3)          # only used when use_extra
        self.frozen = torch.nn.Linear(4, 4)
        for p in self.frozen.parameters():
            p.requires_grad_(False)

    def forward(self, x, use_extra=True):
        h = torch.relu(self.bn(self.conv(x))).flatten(1)
        h = torch.relu(self.fc1(h))
        out = self.fc2(h).square().mean()
        if use_extra:
            out = out + self.extra(h).mean()
        return out


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import copy
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from semi_detr_amd import dp
    dp.init_distributed(backend="gloo")
    try:
        torch.manual_seed(1234 + rank)              # DIFFERENT initial weights per rank: construction must broadcast
        base = _Net()
        flat = dp.FlatDDP(copy.deepcopy(base), broadcast_buffers=False, find_unused_parameters=False,
                          bucket_bytes=8192)
        ref = torch.nn.parallel.DistributedDataParallel(copy.deepcopy(base), broadcast_buffers=False,
                                                        find_unused_parameters=False, bucket_cap_mb=0.008)
        assert len(flat.arena.buckets) > 3
        w0 = [torch.zeros_like(flat.module.fc1.weight) for _ in range(world)]
        dist.all_gather(w0, flat.module.fc1.weight.data)
        assert torch.equal(w0[0], w0[1]), "parameters must be broadcast from rank 0 at construction"
        for a, b in zip(flat.module.state_dict().values(), ref.module.state_dict().values()):
            assert torch.equal(a, b)
        # single-process twin on the concatenated batch (built from the broadcast weights)
        twin = copy.deepcopy(flat.module)
        opts = [torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9) for m in (flat, ref, twin)]
        xs = [[torch.randn(4, 3, 4, 4, generator=torch.Generator().manual_seed(100 * step + r)) for r in range(world)]
              for step in range(3)]
        for step in range(3):
            for m, o in zip((flat, ref), opts[:2]):
                o.zero_grad(set_to_none=(step != 1))          # step 1 keeps the views, steps 0/2 drop them
                m(xs[step][rank]).backward()
            for pa, pb in zip(flat.module.parameters(), ref.module.parameters()):
                if pa.requires_grad:
                    assert torch.equal(pa.grad, pb.grad), "gradient mean must equal torch DDP's bit for bit"
                    assert pa.grad.data_ptr() >= flat.arena.flat.data_ptr()
                    assert pa.grad.data_ptr() < flat.arena.flat.data_ptr() + flat.arena.flat.numel() * 4
            for o in opts[:2]:
                o.step()
            # twin: both ranks' images in one batch; BatchNorm statistics differ (per-rank vs joint), so compare in
            # eval-free terms: only when step == 0 with bn in eval mode
        # --- single-process equivalence with BatchNorm frozen (statistics are per-rank in DDP by design)
        torch.manual_seed(7)
        base2 = _Net().eval()
        flat2 = dp.FlatDDP(copy.deepcopy(base2), find_unused_parameters=False, bucket_bytes=8192)
        twin2 = copy.deepcopy(flat2.module)
        o_f, o_t = torch.optim.SGD(flat2.parameters(), lr=0.05), torch.optim.SGD(twin2.parameters(), lr=0.05)
        for step in range(2):
            o_f.zero_grad()
            o_t.zero_grad()
            flat2(xs[step][rank]).backward()
            # mean over ranks of per-rank mean losses == mean loss of the concatenated batch (equal shard sizes)
            (sum(twin2(xs[step][r]) for r in range(world)) / world).backward()
            o_f.step()
            o_t.step()
            for pa, pb in zip(flat2.module.parameters(), twin2.parameters()):
                assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-6)
        # --- buffers are not synchronised after construction (broadcast_buffers=False)
        flat.module.bn.running_mean.add_(float(rank))
        flat(xs[0][rank]).backward()
        rm = [torch.zeros_like(flat.module.bn.running_mean) for _ in range(world)]
        dist.all_gather(rm, flat.module.bn.running_mean)
        assert not torch.equal(rm[0], rm[1])
        # --- unused parameters: error without the flag, zeros with it
        try:
            flat.zero_grad()
            flat(xs[0][rank], use_extra=False).backward()
            raise AssertionError("expected FlatDDP to report unused parameters")
        except RuntimeError as e:
            assert "did not receive a gradient" in str(e)
        flat3 = dp.FlatDDP(copy.deepcopy(base), find_unused_parameters=True, bucket_bytes=8192)
        flat3.zero_grad()
        flat3(xs[0][rank], use_extra=(rank == 0)).backward()       # only rank 0 uses `extra`
        g = flat3.module.extra.weight.grad
        gs = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(gs, g)
        assert torch.equal(gs[0], gs[1]) and gs[0].abs().sum() > 0    # mean of (rank 0's gradient, zeros)
        # --- a parameter that is unused in the synchronising backward keeps what an earlier no_sync() pass
        #     accumulated (reduced as it stands, exactly like torch DDP), and a globally unused parameter whose grad is
        #     None stays None (so weight decay / momentum never touch it)
        flat4 = dp.FlatDDP(copy.deepcopy(base), find_unused_parameters=True, bucket_bytes=8192)
        ref4 = torch.nn.parallel.DistributedDataParallel(copy.deepcopy(flat4.module), broadcast_buffers=False,
                                                         find_unused_parameters=True, bucket_cap_mb=0.008)
        for mdl in (flat4, ref4):
            for prm in mdl.parameters():
                prm.grad = None
            with mdl.no_sync():
                mdl(xs[1][rank], use_extra=True).backward()        # `extra` accumulates locally
            mdl(xs[2][rank], use_extra=False).backward()           # ... and is unused in the synchronising pass
        assert flat4.module.extra.weight.grad.abs().sum() > 0, "accumulated gradient was discarded"
        for pa, pb in zip(flat4.module.parameters(), ref4.module.parameters()):
            if pa.requires_grad:
                assert torch.allclose(pa.grad, pb.grad, rtol=1e-6, atol=1e-7), "accumulate-then-sync must equal torch DDP"
        for prm in flat4.parameters():
            prm.grad = None                                          # zero_grad(set_to_none=True)
        flat4(xs[0][rank], use_extra=False).backward()               # `extra` unused on EVERY rank
        assert flat4.module.extra.weight.grad is None and flat4.module.extra.bias.grad is None
        assert flat4.module.fc2.weight.grad is not None
        for prm in flat4.parameters():
            prm.grad = None
        flat4(xs[0][rank], use_extra=(rank == 1)).backward()         # used on rank 1 only: every rank gets the mean
        g4 = flat4.module.extra.weight.grad
        assert g4 is not None
        gs4 = [torch.zeros_like(g4) for _ in range(world)]
        dist.all_gather(gs4, g4)
        assert torch.equal(gs4[0], gs4[1]) and gs4[0].abs().sum() > 0
        # --- no_sync: local accumulation, no communication
        flat3.zero_grad()
        with flat3.no_sync():
            flat3(xs[1][rank]).backward()
        local = flat3.arena.flat.clone()
        alls = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(alls, local)
        assert not torch.equal(alls[0], alls[1])
        # --- coalesced reduce_mean of a loss()'s normalisers == per-scalar reduce_mean (mmdet dist_utils.py:67-73)
        vals = [float(rank + 1), torch.tensor([3.0 * (rank + 1)]), torch.tensor(7.0 + rank)]
        many = dp.reduce_mean_many(vals)
        one = [dp.reduce_mean(torch.as_tensor(v, dtype=torch.float32).reshape(-1)) for v in vals]
        for a, b in zip(many, one):
            assert torch.equal(a.reshape(-1), b.reshape(-1))
        # --- ragged gather keeps trailing dimensions
        t2 = torch.arange((2 + rank) * 3, dtype=torch.float32).view(2 + rank, 3) + 100 * rank
        g2 = dp.concat_all_gather_ragged(t2)
        assert g2.shape == (5, 3) and torch.equal(g2[:2], torch.arange(6.0).view(2, 3))
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()[-1500:] or repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_flat_ddp_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=200) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def test_flat_ddp_single_process():
    """World 1: hooks keep gradients in the arena (also after set_to_none), no collective is issued."""
    from semi_detr_amd import dp
    torch.manual_seed(0)
    net = dp.FlatDDP(_Net(), find_unused_parameters=True)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    x = torch.randn(2, 3, 4, 4)
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        net(x).backward()
        for p in net.module.parameters():
            if p.requires_grad:
                off = (p.grad.data_ptr() - net.arena.flat.data_ptr()) // 4
                assert 0 <= off < net.arena.flat.numel()
        opt.step()


# ---------------------------------------------------------------------------------------------------
# bench.py's N > 1 bookkeeping on CPU: the student's parameter list (scaled down 64x, buckets scaled alike) wrapped in
# FlatDDP and driven through mark_ready / finish exactly as bench.Workload.step does -- a counting wrapper around
# dist.all_reduce must see <= 4 bucket collectives per step and nothing else.
# ---------------------------------------------------------------------------------------------------
def _bench_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from semi_detr_amd import dp
    dp.init_distributed(backend="gloo")
    try:
        scale = 64
        student = bench.StudentParams(torch.device("cpu"), scale=scale)
        ddp = dp.FlatDDP(student, broadcast_buffers=False, find_unused_parameters=False, bucket_bytes=(64 << 20) // scale)
        assert abs(ddp.arena.numel - bench.GRAD_ELEMS // scale) < 1000
        calls = {"all_reduce": 0, "elems": 0, "other": 0}
        real_ar, real_ag, real_bc = dist.all_reduce, dist.all_gather, dist.broadcast

        def counted(t, *a, **k):
            calls["all_reduce"] += 1
            calls["elems"] += t.numel()
            return real_ar(t, *a, **k)

        def other(fn):
            def w(*a, **k):
                calls["other"] += 1
                return fn(*a, **k)
            return w
        dist.all_reduce, dist.all_gather, dist.broadcast = counted, other(real_ag), other(real_bc)
        try:
            ddp.profile = True                       # bench.py --gpus N: launch / completion stamps of every bucket
            for step in range(2):
                ddp.arena.flat.fill_(float(rank + 1))
                # the order bench.Workload.step uses: heads + decoder, encoder, backbone (step 1: in four slices, last layers
                # first, as with --backbone-ms), finish
                ddp.mark_ready(student.groups["heads"] + student.groups["decoder"])
                ddp.mark_ready(student.groups["encoder"])
                bb = list(reversed(student.groups["backbone"]))
                parts = 4 if step else 1
                for i in range(parts):
                    ddp.mark_ready(bb[len(bb) * i // parts:len(bb) * (i + 1) // parts])
                ddp.finish()
                assert torch.allclose(ddp.arena.flat, torch.full_like(ddp.arena.flat, (1 + world) / 2.0))
                prof = ddp.comm_profile()            # exposed wait + per-bucket launch -> complete
                assert prof["buckets"] == len(ddp.arena.buckets) == len(prof["bucket_ready_to_done_ms"]), prof
                assert prof["exposed_ms"] >= 0.0 and all(x >= 0.0 for x in prof["bucket_ready_to_done_ms"]), prof
        finally:
            dist.all_reduce, dist.all_gather, dist.broadcast = real_ar, real_ag, real_bc
        assert calls["all_reduce"] == 2 * len(ddp.arena.buckets) and len(ddp.arena.buckets) <= 4, calls
        assert calls["elems"] == 2 * ddp.arena.numel and calls["other"] == 0, calls
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()[-1500:] or repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_bench_bucket_bookkeeping_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=200) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results

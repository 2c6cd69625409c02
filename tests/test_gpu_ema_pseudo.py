"""GPU parity of the fused mean-teacher EMA and the pseudo-label filter."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _ulp(a, b):
    return int(np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)).max())


def test_ema_matches_torch_fixture_and_oracle(golden_ema):
    from semi_detr_amd import ema_update_
    z = golden_ema.z
    for mi in range(4):
        mom = float(z[f"m{mi}.momentum"])
        teachers = [torch.from_numpy(z[f"m{mi}.t{si}.teacher"].copy()).cuda() for si in range(6)]
        students = [torch.from_numpy(z[f"m{mi}.t{si}.student"].copy()).cuda() for si in range(6)]
        ema_update_(teachers, students, mom)
        for si in range(6):
            got = teachers[si].cpu().numpy()
            o = z[f"m{mi}.t{si}.teacher"].copy()
            oracle.ema_update(o, z[f"m{mi}.t{si}.student"], mom)
            assert np.array_equal(got, o), "HIP EMA must be bit-identical to the oracle"
            assert _ulp(got, z[f"m{mi}.t{si}.out"]) <= 1      # torch CPU: mul_ then add_(alpha)


def test_ema_vs_torch_on_the_same_gpu():
    """What the reference executes (mean_teacher.py:60-64) is `tgt.mul_(m).add_(src, alpha=1-m)` by torch ON THE GPU;
    compare with exactly that on this device: many tensor sizes, several momenta (incl. the warm-up values 0 and 0.5).
    torch's GPU add_(alpha) is one fma on the rounded product -- the same two roundings as ema.hip -- so the bar is
    bit equality, reported as ulps if it ever differs."""
    from semi_detr_amd import ema_update_
    torch.manual_seed(7)
    sizes = [1, 7, 64, 255, 4096, 8191, 8193, 65537, 1 << 20, 2359296]
    for mom in (0.0, 0.5, 0.9, 0.999, 0.9996, 1.0):
        ts = [torch.randn(n).cuda() * 3 for n in sizes]
        ss = [torch.randn(n).cuda() * 3 for n in sizes]
        want = [t.clone() for t in ts]
        for w, s in zip(want, ss):
            w.mul_(mom).add_(s, alpha=1 - mom)
        ema_update_(ts, ss, mom)
        torch.cuda.synchronize()
        for t, w in zip(ts, want):
            assert _ulp(t.cpu().numpy(), w.cpu().numpy()) <= 1, (mom, t.numel())


def test_ema_large_unaligned_and_flat():
    from semi_detr_amd import ema_update_, ema_update_flat_
    torch.manual_seed(1)
    sizes = [1, 3, 8191, 8192, 8193, 100003, 1 << 20]
    big_t = torch.randn(sum(sizes) + 16).cuda()
    big_s = torch.randn(sum(sizes) + 16).cuda()
    ts, ss, off = [], [], 1                       # offset 1 float -> 4-byte aligned only (scalar path)
    for n in sizes:
        ts.append(big_t[off:off + n])
        ss.append(big_s[off:off + n])
        off += n
    want = [t.cpu().numpy().copy() for t in ts]
    for w, s in zip(want, ss):
        oracle.ema_update(w, s.cpu().numpy(), 0.999)
    ema_update_(ts, ss, 0.999)
    for t, w in zip(ts, want):
        assert np.array_equal(t.cpu().numpy(), w)
    a, b = torch.randn(3_000_001).cuda(), torch.randn(3_000_001).cuda()
    w = a.cpu().numpy().copy()
    oracle.ema_update(w, b.cpu().numpy(), 0.9996)
    ema_update_flat_(a, b, 0.9996)
    assert np.array_equal(a.cpu().numpy(), w)


def test_mean_teacher_hook_semantics(golden_ema):
    """Schedule (mean_teacher.py:46-48), clone at iter 0 (:32-35), interval, frozen params included, buffers
    untouched, decay schedule (:52-58)."""
    from semi_detr_amd import MeanTeacher, ema_momentum

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = torch.nn.Conv2d(3, 8, 3)
            self.bn = torch.nn.BatchNorm2d(8)
            self.fc = torch.nn.Linear(8, 4)
            self.conv.weight.requires_grad_(False)       # frozen stem: still averaged

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.teacher, self.student = Net(), Net()

    class Runner:
        def __init__(self, model):
            self.model, self.iter = model, 0
            self.log_buffer = type("LB", (), {"output": {}})()

    z = golden_ema.z
    for wu in (0, 100):
        got = [ema_momentum(0.999, wu, int(s)) for s in z["sched_steps"]]
        assert np.array_equal(np.asarray(got), z[f"sched_wu{wu}"])

    torch.manual_seed(0)
    model = Model().cuda()
    model.student.bn.running_mean.fill_(3.0)
    runner = Runner(model)
    hook = MeanTeacher(momentum=0.999, interval=2, warm_up=0)
    hook.before_run(runner)
    for (_, ps), (_, pt) in zip(model.student.named_parameters(), model.teacher.named_parameters()):
        assert torch.equal(ps, pt)
    assert not torch.equal(model.student.bn.running_mean, model.teacher.bn.running_mean)   # buffers untouched
    ref_t = [p.detach().cpu().numpy().copy() for p in model.teacher.parameters()]
    for it in range(1, 6):
        runner.iter = it
        with torch.no_grad():
            for p in model.student.parameters():
                p.add_(torch.randn_like(p) * 0.1)
        hook.before_train_iter(runner)
        if it % 2 == 0:
            m = ema_momentum(0.999, 0, it)
            assert runner.log_buffer.output["ema_momentum"] == m
            for r, ps in zip(ref_t, model.student.parameters()):
                oracle.ema_update(r, ps.detach().cpu().numpy(), m)
        for r, pt in zip(ref_t, model.teacher.parameters()):
            assert np.array_equal(pt.detach().cpu().numpy(), r)
    hook2 = MeanTeacher(momentum=0.999, decay_intervals=[10, 20], decay_factor=0.1)
    runner.iter = 15
    hook2.after_train_iter(runner)
    assert abs(hook2.momentum - (1 - (1 - 0.999) / 0.1)) < 1e-12


def test_ema_fast_path_notices_repointed_storage():
    """the pointer table is reused while the Parameter OBJECTS are the same; a parameter re-pointed to new
    storage (`p.data = ...`, module.to(), flattening) keeps its identity.  EVERY pair's storage address is compared on every
    call, so the very next update already goes to the new storage, whichever parameter it was -- and never is freed memory
    touched (the table keeps the old storage alive until it is rebuilt)."""
    import semi_detr_amd.mean_teacher as mt
    from semi_detr_amd import ema_update_
    torch.manual_seed(0)
    ts = [torch.nn.Parameter(torch.randn(100 + i, device="cuda")) for i in range(40)]
    ss = [torch.nn.Parameter(torch.randn(100 + i, device="cuda")) for i in range(40)]
    for _ in range(3):
        ema_update_(ts, ss, 0.5)
    for victim in (0, 5, 38, 39):
        new = torch.randn_like(ss[victim].data)
        ss[victim].data = new                   # re-pointed: same Parameter object, new storage
        before = ts[victim].detach().clone()
        ema_update_(ts, ss, 0.0)                # momentum 0: teacher <- student
        assert torch.equal(ts[victim].detach(), new), ("the re-pointed storage was not picked up at once", victim)
        assert not torch.equal(before, new)
        newt = torch.randn_like(ts[victim].data)
        ts[victim].data = newt                  # the same for a re-pointed TEACHER parameter (module.to(), load_state_dict copies)
        ema_update_(ts, ss, 0.0)
        assert torch.equal(ts[victim].detach(), ss[victim].detach()), victim
    # a deleted model releases its table (weak references only)
    del ts, ss
    import gc
    gc.collect()
    a, b = [torch.randn(7, device="cuda")], [torch.randn(7, device="cuda")]
    ema_update_(a, b, 0.0)
    assert torch.equal(a[0], b[0]) and len(mt._table_cache) == 1


def test_mean_teacher_vs_reference_hook_sequences(golden_ema):
    """semi_detr_amd.MeanTeacher with the real kernel against the sequences the reference's own hook produced
    (tests/golden/ema.npz seq.*: before_run clone, warm-up, interval 2, decay intervals, wrapped model)."""
    from conftest import drive_mean_teacher_sequence
    z = golden_ema.z
    for name in z["seq.names"]:
        for it, logged, hook_mom, teachers, model in drive_mean_teacher_sequence(z, str(name), "cuda"):
            want = z[f"seq.{name}.logged_momentum"][it]
            assert (np.isnan(want) and np.isnan(logged)) or logged == want, (name, it)
            assert hook_mom == z[f"seq.{name}.hook_momentum"][it], (name, it)
            for i, t in enumerate(teachers):
                ref = z[f"seq.{name}.teacher{it + 1}.{i}"]
                ulp = np.abs(t.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64)).max()
                assert ulp <= 2, (name, it, i, int(ulp))
        assert np.array_equal(model.teacher.buf.cpu().numpy(), z[f"seq.{name}.buf_end"])


def test_pseudo_label_filter(golden_pseudo):
    from semi_detr_amd import filter_pseudo_labels
    names = golden_pseudo.names()
    props = [torch.from_numpy(golden_pseudo[n]["proposal"]).cuda() for n in names]
    labs = [torch.from_numpy(golden_pseudo[n]["labels"]).cuda() for n in names]
    boxes, labels, scores, thr = filter_pseudo_labels(props, labs, return_threshold=True)
    for i, n in enumerate(names):
        g = golden_pseudo[n]
        keep = g["keep"]
        assert np.array_equal(boxes[i].cpu().numpy(), g["proposal"][keep, :4]), n
        assert np.array_equal(scores[i].cpu().numpy(), g["proposal"][keep, 4]), n
        assert np.array_equal(labels[i].cpu().numpy(), g["labels"][keep]), n
        ok, othr = oracle.pseudo_label_filter(g["proposal"])
        assert np.array_equal(ok, keep)
        if g["proposal"].shape[0] > 1:
            np.testing.assert_allclose(thr[i].item(), g["thr"], rtol=2e-7)
        else:
            assert np.isnan(thr[i].item())


def test_pseudo_label_filter_large_ragged():
    from semi_detr_amd import filter_pseudo_labels
    rng = np.random.default_rng(4)
    props, labs = [], []
    for K in (1000, 0, 257, 256, 255, 3):
        xy = rng.random((K, 2)) * 800
        wh = rng.random((K, 2)) * 300 - 30
        props.append(np.concatenate([xy, xy + wh, rng.random((K, 1)) ** 2], -1).astype(np.float32))
        labs.append(rng.integers(0, 80, K).astype(np.int64))
    boxes, labels, scores = filter_pseudo_labels([torch.from_numpy(p).cuda() for p in props],
                                                 [torch.from_numpy(l).cuda() for l in labs])
    for p, l, b, la, s in zip(props, labs, boxes, labels, scores):
        keep, _ = oracle.pseudo_label_filter(p)
        assert np.array_equal(b.cpu().numpy(), p[keep, :4])
        assert np.array_equal(la.cpu().numpy(), l[keep])
        assert np.array_equal(s.cpu().numpy(), p[keep, 4])


def test_rccl_backend_initialises_and_reduces_on_this_gpu():
    """One-rank process group on backend 'nccl' (= RCCL on ROCm), run in a child process: the communicator binds to
    the GPU, an all-reduce and a barrier complete.  (The N > 1 collective path itself is covered with gloo in
    tests/test_dp_gloo.py; this checks that the RCCL side is usable on the box.)"""
    import os
    import subprocess
    import sys
    code = (
        "import os, torch, torch.distributed as dist\n"
        "os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29533')\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "import sys; sys.path.insert(0, os.getcwd())\n"
        "from semi_detr_amd import dp\n"
        "g = torch.arange(1000, dtype=torch.float32, device='cuda')\n"
        "w = dist.all_reduce(g, async_op=True); w.wait(); dist.barrier()\n"
        "print('communicator up', flush=True)\n"
        "r = dp.GradAllReducer(g, bucket_bytes=1024); r.start(); r.launch_ready(0.5); r.finish()\n"
        "s = dp.ScalarReducer(torch.device('cuda', 0)); s.add(3.0); s.add(torch.tensor(5.0, device='cuda'))\n"
        "assert [float(v) for v in s.reduce_mean()] == [3.0, 5.0]\n"
        "assert float(g[999]) == 999.0\n"
        # FlatDDP on the RCCL communicator: construction broadcast, autograd hooks -> bucketed all-reduce (ReduceOp.AVG on
        # RCCL) -> finalize, and the manual mark_ready / finish drive bench.py uses
        "net = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 8)).cuda()\n"
        "ddp = dp.FlatDDP(net, device_ids=[0], broadcast_buffers=False, bucket_bytes=16384, _reduce_when_alone=True)\n"
        "assert ddp._avg and len(ddp.arena.buckets) > 2\n"
        "x = torch.randn(16, 64, device='cuda')\n"
        "ref = [torch.autograd.grad(net(x).square().sum(), list(net.parameters()))]\n"
        "ddp.zero_grad(); ddp(x).square().sum().backward(); torch.cuda.synchronize()\n"
        "for p, g0 in zip(net.parameters(), ref[0]): assert torch.allclose(p.grad, g0, rtol=1e-5, atol=1e-6)\n"
        "ddp.zero_grad(); ddp.mark_ready(list(net.parameters())); ddp.finish(); torch.cuda.synchronize()\n"
        "assert float(ddp.arena.flat.abs().sum()) == 0.0\n"
        "dist.destroy_process_group(); print('rccl ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # The communicator's own bring-up (torch + RCCL, none of this repo's code) was seen to hang on one box in round 6 (300 s, the same
    # suite green on the boxes before and after): two attempts of 150 s; a bring-up that never completes is the BOX and skips, anything
    # that goes wrong once the communicator is up is ours and fails.
    last = None
    for attempt in range(2):
        proc = subprocess.Popen([sys.executable, "-c", code.replace("29533", str(29533 + attempt))], cwd=root, stdout=subprocess.PIPE,
                                stderr=subprocess.PIPE, text=True)
        try:
            stdout, stderr = proc.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            proc.kill()
            stdout, stderr = proc.communicate()
            last = (None, stdout, stderr)
            if "communicator up" in stdout:
                break
            continue
        last = (proc.returncode, stdout, stderr)
        break
    rc, stdout, stderr = last
    if rc is None and "communicator up" not in stdout:
        pytest.skip("RCCL communicator bring-up did not complete within 2 x 150 s on this box (torch.distributed init, before any "
                    "code of this repo runs)")
    assert rc == 0 and "rccl ok" in stdout, (rc, stdout[-500:], stderr[-2000:])

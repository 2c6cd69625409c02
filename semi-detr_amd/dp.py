"""Image-sharded data parallelism for the Semi-DETR hot path: one process per GPU, gradients are the ONLY
bulk exchange (RCCL all-reduce over xGMI; ``backend='nccl'`` is RCCL on ROCm).

What the reference does (detr_ssod/apis/train.py:88-93): wrap the model in MMDistributedDataParallel --
torch DDP with 25 MB buckets, ring all-reduce.  What this module does instead, MI355X-first:

  * every rank owns its images (MSDA, cost matrix, LSAP, pseudo-label filter, EMA are all per-image /
    replica-local: no activation or halo exchange);
  * gradients live in ONE flat fp32 arena in HBM (288 GB per GPU makes the copy-free layout cheap);
    parameters' ``.grad`` are views into it, so a bucket is just a slice -- no gather/scatter copies;
  * buckets are few and large (default 64 MiB: xGMI is 7 point-to-point links of ~153 GB/s per GPU, so
    per-collective latency, not bandwidth, is what erodes weak scaling) and are launched asynchronously
    from the back of the arena as backward produces them, overlapping with the remaining backward work;
  * the scalar normalisers the loss needs across ranks (``reduce_mean`` of num_total_pos etc.,
    mmdet/core/utils/dist_utils.py:67-73 -- ~60 one-element all-reduces per step in the reference) are
    coalesced into one small all-reduce by ``ScalarReducer``.

Works with the ``gloo`` backend on CPU tensors too (that is how tests/test_dp_gloo.py covers it).
"""
import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run contract)
    and create the default process group.  Returns (rank, local_rank, world_size)."""
    import os
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # SEMIDETR_DIST_BACKEND=gloo lets the multi-rank control flow be exercised on a one-GPU box
            backend = os.environ.get("SEMIDETR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)     # binds the communicator to this rank's GPU
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_images(num_images, rank, world):
    """Contiguous image shard of this rank: the path shards by image, nothing else."""
    per = (num_images + world - 1) // world
    lo = min(rank * per, num_images)
    return range(lo, min(lo + per, num_images))


class FlatGradArena:
    """One contiguous fp32 buffer holding every trainable parameter's gradient."""

    def __init__(self, params, device=None, bucket_bytes=64 << 20):
        self.params = [p for p in params if p.requires_grad]
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("FlatGradArena: only fp32 parameters (the reference trains in fp32)")
        device = device or (self.params[0].device if self.params else torch.device("cpu"))
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=device)
        off = 0
        self.offsets = []
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self.offsets.append(off)
            off += n
        per = max(1, bucket_bytes // 4)
        self.buckets = [(s, min(s + per, self.numel)) for s in range(0, self.numel, per)]

    def zero_(self):
        self.flat.zero_()


class GradAllReducer:
    """Mean all-reduce of a flat gradient arena, bucketed and asynchronous."""

    def __init__(self, flat, buckets=None, bucket_bytes=64 << 20, group=None):
        self.flat = flat
        n = flat.numel()
        per = max(1, bucket_bytes // flat.element_size())
        self.buckets = buckets or [(s, min(s + per, n)) for s in range(0, n, per)]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._pending = []
        self._next = len(self.buckets) - 1      # backward fills the arena from the back

    def start(self):
        self._pending = []
        self._next = len(self.buckets) - 1

    def launch_ready(self, fraction_done):
        """Call as backward progresses (fraction in [0,1] of the arena already final): launches every bucket
        that has become ready, last bucket first."""
        if self.world == 1:
            return
        total = len(self.buckets)
        ready = int(fraction_done * total + 1e-9)
        while self._next >= total - ready and self._next >= 0:
            s, e = self.buckets[self._next]
            self._pending.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True))
            self._next -= 1

    def finish(self):
        """Launch what is left, wait, and turn the sums into means."""
        if self.world == 1:
            return
        self.launch_ready(1.0)
        for w in self._pending:
            w.wait()
        self._pending = []
        self.flat.div_(self.world)


class ScalarReducer:
    """Coalesces the per-loss scalar all-reduces (mmdet ``reduce_mean``) into one collective."""

    def __init__(self, device, group=None):
        self.device, self.group, self.vals = device, group, []

    def add(self, value):
        self.vals.append(value if torch.is_tensor(value) else torch.tensor(float(value), device=self.device))
        return len(self.vals) - 1

    def reduce_mean(self):
        if not self.vals:
            return []
        buf = torch.stack([v.to(self.device, torch.float32).reshape(()) for v in self.vals])
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(buf.div_(dist.get_world_size(self.group)), op=dist.ReduceOp.SUM, group=self.group)
        self.vals = []
        return list(buf.unbind(0))


def concat_all_gather_ragged(t, group=None):
    """detr_ssod/models/utils/dist_utils.py:4-30: gather tensors whose FIRST dimension differs between ranks (the GMM
    cost lists, dino_detr_ssod.py:303) and concatenate them along dim 0; trailing dimensions are kept, as the
    reference does by padding dim 0.  Two small collectives: sizes, then padded payloads."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    if t.dim() == 0:
        raise ValueError("concat_all_gather_ragged: need at least one dimension to concatenate along")
    world = dist.get_world_size(group)
    size = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = t.new_zeros((mx,) + tuple(t.shape[1:]))
    pad[:t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad.contiguous(), group=group)
    return torch.cat([o[:n] for o, n in zip(out, sizes)], 0)


def reduce_mean(tensor, group=None):
    """mmdet/core/utils/dist_utils.py:67-73, same contract: mean over ranks of a (small) tensor, returned as a new
    tensor; the input itself when torch.distributed is not initialised."""
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor.div_(dist.get_world_size(group)), op=dist.ReduceOp.SUM, group=group)
    return tensor


def reduce_mean_many(values, device=None, group=None):
    """All the scalar normalisers of one ``loss()`` call in ONE collective.  The reference calls ``reduce_mean`` on
    ~60 one-element tensors per step (num_total_pos / cls_avg_factor / reg_avg_factor / sum_alignment_metrics per
    decoder layer and per loss branch, dino_detr_ssod_head.py:683-852); here the loss collects them first and asks
    once.  ``values``: list of python numbers or 0-d / 1-element tensors -> list of 0-d fp32 tensors (means over
    ranks, in order).  Device = the first tensor's, or ``device``."""
    if not values:
        return []
    if device is None:
        device = next((v.device for v in values if torch.is_tensor(v)), torch.device("cpu"))
    red = ScalarReducer(device, group)
    for v in values:
        red.add(v)
    return red.reduce_mean()


class FlatDDP(torch.nn.Module):
    """Drop-in for the wrap at detr_ssod/apis/train.py:88-93::

        model = MMDistributedDataParallel(model.cuda(), device_ids=[torch.cuda.current_device()],
                                          broadcast_buffers=False, find_unused_parameters=find_unused_parameters)

    Same constructor keywords, same ``.module`` attribute, ``train_step`` / ``val_step`` pass-through (what mmcv's
    runner calls), parameters AND buffers broadcast from rank 0 once at construction (torch DDP's
    ``_sync_module_states``), no per-forward buffer broadcast (only ``broadcast_buffers=False`` is supported -- the
    reference's setting).  What differs is how the gradient mean is produced, MI355X-first:

      * every trainable parameter's ``.grad`` is a view into ONE flat fp32 arena; a bucket is a slice of it, so the
        all-reduce runs in place on HBM the gradients already live in (no bucket copy-in / copy-out);
      * parameters are laid out in REVERSE registration order (the order backward produces them), buckets are few
        and large (64 MiB default; xGMI is point-to-point, per-collective latency is what costs weak scaling);
      * a ``register_post_accumulate_grad_hook`` per parameter counts its bucket down; a bucket is launched
        (asynchronously, in bucket order on every rank) the moment its last gradient has been accumulated, so the
        collectives overlap with the rest of backward; a callback queued on the autograd engine waits for them at
        the end of backward and turns sums into means (RCCL: ``ReduceOp.AVG``, no extra pass);
      * ``optimizer.zero_grad(set_to_none=True)`` (the torch >= 2 default, also what mmcv's OptimizerHook does)
        drops the views: the hook notices a foreign ``.grad``, copies it into the arena and re-installs the view,
        so gradients can never silently bypass the reduction.  ``zero_grad(set_to_none=False)`` or
        ``FlatDDP.zero_grad()`` keeps the views and avoids that copy.
    """

    def __init__(self, module, device_ids=None, output_device=None, dim=0, broadcast_buffers=False,
                 find_unused_parameters=False, bucket_bytes=64 << 20, process_group=None, _reduce_when_alone=False):
        super().__init__()
        if broadcast_buffers:
            raise NotImplementedError("FlatDDP: only broadcast_buffers=False (the reference's setting, "
                                      "detr_ssod/apis/train.py:91) is supported")
        self.module = module
        self.group = process_group
        self.find_unused_parameters = bool(find_unused_parameters)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # test aid: issue the collectives even in a one-rank group, so that the hook -> bucket -> RCCL all-reduce ->
        # finalize path can run on a real RCCL communicator on a one-GPU box (tests/test_gpu_ema_pseudo.py)
        self._alone = self.world == 1 and not (_reduce_when_alone and dist.is_initialized())
        self.require_backward_grad_sync = True
        self._sync_module_states()
        params = [p for p in module.parameters() if p.requires_grad]
        self.arena = FlatGradArena(list(reversed(params)), bucket_bytes=bucket_bytes)
        self.params = self.arena.params                              # arena order = reverse registration order
        self._views = [p.grad for p in self.params]
        # bucket index of every parameter; a parameter that straddles a bucket boundary counts for every bucket it
        # touches (buckets are equal slices of the arena, not parameter-aligned)
        self._param_buckets, self._bucket_need = [], [0] * len(self.arena.buckets)
        per = self.arena.buckets[0][1] - self.arena.buckets[0][0] if self.arena.buckets else 1
        for off, p in zip(self.arena.offsets, self.params):
            bs = list(range(off // per, (off + max(p.numel(), 1) - 1) // per + 1))
            self._param_buckets.append(bs)
            for b in bs:
                self._bucket_need[b] += 1
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._avg = (dist.is_initialized() and dist.get_backend(process_group) == "nccl")
        # opt-in timing of the collectives (bench.py --gpus N): when each bucket was launched / became complete on the
        # compute stream's timeline, and how long the end of backward waited for them
        self.profile = False
        self._stamps, self._last_stamps = None, None
        self._bitmap = None                          # find_unused_parameters: preallocated device tensor of the used-bitmap
        self._reset()
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad_ready) for p in self.params]

    # -- construction-time state sync (torch DDP._sync_module_states): parameters and buffers from rank 0
    def _sync_module_states(self):
        if self.world == 1 and not dist.is_initialized():
            return
        with torch.no_grad():
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                dist.broadcast(t.data, src=dist.get_global_rank(self.group, 0) if self.group else 0, group=self.group)

    def _reset(self):
        self._left = list(self._bucket_need)
        self._fired = [False] * len(self.params)
        self._next = 0
        self._pending = []
        self._callback_queued = False
        self._stamps = None

    # -- the hook autograd calls after a parameter's gradient has been accumulated
    def _on_grad_ready(self, p):
        i = self._index[id(p)]
        view = self._views[i]
        if p.grad is not view and (p.grad is None or p.grad.data_ptr() != view.data_ptr()):
            if p.grad is not None:                   # zero_grad(set_to_none=True) dropped the view: bring it home
                view.copy_(p.grad)
            p.grad = view
        if not self.require_backward_grad_sync or self._alone:
            return
        if not self._callback_queued:
            self._callback_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)
        if self._fired[i]:                           # second accumulation into the same parameter in one backward
            return
        self._fired[i] = True
        for b in self._param_buckets[i]:
            self._left[b] -= 1
        self._launch_ready()

    def _stamp(self):
        """A point on the compute stream's timeline (an event) for GPU arenas, the host clock otherwise (gloo on CPU)."""
        if self.arena.flat.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            return ev
        import time
        return time.perf_counter()

    def _launch_ready(self, force=False):
        nb = len(self.arena.buckets)
        while self._next < nb and (force or self._left[self._next] == 0):
            s, e = self.arena.buckets[self._next]
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            if self.profile:
                if self._stamps is None:
                    self._stamps = {"ready": [], "done": []}
                self._stamps["ready"].append(self._stamp())
            self._pending.append(dist.all_reduce(self.arena.flat[s:e], op=op, group=self.group, async_op=True))
            self._next += 1

    def comm_profile(self):
        """Timing of the LAST synchronising backward (profile = True): per bucket the time from its launch to its
        completion as seen by the compute stream, and `exposed_ms` = how long the compute stream stood waiting for
        collectives after the last gradient was final.  Synchronises the device."""
        st = self._last_stamps
        if not st:
            return None
        if self.arena.flat.is_cuda:
            torch.cuda.synchronize(self.arena.flat.device)
            ms = lambda a, b: a.elapsed_time(b)                          # noqa: E731
        else:
            ms = lambda a, b: (b - a) * 1e3                              # noqa: E731
        return {"bucket_ready_to_done_ms": [ms(r, d) for r, d in zip(st["ready"], st["done"])],
                "exposed_ms": ms(st["enter"], st["exit"]), "buckets": len(st["ready"])}

    def _finalize(self):
        """End of backward: reduce what is left (unused parameters), wait, sums -> means.

        Parameters that did not fire in THIS backward are handled like torch DDP does:
          * a gradient that is already there -- accumulated by an earlier ``no_sync()`` pass or by a multi-backward
            scheme such as mmcv's GradientCumulativeOptimizerHook -- is reduced AS IT STANDS, never zeroed;
          * a parameter whose ``.grad`` is ``None`` contributes zeros to the collective and keeps ``None`` afterwards
            unless some other rank used it (a used-bitmap is all-reduced next to the buckets, as torch DDP does), so
            an optimizer with weight decay / momentum does not touch globally unused parameters.
        """
        try:
            unused = [i for i, f in enumerate(self._fired) if not f]
            if unused and not self.find_unused_parameters:
                raise RuntimeError(
                    "FlatDDP: %d parameters did not receive a gradient in this backward pass. Pass "
                    "find_unused_parameters=True (as cfg.find_unused_parameters does at detr_ssod/apis/train.py:85) "
                    "if that is expected." % len(unused))
            were_none, bitmap, bm_work = [], None, None
            for i in unused:
                p, v = self.params[i], self._views[i]
                if p.grad is None:
                    v.zero_()                       # stale arena content must not enter the mean
                    were_none.append(i)
                elif p.grad is not v and p.grad.data_ptr() != v.data_ptr():
                    v.copy_(p.grad)                  # accumulated outside the arena: bring it home, keep its content
                    p.grad = v
            self._launch_ready(force=True)
            if self.profile and self._stamps is not None:
                self._stamps["enter"] = self._stamp()
            if self.find_unused_parameters:          # every rank takes part, whether or not IT has unused parameters;
                # issued AFTER the last bucket on every rank: collectives must be queued in the same order everywhere.
                # The bitmap lives in a preallocated device tensor fed from pinned host memory (no per-step
                # device allocation, no pageable copy); like torch DDP's local_used_map it is only read
                # back -- the one host sync of this path -- when some parameter of THIS rank had no gradient at all.
                if self._bitmap is None:
                    self._bitmap = torch.zeros(len(self.params), dtype=torch.int32, device=self.arena.flat.device)
                bitmap = self._bitmap
                host = torch.tensor(self._fired, dtype=torch.int32)
                if bitmap.is_cuda:
                    host = host.pin_memory()         # (torch's caching host allocator keeps the block until the copy is done)
                bitmap.copy_(host, non_blocking=True)
                bm_work = dist.all_reduce(bitmap, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            for w in self._pending:
                w.wait()
                if self.profile and self._stamps is not None:
                    self._stamps["done"].append(self._stamp())
            if self.profile and self._stamps is not None:
                self._stamps["exit"] = self._stamp()
                self._last_stamps = self._stamps
            if not self._avg:
                self.arena.flat.div_(self.world)
            if bm_work is not None:
                bm_work.wait()
                if were_none:
                    used = bitmap.tolist()
                    for i in were_none:
                        if used[i] > 0:              # some rank produced a gradient: the mean is this rank's too
                            self.params[i].grad = self._views[i]
        finally:
            self._reset()

    # -- manual driving (bench.py emulates a backward pass whose kernels are not autograd nodes)
    def mark_ready(self, params):
        """Tell the reducer that the gradients of ``params`` are final, exactly as autograd's hook would."""
        for p in params:
            i = self._index[id(p)]
            if not self._fired[i]:
                self._fired[i] = True
                for b in self._param_buckets[i]:
                    self._left[b] -= 1
        if not self._alone and self.require_backward_grad_sync:
            self._launch_ready()

    def finish(self):
        """Counterpart of ``mark_ready`` when no autograd engine callback runs."""
        if not self._alone and self.require_backward_grad_sync:
            self._finalize()
        else:
            self._reset()

    def zero_grad(self, set_to_none=False):
        """Zero the arena in one launch and keep the views installed."""
        self.arena.zero_()
        for p, v in zip(self.params, self._views):
            p.grad = v

    def no_sync(self):
        """Gradient accumulation without communication (torch DDP.no_sync)."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old = self.require_backward_grad_sync
            self.require_backward_grad_sync = False
            try:
                yield
            finally:
                self.require_backward_grad_sync = old
        return ctx()

    def forward(self, *inputs, **kwargs):
        return self.module(*inputs, **kwargs)

    def train_step(self, *inputs, **kwargs):     # mmcv MMDistributedDataParallel.train_step
        return self.module.train_step(*inputs, **kwargs)

    def val_step(self, *inputs, **kwargs):
        return self.module.val_step(*inputs, **kwargs)

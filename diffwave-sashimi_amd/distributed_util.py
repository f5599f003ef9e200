"""Data-parallel gradient exchange with the reference's surface
(``distributed_util.py:44-48,50-60,97-149``): ``init_distributed``,
``apply_gradient_allreduce(module)``, ``reduce_tensor``.

Re-designed for one process per MI355X over RCCL/xGMI instead of translating
the reference's pattern:

* initial weight sync is ONE flattened broadcast per dtype (the reference issues
  one broadcast per state_dict tensor: 686 for SaShiMi, ``distributed_util.py:107-110``);
* gradients are averaged in fixed-size flat buckets filled in reverse
  registration order, each all-reduced asynchronously the moment its last
  gradient has been accumulated, i.e. overlapped with the rest of backward
  (the reference flattens everything and does one blocking all-reduce after
  backward, ``:112-142``).  xGMI is point-to-point (7 links x ~153 GB/s), so a
  ring all-reduce is per-link bound: several ~25 MB buckets in flight keep all
  links busy while backward still runs; a single 94 MB buffer cannot overlap at all.
* the end-of-backward callback only waits for the outstanding buckets and divides
  by the world size.
* ZERO-COPY with the engine modules: the flat bucket buffers are the gradient arena.
  ``grad_views()`` hands out one view per parameter; the engine's backward has
  ``dws_model_get_grads`` write every gradient straight into its view (one launch) and
  returns those views, autograd adopts them as ``p.grad`` without a copy, the hooks find
  the gradient already in place, and after the in-place all-reduce + division ``p.grad``
  IS the averaged gradient: no per-parameter ``torch.empty``, no copy into the bucket,
  no copy back (round 3: 656 of each per step for SaShiMi).  Accumulating a further
  backward into a gradient that already lives in the arena stays in place as well (the
  engine hands the new gradient over in a fresh tensor, autograd adds it into the view);
  gradients that live in their own storage (torch-autograd backbones) still take the
  copy-in / copy-back path.

Backend "nccl" is RCCL on ROCm; the CPU tests use "gloo".
"""
import torch
import torch.distributed as dist
from torch.autograd import Variable

DEFAULT_BUCKET_BYTES = 25 * 1024 * 1024


def reduce_tensor(tensor, num_gpus):
    """Mean over ranks (``distributed_util.py:44-48``); used for loss logging (``train.py:135``)."""
    rt = tensor.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    rt /= num_gpus
    return rt


def init_distributed(rank, num_gpus, group_name, dist_backend, dist_url):
    """``distributed_util.py:50-60`` (``group_name`` is accepted and ignored: current
    torch has no such argument).  Pins the process to its GPU when there is one."""
    if torch.cuda.is_available():
        torch.cuda.set_device(rank % torch.cuda.device_count())
    import datetime
    # rank 0 checkpoints and generates while the other ranks wait in the next exchange (`train.py:166-183`):
    # give the group a timeout that covers a T-step sampling run
    dist.init_process_group(dist_backend, init_method=dist_url, world_size=num_gpus, rank=rank,
                            timeout=datetime.timedelta(minutes=60))


def _real_view(t):
    return torch.view_as_real(t) if t.is_complex() else t


def broadcast_state(module, src=0):
    """One flattened broadcast per dtype of every tensor in ``state_dict()``."""
    groups = {}
    for t in module.state_dict().values():
        if torch.is_tensor(t):
            groups.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, device), tensors in groups.items():
        views = [_real_view(t.data) for t in tensors]
        flat = torch.cat([v.reshape(-1) for v in views]) if len(views) > 1 else views[0].reshape(-1).clone()
        dist.broadcast(flat, src)
        off = 0
        for v in views:
            n = v.numel()
            v.copy_(flat[off:off + n].view_as(v))
            off += n
    if hasattr(module, "invalidate"):    # the .data writes above are invisible to the engine's version-counter cache
        module.invalidate()


class _Bucket:
    def __init__(self, params, device, dtype):
        self.params = params
        self.numel = sum(_real_view(p).numel() for p in params)
        self.flat = torch.zeros(self.numel, device=device, dtype=dtype)
        self.offsets = []
        off = 0
        for p in params:
            n = _real_view(p).numel()
            self.offsets.append((off, n))
            off += n
        self.index = -1          # position in GradientAllReducer.buckets (= the engine's gradient group)
        self.pending = len(params)
        self.ready = set()
        self.foreign = set()     # slots whose gradient lives in its own storage this backward: copied in, copied back
        self.work = None


class GradientAllReducer:
    """Bucketed, backward-overlapped gradient averaging for data-parallel training."""

    def __init__(self, module, bucket_bytes=DEFAULT_BUCKET_BYTES, sync_state=True):
        self.module = module
        self.world = dist.get_world_size()
        if sync_state:
            broadcast_state(module, 0)
        params = [p for p in module.parameters() if p.requires_grad]
        self._bucket_bytes = bucket_bytes
        self.buckets, self.where = [], {}
        cur, cur_bytes, cur_key = [], 0, None
        # reverse registration order ~ the order gradients become available in backward
        for p in reversed(params):
            key = (_real_view(p).dtype, p.device)
            nbytes = _real_view(p).numel() * _real_view(p).element_size()
            if cur and (key != cur_key or cur_bytes + nbytes > bucket_bytes):
                self._close(cur, cur_key)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
            cur_key = key
        if cur:
            self._close(cur, cur_key)
        self._callback_queued = False
        self._t_first, self._span = None, None
        # engine modules (models/engine.py) deliver their gradients bucket by bucket DURING backward and record an event per
        # bucket: set by engine_backward_done() for the backward in flight
        self._engine, self._engine_names, self._bwd_end, self._exposed = None, None, None, None
        self._side = None            # the stream the early all-reduces are launched from (waits for a bucket's event only)
        self._rebucketed = False
        self.overlapped_buckets = 0  # buckets of the last backward whose all-reduce was launched behind their own event
        self.last_stats = None      # {"in_place": slots found in the arena, "copied": slots copied in and back} of the last backward
        self._handles = []
        for p in params:
            self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _close(self, params, key):
        b = _Bucket(list(params), key[1], key[0])
        b.index = len(self.buckets)
        for i, p in enumerate(params):
            self.where[p] = (len(self.buckets), i)
        self.buckets.append(b)

    def engine_backward_done(self, module, names):
        """Called by the engine's autograd node right after ``dws_model_backward`` returned (every backward kernel and every
        bucket's hand-over copy + event is enqueued): from here on a complete, purely engine-written bucket is exchanged
        behind ITS event instead of behind the whole backward."""
        self._engine, self._engine_names = module, names
        self._engine_last = (module, names)
        self._bwd_end = self._stamp()          # the end of backward on the gradients' stream (exposed_ms)

    def _launch(self, b, early_ok=True):
        if self._t_first is None:      # the exchange of a backward starts here: stamp it (allreduce_ms)
            self._t_first = self._stamp()
        if early_ok and self._engine is not None and not b.foreign and b.flat.is_cuda:
            from . import _lib
            if self._side is None:
                self._side = torch.cuda.Stream(device=b.flat.device)
            _lib.check(_lib.load().dws_model_grad_group_wait(self._engine._handle, b.index, self._side.cuda_stream))
            with torch.cuda.stream(self._side):   # the collective's stream waits for what `_side` has queued: the bucket's event
                b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, async_op=True)
            self.overlapped_buckets += 1
            return
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, async_op=True)

    def bucket_ready_points(self):
        """Per bucket, the flush point of the engine's last backward after which the bucket's last gradient was final
        (``dws_model_grad_ready_seq``), and the number of flush points of that backward: a bucket whose point is below the last
        one left -- copy, event, all-reduce -- while backward was still running.  None without an engine backward."""
        import ctypes
        from . import _lib
        if getattr(self, "_engine_last", None) is None:
            return None
        module, names = self._engine_last
        params = dict(module.named_parameters())
        n = len(names)
        seq = (ctypes.c_int32 * n)()
        _lib.check(_lib.load().dws_model_grad_ready_seq(module._handle, n, (ctypes.c_char_p * n)(*[x.encode() for x in names]), seq))
        ready = {id(params[x]): int(seq[i]) for i, x in enumerate(names)}
        return {"bucket_ready_point": [max([ready.get(id(p), -1) for p in b.params]) for b in self.buckets],
                "last_point": max(ready.values())}

    def exposed_ms(self):
        """Milliseconds from the end of the engine's backward (its last kernel) to the completion of the last ``wait()`` of
        the most recent backward: the part of the exchange the step could not hide.  None without an engine backward."""
        if self._exposed is None:
            return None
        t0, t1 = self._exposed
        t1.synchronize()
        return t0.elapsed_time(t1)

    def _rebucket_by_readiness(self):
        """Once, after the first engine backward: re-cut the buckets in the order the engine's backward FINISHES the
        gradients (``dws_model_grad_ready_seq``; the fc_t rows and the embedding MLP of every block come out of one stacked
        GEMM at the very end, registration order would put one of them into every bucket and nothing could leave early).
        Every rank runs the same graph, so every rank cuts the same buckets."""
        import ctypes
        from . import _lib
        names = self._engine_names
        params = dict(self._engine.named_parameters())
        mine = [p for b in self.buckets for p in b.params]
        if set(id(p) for p in mine) != set(id(params[n]) for n in names if params[n].requires_grad):
            return                                   # foreign parameters in the arena: keep the registration order
        n = len(names)
        seq = (ctypes.c_int32 * n)()
        _lib.check(_lib.load().dws_model_grad_ready_seq(self._engine._handle, n, (ctypes.c_char_p * n)(*[x.encode() for x in names]), seq))
        ready = {id(params[x]): int(seq[i]) for i, x in enumerate(names)}
        order = sorted(range(len(mine)), key=lambda i: (ready[id(mine[i])], i))
        bucket_bytes = self._bucket_bytes
        self.buckets, self.where = [], {}
        cur, cur_bytes, cur_key = [], 0, None
        for i in order:
            p = mine[i]
            key = (_real_view(p).dtype, p.device)
            nbytes = _real_view(p).numel() * _real_view(p).element_size()
            if cur and (key != cur_key or cur_bytes + nbytes > bucket_bytes):
                self._close(cur, cur_key)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
            cur_key = key
        if cur:
            self._close(cur, cur_key)

    def _stamp(self):
        """A point in time on the gradients' device: a recorded CUDA event, or the host clock for CPU tensors."""
        dev = self.buckets[0].flat.device if self.buckets else torch.device("cpu")
        if dev.type == "cuda":
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(dev))
            return ev
        import time
        return time.perf_counter()

    def allreduce_ms(self):
        """Milliseconds from the launch of the first bucket's all-reduce to the completion of the last ``wait()`` of the
        most recent backward -- the gradient exchange as the step sees it (exposed + overlapped).  None before the first."""
        if self._span is None:
            return None
        t0, t1 = self._span
        if isinstance(t0, float):
            return (t1 - t0) * 1e3
        t1.synchronize()
        return t0.elapsed_time(t1)

    def grad_views(self):
        """A fresh view of its bucket slot for every parameter (id(p) -> tensor shaped like p): the engine's backward
        writes the gradients there and returns these very tensors, which autograd then adopts as ``p.grad``."""
        out = {}
        for b in self.buckets:
            for (off, n), p in zip(b.offsets, b.params):
                v = b.flat[off:off + n]
                out[id(p)] = torch.view_as_complex(v.view(*p.shape, 2)) if p.is_complex() else v.view(p.shape)
        return out

    def _on_grad(self, p):
        if not self._callback_queued:
            self._callback_queued = True
            Variable._execution_engine.queue_callback(self._finalize)
        bi, pi = self.where[p]
        b = self.buckets[bi]
        off, n = b.offsets[pi]      # (a gradient accumulated twice in one backward is re-copied below, not recounted)
        g = _real_view(p.grad)
        in_place = (g.data_ptr() == b.flat.data_ptr() + off * b.flat.element_size() and g.is_contiguous()
                    and g.dtype == b.flat.dtype)
        if not in_place:       # the gradient lives elsewhere (torch-autograd backbone, accumulation): copy in, copy back later
            b.flat[off:off + n].copy_(g.reshape(-1))
            b.foreign.add(pi)
        if pi not in b.ready:
            b.ready.add(pi)
            b.pending -= 1
            if b.pending == 0:
                self._launch(b)

    def _finalize(self):
        self.last_stats = {"in_place": sum(len(b.ready) - len(b.foreign) for b in self.buckets),
                           "copied": sum(len(b.foreign) for b in self.buckets)}
        # Buckets with parameters that got no gradient this backward: every rank has the same graph,
        # so every rank reaches this point with the same set; missing slots contribute zeros.
        # A slot that got no gradient may still BE a retained ``p.grad`` (a view of the arena from an earlier backward, kept
        # because the caller did not set the gradients to None): its value is set aside, not wiped, and put back afterwards.
        kept = []
        for b in self.buckets:
            if b.work is None and b.ready:
                for pi, p in enumerate(b.params):
                    if pi not in b.ready:
                        off, n = b.offsets[pi]
                        slot = b.flat[off:off + n]
                        g = None if p.grad is None else _real_view(p.grad)
                        if g is not None and g.data_ptr() == slot.data_ptr():
                            kept.append((slot, slot.clone()))
                        slot.zero_()
                self._launch(b, early_ok=False)      # (the zeroing above runs on the gradients' stream: exchange behind it)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.flat /= self.world
                for pi in b.foreign:       # gradients in their own storage get the average copied back
                    p = b.params[pi]
                    if p.grad is not None:
                        off, n = b.offsets[pi]
                        _real_view(p.grad).copy_(b.flat[off:off + n].view_as(_real_view(p.grad)))
            b.work, b.pending, b.ready, b.foreign = None, len(b.params), set(), set()
        for slot, value in kept:
            slot.copy_(value)
        self.last_stats["kept"] = len(kept)
        end = self._stamp() if (self._t_first is not None or self._bwd_end is not None) else None
        if self._t_first is not None:
            self._span = (self._t_first, end)
        self._exposed = (self._bwd_end, end) if (self._bwd_end is not None and not isinstance(end, float)) else None
        self.last_stats["overlapped_buckets"] = self.overlapped_buckets
        self.overlapped_buckets = 0
        if self._engine is not None and not self._rebucketed and self.buckets and self.buckets[0].flat.is_cuda:
            self._rebucketed = True
            self._rebucket_by_readiness()
        self._engine = self._bwd_end = None
        self._t_first = None
        self._callback_queued = False

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def apply_gradient_allreduce(module, bucket_bytes=DEFAULT_BUCKET_BYTES):
    """``distributed_util.py:97-149``: returns the SAME module, now averaging its gradients over
    the process group during ``backward()``."""
    module._dws_grad_reducer = GradientAllReducer(module, bucket_bytes=bucket_bytes)
    return module

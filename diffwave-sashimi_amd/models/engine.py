"""Shared host-side plumbing of the two model shims: owns the ``dws_model``
handle, mirrors the module's state_dict into it, runs forward / sampling."""
import ctypes

import torch
import torch.nn as nn

from .. import _lib


class _EngineTrainFn(torch.autograd.Function):
    """Differentiable call of the engine (`train.py:221`): forward_train keeps the activations inside
    the dws_model, backward fills the gradient of every raw state-dict tensor, which are handed to
    autograd as the gradients of the module's parameters (so optimizers and the DP all-reduce of
    ``distributed_util.apply_gradient_allreduce`` work unchanged)."""

    @staticmethod
    def forward(ctx, module, audio, steps, mel_spec, *params):
        lib = _lib.load()
        B, _, L = audio.shape
        module._sync_params(L)
        module._prepare(B, L)
        module._set_condition(mel_spec)      # conditional models: the conditioner's parameters get gradients too
        x = audio.detach().to(torch.float32).contiguous()
        out = torch.empty((B, module.out_channels, L), device=audio.device, dtype=torch.float32)
        _lib.check(lib.dws_model_forward_train(module._handle, x.data_ptr(), steps.data_ptr(), out.data_ptr(),
                                               _lib.current_stream()))
        ctx.module, ctx.x, ctx.steps = module, x, steps      # x must stay alive until backward (init_conv adjoint)
        module._train_generation += 1                        # the activations inside the dws_model belong to THIS forward
        ctx.generation = module._train_generation
        ctx.meta = [(n, tuple(p.shape), p.dtype) for n, p in module.named_parameters()]
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        m = ctx.module
        if ctx.generation != m._train_generation:
            raise RuntimeError("the engine keeps the activations of ONE training forward: another forward ran on this "
                               "module before this backward (accumulate gradients with forward/backward pairs)")
        d = dout.detach().to(torch.float32).contiguous()
        n = len(ctx.meta)
        # Data-parallel runs (distributed_util.GradientAllReducer): the gradients are written straight into the flat
        # all-reduce buckets -- views handed out per backward, adopted by autograd as p.grad without a copy.  A parameter
        # that still holds a gradient (accumulation over several backwards) gets its own tensor instead: its p.grad may
        # BE last step's view of the same slot.
        reducer = getattr(m, "_dws_grad_reducer", None)
        views = reducer.grad_views() if reducer is not None else {}
        outs, groups = [], []
        for (_, shape, dtype), p in zip(ctx.meta, m.parameters()):
            v = views.get(id(p)) if (p.grad is None and dtype == torch.float32) else None
            if v is not None and v.device == d.device and v.dtype == torch.float32 and tuple(v.shape) == shape:
                outs.append(v)
                groups.append(reducer.where[p][0])          # the bucket the slot belongs to
            elif reducer is not None:
                # outside the arena (accumulation into an existing p.grad, a non-fp32 parameter): a PERSISTENT staging tensor,
                # so that the installed sinks keep their addresses from step to step (autograd adds it into p.grad / the
                # dtype conversion below copies it: nothing keeps a reference to it)
                fb = m.__dict__.setdefault("_grad_fallback", {})
                t = fb.get(ctx.meta[len(outs)][0])
                if t is None or tuple(t.shape) != shape or t.device != d.device:
                    t = fb[ctx.meta[len(outs)][0]] = torch.empty(shape, device=d.device, dtype=torch.float32)
                outs.append(t)
                groups.append(-1)
            else:
                outs.append(torch.empty(shape, device=d.device, dtype=torch.float32))
                groups.append(-1)
        del views
        groups_in = list(groups)        # >= 0: an arena view (handed to autograd as it is)
        if reducer is not None:
            # STAGED hand-over: the engine's backward delivers every gradient to its destination itself, bucket by bucket, the
            # moment a bucket's last gradient exists, and records an event per bucket; the reducer then launches that bucket's
            # all-reduce on a side stream that waits for the event only -- the exchange overlaps the rest of backward.
            nb = len(reducer.buckets)
            groups = [g if g >= 0 else nb for g in groups]          # gradients outside the arena: one last group
            key = (tuple(g.data_ptr() for g in outs), tuple(groups))
            if getattr(m, "_sink_key", None) != key:
                names = (ctypes.c_char_p * n)(*[name.encode() for name, _, _ in ctx.meta])
                dsts = (ctypes.c_void_p * n)(*[g.data_ptr() for g in outs])
                numels = (ctypes.c_int64 * n)(*[g.numel() for g in outs])
                _lib.check(lib.dws_model_set_grad_sinks(m._handle, n, names, dsts, numels, (ctypes.c_int32 * n)(*groups), nb + 1))
                m._sink_key = key
            # (no reference to `outs` may be kept: autograd adopts a returned gradient as p.grad only when nobody else holds
            # it, otherwise it clones.  The destinations outlive the sinks anyway: arena slots and `_grad_fallback` tensors.)
            _lib.check(lib.dws_model_backward(m._handle, d.data_ptr(), _lib.current_stream()))
            reducer.engine_backward_done(m, [name for name, _, _ in ctx.meta])
        else:
            if getattr(m, "_sink_key", None) is not None:           # the module left data-parallel mode
                _lib.check(lib.dws_model_set_grad_sinks(m._handle, 0, None, None, None, None, 0))
                m._sink_key = None
            _lib.check(lib.dws_model_backward(m._handle, d.data_ptr(), _lib.current_stream()))
            names = (ctypes.c_char_p * n)(*[name.encode() for name, _, _ in ctx.meta])
            dsts = (ctypes.c_void_p * n)(*[g.data_ptr() for g in outs])
            numels = (ctypes.c_int64 * n)(*[g.numel() for g in outs])
            _lib.check(lib.dws_model_get_grads(m._handle, n, names, dsts, numels, _lib.current_stream()))   # one launch
        grads = [(g.to(dtype) if (dtype != torch.float32 or gi >= 0 or reducer is None) else g.clone())
                 for g, gi, (_, _, dtype) in zip(outs, groups_in, ctx.meta)]
        return (None, None, None, None, *grads)


class EngineModule(nn.Module):
    """Base of :class:`WaveNet` and :class:`Sashimi`.

    Subclasses register parameters in the reference's state_dict layout and
    implement ``_desc()``.  ``forward`` keeps the reference contract
    ``net((audio[B,in,L], diffusion_steps[B,1]), mel_spec=None)``
    (``models/wavenet.py:202-210``, ``models/sashimi.py:277-313``)."""

    def __init__(self):
        super().__init__()
        self._handle = None
        self._versions = {}
        self._shape = None
        self._mel_key = None
        self._mel_ref = None
        self._train_generation = 0

    # -- handle management ---------------------------------------------------
    def _desc(self):
        raise NotImplementedError

    def _ensure_handle(self):
        if self._handle is None:
            lib = _lib.load()
            h = ctypes.c_void_p()
            desc = self._desc()
            _lib.check(lib.dws_model_create(ctypes.byref(desc), ctypes.byref(h)))
            self._handle = h
            self._versions = {}
        return self._handle

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None and _lib._lib is not None:
                _lib._lib.dws_model_destroy(self._handle)
        except Exception:
            pass

    def set_option(self, key, value):
        """Engine option, e.g. ``set_option("precision", "bf16x3")`` (see dws_model_set_option)."""
        _lib.check(_lib.load().dws_model_set_option(self._ensure_handle(), key.encode(), str(value).encode()))
        # an option marks the engine dirty: its next commit re-packs the weights and DROPS the installed conditioner
        # terms, so the cached mel must be handed over again even if it is the very same tensor object
        self._mel_key = None
        self._mel_ref = None
        return self

    def invalidate(self):
        """Forget what the engine holds: the next call re-sends every parameter and re-runs the conditioner.
        Needed after writes the version counters cannot see (``p.data.copy_``, in-place kernels on ``.data``)."""
        # keep the shapes: same-shape float32 GPU tensors then go through the one-launch batched refresh
        self._versions = {k: (None, -1) + tuple(v[2:]) for k, v in self._versions.items()}
        self._mel_key = None
        self._mel_ref = None

    def _engine_state(self):
        """(name, tensor) pairs handed to dws_model_set_param: the state_dict."""
        return self.state_dict(keep_vars=True).items()

    def _sync_params(self, run_length=None):
        """Hand every changed state_dict tensor to the engine (raw: weight_g /
        weight_v / bias ...; folding and packing happen in dws_model_commit).
        ``run_length`` is the length of the coming call (SaShiMi adapts its S4 kernels to it first)."""
        self._run_L = run_length
        lib = _lib.load()
        h = self._ensure_handle()
        stream = _lib.current_stream()
        batch = []   # (name, tensor) already known to the engine with this shape, float32 on the GPU: one launch for all
        for name, t in self._engine_state():
            key = (t.data_ptr(), t._version, tuple(t.shape), str(t.device))
            old = self._versions.get(name)
            if old == key:
                continue
            if (old is not None and old[2] == key[2] and t.dtype == torch.float32 and t.device.type == "cuda"
                    and t.is_contiguous()):
                batch.append((name, t))
            else:
                if t.dtype == torch.int64:
                    src, dtype = t.detach().contiguous(), 1
                else:
                    src, dtype = t.detach().to(torch.float32).contiguous(), 0
                shape = (ctypes.c_int64 * max(src.dim(), 1))(*src.shape)
                _lib.check(lib.dws_model_set_param(h, name.encode(), src.data_ptr(), shape, src.dim(), dtype, stream))
                if src.device.type != "cuda":
                    torch.cuda.synchronize()
            self._versions[name] = key
            self._mel_key = None  # conditioner terms depend on the weights
        if batch:
            n = len(batch)
            names = (ctypes.c_char_p * n)(*[k.encode() for k, _ in batch])
            srcs = (ctypes.c_void_p * n)(*[t.data_ptr() for _, t in batch])
            _lib.check(lib.dws_model_update_params(h, n, names, srcs, stream))

    def _prepare(self, B, L):
        if self._shape != (B, L):
            _lib.check(_lib.load().dws_model_prepare(self._ensure_handle(), B, L))
            self._shape = (B, L)
            self._mel_key = None

    def _set_condition(self, mel_spec):
        lib = _lib.load()
        h = self._ensure_handle()
        if mel_spec is None:
            if self._mel_key is not None:
                _lib.check(lib.dws_model_set_condition(h, 0, 0, 0, _lib.current_stream()))
                self._mel_key = self._mel_ref = None
            return
        # Same tensor OBJECT with an unchanged version counter: the cached conditioner terms still hold.  The
        # strong reference keeps that tensor alive, so a new mel can never reuse its address and match by accident.
        key = (mel_spec.data_ptr(), mel_spec._version, tuple(mel_spec.shape))
        if mel_spec is self._mel_ref and key == self._mel_key:
            return
        mel = mel_spec.detach().to(torch.float32).contiguous()
        if mel.dim() != 3:
            raise RuntimeError("mel_spec must be [B or 1, mel_bands, Tmel]")
        _lib.check(lib.dws_model_set_condition(h, mel.data_ptr(), mel.shape[0], mel.shape[2], _lib.current_stream()))
        self._mel_key, self._mel_ref = key, mel_spec

    # -- reference surface -----------------------------------------------------
    def forward(self, input_data, mel_spec=None):
        audio, diffusion_steps = input_data
        if audio.device.type != "cuda":
            raise RuntimeError("libdws runs on the GPU only: move the model and inputs to cuda "
                               "(there is no CPU fallback)")
        if audio.dim() != 3:
            raise RuntimeError("audio must be [B, in_channels, L]")
        B, Cin, L = audio.shape
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
            # training call (`train.py:221`): differentiable w.r.t. the parameters
            steps = diffusion_steps.detach().to(device=audio.device, dtype=torch.float32).reshape(-1).contiguous()
            if steps.numel() != B:
                raise RuntimeError(f"diffusion_steps must hold B={B} entries, got {tuple(diffusion_steps.shape)}")
            return _EngineTrainFn.apply(self, audio, steps, mel_spec, *self.parameters())
        with torch.no_grad():
            self._train_generation += 1      # an eval forward overwrites the activations of a pending training forward
            self._sync_params(L)
            self._prepare(B, L)
            self._set_condition(mel_spec)
            x = audio.detach().to(torch.float32).contiguous()
            steps = diffusion_steps.detach().to(device=audio.device, dtype=torch.float32).reshape(-1).contiguous()
            if steps.numel() != B:
                raise RuntimeError(f"diffusion_steps must hold B={B} entries, got {tuple(diffusion_steps.shape)}")
            out = torch.empty((B, self.out_channels, L), device=audio.device, dtype=torch.float32)
            _lib.check(_lib.load().dws_model_forward(self._handle, x.data_ptr(), steps.data_ptr(), out.data_ptr(),
                                                     _lib.current_stream()))
        return out

    def read_tap(self, name, shape):
        """Parity/debug tap of an internal activation (see dws_model_read_tap)."""
        out = torch.empty(shape, device="cuda", dtype=torch.float32)
        _lib.check(_lib.load().dws_model_read_tap(self._handle, name.encode(), out.data_ptr(), out.numel(),
                                                  _lib.current_stream()))
        return out

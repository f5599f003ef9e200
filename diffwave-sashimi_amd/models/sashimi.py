"""Drop-in for ``models/sashimi.py`` of the reference: same constructor kwargs,
``forward((audio, diffusion_steps), mel_spec=None)`` and state_dict keys
(``{d,c,u}_layers.N.{fc_t,norm1.{m,s},norm2.{m,s},ff.ff.{0,2}.conv,layer.D,
layer.output_linear.0,layer.kernel.kernel.{C,log_dt,B,P,inv_w_real,w_imag,L}}``,
pools ``linear.conv``, ``norm.{m,s}``, ``init_conv``/``final_conv``, ``fc_t1/2``).
The module only holds parameters; the forward runs in libdws.so."""
import torch
import torch.nn as nn

from .. import _lib
from . import s4_init
from .engine import EngineModule
from .utils import ConvParams, LinearParams, ZeroConvParams, _uniform_, upsampler_params


class _LNParams(nn.Module):
    """``TransposedLN`` parameters (``models/sashimi.py:11-16``): scalars m=0, s=1."""

    def __init__(self):
        super().__init__()
        self.m = nn.Parameter(torch.zeros(1))
        self.s = nn.Parameter(torch.ones(1))


class _NPLRKernelParams(nn.Module):
    """``SSKernelNPLR`` parameters + the ``L`` buffer (``models/s4.py:631-641``)."""

    def __init__(self, H, N):
        super().__init__()
        p = s4_init.init_s4_params(H, N)
        self.C = nn.Parameter(p["C"])
        self.log_dt = nn.Parameter(p["log_dt"])
        self.B = nn.Parameter(p["B"])
        self.P = nn.Parameter(p["P"])
        self.inv_w_real = nn.Parameter(p["inv_w_real"])
        self.w_imag = nn.Parameter(p["w_imag"])
        self.register_buffer("L", torch.tensor(0))
        # the per-parameter optimizer hints of `OptimModule.register` (`models/s4.py:508-518`, called at `:634-638` with lr=None
        # since `sashimi.py:126` builds `S4(d_model, l_max=L, bidirectional=True)`): weight decay 0, no own learning rate.  C is a
        # plain parameter there (`:631`).  The reference's `train.py:91` ignores them; `train.py`'s `optim_param_groups` reads them.
        for name in ("log_dt", "B", "P", "inv_w_real", "w_imag"):
            getattr(self, name)._optim = {"weight_decay": 0.0}


class _S4Params(nn.Module):
    """``S4(d_model, l_max=L, bidirectional=True)`` parameters (``models/s4.py:1279-1373``)."""

    def __init__(self, H, l_max, N=64):
        super().__init__()
        self.l_max = l_max
        self.D = nn.Parameter(torch.randn(1, H))
        self.kernel = nn.Module()
        self.kernel.kernel = _NPLRKernelParams(H, N)
        conv = nn.Module()
        conv.weight = nn.Parameter(_uniform_(torch.empty(2 * H, H, 1), H))
        conv.bias = nn.Parameter(_uniform_(torch.empty(2 * H), H))
        self.output_linear = nn.ModuleList([conv])


class _FFParams(nn.Module):
    def __init__(self, H, expand):
        super().__init__()
        self.ff = nn.ModuleList([ConvParams(H, expand * H, 1), nn.Identity(), ConvParams(expand * H, H, 1)])


class _BlockParams(nn.Module):
    """``DiffWaveBlock`` parameters (``models/sashimi.py:113-141``)."""
    kind = "block"

    def __init__(self, H, L, ff, embed_out, unconditional, mel_upsample):
        super().__init__()
        self.H, self.L_stage = H, L
        self.fc_t = LinearParams(embed_out, H)
        self.layer = _S4Params(H, L)
        self.ff = _FFParams(H, ff)
        self.norm1 = _LNParams()
        self.norm2 = _LNParams()
        if not unconditional:
            self.upsample_conv2d = upsampler_params(mel_upsample)
            self.mel_conv = ConvParams(80, H, 1)


class _PoolParams(nn.Module):
    """``DownPool`` / ``UpPool`` parameters (``models/sashimi.py:23-58``)."""

    def __init__(self, kind, d_in, d_out):
        super().__init__()
        self.kind = kind
        self.linear = ConvParams(d_in, d_out, 1)


class Sashimi(EngineModule):
    def __init__(self, in_channels=1, out_channels=1,
                 d_model=64, n_layers=8, pool=[4, 4], expand=2, ff=2, unet=True,
                 diffusion_step_embed_dim_in=128,
                 diffusion_step_embed_dim_mid=512,
                 diffusion_step_embed_dim_out=512,
                 unconditional=False,
                 mel_upsample=[16, 16],
                 L=16000,
                 **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.L, self.unet, self.d_model, self.n_layers = L, unet, d_model, n_layers
        self.expand, self.ff, self.pool = expand, ff, list(pool)
        self.unconditional = unconditional
        self.mel_upsample = list(mel_upsample)
        self.embed_dims = (diffusion_step_embed_dim_in, diffusion_step_embed_dim_mid, diffusion_step_embed_dim_out)
        # the reference's `_residual` helper does not forward diffusion_step_embed_dim_out, so every
        # DiffWaveBlock.fc_t takes the default 512 inputs (`sashimi.py:224-229,117`); any other value
        # makes the reference's own forward fail, so it is rejected here up front
        if diffusion_step_embed_dim_out != 512:
            raise ValueError("Sashimi: diffusion_step_embed_dim_out must be 512 (DiffWaveBlock.fc_t is hard-wired "
                             "to 512 inputs in the reference, sashimi.py:117,224-229)")
        eo = 512

        self.init_conv = nn.ModuleList([ConvParams(in_channels, d_model, 1)])
        self.fc_t1 = LinearParams(diffusion_step_embed_dim_in, diffusion_step_embed_dim_mid)
        self.fc_t2 = LinearParams(diffusion_step_embed_dim_mid, eo)

        def block(H, Ls):
            return _BlockParams(H, Ls, ff, eo, unconditional, self.mel_upsample)

        H, Ls = d_model, L
        d_layers = []
        for p in self.pool:
            if unet:
                d_layers += [block(H, Ls) for _ in range(n_layers)]
            d_layers.append(_PoolParams("down", H * p, H * expand))
            Ls //= p
            H *= expand
        self.d_layers = nn.ModuleList(d_layers)
        self.c_layers = nn.ModuleList([block(H, Ls) for _ in range(n_layers)])
        u_layers = []
        for p in self.pool[::-1]:
            H //= expand
            Ls *= p
            u_layers.append(_PoolParams("up", H * expand, H * p))
            u_layers += [block(H, Ls) for _ in range(n_layers)]
        self.u_layers = nn.ModuleList(u_layers)
        self.norm = _LNParams()
        self.final_conv = nn.ModuleList([ConvParams(d_model, d_model, 1), nn.Identity(),
                                         ZeroConvParams(d_model, out_channels)])
        self._nodes = {}
        self._L_host = {}     # id(kernel) -> ((data_ptr, version) of its L buffer, value): see _kernel_L

    def _blocks(self):
        for ml in (self.d_layers, self.c_layers, self.u_layers):
            for m in ml:
                if isinstance(m, _BlockParams):
                    yield m

    def _desc(self):
        d = _lib.ModelDesc()
        d.kind = _lib.DWS_KIND_SASHIMI
        d.in_channels, d.out_channels = self.in_channels, self.out_channels
        (d.diffusion_step_embed_dim_in, d.diffusion_step_embed_dim_mid,
         d.diffusion_step_embed_dim_out) = self.embed_dims
        d.unconditional = 1 if self.unconditional else 0
        d.mel_upsample[0], d.mel_upsample[1] = self.mel_upsample
        d.mel_bands = 80
        d.d_model, d.n_layers, d.n_pool = self.d_model, self.n_layers, len(self.pool)
        if len(self.pool) > _lib.DWS_MAX_POOL:
            raise NotImplementedError(f"more than {_lib.DWS_MAX_POOL} pooling stages")
        for i, p in enumerate(self.pool):
            d.pool[i] = p
        d.expand, d.ff, d.unet, d.L = self.expand, self.ff, 1 if self.unet else 0, self.L
        return d

    def invalidate(self):
        """Also forget the host copies of the kernels' ``L`` buffers: a ``.data`` write (``broadcast_state``) changes
        them without touching the version counter."""
        self._L_host.clear()
        super().invalidate()

    def _kernel_L(self, k):
        """Host copy of a kernel's ``L`` buffer, re-read only when the buffer changed (in place, or replaced by
        ``.to()`` / ``load_state_dict``).  ``int(k.L)`` on a GPU buffer is a device-to-host copy that waits for everything
        queued before it; twice per block on every call it kept the host from ever running ahead of the GPU."""
        key = (k.L.data_ptr(), k.L._version, str(k.L.device))
        hit = self._L_host.get(id(k))
        if hit is None or hit[0] != key:
            hit = self._L_host[id(k)] = (key, int(k.L))
        return hit[1]

    @torch.no_grad()
    def _setup_C(self):
        """First-forward mutation of the reference (``s4.py:531-551,686-687``): a kernel
        whose ``L`` buffer is 0 gets ``C <- C (I - dA^l_max)`` in place and ``L <- l_max``,
        so a state_dict saved afterwards stores C~ exactly like a reference checkpoint.
        Other input lengths never touch the parameters: ``S4.forward`` asks the kernel for
        ``min(L_in, l_max)`` taps (``s4.py:1387``), so the length-doubling branch is unreachable."""
        for blk in self._blocks():
            k = blk.layer.kernel.kernel
            if self._kernel_L(k) == 0:
                Ct = s4_init.setup_C(k.C, k.P, k.inv_w_real, k.w_imag, k.log_dt, blk.L_stage)
                k.C.copy_(Ct.to(k.C.device))
                k.L.fill_(blk.L_stage)

    def _engine_state(self):
        L_in = getattr(self, "_run_L", None) or self.L
        span = 1
        for p in self.pool:
            span *= p
        if L_in % span != 0:
            raise RuntimeError(f"sashimi: input length {L_in} is not divisible by the pooling factors {self.pool}")
        self._setup_C()
        items = list(self.state_dict(keep_vars=True).items())
        for Lk in sorted({self._kernel_L(blk.layer.kernel.kernel) for blk in self._blocks()}):
            if Lk not in self._nodes:
                omega, z = s4_init.omega_z(Lk)
                self._nodes[Lk] = (torch.view_as_real(omega).contiguous(), torch.view_as_real(z).contiguous())
            items.append((f"__omega.{Lk}", self._nodes[Lk][0]))
            items.append((f"__z.{Lk}", self._nodes[Lk][1]))
        return items

    @classmethod
    def name(cls, cfg):
        return "{}_d{}_n{}_pool_{}_expand{}_ff{}".format(
            "unet" if cfg["unet"] else "snet", cfg["d_model"], cfg["n_layers"], len(cfg["pool"]),
            cfg["expand"], cfg["ff"])

    def __repr__(self):
        # the reference's __repr__ raises (`sashimi.py:316`: ''.join of ints, missing attribute)
        return (f"sashimi_h{self.d_model}_d{self.n_layers}_pool{''.join(map(str, self.pool))}_expand{self.expand}"
                f"_ff{self.ff}_{'uncond' if self.unconditional else 'cond'}")

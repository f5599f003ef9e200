"""GPU: the Cauchy operator against the reference's known-answer method
(`extensions/cauchy/test_cauchy.py:53-95`): fp64 formula as truth, error budget
relative to a second fp32 implementation (there pykeops, here the fp32 torch
evaluation of the same sum)."""
import pytest
import torch

from oracle import cauchy as oc
from tests.conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

ATOL = 1e-4
TOL_FACTOR = 10.0


def _relerr(x, ref):
    return (x.cdouble().cpu() - ref).abs() / ref.abs()


@pytest.mark.parametrize("L", [3, 17, 489, 2 ** 10, 1047, 2 ** 11, 2 ** 12, 2 ** 13, 2 ** 14, 2 ** 18])
@pytest.mark.parametrize("N", [4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048])
def test_cauchy_mult_symmetric(gpu, N, L):
    from diffwave_sashimi_amd.extensions.cauchy import cauchy_mult
    if N * L > 2 ** 27:
        pytest.skip("oracle too slow on the host for this size")
    batch_size = 4
    v_half, z, w_half = oc.generate_data(batch_size, N, L, symmetric=True, seed=2357)
    ref = oc.cauchy_sym_direct(v_half.cdouble(), z.cdouble(), w_half.cdouble())
    alt = oc.cauchy_sym_direct(v_half, z, w_half)  # fp32 comparison implementation
    vg = v_half.to(gpu).requires_grad_(True)
    wg = w_half.to(gpu).requires_grad_(True)
    out = cauchy_mult(vg, z.to(gpu), wg, symmetric=True)
    e, ea = _relerr(out.detach(), ref), _relerr(alt, ref)
    assert e.amax() <= ea.amax() * TOL_FACTOR + ATOL
    assert e.mean() <= ea.mean() * TOL_FACTOR + ATOL
    g = torch.Generator().manual_seed(5)
    dout = torch.randn(out.shape, dtype=torch.complex64, generator=g)
    dv_ref, dw_ref = oc.cauchy_sym_bwd(v_half.cdouble(), z.cdouble(), w_half.cdouble(), dout.cdouble())
    dv_alt, dw_alt = oc.cauchy_sym_bwd(v_half, z, w_half, dout)
    dv, dw = torch.autograd.grad(out, (vg, wg), dout.to(gpu))
    for got, alt_, ref_ in ((dv, dv_alt, dv_ref), (dw, dw_alt, dw_ref)):
        e, ea = _relerr(got, ref_), _relerr(alt_, ref_)
        assert e.amax() <= ea.amax() * TOL_FACTOR + ATOL
        assert e.mean() <= ea.mean() * TOL_FACTOR + ATOL


@pytest.mark.parametrize("N,L", [(4, 3), (16, 17), (64, 489), (64, 1024)])
def test_cauchy_golden_vectors(gpu, N, L):
    """Committed fp64 answers from the reference formula + autograd."""
    from diffwave_sashimi_amd.extensions import cauchy as ext
    g = load_golden("cauchy")
    t = f"sym/N{N}_L{L}"
    v, z, w, dout = (torch.from_numpy(g[f"{t}/{k}"]).to(gpu) for k in ("v_half", "z", "w_half", "dout"))
    out = ext.cauchy_mult_sym_fwd(v, z, w)
    dv, dw = ext.cauchy_mult_sym_bwd(v, z, w, dout)
    assert rel_err(torch.view_as_real(out), torch.view_as_real(torch.from_numpy(g[f"{t}/out"]))) < 1e-4
    assert rel_err(torch.view_as_real(dv), torch.view_as_real(torch.from_numpy(g[f"{t}/dv"]))) < 1e-4
    assert rel_err(torch.view_as_real(dw), torch.view_as_real(torch.from_numpy(g[f"{t}/dw"]))) < 1e-4
    t = f"nonsym/N{N}_L{L}"
    v, z, w, dout = (torch.from_numpy(g[f"{t}/{k}"]).to(gpu) for k in ("v", "z", "w", "dout"))
    out = ext.cauchy_mult(v, z, w, symmetric=False)
    dv, dw = ext.cauchy_mult_bwd(v, z, w, dout)
    assert rel_err(torch.view_as_real(out), torch.view_as_real(torch.from_numpy(g[f"{t}/out"]))) < 1e-4
    assert rel_err(torch.view_as_real(dv), torch.view_as_real(torch.from_numpy(g[f"{t}/dv"]))) < 1e-4
    assert rel_err(torch.view_as_real(dw), torch.view_as_real(torch.from_numpy(g[f"{t}/dw"]))) < 1e-4


@pytest.mark.parametrize("B,NH,L", [(11, 3, 37), (9, 5, 1500), (1, 1, 5000), (17, 13, 4097), (8, 9, 4096), (3, 7, 1024)])
def test_cauchy_ragged_shapes(gpu, B, NH, L):
    """Shapes off the kernels' granules: row counts that are not multiples of 8 (the backward pads its XCD-local block order),
    state sizes that are not multiples of the 8 n a backward thread owns (or of the packed pairs), row lengths on both
    sides of the 1024 / 4096-bin block-shape switches and of the forward's outputs-per-thread choice."""
    from diffwave_sashimi_amd.extensions import cauchy as ext
    g = torch.Generator().manual_seed(100 * B + NH)
    v = torch.randn(B, NH, dtype=torch.complex64, generator=g)
    w = torch.randn(B, NH, dtype=torch.complex64, generator=g)
    w = torch.complex(-w.real.abs() - 0.05, w.imag)
    z = torch.exp(1j * torch.randn(L, dtype=torch.float32, generator=g))
    dout = torch.randn(B, L, dtype=torch.complex64, generator=g)
    vg, zg, wg, dg = v.to(gpu), z.to(gpu), w.to(gpu), dout.to(gpu)
    ref = oc.cauchy_sym_direct(v.cdouble(), z.cdouble(), w.cdouble())
    dv_ref, dw_ref = oc.cauchy_sym_bwd(v.cdouble(), z.cdouble(), w.cdouble(), dout.cdouble())
    out = ext.cauchy_mult_sym_fwd(vg, zg, wg)
    dv, dw = ext.cauchy_mult_sym_bwd(vg, zg, wg, dg)
    for got, want in ((out, ref), (dv, dv_ref), (dw, dw_ref)):
        assert got.shape == want.shape
        assert rel_err(torch.view_as_real(got), torch.view_as_real(want.to(torch.complex64))) < 1e-4
    ref = oc.cauchy_direct(v.cdouble(), z.cdouble(), w.cdouble())
    dv_ref, dw_ref = oc.cauchy_bwd(v.cdouble(), z.cdouble(), w.cdouble(), dout.cdouble())
    out = ext.cauchy_mult(vg, zg, wg, symmetric=False)
    dv, dw = ext.cauchy_mult_bwd(vg, zg, wg, dg)
    for got, want in ((out, ref), (dv, dv_ref), (dw, dw_ref)):
        assert rel_err(torch.view_as_real(got), torch.view_as_real(want.to(torch.complex64))) < 1e-4


def test_cauchy_broadcast_front_end_and_errors(gpu):
    from diffwave_sashimi_amd.extensions import cauchy as ext
    g = torch.Generator().manual_seed(1)
    # the model's call shape: v (2,3,H,N), w (H,N) broadcast, z (L/2+1)  (`s4.py:752-758`)
    v = torch.randn(2, 3, 5, 32, dtype=torch.complex64, generator=g)
    w = torch.randn(5, 32, dtype=torch.complex64, generator=g) - 2.0
    z = torch.exp(1j * torch.randn(33, generator=g))
    out = ext.cauchy_mult(v.to(gpu), z.to(gpu), w.to(gpu), symmetric=True)
    assert out.shape == (2, 3, 5, 33)
    wb = w.expand(2, 3, 5, 32).reshape(-1, 32)
    ref = oc.cauchy_sym_direct(v.reshape(-1, 32).cdouble(), z.cdouble(), wb.cdouble()).reshape(2, 3, 5, 33)
    assert rel_err(torch.view_as_real(out), torch.view_as_real(ref)) < 1e-4
    with pytest.raises(RuntimeError):
        ext.cauchy_mult_sym_fwd(v.reshape(-1, 32), z.to(gpu), w.to(gpu))          # v on CPU
    with pytest.raises(RuntimeError):
        ext.cauchy_mult_sym_fwd(v.reshape(-1, 32).to(gpu), z.to(gpu), w.to(gpu))  # shape mismatch
    with pytest.raises(NotImplementedError):
        big = torch.zeros(1, 2048, dtype=torch.complex64, device=gpu)
        ext.cauchy_mult_sym_fwd(big, z.to(gpu), big)
    # empty inputs
    e = ext.cauchy_mult_sym_fwd(torch.zeros(0, 4, dtype=torch.complex64, device=gpu), z.to(gpu),
                                torch.zeros(0, 4, dtype=torch.complex64, device=gpu))
    assert e.shape == (0, 33)


def _import_cauchy_mult_by_name():
    """The way the reference gets it (`extensions/cauchy/cauchy.py:5`, `models/s4.py:35-42`): a top-level module
    called `cauchy_mult` on sys.path -- here the standalone ctypes binding diffwave-sashimi_amd/extensions/cauchy_mult.py."""
    import importlib
    import os
    import sys
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffwave-sashimi_amd", "extensions")
    if d not in sys.path:
        sys.path.insert(0, d)
    return importlib.import_module("cauchy_mult")


@pytest.mark.parametrize("N,L", [(4, 3), (16, 17), (64, 489), (64, 1024)])
def test_cauchy_mult_module_by_its_reference_name(gpu, N, L):
    """INTEGRATION.md section 1 as a test: `from cauchy_mult import cauchy_mult_fwd, cauchy_mult_bwd,
    cauchy_mult_sym_fwd, cauchy_mult_sym_bwd` (the reference's import line) and the four entry points against the
    committed fp64 answers (tests/golden/cauchy.npz, generated from the reference's formula + autograd)."""
    cm = _import_cauchy_mult_by_name()
    from cauchy_mult import cauchy_mult_bwd, cauchy_mult_fwd, cauchy_mult_sym_bwd, cauchy_mult_sym_fwd  # noqa: F401
    assert cm.__name__ == "cauchy_mult" and "diffwave_sashimi_amd" not in cm.__name__
    g = load_golden("cauchy")
    t = f"sym/N{N}_L{L}"
    v, z, w, dout = (torch.from_numpy(g[f"{t}/{k}"]).to(gpu) for k in ("v_half", "z", "w_half", "dout"))
    out = cauchy_mult_sym_fwd(v, z, w)
    dv, dw = cauchy_mult_sym_bwd(v, z, w, dout)
    assert out.shape == (v.shape[0], L) and out.dtype == torch.complex64 and out.device == v.device
    for got, key in ((out, "out"), (dv, "dv"), (dw, "dw")):
        assert rel_err(torch.view_as_real(got), torch.view_as_real(torch.from_numpy(g[f"{t}/{key}"]))) < 1e-4, key
    t = f"nonsym/N{N}_L{L}"
    v, z, w, dout = (torch.from_numpy(g[f"{t}/{k}"]).to(gpu) for k in ("v", "z", "w", "dout"))
    out = cauchy_mult_fwd(v, z, w)
    dv, dw = cauchy_mult_bwd(v, z, w, dout)
    for got, key in ((out, "out"), (dv, "dv"), (dw, "dw")):
        assert rel_err(torch.view_as_real(got), torch.view_as_real(torch.from_numpy(g[f"{t}/{key}"]))) < 1e-4, key
    # the reference's error behaviour at this boundary: TORCH_CHECK -> RuntimeError, unsupported N -> NotImplementedError
    with pytest.raises(RuntimeError):
        cauchy_mult_sym_fwd(v.cpu(), z, w)
    with pytest.raises(RuntimeError):
        cauchy_mult_sym_fwd(v, z, w[:, :-1])
    with pytest.raises(NotImplementedError):
        big = torch.zeros(1, 2048, dtype=torch.complex64, device=gpu)
        cauchy_mult_sym_fwd(big, z, big)


def test_reference_style_autograd_wrapper_over_cauchy_mult(gpu):
    """What `extensions/cauchy/cauchy.py:80-111` builds on top of the module (an autograd.Function calling
    cauchy_mult_sym_fwd / _bwd): gradients through it agree with the package's own wrapper."""
    cm = _import_cauchy_mult_by_name()
    from diffwave_sashimi_amd.extensions import cauchy as ext

    class Sym(torch.autograd.Function):
        @staticmethod
        def forward(ctx, v, z, w):
            ctx.save_for_backward(v, z, w)
            return cm.cauchy_mult_sym_fwd(v, z, w)

        @staticmethod
        def backward(ctx, dout):
            v, z, w = ctx.saved_tensors
            dv, dw = cm.cauchy_mult_sym_bwd(v, z, w, dout)
            return dv, None, dw

    v_half, z, w_half = oc.generate_data(4, 64, 1000, symmetric=True, seed=11)
    v1, w1 = v_half.to(gpu).requires_grad_(True), w_half.to(gpu).requires_grad_(True)
    v2, w2 = v_half.to(gpu).requires_grad_(True), w_half.to(gpu).requires_grad_(True)
    o1 = Sym.apply(v1, z.to(gpu), w1)
    o2 = ext.cauchy_mult(v2, z.to(gpu), w2, symmetric=True)
    assert torch.equal(o1, o2)
    dout = torch.randn(o1.shape, dtype=torch.complex64, generator=torch.Generator().manual_seed(2)).to(gpu)
    g1 = torch.autograd.grad(o1, (v1, w1), dout)
    g2 = torch.autograd.grad(o2, (v2, w2), dout)
    assert torch.equal(g1[0], g2[0]) and torch.equal(g1[1], g2[1])

"""CPU: pin the Cauchy oracle (torch restatement + plain-C restatement) against
the reference's own fp64 formula and autograd (tests/golden/cauchy.npz)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import cauchy as oc
from tests.conftest import ROOT, load_golden, rel_err

CASES = [(N, L) for N in (4, 16, 64) for L in (3, 17, 489, 1024)]


@pytest.mark.parametrize("N,L", CASES)
def test_sym_oracle_matches_reference_formula(N, L):
    g = load_golden("cauchy")
    t = f"sym/N{N}_L{L}"
    v, z, w, dout = (torch.from_numpy(g[f"{t}/{k}"]).cdouble() for k in ("v_half", "z", "w_half", "dout"))
    # the seeded draw is the reference test's distribution
    v0, z0, w0 = oc.generate_data(4, N, L, True, 2357)
    assert torch.equal(v0.cdouble(), v) and torch.equal(z0.cdouble(), z) and torch.equal(w0.cdouble(), w)
    assert rel_err(torch.view_as_real(oc.cauchy_sym_direct(v, z, w)), torch.view_as_real(torch.from_numpy(g[f"{t}/out"]))) < 1e-11
    assert rel_err(torch.view_as_real(oc.cauchy_sym_formula(v, z, w)), torch.view_as_real(torch.from_numpy(g[f"{t}/out"]))) < 1e-11
    dv, dw = oc.cauchy_sym_bwd(v, z, w, dout)
    assert rel_err(torch.view_as_real(dv), torch.view_as_real(torch.from_numpy(g[f"{t}/dv"]))) < 1e-10
    assert rel_err(torch.view_as_real(dw), torch.view_as_real(torch.from_numpy(g[f"{t}/dw"]))) < 1e-10


@pytest.mark.parametrize("N,L", CASES)
def test_nonsym_oracle_matches_reference_formula(N, L):
    g = load_golden("cauchy")
    t = f"nonsym/N{N}_L{L}"
    v, z, w, dout = (torch.from_numpy(g[f"{t}/{k}"]).cdouble() for k in ("v", "z", "w", "dout"))
    assert rel_err(torch.view_as_real(oc.cauchy_direct(v, z, w)), torch.view_as_real(torch.from_numpy(g[f"{t}/out"]))) < 1e-11
    dv, dw = oc.cauchy_bwd(v, z, w, dout)
    assert rel_err(torch.view_as_real(dv), torch.view_as_real(torch.from_numpy(g[f"{t}/dv"]))) < 1e-10
    assert rel_err(torch.view_as_real(dw), torch.view_as_real(torch.from_numpy(g[f"{t}/dw"]))) < 1e-10


@pytest.fixture(scope="module")
def clib():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libcauchy_ref.so"))
    return lib


@pytest.mark.parametrize("N,L", [(4, 17), (64, 489)])
def test_c_restatement_matches_reference(clib, N, L):
    g = load_golden("cauchy")
    t = f"sym/N{N}_L{L}"
    arr = {k: np.ascontiguousarray(g[f"{t}/{k}"].astype(np.complex128)) for k in ("v_half", "z", "w_half", "dout")}
    B, n = arr["v_half"].shape
    out = np.zeros((B, L), np.complex128)
    dv = np.zeros((B, n), np.complex128)
    dw = np.zeros((B, n), np.complex128)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    I = ctypes.c_int64
    clib.cauchy_sym_fwd_ref(P(arr["v_half"]), P(arr["z"]), P(arr["w_half"]), P(out), I(B), I(n), I(L))
    clib.cauchy_sym_bwd_ref(P(arr["v_half"]), P(arr["z"]), P(arr["w_half"]), P(arr["dout"]), P(dv), P(dw), I(B), I(n), I(L))
    assert np.abs(out - g[f"{t}/out"]).max() <= 1e-11 * np.abs(g[f"{t}/out"]).max()
    assert np.abs(dv - g[f"{t}/dv"]).max() <= 1e-10 * np.abs(g[f"{t}/dv"]).max()
    assert np.abs(dw - g[f"{t}/dw"]).max() <= 1e-10 * np.abs(g[f"{t}/dw"]).max()

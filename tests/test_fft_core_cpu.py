"""CPU: the FFT core of the S4 convolution kernels (csrc/fft_core.h: radix-16 register passes, bit-reversed spectrum,
real-input pointwise stage) emulated on the host thread by thread (tests/native/fft_core_host.cpp, built here with g++)
against numpy: the transforms themselves, and one row of `fftconv_kernel`'s sequence against the definition of the
two-sided S4 convolution (`s4.py:1391-1406`) evaluated in float64."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import ROOT

SRC = os.path.join(ROOT, "tests", "native", "fft_core_host.cpp")
LIB = os.path.join(ROOT, "tests", "native", "libfft_core_host.so")


@pytest.fixture(scope="module")
def host():
    hdr = os.path.join(ROOT, "diffwave-sashimi_amd", "csrc", "fft_core.h")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", LIB, SRC], check=True)
    lib = ctypes.CDLL(LIB)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.dws_host_fft.argtypes = [ctypes.c_int, fp, fp, ctypes.c_int]
    lib.dws_host_conv_row.argtypes = [ctypes.c_int, fp, ctypes.c_int, fp, fp, fp, fp, fp, ctypes.c_float, fp]
    lib.dws_host_conv_row_fused.argtypes = lib.dws_host_conv_row.argtypes
    lib.dws_host_conv_row_fused4.argtypes = lib.dws_host_conv_row.argtypes
    lib.dws_host_conv_row_x4.argtypes = lib.dws_host_conv_row.argtypes
    fpp = ctypes.POINTER(fp)
    lib.dws_host_conv_long_row.argtypes = [ctypes.c_int, fp, ctypes.c_int, fp, fp, fpp, fpp, fpp, fp]
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _brev(k, bits):
    return int(format(k, f"0{bits}b")[::-1], 2) if bits else 0


def _tables(lg):
    """`build_fft_tables` / `kf_permute_kernel` of fftconv_kernels.hip, restated."""
    M = 1 << lg
    k = np.arange(M // 2)
    tw = np.exp(-2j * np.pi * k / M).astype(np.complex64)
    kk = np.array([_brev(2 * q, lg) for q in range(M // 2)])
    twp = np.exp(-1j * np.pi * kk / M).astype(np.complex64)
    return tw, twp, kk


def _c2f(z):
    return np.ascontiguousarray(z.astype(np.complex64)).view(np.float32)


@pytest.mark.parametrize("lg", [6, 7, 8, 10, 11, 12, 13, 14])
def test_forward_is_the_bit_reversed_dft_and_inverse_undoes_it(host, lg):
    M = 1 << lg
    rng = np.random.default_rng(lg)
    x = (rng.standard_normal(M) + 1j * rng.standard_normal(M)).astype(np.complex64)
    tw, _, _ = _tables(lg)
    d = _c2f(x).copy()
    assert host.dws_host_fft(lg, _p(d), _p(_c2f(tw)), 0) == 0
    got = d.view(np.complex64)
    ref = np.fft.fft(x.astype(np.complex128))
    perm = np.array([_brev(p, lg) for p in range(M)])
    err = np.abs(got - ref[perm]).max() / np.abs(ref).max()
    assert err < 2e-6, err
    assert host.dws_host_fft(lg, _p(d), _p(_c2f(tw)), 1) == 0          # consumes the bit-reversed order
    back = d.view(np.complex64) / M
    assert np.abs(back - x).max() / np.abs(x).max() < 2e-6


@pytest.mark.parametrize("lg,L", [(6, 60), (10, 1000), (10, 64), (12, 4000), (14, 16000), (14, 16384), (11, 1500), (13, 8000)])
@pytest.mark.parametrize("csign", [1.0, -1.0])
def test_convolution_row_matches_the_definition(host, lg, L, csign):
    """y[i] = sum_j k0[j] u[i-j] + sum_{m>=1} k1[m-1] u[i+m] (csign = -1: the adjoint, i.e. correlation with the same
    two-sided kernel), via the kernel's own sequence: half-size complex FFT of the packed real row, pointwise stage in
    bit-reversed pair order against K_f of the re-placed two-sided kernel, mirrored inverse."""
    M, Nf = 1 << lg, 2 << lg
    rng = np.random.default_rng(100 * lg + L)
    u = rng.standard_normal(L).astype(np.float32)
    decay = np.exp(-np.arange(L) / (L / 6.0))
    k0 = (rng.standard_normal(L) * decay).astype(np.float32)
    k1 = (rng.standard_normal(L) * decay).astype(np.float32)
    K = np.zeros(Nf)
    K[:L] = k0
    K[Nf - L:] = k1[::-1]                                          # K[Nf - m] = k1[m - 1], m = 1..L
    Kf = np.fft.rfft(K)                                            # Nf/2 + 1 = M + 1 bins
    tw, twp, kk = _tables(lg)
    kfa, kfb = Kf[kk].astype(np.complex64), Kf[M - kk].astype(np.complex64)
    kfa[0], kfb[0] = Kf[0], Kf[M]
    kfs = np.array([Kf[0], Kf[M], Kf[M // 2]]).astype(np.complex64)
    out = np.zeros(L, dtype=np.float32)
    rc = host.dws_host_conv_row(lg, _p(u), L, _p(_c2f(tw)), _p(_c2f(twp)), _p(_c2f(kfa)), _p(_c2f(kfb)), _p(_c2f(kfs)),
                                csign, _p(out))
    assert rc == 0
    U = np.fft.rfft(np.concatenate([u.astype(np.float64), np.zeros(Nf - L)]))
    ref = np.fft.irfft(U * (Kf if csign > 0 else np.conj(Kf)), n=Nf)[:L]
    if csign > 0 and L <= 4000:                                    # the FFT reference itself against the direct sums
        k0d, k1d, ud = k0.astype(np.float64), k1.astype(np.float64), u.astype(np.float64)
        direct = np.array([np.dot(k0d[:i + 1][::-1], ud[:i + 1]) + np.dot(k1d[:L - 1 - i], ud[i + 1:]) for i in range(L)])
        assert np.abs(direct - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())
    err = np.abs(out - ref).max() / np.abs(ref).max()
    assert err < 5e-6, err
    # the kernel's pair arithmetic (`pointwise_pair4`: packed, fused multiply-adds, the two exact halvings moved into the
    # final 1/M): other roundings, same accuracy against the float64 definition
    x4 = np.zeros(L, dtype=np.float32)
    assert host.dws_host_conv_row_x4(lg, _p(u), L, _p(_c2f(tw)), _p(_c2f(twp)), _p(_c2f(kfa)), _p(_c2f(kfb)), _p(_c2f(kfs)),
                                     csign, _p(x4)) == 0
    errx = np.abs(x4 - ref).max() / np.abs(ref).max()
    assert errx < 5e-6 and errx < 1.5 * err + 1e-7, (errx, err)
    if lg in (6, 10, 14):
        # plans that end in a radix-4 tail run forward tail + pair stage + inverse tail as ONE pass in the kernel
        # (`pass_tail_pointwise`: a thread holds half of a block of 16 positions and half of its mirror block): the same
        # butterflies and pair arithmetic on the same operands, so the row must come out bit for bit the same
        fused = np.zeros(L, dtype=np.float32)
        assert host.dws_host_conv_row_fused(lg, _p(u), L, _p(_c2f(tw)), _p(_c2f(twp)), _p(_c2f(kfa)), _p(_c2f(kfb)),
                                            _p(_c2f(kfs)), csign, _p(fused)) == 0
        assert np.array_equal(fused, out), np.abs(fused - out).max()
        # ... and the fused pass with the packed pair arithmetic (what the kernel runs) equals the separate sequence with it
        f4 = np.zeros(L, dtype=np.float32)
        assert host.dws_host_conv_row_fused4(lg, _p(u), L, _p(_c2f(tw)), _p(_c2f(twp)), _p(_c2f(kfa)), _p(_c2f(kfb)),
                                             _p(_c2f(kfs)), csign, _p(f4)) == 0
        assert np.array_equal(f4, x4), np.abs(f4 - x4).max()


@pytest.mark.parametrize("lg,L,Lt", [(8, 1000, 256), (8, 1000, 100), (8, 513, 256), (10, 2500, 1000), (14, 40000, 16000)])
def test_segmented_convolution_of_rows_longer_than_the_transform(host, lg, L, Lt):
    """`fftconv_seg_kernel`'s sequence (inputs longer than 16384 samples, `generate.py:156` / `s4.py:1387`): segments of
    S = Nf/2 samples; output segment j = first half of IFFT(A_j K_f + A_{j-1} Kc' + A_{j+1} Ka') with Kc' / Ka' the
    spectra of the causal / anti-causal kernel half alone times (-1)^k (a shift by S).  Against the definition with
    Lt <= S taps per direction, in float64."""
    M, Nf = 1 << lg, 2 << lg
    S = M
    rng = np.random.default_rng(7 * lg + L + Lt)
    u = rng.standard_normal(L).astype(np.float32)
    decay = np.exp(-np.arange(Lt) / (Lt / 5.0))
    k0 = (rng.standard_normal(Lt) * decay).astype(np.float32)
    k1 = (rng.standard_normal(Lt) * decay).astype(np.float32)
    Kc, Ka = np.zeros(Nf), np.zeros(Nf)
    Kc[:Lt] = k0
    Ka[Nf - Lt:] = k1[::-1]
    sign = (-1.0) ** np.arange(M + 1)
    specs = [np.fft.rfft(Kc + Ka), sign * np.fft.rfft(Kc), sign * np.fft.rfft(Ka)]
    tw, twp, kk = _tables(lg)
    keep, kfa, kfb, kfs = [], [], [], []
    for Kf in specs:
        a, b = Kf[kk].astype(np.complex64), Kf[M - kk].astype(np.complex64)
        a[0], b[0] = Kf[0], Kf[M]
        c = np.array([Kf[0], Kf[M], Kf[M // 2]]).astype(np.complex64)
        arrs = [_c2f(a), _c2f(b), _c2f(c)]
        keep.append(arrs)
        kfa.append(_p(arrs[0])); kfb.append(_p(arrs[1])); kfs.append(_p(arrs[2]))
    fp = ctypes.POINTER(ctypes.c_float)
    arr3 = lambda ps: (fp * 3)(*ps)
    out = np.zeros(L, dtype=np.float32)
    assert host.dws_host_conv_long_row(lg, _p(u), L, _p(_c2f(tw)), _p(_c2f(twp)), arr3(kfa), arr3(kfb), arr3(kfs), _p(out)) == 0
    # definition: y[i] = sum_{j<Lt} k0[j] u[i-j] + sum_{1<=m<=Lt} k1[m-1] u[i+m], via one big float64 FFT
    n = 1
    while n < L + 2 * Lt:
        n *= 2
    K = np.zeros(n)
    K[:Lt] = k0
    K[n - Lt:] = k1[::-1]
    ref = np.fft.irfft(np.fft.rfft(np.concatenate([u.astype(np.float64), np.zeros(n - L)])) * np.fft.rfft(K), n=n)[:L]
    err = np.abs(out - ref).max() / np.abs(ref).max()
    assert err < 5e-6, err


@pytest.mark.parametrize("lg", [6, 10, 12, 14])
def test_partner_of_a_position_sits_in_the_mirror_block(host, lg):
    """The pair stage couples frequency k with M - k.  In the bit-reversed layout the partner of position 16 t + r is
    position 16 t' + 15 - r with t' = mirror_block(t) = 3 msb(t) - 1 - t (block 0 pairs inside itself): the index fact the
    fused tail pass (`pass_tail_pointwise`) is built on -- a thread holds the r3 = 0 half of block t and the r3 = 1 half of
    block t' and finds all eight pairs (x[e], x[15 - e]) in its registers."""
    M = 1 << lg
    host.dws_host_mirror_block.restype = ctypes.c_int
    for t in range(1, M // 16):
        tm = host.dws_host_mirror_block(t)
        assert host.dws_host_mirror_block(tm) == t and (tm == t) == (t == 1)
        for r in range(16):
            k = _brev(16 * t + r, lg)
            assert _brev((M - k) % M, lg) == 16 * tm + 15 - r
    # block 0: k = 0 and M/2 (positions 0, 1) are their own partners, the rest pairs inside the block
    inside = {r: _brev((M - _brev(r, lg)) % M, lg) for r in range(16)}
    assert inside[0] == 0 and inside[1] == 1
    assert {r: inside[r] for r in (2, 4, 6, 8, 10, 12, 14)} == {2: 3, 4: 7, 6: 5, 8: 15, 10: 13, 12: 11, 14: 9}

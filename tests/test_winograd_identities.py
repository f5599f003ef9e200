"""CPU: the algebra the three Winograd kernels rely on, checked in float64 against the definition of the dilated 3-tap
convolution (`models/wavenet.py:20,95`: `h[o, l] = sum_c sum_t W[o, c, t] x[c, l + (t - 1) d]`, zero padded) and its two
adjoints -- the pairing of positions, the input / weight / gradient transforms and the zero-padding rules, for ragged
lengths, positions past L in the last pair block and d > L.

  forward   csrc/wavenet_wino.hip            y[p], y[p+d] from t0..t3 and G0..G3
  dgrad     csrc/wavenet_backward_wino.hip   the same pairing on dH with the taps reversed (tapwino_mfma_kernel)
  wgrad     csrc/wavenet_backward_wino.hip   dG_j = sum u_j t_j, dW from dG (wgrad_wino_kernel / _reduce_kernel)
"""
import numpy as np
import pytest


def direct_conv(W, x, d):
    """h[o, l] = sum_c sum_t W[o, c, t] x[c, l + (t - 1) d], x zero outside [0, L)."""
    O, C, _ = W.shape
    L = x.shape[1]
    xp = np.zeros((C, L + 2 * d))
    xp[:, d:d + L] = x
    return sum(W[:, :, t] @ xp[:, t * d:t * d + L] for t in range(3))


def pair_columns(L, d):
    """Pair column q -> first position p(q) = (q // d) 2d + q % d, over whole 2d-blocks (the kernels' `nq`)."""
    nq = -(-L // (2 * d)) * d
    q = np.arange(nq)
    return (q // d) * 2 * d + q % d


def at(x, pos):
    """x[:, pos] with zeros outside [0, L) -- what the buffer descriptors' bounds check returns."""
    L = x.shape[1]
    out = np.zeros((x.shape[0], len(pos)))
    ok = (pos >= 0) & (pos < L)
    out[:, ok] = x[:, pos[ok]]
    return out


def transforms(x, p, d):
    d0, d1, d2, d3 = at(x, p - d), at(x, p), at(x, p + d), at(x, p + 2 * d)
    return d0 - d2, d1 + d2, d2 - d1, d1 - d3


def weight_transform(g0, g1, g2):
    return g0, 0.5 * (g0 + g1 + g2), 0.5 * (g0 - g1 + g2), g2


CASES = [(50, 1), (50, 2), (333, 64), (600, 2048), (130, 8), (4100, 256), (7, 4), (1, 1)]


@pytest.mark.parametrize("L,d", CASES)
def test_forward_pairing(L, d):
    rng = np.random.default_rng(L * 7 + d)
    O, C = 6, 5
    W, x = rng.standard_normal((O, C, 3)), rng.standard_normal((C, L))
    ref = direct_conv(W, x, d)
    p = pair_columns(L, d)
    t0, t1, t2, t3 = transforms(x, p, d)
    G0, G1, G2, G3 = weight_transform(W[:, :, 0], W[:, :, 1], W[:, :, 2])
    m0, m1, m2, m3 = G0 @ t0, G1 @ t1, G2 @ t2, G3 @ t3
    y = np.zeros((O, L))
    ok0, ok1 = p < L, p + d < L
    y[:, p[ok0]] = (m0 + m1 + m2)[:, ok0]
    y[:, (p + d)[ok1]] = (m1 - m2 - m3)[:, ok1]
    # every position is the first or the second member of exactly one pair
    covered = np.zeros(L, int)
    np.add.at(covered, p[ok0], 1)
    np.add.at(covered, (p + d)[ok1], 1)
    assert (covered == 1).all()
    np.testing.assert_allclose(y, ref, rtol=0, atol=1e-12 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("L,d", CASES)
def test_data_gradient_pairing(L, d):
    """dx[c, l] = sum_o sum_t W[o, c, t] dH[o, l - (t - 1) d]: the same pairing with g0 = W[.,.,2], g1 = W[.,.,1],
    g2 = W[.,.,0] (`tapwino_pack_transposed_kernel`)."""
    rng = np.random.default_rng(L * 11 + d)
    O, C = 6, 5
    W, dH = rng.standard_normal((O, C, 3)), rng.standard_normal((O, L))
    # the adjoint by its definition: <dH, conv(W, x)> = <dx, x> for every x  ->  dx = conv with transposed, reversed taps
    Wt = np.transpose(W[:, :, ::-1], (1, 0, 2))          # [C][O][3], tap t pairs with dH[l + (t - 1) d]
    ref = direct_conv(Wt, dH, d)
    x = rng.standard_normal((C, L))
    assert abs(np.vdot(dH, direct_conv(W, x, d)) - np.vdot(ref, x)) < 1e-9 * L
    p = pair_columns(L, d)
    t0, t1, t2, t3 = transforms(dH, p, d)
    G0, G1, G2, G3 = weight_transform(W[:, :, 2].T, W[:, :, 1].T, W[:, :, 0].T)
    m0, m1, m2, m3 = G0 @ t0, G1 @ t1, G2 @ t2, G3 @ t3
    dx = np.zeros((C, L))
    ok0, ok1 = p < L, p + d < L
    dx[:, p[ok0]] = (m0 + m1 + m2)[:, ok0]
    dx[:, (p + d)[ok1]] = (m1 - m2 - m3)[:, ok1]
    np.testing.assert_allclose(dx, ref, rtol=0, atol=1e-12 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("L,d", CASES)
def test_weight_gradient_pairing(L, d):
    """dW[o, c, t] = sum_l dH[o, l] Xh[c, l + (t - 1) d] with Xh = x + addc[c] inside [0, L), 0 outside: four products over the
    pair columns, the per-channel constant entering each with the factor in(pa) +- in(pb) (`wgrad_wino_kernel`), and the
    bias gradient as the row sums of u1."""
    rng = np.random.default_rng(L * 13 + d)
    O, C = 6, 5
    dH, x, addc = rng.standard_normal((O, L)), rng.standard_normal((C, L)), rng.standard_normal((C, 1))
    xh = x + addc
    xp = np.zeros((C, L + 2 * d))
    xp[:, d:d + L] = xh
    ref = np.stack([dH @ xp[:, t * d:t * d + L].T for t in range(3)], axis=-1)       # [O][C][3]
    p = pair_columns(L, d)
    y0, y1 = at(dH, p), at(dH, p + d)
    u = (y0, y0 + y1, y0 - y1, -y1)
    offs = ((-d, d, -1.0), (0, d, 1.0), (d, 0, -1.0), (0, 2 * d, -1.0))      # (xa, xb, sign) of t_j = Xh[p+xa] + sign Xh[p+xb]
    dG = []
    for j, (xa, xb, sg) in enumerate(offs):
        pa, pb = p + xa, p + xb
        ina, inb = ((pa >= 0) & (pa < L)).astype(float), ((pb >= 0) & (pb < L)).astype(float)
        t = at(x, pa) + sg * at(x, pb) + addc * (ina + sg * inb)                     # raw loads + constant * mask factor
        dG.append(u[j] @ t.T)
    hs = 0.5 * (dG[1] + dG[2])
    dW = np.stack([dG[0] + hs, 0.5 * (dG[1] - dG[2]), dG[3] + hs], axis=-1)
    np.testing.assert_allclose(dW, ref, rtol=0, atol=1e-11 * max(1.0, np.abs(ref).max()))
    np.testing.assert_allclose(u[1].sum(axis=1), dH.sum(axis=1), rtol=0, atol=1e-11 * L)   # bias gradient rides on u1

"""Pins for the RNN-T loss restatement (numpy and C): the public warp-transducer
known-answer vector, brute-force enumeration of all alignments,
finite differences, numpy-vs-C agreement, ragged lengths."""
import numpy as np
import pytest

from conftest import rel_err
from oracle import tt_oracle as O
from oracle.rnnt_c import rnnt_loss_c

KAT_LOGITS = np.array([[[[.1, .6, .1, .1, .1], [.1, .1, .6, .1, .1], [.1, .1, .2, .8, .1]],
                        [[.1, .6, .1, .1, .1], [.1, .1, .2, .1, .1], [.7, .1, .2, .1, .1]]]])
KAT_LABELS = np.array([[1, 2]])
KAT_COST = 4.49566677
KAT_GRAD = np.array([
    [-0.13116686, -0.39992680, 0.17703122, 0.17703122, 0.17703122],
    [-0.18572753, 0.12247054, -0.18168408, 0.12247054, 0.12247054],
    [-0.32091246, 0.06269139, 0.06928471, 0.12624497, 0.06269139],
    [0.05456068, -0.21824272, 0.05456068, 0.05456068, 0.05456068],
    [0.12073957, 0.12073957, -0.48295828, 0.12073957, 0.12073957],
    [-0.69258820, 0.16871117, 0.18645468, 0.16871117, 0.16871117]]).reshape(1, 2, 3, 5)


@pytest.mark.parametrize("impl", ["numpy_loops", "numpy_diag", "c"])
def test_known_answer_vector(impl):
    if impl == "c":
        loss, costs, g = rnnt_loss_c(KAT_LOGITS, KAT_LABELS, [2], [2])
    else:
        lat = O.rnnt_lattice if impl == "numpy_loops" else O.rnnt_lattice_diag
        loss, costs, g = O.rnnt_loss(KAT_LOGITS, KAT_LABELS, [2], [2], lattice=lat)
    assert abs(costs[0] - KAT_COST) < 2e-6
    assert np.abs(g - KAT_GRAD).max() < 2e-6


@pytest.mark.parametrize("T,U", [(1, 0), (1, 1), (2, 2), (3, 4), (4, 3), (4, 4)])
def test_brute_force_enumeration(T, U):
    rng = np.random.default_rng(T * 10 + U)
    V = 5
    x = rng.normal(size=(1, T, U + 1, V))
    y = rng.integers(1, V, size=(1, max(U, 1)))[:, :U] if U > 0 else np.zeros((1, 0), dtype=np.int64)
    want = O.rnnt_brute_force(x[0], y[0] if U else [])
    _, costs, _ = O.rnnt_loss(x, y.reshape(1, U), [T], [U])
    assert abs(costs[0] - want) < 1e-10
    yc = y.reshape(1, U) if U else np.zeros((1, 1), dtype=np.int32)
    xc = x if U else x
    if U > 0:
        _, cc, _ = rnnt_loss_c(x, yc, [T], [U])
        assert abs(cc[0] - want) < 1e-5


def test_finite_differences_ragged():
    rng = np.random.default_rng(7)
    B, T, U, V = 2, 5, 3, 6
    x = rng.normal(size=(B, T, U + 1, V))
    y = rng.integers(1, V, size=(B, U))
    tl, ul = np.array([5, 3]), np.array([3, 2])
    loss, costs, g = O.rnnt_loss(x, y, tl, ul)
    assert np.all(g[1, 3:] == 0) and np.all(g[1, :, 3:] == 0)
    eps = 1e-6
    for idx in [(0, 0, 0, 0), (0, 4, 3, 0), (0, 2, 1, int(y[0, 1])), (1, 2, 2, 0), (1, 1, 0, int(y[1, 0])), (1, 0, 1, 3)]:
        xp, xm = x.copy(), x.copy()
        xp[idx] += eps
        xm[idx] -= eps
        fd = (O.rnnt_loss(xp, y, tl, ul)[0] - O.rnnt_loss(xm, y, tl, ul)[0]) / (2 * eps)
        assert abs(fd - g[idx]) < 1e-7, idx


def test_numpy_loops_diag_and_c_agree():
    rng = np.random.default_rng(3)
    B, T, U, V = 3, 17, 9, 11
    x = rng.normal(size=(B, T, U + 1, V)).astype(np.float32) * 3
    y = rng.integers(1, V, size=(B, U))
    tl, ul = np.array([17, 12, 1]), np.array([9, 0, 4])
    a = O.rnnt_loss(x.astype(np.float64), y, tl, ul, lattice=O.rnnt_lattice)
    b = O.rnnt_loss(x.astype(np.float64), y, tl, ul, lattice=O.rnnt_lattice_diag)
    c = rnnt_loss_c(x, y, tl, ul)
    assert np.allclose(a[1], b[1], rtol=0, atol=1e-10) and rel_err(a[2], b[2]) < 1e-12
    assert rel_err(c[1], a[1]) < 1e-6 and rel_err(c[2], a[2]) < 1e-6
    assert abs(c[0] - a[0]) / abs(a[0]) < 1e-6


def test_reductions():
    rng = np.random.default_rng(5)
    x = rng.normal(size=(2, 4, 3, 5))
    y = rng.integers(1, 5, size=(2, 2))
    m = O.rnnt_loss(x, y, [4, 4], [2, 2], reduction="mean")
    s = O.rnnt_loss(x, y, [4, 4], [2, 2], reduction="sum")
    assert abs(m[0] * 2 - s[0]) < 1e-12 and rel_err(m[2] * 2, s[2]) < 1e-14


def test_c_oracle_matches_golden_costs(golden):
    z, sd = golden
    for tag in ("full", "ragged"):
        loss, costs, _ = rnnt_loss_c(z["logits"], z["targets"], z[tag + "/act_lens"], z[tag + "/label_lens"])
        assert rel_err(costs, z[tag + "/costs"]) < 1e-6

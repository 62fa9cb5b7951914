"""The oracle's stand-ins for the reference's two third-party eigen-solvers, against scipy / numpy (LAPACK) on inputs made
here: the ONE eigenpair of the Taubin pencil the reference takes from dggev (quadric.cpp:143-153; oracle solve_taubin:
Cholesky with deflation, tridiagonalisation, bisection, twisted factorisation) and the axis it takes from EigenSolver
(quadric.cpp:268-280; oracle smallest_eigvec3).  The end-to-end pin on real scenes is tests/test_e2e_lapack.py; this file
covers what scenes rarely hold: random surfaces, exactly planar patches in every lattice orientation, fewer than nine
points, tiny and huge coordinate scales, repeated eigenvalues."""
import numpy as np
import pytest
import scipy.linalg

from oracle import oracle_py as O

pytestmark = pytest.mark.filterwarnings("ignore")


def build_MN(p):
    """quadric.cpp:24-141 (sum order irrelevant here: the comparison is against a solver fed the same matrices)."""
    x, y, z = (p[:, k].astype(np.float64) for k in range(3))
    one = np.ones_like(x)
    L = np.stack([x * x, y * y, z * z, x * y, y * z, x * z, x, y, z, one], 1)
    M = L.T @ L
    zero = np.zeros_like(x)
    gx = np.stack([2 * x, zero, zero, y, zero, z, one, zero, zero, zero], 1)
    gy = np.stack([zero, 2 * y, zero, x, z, zero, zero, one, zero, zero], 1)
    gz = np.stack([zero, zero, 2 * z, zero, y, x, zero, zero, one, zero], 1)
    N = gx.T @ gx + gy.T @ gy + gz.T @ gz
    return M, N


def lapack_smallest(M, N):
    """quadric.cpp:146-153 with real LAPACK: dggev, alphar / beta, first minimum over the first nine."""
    alphar, alphai, beta, _vl, vr, _w, info = scipy.linalg.lapack.dggev(np.asfortranarray(M), np.asfortranarray(N), compute_vl=0,
                                                                      compute_vr=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        ev = alphar / beta
    mi = 0
    for k in range(1, 9):
        if ev[k] < ev[mi]:
            mi = k
    return ev, vr[:, mi], mi


def angle(u, v):
    c = abs(u @ v) / (np.linalg.norm(u) * np.linalg.norm(v))
    return float(np.arccos(min(1.0, c)))


def gradient_normals(v, p):
    a, b, c, d, e, f, g, h, i = v[:9]
    x, y, z = (p[:, k].astype(np.float64) for k in range(3))
    n = np.stack([2 * a * x + d * y + f * z + g, 2 * b * y + d * x + e * z + h, 2 * c * z + e * y + f * x + i], 1)
    return n / np.linalg.norm(n, axis=1, keepdims=True)


def surface_patch(rng, kind, n=400, scale=1.0, centre=(0.7, 0.05, -0.05)):
    u, v = rng.uniform(-0.03, 0.03, (2, n))
    if kind == "paraboloid":
        w = 8.0 * u * u - 5.0 * v * v + 0.3 * u
    elif kind == "cylinder":
        w = np.sqrt(0.04 ** 2 - np.clip(u, -0.03, 0.03) ** 2) - 0.04
    elif kind == "saddle_noise":
        w = 6.0 * u * v + rng.normal(0, 3e-4, n)
    else:  # blob: no surface at all
        w = rng.uniform(-0.03, 0.03, n)
    R = scipy.linalg.qr(rng.normal(size=(3, 3)))[0]
    p = (np.stack([u, v, w], 1) @ R.T) * scale + np.asarray(centre) * scale
    return p.astype(np.float32)


@pytest.mark.parametrize("kind", ["paraboloid", "cylinder", "saddle_noise", "blob"])
def test_smallest_eigenpair_of_regular_pencils_matches_dggev(kind):
    rng = np.random.default_rng(sum(map(ord, kind)))
    worst = 0.0
    for trial in range(25):
        p = surface_patch(rng, kind)
        M, N = build_MN(p)
        ev, v_ref, mi = lapack_smallest(M, N)
        rc, v, lam = O.solve_taubin(M, N)
        assert rc == 0 and np.isfinite(v).all()
        gap = np.sort(ev[:9])[1] - np.sort(ev[:9])[0]
        if gap < 1e-3 * abs(np.sort(ev[:9])[1]):
            continue  # (near-repeated smallest eigenvalues: the eigenvector is ill-defined for any solver)
        a = angle(v, v_ref)
        worst = max(worst, a)
        assert abs(lam - ev[mi]) <= 1e-6 * abs(ev[mi]) + 1e-9 * np.abs(ev[:9]).max(), (kind, trial)
        # the generalised residual of the oracle's vector is at rounding level
        r = M @ v - lam * (N @ v)
        assert np.linalg.norm(r) <= 1e-8 * np.linalg.norm(M) * np.linalg.norm(v)
    assert worst < 5e-5, worst  # (un-centred coordinates: cond(N9) ~ 1e6; dggev's own noise is of this order)


@pytest.mark.parametrize("normal", [(0, 0, 1), (0, 1, 0), (1, 0, 0), (1, 1, 0), (1, 0, -1), (1, 1, 1)])
def test_exactly_planar_lattice_patch_yields_the_planes_normal(normal):
    """The singular pencil of an exactly planar neighbourhood (both matrices annihilate (n.p - c)^2): the deflated solve
    returns a quadric that vanishes on the points and whose gradients are +-n there, like dggev's."""
    nrm = np.asarray(normal, np.float64)
    g = np.arange(-9, 10)
    pts = []
    for i in g:
        for j in g:
            for k in g:
                if i * normal[0] + j * normal[1] + k * normal[2] == 0 and i * i + j * j + k * k <= 58:
                    pts.append((i, j, k))
    # a lattice of 2^-8 m (3.9 mm) with an offset on the same lattice: every coordinate, and n.p, is exact in float32
    h = 2.0 ** -8
    p = (np.asarray(pts, np.float64) * h + np.array([180, 13, -12]) * h).astype(np.float32)
    q = p.astype(np.float64)
    assert np.ptp(q @ nrm) == 0.0
    M, N = build_MN(p)
    rc, v, lam = O.solve_taubin(M, N)
    assert rc == 0 and np.isfinite(v).all() and abs(lam) < 1e-9
    L = np.stack([q[:, 0] ** 2, q[:, 1] ** 2, q[:, 2] ** 2, q[:, 0] * q[:, 1], q[:, 1] * q[:, 2], q[:, 0] * q[:, 2], q[:, 0], q[:, 1],
                  q[:, 2], np.ones(len(q))], 1)
    assert np.abs(L @ v).max() <= 1e-9 * np.abs(v).max()  # the quadric passes through every point
    n = gradient_normals(v, p)
    cosang = np.abs(n @ (nrm / np.linalg.norm(nrm)))
    assert np.median(cosang) > 1 - 1e-9 and (cosang > 1 - 1e-6).mean() > 0.95  # +-n (a few points sit on the linear factor's zero line)
    # and LAPACK agrees on that normal (its eigenvector differs: the null space is three-dimensional)
    _ev, v_ref, _mi = lapack_smallest(M, N)
    n_ref = gradient_normals(v_ref, p)
    assert np.median(np.abs(n_ref @ (nrm / np.linalg.norm(nrm)))) > 1 - 1e-6


@pytest.mark.parametrize("n_pts", [1, 2, 3, 5, 8])
def test_underdetermined_neighbourhoods_get_a_finite_quadric_through_their_points(n_pts):
    rng = np.random.default_rng(n_pts)
    p = (np.round(rng.uniform(-0.02, 0.02, (n_pts, 3)) / 0.003) * 0.003 + np.array([0.7, 0.05, -0.05])).astype(np.float32)
    p = np.unique(p, axis=0)
    M, N = build_MN(p)
    rc, v, lam = O.solve_taubin(M, N)
    assert rc == 0 and np.isfinite(v).all() and np.abs(v).max() > 0
    q = p.astype(np.float64)
    L = np.stack([q[:, 0] ** 2, q[:, 1] ** 2, q[:, 2] ** 2, q[:, 0] * q[:, 1], q[:, 1] * q[:, 2], q[:, 0] * q[:, 2], q[:, 0], q[:, 1],
                  q[:, 2], np.ones(len(q))], 1)
    assert np.abs(L @ v).max() <= 1e-6 * np.abs(v).max()  # an interpolating quadric (there are infinitely many)


@pytest.mark.parametrize("scale", [1e-3, 1.0, 1e3])
def test_solver_is_scale_covariant(scale):
    """Millimetres or kilometres: the eigenvector's direction in the scaled monomial basis is the same."""
    rng = np.random.default_rng(3)
    p1 = surface_patch(rng, "saddle_noise")  # (a fit with a residual: the eigenvalue is not at the noise level)
    M1, N1 = build_MN(p1)
    _rc, v1, lam1 = O.solve_taubin(M1, N1)
    ps = (p1.astype(np.float64) * scale).astype(np.float64)
    Ms, Ns = build_MN(ps)
    rc, vs, lams = O.solve_taubin(Ms, Ns)
    assert rc == 0 and np.isfinite(vs).all()
    # q_s(p s) = q_1(p): coefficients scale by s^-2 (quadratic), s^-1 (linear), 1 (constant)
    back = vs * np.array([scale ** 2] * 6 + [scale] * 3 + [1.0])
    # (not to rounding: the fit is done in un-centred coordinates, whose conditioning changes with the unit)
    assert angle(back, v1) < 1e-3
    assert abs(lams / scale ** 2 - lam1) <= 1e-3 * abs(lam1)


def test_axis_solver_matches_numpy_eig_and_handles_repeated_eigenvalues():
    rng = np.random.default_rng(11)
    worst = 0.0
    for trial in range(200):
        n = rng.normal(size=(rng.integers(3, 200), 3))
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        n[:, 2] *= rng.choice([1.0, 0.3, 0.05])  # flatten some sets
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        M3 = n.T @ n
        a = O.smallest_eigvec3(M3)
        w, V = np.linalg.eigh(M3)
        assert abs(np.linalg.norm(a) - 1.0) < 1e-12
        if w[1] - w[0] > 1e-6 * w[2]:
            worst = max(worst, angle(a, V[:, 0]))
        assert abs(a @ M3 @ a - w[0]) <= 1e-9 * w[2] + 1e-12  # the Rayleigh quotient is the smallest eigenvalue in any case
    assert worst < 1e-6
    # exactly diagonal with a double zero: the x axis, as a general solver's identity eigenvectors give
    assert np.array_equal(O.smallest_eigvec3(np.diag([0.0, 0.0, 57.0])), [1.0, 0.0, 0.0])
    # all normals parallel up to noise: a unit vector orthogonal to them
    nz = np.array([0.0, 0.0, 1.0]) + rng.normal(0, 1e-11, (300, 3))
    nz /= np.linalg.norm(nz, axis=1, keepdims=True)
    a = O.smallest_eigvec3(nz.T @ nz)
    assert abs(np.linalg.norm(a) - 1.0) < 1e-12 and abs(a[2]) < 1e-6

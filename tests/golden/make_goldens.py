"""Generates the committed golden fixtures in this directory.  Run from the repo root, in the build container:

    python tests/golden/make_goldens.py

What each fixture pins (the reference's own tests hold no expected values -- SURVEY.md section 4 -- so these are
generated here, from the real third-party routines where the container has them):

* taubin_dggev.npz   M, N (10x10 scatter / constraint matrices of seeded neighbourhoods) and the answer of LAPACK
                     ``dggev`` -- the very routine quadric.cpp:353,359 calls -- through scipy: alphar, alphai, beta
                     and the eigenvector of the smallest of the first nine eigenvalues (quadric.cpp:149-152).
* glibc_rand.npz     the first outputs of this container's real glibc ``srand(seed); rand()`` (quadric.cpp:184).
* svm_032015_linear_20_20_same   the linear SVM model *data file* shipped with the reference (model data, not source),
                     needed on the GPU box where /root/reference does not exist.
* svm_weights.npz    the same model parsed with a plain-Python reader (cross-check for the C++/oracle parsers).
"""
from __future__ import annotations

import ctypes
import os
import re
import shutil
import sys

import numpy as np
import scipy.linalg as sl

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from agile_grasp_amd import synthetic  # noqa: E402


def scatter_matrices(P: np.ndarray):
    """M = sum l l^T, N = sum grad(l) grad(l)^T with l = [x2,y2,z2,xy,yz,xz,x,y,z,1] (quadric.cpp:40-141)."""
    x, y, z = P.T.astype(np.float64)
    one, o = np.ones_like(x), np.zeros_like(x)
    l = np.stack([x * x, y * y, z * z, x * y, y * z, x * z, x, y, z, one], 1)
    lx = np.stack([2 * x, o, o, y, o, z, one, o, o, o], 1)
    ly = np.stack([o, 2 * y, o, x, z, o, o, one, o, o], 1)
    lz = np.stack([o, o, 2 * z, o, y, x, o, o, one, o], 1)
    return l.T @ l, lx.T @ lx + ly.T @ ly + lz.T @ lz


def brute_ball(xyz: np.ndarray, q: np.ndarray, r: float) -> np.ndarray:
    d = q[None, :].astype(np.float32) - xyz
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    idx = np.nonzero(d2 < np.float32(r * r))[0]
    order = np.lexsort((idx, d2[idx]))
    return idx[order]


def main():
    sc = synthetic.config("tiny")
    rng = np.random.default_rng(0)
    Ms, Ns, AR, AI, BE, V, MI = [], [], [], [], [], [], []
    picks = list(sc.samples[::3][:20])
    neighbourhoods = [sc.xyz[brute_ball(sc.xyz, sc.xyz[s], 0.03)] for s in picks]
    # analytic patches: cylinder and sphere caps with noise, away from the origin like real clouds
    for k in range(6):
        n = 400
        if k % 2 == 0:
            a = rng.uniform(-0.6, 0.6, n)
            h = rng.uniform(-0.03, 0.03, n)
            P = np.stack([0.7 + 0.04 * np.cos(a), 0.1 + 0.04 * np.sin(a), -0.05 + h], 1)
        else:
            u, v = rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n)
            P = np.stack([0.9 + 0.06 * np.sin(u), -0.2 + 0.06 * np.sin(v) * np.cos(u), 0.06 * np.cos(u) * np.cos(v)], 1)
        P = P + rng.normal(0, 2e-4, P.shape)
        neighbourhoods.append(P.astype(np.float32))
    for P in neighbourhoods:
        M, N = scatter_matrices(P)
        ar, ai, be, _vl, vr, _work, info = sl.lapack.dggev(M, N, compute_vl=0, compute_vr=1)
        assert info == 0
        with np.errstate(divide="ignore", invalid="ignore"):
            ev = ar / be
        mi = int(np.argmin(ev[:9]))
        Ms.append(M), Ns.append(N), AR.append(ar), AI.append(ai), BE.append(be), V.append(vr[:, mi]), MI.append(mi)
    np.savez_compressed(os.path.join(HERE, "taubin_dggev.npz"), M=np.array(Ms), N=np.array(Ns), alphar=np.array(AR),
                        alphai=np.array(AI), beta=np.array(BE), v=np.array(V), min_index=np.array(MI))

    libc = ctypes.CDLL("libc.so.6")
    libc.rand.restype = ctypes.c_int
    seeds = [1, 42, 20150501]
    vals = []
    for s in seeds:
        libc.srand(ctypes.c_uint(s))
        vals.append([libc.rand() for _ in range(500)])
    np.savez_compressed(os.path.join(HERE, "glibc_rand.npz"), seeds=np.array(seeds, np.uint32),
                        values=np.array(vals, np.int32))

    src = "/root/reference/svm_032015_linear_20_20_same"
    dst = os.path.join(HERE, "svm_032015_linear_20_20_same")
    if os.path.exists(src):
        shutil.copyfile(src, dst)
    txt = open(dst).read()
    body = txt[txt.index("support_vectors:"):txt.index("decision_functions:")]
    body = body[body.index("[") + 1:body.index("]")]
    w = np.array([np.float32(float(t)) for t in re.split(r"[,\s]+", body.strip()) if t], np.float32)
    rho = float(re.search(r"rho:\s*([-+0-9.eE]+)", txt).group(1))
    assert w.size == 3528
    np.savez_compressed(os.path.join(HERE, "svm_weights.npz"), w=w, rho=np.float64(rho))
    print("goldens written:", len(Ms), "eigen cases,", len(seeds), "rand seeds, svm", w.size, rho)


if __name__ == "__main__":
    main()

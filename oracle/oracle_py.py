"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by
the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NORMALS_DETERMINISTIC = 0
NORMALS_RAND50 = 1


class OrcParams(C.Structure):
    _fields_ = [
        ("finger_width", C.c_double),
        ("hand_outer_diameter", C.c_double),
        ("hand_depth", C.c_double),
        ("hand_height", C.c_double),
        ("init_bite", C.c_double),
        ("nn_radius_taubin", C.c_double),
        ("nn_radius_hands", C.c_double),
        ("nn_radius_normals", C.c_double),
        ("cam_origin", (C.c_double * 3) * 2),
        ("normals_mode", C.c_int32),
        ("rand_seed", C.c_uint32),
        ("num_threads", C.c_int32),
        ("pow6_libm", C.c_int32),
    ]


HYP_DTYPE = np.dtype(
    [
        ("axis", "<f8", 3),
        ("approach", "<f8", 3),
        ("binormal", "<f8", 3),
        ("bottom", "<f8", 3),
        ("surface", "<f8", 3),
        ("width", "<f8"),
        ("sample", "<i4"),
        ("orientation", "<i4"),
        ("cam_source", "<i4"),
        ("n_in_box", "<i4"),
        ("half_antipodal", "u1"),
        ("full_antipodal", "u1"),
        ("svm_keep", "u1"),
        ("valid", "u1"),
        ("finger_index", "<i4"),
        ("depth_index", "<i4"),
        ("epoch", "<i4"),
    ]
)
assert HYP_DTYPE.itemsize == 160

FRAME_DTYPE = np.dtype(
    [
        ("sample", "<f8", 3),
        ("normal", "<f8", 3),
        ("axis", "<f8", 3),
        ("binormal", "<f8", 3),
        ("params", "<f8", 10),
        ("eigenvalue", "<f8"),
        ("n_nb", "<i4"),
        ("majority_cam", "<i4"),
        ("max_index", "<i4"),
        ("valid", "<i4"),
    ]
)
assert FRAME_DTYPE.itemsize == 200


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "agile_oracle.cpp")
    hdr = os.path.join(_HERE, "agile_oracle.h")
    stale = (not os.path.exists(so)) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_radius_search.restype = C.c_int64
    return _LIB


def default_params(cam_origins: np.ndarray, normals_mode: int = NORMALS_DETERMINISTIC, num_threads: int = 0,
                   **over) -> OrcParams:
    p = OrcParams()
    p.finger_width, p.hand_outer_diameter, p.hand_depth = 0.01, 0.09, 0.06
    p.hand_height, p.init_bite = 0.02, 0.01
    p.nn_radius_taubin, p.nn_radius_hands, p.nn_radius_normals = 0.03, 0.08, 0.01
    for c in range(2):
        for r in range(3):
            p.cam_origin[c][r] = float(cam_origins[c][r])
    p.normals_mode = normals_mode
    p.rand_seed = 1
    p.num_threads = num_threads if num_threads > 0 else (os.cpu_count() or 1)
    p.pow6_libm = 0
    for k, v in over.items():
        setattr(p, k, v)
    return p


def _fp(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def radius_search(xyz: np.ndarray, q, radius: float):
    xyz = np.ascontiguousarray(xyz, np.float32)
    cap = xyz.shape[0]
    idx = np.empty(cap, np.int32)
    d2 = np.empty(cap, np.float32)
    qq = (C.c_float * 3)(*[float(v) for v in q])
    n = lib().orc_radius_search(_fp(xyz, C.c_float), C.c_int64(3), C.c_int64(xyz.shape[0]), qq, C.c_double(radius),
                                _fp(idx, C.c_int32), _fp(d2, C.c_float), C.c_int64(cap))
    return idx[:n].copy(), d2[:n].copy()


def fit_frames(p: OrcParams, xyz, cam, samples, radius: float) -> np.ndarray:
    xyz = np.ascontiguousarray(xyz, np.float32)
    cam = np.ascontiguousarray(cam, np.int32)
    samples = np.ascontiguousarray(samples, np.int32)
    fr = np.zeros(samples.shape[0], FRAME_DTYPE)
    rc = lib().orc_fit_frames(C.byref(p), _fp(xyz, C.c_float), C.c_int64(3), _fp(cam, C.c_int32),
                              C.c_int64(xyz.shape[0]), _fp(samples, C.c_int32), C.c_int64(samples.shape[0]),
                              C.c_double(radius), fr.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return fr


def find_hands(p: OrcParams, xyz, cam, samples, calculates_antipodal: bool = False, want_images: bool = False,
               want_frames: bool = True):
    xyz = np.ascontiguousarray(xyz, np.float32)
    cam = np.ascontiguousarray(cam, np.int32)
    samples = np.ascontiguousarray(samples, np.int32)
    S = samples.shape[0]
    cap = 8 * S
    out = np.zeros(cap, HYP_DTYPE)
    frames = np.zeros(S, FRAME_DTYPE)
    nh = np.zeros(S, np.int32)
    images = np.zeros((cap, 8000), np.uint8) if want_images else None
    n_out = C.c_int64(0)
    rc = lib().orc_find_hands(C.byref(p), _fp(xyz, C.c_float), C.c_int64(3), _fp(cam, C.c_int32),
                              C.c_int64(xyz.shape[0]), _fp(samples, C.c_int32), C.c_int64(S),
                              C.c_int(1 if calculates_antipodal else 0), out.ctypes.data_as(C.c_void_p), C.c_int64(cap),
                              C.byref(n_out), frames.ctypes.data_as(C.c_void_p), _fp(nh, C.c_int32),
                              images.ctypes.data_as(C.c_void_p) if want_images else None)
    assert rc == 0, rc
    n = n_out.value
    return {"hyps": out[:n].copy(), "frames": frames, "nh": nh, "images": images[:n].copy() if want_images else None}


def hands_from_frames(p: OrcParams, xyz, cam, samples, frames, normals=None, want_images: bool = False):
    xyz = np.ascontiguousarray(xyz, np.float32)
    cam = np.ascontiguousarray(cam, np.int32)
    samples = np.ascontiguousarray(samples, np.int32)
    frames = np.ascontiguousarray(frames)
    assert frames.dtype == FRAME_DTYPE
    S = samples.shape[0]
    cap = 8 * S
    out = np.zeros(cap, HYP_DTYPE)
    nh = np.zeros(S, np.int32)
    images = np.zeros((cap, 8000), np.uint8) if want_images else None
    n_out = C.c_int64(0)
    if normals is not None:
        normals = np.ascontiguousarray(normals, np.float64)
    rc = lib().orc_hands_from_frames(C.byref(p), _fp(xyz, C.c_float), C.c_int64(3), _fp(cam, C.c_int32),
                                     C.c_int64(xyz.shape[0]), _fp(samples, C.c_int32), C.c_int64(S),
                                     frames.ctypes.data_as(C.c_void_p),
                                     normals.ctypes.data_as(C.c_void_p) if normals is not None else None,
                                     out.ctypes.data_as(C.c_void_p), C.c_int64(cap), C.byref(n_out),
                                     _fp(nh, C.c_int32), images.ctypes.data_as(C.c_void_p) if want_images else None)
    assert rc == 0, rc
    n = n_out.value
    return {"hyps": out[:n].copy(), "nh": nh, "images": images[:n].copy() if want_images else None}


def hog(image: np.ndarray) -> np.ndarray:
    image = np.ascontiguousarray(image, np.uint8).reshape(8000)
    desc = np.zeros(3528, np.float32)
    lib().orc_hog(_fp(image, C.c_uint8), _fp(desc, C.c_float))
    return desc


def load_svm(path: str):
    w = np.zeros(3528, np.float32)
    rho = C.c_double(0)
    n = lib().orc_load_svm(path.encode(), _fp(w, C.c_float), C.c_int32(3528), C.byref(rho))
    if n != 3528:
        raise RuntimeError(f"SVM parse failed ({n})")
    return w, rho.value


def classify(images: np.ndarray, w: np.ndarray, rho: float, num_threads: int = 0):
    images = np.ascontiguousarray(images, np.uint8).reshape(-1, 8000)
    w = np.ascontiguousarray(w, np.float32)
    keep = np.zeros(images.shape[0], np.uint8)
    sums = np.zeros(images.shape[0], np.float64)
    rc = lib().orc_classify(_fp(images, C.c_uint8), C.c_int64(images.shape[0]), _fp(w, C.c_float), C.c_int32(w.size),
                            C.c_double(rho), _fp(keep, C.c_uint8), _fp(sums, C.c_double),
                            C.c_int(num_threads if num_threads > 0 else (os.cpu_count() or 1)))
    assert rc == 0
    return keep, sums


def glibc_rand(seed: int, count: int) -> np.ndarray:
    out = np.zeros(count, np.int32)
    lib().orc_glibc_rand(C.c_uint32(seed), _fp(out, C.c_int32), C.c_int64(count))
    return out


def solve_taubin(M: np.ndarray, N: np.ndarray):
    M = np.ascontiguousarray(M, np.float64)
    N = np.ascontiguousarray(N, np.float64)
    v = np.zeros(10)
    lam = C.c_double(0)
    rc = lib().orc_solve_taubin(_fp(M, C.c_double), _fp(N, C.c_double), _fp(v, C.c_double), C.byref(lam))
    return rc, v, lam.value


def smallest_eigvec3(M3: np.ndarray) -> np.ndarray:
    M3 = np.ascontiguousarray(M3, np.float64)
    out = np.zeros(3)
    lib().orc_smallest_eigvec3(_fp(M3, C.c_double), _fp(out, C.c_double))
    return out


def preprocess(xyz: np.ndarray, size_left: int, workspace, cell_size: float = 0.003, dense: bool = False):
    """NaN removal + workspace filter + per-camera voxelisation (localization.cpp:17-45,216-355)."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    n = xyz.shape[0]
    ws = np.ascontiguousarray(workspace, np.float64)
    out = np.zeros((max(n, 1), 3), np.float32)
    cam = np.zeros(max(n, 1), np.int32)
    f = lib().orc_preprocess
    f.restype = C.c_int64
    k = f(_fp(xyz, C.c_float), C.c_int64(xyz.shape[1] if n else 3), C.c_int64(n), C.c_int64(size_left),
          C.c_int(1 if dense else 0), _fp(ws, C.c_double), C.c_double(cell_size), _fp(out, C.c_float),
          _fp(cam, C.c_int32), C.c_int64(out.shape[0]))
    return out[:k].copy(), cam[:k].copy()


HANDLE_DTYPE = np.dtype([("axis", "<f8", 3), ("center", "<f8", 3), ("approach", "<f8", 3), ("binormal", "<f8", 3),
                         ("hands_center", "<f8", 3), ("width", "<f8"), ("n_inliers", "<i4"), ("first_inlier", "<i4")])
assert HANDLE_DTYPE.itemsize == 136


def find_handles(hands: np.ndarray, min_inliers: int = 3, min_length: float = 0.005):
    """HandleSearch::findHandles + Handle (handle_search.cpp:4-128, handle.cpp:3-74) on hypothesis records."""
    hands = np.ascontiguousarray(hands, HYP_DTYPE)
    H = hands.shape[0]
    out = np.zeros(max(H, 1), HANDLE_DTYPE)
    idx = np.zeros(max(H, 1), np.int32)
    f = lib().orc_find_handles
    f.restype = C.c_int64
    n = f(hands.ctypes.data_as(C.c_void_p), C.c_int64(H), C.c_int32(min_inliers), C.c_double(min_length),
          out.ctypes.data_as(C.c_void_p), C.c_int64(out.shape[0]), _fp(idx, C.c_int32), C.c_int64(idx.shape[0]))
    assert n >= 0
    out = out[:n].copy()
    total = int(out["n_inliers"].sum())
    return out, idx[:total].copy()


def find_hands_points(p: OrcParams, xyz, cam, samples):
    """Hypotheses plus, per hypothesis, points_for_learning (3, n_b) and the camera id of each column."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    cam = np.ascontiguousarray(cam, np.int32)
    samples = np.ascontiguousarray(samples, np.int32)
    S = samples.shape[0]
    cap = 8 * S
    out = np.zeros(cap, HYP_DTYPE)
    ofs = np.zeros(cap + 1, np.int64)
    n_out = C.c_int64(0)
    f = lib().orc_find_hands_points
    f.restype = C.c_int64
    args = lambda pts, pc, pcap: (C.byref(p), _fp(xyz, C.c_float), C.c_int64(3), _fp(cam, C.c_int32), C.c_int64(xyz.shape[0]),
                                  _fp(samples, C.c_int32), C.c_int64(S), out.ctypes.data_as(C.c_void_p), C.c_int64(cap),
                                  C.byref(n_out), _fp(pts, C.c_double), _fp(pc, C.c_int32), C.c_int64(pcap),
                                  _fp(ofs, C.c_int64))
    total = f(*args(np.zeros(3), np.zeros(1, np.int32), 0))
    assert total >= 0, total
    pts = np.zeros((max(total, 1), 3), np.float64)
    pc = np.zeros(max(total, 1), np.int32)
    assert f(*args(pts, pc, total)) == total
    n = n_out.value
    return {"hyps": out[:n].copy(), "points": [pts[ofs[k]:ofs[k + 1]].T.copy() for k in range(n)],
            "cams": [pc[ofs[k]:ofs[k + 1]].copy() for k in range(n)]}


# ---- f4: the training side (learning.cpp:3-163, 249-318) -----------------------------------------------------
def find_hands_training(p: OrcParams, xyz, cam, samples):
    """find_hands(calculates_antipodal=True) plus, per hypothesis, the images of the three training instances
    createInstance(h, cam_pos, cam = -1 / 0 / 1): 'images' is (H, 3, 8000)."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    cam = np.ascontiguousarray(cam, np.int32)
    samples = np.ascontiguousarray(samples, np.int32)
    S = samples.shape[0]
    cap = 8 * S
    out = np.zeros(cap, HYP_DTYPE)
    images = np.zeros((cap, 8000), np.uint8)
    cam_images = np.zeros((cap, 2, 8000), np.uint8)
    n_out = C.c_int64(0)
    rc = lib().orc_find_hands_training(C.byref(p), _fp(xyz, C.c_float), C.c_int64(3), _fp(cam, C.c_int32),
                                       C.c_int64(xyz.shape[0]), _fp(samples, C.c_int32), C.c_int64(S),
                                       out.ctypes.data_as(C.c_void_p), C.c_int64(cap), C.byref(n_out),
                                       _fp(images, C.c_uint8), _fp(cam_images, C.c_uint8))
    assert rc == 0, rc
    n = n_out.value
    return {"hyps": out[:n].copy(), "images": np.concatenate([images[:n, None, :], cam_images[:n]], axis=1)}


def hog_many(images: np.ndarray) -> np.ndarray:
    images = np.ascontiguousarray(images, np.uint8).reshape(-1, 8000)
    return np.stack([hog(im) for im in images]) if images.shape[0] else np.zeros((0, 3528), np.float32)


def train_svm(features: np.ndarray, labels: np.ndarray, C_: float = 1.0, max_iter: int = 1000,
              eps: float = 1.1920928955078125e-07, num_threads: int = 0, kernel: int = 0):
    """CvSVM::train(C_SVC, LINEAR or POLY degree 2) restated (parity with OpenCV itself unpinned).
    Returns dict(w [LINEAR: the compacted vector], rho, iterations, n_sv, alpha, sv_order, model) where model =
    (kernel, sv, alpha, rho) is what CvSVM::save would hold."""
    features = np.ascontiguousarray(features, np.float32)
    n, d = features.shape
    lab = np.ascontiguousarray(np.where(np.asarray(labels) > 0, 1, -1), np.int8)
    w = np.zeros(d, np.float32)
    rho = C.c_double(0)
    info = np.zeros(4, np.int32)
    alpha = np.zeros(n, np.float64)
    sv_order = np.zeros(n, np.int32)
    rc = lib().orc_train_svm(_fp(features, C.c_float), lab.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_int32(d),
                             C.c_int32(kernel), C.c_double(C_), C.c_int32(max_iter), C.c_double(eps), _fp(w, C.c_float),
                             C.byref(rho), _fp(info, C.c_int32), _fp(alpha, C.c_double), _fp(sv_order, C.c_int32),
                             C.c_int(num_threads if num_threads > 0 else min(os.cpu_count() or 1, 32)))
    if rc != 0:
        raise RuntimeError(f"orc_train_svm failed ({rc})")
    sv_order = sv_order[: int(info[1])]
    if kernel == 0:
        model = (0, w[None, :].copy(), np.ones(1), rho.value)
    else:
        model = (1, features[sv_order].copy(), alpha[sv_order].copy(), rho.value)
    return {"w": w, "rho": rho.value, "iterations": int(info[0]), "n_sv": int(info[1]), "alpha": alpha,
            "sv_order": sv_order, "model": model}


def save_svm_model(path: str, model) -> None:
    kernel, sv, alpha, rho = model
    sv = np.ascontiguousarray(sv, np.float32)
    alpha = np.ascontiguousarray(alpha, np.float64)
    rc = lib().orc_save_svm_model(path.encode(), C.c_int32(kernel), _fp(sv, C.c_float), C.c_int32(sv.shape[0]),
                                  C.c_int32(sv.shape[1]), _fp(alpha, C.c_double), C.c_double(rho))
    assert rc == 0, rc


def load_svm_model(path: str, sv_cap: int = 1 << 16):
    kernel = C.c_int32(0)
    rho = C.c_double(0)
    probe = np.zeros((1, 3528), np.float32)
    a1 = np.zeros(1, np.float64)
    n = lib().orc_load_svm_model(path.encode(), C.byref(kernel), _fp(probe, C.c_float), C.c_int32(1), C.c_int32(3528),
                                 _fp(a1, C.c_double), C.byref(rho))
    if n < 0:
        raise RuntimeError(f"SVM parse failed ({n})")
    sv = np.zeros((n, 3528), np.float32)
    alpha = np.zeros(n, np.float64)
    n2 = lib().orc_load_svm_model(path.encode(), C.byref(kernel), _fp(sv, C.c_float), C.c_int32(n), C.c_int32(3528),
                                  _fp(alpha, C.c_double), C.byref(rho))
    assert n2 == n
    return kernel.value, sv, alpha, rho.value


def classify_model(images: np.ndarray, model, num_threads: int = 0):
    kernel, sv, alpha, rho = model
    images = np.ascontiguousarray(images, np.uint8).reshape(-1, 8000)
    sv = np.ascontiguousarray(sv, np.float32)
    alpha = np.ascontiguousarray(alpha, np.float64)
    keep = np.zeros(images.shape[0], np.uint8)
    sums = np.zeros(images.shape[0], np.float64)
    rc = lib().orc_classify_model(_fp(images, C.c_uint8), C.c_int64(images.shape[0]), C.c_int32(kernel), _fp(sv, C.c_float),
                                  C.c_int32(sv.shape[0]), C.c_int32(3528), _fp(alpha, C.c_double), C.c_double(rho),
                                  _fp(keep, C.c_uint8), _fp(sums, C.c_double),
                                  C.c_int(num_threads if num_threads > 0 else (os.cpu_count() or 1)))
    assert rc == 0
    return keep, sums


def save_svm(path: str, w: np.ndarray, rho: float) -> None:
    w = np.ascontiguousarray(w, np.float32)
    rc = lib().orc_save_svm(path.encode(), _fp(w, C.c_float), C.c_int32(w.size), C.c_double(rho))
    assert rc == 0, rc

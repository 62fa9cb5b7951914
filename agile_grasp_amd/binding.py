"""ctypes binding of libagile_grasp_hip.so (the C ABI of include/agh.h) for tests and bench.py.

The product is the shared library; this module only marshals numpy / torch buffers into it.  There is no CPU
fallback: if the library or a gfx950 device is missing, loading or ``Context()`` raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NORMALS_DETERMINISTIC = 0
NORMALS_RAND50 = 1


class AghParams(C.Structure):
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
        ("device", C.c_int32),
        ("profile", C.c_int32),
    ]


class AghTiming(C.Structure):
    _fields_ = [("ms", C.c_float * 16), ("name", C.c_char_p * 16), ("n", C.c_int32), ("total_ms", C.c_float)]


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
HANDLE_DTYPE = np.dtype([("axis", "<f8", 3), ("center", "<f8", 3), ("approach", "<f8", 3), ("binormal", "<f8", 3),
                         ("hands_center", "<f8", 3), ("width", "<f8"), ("n_inliers", "<i4"), ("first_inlier", "<i4")])
assert HYP_DTYPE.itemsize == 160 and FRAME_DTYPE.itemsize == 200 and HANDLE_DTYPE.itemsize == 136

EXPORTS = [
    "agh_default_params", "agh_create", "agh_destroy", "agh_last_error", "agh_set_cloud", "agh_set_cloud_device", "agh_set_cloud_batch", "agh_set_cloud_batch_device",
    "agh_preprocess", "agh_preprocess_device", "agh_localize", "agh_localize_device", "agh_localize_begin", "agh_localize_stage", "agh_localize_end", "agh_get_cloud", "agh_find_handles", "agh_find_hands", "agh_find_hands_device", "agh_load_svm", "agh_load_svm_file", "agh_classify",
    "agh_classify_device", "agh_get_frames", "agh_get_neighbor_counts", "agh_get_images", "agh_get_hog",
    "agh_get_normals", "agh_get_timing", "agh_get_timing_counts", "agh_set_profile", "agh_synchronize", "agh_selftest_math",
    "agh_set_training_images", "agh_get_training_images", "agh_hog_images", "agh_train_svm", "agh_save_svm_file",
    "agh_load_svm_model", "agh_get_learning_points", "agh_get_epoch", "agh_get_packed_images", "agh_classify_images", "agh_comm_rccl_origin",
    "agh_save_svm_file_ex", "agh_comm_unique_id", "agh_comm_init", "agh_comm_init_local", "agh_comm_destroy", "agh_comm_rank", "agh_comm_last_count", "agh_comm_last_exchange", "agh_comm_set_segment_records", "agh_comm_inject_fault",
    "agh_shard_slice", "agh_find_hands_sharded_device", "agh_find_hands_sharded", "agh_classify_sharded_device",
    "agh_classify_sharded",
]


def comm_unique_id() -> bytes:
    """ncclGetUniqueId: call on one rank, hand the 128 bytes to the others."""
    buf = (C.c_uint8 * 128)()
    rc = load_library().agh_comm_unique_id(buf)
    if rc != 0:
        raise AghError(rc, "agh_comm_unique_id failed (is RCCL installed?)")
    return bytes(buf)


def comm_init_local(contexts) -> None:
    """The contexts (one host thread each) become the ranks of an in-process communicator: the sharded schedule with
    device copies instead of RCCL, for validation on a single GPU."""
    arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    rc = load_library().agh_comm_init_local(arr, C.c_int32(len(contexts)))
    if rc != 0:
        raise AghError(rc, "agh_comm_init_local failed")


def shard_slice(n: int, rank: int, n_ranks: int):
    lo, hi = C.c_int64(0), C.c_int64(0)
    load_library().agh_shard_slice(C.c_int64(n), C.c_int32(rank), C.c_int32(n_ranks), C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def comm_rccl_origin() -> str:
    """Which RCCL image the library bound (binds it if that has not happened yet)."""
    lib = load_library()
    lib.agh_comm_rccl_origin.restype = C.c_char_p
    return lib.agh_comm_rccl_origin().decode()


def pack_images(images: np.ndarray) -> np.ndarray:
    """(n, 8000) uint8 images (0 / 255) -> (n, 250) uint32 words, bit (b & 31) of word (b >> 5) = pixel b."""
    im = np.ascontiguousarray(images, np.uint8).reshape(-1, 8000) != 0
    return np.packbits(im, axis=1, bitorder="little").view("<u4").reshape(-1, 250).copy()


def unpack_images(words: np.ndarray) -> np.ndarray:
    w = np.ascontiguousarray(words, "<u4").reshape(-1, 250)
    return (np.unpackbits(w.view(np.uint8), axis=1, bitorder="little") * np.uint8(255)).reshape(-1, 8000)


SVM_LINEAR = 0
SVM_POLY2 = 1


def save_svm_file(path: str, w: np.ndarray, rho: float, kernel: int = SVM_LINEAR, alpha: np.ndarray | None = None) -> None:
    """CvSVM::save (needs no device): the compacted linear vector (w: 3528 floats) or, with `alpha`, the support
    vectors (w: n_sv x 3528) of a model with the given kernel."""
    sv = np.ascontiguousarray(w, np.float32).reshape(-1, 3528)
    al = np.ones(1, np.float64) if alpha is None else np.ascontiguousarray(alpha, np.float64)
    assert al.shape[0] == sv.shape[0]
    rc = load_library().agh_save_svm_file(path.encode(), C.c_int32(kernel), _p(sv, C.c_float), C.c_int32(sv.shape[0]),
                                          C.c_int32(3528), _p(al, C.c_double), C.c_double(rho))
    if rc != 0:
        raise AghError(rc, f"cannot write {path}")


class AghLocalizeParams(C.Structure):
    _fields_ = [("size_left", C.c_int64), ("dense", C.c_int32), ("classify", C.c_int32), ("workspace", C.c_double * 6),
                ("cell_size", C.c_double), ("sample_idx", C.POINTER(C.c_int32)), ("n_samples", C.c_int64),
                ("sample_seed", C.c_uint64), ("min_inliers", C.c_int32), ("reserved", C.c_int32), ("min_length", C.c_double)]


class AghLocalizeResult(C.Structure):
    _fields_ = [("n_voxels", C.c_int64), ("n_hypotheses", C.c_int64), ("n_hands", C.c_int64), ("n_handles", C.c_int64),
                ("n_inlier_idx", C.c_int64)]


def draw_samples(n_points: int, n_samples: int, seed: int) -> np.ndarray:
    """The sample list agh_localize draws on the device for sample_idx = NULL (include/agh.h): one index per stratum."""
    M = (1 << 64) - 1

    def splitmix64(x):
        x = (x + 0x9E3779B97F4A7C15) & M
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
        return x ^ (x >> 31)

    out = np.empty(n_samples, np.int32)
    for k in range(n_samples):
        if n_points >= n_samples:
            lo, hi = (k * n_points) // n_samples, ((k + 1) * n_points) // n_samples
            out[k] = lo + splitmix64((seed ^ (k * 0x9E3779B97F4A7C15)) & M) % (hi - lo)
        else:
            out[k] = k if k < n_points else -(1 << 31)
    return out


AGH_ERR_INVALID_ARGUMENT, AGH_ERR_HIP, AGH_ERR_CAPACITY, AGH_ERR_NO_CLOUD, AGH_ERR_NO_SVM, AGH_ERR_STATE = -1, -3, -4, -5, -6, -8
AGH_ERR_RETRY = -9  # the context adapted its configuration to the input (include/agh.h): repeat the call


class AghError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"agh error {code}: {msg}")
        self.code = code


def library_path() -> str:
    return os.path.join(_HERE, "lib", "libagile_grasp_hip.so")


def load_library():
    """dlopen the HIP library.  torch is imported first when available so that both share one HIP runtime."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                "agile_grasp_amd has no CPU implementation.")
        if os.environ.get("AGH_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401  (loads torch's libamdhip64 first; ours then binds to the same runtime)
            except Exception:
                pass
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        lib.agh_last_error.restype = C.c_char_p
        lib.agh_last_error.argtypes = [C.c_void_p]
        lib.agh_selftest_math.restype = C.c_int64
        lib.agh_destroy.restype = None
        lib.agh_default_params.restype = None
        lib.agh_shard_slice.restype = None
        _LIB = lib
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Context:
    """One agh_ctx: HandSearch + Learning::classify state for one cloud on one GPU."""

    def __init__(self, cam_origins, normals_mode: int = NORMALS_DETERMINISTIC, device: int = 0, profile: bool = False,
                 rand_seed: int = 1, **geometry):
        self.lib = load_library()
        p = AghParams()
        self.lib.agh_default_params(C.byref(p))
        for c in range(2):
            for r in range(3):
                p.cam_origin[c][r] = float(cam_origins[c][r])
        p.normals_mode, p.device, p.profile, p.rand_seed = normals_mode, device, int(profile), rand_seed
        for k, v in geometry.items():
            setattr(p, k, v)
        self.params = p
        self._h = C.c_void_p()
        rc = self.lib.agh_create(C.byref(p), C.byref(self._h))
        if rc != 0:
            raise AghError(rc, self.lib.agh_last_error(None).decode())
        self._keep = []  # device tensors that must outlive the context's use of them

    def close(self):
        if self._h:
            self.lib.agh_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc < 0:
            raise AghError(rc, self.lib.agh_last_error(self._h).decode())
        return rc

    # ---- host-buffer API ----
    def set_cloud(self, xyz: np.ndarray, cam: np.ndarray | None):
        xyz = np.ascontiguousarray(xyz, np.float32)
        assert xyz.ndim == 2 and xyz.shape[1] >= 3
        stride = xyz.shape[1] * 4  # (numpy reports arbitrary strides for empty arrays)
        camp = None
        if cam is not None:
            cam = np.ascontiguousarray(cam, np.int32)
            camp = _p(cam, C.c_int32)
        self._check(self.lib.agh_set_cloud(self._h, _p(xyz, C.c_float), C.c_int64(stride), camp,
                                           C.c_int64(xyz.shape[0])))
        self.n = xyz.shape[0]

    def set_cloud_batch(self, clouds, cams):
        """A batch of clouds in one context: `clouds` / `cams` are lists of (n_k, 3) float32 / (n_k,) int32 arrays.  Returns
        the offsets; point and sample indices of later calls are positions in the concatenation."""
        xyz = np.ascontiguousarray(np.concatenate([np.asarray(c, np.float32)[:, :3] for c in clouds]), np.float32)
        cam = np.ascontiguousarray(np.concatenate([np.asarray(c, np.int32) for c in cams]), np.int32)
        off = np.zeros(len(clouds) + 1, np.int64)
        off[1:] = np.cumsum([len(c) for c in clouds])
        self._check(self.lib.agh_set_cloud_batch(self._h, _p(xyz, C.c_float), C.c_int64(12), _p(cam, C.c_int32),
                                                 _p(off, C.c_int64), C.c_int32(len(clouds))))
        self.n = xyz.shape[0]
        return off

    def set_cloud_batch_torch(self, xyz_t, cam_t, offsets, stream=None):
        assert xyz_t.is_cuda and xyz_t.is_contiguous()
        off = np.ascontiguousarray(offsets, np.int64)
        self._keep = [xyz_t, cam_t]
        self.n = xyz_t.shape[0]
        self._check(self.lib.agh_set_cloud_batch_device(
            self._h, C.c_void_p(xyz_t.data_ptr()), C.c_int64(xyz_t.stride(0) * 4),
            C.c_void_p(cam_t.data_ptr()) if cam_t is not None else None, _p(off, C.c_int64), C.c_int32(off.shape[0] - 1),
            C.c_void_p(stream) if stream else None))

    def preprocess(self, xyz: np.ndarray, size_left: int, workspace, cell_size: float = 0.003, dense: bool = False) -> int:
        """NaN removal + workspace box + per-camera voxelisation on the GPU; the result becomes the context's cloud."""
        xyz = np.ascontiguousarray(xyz, np.float32)
        assert xyz.ndim == 2 and xyz.shape[1] >= 3
        ws = np.ascontiguousarray(workspace, np.float64)
        assert ws.size == 6
        nv = C.c_int64(0)
        self._check(self.lib.agh_preprocess(self._h, _p(xyz, C.c_float), C.c_int64(xyz.shape[1] * 4),
                                            C.c_int64(xyz.shape[0]), C.c_int64(size_left), C.c_int(1 if dense else 0),
                                            _p(ws, C.c_double), C.c_double(cell_size), C.byref(nv)))
        self.n = nv.value
        return nv.value

    def preprocess_torch(self, xyz_t, size_left: int, workspace, cell_size: float = 0.003, dense: bool = False,
                         stream=None) -> int:
        assert xyz_t.is_cuda and xyz_t.is_contiguous()
        ws = np.ascontiguousarray(workspace, np.float64)
        nv = C.c_int64(0)
        self._keep = [xyz_t]
        self._check(self.lib.agh_preprocess_device(
            self._h, C.c_void_p(xyz_t.data_ptr()), C.c_int64(xyz_t.stride(0) * 4), C.c_int64(xyz_t.shape[0]),
            C.c_int64(size_left), C.c_int(1 if dense else 0), _p(ws, C.c_double), C.c_double(cell_size), C.byref(nv),
            C.c_void_p(stream) if stream else None))
        self.n = nv.value
        return nv.value

    def find_handles(self, hands: np.ndarray, min_inliers: int = 3, min_length: float = 0.005):
        """HandleSearch::findHandles on hypothesis records; returns (handles, concatenated inlier indices)."""
        hands = np.ascontiguousarray(hands, HYP_DTYPE)
        H = hands.shape[0]
        out = np.zeros(max(H, 1), HANDLE_DTYPE)
        idx = np.zeros(max(H, 1), np.int32)
        n = C.c_int64(0)
        self._check(self.lib.agh_find_handles(self._h, hands.ctypes.data_as(C.c_void_p), C.c_int64(H), C.c_int32(min_inliers),
                                              C.c_double(min_length), out.ctypes.data_as(C.c_void_p), C.c_int64(out.shape[0]),
                                              _p(idx, C.c_int32), C.c_int64(idx.shape[0]), C.byref(n)))
        out = out[:n.value].copy()
        return out, idx[:int(out["n_inliers"].sum())].copy()

    def localize(self, xyz, size_left: int, workspace, samples=None, n_samples: int = 0, sample_seed: int = 1,
                 classify: bool = True, min_inliers: int = 3, min_length: float = 0.005, cell_size: float = 0.003,
                 dense: bool = False, phase: str = "both"):
        """agh_localize: raw capture -> voxels -> search -> SVM -> handles in one call with one synchronisation
        (grasp_localizer.cpp:95-103).  `samples`: indices into the voxelised cloud, or None: n_samples are drawn on the device.
        Returns a dict: handles, inlier_idx, hands (what the handle search ran on), samples, n_voxels, n_hypotheses."""
        on_device = hasattr(xyz, "is_cuda") and xyz.is_cuda  # a torch CUDA tensor (N, >= 3) float32: agh_localize_device
        if on_device:
            assert xyz.is_contiguous() and xyz.dim() == 2 and xyz.shape[1] >= 3
            xyz_ptr, n_pts, stride_b = C.c_void_p(xyz.data_ptr()), int(xyz.shape[0]), int(xyz.stride(0)) * 4
        else:
            xyz = np.ascontiguousarray(xyz, np.float32)
            assert xyz.ndim == 2 and xyz.shape[1] >= 3
            xyz_ptr, n_pts, stride_b = _p(xyz, C.c_float), xyz.shape[0], xyz.shape[1] * 4
        lp = AghLocalizeParams()
        lp.size_left, lp.dense, lp.classify = size_left, 1 if dense else 0, 1 if classify else 0
        ws = np.ascontiguousarray(workspace, np.float64)
        assert ws.size == 6
        for k in range(6):
            lp.workspace[k] = float(ws[k])
        lp.cell_size = cell_size
        if samples is not None:
            samples = np.ascontiguousarray(samples, np.int32)
            lp.sample_idx = samples.ctypes.data_as(C.POINTER(C.c_int32))
            S = samples.shape[0]
        else:
            lp.sample_idx = None
            S = int(n_samples)
        lp.n_samples, lp.sample_seed, lp.min_inliers, lp.min_length = S, sample_seed, min_inliers, min_length
        bufs = getattr(self, "_loc_bufs", None)
        hcap = max(min(8 * S, 8192), 1)
        if bufs is None or bufs[0].shape[0] < hcap or bufs[3].shape[0] < max(S, 1):
            bufs = self._loc_bufs = (np.zeros(hcap, HANDLE_DTYPE), np.zeros(hcap, np.int32), np.zeros(hcap, HYP_DTYPE),
                                     np.zeros(max(S, 1), np.int32))
        handles, idx, hands, sout = bufs
        res = AghLocalizeResult()
        if phase == "begin":  # agh_localize_begin: everything queued; localize_end() collects
            assert not on_device
            self._loc_keep = (xyz, samples, lp)  # (the capture must stay valid until the end call)
            self._loc_S = S
            self._check(self.lib.agh_localize_begin(self._h, xyz_ptr, C.c_int64(stride_b), C.c_int64(n_pts), C.byref(lp)))
            return None
        fn = self.lib.agh_localize_device if on_device else self.lib.agh_localize
        self._check(fn(self._h, xyz_ptr, C.c_int64(stride_b), C.c_int64(n_pts), C.byref(lp), handles.ctypes.data_as(C.c_void_p),
                       C.c_int64(hcap), _p(idx, C.c_int32), C.c_int64(hcap), hands.ctypes.data_as(C.c_void_p), C.c_int64(hcap),
                       _p(sout, C.c_int32), C.byref(res)))
        return self._localize_result(res, S)

    def localize_begin(self, xyz, size_left: int, workspace, **kw):
        """agh_localize_begin: the chain of this capture queued, nothing waited for (see include/agh.h)."""
        return self.localize(xyz, size_left, workspace, phase="begin", **kw)

    def localize_stage(self, xyz):
        """agh_localize_stage: the NEXT capture up on a second stream, beside the chain in flight.  Pass the same array object to
        the next localize_begin (the library recognises the capture by pointer, stride and count)."""
        xyz = np.ascontiguousarray(xyz, np.float32)
        assert xyz.ndim == 2 and xyz.shape[1] >= 3
        self._stage_keep = xyz
        self._check(self.lib.agh_localize_stage(self._h, _p(xyz, C.c_float), C.c_int64(xyz.shape[1] * 4), C.c_int64(xyz.shape[0])))
        return xyz

    def localize_end(self):
        """agh_localize_end: the one synchronisation and the results of the chain localize_begin queued."""
        if getattr(self, "_loc_bufs", None) is None:  # (no begin before: the library says so)
            self._loc_bufs = (np.zeros(1, HANDLE_DTYPE), np.zeros(1, np.int32), np.zeros(1, HYP_DTYPE), np.zeros(1, np.int32))
            self._loc_S = 0
        handles, idx, hands, sout = self._loc_bufs
        hcap = handles.shape[0]
        res = AghLocalizeResult()
        self._check(self.lib.agh_localize_end(self._h, handles.ctypes.data_as(C.c_void_p), C.c_int64(hcap), _p(idx, C.c_int32),
                                              C.c_int64(hcap), hands.ctypes.data_as(C.c_void_p), C.c_int64(hcap),
                                              _p(sout, C.c_int32), C.byref(res)))
        self._loc_keep = None
        return self._localize_result(res, self._loc_S)

    def _localize_result(self, res, S):
        handles, idx, hands, sout = self._loc_bufs
        self.n = res.n_voxels
        self.last_samples = S
        self.last_n = res.n_hypotheses
        return {"handles": handles[:res.n_handles].copy(), "inlier_idx": idx[:res.n_inlier_idx].copy(),
                "hands": hands[:res.n_hands].copy(), "samples": sout[:S].copy(), "n_voxels": int(res.n_voxels),
                "n_hypotheses": int(res.n_hypotheses)}

    def cloud(self):
        xyz = np.zeros((max(self.n, 1), 3), np.float32)
        cam = np.zeros(max(self.n, 1), np.int32)
        k = self._check(self.lib.agh_get_cloud(self._h, _p(xyz, C.c_float), _p(cam, C.c_int32), C.c_int64(self.n)))
        return xyz[:k], cam[:k]

    def find_hands(self, samples: np.ndarray, calculates_antipodal: bool = False) -> np.ndarray:
        samples = np.ascontiguousarray(samples, np.int32)
        cap = max(8 * samples.shape[0], 1)
        out = getattr(self, "_out_buf", None)  # (a fresh 2.5 MB numpy array per call is an mmap / munmap pair and page faults)
        if out is None or out.shape[0] < cap:
            out = self._out_buf = np.zeros(cap, HYP_DTYPE)
        n = C.c_int64(0)
        self._check(self.lib.agh_find_hands(self._h, _p(samples, C.c_int32), C.c_int64(samples.shape[0]),
                                            C.c_int(1 if calculates_antipodal else 0), out.ctypes.data_as(C.c_void_p),
                                            C.c_int64(cap), C.byref(n)))
        self.last_samples = samples.shape[0]
        self.last_n = n.value
        return out[:n.value].copy()

    def frames(self) -> np.ndarray:
        fr = np.zeros(self.last_samples, FRAME_DTYPE)
        n = self._check(self.lib.agh_get_frames(self._h, fr.ctypes.data_as(C.c_void_p), C.c_int64(fr.shape[0])))
        return fr[:n]

    def neighbor_counts(self):
        nt = np.zeros(self.last_samples, np.int32)
        nh = np.zeros(self.last_samples, np.int32)
        self._check(self.lib.agh_get_neighbor_counts(self._h, _p(nt, C.c_int32), _p(nh, C.c_int32),
                                                     C.c_int64(nt.shape[0])))
        return nt, nh

    def images(self) -> np.ndarray:
        im = np.zeros((max(self.last_n, 1), 8000), np.uint8)
        n = self._check(self.lib.agh_get_images(self._h, _p(im, C.c_uint8), C.c_int64(self.last_n)))
        return im[:n]

    def normals(self) -> np.ndarray:
        nr = np.zeros((self.n, 3), np.float64)
        self._check(self.lib.agh_get_normals(self._h, _p(nr, C.c_double), C.c_int64(self.n)))
        return nr

    # ---- training side (learning.cpp:3-163, 249-318) ----
    def set_training_images(self, on: bool = True):
        self._check(self.lib.agh_set_training_images(self._h, C.c_int(1 if on else 0)))

    def training_images(self) -> np.ndarray:
        """(H, 3, 250) packed images of the last find_hands(calculates_antipodal=True): cam = -1, 0, 1."""
        last_n = getattr(self, "last_n", 0)
        im = np.zeros((max(last_n, 1), 3, 250), "<u4")
        n = self._check(self.lib.agh_get_training_images(self._h, _p(im, C.c_uint32), C.c_int64(last_n)))
        return im[:n]

    def hog_images(self, packed: np.ndarray) -> np.ndarray:
        packed = np.ascontiguousarray(packed, "<u4").reshape(-1, 250)
        desc = np.zeros((packed.shape[0], 3528), np.float32)
        self._check(self.lib.agh_hog_images(self._h, _p(packed, C.c_uint32), C.c_int64(packed.shape[0]), _p(desc, C.c_float)))
        return desc

    def train_svm(self, packed: np.ndarray, labels: np.ndarray, C_: float = 1.0, max_iter: int = 1000,
                  eps: float = 1.1920928955078125e-07, kernel: int = SVM_LINEAR) -> dict:
        """convertData's CvSVM::train.  LINEAR: 'w' is the compacted vector; POLY2: 'sv' (n_sv x 3528) and 'alpha'."""
        packed = np.ascontiguousarray(packed, "<u4").reshape(-1, 250)
        lab = np.ascontiguousarray(np.where(np.asarray(labels) > 0, 1, -1), np.int8)
        n = packed.shape[0]
        assert lab.shape[0] == n
        cap = 1 if kernel == SVM_LINEAR else n
        sv = np.zeros((cap, 3528), np.float32)
        alpha = np.zeros(cap, np.float64)
        n_sv = C.c_int32(0)
        rho = C.c_double(0)
        info = np.zeros(6, np.int32)
        self._check(self.lib.agh_train_svm(self._h, _p(packed, C.c_uint32), _p(lab, C.c_int8), C.c_int64(n), C.c_int32(kernel),
                                           C.c_double(C_), C.c_int32(max_iter), C.c_double(eps), _p(sv, C.c_float),
                                           C.c_int64(cap), _p(alpha, C.c_double), C.byref(n_sv), C.byref(rho),
                                           _p(info, C.c_int32)))
        return {"w": sv[0].copy(), "sv": sv[: n_sv.value].copy(), "alpha": alpha[: n_sv.value].copy(), "rho": rho.value,
                "kernel": kernel, "iterations": int(info[0]), "n_sv": int(info[1]), "n_neg": int(info[2]),
                "n_pos": int(info[3]), "rows_computed": int(info[4]), "rows_reused": int(info[5])}

    def load_svm_model(self, kernel: int, sv: np.ndarray, alpha: np.ndarray, rho: float):
        sv = np.ascontiguousarray(sv, np.float32).reshape(-1, 3528)
        alpha = np.ascontiguousarray(alpha, np.float64)
        self._check(self.lib.agh_load_svm_model(self._h, C.c_int32(kernel), _p(sv, C.c_float), C.c_int32(sv.shape[0]),
                                                C.c_int32(3528), _p(alpha, C.c_double), C.c_double(rho)))

    def learning_points(self, hyp: int):
        """(3, n_b) points_for_learning of hypothesis `hyp` and the camera id of each column."""
        n = C.c_int64(0)
        rc = self.lib.agh_get_learning_points(self._h, C.c_int64(hyp), None, None, C.c_int64(0), C.byref(n))
        if rc not in (0, -4):  # AGH_ERR_CAPACITY reports the size
            self._check(rc)
        pts = np.zeros((max(n.value, 1), 3), np.float64)
        cam = np.zeros(max(n.value, 1), np.int32)
        self._check(self.lib.agh_get_learning_points(self._h, C.c_int64(hyp), _p(pts, C.c_double), _p(cam, C.c_int32),
                                                     C.c_int64(n.value), C.byref(n)))
        return pts[: n.value].T.copy(), cam[: n.value].copy()

    def load_svm(self, w: np.ndarray, rho: float):
        w = np.ascontiguousarray(w, np.float32)
        self._check(self.lib.agh_load_svm(self._h, _p(w, C.c_float), C.c_int32(w.size), C.c_double(rho)))

    def load_svm_file(self, path: str):
        self._check(self.lib.agh_load_svm_file(self._h, path.encode()))

    def classify(self) -> np.ndarray:
        keep = np.zeros(max(self.last_n, 1), np.uint8)
        nk = C.c_int64(0)
        self._check(self.lib.agh_classify(self._h, _p(keep, C.c_uint8), C.c_int64(self.last_n), C.byref(nk)))
        return keep[:self.last_n]

    def epoch(self):
        """(stamp, hypothesis count) of the last find_hands call."""
        e, n = C.c_int32(0), C.c_int64(0)
        self._check(self.lib.agh_get_epoch(self._h, C.byref(e), C.byref(n)))
        return e.value, n.value

    def packed_images(self) -> np.ndarray:
        """(H, 250) packed occupancy images of the last find_hands call."""
        im = np.zeros((max(self.last_n, 1), 250), "<u4")
        n = self._check(self.lib.agh_get_packed_images(self._h, _p(im, C.c_uint32), C.c_int64(self.last_n)))
        return im[:n]

    def classify_images(self, packed: np.ndarray):
        """Learning::classify on packed images that belong to no search: (keep, decision values)."""
        packed = np.ascontiguousarray(packed, "<u4").reshape(-1, 250)
        n = packed.shape[0]
        keep = np.zeros(max(n, 1), np.uint8)
        sums = np.zeros(max(n, 1), np.float64)
        self._check(self.lib.agh_classify_images(self._h, _p(packed, C.c_uint32), C.c_int64(n), _p(keep, C.c_uint8),
                                                 _p(sums, C.c_double)))
        return keep[:n], sums[:n]

    # ---- multi-GPU: samples of one cloud sharded over the ranks of a communicator ----
    def comm_init(self, rank: int, n_ranks: int, unique_id: bytes):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self.lib.agh_comm_init(self._h, C.c_int32(rank), C.c_int32(n_ranks), buf))

    def comm_destroy(self):
        self._check(self.lib.agh_comm_destroy(self._h))

    def comm_set_segment_records(self, records: int):
        self._check(self.lib.agh_comm_set_segment_records(self._h, C.c_int64(records)))

    def comm_inject_fault(self, sites: int):
        """agh_comm_inject_fault (testing aid): this rank fails on its own at the named sites of its next sharded call."""
        self._check(self.lib.agh_comm_inject_fault(self._h, C.c_int32(sites)))

    def comm_rank(self):
        r, n = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.agh_comm_rank(self._h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def comm_last_exchange(self):
        """(bytes one rank contributed, ranks, True if RCCL moved them) of the last sharded search's hypothesis all-gather."""
        b, n, v = C.c_int64(0), C.c_int32(0), C.c_int32(0)
        self._check(self.lib.agh_comm_last_exchange(self._h, C.byref(b), C.byref(n), C.byref(v)))
        return b.value, n.value, bool(v.value)

    def find_hands_sharded(self, samples: np.ndarray, calculates_antipodal: bool = False, cap: int | None = None) -> np.ndarray:
        samples = np.ascontiguousarray(samples, np.int32)
        cap = max(8 * samples.shape[0], 1) if cap is None else cap
        out = np.zeros(cap, HYP_DTYPE)
        n = C.c_int64(0)
        self._check(self.lib.agh_find_hands_sharded(self._h, _p(samples, C.c_int32), C.c_int64(samples.shape[0]),
                                                    C.c_int(1 if calculates_antipodal else 0),
                                                    out.ctypes.data_as(C.c_void_p), C.c_int64(cap), C.byref(n)))
        r, g = self.comm_rank()
        lo, hi = shard_slice(samples.shape[0], r, g)
        self.last_samples = hi - lo
        self.shard_n = n.value
        e, mine = self.epoch()
        self.last_n = max(mine, 0)
        return out[:n.value].copy()

    def classify_sharded(self):
        """(records with svm_keep set, keep flags) of the last find_hands_sharded."""
        n = self.shard_n
        out = np.zeros(max(n, 1), HYP_DTYPE)
        keep = np.zeros(max(n, 1), np.uint8)
        nk = C.c_int64(0)
        self._check(self.lib.agh_classify_sharded(self._h, out.ctypes.data_as(C.c_void_p), _p(keep, C.c_uint8), C.c_int64(n),
                                                  C.byref(nk)))
        return out[:n].copy(), keep[:n].copy()

    def find_hands_sharded_torch(self, samples_t, out_t, nout_t, calculates_antipodal: bool = False, stream=None):
        S = samples_t.shape[0]
        cap = out_t.numel() // 160
        r, g = self.comm_rank()
        lo, hi = shard_slice(S, r, g)
        self.last_samples = hi - lo
        self._check(self.lib.agh_find_hands_sharded_device(
            self._h, C.c_void_p(samples_t.data_ptr()), C.c_int64(S), C.c_int(1 if calculates_antipodal else 0),
            C.c_void_p(out_t.data_ptr()), C.c_int64(cap), C.c_void_p(nout_t.data_ptr()),
            C.c_void_p(stream) if stream else None))

    def classify_sharded_torch(self, keep_t=None, stream=None):
        self._check(self.lib.agh_classify_sharded_device(
            self._h, C.c_void_p(keep_t.data_ptr()) if keep_t is not None else None, C.c_void_p(stream) if stream else None))

    def hog(self):
        desc = np.zeros((max(self.last_n, 1), 3528), np.float32)
        sums = np.zeros(max(self.last_n, 1), np.float64)
        n = self._check(self.lib.agh_get_hog(self._h, _p(desc, C.c_float), _p(sums, C.c_double), C.c_int64(self.last_n)))
        return desc[:n], sums[:n]

    def timing(self, counts: bool = False):
        """Summed kernel times [ms] per phase since the previous call; with counts=True also the number of timed launches behind
        each sum (profile level 3 times every fourth call only)."""
        t = AghTiming()
        self._check(self.lib.agh_get_timing(self._h, C.byref(t)))
        ms = {t.name[i].decode(): float(t.ms[i]) for i in range(t.n)}
        if counts:
            cnt = (C.c_int32 * 16)()
            self._check(self.lib.agh_get_timing_counts(self._h, cnt, C.c_int32(16)))
            return ms, {t.name[i].decode(): int(cnt[i]) for i in range(t.n)}
        return ms

    def set_profile(self, level: int):
        self._check(self.lib.agh_set_profile(self._h, C.c_int32(level)))

    def synchronize(self):
        self._check(self.lib.agh_synchronize(self._h))

    def selftest_math(self, n: int = 1 << 20, seed: int = 1) -> int:
        return int(self.lib.agh_selftest_math(self._h, C.c_int64(n), C.c_uint64(seed)))

    # ---- device-resident API (torch tensors) ----
    def set_cloud_torch(self, xyz_t, cam_t, stream=None):
        """xyz_t: float32 CUDA tensor (N, 3) or (N, 8) contiguous; cam_t: int32 CUDA tensor (N,) or None."""
        assert xyz_t.is_cuda and xyz_t.is_contiguous()
        self._keep = [xyz_t, cam_t]
        self.n = xyz_t.shape[0]
        self._check(self.lib.agh_set_cloud_device(
            self._h, C.c_void_p(xyz_t.data_ptr()), C.c_int64(xyz_t.stride(0) * 4),
            C.c_void_p(cam_t.data_ptr()) if cam_t is not None else None, C.c_int64(self.n),
            C.c_void_p(stream) if stream else None))

    def find_hands_torch(self, samples_t, out_t, nout_t, calculates_antipodal: bool = False, stream=None):
        """samples_t int32 CUDA (S,), out_t uint8 CUDA (cap*160,), nout_t int64 CUDA (1,).  Asynchronous."""
        S = samples_t.shape[0]
        cap = out_t.numel() // 160
        self.last_samples = S
        self._check(self.lib.agh_find_hands_device(
            self._h, C.c_void_p(samples_t.data_ptr()), C.c_int64(S), C.c_int(1 if calculates_antipodal else 0),
            C.c_void_p(out_t.data_ptr()), C.c_int64(cap), C.c_void_p(nout_t.data_ptr()),
            C.c_void_p(stream) if stream else None))

    def classify_torch(self, keep_t=None, stream=None):
        self._check(self.lib.agh_classify_device(self._h, C.c_void_p(keep_t.data_ptr()) if keep_t is not None else None,
                                                 C.c_void_p(stream) if stream else None))

"""Synthetic tabletop scenes for parity tests and benchmarks.

The reference ships no PCD files, so the workloads are generated: a table plane plus 14 objects (upright
cylinders, boxes, lying cylinders) sampled at 1.5 mm, kept with probability 0.8 per camera, each camera snapped to
its own 3 mm voxel lattice exactly as ``Localization::voxelizeCloud`` leaves it
(/root/reference/src/agile_grasp/localization.cpp:282-351: ``floor((p - min)/0.003)*0.003 + min``, de-duplicated,
lexicographic voxel order, camera 0 block then camera 1 block).  The result is the cloud *as it enters*
``HandSearch::findHands``: float32 xyz + camera id per point + explicit sample indices.
"""
from __future__ import annotations

import dataclasses

import numpy as np

# camera poses of the reference node (/root/reference/src/nodes/find_grasps.cpp:35-45)
_BASE_TF = np.array(
    [[0, 0.445417, 0.895323, 0.215], [1, 0, 0, -0.015], [0, 0.895323, -0.445417, 0.23], [0, 0, 0, 1]], dtype=np.float64
)
_SQRT_TF = np.array(
    [
        [0.9366, -0.0162, 0.3500, -0.2863],
        [0.0151, 0.9999, 0.0058, 0.0058],
        [-0.3501, -0.0002, 0.9367, 0.0554],
        [0, 0, 0, 1],
    ],
    dtype=np.float64,
)


def camera_poses() -> tuple[np.ndarray, np.ndarray]:
    """cam_tf_left = base_tf * sqrt_tf^-1, cam_tf_right = base_tf * sqrt_tf (find_grasps.cpp:44-45)."""
    return _BASE_TF @ np.linalg.inv(_SQRT_TF), _BASE_TF @ _SQRT_TF


def camera_origins() -> np.ndarray:
    left, right = camera_poses()
    return np.stack([left[:3, 3], right[:3, 3]]).astype(np.float64)


# Two variants of every scene:
#  * AXIS-ALIGNED (name suffix "u", `tilt=False`): SURVEY section 8d to the letter.  The table top and the box faces are
#    parallel to the axes of the frame the cloud is voxelised in -- the `/base` frame of the two-view setup
#    (launch/baxter_grasps.launch:4) -- so every point of a horizontal face lands on ONE lattice level: two thirds of
#    the Taubin neighbourhoods are EXACTLY planar and the generalised eigenproblem of quadric.cpp:143-153 is singular.
#    LAPACK's dggev returns the plane's normal for them all the same, and so does this repository since round 3.
#  * TILTED (the default, the names BASELINE's configs use since round 1): the whole scene (points and camera origins
#    alike) is expressed in a frame rotated by 33 / -19 degrees about the scene pivot before the per-camera voxel snap,
#    as in a camera optical frame (single_camera_grasps.launch); no neighbourhood is exactly planar.
_PIVOT = np.array([0.9, 0.0, -0.1])


def _tilt() -> np.ndarray:
    ax, ay = np.deg2rad(33.0), np.deg2rad(-19.0)
    rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    return ry @ rx


def to_scene_frame(p: np.ndarray, tilt: bool = True) -> np.ndarray:
    if not tilt:
        return np.array(p, dtype=np.float64, copy=True)
    return (p - _PIVOT) @ _tilt().T + _PIVOT


@dataclasses.dataclass
class Scene:
    xyz: np.ndarray  # (N, 3) float32
    cam: np.ndarray  # (N,) int32, 0 = left, 1 = right
    cam_origins: np.ndarray  # (2, 3) float64
    samples: np.ndarray  # (S,) int32 sorted ascending
    seed: int
    name: str = ""

    @property
    def n(self) -> int:
        return int(self.xyz.shape[0])


def _grid2(u0, u1, v0, v1, step):
    u = np.arange(u0, u1, step)
    v = np.arange(v0, v1, step)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    return uu.ravel(), vv.ravel()


def _surface_points(rng: np.random.Generator, table: tuple[float, float, float, float, float], n_objects: int,
                    step: float) -> np.ndarray:
    """Raw surface samples (float64) of the table and the objects, before the per-camera keep/voxelisation."""
    x0, x1, y0, y1, zt = table
    parts = []
    u, v = _grid2(x0, x1, y0, y1, step)
    parts.append(np.stack([u, v, np.full_like(u, zt)], 1))
    # object centres on a jittered grid inside the table, 12 cm margin
    gx = int(np.ceil(np.sqrt(n_objects * (x1 - x0) / max(y1 - y0, 1e-9))))
    gy = int(np.ceil(n_objects / gx))
    cells = [(i, j) for i in range(gx) for j in range(gy)][:n_objects]
    for k, (i, j) in enumerate(cells):
        cx = x0 + 0.12 + (i + 0.5) * (x1 - x0 - 0.24) / gx + rng.uniform(-0.02, 0.02)
        cy = y0 + 0.12 + (j + 0.5) * (y1 - y0 - 0.24) / gy + rng.uniform(-0.02, 0.02)
        kind = k % 3
        if kind == 0:  # upright cylinder r 2-4 cm, h 8-20 cm
            r, h = rng.uniform(0.02, 0.04), rng.uniform(0.08, 0.20)
            a, z = _grid2(0.0, 2 * np.pi * r, 0.0, h, step)
            parts.append(np.stack([cx + r * np.cos(a / r), cy + r * np.sin(a / r), zt + z], 1))
            u, v = _grid2(-r, r, -r, r, step)
            m = u * u + v * v <= r * r
            parts.append(np.stack([cx + u[m], cy + v[m], np.full(m.sum(), zt + h)], 1))
        elif kind == 1:  # box 3-8 x 3-8 x 5-15 cm, rotated about z
            lx, ly, lz = rng.uniform(0.03, 0.08), rng.uniform(0.03, 0.08), rng.uniform(0.05, 0.15)
            th = rng.uniform(0, np.pi)
            c, s = np.cos(th), np.sin(th)
            faces = []
            u, v = _grid2(-lx / 2, lx / 2, -ly / 2, ly / 2, step)
            faces.append(np.stack([u, v, np.full_like(u, lz)], 1))
            u, v = _grid2(-lx / 2, lx / 2, 0, lz, step)
            faces.append(np.stack([u, np.full_like(u, -ly / 2), v], 1))
            faces.append(np.stack([u, np.full_like(u, ly / 2), v], 1))
            u, v = _grid2(-ly / 2, ly / 2, 0, lz, step)
            faces.append(np.stack([np.full_like(u, -lx / 2), u, v], 1))
            faces.append(np.stack([np.full_like(u, lx / 2), u, v], 1))
            f = np.concatenate(faces)
            parts.append(np.stack([cx + c * f[:, 0] - s * f[:, 1], cy + s * f[:, 0] + c * f[:, 1], zt + f[:, 2]], 1))
        else:  # lying cylinder r 1.5-3 cm, L 10-25 cm, axis in the table plane
            r, ln = rng.uniform(0.015, 0.03), rng.uniform(0.10, 0.25)
            th = rng.uniform(0, np.pi)
            c, s = np.cos(th), np.sin(th)
            a, t = _grid2(0.0, 2 * np.pi * r, -ln / 2, ln / 2, step)
            lx_, ly_, lz_ = t, r * np.cos(a / r), r + r * np.sin(a / r)
            parts.append(np.stack([cx + c * lx_ - s * ly_, cy + s * lx_ + c * ly_, zt + lz_], 1))
            u, v = _grid2(-r, r, -r, r, step)
            m = u * u + v * v <= r * r
            for end in (-ln / 2, ln / 2):
                ex, ey, ez = np.full(m.sum(), end), u[m], r + v[m]
                parts.append(np.stack([cx + c * ex - s * ey, cy + s * ex + c * ey, zt + ez], 1))
    return np.concatenate(parts).astype(np.float64)


def _voxelize(pts: np.ndarray, cell: float = 0.003) -> np.ndarray:
    """Per-camera voxel snap in lexicographic voxel order (localization.cpp:282-351)."""
    mn = pts.min(0)
    vox = np.floor((pts - mn) / cell).astype(np.int64)
    vox = np.unique(vox, axis=0)  # lexicographic (ix, iy, iz), like std::set<Vector3i, comparator>
    return vox.astype(np.float64) * cell + mn


def make_scene(n_points: int, n_samples: int, seed: int, two_view: bool = True, n_objects: int = 14,
               name: str = "", tilt: bool = True) -> Scene:
    """Build a scene with exactly ``n_points`` points and ``n_samples`` sorted sample indices."""
    rng = np.random.default_rng(seed)
    views = 2 if two_view else 1
    # voxel density: 1/9e-6 per m^2 per camera on flat surfaces; objects add roughly 0.023 m^2 each
    want_area = 1.08 * n_points * 9e-6 / views
    obj_area = 0.023 * n_objects
    table_area = max(want_area - obj_area, 0.25 * want_area)
    aspect = 1.3
    lx, ly = np.sqrt(table_area * aspect), np.sqrt(table_area / aspect)
    for _ in range(6):
        table = (0.5, 0.5 + lx, -ly / 2, ly / 2, -0.10)
        raw = to_scene_frame(_surface_points(np.random.default_rng(seed), table, n_objects, 0.0015), tilt)
        clouds = []
        crng = np.random.default_rng(seed + 7919)
        for _cam in range(views):
            keep = crng.random(raw.shape[0]) < 0.8
            jit = crng.uniform(-0.0003, 0.0003, size=(int(keep.sum()), 3))
            clouds.append(_voxelize(raw[keep] + jit))
        total = sum(c.shape[0] for c in clouds)
        if total >= n_points:
            break
        grow = np.sqrt(1.1 * n_points / total)
        lx, ly = lx * grow, ly * grow
    else:
        raise RuntimeError("could not reach the requested point count")
    xyz = np.concatenate(clouds).astype(np.float32)
    cam = np.concatenate([np.full(c.shape[0], i, np.int32) for i, c in enumerate(clouds)])
    if total > n_points:  # drop random points, order preserved
        keep_idx = np.sort(rng.permutation(total)[:n_points])
        xyz, cam = xyz[keep_idx], cam[keep_idx]
    samples = np.sort(rng.permutation(n_points)[:n_samples]).astype(np.int32)
    return Scene(np.ascontiguousarray(xyz), np.ascontiguousarray(cam), to_scene_frame(camera_origins(), tilt), samples,
                 seed, name)


# BASELINE.json configs made concrete
def config(name: str) -> Scene:
    """`C2`, `C4`, ... are the tilted scenes; a trailing `u` (`C2u`, `smallu`, ...) selects the axis-aligned variant of the
    same scene (same seed, same objects), `boxu` an axis-aligned table with one 6 x 6 x 9 cm box."""
    if name == "boxu":
        return make_box_scene()
    if name.startswith("seed"):  # `seed123` / `seed123u`: a C2-sized scene from any seed (sweeps over many scenes)
        axis_aligned = name.endswith("u")
        return make_scene(300_000, 2000, seed=int(name[4:-1] if axis_aligned else name[4:]), two_view=True, name=name, tilt=not axis_aligned)
    tilt = True
    base = name
    if name.endswith("u") and name[:-1] in ("C1", "C2", "C3", "C4", "tiny", "small") or (name.startswith("C5") and name.endswith("u")):
        tilt, base = False, name[:-1]
    if base == "C1":
        return make_scene(50_000, 500, seed=1, two_view=False, name=name, tilt=tilt)
    if base in ("C2", "C3"):
        return make_scene(300_000, 2000, seed=2, two_view=True, name=name, tilt=tilt)
    if base == "C4":
        return make_scene(1_000_000, 8000, seed=4, two_view=True, n_objects=48, name=name, tilt=tilt)
    if base.startswith("C5"):
        k = int(base[3:]) if len(base) > 2 else 0
        return make_scene(300_000, 2000, seed=10 + k, two_view=True, name=name, tilt=tilt)
    if base == "tiny":
        return make_scene(12_000, 64, seed=3, two_view=True, n_objects=3, name=name, tilt=tilt)
    if base == "small":
        return make_scene(40_000, 200, seed=5, two_view=True, n_objects=6, name=name, tilt=tilt)
    raise KeyError(name)


def make_box_scene(n_samples: int = 400, seed: int = 21) -> Scene:
    """An ideal axis-aligned 36 x 36 cm table with one axis-aligned 6 x 6 x 9 cm box on it, both cameras, every point
    exactly on the 3 mm lattice (no jitter, no drop-outs): the textbook case of exactly planar neighbourhoods."""
    rng = np.random.default_rng(seed)
    g = np.arange(-60, 61) * 0.003
    u, v = np.meshgrid(g, g, indexing="ij")
    zt = -0.099
    table = np.stack([0.75 + u.ravel(), 0.0 + v.ravel(), np.full(u.size, zt)], 1)
    inside = (np.abs(table[:, 0] - 0.75) < 0.0301) & (np.abs(table[:, 1]) < 0.0301)
    table = table[~inside]
    b = np.arange(-10, 11) * 0.003
    h = np.arange(0, 31) * 0.003
    bu, bv = np.meshgrid(b, b, indexing="ij")
    top = np.stack([0.75 + bu.ravel(), bv.ravel(), np.full(bu.size, zt + 0.09)], 1)
    su, sh = np.meshgrid(b, h[:-1], indexing="ij")
    sides = [np.stack([0.75 + su.ravel(), np.full(su.size, sgn * 0.03), zt + sh.ravel()], 1) for sgn in (-1, 1)]
    sides += [np.stack([0.75 + np.full(su.size, sgn * 0.03), su.ravel(), zt + sh.ravel()], 1) for sgn in (-1, 1)]
    pts = np.unique(np.round(np.concatenate([table, top] + sides) / 0.003).astype(np.int64), axis=0) * 0.003
    clouds = []
    for _cam in range(2):
        keep = rng.random(pts.shape[0]) < 0.8
        clouds.append(pts[keep])
    xyz = np.concatenate(clouds).astype(np.float32)
    cam = np.concatenate([np.full(c.shape[0], i, np.int32) for i, c in enumerate(clouds)])
    samples = np.sort(rng.permutation(xyz.shape[0])[:n_samples]).astype(np.int32)
    return Scene(np.ascontiguousarray(xyz), np.ascontiguousarray(cam), camera_origins(), samples, seed, "boxu")


@dataclasses.dataclass
class RawCloud:
    """A two-camera cloud as it enters Localization::localizeHands (before NaN removal / workspace / voxels)."""
    xyz: np.ndarray  # (N, 3) float32, camera 0 block then camera 1 block, may hold NaN rows
    size_left: int
    workspace: np.ndarray  # (6,) float64 {xmin, xmax, ymin, ymax, zmin, zmax}
    cam_origins: np.ndarray


def make_raw_cloud(n_points: int, seed: int, nan_frac: float = 0.01, n_objects: int = 14, trim: float = 0.03) -> RawCloud:
    """Raw (un-voxelised) two-view capture of the tabletop scene: 1.5 mm surface samples kept with probability 0.8
    per camera and jittered, ``nan_frac`` of the rows replaced by NaN (sensor drop-outs), and a workspace box that
    cuts ``trim`` metres off every side of the scene's bounding box so that the filter has work to do."""
    rng = np.random.default_rng(seed)
    want_area = 0.5 * n_points * (0.0015 ** 2) / 0.8
    table_area = max(want_area - 0.023 * n_objects, 0.25 * want_area)
    lx, ly = np.sqrt(table_area * 1.3), np.sqrt(table_area / 1.3)
    raw = to_scene_frame(_surface_points(np.random.default_rng(seed), (0.5, 0.5 + lx, -ly / 2, ly / 2, -0.10),
                                         n_objects, 0.0015))
    views = []
    for _cam in range(2):
        keep = rng.random(raw.shape[0]) < 0.8
        views.append(raw[keep] + rng.uniform(-0.0003, 0.0003, size=(int(keep.sum()), 3)))
    # trim both views by the same factor to reach n_points in total (order preserved)
    total = views[0].shape[0] + views[1].shape[0]
    if total > n_points:
        f = n_points / total
        views = [v[np.sort(rng.permutation(v.shape[0])[:int(v.shape[0] * f)])] for v in views]
    xyz = np.concatenate(views).astype(np.float32)
    size_left = views[0].shape[0]
    if nan_frac > 0:
        bad = rng.random(xyz.shape[0]) < nan_frac
        xyz[bad] = np.nan
    lo, hi = np.nanmin(xyz, 0).astype(np.float64) + trim, np.nanmax(xyz, 0).astype(np.float64) - trim
    ws = np.array([lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]])
    return RawCloud(np.ascontiguousarray(xyz), int(size_left), ws, to_scene_frame(camera_origins()))

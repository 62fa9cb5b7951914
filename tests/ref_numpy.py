"""Independent numpy transcription of the reference's hand-evaluation logic, used ONLY to cross-check the C++ oracle.

Second implementation written line by line from /root/reference/src/agile_grasp/finger_hand.cpp (3-233),
rotating_hand.cpp (78-177), antipodal.cpp (12-86) and learning.cpp (320-365).  It is self-generated (the reference
pins no numbers), so agreement with the oracle shows two independent readings of the source
coincide; it is not a reference-pinned golden.
"""
from __future__ import annotations

import math

import numpy as np


class FingerHand:
    def __init__(self, finger_width, hand_outer_diameter, hand_depth):
        self.fw, self.od, self.depth = finger_width, hand_outer_diameter, hand_depth
        n = 10
        low, high = 0.0, hand_outer_diameter - finger_width
        step = (high - low) / (n - 1)
        fs_half = np.array([low + i * step for i in range(n)])
        self.fs = np.concatenate([(fs_half - hand_outer_diameter) + finger_width, fs_half])
        self.fingers = np.zeros(2 * n, bool)
        self.hand = np.zeros(n, bool)
        self.back_of_hand = 0.0
        self.pts = None

    def copy(self):
        o = FingerHand.__new__(FingerHand)
        o.__dict__.update({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in self.__dict__.items()})
        return o

    def evaluate_fingers(self, bite):
        self.back_of_hand = -1.0 * (self.depth - bite)
        self.fingers[:] = False
        cropped = []
        for i in range(self.pts.shape[1]):
            if self.pts[1, i] < bite:
                cropped.append(i)
                if self.pts[1, i] < self.back_of_hand:
                    return
        cp = self.pts[:, cropped]
        m = self.fs.size
        for i in range(m):
            num_in_gap = int(np.count_nonzero((cp[0] > self.fs[i]) & (cp[0] < self.fs[i] + self.fw)))
            if num_in_gap == 0:
                if i <= m // 2:
                    s = int(np.count_nonzero(cp[0] > self.fs[i] + self.fw))
                else:
                    s = int(np.count_nonzero(cp[0] < self.fs[i]))
                if s > 0:
                    self.fingers[i] = True

    def evaluate_hand(self):
        n = self.fingers.size // 2
        self.hand = self.fingers[:n] & self.fingers[n:]

    def deepen_hand(self, init_deepness, max_deepness):
        hand_idx = [i for i in range(self.hand.size) if self.hand[i]]
        if not hand_idx:
            return None, 0
        e = hand_idx[int(math.ceil(len(hand_idx) / 2.0)) - 1]
        new_hand = self.copy()
        last = new_hand.copy()
        d = init_deepness + 0.005
        steps = 0
        while d <= max_deepness:
            new_hand.evaluate_fingers(d)
            new_hand.evaluate_hand()
            if not new_hand.hand[e]:
                break
            last = new_hand.copy()
            steps += 1
            d += 0.005
        self.__dict__.update(last.__dict__)
        self.hand = np.zeros(10, bool)
        self.hand[e] = True
        return e, steps

    def grasp_parameters(self, bite):
        fs_sum = 0.0
        for i in range(self.hand.size):
            fs_sum += self.fs[i] * float(self.hand[i])
        hor_pos = (self.od / 2.0) + (fs_sum / int(self.hand.sum()))
        bottom = (hor_pos, self.pts[1].max())
        surface = (hor_pos, self.pts[1].min())
        hand_idx = [i for i in range(self.hand.size) if self.hand[i]]
        e = hand_idx[len(hand_idx) // 2]
        left, right = self.fs[e], self.fs[self.hand.size + e]
        mx, mn = -100000.0, 100000.0
        sel = (self.pts[1] < bite) & (self.pts[0] > left) & (self.pts[0] < right)
        if sel.any():
            mn = min(mn, self.pts[0][sel].min())
            mx = max(mx, self.pts[0][sel].max())
        return bottom, surface, mx - mn


def evaluate_hand(points, normals, cam_ids, frame, cams, geom, init_bite, sample):
    """rotating_hand.cpp:78-177 on already transformed+cropped points (3 x n), normals (3 x n), frame (3x3)."""
    fw, od, depth = geom
    out = []
    angles = [-1.0 * math.pi + k * ((math.pi - (-1.0 * math.pi)) / 8.0) for k in range(8)]
    fh = FingerHand(fw, od, depth)
    for o, ang in enumerate(angles):
        c, s = math.cos(ang), math.sin(ang)
        rot = np.array([[c, -1.0 * s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
        # sequential left-to-right 3-term sums, like the oracle's stated order
        pr = np.empty_like(points)
        nr = np.empty_like(normals)
        for r in range(2):
            pr[r] = rot[r, 0] * points[0] + rot[r, 1] * points[1]
            nr[r] = rot[r, 0] * normals[0] + rot[r, 1] * normals[1]
        pr[2], nr[2] = points[2], normals[2]
        T = np.empty((3, 3))
        for i in range(3):
            for j in range(3):
                T[i, j] = (frame[i, 0] * rot[j, 0] + frame[i, 1] * rot[j, 1]) + frame[i, 2] * rot[j, 2]
        approach = np.array([(T[i, 0] * 0.0 + T[i, 1] * 1.0) + T[i, 2] * 0.0 for i in range(3)])

        def dot(a, b):
            return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]

        if dot(approach, cams[:, 0]) > 0 and dot(approach, cams[:, 1]) > 0:
            continue
        binormal = np.array([(T[i, 0] * 1.0 + T[i, 1] * 0.0) + T[i, 2] * 0.0 for i in range(3)])
        if pr.shape[1] == 0:
            continue
        fh.pts = pr
        fh.evaluate_fingers(init_bite)
        fh.evaluate_hand()
        if fh.hand.sum() > 0:
            e, steps = fh.deepen_hand(init_bite, depth)
            bottom2, surface2, width = fh.grasp_parameters(init_bite)
            surface = np.array([(T[i, 0] * surface2[0] + T[i, 1] * surface2[1]) + T[i, 2] * 0.0 for i in range(3)])
            bottom = np.array([(T[i, 0] * bottom2[0] + T[i, 1] * bottom2[1]) + T[i, 2] * 0.0 for i in range(3)])
            box = pr[1] < fh.back_of_hand + depth
            pib = pr[:, box] - surface[:, None]
            nib = nr[:, box]
            cos_t = math.cos(20 * math.pi / 180.0)
            numl = int(np.count_nonzero(-1.0 * nib[0] > cos_t))
            numr = int(np.count_nonzero(nib[0] > cos_t))
            out.append(dict(orientation=o, approach=approach, binormal=binormal, surface=surface + sample,
                            bottom=bottom + sample, width=width, n_in_box=int(box.sum()), finger_index=e,
                            depth_index=steps, half=(numl > 6 or numr > 6), full=(numl > 6 and numr > 6),
                            points_in_box=pib, cam_in_box=np.asarray(cam_ids)[box]))
    return out


def convert_to_image(pts, binormal, source_to_center):
    """learning.cpp:320-365 (80 rows x 100 cols)."""
    cell = (0.05 - (-0.05)) / 100.0
    if float((binormal[0] * source_to_center[0] + binormal[1] * source_to_center[1])
             + binormal[2] * source_to_center[2]) > 0:
        hc = np.floor((pts[0] - (-0.05)) / cell).astype(np.int64)
    else:
        hc = np.floor((-pts[0] - (-0.05)) / cell).astype(np.int64)
    vc = np.floor((pts[1] - 0.0) / cell).astype(np.int64)
    img = np.zeros((80, 100), np.uint8)
    hc = np.minimum(99, np.maximum(0, hc))
    vc = np.minimum(79, np.maximum(0, vc))
    img[79 - vc, hc] = 255
    return img


def preprocess(xyz, size_left, ws, cell=0.003, dense=False):
    """Second transcription of the head of Localization::localizeHands (localization.cpp:17-45, 216-355): camera id by
    position in the NaN-free cloud, workspace box, per-camera lattice floor((p - min) / cell), unique voxels in
    lexicographic order, coordinates v * cell + min as float32, camera 0 block then camera 1 block."""
    xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
    fin = np.isfinite(xyz).all(1) if not dense else np.ones(len(xyz), bool)
    p = xyz[fin]
    cam = (np.arange(len(p)) >= size_left).astype(np.int32)
    pd = p.astype(np.float64)
    with np.errstate(invalid="ignore"):
        inb = ((pd[:, 0] >= ws[0]) & (pd[:, 0] <= ws[1]) & (pd[:, 1] >= ws[2]) & (pd[:, 1] <= ws[3]) &
               (pd[:, 2] >= ws[4]) & (pd[:, 2] <= ws[5]))
    pd, cam = pd[inb], cam[inb]
    outs, cams = [np.zeros((0, 3), np.float32)], [np.zeros(0, np.int32)]
    for c in range(2):
        q = pd[cam == c]
        if len(q) == 0:
            continue
        mn = np.minimum(q.min(0), 10000.0)
        vx = np.unique(np.floor((q - mn) / cell).astype(np.int64), axis=0)
        outs.append((vx.astype(np.float64) * cell + mn).astype(np.float32))
        cams.append(np.full(len(vx), c, np.int32))
    return np.concatenate(outs), np.concatenate(cams)


def find_handles(hands, min_inliers=3, min_length=0.005):
    """Second transcription of HandleSearch::findHandles + Handle (handle_search.cpp:4-128, handle.cpp:3-74), with the
    oracle's three stated choices (cut keeps the elements before the gap position, ties by index, axis sign).  Returns
    [(inlier index list, axis, center, approach, binormal, hands_center, width)]."""
    H = len(hands)
    width = hands["width"].astype(np.float64).copy()
    res = []
    for i in range(H):
        if width[i] == -1:
            continue
        ia, ip, inn = hands["axis"][i], hands["bottom"][i], hands["approach"][i]
        inl = []
        for j in range(H):
            if width[j] == -1:
                continue
            d = hands["bottom"][j] - ip
            P = np.eye(3) - np.outer(ia, ia)
            v = np.array([(P[r, 0] * d[0] + P[r, 1] * d[1]) + P[r, 2] * d[2] for r in range(3)])
            dist_line = math.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
            along = (ia[0] * d[0] + ia[1] * d[1]) + ia[2] * d[2]
            sa = lambda x: math.acos(min(1.0, max(-1.0, x)))
            aa = (ia[0] * hands["axis"][j][0] + ia[1] * hands["axis"][j][1]) + ia[2] * hands["axis"][j][2]
            nn = (inn[0] * hands["approach"][j][0] + inn[1] * hands["approach"][j][1]) + inn[2] * hands["approach"][j][2]
            if dist_line < 0.01 and min(sa(aa), math.pi - sa(aa)) < 0.34 and sa(nn) < 0.34:
                inl.append((along, j))
        if len(inl) < min_inliers:
            continue
        inl.sort()
        for k in range(len(inl) - 1):
            if inl[k + 1][0] - inl[k][0] > 0.02:
                inl = inl[:k]
                break
        if len(inl) < min_inliers:
            continue
        ds = [a for a, _ in inl]
        if not (max(ds) - min(ds) > min_length):
            continue
        idx = [j for _, j in inl]
        res.append(idx)
        for j in idx:
            width[j] = -1
    return res


# ---- f4: CvSVM::train(C_SVC, LINEAR) + optimize_linear_svm, OpenCV 2.4 svm.cpp, transcribed for small problems -------
def _kernel_row(X, i, poly=False):
    """calc_non_rbf_base: float products, groups of four summed in float, accumulated in double in index order."""
    p = X * X[i][None, :]  # float32
    n, d = X.shape
    g = p[:, : d - d % 4].reshape(n, -1, 4)
    gs = ((g[:, :, 0] + g[:, :, 1]) + g[:, :, 2]) + g[:, :, 3]  # float32
    s = np.cumsum(gs.astype(np.float64), axis=1)[:, -1] if gs.shape[1] else np.zeros(n)
    for k in range(d - d % 4, d):
        s = s + p[:, k].astype(np.float64)
    q = (s * 1.0 + 0.0).astype(np.float32)
    if poly:
        q = q * q  # calc_poly: cvPow(R, R, 2) = multiply(src, src), float
    return np.minimum(q, np.float32(np.finfo(np.float32).max * 1e-3))


def train_svm(features, labels, C=1.0, max_iter=1000, eps=float(np.finfo(np.float32).eps), poly=False):
    X0 = np.ascontiguousarray(features, np.float32)
    lab = np.asarray(labels)
    order = np.concatenate([np.nonzero(lab <= 0)[0], np.nonzero(lab > 0)[0]])  # class 0 (label -1) first
    n0 = int((lab <= 0).sum())
    X = X0[order]
    n = X.shape[0]
    y = np.where(np.arange(n) < n0, 1, -1).astype(np.int8)
    alpha = np.zeros(n)
    G = -np.ones(n)
    status = -np.ones(n, np.int8)
    feps = float(np.finfo(np.float32).eps)
    rows = {}

    def row(i):
        if i not in rows:
            q = _kernel_row(X, i, poly)
            rows[i] = (y.astype(np.float32) * q) if y[i] > 0 else (-y.astype(np.float32) * q)
        return rows[i]

    it = 0
    while True:
        g1, g2, i1, i2 = -np.finfo(np.float64).max, -np.finfo(np.float64).max, -1, -1
        for k in range(n):
            ub, lb = status[k] > 0, status[k] < 0
            if y[k] > 0:
                if not ub and -G[k] > g1:
                    g1, i1 = -G[k], k
                if not lb and G[k] > g2:
                    g2, i2 = G[k], k
            else:
                if not ub and -G[k] > g2:
                    g2, i2 = -G[k], k
                if not lb and G[k] > g1:
                    g1, i1 = G[k], k
        if g1 + g2 < eps:
            break
        it += 1
        if it - 1 >= max_iter:
            break
        i, j = i1, i2
        Qi, Qj = row(i), row(j)
        ai, aj = alpha[i], alpha[j]
        oi, oj = ai, aj
        if y[i] != y[j]:
            denom = float(np.float32(np.float32(Qi[i] + Qj[j]) + np.float32(2) * Qi[j]))
            delta = (-G[i] - G[j]) / max(abs(denom), feps)
            diff = ai - aj
            ai += delta
            aj += delta
            if diff > 0 and aj < 0:
                aj, ai = 0.0, diff
            elif diff <= 0 and ai < 0:
                ai, aj = 0.0, -diff
            if diff > C - C and ai > C:
                ai, aj = C, C - diff
            elif diff <= C - C and aj > C:
                aj, ai = C, C + diff
        else:
            denom = float(np.float32(np.float32(Qi[i] + Qj[j]) - np.float32(2) * Qi[j]))
            delta = (G[i] - G[j]) / max(abs(denom), feps)
            sm = ai + aj
            ai -= delta
            aj += delta
            if sm > C and ai > C:
                ai, aj = C, sm - C
            elif sm <= C and aj < 0:
                aj, ai = 0.0, sm
            if sm > C and aj > C:
                aj, ai = C, sm - C
            elif sm <= C and ai < 0:
                ai, aj = 0.0, sm
        alpha[i], alpha[j] = ai, aj
        for k in (i, j):
            status[k] = 1 if alpha[k] >= C else (-1 if alpha[k] <= 0 else 0)
        G = G + (Qi.astype(np.float64) * (ai - oi) + Qj.astype(np.float64) * (aj - oj))
    yG = y * G
    free = status == 0
    if free.any():
        rho = float(np.cumsum(yG[free])[-1]) / int(free.sum())
    else:
        ubm = (status < 0) & (y > 0) | (status > 0) & (y < 0)
        lbm = ~ubm
        ub = yG[ubm].min() if ubm.any() else np.finfo(np.float64).max
        lb = yG[lbm].max() if lbm.any() else -np.finfo(np.float64).max
        rho = (ub + lb) * 0.5
    a = alpha * y
    v = np.zeros(X.shape[1])
    for k in range(n):
        if abs(a[k]) > 0:
            v = v + X[k].astype(np.float64) * a[k]
    alpha_out = np.zeros(n)
    alpha_out[order] = a
    return dict(w=v.astype(np.float32), rho=rho, iterations=min(it, max_iter), n_sv=int((np.abs(a) > 0).sum()),
                alpha=alpha_out, sv_order=order[np.abs(a) > 0])


# ---------------------------------------------------------------------------------------------------------------------
# An INDEPENDENT HOG (row a18: cv::HOGDescriptor::compute, call site learning.cpp:194-195,220), written from the PUBLISHED
# algorithm (Dalal & Triggs 2005 with OpenCV's documented parameters: 64x64 window, 16x16 blocks at stride 8, 8x8 cells,
# 9 unsigned orientation bins, Gaussian block window sigma = 4, L2-Hys 0.2, sqrt gamma) -- NOT from OpenCV's source layout:
# no PixData tables, no count1/2/4 pixel groups, no per-block accumulation order.  Every vote is spelled out: for each pixel
# of each block, its Gaussian weight, its two orientation bins (linear interpolation), its up-to-four cells (bilinear
# interpolation; cells outside the block get nothing), accumulated in float64 in plain raster order.  The arrays run over
# the IMAGES (axis 0), the loops over windows, blocks and pixels -- so the structure of the sum is the textbook one.
#
# `angle`: "exact" = atan2 of the gradient (the algorithm as published); "opencv24" = the documented polynomial of OpenCV
# 2.4's fastAtan2 ("accuracy about 0.3 degrees"), which cartToPolar uses there -- the one place where the library's
# arithmetic, not the algorithm, decides digits: on a binary image the gradient has eight directions, the axis-aligned
# four are exact in both, the diagonals differ by 1.7e-4 rad, i.e. 4.8e-4 of a bin.
# ---------------------------------------------------------------------------------------------------------------------
def _reflect101(p, n):
    return -p if p < 0 else (2 * n - 2 - p if p >= n else p)


def _fast_atan2_deg(y, x):
    """OpenCV 2.4 fastAtan2 (documented 7th-order odd polynomial on the octant), float32, degrees in [0, 360)."""
    f = np.float32
    sc = f(180.0 / np.pi)
    p1, p3, p5, p7 = (f(0.9997878412794807) * sc, f(-0.3258083974640975) * sc, f(0.1555786518463281) * sc,
                      f(-0.04432655554792128) * sc)
    ax, ay = np.abs(x), np.abs(y)
    eps = f(2.2204460492503131e-16)
    swap = ax < ay
    num = np.where(swap, ax, ay)
    den = np.where(swap, ay, ax) + eps
    c = (num / den).astype(f)
    c2 = c * c
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    a = np.where(swap, f(90.0) - a, a)
    a = np.where(x < 0, f(180.0) - a, a)
    a = np.where(y < 0, f(360.0) - a, a)
    return a.astype(f)


def hog_bruteforce(images, angle="opencv24"):
    """images: (n, 80, 100) uint8 -> (n, 3528) float32 descriptors, layout window / block (x-major) / cell (x-major) / bin."""
    f = np.float32
    img = np.asarray(images, np.uint8).reshape(-1, 80, 100)
    n, H, W = img.shape
    g = np.sqrt(img.astype(f))  # gamma correction
    # centred differences, reflect-101 at the image border, on the WHOLE image (OpenCV computes the gradient image first)
    xs_p = [_reflect101(x + 1, W) for x in range(W)]
    xs_m = [_reflect101(x - 1, W) for x in range(W)]
    ys_p = [_reflect101(y + 1, H) for y in range(H)]
    ys_m = [_reflect101(y - 1, H) for y in range(H)]
    dx = (g[:, :, xs_p] - g[:, :, xs_m]).astype(f)
    dy = (g[:, ys_p, :] - g[:, ys_m, :]).astype(f)
    mag = np.sqrt(dx * dx + dy * dy).astype(f)
    if angle == "exact":
        th = np.arctan2(dy.astype(np.float64), dx.astype(np.float64))
        th = np.where(th < 0, th + 2 * np.pi, th)
        a = th * (9.0 / np.pi) - 0.5
        h0 = np.floor(a)
        w1 = (a - h0)
    else:
        deg = _fast_atan2_deg(dy, dx)
        a = (deg * f(np.pi / 180)).astype(f) * f(9 / np.pi) - f(0.5)
        h0 = np.floor(a)
        w1 = (a - h0.astype(f)).astype(f)
    b0 = np.mod(h0.astype(np.int64), 9)
    b1 = np.mod(b0 + 1, 9)
    v0 = (mag * (f(1) - w1.astype(f))).astype(f)  # the two votes of a pixel, before the spatial weights
    v1 = (mag * w1.astype(f)).astype(f)
    out = np.zeros((n, 2, 7, 7, 2, 2, 9), np.float64)  # window, block x, block y, cell x, cell y, bin
    rows = np.arange(n)
    for win in range(2):
        wx = 32 * win
        for bx in range(7):
            for by in range(7):
                hist = out[:, win, bx, by]
                for i in range(16):  # row inside the block
                    for j in range(16):  # column inside the block
                        gw = f(np.exp(f(-((i - 8.0) ** 2 + (j - 8.0) ** 2) / (2.0 * 4.0 * 4.0))))
                        cxf = (j + 0.5) / 8.0 - 0.5
                        cyf = (i + 0.5) / 8.0 - 0.5
                        x0, y0 = int(np.floor(cxf)), int(np.floor(cyf))
                        fx, fy = f(cxf - x0), f(cyf - y0)
                        py, px = by * 8 + i, wx + bx * 8 + j
                        for cx, wxc in ((x0, f(1) - fx), (x0 + 1, fx)):
                            if cx < 0 or cx > 1:
                                continue
                            for cy, wyc in ((y0, f(1) - fy), (y0 + 1, fy)):
                                if cy < 0 or cy > 1:
                                    continue
                                wgt = f(gw * f(wxc * wyc))
                                np.add.at(hist[:, cx, cy], (rows, b0[:, py, px]), (v0[:, py, px] * wgt).astype(f))
                                np.add.at(hist[:, cx, cy], (rows, b1[:, py, px]), (v1[:, py, px] * wgt).astype(f))
    blk = out.reshape(n, 2 * 49, 36).astype(f)
    # L2-Hys: normalise, clip at 0.2, normalise again
    s = (1.0 / (np.sqrt((blk.astype(np.float64) ** 2).sum(-1)) + 36 * 0.1)).astype(f)
    blk = np.minimum(blk * s[..., None], f(0.2))
    s = (1.0 / (np.sqrt((blk.astype(np.float64) ** 2).sum(-1)) + 1e-3)).astype(f)
    return (blk * s[..., None]).astype(f).reshape(n, 3528)

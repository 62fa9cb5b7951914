"""f4, the training side (Learning::train / trainBalanced / convertData, learning.cpp:3-163, 249-318).

CPU part: the oracle's per-camera instance images against a numpy transcription; the oracle's restatement of OpenCV's
C_SVC solver against an independent numpy transcription (bit for bit) and against the KKT conditions of the dual; the
model writer against the reference's shipped model file (byte for byte).  GPU part: images, descriptors and the
trained model from the HIP path against the oracle, bit for bit.
"""
import os

import numpy as np
import pytest

from oracle import oracle_py as O
from tests import ref_numpy as R
from tests.test_oracle import _transcription_inputs

GOLD = os.path.join(os.path.dirname(__file__), "golden")
MODEL = os.path.join(GOLD, "svm_032015_linear_20_20_same")


def _toy_problem(n, seed, d=3528, overlap=0.3):
    rng = np.random.default_rng(seed)
    X = (np.abs(rng.normal(size=(n, d))) * 0.05).astype(np.float32)
    X[:, rng.random(d) < 0.3] = 0.0  # HOG descriptors hold exact zeros
    wt = rng.normal(size=d).astype(np.float32)
    sc = X @ wt
    y = np.where(sc + overlap * sc.std() * rng.normal(size=n) > np.median(sc), 1, -1)
    return X, y


def test_writer_reproduces_the_shipped_model_byte_for_byte(tmp_path):
    from agile_grasp_amd import binding, build

    build.build()
    w, rho = O.load_svm(MODEL)
    gold = open(MODEL, "rb").read()
    O.save_svm(str(tmp_path / "o.yaml"), w, rho)
    assert open(tmp_path / "o.yaml", "rb").read() == gold
    binding.save_svm_file(str(tmp_path / "p.yaml"), w, rho)  # the product's writer (host code: needs no device)
    assert open(tmp_path / "p.yaml", "rb").read() == gold
    with pytest.raises(binding.AghError):
        binding.save_svm_file(str(tmp_path / "no_such_dir" / "p.yaml"), w, rho)


def test_oracle_camera_images_match_numpy_transcription(tiny_scene):
    sc = tiny_scene
    p = O.default_params(sc.cam_origins)
    res = O.find_hands_training(p, sc.xyz, sc.cam, sc.samples)
    plain = O.find_hands(p, sc.xyz, sc.cam, sc.samples, calculates_antipodal=True, want_images=True)
    hyps, images = res["hyps"], res["images"]
    assert np.array_equal(hyps, plain["hyps"]) and np.array_equal(images[:, 0], plain["images"])
    assert np.array_equal(images[:, 0], images[:, 1] | images[:, 2])  # ins.pts(cam = -1) is the union
    assert (images[:, 1] != images[:, 2]).any() and images[:, 1].any() and images[:, 2].any()
    fr = plain["frames"]
    checked = 0
    for si in range(0, sc.samples.size, 3):
        pts, cam_ids, frame = _transcription_inputs(sc, p, fr, si)
        sample = fr["sample"][si]
        cams = (sc.cam_origins - sample[None, :]).T
        got = R.evaluate_hand(pts, np.zeros_like(pts), cam_ids, frame, cams, (0.01, 0.09, 0.06), 0.01, sample)
        for g in got:
            k = np.nonzero((hyps["sample"] == si) & (hyps["orientation"] == g["orientation"]))[0][0]
            s2c = hyps["surface"][k] - sc.cam_origins[hyps["cam_source"][k]]
            for c in (0, 1):  # createInstance(h, cam_pos, c): the camera's subset, the hand's own source_to_center
                sub = g["points_in_box"][:, g["cam_in_box"] == c]
                assert np.array_equal(R.convert_to_image(sub, g["binormal"], s2c).reshape(-1), images[k, 1 + c])
                checked += 1
    assert checked > 10


@pytest.mark.parametrize("n,seed,max_iter,kernel", [(40, 1, 1000, 0), (90, 2, 60, 0), (64, 3, 1000, 0), (64, 4, 1000, 1),
                                                     (90, 5, 45, 1)])
def test_oracle_solver_matches_numpy_transcription(n, seed, max_iter, kernel):
    X, y = _toy_problem(n, seed)
    a = O.train_svm(X, y, max_iter=max_iter, kernel=kernel)
    b = R.train_svm(X, y, max_iter=max_iter, poly=bool(kernel))
    assert a["iterations"] == b["iterations"] and a["n_sv"] == b["n_sv"]
    assert np.array_equal(a["alpha"], b["alpha"]) and np.array_equal(a["sv_order"], b["sv_order"])
    assert a["rho"] == b["rho"]
    if kernel == 0:
        assert np.array_equal(a["w"], b["w"])


def test_model_files_round_trip_both_shapes(tmp_path):
    from agile_grasp_amd import binding, build

    build.build()
    X, y = _toy_problem(120, 8)
    for kernel in (0, 1):
        r = O.train_svm(X, y, kernel=kernel)
        k, sv, alpha, rho = r["model"]
        assert sv.shape[0] == (1 if kernel == 0 else r["n_sv"])
        po, pp = str(tmp_path / f"o{kernel}.yaml"), str(tmp_path / f"p{kernel}.yaml")
        O.save_svm_model(po, r["model"])
        binding.save_svm_file(pp, sv, rho, kernel=kernel, alpha=alpha)
        assert open(po, "rb").read() == open(pp, "rb").read()  # the two writers agree byte for byte
        k2, sv2, alpha2, rho2 = O.load_svm_model(pp)
        assert k2 == kernel and np.array_equal(sv2, sv) and np.array_equal(alpha2, alpha) and rho2 == rho
    # the quadratic model separates what it was trained on (sum > 0 <=> label -1)
    r = O.train_svm(X, y, kernel=1, max_iter=100000, eps=1e-3)
    _, sv, alpha, rho = r["model"]
    dec = ((X.astype(np.float64) @ sv.astype(np.float64).T) ** 2) @ alpha - rho
    assert (np.where(dec > 0, -1, 1) == y).mean() > 0.95


def test_oracle_solver_satisfies_kkt_when_run_to_convergence():
    X, y = _toy_problem(400, 11, overlap=0.6)
    r = O.train_svm(X, y, max_iter=200000, eps=1e-4)
    assert r["iterations"] < 200000
    a = r["alpha"]  # signed: alpha * y_solver, y_solver = +1 for label -1
    ys = np.where(y > 0, -1.0, 1.0)
    assert np.all(a * ys >= 0) and np.all(np.abs(a) <= 1.0) and abs(a.sum()) < 1e-9  # box + equality constraint
    dec = X.astype(np.float64) @ (X.astype(np.float64).T @ a) - r["rho"]
    m = ys * dec  # functional margin
    tol = 2e-3
    free = (np.abs(a) > 0) & (np.abs(a) < 1.0)
    assert np.all(np.abs(m[free] - 1) < tol)
    assert np.all(m[a == 0] > 1 - tol) and np.all(m[np.abs(a) == 1.0] < 1 + tol)
    # optimize_linear_svm: the compacted vector scores like the support-vector sum; sum > 0 <=> label -1
    dec_w = X.astype(np.float64) @ r["w"].astype(np.float64) - r["rho"]
    assert np.abs(dec_w - dec).max() < 1e-4
    keep = np.array([O.lib().orc_svm_keep(O._fp(np.ascontiguousarray(x), O.C.c_float), O._fp(r["w"], O.C.c_float),
                                          O.C.c_int32(3528), O.C.c_double(r["rho"]), None) for x in X])
    assert (np.where(keep == 1, 1, -1) == y).mean() > 0.9


@pytest.mark.parametrize("kernel", [0, 1])
def test_oracle_solver_fixed_point_matches_libsvm(kernel):
    """A real third-party pin of the solver: OpenCV's CvSVMSolver descends from libsvm, and scikit-learn ships libsvm.
    Run to convergence, the restated solver and libsvm's C-SVC must reach the same dual solution: the same support
    vectors, alphas and rho to ~1e-6, decision values to ~2e-6, identical labels.  (What this does NOT pin: the iterate at
    which OpenCV's default criteria -- 1000 steps -- stop; that is the restated pair selection, checked against the numpy
    transcription.)"""
    from sklearn.svm import SVC

    X, y = _toy_problem(400, 11, overlap=0.6)
    r = O.train_svm(X, y, max_iter=400000, eps=1e-6, kernel=kernel)
    assert r["iterations"] < 400000
    kw = dict(kernel="linear") if kernel == 0 else dict(kernel="poly", degree=2, gamma=1.0, coef0=0.0)
    m = SVC(C=1.0, tol=1e-6, shrinking=False, cache_size=500, **kw).fit(X.astype(np.float64), y)
    Xd = X.astype(np.float64)
    K = Xd @ Xd.T if kernel == 0 else (Xd @ Xd.T) ** 2
    dec_orc = K @ r["alpha"] - r["rho"]              # CvSVM::predict: sum > 0 <=> label -1
    dec_lib = m.decision_function(Xd)                # libsvm: > 0 <=> label +1
    assert np.abs(dec_lib + dec_orc).max() < 2e-5
    assert np.array_equal(np.sign(dec_lib), -np.sign(dec_orc))
    al = np.zeros(len(y))
    al[m.support_] = m.dual_coef_[0]                 # y_i alpha_i
    assert np.abs(al + r["alpha"]).max() < 1e-5 and abs(m.intercept_[0] - r["rho"]) < 1e-5
    assert set(np.nonzero(np.abs(r["alpha"]) > 1e-9)[0]) == set(m.support_.tolist())


def test_oracle_solver_default_criteria_and_single_class():
    X, y = _toy_problem(300, 5)
    r = O.train_svm(X, y)  # CvSVMParams defaults: 1000 steps at most
    assert 0 < r["iterations"] <= 1000 and r["n_sv"] > 0
    with pytest.raises(RuntimeError):
        O.train_svm(X, np.ones(300))


# ---- GPU ----------------------------------------------------------------------------------------------------------
def _training_set(sc, ctx):
    from agile_grasp_amd import binding

    ctx.set_training_images(True)
    ctx.set_cloud(sc.xyz, sc.cam)
    hyps = ctx.find_hands(sc.samples, calculates_antipodal=True)
    packed = ctx.training_images()
    return hyps, packed, binding.unpack_images(packed.reshape(-1, 250)).reshape(-1, 3, 8000)


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", ["tiny", "small"])
def test_training_images_and_descriptors_bit_exact(scene_name):
    from agile_grasp_amd import binding, synthetic
    from tests.test_gpu_parity import assert_hyps_equal

    sc = synthetic.config(scene_name)
    ctx = binding.Context(sc.cam_origins)
    hyps, packed, images = _training_set(sc, ctx)
    ref = O.find_hands_training(O.default_params(sc.cam_origins), sc.xyz, sc.cam, sc.samples)
    assert_hyps_equal(hyps, ref["hyps"])
    assert np.array_equal(images, ref["images"])
    assert np.array_equal(ctx.images(), ref["images"][:, 0])  # the prediction path's image is unchanged
    desc = ctx.hog_images(packed.reshape(-1, 250))
    assert np.array_equal(desc, O.hog_many(ref["images"].reshape(-1, 8000)))
    # switching the mode off restores the plain sweep
    ctx.set_training_images(False)
    again = ctx.find_hands(sc.samples, calculates_antipodal=True)
    assert_hyps_equal(again, ref["hyps"])
    with pytest.raises(binding.AghError):
        ctx.training_images()


@pytest.mark.gpu
@pytest.mark.parametrize("max_iter", [1000, 41])
def test_quadratic_model_trained_saved_loaded_classified_bit_exact(small_scene, tmp_path, max_iter):
    """Learning::train* as the reference calls convertData: uses_linear_kernel = false (learning.h:180-182)."""
    from agile_grasp_amd import binding

    sc = small_scene
    ctx = binding.Context(sc.cam_origins)
    hyps, packed, images = _training_set(sc, ctx)
    use = (hyps["half_antipodal"] == 0) | (hyps["full_antipodal"] == 1)
    labels = np.repeat(hyps["full_antipodal"][use].astype(np.int8), 3)
    got = ctx.train_svm(packed[use].reshape(-1, 250), labels, max_iter=max_iter, kernel=binding.SVM_POLY2)
    feats = O.hog_many(images[use].reshape(-1, 8000))
    ref = O.train_svm(feats, labels, max_iter=max_iter, kernel=1)
    _, rsv, ralpha, rrho = ref["model"]
    assert got["iterations"] == ref["iterations"] and got["n_sv"] == ref["n_sv"] == len(got["alpha"])
    assert got["rho"] == rrho and np.array_equal(got["alpha"], ralpha) and np.array_equal(got["sv"], rsv)
    path = str(tmp_path / "poly.yaml")
    binding.save_svm_file(path, got["sv"], got["rho"], kernel=binding.SVM_POLY2, alpha=got["alpha"])
    model = O.load_svm_model(path)
    assert model[0] == 1 and np.array_equal(model[1], rsv) and np.array_equal(model[2], ralpha) and model[3] == rrho
    ctx.load_svm_file(path)
    keep = ctx.classify()
    desc, sums = ctx.hog()
    okeep, osums = O.classify_model(images[:, 0], model)
    assert np.array_equal(desc, feats.reshape(-1, 3, 3528)[:, 0]) if use.all() else True
    assert np.array_equal(sums, osums) and np.array_equal(keep, okeep)
    assert 0 < keep.sum() < keep.size
    # an uncompacted LINEAR model (several support vectors) goes through the same general path
    lin = O.train_svm(feats, labels, max_iter=max_iter, kernel=0)
    order = lin["sv_order"]
    ctx.load_svm_model(binding.SVM_LINEAR, feats[order], lin["alpha"][order], lin["rho"])
    keep2 = ctx.classify()
    okeep2, _ = O.classify_model(images[:, 0], (0, feats[order], lin["alpha"][order], lin["rho"]))
    assert np.array_equal(keep2, okeep2)
    # and loading the compacted vector switches back to the fused path
    ctx.load_svm(lin["w"], lin["rho"])
    okeep3, _ = O.classify(images[:, 0], lin["w"], lin["rho"])
    assert np.array_equal(ctx.classify(), okeep3)


@pytest.mark.gpu
@pytest.mark.parametrize("max_iter", [1000, 37])
def test_trained_model_bit_exact(small_scene, tmp_path, max_iter):
    from agile_grasp_amd import binding

    sc = small_scene
    ctx = binding.Context(sc.cam_origins)
    hyps, packed, images = _training_set(sc, ctx)
    # Learning::train(hands_list, file, cam_pos): every hand that is not merely half-antipodal, three instances each
    use = (hyps["half_antipodal"] == 0) | (hyps["full_antipodal"] == 1)
    labels = np.repeat(hyps["full_antipodal"][use].astype(np.int8), 3)
    assert 0 < labels.sum() < labels.size
    inst = packed[use].reshape(-1, 250)
    got = ctx.train_svm(inst, labels, max_iter=max_iter)
    feats = O.hog_many(images[use].reshape(-1, 8000))
    ref = O.train_svm(feats, labels, max_iter=max_iter)
    assert got["iterations"] == ref["iterations"] and got["n_sv"] == ref["n_sv"]
    assert got["n_pos"] == int(labels.sum()) and got["n_neg"] == int(labels.size - labels.sum())
    assert got["rho"] == ref["rho"]
    assert np.array_equal(got["w"], ref["w"])
    # CvSVM::save -> CvSVM::load -> Learning::classify round trip
    path = str(tmp_path / "model.yaml")
    binding.save_svm_file(path, got["w"], got["rho"])
    w2, rho2 = O.load_svm(path)
    assert np.array_equal(w2, got["w"]) and rho2 == got["rho"]
    ctx.load_svm_file(path)
    keep = ctx.classify()
    okeep, _ = O.classify(images[:, 0], got["w"], got["rho"])
    assert np.array_equal(keep, okeep)


@pytest.mark.gpu
def test_train_svm_argument_errors(tiny_scene):
    from agile_grasp_amd import binding

    ctx = binding.Context(tiny_scene.cam_origins)
    im = np.zeros((6, 250), "<u4")
    with pytest.raises(binding.AghError):
        ctx.train_svm(im, np.ones(6))  # one class
    with pytest.raises(binding.AghError):
        ctx.training_images()  # nothing searched yet


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 3, 63, 64, 65, 257, 600])
def test_solver_sizes_around_the_tile_boundaries(tiny_scene, n):
    """Instance counts below, at and above the 64-instance tile and the 256-thread work-group, random images."""
    from agile_grasp_amd import binding

    ctx = binding.Context(tiny_scene.cam_origins)
    rng = np.random.default_rng(n)
    images = np.zeros((n, 80, 100), np.uint8)
    for im in images:  # a few random strokes: descriptors with many exact zeros, like real grasp images
        for _ in range(int(rng.integers(1, 6))):
            r, c = int(rng.integers(0, 80)), int(rng.integers(0, 100))
            im[r:r + int(rng.integers(1, 30)), c:c + int(rng.integers(1, 4))] = 255
    packed = binding.pack_images(images.reshape(n, 8000))
    labels = np.where(rng.random(n) < 0.4, 1, -1)
    labels[0], labels[1] = 1, -1
    feats = O.hog_many(images.reshape(n, 8000))
    assert np.array_equal(ctx.hog_images(packed), feats)
    for kernel in (binding.SVM_LINEAR, binding.SVM_POLY2):
        for max_iter in (0, 1000):
            got = ctx.train_svm(packed, labels, kernel=kernel, max_iter=max_iter)
            ref = O.train_svm(feats, labels, kernel=kernel, max_iter=max_iter)
            assert got["iterations"] == ref["iterations"] and got["n_sv"] == ref["n_sv"] and got["rho"] == ref["rho"]
            if kernel == binding.SVM_LINEAR:
                assert np.array_equal(got["w"], ref["w"])
            else:
                assert np.array_equal(got["sv"], ref["model"][1]) and np.array_equal(got["alpha"], ref["model"][2])


@pytest.mark.gpu
@pytest.mark.parametrize("cache_rows", [2, 3, 17])
def test_row_cache_evictions_do_not_change_the_model(small_scene, cache_rows, monkeypatch):
    """A cache that holds only a few kernel rows evicts on nearly every step; the values are the same floats."""
    from agile_grasp_amd import binding

    sc = small_scene
    ctx = binding.Context(sc.cam_origins)
    hyps, packed, images = _training_set(sc, ctx)
    labels = np.repeat(hyps["full_antipodal"].astype(np.int8), 3)
    inst = packed.reshape(-1, 250)
    full = ctx.train_svm(inst, labels, max_iter=300)
    monkeypatch.setenv("AGH_SVM_CACHE_ROWS", str(cache_rows))
    small = ctx.train_svm(inst, labels, max_iter=300)
    assert small["rows_computed"] > full["rows_computed"]
    assert small["iterations"] == full["iterations"] and small["rho"] == full["rho"] and small["n_sv"] == full["n_sv"]
    assert np.array_equal(small["w"], full["w"])
    poly_full = ctx.train_svm(inst, labels, max_iter=120, kernel=binding.SVM_POLY2)
    monkeypatch.delenv("AGH_SVM_CACHE_ROWS")
    poly_ref = ctx.train_svm(inst, labels, max_iter=120, kernel=binding.SVM_POLY2)
    assert poly_full["rho"] == poly_ref["rho"] and np.array_equal(poly_full["alpha"], poly_ref["alpha"])


@pytest.mark.gpu
def test_model_loader_shapes_and_refusals(tiny_scene, tmp_path, svm_model):
    from agile_grasp_amd import binding

    sc = tiny_scene
    ctx = binding.Context(sc.cam_origins)
    ctx.set_cloud(sc.xyz, sc.cam)
    hyps = ctx.find_hands(sc.samples)
    images = ctx.images()
    w, rho = svm_model
    # a one-support-vector quadratic model, and a linear model with alpha != 1 (not the compacted form)
    for kernel, alpha in ((binding.SVM_POLY2, [1.0]), (binding.SVM_LINEAR, [0.5]), (binding.SVM_POLY2, [-2.5])):
        path = str(tmp_path / f"m{kernel}_{alpha[0]}.yaml")
        binding.save_svm_file(path, w, rho, kernel=kernel, alpha=np.array(alpha))
        ctx.load_svm_file(path)
        keep = ctx.classify()
        _, sums = ctx.hog()
        okeep, osums = O.classify_model(images, O.load_svm_model(path))
        assert np.array_equal(keep, okeep) and np.array_equal(sums, osums)
    # refused: kernels / parameters convertData never writes
    good = open(str(tmp_path / f"m{binding.SVM_POLY2}_1.0.yaml")).read()
    for bad in (good.replace("type:POLY, degree:2.", "type:POLY, degree:3."), good.replace("type:POLY", "type:RBF"),
                good.replace("gamma:1.", "gamma:5.0000000000000000e-01"), good.replace("C_SVC", "NU_SVC"),
                good.replace("alpha: [ 1. ]", "alpha: [ 1., 2. ]")):
        path = str(tmp_path / "bad.yaml")
        open(path, "w").write(bad)
        with pytest.raises(binding.AghError):
            ctx.load_svm_file(path)
        with pytest.raises(RuntimeError):
            O.load_svm_model(path)
    with pytest.raises(binding.AghError):
        ctx.load_svm_file(str(tmp_path / "missing.yaml"))
    # the previous model stays usable after a refused load
    ctx.load_svm(w, rho)
    okeep, _ = O.classify(images, w, rho)
    assert np.array_equal(ctx.classify(), okeep) and len(hyps) == len(okeep)


@pytest.mark.gpu
def test_descriptor_of_dense_images_bit_exact(svm_model):
    """Images far denser than any hand's: stripes whose horizontal (vertical) gradient is non-zero at EVERY pixel, so that every
    block's list of voting pixData entries is full, half-density noise (three quarters of the pixels vote), a blank image and a
    full one.  Descriptor, decision value and label against the oracle, bit for bit."""
    from agile_grasp_amd import binding, synthetic

    rng = np.random.default_rng(11)
    imgs = np.zeros((6, 80, 100), np.uint8)
    imgs[0][:, (np.arange(100) % 4) < 2] = 255                      # 1100 1100 ... : sx != 0 everywhere
    imgs[1][(np.arange(80) % 4) < 2, :] = 255                       # the same in y
    imgs[2] = (rng.random((80, 100)) < 0.5) * 255
    imgs[3] = imgs[0] ^ ((rng.random((80, 100)) < 0.03) * 255).astype(np.uint8)
    imgs[5][:] = 255
    flat = imgs.reshape(6, 8000)
    sc = synthetic.config("tiny")
    ctx = binding.Context(sc.cam_origins)
    packed = binding.pack_images(flat)
    desc = ctx.hog_images(packed)
    assert np.array_equal(desc, O.hog_many(flat))
    assert np.abs(desc[0]).max() > 0 and not desc[4].any()
    ctx.load_svm(*svm_model)
    keep, sums = ctx.classify_images(packed)
    okeep, osums = O.classify(flat, *svm_model)
    assert np.array_equal(keep, np.asarray(okeep, np.uint8)) and np.array_equal(sums, osums)

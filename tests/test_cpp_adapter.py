"""The header-only C++ adapter (include/agile_grasp_amd/) that keeps the reference's HandSearch / GraspHypothesis /
Learning signatures: it must compile with a plain g++ against the C ABI, follow the reference's print-and-return-empty
error convention without a GPU, and on a GPU return exactly what the C ABI returns."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _build(tmp_path):
    from agile_grasp_amd import build

    build.build()
    exe = str(tmp_path / "adapter_test")
    libdir = os.path.join(ROOT, "agile_grasp_amd", "lib")
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "adapter_test.cpp"), "-o", exe, "-L" + libdir, "-lagile_grasp_hip",
           "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def _dump(sc, path):
    with open(path, "wb") as f:
        f.write(struct.pack("<qq", sc.n, sc.samples.size))
        f.write(np.asarray(sc.cam_origins, np.float64).tobytes())
        f.write(sc.xyz.astype(np.float32).tobytes())
        f.write(sc.cam.astype(np.int32).tobytes())
        f.write(sc.samples.astype(np.int32).tobytes())


def test_adapter_compiles_and_fails_loudly_without_gpu(tmp_path, tiny_scene):
    import torch

    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    cloud = str(tmp_path / "cloud.bin")
    _dump(tiny_scene, cloud)
    out = subprocess.run([exe, cloud, os.path.join(GOLD, "svm_032015_linear_20_20_same"), "1"], capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0
    assert "Error: cannot create the MI355X grasp-search context" in out.stdout
    assert "RESULT 0 0" in out.stdout  # empty vectors, like the reference's error paths (localization.cpp:9-15)


@pytest.mark.gpu
@pytest.mark.parametrize("deterministic", [1, 0])
def test_adapter_matches_c_abi(tmp_path, tiny_scene, svm_model, deterministic):
    from agile_grasp_amd import binding

    sc = tiny_scene
    exe = _build(tmp_path)
    cloud = str(tmp_path / "cloud.bin")
    _dump(sc, cloud)
    out = subprocess.run([exe, cloud, os.path.join(GOLD, "svm_032015_linear_20_20_same"), str(deterministic)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.splitlines()
    res = [l for l in lines if l.startswith("RESULT")][0].split()
    ctx = binding.Context(sc.cam_origins,
                          normals_mode=binding.NORMALS_DETERMINISTIC if deterministic else binding.NORMALS_RAND50)
    ctx.set_cloud(sc.xyz, sc.cam)
    hyps = ctx.find_hands(sc.samples)
    ctx.load_svm(*svm_model)
    keep = ctx.classify()
    assert int(res[1]) == len(hyps) and int(res[2]) == int(keep.sum())
    hl = [l.split()[1:] for l in lines if l.startswith("H ")]
    for row, h in zip(hl, hyps):
        vals = [float(v) for v in row[:7]]
        exp = [h["surface"][0], h["surface"][1], h["surface"][2], h["approach"][0], h["axis"][1], h["binormal"][2],
               h["width"]]
        assert vals == [float(e) for e in exp]
        assert int(row[7]) == h["cam_source"] and int(row[8]) == h["n_in_box"]
    kept_idx = [int(l.split()[1]) for l in lines if l.startswith("A ")]
    assert kept_idx == list(np.nonzero(keep)[0])

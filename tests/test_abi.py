"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/agh.h declares, and
fails loudly (never falls back to a CPU path) when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "agh.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(agh_[a-z0-9_]+)\s*\(", hdr)))


def test_library_builds_and_exports_every_declared_symbol():
    from agile_grasp_amd import binding, build

    build.build()
    lib = binding.load_library()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert set(binding.EXPORTS) == set(names)


def test_status_codes_match_the_header():
    """The Python side names two codes (callers of the asynchronous entry points repeat on AGH_ERR_RETRY): they must be the
    header's, and the header's codes must be distinct."""
    from agile_grasp_amd import binding

    hdr = open(os.path.join(ROOT, "include", "agh.h")).read()
    codes = dict((k, int(v)) for k, v in re.findall(r"\b(AGH_(?:OK|ERR_[A-Z_]+))\s*=\s*(-?\d+)", hdr))
    assert codes["AGH_OK"] == 0 and len(set(codes.values())) == len(codes) >= 10
    assert binding.AGH_ERR_RETRY == codes["AGH_ERR_RETRY"]


def test_struct_layouts_match_the_header():
    from agile_grasp_amd import binding

    assert binding.HYP_DTYPE.itemsize == 160
    assert binding.FRAME_DTYPE.itemsize == 200
    assert C.sizeof(binding.AghParams) == 8 * 8 + 6 * 8 + 4 * 4
    from oracle import oracle_py as O

    assert O.HYP_DTYPE == binding.HYP_DTYPE and O.FRAME_DTYPE == binding.FRAME_DTYPE
    assert binding.HANDLE_DTYPE.itemsize == 136 and O.HANDLE_DTYPE == binding.HANDLE_DTYPE


def test_no_cpu_fallback_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from agile_grasp_amd import binding

    with pytest.raises(binding.AghError) as e:
        binding.Context(np.zeros((2, 3)))
    assert e.value.code == -2  # AGH_ERR_NO_DEVICE
    assert "no CPU path" in str(e.value) or "HIP device" in str(e.value)


def test_product_never_references_the_oracle():
    pkg = os.path.join(ROOT, "agile_grasp_amd")
    inc = os.path.join(ROOT, "include")
    for base in (pkg, inc):
        for dp, _dn, fn in os.walk(base):
            for f in fn:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    assert "oracle_py" not in txt and "liboracle" not in txt and "agile_oracle" not in txt, f

"""CPU-only checks of the drop-in boundary: libaleppo.so builds, loads without a GPU, exports every
symbol include/aleppo.h declares, and refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from __graft_entry__ import build, load_package


@pytest.fixture(scope="module")
def pkg():
    return build()


def header_functions():
    txt = open(os.path.join(ROOT, "include", "aleppo.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(aleppo_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(pkg):
    names = header_functions()
    assert len(names) >= 30
    lib = pkg.lib()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/aleppo.h but not exported by libaleppo.so"
    assert sorted(pkg.EXPORTS) == names
    assert lib.aleppo_abi_version() == pkg.ABI_VERSION


def test_config_struct_layout_matches_header(pkg):
    # 13 x int32, 9 x float, uint64 (include/aleppo.h aleppo_config)
    assert C.sizeof(pkg.Config) == 96 and pkg.Config.seed.offset == 88
    assert C.sizeof(pkg.MinibatchMetrics) == 28


def test_fails_loudly_without_a_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.AleppoError, match="no CPU fallback"):
        pkg.Engine(8, 8)
    import numpy as np
    with pytest.raises(pkg.AleppoError, match="no CPU fallback"):
        pkg.gae.gae(np.zeros((1, 3), np.float32), np.ones((1, 3)), np.ones((1, 3)), np.ones(1), np.zeros((1, 3)),
                    np.zeros((1, 3)), np.zeros((1, 3)), 0.99, 0.95)


def test_argument_validation_happens_before_the_device(pkg):
    import numpy as np
    with pytest.raises(pkg.AleppoInvalidArgument, match="Horizon must be greater than 0"):
        pkg.Engine(8, 0)
    with pytest.raises(pkg.AleppoInvalidArgument, match="Total environments must be greater than 0"):
        pkg.Engine(0, 8)
    with pytest.raises(pkg.AleppoInvalidArgument, match="2D except next_values"):
        pkg.gae.gae(np.zeros(3, np.float32), np.ones(3), np.ones(3), np.ones(1), np.zeros(3), np.zeros(3),
                    np.zeros(3), 0.99, 0.95)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ale-libtorch-ppo_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in txt and "oracle_lib" not in txt and "oracle/" not in txt.replace(
                    "``oracle/``", ""), f

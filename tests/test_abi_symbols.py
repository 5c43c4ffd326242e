"""CPU: the C-ABI library builds/loads and exports every symbol include/mimo_hip.h declares; without a
GPU every compute entry point must fail loudly (no silent CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from mimo_amd import _lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mimo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mimo_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mimo_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names
    assert b"gfx950" in lib.mimo_version()


def test_philox_host_mirror_matches_oracle():
    import numpy as np
    from oracle.mimo_oracle import philox_uniforms
    lib = _lib.load()
    rows = np.array([0, 1, 2, 12345, 2**32 + 7, 10**12], dtype=np.uint64)
    for seed, sweep in [(0, 0), (1337, 3), (2**40 + 5, 2**33)]:
        ref = philox_uniforms(seed, rows, sweep)
        got = [lib.mimo_philox_uniform(seed, int(r), sweep) for r in rows]
        assert list(ref) == got


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mimo_amd.engine import HipEngine
    with pytest.raises(_lib.MimoHipError):
        HipEngine(0)
    lib = _lib.load()
    assert lib.mimo_estep(None, None, None, None, 1, 0, None, None) == _lib.E_INVALID
    assert lib.mimo_upload(None, None, 0, 1) == _lib.E_INVALID

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def rel_err(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    if a.size == 0 and b.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


GMM_CASES = ["gmm_c1_d2_k4_dir", "gmm_c2_d16_k16_dir", "gmm_c2_d16_k64_dir", "gmm_c3_d8_k32_stick",
             "gmm_c5_d32_k16_dir", "gmm_tail_d5_k7_stick"]
ILR_CASES = ["ilr_c4_dx8_dy4_k16_stick", "ilr_dx1_dy1_k6_dir"]
GIBBS_CASES = ["gibbs_c1_trace", "gibbs_stick_trace"]
# one full-K fixture per BASELINE config whose K is capped above (C3: K = 256, C4: K = 64, C5: K = 128), generated from the
# reference by `make_golden.py fullk`
# ... and N = 4099 per config: tile tails across many workgroups against REFERENCE output; the
# reference's ILR default shape (examples/ilr/evaluate_sine.py:35: 50 experts over dx = dy = 1) at N = 257 and N = 4099
GMM_FULLK_CASES = ["gmm_c3_d8_k256_stick", "gmm_c5_d32_k128_dir", "gmm_c2_d16_k16_n4099", "gmm_c3_d8_k32_n4099"]
ILR_FULLK_CASES = ["ilr_c4_dx8_dy4_k64_stick", "ilr_c4_dx8_dy4_k16_n4099", "ilr_dx1_dy1_k50_stick", "ilr_dx1_dy1_k50_n4099"]


def gating_of(g, prefix):
    kind = str(g["gating_kind"])
    if kind == "dirichlet":
        return kind, g[prefix + "_alphas"]
    return kind, (g[prefix + "_gammas"], g[prefix + "_deltas"])


def nw_of(g, prefix):
    return tuple(g[f"{prefix}_{k}"] for k in ("mus", "kappas", "psis", "nus"))


def mnw_of(g, prefix):
    return tuple(g[f"{prefix}_{k}"] for k in ("Ms", "Ks", "psis", "nus"))


@pytest.fixture(scope="session")
def _session_engine():
    from mimo_amd.engine import HipEngine
    return HipEngine(0)


@pytest.fixture
def engine(_session_engine):
    """One HIP engine for the whole GPU session (fails loudly if the library / GPU is missing).  The structure hint
    is engine state (a tied model leaves it at 'linear'): every test starts from the full feature map."""
    _session_engine.set_structure('full')
    _session_engine.set_row_offset(0)
    _session_engine.tune("num_cu", 0)              # launch-geometry overrides of a previous test (mimo_tune) are undone
    _session_engine.tune("sorted_range", 0)
    _session_engine.tune("mid_min_d", 0)
    _session_engine.tune("mid_narrow_k", 0)
    _session_engine.tune("mid_labels_min_d", 0)
    _session_engine.tune("mid_labels_narrow_k", 0)
    _session_engine.tune("narrow_big_vi", 0)
    return _session_engine

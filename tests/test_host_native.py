"""CPU: the native batched conjugate update (mimo_amd/csrc/mimo_host.cpp, host-only entry points of
libmimo_hip.so) against the NumPy route of the same classes, which is pinned to the reference by the
golden-vector tests."""
import os

import numpy as np
import pytest
from scipy.special import digamma

from mimo_amd import _lib
from mimo_amd.distributions import composite
from mimo_amd.distributions import StackedNormalWisharts, StackedMatrixNormalWisharts
from mimo_amd.utils.abstraction import Statistics as Stats
from conftest import rel_err


def _both(make, nat):
    """posterior built twice from the same natural parameters: native route, NumPy route."""
    outs = []
    for native in (True, False):
        old = composite.NATIVE_HOST
        composite.NATIVE_HOST = native
        try:
            p = make()
            p.nat_param = Stats([v.copy() for v in nat])
            outs.append(p)
        finally:
            composite.NATIVE_HOST = old
    return outs


def test_digamma():
    lib = _lib.load()
    x = np.concatenate([np.linspace(1e-3, 12., 400), np.logspace(1.1, 8, 100)])
    got = np.array([lib.mimo_host_digamma(float(v)) for v in x])
    assert np.max(np.abs(got - digamma(x)) / np.maximum(1., np.abs(digamma(x)))) < 2e-15


@pytest.mark.parametrize("K,D", [(1, 1), (3, 2), (5, 16), (64, 16), (130, 32), (7, 33)])
def test_normal_wishart_native_equals_numpy(K, D):
    rng = np.random.default_rng(K * 100 + D)
    A = rng.standard_normal((K, D, D))
    kappas = rng.uniform(0.5, 200., K)
    mus = rng.standard_normal((K, D))
    nat = [kappas[:, None] * mus, kappas, A @ A.transpose(0, 2, 1) + D * np.eye(D)
           + kappas[:, None, None] * np.einsum('kd,kl->kdl', mus, mus), rng.uniform(1., 5000., K)]
    nat_p, np_p = _both(lambda: StackedNormalWisharts(K, D), nat)
    assert 'canon' in nat_p._memo and 'hld' in nat_p._memo            # the native route really ran
    for a, b in zip(nat_p.params, np_p.params):
        assert rel_err(a, b) < 1e-12
    for a, b in zip(nat_p.canonical_expected(), np_p.canonical_expected()):
        assert rel_err(a, b) < 1e-12
    for a, b in zip(nat_p.expected_statistics(), np_p.expected_statistics()):
        assert rel_err(a, b) < 1e-12
    assert rel_err(nat_p.log_partition(), np_p.log_partition()) < 1e-12
    assert rel_err(nat_p.entropy(), np_p.entropy()) < 1e-9
    for a, b in zip(nat_p.nat_param, nat):
        assert np.array_equal(a, b)                                    # the assigned block is kept exactly


@pytest.mark.parametrize("K,dy,dc", [(1, 1, 2), (6, 2, 4), (64, 4, 9), (33, 8, 17), (9, 3, 33)])
def test_matrix_normal_wishart_native_equals_numpy(K, dy, dc):
    rng = np.random.default_rng(K + 10 * dy + 100 * dc)
    A = rng.standard_normal((K, dc, dc)); Ks = A @ A.transpose(0, 2, 1) + dc * np.eye(dc)
    Ms = rng.standard_normal((K, dy, dc))
    A = rng.standard_normal((K, dy, dy)); Pinv = A @ A.transpose(0, 2, 1) + dy * np.eye(dy)
    nat = [Ms @ Ks, Ks, Pinv + Ms @ Ks @ Ms.transpose(0, 2, 1), rng.uniform(dy + 2., 3000., K)]
    nat_p, np_p = _both(lambda: StackedMatrixNormalWisharts(K, dc, dy), nat)
    assert 'canon_affine' in nat_p._memo
    for a, b in zip(nat_p.params, np_p.params):
        assert rel_err(a, b) < 1e-11
    for affine in (True, False):
        for a, b in zip(nat_p.canonical_expected(affine), np_p.canonical_expected(affine)):
            assert rel_err(a, b) < 1e-11
    for a, b in zip(nat_p.expected_statistics(), np_p.expected_statistics()):
        assert rel_err(a, b) < 1e-11
    assert rel_err(nat_p.entropy(), np_p.entropy()) < 1e-9


def test_not_positive_definite_takes_the_numpy_route():
    """A non-SPD block makes the native call report failure; the NumPy route then decides (as before)."""
    K, D = 5, 3
    nat = [np.zeros((K, D)), np.ones(K), np.stack(K * [np.eye(D)]), 4. * np.ones(K)]
    nat[2][2] = -np.eye(D)
    p = StackedNormalWisharts(K, D)
    p.nat_param = Stats(nat)
    assert 'canon' not in p._memo
    assert np.allclose(p.psis[2], -np.eye(D)) and np.allclose(p.psis[0], np.eye(D))


def test_host_routines_under_asan_ubsan(tmp_path):
    """mimo_host.cpp (the only native code that runs on the CPU) compiled with AddressSanitizer + UBSan and driven
    over K in {1..130}, D in {1..32}, affine / non-affine experts and a non-SPD block (GPU sanitizers are not
    available on the pool; this is the CPU build the brief asks to sanitise)."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_sanitize")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", "-mavx2", "-mfma", "-pthread", "-I", os.path.join(root, "include"),
           os.path.join(root, "tests", "host_sanitize.cpp"), os.path.join(root, "mimo_amd", "csrc", "mimo_host.cpp"),
           "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "sanitizer run ok" in run.stdout, run.stdout[-1000:] + run.stderr[-3000:]

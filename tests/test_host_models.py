"""CPU: host logic of mimo_amd (canonical forms, conjugate updates, drivers) against the reference's
golden vectors, with the oracle-backed engine double standing in for the GPU (tests/oracle_engine.py)."""
import pytest

from conftest import GMM_CASES, ILR_CASES, GIBBS_CASES
from oracle_engine import OracleEngine
import model_checks as mc


@pytest.mark.parametrize("name", GMM_CASES)
def test_gmm_tables_stats_elbo(name):
    mc.check_gmm_case(name, OracleEngine())


@pytest.mark.parametrize("name", GMM_CASES)
def test_gmm_vi_trace(name):
    mc.check_gmm_vi_trace(name, OracleEngine())


@pytest.mark.parametrize("name", GIBBS_CASES)
def test_gibbs_trace(name):
    mc.check_gibbs_trace(name, OracleEngine())


@pytest.mark.parametrize("name", ILR_CASES)
def test_ilr_tables_stats_elbo(name):
    mc.check_ilr_case(name, OracleEngine())


@pytest.mark.parametrize("name", ILR_CASES)
def test_ilr_vi_trace(name):
    mc.check_ilr_vi_trace(name, OracleEngine())

"""Rows of more than 512 entries against the LITERAL oracle (round-3 review, item 3).

The cuts of 32x32x32_15096_L the literal oracle can walk stop at rows of 288 / 256 entries (tests/test_gpu_l32cut.py); the rows of 512
and 1024 entries of the metric's input reach the flat sweep of plo::cse_big_kernel only through goldens made by the build's own
scalable engine.  tests/golden/longrow_costs.json holds (adds, muls) of oracle/plo_oracle.c on a synthetic matrix with rows of 461-627
entries (plo_testlib.longrow_matrix; tests/golden/make_longrow_costs.py, tens of minutes per seed).  The kernel must reproduce them in
all three modes, and with a window of 256 entries (PLO_BIG_FWIN) every long row is swept in several windows of the flat sweep
(reference include/plinopt_optimize.inl:92-146: every entry of a row that holds the pair is visited)."""
import json
import os

import pytest

from plo_testlib import GOLDEN, longrow_matrix

pytestmark = pytest.mark.gpu
P = 131071


@pytest.fixture(scope="module")
def gold():
    G = json.load(open(os.path.join(GOLDEN, "longrow_costs.json")))
    m, n, rp, c, v = longrow_matrix(P)
    assert len(c) == G["nnz"] and [rp[i + 1] - rp[i] for i in range(m)] == G["row_lengths"] and max(G["row_lengths"]) > 512
    return G, (m, n, rp, c, v)


def _run(csr, G):
    from plinopt_amd import CSEPlan
    m, n, rp, c, v = csr
    plan = CSEPlan(m, n, rp, c, v, P)
    assert plan.is_hbm                                        # rows beyond 64 entries: the HBM-resident family
    got = plan.cost_many(seed0=G["seed0"], n=len(G["adds"]))
    cnt = plan.hbm_counters()
    plan.close()
    return got, cnt


def test_long_rows_equal_the_literal_oracle(hip, gold):
    G, csr = gold
    got, cnt = _run(csr, G)
    assert got == (G["adds"], G["muls"])
    assert cnt["extra_sweep_windows"] == 0                    # one row per wave: 627 entries fit the 2048-entry window


def test_long_rows_in_several_windows_of_the_flat_sweep(hip, gold, monkeypatch):
    G, csr = gold
    monkeypatch.setenv("PLO_BIG_FWIN", "256")
    got, cnt = _run(csr, G)
    assert got == (G["adds"], G["muls"])
    assert cnt["extra_sweep_windows"] > 100 * len(G["adds"]), cnt      # every step of a row longer than 256 entries takes 2 or 3 windows


@pytest.mark.parametrize("knob", ["PLO_BIG_NORID", "PLO_BIG_VT_GLOBAL", "PLO_BIG_EAGER"])
def test_long_rows_in_the_other_kernel_modes(hip, gold, monkeypatch, knob):
    """mode 1 (value table in LDS, no ratio identifiers), mode 0 (value table in global memory) and the eager pair table of round 2"""
    G, csr = gold
    monkeypatch.setenv(knob, "1")
    got, _ = _run(csr, G)
    assert got == (G["adds"], G["muls"])
